#include <hip/hip_runtime.h>
#include <stdio.h>
template <int SH> __device__ int row_shr(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x110 + SH, 0xF, 0xF, true); }
__global__ void k(int P, int* out) {
    int lane = threadIdx.x, part = lane & (P - 1);
    int x = lane + 1, incl = x;
    if (P > 1) incl += (part >= 1) ? row_shr<1>(incl) : 0;
    if (P > 2) incl += (part >= 2) ? row_shr<2>(incl) : 0;
    if (P > 4) incl += (part >= 4) ? row_shr<4>(incl) : 0;
    if (P > 8) incl += (part >= 8) ? row_shr<8>(incl) : 0;
    out[lane] = incl;
    out[64 + lane] = row_shr<1>(x);
}
int main() {
    int* d; hipMalloc(&d, 128 * 4); int h[128];
    for (int P : {2, 4, 8, 16}) {
        hipLaunchKernelGGL(k, 1, 64, 0, 0, P, d); hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("P=%d incl:", P); for (int i = 0; i < 20; ++i) printf(" %d", h[i]); printf("\n");
        if (P == 2) { printf("shr1 of (lane+1):"); for (int i = 0; i < 20; ++i) printf(" %d", h[64 + i]); printf("\n"); }
    }
}
