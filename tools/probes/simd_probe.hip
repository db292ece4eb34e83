// Which SIMD does wave w of a 512-/1024-thread workgroup land on?  (gfx950 probe)
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void probe(unsigned* out) {
    unsigned hwid = __builtin_amdgcn_s_getreg((4 /*HW_ID*/) | (0 << 6) | (31 << 11));
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = hwid;
}
int main() {
    for (int nt : {512, 1024}) {
        unsigned* d; int nb = 8, nw = nt / 64;
        hipMalloc(&d, nb * nw * 4);
        hipLaunchKernelGGL(probe, dim3(nb), dim3(nt), 100 * 1024, 0, d);
        unsigned h[256]; hipMemcpy(h, d, nb * nw * 4, hipMemcpyDeviceToHost);
        for (int b = 0; b < nb; ++b) { printf("nt=%d blk %d simd:", nt, b); for (int w = 0; w < nw; ++w) printf(" %u", (h[b * nw + w] >> 4) & 3); printf("  cu %u\n", (h[b*nw] >> 8) & 15); }
        hipFree(d);
    }
    return 0;
}
