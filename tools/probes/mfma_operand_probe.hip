// v_mfma_f32_16x16x32_bf16 issue rate when consecutive instructions read DIFFERENT operand registers
// (three A planes x three B planes, the six-product pattern of the kernels) vs the same pair every time.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
template <int MODE>
__global__ __launch_bounds__(1024) void probe(int iters, float* sink) {
    f32x4 c = {0, 0, 0, 0};
    bf16x8 ah, am, al, bh, bm, bl;
    for (int q = 0; q < 8; ++q) {
        ah[q] = (short)(threadIdx.x + q); am[q] = (short)(threadIdx.x * 5 + q); al[q] = (short)(threadIdx.x * 7 + q);
        bh[q] = (short)(threadIdx.x * 3 + q); bm[q] = (short)(threadIdx.x * 11 + q); bl[q] = (short)(threadIdx.x * 13 + q);
    }
    asm volatile("" : "+v"(ah), "+v"(am), "+v"(al), "+v"(bh), "+v"(bm), "+v"(bl));
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            if (MODE == 0) {
                for (int j = 0; j < 6; ++j) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, c, 0, 0, 0);
            } else {
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, c, 0, 0, 0);
            }
        }
    }
    if (c[0] == 12345.678f) sink[0] = c[0];
}
template <int MODE>
void run(const char* name) {
    float* sink; (void)hipMalloc(&sink, 4);
    const int iters = 20000;
    for (int wps = 1; wps <= 4; wps += 3) {
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        probe<MODE><<<1, 256 * wps>>>(iters, sink);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        probe<MODE><<<1, 256 * wps>>>(iters, sink);
        (void)hipEventRecord(e1);
        (void)hipDeviceSynchronize();
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        const double per_simd = (double)iters * 12 * wps;
        printf("%s, waves/SIMD %d: %.1f cycles per MFMA per SIMD\n", name, wps, ms * 1e6 / per_simd * 2.39);
    }
}
int main() { run<0>("same operands"); run<1>("six-product pattern"); return 0; }
