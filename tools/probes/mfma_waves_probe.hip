// Aggregate issue rate of v_mfma_f32_16x16x32_bf16 per SIMD when 1..4 waves per SIMD each run ONE dependent chain
// (does interleaving chains of different waves cost what interleaving chains inside a wave costs?)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(1024) void probe(int iters, float* sink) {
    f32x4 c = {0, 0, 0, 0};
    bf16x8 x, y;
    for (int q = 0; q < 8; ++q) { x[q] = (short)(threadIdx.x + q); y[q] = (short)(threadIdx.x * 3 + q); }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 12; ++r) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, c, 0, 0, 0);
    }
    if (c[0] == 12345.678f) sink[0] = c[0];
}
int main() {
    float* sink; (void)hipMalloc(&sink, 4);
    const int iters = 20000;
    for (int wps = 1; wps <= 4; ++wps) {
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        probe<<<1, 256 * wps>>>(iters, sink);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        probe<<<1, 256 * wps>>>(iters, sink);
        (void)hipEventRecord(e1);
        (void)hipDeviceSynchronize();
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        const double per_simd = (double)iters * 12 * wps;     // MFMAs issued on each SIMD
        printf("waves/SIMD %d: %.2f ns per MFMA per SIMD  (= %.1f cycles at 2.39 GHz)\n", wps, ms * 1e6 / per_simd, ms * 1e6 / per_simd * 2.39);
    }
    return 0;
}
