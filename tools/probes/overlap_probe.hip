// Do bf16 MFMA waves and VALU waves on one SIMD overlap at THROUGHPUT (2 + 2 waves per SIMD), or do their cycles add?
// 1024-thread workgroup, waves 0-7 run a VALU stream, waves 8-15 dependent bf16 MFMA chains; mode bit0 / bit1 enable.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(1024) void probe(int mode, int iters, float* sink) {
    const int wave = threadIdx.x >> 6;
    float acc = threadIdx.x * 0.001f;
    if (wave < 8) {
        if (mode & 1) {
            float a = acc, b = acc + 1.f, c = acc + 2.f, d = acc + 3.f, e = acc + 4.f, f = acc + 5.f, g = acc + 6.f, h = acc + 7.f;
            for (int i = 0; i < iters; ++i) {   // 16 VALU ops per iteration, 8 independent values
                float lo;
                lo = __builtin_amdgcn_fmed3f(a, b, -INFINITY); b = __builtin_amdgcn_fmed3f(a, b, INFINITY); a = lo;
                lo = __builtin_amdgcn_fmed3f(c, d, -INFINITY); d = __builtin_amdgcn_fmed3f(c, d, INFINITY); c = lo;
                lo = __builtin_amdgcn_fmed3f(e, f, -INFINITY); f = __builtin_amdgcn_fmed3f(e, f, INFINITY); e = lo;
                lo = __builtin_amdgcn_fmed3f(g, h, -INFINITY); h = __builtin_amdgcn_fmed3f(g, h, INFINITY); g = lo;
                lo = __builtin_amdgcn_fmed3f(a, c, -INFINITY); c = __builtin_amdgcn_fmed3f(a, c, INFINITY); a = lo;
                lo = __builtin_amdgcn_fmed3f(b, d, -INFINITY); d = __builtin_amdgcn_fmed3f(b, d, INFINITY); b = lo;
                lo = __builtin_amdgcn_fmed3f(e, g, -INFINITY); g = __builtin_amdgcn_fmed3f(e, g, INFINITY); e = lo;
                lo = __builtin_amdgcn_fmed3f(f, h, -INFINITY); h = __builtin_amdgcn_fmed3f(f, h, INFINITY); f = lo;
                asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));
            }
            acc = a + b + c + d + e + f + g + h;
        }
    } else if (mode & 2) {
        f32x4 c0 = {0, 0, 0, 0};
        bf16x8 x, y;
        for (int q = 0; q < 8; ++q) { x[q] = (short)(threadIdx.x + q); y[q] = (short)(threadIdx.x * 3 + q); }
        for (int i = 0; i < iters; ++i) {       // 4 dependent MFMAs per iteration
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, c0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(y, x, c0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, c0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(y, x, c0, 0, 0, 0);
        }
        acc = c0[0];
    }
    if (acc == 12345.678f) sink[0] = acc;
}
int main() {
    float* sink; (void)hipMalloc(&sink, 4);
    const int iters = 20000;
    for (int mode = 1; mode <= 3; ++mode) {
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        probe<<<1, 1024>>>(mode, iters, sink);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        probe<<<1, 1024>>>(mode, iters, sink);
        (void)hipEventRecord(e1);
        (void)hipDeviceSynchronize();
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        const double cyc = ms * 1e6 * 2.39;
        printf("mode %d (VALU %s, MFMA %s): %.0f cycles total = %.1f cycles per iteration;  per SIMD: %s%s\n", mode,
               mode & 1 ? "on" : "off", mode & 2 ? "on" : "off", cyc, cyc / iters,
               mode & 1 ? "32 VALU ops " : "", mode & 2 ? "8 MFMAs" : "");
    }
    return 0;
}
