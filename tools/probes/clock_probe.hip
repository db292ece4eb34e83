// What shader clock does the chip hold under (a) a vector-FMA load, (b) an f16 MFMA load, (c) the alternating
// MFMA / vector pattern of the all-pairs tail?  clock64() (s_memtime) against wall_clock64() (s_memrealtime, 100 MHz),
// 1024 workgroups x 256 threads (4 waves per SIMD), a few milliseconds per mode.  Also prints cycles per MFMA per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <int MODE>
__global__ __launch_bounds__(256, 4) void probe(int iters, float* sink, unsigned long long* out) {
    f16x8 a, b;
    for (int q = 0; q < 8; ++q) { a[q] = (_Float16)(0.001f * (threadIdx.x + q)); b[q] = (_Float16)(0.002f * (threadIdx.x * 3 + q)); }
    f32x4 c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
    float v0 = threadIdx.x * 0.5f, v1 = 1.f, v2 = 2.f, v3 = 3.f, v4 = 4.f, v5 = 5.f, v6 = 6.f, v7 = 7.f;
    const unsigned long long t0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
        if (MODE & 1) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {   // 32 independent-ish FMAs
                v0 = fmaf(v0, 1.0001f, 0.5f); v1 = fmaf(v1, 1.0001f, 0.5f); v2 = fmaf(v2, 1.0001f, 0.5f); v3 = fmaf(v3, 1.0001f, 0.5f);
                v4 = fmaf(v4, 1.0001f, 0.5f); v5 = fmaf(v5, 1.0001f, 0.5f); v6 = fmaf(v6, 1.0001f, 0.5f); v7 = fmaf(v7, 1.0001f, 0.5f);
            }
        }
        if (MODE & 2) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(b, a, c1, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(b, a, c1, 0, 0, 0);
        }
        asm volatile("" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(c0), "+v"(c1));
    }
    const unsigned long long t1 = clock64(), w1 = wall_clock64();
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = t1 - t0; out[2 * blockIdx.x + 1] = w1 - w0; }
    sink[blockIdx.x * 256 + threadIdx.x] = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7 + c0[0] + c1[1];
}
template <int MODE>
static void run(const char* name, int iters, float* sink, unsigned long long* out) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(probe<MODE>, dim3(1024), dim3(256), 0, 0, iters, sink, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[2048]; hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
        double cyc = 0, wall = 0; for (int i = 0; i < 1024; ++i) { cyc += h[2 * i]; wall += h[2 * i + 1]; }
        cyc /= 1024; wall /= 1024;
        printf("%-10s rep %d: %.3f ms  clock64 %.0f  wall(100MHz) %.0f  -> clock64 rate %.0f MHz", name, rep, ms, cyc, wall, cyc / wall * 100.0);
        if (MODE & 2) printf("  | per SIMD: %.1f clock64 ticks, %.2f ns per MFMA", cyc / (iters * 16.0), wall * 10.0 / (iters * 16.0));
        if (MODE & 1) printf("  | per SIMD: %.2f ticks, %.3f ns per FMA", cyc / (iters * 128.0), wall * 10.0 / (iters * 128.0));
        printf("\n");
    }
}
int main() {
    float* sink; unsigned long long* out;
    hipMalloc(&sink, 1024 * 256 * 4); hipMalloc(&out, 2048 * 8);
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(probe<1>, dim3(1024), dim3(256), 0, 0, 20000, sink, out);   // ~0.5 s of warm-up
    hipDeviceSynchronize();
    run<1>("valu", 40000, sink, out);
    run<2>("mfma", 40000, sink, out);
    run<3>("both", 40000, sink, out);
    run<1>("valu", 40000, sink, out);
    return 0;
}
