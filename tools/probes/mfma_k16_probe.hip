// What does the legacy v_mfma_f32_16x16x16_f16 (K = 16) cost on gfx950 beside v_mfma_f32_16x16x32_f16 (K = 32)?
// The all-pairs tail's fifth matrix instruction (W1lo . Hhi) only needs K = 16.  1024 workgroups x 256 threads (4 waves per
// SIMD), independent accumulators, HIP events; prints ns per instruction per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(256, 4) void probe(int iters, float* sink) {
    f16x8 a, b;
    for (int q = 0; q < 8; ++q) { a[q] = (_Float16)(0.001f * (threadIdx.x + q)); b[q] = (_Float16)(0.002f * (threadIdx.x * 3 + q)); }
    f16x4 a4 = {a[0], a[1], a[2], a[3]}, b4 = {b[0], b[1], b[2], b[3]};
    f32x4 c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0}, c2 = {0, 0, 0, 0}, c3 = {0, 0, 0, 0}, c4 = {0, 0, 0, 0};
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {          // five K = 32
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(b, a, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(b, a, c3, 0, 0, 0);
            c4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, a, c4, 0, 0, 0);
        } else if (MODE == 1) {   // five K = 16
            c0 = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x16f16(b4, a4, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_16x16x16f16(b4, a4, c3, 0, 0, 0);
            c4 = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, a4, c4, 0, 0, 0);
        } else {                  // four K = 32 + one K = 16 (the tail's mix)
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(b, a, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(b, a, c3, 0, 0, 0);
            c4 = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, c4, 0, 0, 0);
        }
        asm volatile("" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4));
    }
    sink[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3] + c4[0];
}
template <int MODE>
static void run(const char* name, int iters, float* sink) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(probe<MODE>, dim3(1024), dim3(256), 0, 0, iters, sink);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-28s rep %d: %.3f ms -> %.2f ns per 5 instructions per SIMD (%.2f each)\n", name, rep, ms,
               ms * 1e6 / (iters * 4.0), ms * 1e6 / (iters * 20.0));
    }
}
int main() {
    float* sink;
    hipMalloc(&sink, 1024 * 256 * 4);
    for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(probe<0>, dim3(1024), dim3(256), 0, 0, 20000, sink);   // warm-up
    hipDeviceSynchronize();
    run<0>("5 x 16x16x32 f16", 40000, sink);
    run<1>("5 x 16x16x16 f16", 40000, sink);
    run<2>("4 x K=32 + 1 x K=16", 40000, sink);
    run<0>("5 x 16x16x32 f16", 40000, sink);
    return 0;
}
