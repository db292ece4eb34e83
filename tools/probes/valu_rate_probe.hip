// VALU issue rate on gfx950 as a function of waves per SIMD (1, 2, 4) for a few op kinds.
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int KIND>
__global__ __launch_bounds__(1024) void probe(int iters, float* sink, unsigned long long* cyc) {
    float a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = threadIdx.x * 0.001f + i;
    __syncthreads();
    unsigned long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; i += 2) {       // 8 independent pairs, 16 ops
            if (KIND == 0) { float lo = __builtin_amdgcn_fmed3f(a[i], a[i + 1], -INFINITY); a[i + 1] = __builtin_amdgcn_fmed3f(a[i], a[i + 1], INFINITY); a[i] = lo; }
            if (KIND == 1) { a[i] = fmaf(a[i], 1.0001f, a[i + 1]); a[i + 1] = fmaf(a[i + 1], 0.9999f, 0.5f); }
            if (KIND == 2) { int x = __float_as_int(a[i]), y = __float_as_int(a[i + 1]); int lo = min(x, y); y = max(x, y); a[i] = __int_as_float(lo); a[i + 1] = __int_as_float(y); }
            if (KIND == 3) { a[i] = (a[i] < a[i + 1]) ? a[i + 1] : a[i]; a[i + 1] = a[i + 1] + 1.0f; }
        }
        asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]));
        asm volatile("" : "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]));
    }
    unsigned long long t1 = clock64();
    if ((threadIdx.x & 63) == 0) atomicAdd(cyc, t1 - t0);
    float s = 0; for (int i = 0; i < 16; ++i) s += a[i];
    if (s == 12345.678f) sink[0] = s;
}
template <int KIND>
void run(const char* name) {
    float* sink; unsigned long long* cyc;
    hipMalloc(&sink, 4); hipMalloc(&cyc, 8);
    const int iters = 4000, nb = 256;
    for (int nt : {256, 512, 1024}) {
        hipMemset(cyc, 0, 8);
        hipLaunchKernelGGL(probe<KIND>, dim3(nb), dim3(nt), 0, 0, iters, sink, cyc);
        hipDeviceSynchronize();
        unsigned long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        double per_wave_iter = h / (double)(nb * (nt / 64)) / iters;
        printf("%-12s %d waves/SIMD: %.1f cycles per 16-op iteration per wave -> %.2f cycles/op/SIMD\n", name, nt / 256, per_wave_iter,
               per_wave_iter / 16 / (nt / 256));
    }
}
int main() { run<0>("med3 min/max"); run<1>("v_fma_f32"); run<2>("int min/max"); run<3>("cmp+cndmask"); return 0; }
