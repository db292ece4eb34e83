// Do f32 MFMA waves and VALU waves on the same SIMD overlap on gfx950?  512-thread WG:
// waves 0-3 run a VALU chain (role A), waves 4-7 run an MFMA chain (role B).  mode bit0: run A, bit1: run B.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
template <int KIND>
__global__ __launch_bounds__(512) void probe(int mode, int iters, float* sink, unsigned long long* cyc) {
    const int wave = threadIdx.x >> 6;
    float acc = threadIdx.x * 0.001f;
    __syncthreads();
    unsigned long long t0 = clock64();
    if (wave < 4) {
        if (mode & 1) {
            float a = acc, b = acc + 1.f, c = acc + 2.f, d = acc + 3.f, e = acc + 4.f, f = acc + 5.f, g = acc + 6.f, h = acc + 7.f;
            for (int i = 0; i < iters; ++i) {   // 16 independent-ish VALU ops per iteration (min/max int-free float)
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
        if (KIND == 0) {
            f32x4 c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
            float x = acc, y = acc * 0.5f;
            for (int i = 0; i < iters; ++i) {   // 4 MFMAs per iteration, two chains
                c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, x, c1, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, x, c1, 0, 0, 0);
            }
            acc = c0[0] + c1[1];
        } else if (KIND == 2) {   // ONE dependent chain of bf16 MFMAs (the pattern of the production kernels)
            f32x4 c0 = {0, 0, 0, 0};
            bf16x8 x, y;
            for (int q = 0; q < 8; ++q) { x[q] = (short)(threadIdx.x + q); y[q] = (short)(threadIdx.x * 3 + q); }
            for (int i = 0; i < iters; ++i) {
                c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, c0, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(y, x, c0, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, c0, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(y, x, c0, 0, 0, 0);
            }
            acc = c0[0];
        } else {
            f32x4 c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
            bf16x8 x, y;
            for (int q = 0; q < 8; ++q) { x[q] = (short)(threadIdx.x + q); y[q] = (short)(threadIdx.x * 3 + q); }
            for (int i = 0; i < iters; ++i) {
                c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(y, x, c1, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(y, x, c1, 0, 0, 0);
            }
            acc = c0[0] + c1[1];
        }
    }
    unsigned long long t1 = clock64();
    if ((threadIdx.x & 63) == 0) atomicAdd(&cyc[wave < 4 ? 0 : 1], t1 - t0);
    if (acc == 12345.678f) sink[0] = acc;
}
template <int KIND>
void run(const char* name) {
    float* sink; unsigned long long* cyc;
    hipMalloc(&sink, 4); hipMalloc(&cyc, 16);
    const int iters = 2000, nb = 256;
    for (int mode = 1; mode <= 3; ++mode) {
        hipMemset(cyc, 0, 16);
        hipLaunchKernelGGL(probe<KIND>, dim3(nb), dim3(512), 0, 0, mode, iters, sink, cyc);
        hipDeviceSynchronize();
        unsigned long long h[2]; hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
        printf("%s mode %d (A=VALU %s, B=MFMA %s): VALU wave %.1f cyc/iter (16 ops), MFMA wave %.1f cyc/iter (4 mfma)\n", name, mode,
               mode & 1 ? "on" : "off", mode & 2 ? "on" : "off", h[0] / (double)(nb * 4) / iters, h[1] / (double)(nb * 4) / iters);
    }
}
int main() { run<0>("f32 16x16x4 "); run<1>("bf16 16x16x32, 2 chains"); run<2>("bf16 16x16x32, 1 chain "); return 0; }
