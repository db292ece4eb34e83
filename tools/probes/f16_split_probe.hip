// Probes for the f16 two-plane all-pairs tail (sgpr_score.hip):
//  1. does v_mfma_f32_16x16x32_f16 honour f16 DENORMAL inputs (the lo planes of small values are subnormal)?
//  2. v_cvt_pkrtz_f16_f32 + v_fma_mixlo/mixhi_f16 (with clamp) as the split  h = hi + lo  of a relu'd fp32 value:
//     semantics on negative inputs, subnormal residuals and values >= 2048.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void denorm_kernel(float a_val, float b_val, float* out) {
    const _Float16 a = (_Float16)a_val, b = (_Float16)b_val;
    f16x8 av = {a, a, a, a, a, a, a, a}, bv = {b, b, b, b, b, b, b, b};
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bv, c, 0, 0, 0);
    if (threadIdx.x == 0) out[0] = c[0];
}

__device__ __forceinline__ unsigned split_lo(unsigned hi_packed, float h0, float h1) {
    // lo.x = clamp01(h0 - hi.x), lo.y = clamp01(h1 - hi.y) as packed f16
    unsigned lo = 0;
    asm volatile("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0] clamp\n\t"
                 "v_fma_mixhi_f16 %0, %1, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0] clamp"
                 : "+v"(lo)
                 : "v"(hi_packed), "v"(h0), "v"(h1));
    return lo;
}

__global__ void split_kernel(const float* in, int n, float* hi_out, float* lo_out) {
    const int i = threadIdx.x;
    if (2 * i + 1 >= n + 1) return;
    const float h0 = in[2 * i], h1 = in[2 * i + 1];
    const f16x2 hp = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(h0, h1));
    unsigned hpu = __builtin_bit_cast(unsigned, hp);
    unsigned lo = split_lo(hpu, h0, h1);
    // relu on the hi plane: v_pk_max_f16 with 0
    f16x2 z = {(_Float16)0.f, (_Float16)0.f};
    f16x2 hr = __builtin_elementwise_max(hp, z);
    f16x2 lp = __builtin_bit_cast(f16x2, lo);
    hi_out[2 * i] = (float)hr[0];
    hi_out[2 * i + 1] = (float)hr[1];
    lo_out[2 * i] = (float)lp[0];
    lo_out[2 * i + 1] = (float)lp[1];
}

int main() {
    float* d;
    hipMalloc(&d, 64);
    float h;
    const float cases[][2] = {{ldexpf(1.f, -20), 1024.f}, {ldexpf(1.f, -24), 1024.f}, {ldexpf(1.f, -14), 1.f}, {ldexpf(1.f, -15), ldexpf(1.f, -15)}};
    for (auto& c : cases) {
        hipLaunchKernelGGL(denorm_kernel, dim3(1), dim3(64), 0, 0, c[0], c[1], d);
        hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
        printf("mfma f16: a=%g b=%g -> sum of 32 products %g (exact %g)\n", c[0], c[1], h, 32.0 * c[0] * c[1]);
    }
    const int n = 16;
    float in[n] = {1.000123f, -1.000123f, 0.1f, -0.1f, 3.14159274f, 1e-3f, 1e-5f, -1e-5f, 2047.7f, 2049.3f, 70000.f, 0.f, 6.1e-5f, 1e-7f, -70000.f, 0.33333334f};
    float *di, *dh, *dl, hh[n], hl[n];
    hipMalloc(&di, sizeof(in)); hipMalloc(&dh, sizeof(in)); hipMalloc(&dl, sizeof(in));
    hipMemcpy(di, in, sizeof(in), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(split_kernel, dim3(1), dim3(64), 0, 0, di, n, dh, dl);
    hipMemcpy(hh, dh, sizeof(in), hipMemcpyDeviceToHost);
    hipMemcpy(hl, dl, sizeof(in), hipMemcpyDeviceToHost);
    for (int i = 0; i < n; ++i) {
        const float r = in[i] > 0 ? in[i] : 0.f;
        printf("split: h=%-14.9g relu=%-14.9g hi=%-14.9g lo=%-14.9g hi+lo-relu=%g\n", in[i], r, hh[i], hl[i], (double)hh[i] + hl[i] - r);
    }
    return 0;
}
