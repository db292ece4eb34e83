// Issue cost of the vector instructions the all-pairs tail is made of (cycles per instruction per SIMD at 1..4 waves
// per SIMD, 16 independent instructions per iteration, registers only).
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)
template <int KIND>
__global__ __launch_bounds__(1024) void probe(int iters, float* sink, unsigned long long* cyc) {
    float a[16], b[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { a[i] = threadIdx.x * 0.001f + i; b[i] = 1.f + i; }
    __syncthreads();
    unsigned long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#define K0(i) asm volatile("v_fmac_f32 %0, %1, %1" : "+v"(a[i]) : "v"(b[i]));
#define K1(i) asm volatile("v_max_i32 %0, 0, %1" : "=v"(a[i]) : "v"(b[i]));
#define K2(i) asm volatile("v_pk_max_i16 %0, %1, 0" : "=v"(a[i]) : "v"(b[i]));
#define K3(i) asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %1" : "=v"(a[i]) : "v"(b[i]));
#define K4(i) asm volatile("v_fma_mixlo_f16 %0, %1, -1.0, %1 op_sel_hi:[1,0,0]" : "+v"(a[i]) : "v"(b[i]));
#define K5(i) asm volatile("v_fma_mixhi_f16 %0, %1, -1.0, %1 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(a[i]) : "v"(b[i]));
#define K6(i) asm volatile("v_pk_max_f16 %0, %1, 0" : "=v"(a[i]) : "v"(b[i]));
#define K7(i) asm volatile("v_exp_f32 %0, %1" : "=v"(a[i]) : "v"(b[i]));
#define K9(i) asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(a[i]) : "v"(b[i]));
#define K10(i) asm volatile("v_and_b32 %0, 0xffffe000, %1" : "=v"(a[i]) : "v"(b[i]));
#define K11(i) asm volatile("v_cvt_pk_f16_f32 %0, %1, %1" : "=v"(a[i]) : "v"(b[i]));
        if (KIND == 0) { REP16(K0) }
        if (KIND == 1) { REP16(K1) }
        if (KIND == 2) { REP16(K2) }
        if (KIND == 3) { REP16(K3) }
        if (KIND == 4) { REP16(K4) }
        if (KIND == 5) { REP16(K5) }
        if (KIND == 6) { REP16(K6) }
        if (KIND == 7) { REP16(K7) }
        if (KIND == 9) { REP16(K9) }
        if (KIND == 10) { REP16(K10) }
        if (KIND == 11) { REP16(K11) }
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
    printf("%-22s", name);
    for (int nt : {256, 512, 1024}) {
        hipMemset(cyc, 0, 8);
        hipLaunchKernelGGL(probe<KIND>, dim3(nb), dim3(nt), 0, 0, iters, sink, cyc);
        hipDeviceSynchronize();
        unsigned long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        double per_wave_iter = h / (double)(nb * (nt / 64)) / iters;
        printf("  %dw/SIMD: %5.2f cyc/inst/SIMD", nt / 256, per_wave_iter / 16 / (nt / 256));
    }
    printf("\n");
}
int main() {
    run<0>("v_fmac_f32"); run<1>("v_max_i32"); run<2>("v_pk_max_i16"); run<3>("v_cvt_pkrtz_f16_f32"); run<4>("v_fma_mixlo_f16");
    run<5>("v_fma_mixhi_f16"); run<6>("v_pk_max_f16"); run<7>("v_exp_f32"); run<9>("v_cvt_f32_f16");
    run<10>("v_and_b32"); run<11>("v_cvt_pk_f16_f32");
    return 0;
}
