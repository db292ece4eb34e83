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
    float ninf = -__builtin_inff();
    asm volatile("" : "+v"(ninf));
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
#define K12(i) asm volatile("v_min_f32 %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
#define K13(i) asm volatile("v_max_f32 %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
#define K14(i) asm volatile("v_med3_f32 %0, %1, %0, %2" : "+v"(a[i]) : "v"(b[i]), "v"(ninf));
#define K15(i) asm volatile("v_max3_f32 %0, %1, %0, %1" : "+v"(a[i]) : "v"(b[i]));
#define K16(i) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b[i]) : "vcc");
#define K17(i) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(a[i]) : "v"(b[i]));
#define K18(i) asm volatile("v_add_f32 %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
#define K19(i) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
#define K20(i) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %1" : "=v"(a[i]) : "v"(b[i]));
#define K21(i) asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(a[i]) : "v"(b[i]));
#define K22(i) asm volatile("v_min_i32 %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
#define K23(i) asm volatile("v_perm_b32 %0, %1, %0, %1" : "+v"(a[i]) : "v"(b[i]));
#define K24(i) asm volatile("v_fma_f32 %0, %1, %1, %0" : "+v"(a[i]) : "v"(b[i]));
#define K25(i) asm volatile("v_add_u32 %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
#define K26(i) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(*reinterpret_cast<double*>(&a[(i) & 14])) : "v"(*reinterpret_cast<double*>(&b[(i) & 14])));
#define K27(i) asm volatile("v_min_u32 %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
#define K28(i) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[i]), "v"(b[i]) : "vcc");
        if (KIND == 12) { REP16(K12) }
        if (KIND == 13) { REP16(K13) }
        if (KIND == 14) { REP16(K14) }
        if (KIND == 15) { REP16(K15) }
        if (KIND == 16) { REP16(K16) }
        if (KIND == 17) { REP16(K17) }
        if (KIND == 18) { REP16(K18) }
        if (KIND == 19) { REP16(K19) }
        if (KIND == 20) { REP16(K20) }
        if (KIND == 21) { REP16(K21) }
        if (KIND == 22) { REP16(K22) }
        if (KIND == 23) { REP16(K23) }
        if (KIND == 24) { REP16(K24) }
        if (KIND == 25) { REP16(K25) }
        if (KIND == 26) { REP16(K26) }
        if (KIND == 27) { REP16(K27) }
        if (KIND == 28) { REP16(K28) }
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
    run<12>("v_min_f32"); run<13>("v_max_f32"); run<14>("v_med3_f32"); run<15>("v_max3_f32"); run<16>("v_cmp+v_cndmask (2)");
    run<17>("v_mov_b32_dpp quad"); run<18>("v_add_f32"); run<19>("v_mul_f32"); run<20>("v_cvt_pk_bf16_f32"); run<21>("v_lshlrev_b32");
    run<22>("v_min_i32"); run<23>("v_perm_b32"); run<24>("v_fma_f32 (VOP3)"); run<25>("v_add_u32"); run<26>("v_pk_add_f32 (2 values)");
    run<27>("v_min_u32"); run<28>("v_cmp_lt_f32");
    return 0;
}
