// The all-pairs tail's instruction mix per 16 pairs - five v_mfma_f32_16x16x32_f16 (a chain of three, a chain of two)
// and ~15 vector instructions (2 v_cvt_pkrtz, 4 v_fma_mix, 2 v_pk_max_i16, 4 v_med3, 3 v_add) - issued four ways, 4 waves
// per SIMD on every SIMD of the chip (512 workgroups x 512 threads):
//   mixed   every wave carries both streams, the vector work depending on the matrix results (what the kernel does)
//   split   waves 0-3 of a workgroup (one per SIMD) issue ONLY the matrix instructions of two units, waves 4-7 ONLY the
//           vector instructions of two units: per SIMD the same work as `mixed`, on specialised waves, no data hand-over
//           (an upper bound on what wave specialisation could buy)
//   mfma    only the matrix instructions, every wave;   valu    only the vector instructions, every wave
// If vector and matrix issue overlapped between the waves of a SIMD (MI355X_MICROARCH.md, "MFMA and VALU pipes are
// separate"), split ~ max(mfma, valu); if their issue cycles add, split ~ mixed ~ mfma + valu.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef short i16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x4 mm(f16x8 a, f16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }

// the vector work between layer 1 and layer 2 (8 instructions) and after layer 2 (7)
__device__ __forceinline__ void valu_mid(const f32x4 h, unsigned& a, unsigned& b, unsigned& l01, unsigned& l23) {
    const unsigned h01 = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(h[0], h[1]));
    const unsigned h23 = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(h[2], h[3]));
    const i16x2 z = {0, 0};
    a = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(i16x2, h01), z));
    b = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(i16x2, h23), z));
    asm volatile("v_fma_mixlo_f16 %0, %2, -1.0, %4 op_sel_hi:[1,0,0] clamp\n\t"
                 "v_fma_mixlo_f16 %1, %3, -1.0, %6 op_sel_hi:[1,0,0] clamp\n\t"
                 "v_fma_mixhi_f16 %0, %2, -1.0, %5 op_sel:[1,0,0] op_sel_hi:[1,0,0] clamp\n\t"
                 "v_fma_mixhi_f16 %1, %3, -1.0, %7 op_sel:[1,0,0] op_sel_hi:[1,0,0] clamp"
                 : "=&v"(l01), "=&v"(l23)
                 : "v"(h01), "v"(h23), "v"(h[0]), "v"(h[1]), "v"(h[2]), "v"(h[3]));
}
__device__ __forceinline__ float valu_end(const f32x4 q) {
    const float t0 = __builtin_amdgcn_fmed3f(q[0], 0.f, INFINITY), t1 = __builtin_amdgcn_fmed3f(q[1], 0.f, -INFINITY);
    const float t2 = __builtin_amdgcn_fmed3f(q[2], 0.f, INFINITY), t3 = __builtin_amdgcn_fmed3f(q[3], 0.f, -INFINITY);
    return (t0 + t1) + (t2 + t3);
}

template <int MODE>   // 0 mixed, 1 split, 2 mfma only, 3 valu only
__global__ __launch_bounds__(512, 2) void probe(int units, float* sink) {
    const int wave = threadIdx.x >> 6;
    f16x8 a, b;
    for (int q = 0; q < 8; ++q) { a[q] = (_Float16)(0.001f * ((threadIdx.x & 63) + q)); b[q] = (_Float16)(0.002f * ((threadIdx.x & 31) + q)); }
    f32x4 u = {0.1f, 0.2f, 0.3f, 0.4f};
    float acc = 0.f;
    const bool do_m = MODE == 0 || MODE == 2 || (MODE == 1 && wave < 4);
    const bool do_v = MODE == 0 || MODE == 3 || (MODE == 1 && wave >= 4);
    const int n = MODE == 1 ? 2 * units : units;          // split: a specialised wave carries two units' worth of its kind
    f32x4 h = u, q = u;
    for (int i = 0; i < n; ++i) {
        asm volatile("" : "+v"(u));                       // (nothing of a unit is loop-invariant)
        if (do_m) {
            h = mm(a, b, u);
            h = mm(b, a, h);
            h = mm(a, b, h);
        }
        unsigned p0 = 0, p1 = 0, p2 = 0, p3 = 0;
        if (do_v) valu_mid(h, p0, p1, p2, p3);
        if (do_m) {
            const f16x8 hb = (MODE == 0) ? __builtin_bit_cast(f16x8, (uint4){p0, p1, p2, p3}) : b;
            q = mm(b, hb, u);
            q = mm(a, hb, q);
        }
        if (do_v) {
            float s = valu_end(MODE == 0 ? q : h);
            if (MODE != 0) s += __uint_as_float(p0 ^ p1 ^ p2 ^ p3);       // keep the mid results alive
            acc += s;
            if (MODE != 0) h[0] = acc;                                    // the vector stream depends on itself, unit to unit
        }
        asm volatile("" : "+v"(h), "+v"(q), "+v"(acc));
    }
    sink[blockIdx.x * 512 + threadIdx.x] = acc + h[1] + q[2];
}

template <int MODE>
static double run(const char* name, int units, float* sink) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    double best = 1e30;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(probe<MODE>, dim3(512), dim3(512), 0, 0, units, sink);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    // per SIMD: 4 waves x `units` units each (mixed / mfma / valu), or 2 + 2 specialised waves x 2 units (split) = 4 units' work
    printf("%-6s %8.3f ms  -> %7.2f ns per unit (5 matrix + 15 vector instructions) per SIMD\n", name, best,
           best * 1e6 / (4.0 * units));
    return best;
}

int main() {
    float* sink;
    hipMalloc(&sink, 512 * 512 * 4);
    const int units = 40000;
    for (int i = 0; i < 40; ++i) hipLaunchKernelGGL(probe<0>, dim3(512), dim3(512), 0, 0, units, sink);   // clock ramp
    hipDeviceSynchronize();
    const double mixed = run<0>("mixed", units, sink);
    const double split = run<1>("split", units, sink);
    const double mf = run<2>("mfma", units, sink);
    const double va = run<3>("valu", units, sink);
    printf("mixed / (mfma + valu) = %.3f   split / (mfma + valu) = %.3f   split / max(mfma, valu) = %.3f   split / mixed = %.3f\n",
           mixed / (mf + va), split / (mf + va), split / (mf > va ? mf : va), split / mixed);
    return 0;
}
