// Per-graph operands of the all-pairs tail (sgpr_score.hip), shared by ntn_prep_kernel and by the epilogue of the embed
// kernel (sgpr_embed.hip: a graph that will be a row and a column of a score matrix leaves the embed launch with its
// tail operands already in place).  One function, so both producers give the same bits:
//   row r   A'_r[t][j] = sum_i e[i] W[i][j][t] + Wb[t][F + j]  (fp32 FMA chain over i ascending, one add), as two f16 planes
//           (hi = RNE(a), lo = RNE(a - hi)) in MFMA A-operand order: element (t, j) of plane p at
//           Ab[((r * 2 + p) * 64 + (j >> 3) * 16 + t) * 8 + (j & 7)]
//           u_r[t] = sum_m Wb[t][m] e[m] + bias[t]             (FMA chain over m ascending, one add)
//   column c  e itself as two f16 planes in MFMA B-operand order, 64 columns per super-block:
//           Cb[(((sb * 2 + p) * 4 + b) * 64 + (j >> 3) * 16 + c15) * 8 + (j & 7)], sb = c >> 6, c15 = (c & 63) >> 2, b = c & 3
// and the magnitudes the tail's f16 range check needs.  No barriers inside.
#pragma once
#include "sgpr_internal.hpp"

namespace sgpr {

__device__ __forceinline__ void prep_split2_f16(float a, unsigned short& h, unsigned short& l) {
    const _Float16 hh = (_Float16)a;                     // round to nearest even
    const _Float16 ll = (_Float16)(a - (float)hh);       // exact residual, rounded once
    h = __builtin_bit_cast(unsigned short, hh);
    l = __builtin_bit_cast(unsigned short, ll);
}

// Row operands of NG graphs at once (rows row0 .. row0 + n - 1, n <= NG; e[g] = pooled vector g in LDS): a weight is
// fetched once and meets all NG vectors - every output still is its own FMA chain over i ascending, so NG does not
// change a bit.  amax / umax: this thread's running maxima of |A'| and |u| (reduced by the caller).
template <int NG>
__device__ __forceinline__ void prep_rows(const DevWeights& w, const float (*e)[kF3], int tid, int nthreads, long long row0,
                                          int n, unsigned short* __restrict__ Ab, float* __restrict__ ur, float& amax,
                                          float& umax) {
    constexpr int F = kF3, T = kT;
    for (int o = tid; o < T * F; o += nthreads) {                        // o = t * 32 + j: neighbours along j
        const int t = o >> 5, j = o & 31;
        const float* wp = w.ntn_wt + (size_t)t * F + j;                  // Wt[i][t][j]
        float a[NG];
#pragma unroll
        for (int g = 0; g < NG; ++g) a[g] = 0.f;
#pragma unroll 8
        for (int i = 0; i < F; ++i) {
            const float wi = wp[(size_t)i * T * F];
#pragma unroll
            for (int g = 0; g < NG; ++g) a[g] = fmaf(e[g][i], wi, a[g]);
        }
        const float wb = w.ntn_wb[t * 2 * F + F + j];
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (g >= n) break;
            const float v = a[g] + wb;
            amax = fmaxf(amax, fabsf(v));
            unsigned short h, l;
            prep_split2_f16(v, h, l);
            unsigned short* dst = Ab + ((size_t)(row0 + g) * 2 * 64 + (j >> 3) * 16 + t) * 8 + (j & 7);
            dst[0] = h;
            dst[64 * 8] = l;
        }
    }
    for (int o = tid; o < NG * T; o += nthreads) {                       // u_r: graph o / 16, neuron o % 16
        const int g = o / T, t = o - g * T;
        if (g >= n) continue;
        float s = 0.f;
#pragma unroll 8
        for (int m = 0; m < F; ++m) s = fmaf(w.ntn_wb[t * 2 * F + m], e[g][m], s);
        s += w.ntn_bias[t];
        umax = fmaxf(umax, fabsf(s));
        ur[(size_t)(row0 + g) * T + t] = s;
    }
}

// Column operand of one graph: lanes j = 0 .. 31 of the calling wave (x = e_c[j], or 0 for a padding column past M)
__device__ __forceinline__ void prep_col(float x, int j, long long col, unsigned short* __restrict__ Cb, float& emax) {
    emax = fmaxf(emax, fabsf(x));
    unsigned short h, l;
    prep_split2_f16(x, h, l);
    const long long sb = col >> 6;
    const int cl = (int)(col & 63), c15 = cl >> 2, b = cl & 3;
    unsigned short* dst = Cb + (((size_t)sb * 2 * 4 + b) * 64 + (j >> 3) * 16 + c15) * 8 + (j & 7);
    dst[0] = h;
    dst[4 * 64 * 8] = l;
}

}  // namespace sgpr
