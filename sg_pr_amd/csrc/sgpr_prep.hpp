// Row / column operands of the all-pairs tail (sgpr_score.hip): the stand-alone prep kernels' body, its A' tile code
// (shared with score_all_pairs_self_kernel, which forms A' of the row graphs it works on by itself) and the f16 plane split
// (shared with the embed kernels' epilogue, which leaves a graph's own column planes behind: sgpr_embed.hip).  Reference:
// TenorNetworkModule.forward, layers_batch.py:77-81 - the bilinear form e1^T W and the block term's halves are hoisted per graph.
#pragma once
#include "sgpr_internal.hpp"

namespace sgpr {

typedef float f32x4_t __attribute__((ext_vector_type(4)));

constexpr int AP_SB = 64;      // columns per super-block of the dense tail (4 MFMA column blocks)

// bytes of LDS the body needs: the staged operand units of 16 graphs + the waves' range partials
constexpr int kPrepStageBytes = 16 * 2 * 4 * 8 * 8 * 2;
constexpr int kPrepLdsBytes = kPrepStageBytes + 4 * 4 * 4;

__device__ __forceinline__ void prep_split2_f16(float a, _Float16& h, _Float16& l) {
    h = (_Float16)a;                       // round to nearest even
    l = (_Float16)(a - (float)h);          // exact residual, rounded once: |a - (h + l)| <= 2^-23 |a| (or 2^-25 absolute)
}

__device__ __forceinline__ float prep_wave_max(float v) {
    v = fmaxf(v, __shfl_xor(v, 1));
    v = fmaxf(v, __shfl_xor(v, 2));
    v = fmaxf(v, __shfl_xor(v, 4));
    v = fmaxf(v, __shfl_xor(v, 8));
    v = fmaxf(v, __shfl_xor(v, 16));
    return fmaxf(v, __shfl_xor(v, 32));
}

constexpr int PREP_DENSE = 0;     // rows [R][F] by index, operands written at the same index, columns in super-block order
constexpr int PREP_LIST = 1;      // pair-list mode: the R row graphs are rows[row_ids[0 .. R)] (gathered, written compactly);
                                  // column operands per graph, Cb [M][2 planes][32] f16

// A' = e^T W + Wb[:, F:] of the 16 row graphs g0 .. g0 + 15 (rows past R skipped), output tiles `half` (0 / 1: 16 of the
// 32 tiles; tile = (t, j half), this wave takes four of them) -> two f16 planes in `stage` ([graph][plane][j >> 3][t & 7]
// [j & 7]: the tail's A-operand units); amax / l1max: running max |A'| and max over (graph, t) of sum_j |A'[t][j]|.
// QB tiles' weights are requested at a time (4: one L2 round trip per call).  256 threads.
template <int MODE, int QB>
__device__ __forceinline__ void prep_rows_tiles(const float* __restrict__ ntn_wt, const float* __restrict__ ntn_wb,
                                                const float* __restrict__ rows, int R, const int g0,
                                                const int half, const int32_t* __restrict__ row_ids,
                                                unsigned short* __restrict__ stage, float& amax, float& l1max) {
    constexpr int F = kF3, T = kT;
    const int lane = threadIdx.x & 63, wave = (threadIdx.x >> 6) & 3;
    const int l15 = lane & 15, lq = lane >> 4;
    auto ld = [](const float* p) { return *p; };
    auto ld4 = [&](const float* p) { return *reinterpret_cast<const float4*>(p); };
    auto src_of = [&](int s) { return (MODE != PREP_DENSE && row_ids) ? row_ids[s] : s; };
    (void)ld;
    // A operand: E[g0 + l15][16 blk + 4 lq .. +3] (k order permuted: lane group q supplies k = 4q + s at step s)
    const int ga = min(g0 + l15, R - 1);
    const float* e = rows + (size_t)src_of(ga) * F + 4 * lq;
    const float4 ea0 = ld4(e), ea1 = ld4(e + 16);
    // the wave's four output tiles: all 32 weight operands (and the four block-term values) are requested before the
    // first matrix instruction - one L2 round trip for the workgroup's critical path instead of four
    float l1r[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int qb = 0; qb < 4; qb += QB) {
    float wv[QB][8], wbv[QB];
#pragma unroll
    for (int q = 0; q < QB; ++q) {
        const int tile = half * 16 + wave * 4 + qb + q;
        const int t = tile >> 1, j = (tile & 1) * 16 + l15;
        const float* wp = ntn_wt + ((size_t)(4 * lq) * T + t) * F + j;            // Wt[i = 4 lq + s][t][j]
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            wv[q][s4] = wp[s4 * T * F];
            wv[q][4 + s4] = wp[(16 + s4) * T * F];
        }
        wbv[q] = ntn_wb[t * 2 * F + F + j];
    }
#pragma unroll
    for (int q = 0; q < QB; ++q) {
        const int tile = half * 16 + wave * 4 + qb + q;
        const int t = tile >> 1, j = (tile & 1) * 16 + l15;
        f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ea0.x, wv[q][0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ea0.y, wv[q][1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ea0.z, wv[q][2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ea0.w, wv[q][3], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ea1.x, wv[q][4], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ea1.y, wv[q][5], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ea1.z, wv[q][6], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ea1.w, wv[q][7], acc, 0, 0, 0);
        // acc[r] = (e1^T W)_{g0 + 4 lq + r}[t][j]; the column half of the block term rides along: A' = A + Wb[t][F + j]
        const float wbc = wbv[q];
        // tiles q = 0, 1 (and 2, 3) are the two halves j < 16 / j >= 16 of the same t: a row of A' is the 16 lanes of
        // a lane group in both of them
        if ((q & 1) == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) l1r[r] = 0.f;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int g = g0 + 4 * lq + r;
            if (g < R) {
                const float a = acc[r] + wbc;
                amax = fmaxf(amax, fabsf(a));
                l1r[r] += fabsf(a);
                _Float16 h, l;
                prep_split2_f16(a, h, l);
                // staged through LDS in the operand layout: the lanes hold one f16 each of a 16-byte operand unit
                // (8 consecutive j of one (graph, plane, t)); 2-byte global stores cost the kernel a quarter of its time
                unsigned short* dst = stage + (((4 * lq + r) * 2 * 4 + (j >> 3)) * 8 + (t & 7)) * 8 + (j & 7);
                dst[0] = __builtin_bit_cast(unsigned short, h);
                dst[4 * 8 * 8] = __builtin_bit_cast(unsigned short, l);
            }
        }
        if (q & 1) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = l1r[r];
                v += __shfl_xor(v, 1);
                v += __shfl_xor(v, 2);
                v += __shfl_xor(v, 4);
                v += __shfl_xor(v, 8);
                l1max = fmaxf(l1max, v);
            }
        }
    }
    }
}

// 16 graphs per pair of work units (block >> 1 = the group, block & 1 = which half of the 32 output tiles and which 8 of
// the 16 graphs' block terms / column operands).  Row graphs get A' (16 x 32 per graph) as ONE small GEMM,
// [16 graphs x 32] x [32 x 512] on the fp32 matrix cores - the 64 KB weight tensor crosses L2 -> CU once per 16 graphs
// instead of once per graph - plus the column half of the block term, split into two f16 planes on the way out, and
// u_r; column graphs get their two-plane copy (columns past M are zero-filled).
// Runs on the first 256 threads of the workgroup (the others only join the barriers); `lds`: kPrepLdsBytes, 16-byte aligned.
template <int MODE>
__device__ __forceinline__ void ntn_prep_body(const DevWeights& w, const float* __restrict__ rows, int R,
                                              const float* __restrict__ cols, int M, unsigned short* __restrict__ Ab,
                                              float* __restrict__ ur, float* __restrict__ rng,
                                              unsigned short* __restrict__ Cb, const int block,
                                              const int32_t* __restrict__ row_ids, unsigned char* __restrict__ lds) {
    constexpr int F = kF3, T = kT;
    unsigned short* stage = reinterpret_cast<unsigned short*>(lds);           // [graph][plane][j >> 3][t & 7][j & 7]
    float(*red)[4] = reinterpret_cast<float(*)[4]>(lds + kPrepStageBytes);
    const bool worker = threadIdx.x < 256;
    const int lane = threadIdx.x & 63, wave = (threadIdx.x >> 6) & 3;
    const int l15 = lane & 15, lq = lane >> 4;
    // two work units per 16 graphs (half of the 32 output tiles each): twice the resident waves for a latency-bound job
    const int g0 = (block >> 1) * 16, half = block & 1;
    // where row slot s is read from / its operands are written to
    auto ld = [](const float* p) { return *p; };
    auto src_of = [&](int s) { return (MODE != PREP_DENSE && row_ids) ? row_ids[s] : s; };
    auto dst_of = [&](int s) { return s; };
    float amax = 0.f, umax = 0.f, emax = 0.f, l1max = 0.f;       // l1max: max over (graph, t) of sum_j |A'[t][j]|
    if (g0 < R && worker) prep_rows_tiles<MODE, 4>(w.ntn_wt, w.ntn_wb, rows, R, g0, half, row_ids, stage, amax, l1max);
    if (g0 < R) {
        __syncthreads();
        // 16 graphs x 2 planes x 4 (j >> 3) x 8 t of this half = 1024 units of 16 bytes, 8 consecutive t contiguous in memory
        for (int u = threadIdx.x; worker && u < 16 * 2 * 4 * 8; u += 256) {
            const int tl = u & 7, jb = (u >> 3) & 3, pl = (u >> 5) & 1, gi = u >> 6;
            if (g0 + gi < R)
                *reinterpret_cast<uint4*>(Ab + (((size_t)dst_of(g0 + gi) * 2 + pl) * 64 + jb * 16 + half * 8 + tl) * 8) =
                    *reinterpret_cast<const uint4*>(stage + (size_t)u * 8);
        }
    }
    // block term of the row graphs and the column operands: one graph per wave pass
    for (int gi = half * 8 + wave * 2; worker && gi < half * 8 + wave * 2 + 2; ++gi) {
        const int g = g0 + gi;
        if (g < R) {
            const float* e1 = rows + (size_t)src_of(g) * F;
            float s = 0.f;
            for (int m = 0; m < 8; ++m) s = fmaf(w.ntn_wb[l15 * 2 * F + lq * 8 + m], ld(e1 + lq * 8 + m), s);
            s += __shfl_xor(s, 16);
            s += __shfl_xor(s, 32);
            s += w.ntn_bias[l15];
            umax = fmaxf(umax, fabsf(s));
            if (lq == 0) ur[(size_t)dst_of(g) * T + l15] = s;
        }
        const int msb = MODE == PREP_LIST ? M : (M + AP_SB - 1) / AP_SB * AP_SB;
        const bool has_col = g < msb;                       // (columns past M: zeros)
        if (has_col && lane < F) {                          // the column operand itself, two f16 planes
            const int c = g;
            const float x = g < M ? ld(cols + (size_t)c * F + lane) : 0.f;
            emax = fmaxf(emax, fabsf(x));
            _Float16 h, l;
            prep_split2_f16(x, h, l);
            if (MODE == PREP_LIST) {
                unsigned short* dst = Cb + (size_t)c * (2 * F) + lane;
                dst[0] = __builtin_bit_cast(unsigned short, h);
                dst[F] = __builtin_bit_cast(unsigned short, l);
            } else {
                const int sb = c >> 6, cl = c & 63, c15 = cl >> 2, b = cl & 3, j = lane;
                unsigned short* dst = Cb + ((((size_t)sb * 2) * 4 + b) * 64 + (j >> 3) * 16 + c15) * 8 + (j & 7);
                dst[0] = __builtin_bit_cast(unsigned short, h);
                dst[4 * 64 * 8] = __builtin_bit_cast(unsigned short, l);
            }
        }
    }
    // NaN inputs: fmaxf drops them, so fold an explicit "not finite" marker in (infinity fails every bound)
    amax = prep_wave_max(amax);
    umax = prep_wave_max(umax);
    emax = prep_wave_max(emax);
    l1max = prep_wave_max(l1max);
    if (lane == 0 && worker) {
        red[wave][0] = amax;
        red[wave][1] = umax;
        red[wave][2] = emax;
        red[wave][3] = l1max;
    }
    __syncthreads();
    if (threadIdx.x < 4) {
        const int q = threadIdx.x;
        rng[(size_t)block * 4 + q] = fmaxf(fmaxf(red[0][q], red[1][q]), fmaxf(red[2][q], red[3][q]));
    }
}

}  // namespace sgpr
