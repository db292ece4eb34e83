// Pair-coupled tail of SG.forward on gfx950: Neural Tensor Network
// (TenorNetworkModule.forward, reference layers_batch.py:70-83) + fully_connected_first
// / ReLU + scoring_layer / sigmoid (sg_net.py:131-136).
//
//  * score_pairs_kernel      one wave64 per pair (pair-list mode: eval_batch.py:30-36,
//                            SG.forward's per-pair tail).
//  * ntn_prep_kernel + score_all_pairs_kernel
//                            dense R x M rectangle.  The bilinear form AND the column half of the block term are
//                            hoisted per row graph (A'_r = e1^T W + Wb[:, F:], 16x32; u_r = Wb[:, :F] e1 + bias), so a
//                            pair costs 512 + 256 + 16 FMA.  Both dense layers run on v_mfma_f32_16x16x32_f16 with
//                            two-plane f16 operands (x = hi + lo, 22 bits; three cross products per layer), layer 2
//                            chained off the accumulator layout of layer 1; a wave keeps 4 row graphs in registers and
//                            walks 64-column super-blocks; every store instruction writes 4 rows x 256 contiguous bytes.
#include <math.h>

#include <string.h>

#include "sgpr_internal.hpp"

namespace sgpr {

constexpr int F = kF3;   // 32 pooled features
constexpr int T = kT;    // 16 tensor neurons
constexpr int BN_ = kB;  // 16 bottleneck neurons

// ------------------------------------------------------------------ per-pair list
// TenorNetworkModule.forward for one pair on one wave64: lanes with equal t = lane & 15 return the same
// relu(e1^T W[:, :, t] e2 + Wb[t, :] . [e1; e2] + bias[t])      (layers_batch.py:77-83)
//   ntn_w [32][32*16] = weight_matrix.view(F3, -1) (col = j*16 + t),  ntn_wb [16][64],  bias [16]
__device__ __forceinline__ float ntn_neuron(const float* __restrict__ ntn_w, const float* __restrict__ ntn_wb,
                                            const float* __restrict__ bias, const float* __restrict__ e1,
                                            const float* __restrict__ e2, int lane) {
    const int t = lane & 15, q = lane >> 4;
    // v[r] = sum_i e1[i] * W[i][col_r],  col_r = lane + 64 r  ->  j = q + 4r, same t for every r
    float v[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) v[r] = 0.f;
    for (int i = 0; i < F; ++i) {
        const float a = e1[i];
        const float* wr = ntn_w + i * (F * T) + lane;
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = fmaf(a, wr[64 * r], v[r]);
    }
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r) s = fmaf(v[r], e2[q + 4 * r], s);
    // block term  Wb[t][:] . [e1; e2], 16 of the 64 products per lane group
    for (int m = 0; m < 16; ++m) {
        const int mm = q * 16 + m;
        const float x = mm < F ? e1[mm] : e2[mm - F];
        s = fmaf(ntn_wb[t * 2 * F + mm], x, s);
    }
    s += __shfl_xor(s, 16);
    s += __shfl_xor(s, 32);
    return fmaxf(s + bias[t], 0.f);
}

__global__ __launch_bounds__(256) void score_pairs_kernel(const DevWeights w, const float* __restrict__ p1,
                                                          const int32_t* __restrict__ i1,
                                                          const float* __restrict__ p2,
                                                          const int32_t* __restrict__ i2, int64_t P,
                                                          float* __restrict__ score) {
    const int lane = threadIdx.x & 63;
    const int64_t pair = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pair >= P) return;
    const int64_t r1 = i1 ? i1[pair] : pair;
    const int64_t r2 = i2 ? i2[pair] : pair;
    const int t = lane & 15;
    const float h = ntn_neuron(w.ntn_w, w.ntn_wb, w.ntn_bias, p1 + r1 * F, p2 + r2 * F, lane);
    // fully_connected_first + ReLU: lane t computes output neuron t
    float gacc = w.fc1_b[t];
    for (int tt = 0; tt < T; ++tt) gacc = fmaf(w.fc1_w[t * T + tt], __shfl(h, tt), gacc);
    float z = fmaxf(gacc, 0.f) * w.fc2_w[t];
    z += __shfl_xor(z, 1);
    z += __shfl_xor(z, 2);
    z += __shfl_xor(z, 4);
    z += __shfl_xor(z, 8);
    if (lane == 0) score[pair] = 1.f / (1.f + expf(-(z + w.fc2_b[0])));
}

// stand-alone TenorNetworkModule.forward: out [B][16] = the similarity vector before the FC head
__global__ __launch_bounds__(256) void ntn_kernel(const float* __restrict__ ntn_w, const float* __restrict__ ntn_wb,
                                                  const float* __restrict__ bias, const float* __restrict__ e1,
                                                  const float* __restrict__ e2, int64_t B, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t pair = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pair >= B) return;
    const float h = ntn_neuron(ntn_w, ntn_wb, bias, e1 + pair * F, e2 + pair * F, lane);
    if (lane < T) out[pair * T + lane] = h;
}

int launch_ntn(const float* w, const float* wb, const float* bias, const float* e1, const float* e2, int64_t B,
               float* out, hipStream_t stream) {
    if (B == 0) return SGPR_OK;
    const int64_t blocks = (B + 3) / 4;
    if (blocks > 0x7fffffffLL) {
        set_error("sgpr_ntn: too many pairs for one launch");
        return SGPR_E_INVALID;
    }
    hipLaunchKernelGGL(ntn_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, w, wb, bias, e1, e2, B, out);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "ntn_kernel launch");
    return SGPR_OK;
}

int launch_score_pairs(const sgpr_handle* h, const float* p1, const int32_t* i1, const float* p2, const int32_t* i2,
                       int64_t P, float* score, hipStream_t stream) {
    if (P == 0) return SGPR_OK;
    const int64_t blocks = (P + 3) / 4;
    if (blocks > 0x7fffffffLL) {
        set_error("sgpr_score_pairs: too many pairs for one launch");
        return SGPR_E_INVALID;
    }
    hipLaunchKernelGGL(score_pairs_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, h->w, p1, i1, p2, i2, P, score);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "score_pairs_kernel launch");
    return SGPR_OK;
}

// ------------------------------------------------------------------ dense all-pairs
// workspace layout:  ur [R][T] f32 | rng [NWG][4] f32 | Ab [R][2][64][8] f16 | Cb [NSB][2][4][64][8] f16
//   ur   u_r[t] = Wb[t][:F] . e1_r + bias[t]
//   Ab   A'_r[t][j] = sum_i e1_r[i] W[i][j][t] + Wb[t][F + j]  as two f16 planes (hi = RNE(a), lo = RNE(a - hi): 22 bits)
//        in MFMA A-operand order: lane (g = j >> 3, l15 = t) holds j = 8g .. 8g+7 -> a wave's operand = 1 KB contiguous
//   Cb   column vectors e2_c as two f16 planes in MFMA B-operand order, 64 columns per super-block: block b (0..3) of a
//        super-block feeds MFMA column l15 with graph column 64 sb + 4 l15 + b, so that after the head a lane owns four
//        CONSECUTIVE columns of one row (one 16-byte store); lane (g, l15) holds j = 8g .. 8g+7
//   rng  per prep workgroup: max |A'|, max |u|, max |e2| - the main kernel derives a bound on every f16 it will form
// f16 has 11 significant bits and a 65504 range: inputs whose bound reaches that range take the exact fp32 per-pair
// path inside score_all_pairs_kernel (never on real data: |pooled| ~ 10); subnormal lo planes are honoured by the
// matrix core (tools/probes/f16_split_probe.hip), so small values only lose what fp32 would lose as well.
constexpr int AP_RW = 4;       // row graphs per wave: their A' operands (2 planes) stay in registers
constexpr int AP_ROWS = 16;    // row graphs per workgroup: 4 waves x AP_RW
constexpr int AP_SB = 64;      // columns per super-block (4 MFMA column blocks)
constexpr int AP_COLS = 256;   // column graphs per work item (4 super-blocks)
#ifndef SGPR_AP_OCC
#define SGPR_AP_OCC 4
#endif
// the multi-rectangle kernel carries the job table on top: at four workgroups per CU (128 VGPRs) it spills 360 bytes per
// lane, at three (166 VGPRs) nothing - config 4's tail 368.5 -> 358.9 us on one box (profiles/r05_tail_variants.txt)
constexpr int AP_MULTI_OCC = 3;
constexpr int AP_OCC = SGPR_AP_OCC;   // resident workgroups per CU the kernel is compiled for (waves per SIMD)
#ifndef SGPR_AP_NI
#define SGPR_AP_NI 2            // (same-box A/B, round 5: 97.7 -> 96.5 us per KITTI-00 matrix, twice; bit-identical - program order only)
#endif
#ifndef SGPR_AP_CHAINS
#define SGPR_AP_CHAINS 1        // 2: layer 1's correction products (lo.hi, hi.lo) in an accumulator chain of their own, met by the
#endif                          // hi.hi chain in one vector add - a shorter dependent chain for four more vector instructions per
                                // (row, block); A/B builds (tools/build_variant.sh): measured slower, profiles/r05_tail_variants.txt
#ifndef SGPR_AP_NT_STORE
#define SGPR_AP_NT_STORE 0      // 1: the matrix leaves through non-temporal stores (A/B builds, tools/build_variant.sh)
#endif
constexpr int AP_NI = SGPR_AP_NI;   // row graphs interleaved in program order (see score_all_pairs_kernel)
constexpr float AP_F16_SAFE = 60000.f;

static inline int ap_prep_groups(int R, int M) {
    const int msb = (M + AP_SB - 1) / AP_SB * AP_SB;
    return ((R > msb ? R : msb) + 15) / 16;
}

// (sized for the three bf16 planes of the wide-range instance, score_all_pairs_wide_kernel; the default two f16 planes
//  use two thirds of the operand regions)
size_t score_all_pairs_ws_bytes(int R, int M) {
    const size_t nsb = (size_t)(M + AP_SB - 1) / AP_SB;
    return (size_t)R * T * sizeof(float) + (size_t)2 * ap_prep_groups(R, M) * 4 * sizeof(float) +
           (size_t)R * 3 * 64 * 8 * sizeof(unsigned short) + nsb * 3 * 4 * 64 * 8 * sizeof(unsigned short);
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void split2_f16(float a, _Float16& h, _Float16& l) {
    h = (_Float16)a;                       // round to nearest even
    l = (_Float16)(a - (float)h);          // exact residual, rounded once: |a - (h + l)| <= 2^-23 |a| (or 2^-25 absolute)
}

// x = hi + mid + lo EXACTLY, as three bf16 planes: each plane is the upper half of what is left (truncation: 8 + 8 + 8 bits =
// fp32's 24 significant bits; every plane carries x's sign, so a ReLU of x is a ReLU of each plane)
__device__ __forceinline__ void split3_bf16(float x, unsigned& hb, unsigned& mb, unsigned& lb) {
    hb = __float_as_uint(x);
    const float r1 = x - __uint_as_float(hb & 0xffff0000u);
    mb = __float_as_uint(r1);
    const float r2 = r1 - __uint_as_float(mb & 0xffff0000u);
    lb = __float_as_uint(r2);                // (the plane is the upper 16 bits of each word)
}

__device__ __forceinline__ float wave_max_f32(float v) {
    v = fmaxf(v, __shfl_xor(v, 1));
    v = fmaxf(v, __shfl_xor(v, 2));
    v = fmaxf(v, __shfl_xor(v, 4));
    v = fmaxf(v, __shfl_xor(v, 8));
    v = fmaxf(v, __shfl_xor(v, 16));
    return fmaxf(v, __shfl_xor(v, 32));
}

// 16 graphs per pair of workgroups.  Row graphs get A' (16 x 32 per graph) as ONE small GEMM per workgroup,
// [16 graphs x 32] x [32 x 512] on the fp32 matrix cores - the 64 KB weight tensor crosses L2 -> CU once per 16 graphs
// instead of once per graph - plus the column half of the block term, split into two f16 planes on the way out, and
// u_r; column graphs get their two-plane copy in super-block order (columns past M are zero-filled).
// LIST (pair-list mode, score_pair_list_kernel): the R row graphs are rows[row_ids[0 .. R)] (the distinct row graphs of
// the list, gathered) and the column operands are laid out per graph, Cb [M][2 planes][32] f16, instead of per super-block.
// NPL: operand planes - 2 (f16: x = hi + lo, 22 bits) or 3 (bf16, the wide-range instance: x exactly, fp32's range)
template <bool LIST, int NPL = 2>
__device__ __forceinline__ void ntn_prep_body(const DevWeights& w, const float* __restrict__ rows, int R,
                                              const float* __restrict__ cols, int M, unsigned short* __restrict__ Ab,
                                              float* __restrict__ ur, float* __restrict__ rng,
                                              unsigned short* __restrict__ Cb, const int block,
                                              const int32_t* __restrict__ row_ids = nullptr) {
    __shared__ float red[4][4];
    __shared__ __attribute__((aligned(16))) unsigned short stage[16 * NPL * 4 * 8 * 8];   // [graph][plane][j >> 3][t & 7][j & 7]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, lq = lane >> 4;
    // two workgroups per 16 graphs (half of the 32 output tiles each): twice the resident waves for a latency-bound job
    const int g0 = (block >> 1) * 16, half = block & 1;
    float amax = 0.f, umax = 0.f, emax = 0.f, l1max = 0.f;       // l1max: max over (graph, t) of sum_j |A'[t][j]|
    if (g0 < R) {
        // A operand: E[g0 + l15][16 blk + 4 lq .. +3] (k order permuted: lane group q supplies k = 4q + s at step s)
        const int ga = min(g0 + l15, R - 1);
        const float* e = rows + (size_t)(LIST ? row_ids[ga] : ga) * F + 4 * lq;
        const float4 ea0 = *reinterpret_cast<const float4*>(e), ea1 = *reinterpret_cast<const float4*>(e + 16);
        // the wave's four output tiles: all 32 weight operands (and the four block-term values) are requested before the
        // first matrix instruction - one L2 round trip for the workgroup's critical path instead of four
        float wv[4][8], wbv[4], l1r[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int tile = half * 16 + wave * 4 + q;
            const int t = tile >> 1, j = (tile & 1) * 16 + l15;
            const float* wp = w.ntn_wt + ((size_t)(4 * lq) * T + t) * F + j;            // Wt[i = 4 lq + s][t][j]
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                wv[q][s4] = wp[s4 * T * F];
                wv[q][4 + s4] = wp[(16 + s4) * T * F];
            }
            wbv[q] = w.ntn_wb[t * 2 * F + F + j];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int tile = half * 16 + wave * 4 + q;
            const int t = tile >> 1, j = (tile & 1) * 16 + l15;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
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
                    // staged through LDS in the operand layout: the lanes hold one f16 each of a 16-byte operand unit
                    // (8 consecutive j of one (graph, plane, t)); 2-byte global stores cost the kernel a quarter of its time
                    unsigned short* dst = stage + (((4 * lq + r) * NPL * 4 + (j >> 3)) * 8 + (t & 7)) * 8 + (j & 7);
                    if constexpr (NPL == 3) {
                        unsigned hb, mb, lb;
                        split3_bf16(a, hb, mb, lb);
                        dst[0] = (unsigned short)(hb >> 16);
                        dst[4 * 8 * 8] = (unsigned short)(mb >> 16);
                        dst[2 * 4 * 8 * 8] = (unsigned short)(lb >> 16);
                    } else {
                        _Float16 h, l;
                        split2_f16(a, h, l);
                        dst[0] = __builtin_bit_cast(unsigned short, h);
                        dst[4 * 8 * 8] = __builtin_bit_cast(unsigned short, l);
                    }
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
    if (g0 < R) {
        __syncthreads();
        // 16 graphs x 2 planes x 4 (j >> 3) x 8 t of this half = 1024 units of 16 bytes, 8 consecutive t contiguous in memory
        for (int u = threadIdx.x; u < 16 * NPL * 4 * 8; u += 256) {
            const int tl = u & 7, jb = (u >> 3) & 3, pl = (u >> 5) % NPL, gi = (u >> 5) / NPL;
            if (g0 + gi < R)
                *reinterpret_cast<uint4*>(Ab + (((size_t)(g0 + gi) * NPL + pl) * 64 + jb * 16 + half * 8 + tl) * 8) =
                    *reinterpret_cast<const uint4*>(stage + (size_t)u * 8);
        }
    }
    // block term of the row graphs and the column operands: one graph per wave pass
    for (int gi = half * 8 + wave * 2; gi < half * 8 + wave * 2 + 2; ++gi) {
        const int g = g0 + gi;
        if (g < R) {
            const float* e1 = rows + (size_t)(LIST ? row_ids[g] : g) * F;
            float s = 0.f;
            for (int m = 0; m < 8; ++m) s = fmaf(w.ntn_wb[l15 * 2 * F + lq * 8 + m], e1[lq * 8 + m], s);
            s += __shfl_xor(s, 16);
            s += __shfl_xor(s, 32);
            s += w.ntn_bias[l15];
            umax = fmaxf(umax, fabsf(s));
            if (lq == 0) ur[(size_t)g * T + l15] = s;
        }
        const int msb = LIST ? M : (M + AP_SB - 1) / AP_SB * AP_SB;
        if (g < msb && lane < F) {                          // the column operand itself, two f16 planes (zeros past M)
            const float x = g < M ? cols[(size_t)g * F + lane] : 0.f;
            emax = fmaxf(emax, fabsf(x));
            _Float16 h, l;
            split2_f16(x, h, l);
            if (LIST) {
                unsigned short* dst = Cb + (size_t)g * (2 * F) + lane;
                dst[0] = __builtin_bit_cast(unsigned short, h);
                dst[F] = __builtin_bit_cast(unsigned short, l);
            } else {
                const int sb = g >> 6, cl = g & 63, c15 = cl >> 2, b = cl & 3, j = lane;
                unsigned short* dst = Cb + ((((size_t)sb * NPL) * 4 + b) * 64 + (j >> 3) * 16 + c15) * 8 + (j & 7);
                if constexpr (NPL == 3) {
                    unsigned hb, mb, lb;
                    split3_bf16(x, hb, mb, lb);
                    dst[0] = (unsigned short)(hb >> 16);
                    dst[4 * 64 * 8] = (unsigned short)(mb >> 16);
                    dst[2 * 4 * 64 * 8] = (unsigned short)(lb >> 16);
                } else {
                    dst[0] = __builtin_bit_cast(unsigned short, h);
                    dst[4 * 64 * 8] = __builtin_bit_cast(unsigned short, l);
                }
            }
        }
    }
    // NaN inputs: fmaxf drops them, so fold an explicit "not finite" marker in (infinity fails every bound)
    amax = wave_max_f32(amax);
    umax = wave_max_f32(umax);
    emax = wave_max_f32(emax);
    l1max = wave_max_f32(l1max);
    if (lane == 0) {
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

__global__ __launch_bounds__(256) void ntn_prep_kernel(const DevWeights w, const float* __restrict__ rows, int R,
                                                       const float* __restrict__ cols, int M,
                                                       unsigned short* __restrict__ Ab, float* __restrict__ ur,
                                                       float* __restrict__ rng, unsigned short* __restrict__ Cb) {
    ntn_prep_body<false>(w, rows, R, cols, M, Ab, ur, rng, Cb, (int)blockIdx.x);
}

__global__ __launch_bounds__(256) void ntn_prep_wide_kernel(const DevWeights w, const float* __restrict__ rows, int R,
                                                            const float* __restrict__ cols, int M,
                                                            unsigned short* __restrict__ Ab, float* __restrict__ ur,
                                                            float* __restrict__ rng, unsigned short* __restrict__ Cb) {
    ntn_prep_body<false, 3>(w, rows, R, cols, M, Ab, ur, rng, Cb, (int)blockIdx.x);
}

// several independent rectangles in one launch (sgpr_score_all_pairs_multi): job j owns the prep workgroups
// [block0[j], block0[j+1]) and the work items [item0[j], item0[j+1]) of the main kernel
constexpr int AP_MAX_JOBS = 8;
struct ApJob {
    const float* rows;
    const float* cols;
    float* score;
    int64_t ld;
    unsigned short* Ab;
    unsigned short* Cb;
    float* ur;
    float* rng;
    int R, M, nrng;
};
struct ApJobs {
    int n;
    int block0[AP_MAX_JOBS + 1];
    int item0[AP_MAX_JOBS + 1];
    ApJob job[AP_MAX_JOBS];
};

__global__ __launch_bounds__(256) void ntn_prep_multi_kernel(const DevWeights w, const ApJobs jobs) {
    int j = 0;
    while (j + 1 < jobs.n && (int)blockIdx.x >= jobs.block0[j + 1]) ++j;
    const ApJob& q = jobs.job[j];
    ntn_prep_body<false>(w, q.rows, q.R, q.cols, q.M, q.Ab, q.ur, q.rng, q.Cb, (int)blockIdx.x - jobs.block0[j]);
}

__device__ __forceinline__ f32x4 mfma_f16(f16x8 a, f16x8 b, f32x4 c) {
    // 16x16x32: D[4*(l>>4)+r][l&15] += sum_k A[row][k] B[k][col]; lane l supplies A[l&15][8*(l>>4) .. +7] and
    // B[8*(l>>4) .. +7][l&15]
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

// v_permlane32_swap: lanes 32-63 of the first operand trade places with lanes 0-31 of the second; the sum of the two
// results is, in the lower half, a[l] + a[l+32] and, in the upper half, b[l-32] + b[l]
__device__ __forceinline__ float swap32_add(float a, float b) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// v_permlane16_swap: odd 16-lane rows of the first operand trade places with even rows of the second; the sum is
// a[l] + a[l+16] in even rows and b[l-16] + b[l] in odd rows
__device__ __forceinline__ float swap16_add(float a, float b) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// ReLU as ONE instruction: fmaxf (and fmed3 with an infinite bound, which the optimizer folds back into it) first
// canonicalises the operand with a second v_max x, x.  A signed-integer max with 0 does the same job on the bit
// pattern: every negative float (and -0) is a negative integer.
__device__ __forceinline__ float relu(float x) { return __int_as_float(max(__float_as_int(x), 0)); }

// relu(h[0..3]) -> the layer-2 B operand {hi01, hi23, lo01, lo23} (packed f16 planes, hi + lo = relu(h) to 22 bits):
//   hi = the values truncated to f16 (v_cvt_pkrtz_f16_f32: |hi| <= |h|, same sign)
//   lo = f16(h - hi) straight out of one mixed-precision FMA per value (v_fma_mixlo/mixhi_f16: f16 * f32 + f32, the f16
//        result written to one half of the destination)
//   truncation makes both planes carry h's sign, so the ReLU is one packed SIGNED-INTEGER max with 0 per plane pair
//   (v_pk_max_i16: a negative f16 is a negative int16; no canonicalising pre-pass like the float max gets).
// Only the four mix instructions are inline asm (the compiler has no builtin for them): the conversions before and the
// maxima after are visible instructions, so the wait states an MFMA result needs before a vector read and a vector
// result needs before an MFMA read are inserted by the compiler - it does not look inside inline asm (a first version
// with everything in asm read half-written accumulators).
typedef short i16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
// CL: the low plane's ReLU as the `clamp` bit of the instruction that forms it.  lo = h - hi lies in [0, ulp(hi)) for
// h >= 0 and in (-ulp(hi), 0] for h < 0 (hi truncates towards zero), and clamp cuts to [0, 1]: exact whenever ulp(hi) <= 1,
// i.e. |h| < 2048 - which the launch guarantees through a bound on |H| (ap_mode) before it picks this variant.
template <bool CL>
__device__ __forceinline__ f16x8 split_relu4(f32x4 h) {
    const unsigned h01 = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(h[0], h[1]));
    const unsigned h23 = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(h[2], h[3]));
    unsigned l01, l23;
    const i16x2 z = {0, 0};
    const unsigned a = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(i16x2, h01), z));
    const unsigned b = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(i16x2, h23), z));
    if constexpr (CL) {
        asm("v_fma_mixlo_f16 %0, %2, -1.0, %4 op_sel_hi:[1,0,0] clamp\n\t"
            "v_fma_mixlo_f16 %1, %3, -1.0, %6 op_sel_hi:[1,0,0] clamp\n\t"
            "v_fma_mixhi_f16 %0, %2, -1.0, %5 op_sel:[1,0,0] op_sel_hi:[1,0,0] clamp\n\t"
            "v_fma_mixhi_f16 %1, %3, -1.0, %7 op_sel:[1,0,0] op_sel_hi:[1,0,0] clamp"
            : "=&v"(l01), "=&v"(l23)
            : "v"(h01), "v"(h23), "v"(h[0]), "v"(h[1]), "v"(h[2]), "v"(h[3]));
        return __builtin_bit_cast(f16x8, u32x4{a, b, l01, l23});
    } else {
        asm("v_fma_mixlo_f16 %0, %2, -1.0, %4 op_sel_hi:[1,0,0]\n\t"
            "v_fma_mixlo_f16 %1, %3, -1.0, %6 op_sel_hi:[1,0,0]\n\t"
            "v_fma_mixhi_f16 %0, %2, -1.0, %5 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
            "v_fma_mixhi_f16 %1, %3, -1.0, %7 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
            : "=&v"(l01), "=&v"(l23)
            : "v"(h01), "v"(h23), "v"(h[0]), "v"(h[1]), "v"(h[2]), "v"(h[3]));
        const unsigned c = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(i16x2, l01), z));
        const unsigned d = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(i16x2, l23), z));
        return __builtin_bit_cast(f16x8, u32x4{a, b, c, d});
    }
}

// Exact fp32 evaluation of the rectangle [r0, r1) x [c0, c1), one pair per wave iteration, for inputs outside the f16
// range.  Deliberately rolled loops: it shares the kernel with the hot loop and must not cost it registers.
__device__ __forceinline__ float slow_pair(const DevWeights& w, const float* __restrict__ e1, const float* __restrict__ e2) {
    const int lane = threadIdx.x & 63, t = lane & 15, q = lane >> 4;
    float s = 0.f;
#pragma unroll 1
    for (int j = 8 * q; j < 8 * q + 8; ++j) {                     // this lane group's quarter of the 32 j
        float v = 0.f;
#pragma unroll 1
        for (int i = 0; i < F; ++i) v = fmaf(e1[i], w.ntn_w[i * (F * T) + j * T + t], v);
        s = fmaf(v, e2[j], s);
    }
#pragma unroll 1
    for (int m = 16 * q; m < 16 * q + 16; ++m) s = fmaf(w.ntn_wb[t * 2 * F + m], m < F ? e1[m] : e2[m - F], s);
    s += __shfl_xor(s, 16);
    s += __shfl_xor(s, 32);
    const float h = fmaxf(s + w.ntn_bias[t], 0.f);
    float gacc = w.fc1_b[t];
#pragma unroll 1
    for (int tt = 0; tt < T; ++tt) gacc = fmaf(w.fc1_w[t * T + tt], __shfl(h, tt), gacc);
    float z = fmaxf(gacc, 0.f) * w.fc2_w[t];
    z += __shfl_xor(z, 1);
    z += __shfl_xor(z, 2);
    z += __shfl_xor(z, 4);
    z += __shfl_xor(z, 8);
    return 1.f / (1.f + expf(-(z + w.fc2_b[0])));
}

__device__ __forceinline__ void slow_tile(const DevWeights& w, const float* __restrict__ prow,
                                          const float* __restrict__ pcol, int r0, int r1, int c0, int c1,
                                          float* __restrict__ score, int64_t ld) {
    const int lane = threadIdx.x & 63;
#pragma unroll 1
    for (int r = r0; r < r1; ++r)
#pragma unroll 1
        for (int c = c0; c < c1; ++c) {
            const float sc = slow_pair(w, prow + (size_t)r * F, pcol + (size_t)c * F);
            if (lane == 0) score[(size_t)r * ld + c] = sc;
        }
}

// One wave owns AP_RW = 4 row graphs - their A' operands (two f16 planes) are fetched once per work item and kept in
// registers - and walks super-blocks of 64 column graphs, whose operands (0.6 MB in total: L2-resident) are fetched
// one 16-column block ahead.  Per (row, block of 16 columns):
//   layer 1  H[t][c] = relu(u_r[t] + sum_j A'_r[t][j] e2_c[j]),  K = 32: three f16 MFMAs (lo.hi, hi.lo, hi.hi) on the
//            accumulator initialised with u_r - no vector instruction at all
//   layer 2  G[o][c] = relu(b1[o] + sum_t W1[o][t] H[t][c])   two f16 MFMAs: H is consumed straight from the
//            accumulator layout (lane group g holds t = 4g..4g+3), split into two f16 planes; the K slots
//            8g..8g+3 / 8g+4..8g+7 carry hi / lo, the A operand is W1 laid out to match
//   head     z[c] = b2 + sum_o w2[o] G[o][c]: w2 is folded into layer 2 (ap_consts), so a lane's four terms are four
//            v_med3_f32 of the accumulator against its own side of zero and three adds; then per super-block a lane-swap
//            transpose-reduce over the 4 lane groups leaves lane (g, l15) with row g, columns 4 l15 .. 4 l15 + 3:
//            sigmoid, one 16-byte store.
// OCC = resident workgroups per CU the instance is compiled for; NI = row graphs whose dependent MFMA -> vector ->
// MFMA -> vector chains are interleaved in program order (1, 2 or 4); VAR = timing experiments only (tools/probes):
// bit 1 drops the stores, bit 2 the operand loads of the next block, bit 4 (16) the matrix instructions, bit 5 (32) the
// vector work between them
// per-wave constants of the tail: the head's weights in the lane layout of the MFMA results
struct ApConsts {
    f16x8 w1hi, w1lo;
    float4 b1v, side;
    float nb2;
};

// The head's weight w2[o] is folded into layer 2 (row o of W1 and b1[o] scaled by it - in fp32, before the planes are
// cut), so that the accumulator holds q''[o] = w2[o] (b1[o] + W1[o].H) and the pair's term w2[o] relu(q[o]) is q''[o]
// clamped to its own side of zero: max(q'', 0) for w2 > 0, min(q'', 0) for w2 < 0 = ONE v_med3_f32 against
// (0, side[o]), side = +inf / -inf - instead of an integer maximum and a multiply-add per value.
__device__ __forceinline__ ApConsts ap_consts(const DevWeights& w, int l15, int g) {
    ApConsts c;
    const float s = w.fc2_w[l15];                                                      // the A operand's row is o = l15
    float4 w1v = *reinterpret_cast<const float4*>(w.fc1_w + l15 * T + 4 * g);         // W1[o = l15][t = 4g..4g+3]
    w1v = make_float4(s * w1v.x, s * w1v.y, s * w1v.z, s * w1v.w);
    const _Float16 wh0 = (_Float16)w1v.x, wh1 = (_Float16)w1v.y, wh2 = (_Float16)w1v.z, wh3 = (_Float16)w1v.w;
    const _Float16 z16 = (_Float16)0.f;
    c.w1hi = f16x8{wh0, wh1, wh2, wh3, wh0, wh1, wh2, wh3};                      // meets H's hi and lo planes
    c.w1lo = f16x8{(_Float16)(w1v.x - (float)wh0), (_Float16)(w1v.y - (float)wh1), (_Float16)(w1v.z - (float)wh2),
                        (_Float16)(w1v.w - (float)wh3), z16, z16, z16, z16};            // meets the hi plane only
    const float4 b1 = *reinterpret_cast<const float4*>(w.fc1_b + 4 * g);              // the accumulator's rows are o = 4g + r
    const float4 w2 = *reinterpret_cast<const float4*>(w.fc2_w + 4 * g);
    c.b1v = make_float4(w2.x * b1.x, w2.y * b1.y, w2.z * b1.z, w2.w * b1.w);
    c.side = make_float4(w2.x < 0.f ? -INFINITY : INFINITY, w2.y < 0.f ? -INFINITY : INFINITY,
                         w2.z < 0.f ? -INFINITY : INFINITY, w2.w < 0.f ? -INFINITY : INFINITY);
    c.nb2 = -w.fc2_b[0] * 1.4426950408889634f;
    return c;
}

// max |A'|, max |u|, max |e2| partials -> can every f16 the launch forms be represented?
// The whole workgroup reads the partials, four independent 16-byte loads per thread and round (a wave reading them alone,
// one dependent load per loop trip, spent ~0.7 us per 64 partials before its first matrix instruction: 6 us of the
// KITTI-00 tail, 25 us of a 12 k-graph pair list); the waves' maxima meet in LDS.  Every wave returns the same values.
__device__ __forceinline__ void ap_range(const float* __restrict__ rng, int nrng, float& am, float& um, float& em, float& l1) {
    __shared__ float4 part[4];
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i = threadIdx.x; i < nrng; i += 4 * 256) {
        const float4 v0 = *reinterpret_cast<const float4*>(rng + (size_t)i * 4);
        const float4 v1 = i + 256 < nrng ? *reinterpret_cast<const float4*>(rng + (size_t)(i + 256) * 4) : z;
        const float4 v2 = i + 512 < nrng ? *reinterpret_cast<const float4*>(rng + (size_t)(i + 512) * 4) : z;
        const float4 v3 = i + 768 < nrng ? *reinterpret_cast<const float4*>(rng + (size_t)(i + 768) * 4) : z;
        am = fmaxf(fmaxf(am, v0.x), fmaxf(fmaxf(v1.x, v2.x), v3.x));
        um = fmaxf(fmaxf(um, v0.y), fmaxf(fmaxf(v1.y, v2.y), v3.y));
        em = fmaxf(fmaxf(em, v0.z), fmaxf(fmaxf(v1.z, v2.z), v3.z));
        l1 = fmaxf(fmaxf(l1, v0.w), fmaxf(fmaxf(v1.w, v2.w), v3.w));
    }
    am = wave_max_f32(am);
    um = wave_max_f32(um);
    em = wave_max_f32(em);
    l1 = wave_max_f32(l1);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = make_float4(am, um, em, l1);
    __syncthreads();
    const float4 p0 = part[0], p1 = part[1], p2 = part[2], p3 = part[3];
    am = fmaxf(fmaxf(p0.x, p1.x), fmaxf(p2.x, p3.x));
    um = fmaxf(fmaxf(p0.y, p1.y), fmaxf(p2.y, p3.y));
    em = fmaxf(fmaxf(p0.z, p1.z), fmaxf(p2.z, p3.z));
    l1 = fmaxf(fmaxf(p0.w, p1.w), fmaxf(p2.w, p3.w));
    __syncthreads();                                      // (the multi-job kernel calls this once per rectangle)
}
// 0: exact fp32 per-pair path (an f16 of the launch could overflow); 1: the f16-plane path; 2: the same with the low
// plane's ReLU folded into its conversion (split_relu4<true>): |H[t]| <= |u[t]| + sum_j |A'[t][j]| max|e2| < 1024
__device__ __forceinline__ int ap_mode(float am, float um, float em, float l1) {
    if (!((am < AP_F16_SAFE) && (em < AP_F16_SAFE) && (um + 32.f * am * em < AP_F16_SAFE))) return 0;
    return (um + l1 * em < 1024.f) ? 2 : 1;
}

// the work items [it0, it1) of one R x M rectangle
template <int NI, int VAR, bool CL>
__device__ __forceinline__ void ap_items(const DevWeights& w, const ApConsts& k, const bool fast, int R, int M,
                                         const unsigned short* __restrict__ Ab, const unsigned short* __restrict__ Cb,
                                         const float* __restrict__ ur, const float* __restrict__ prow,
                                         const float* __restrict__ pcol, float* __restrict__ score, int64_t ld,
                                         const int it0, const int it1) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const f16x8 w1hi = k.w1hi, w1lo = k.w1lo;
    const float4 b1v = k.b1v, side = k.side;
    const float kL2E = 1.4426950408889634f;
    const float nb2 = k.nb2;
    const int ncc = (M + AP_COLS - 1) / AP_COLS;
    f16x8 ah[AP_RW], al[AP_RW];
    f32x4 u4[AP_RW];
    int cur_rg = -1, rbase = 0;
    const int nsb = (M + AP_SB - 1) / AP_SB;
    for (int it = it0; it < it1; ++it) {
        const int rg = it / ncc, cc = it - rg * ncc;
        if (rg != cur_rg) {
            cur_rg = rg;
            rbase = rg * AP_ROWS + wave * AP_RW;
#pragma unroll
            for (int rr = 0; rr < AP_RW; ++rr) {
                const int r = min(rbase + rr, R - 1);
                const unsigned short* ap = Ab + ((size_t)r * 2 * 64 + lane) * 8;      // A'_r[t = l15][8g .. 8g+7]
                ah[rr] = *reinterpret_cast<const f16x8*>(ap);
                al[rr] = *reinterpret_cast<const f16x8*>(ap + 64 * 8);
                const float4 u = *reinterpret_cast<const float4*>(ur + (size_t)r * T + 4 * g);
                u4[rr] = f32x4{u.x, u.y, u.z, u.w};
            }
        }
        const int sb0 = cc * (AP_COLS / AP_SB), sb1 = min(nsb, sb0 + AP_COLS / AP_SB);
        if (rbase >= R) continue;                          // this wave's rows lie past the matrix edge
        if (!fast) {      // inputs outside the f16 range: exact fp32 per-pair arithmetic
            slow_tile(w, prow, pcol, rbase, min(R, rbase + AP_RW), sb0 * AP_SB, min(M, sb1 * AP_SB), score, ld);
            continue;
        }
        // column operands of block (sb, b): e2_c[8g .. 8g+7], c = 64 sb + 4 l15 + b; 1 KB contiguous per wave and plane,
        // straight from L2 / L1 (the four waves of a workgroup read the same blocks).  Staging them through LDS once per
        // workgroup was measured and is no faster (112 vs 108 us): the kernel is bound by vector issue, not by operands.
        const unsigned short* cp = Cb + (size_t)sb0 * (2 * 4 * 64 * 8) + (size_t)lane * 8;
        f16x8 bh = *reinterpret_cast<const f16x8*>(cp);
        f16x8 bl = *reinterpret_cast<const f16x8*>(cp + 4 * 64 * 8);
        for (int sb = sb0; sb < sb1; ++sb) {
            float zb[4][AP_RW];
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                // next block: b + 1 of this super-block, or block 0 of the next one (the last one re-reads itself)
                const int nb = b + 1 < 4 ? b + 1 : 0;
                const int nsbk = b + 1 < 4 ? sb : min(sb + 1, sb1 - 1);
                const unsigned short* np = Cb + ((size_t)nsbk * 2 * 4 + nb) * (64 * 8) + (size_t)lane * 8;
                const f16x8 nbh = (VAR & 4) ? bh : *reinterpret_cast<const f16x8*>(np);
                const f16x8 nbl = (VAR & 4) ? bl : *reinterpret_cast<const f16x8*>(np + 4 * 64 * 8);
#pragma unroll
                for (int r0 = 0; r0 < AP_RW; r0 += NI) {
                    f32x4 h[NI], q[NI];
                    f16x8 hb[NI];
                    if (VAR & 8) __builtin_amdgcn_s_setprio(1);
                    if (VAR & 16) {                      // timing: no matrix instructions (opaque copies keep the vector work alive)
#pragma unroll
                        for (int i = 0; i < NI; ++i) {
                            h[i] = u4[r0 + i];
                            asm volatile("" : "+v"(h[i]) : "v"(bh), "v"(bl));
                        }
                    } else {
#if SGPR_AP_CHAINS == 2
                        f32x4 hc[NI];
#pragma unroll
                        for (int i = 0; i < NI; ++i) hc[i] = mfma_f16(al[r0 + i], bh, f32x4{0.f, 0.f, 0.f, 0.f});
#pragma unroll
                        for (int i = 0; i < NI; ++i) h[i] = mfma_f16(ah[r0 + i], bh, u4[r0 + i]);
#pragma unroll
                        for (int i = 0; i < NI; ++i) hc[i] = mfma_f16(ah[r0 + i], bl, hc[i]);
#pragma unroll
                        for (int i = 0; i < NI; ++i) h[i] = h[i] + hc[i];
#else
#pragma unroll
                        for (int i = 0; i < NI; ++i) h[i] = mfma_f16(al[r0 + i], bh, u4[r0 + i]);
#pragma unroll
                        for (int i = 0; i < NI; ++i) h[i] = mfma_f16(ah[r0 + i], bl, h[i]);
#pragma unroll
                        for (int i = 0; i < NI; ++i) h[i] = mfma_f16(ah[r0 + i], bh, h[i]);
#endif
                    }
                    if (VAR & 8) __builtin_amdgcn_s_setprio(0);
#pragma unroll
                    for (int i = 0; i < NI; ++i) hb[i] = (VAR & 32) ? __builtin_bit_cast(f16x8, h[i]) : split_relu4<CL>(h[i]);
                    if (VAR & 8) __builtin_amdgcn_s_setprio(1);
                    if (VAR & 16) {
#pragma unroll
                        for (int i = 0; i < NI; ++i) {
                            q[i] = f32x4{b1v.x, b1v.y, b1v.z, b1v.w};
                            asm volatile("" : "+v"(q[i]) : "v"(hb[i]));
                        }
                    } else {
#pragma unroll
                        for (int i = 0; i < NI; ++i) q[i] = mfma_f16(w1lo, hb[i], f32x4{b1v.x, b1v.y, b1v.z, b1v.w});   // hi . W1lo
#pragma unroll
                        for (int i = 0; i < NI; ++i) q[i] = mfma_f16(w1hi, hb[i], q[i]);                                 // (hi + lo) . W1hi
                    }
                    if (VAR & 8) __builtin_amdgcn_s_setprio(0);
#pragma unroll
                    for (int i = 0; i < NI; ++i) {
                        if (VAR & 32) {
                            zb[b][r0 + i] = q[i][0];
                            continue;
                        }
                        const float t0 = __builtin_amdgcn_fmed3f(q[i][0], 0.f, side.x), t1 = __builtin_amdgcn_fmed3f(q[i][1], 0.f, side.y);
                        const float t2 = __builtin_amdgcn_fmed3f(q[i][2], 0.f, side.z), t3 = __builtin_amdgcn_fmed3f(q[i][3], 0.f, side.w);
                        zb[b][r0 + i] = (t0 + t1) + (t2 + t3);           // partial over o = 4g..4g+3 of row r0+i, column 4 l15 + b
                    }
                }
                bh = nbh;
                bl = nbl;
            }
            // transpose-reduce over the four lane groups with the gfx950 lane-swap ops: lane group g ends up with the
            // full sum of row g (3 swaps + 3 adds per column block instead of 8 bpermutes), for its 4 columns
            float sc[4];
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const float p02 = swap32_add(zb[b][0], zb[b][2]);    // lanes 0-31: row 0 over groups {g, g+2}; 32-63: row 2
                const float p13 = swap32_add(zb[b][1], zb[b][3]);    // likewise rows 1 / 3
                const float zsel = swap16_add(p02, p13);             // even 16-lane rows: row 0 / 2, odd: row 1 / 3
                // sigmoid: v_exp_f32 / v_rcp_f32 (1 ulp each) - far inside the 1e-4 score tolerance
                sc[b] = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(fmaf(zsel, -kL2E, nb2)));
            }
            const int r = rbase + g, c0 = sb * AP_SB + 4 * l15;
            if ((VAR & 2) && sc[0] + sc[1] + sc[2] + sc[3] != 12345.678f) continue;
            if (r < R) {
                float* dst = score + (size_t)r * ld + c0;
                if (c0 + 3 < M) {
                    typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
#if SGPR_AP_NT_STORE
                    __builtin_nontemporal_store(f32x4u{sc[0], sc[1], sc[2], sc[3]}, reinterpret_cast<f32x4u*>(dst));
#else
                    *reinterpret_cast<f32x4u*>(dst) = f32x4u{sc[0], sc[1], sc[2], sc[3]};
#endif
                } else {
#pragma unroll
                    for (int b = 0; b < 4; ++b)
                        if (c0 + b < M) dst[b] = sc[b];
                }
            }
        }
    }
}

template <int OCC, int NI, int VAR>
__global__ __launch_bounds__(256, OCC) void score_all_pairs_kernel(const DevWeights w, int R, int M,
                                                              const unsigned short* __restrict__ Ab,
                                                              const unsigned short* __restrict__ Cb,
                                                              const float* __restrict__ ur,
                                                              const float* __restrict__ rng, int nrng,
                                                              const float* __restrict__ prow,
                                                              const float* __restrict__ pcol,
                                                              float* __restrict__ score, int64_t ld) {
    const int lane = threadIdx.x & 63;
    const int l15 = lane & 15, g = lane >> 4;
    // ---- can every f16 this launch forms be represented?  |A'|, |e2| and |H| <= |u| + 32 max|A'| max|e2|
    float am = 0.f, um = 0.f, em = 0.f, l1 = 0.f;
    ap_range(rng, nrng, am, um, em, l1);
    const int mode = ap_mode(am, um, em, l1);
    const ApConsts k = ap_consts(w, l15, g);
    // work items = (row group of AP_ROWS, column chunk of AP_COLS), row-major; every workgroup takes a contiguous,
    // equally long range (the grid is sized to one resident slot per workgroup, so there is no second,
    // under-occupied round) and reloads its row operands only when the row group changes
    const int ncc = (M + AP_COLS - 1) / AP_COLS;
    const int64_t items = (int64_t)ncc * ((R + AP_ROWS - 1) / AP_ROWS);
    // workgroups go to the 8 XCDs round-robin and every XCD has its own L2: the workgroups of ONE XCD take neighbouring
    // ranges, so that a row group's A' operands (shared by the ~4 workgroups that split its column chunks) are fetched
    // through one L2 instead of four (FETCH_SIZE 37 -> see profiles)
    const unsigned nwg = gridDim.x;
    const unsigned wg = (nwg & 7u) == 0u ? (blockIdx.x & 7u) * (nwg >> 3) + (blockIdx.x >> 3) : blockIdx.x;
    const int it0 = (int)(items * wg / nwg), it1 = (int)(items * (wg + 1) / nwg);
    if (mode == 2)
        ap_items<NI, VAR, true>(w, k, true, R, M, Ab, Cb, ur, prow, pcol, score, ld, it0, it1);
    else
        ap_items<NI, VAR, false>(w, k, mode != 0, R, M, Ab, Cb, ur, prow, pcol, score, ld, it0, it1);
}

// the same for several rectangles: the work items of all jobs form one row-major list that the workgroups split evenly
template <int OCC, int NI>
__global__ __launch_bounds__(256, OCC) void score_all_pairs_multi_kernel(const DevWeights w, const ApJobs jobs) {
    const int lane = threadIdx.x & 63;
    const int l15 = lane & 15, g = lane >> 4;
    const ApConsts k = ap_consts(w, l15, g);
    const int64_t items = jobs.item0[jobs.n];
    const unsigned nwg = gridDim.x;
    const unsigned wg = (nwg & 7u) == 0u ? (blockIdx.x & 7u) * (nwg >> 3) + (blockIdx.x >> 3) : blockIdx.x;
    const int g0 = (int)(items * wg / nwg), g1 = (int)(items * (wg + 1) / nwg);
#pragma unroll 1
    for (int j = 0; j < jobs.n; ++j) {
        const int lo = max(g0, jobs.item0[j]) - jobs.item0[j], hi = min(g1, jobs.item0[j + 1]) - jobs.item0[j];
        if (lo >= hi) continue;
        const ApJob& q = jobs.job[j];
        float am = 0.f, um = 0.f, em = 0.f, l1 = 0.f;        // the f16 range question is answered per rectangle, like
        ap_range(q.rng, q.nrng, am, um, em, l1);             // a call of its own would
        const int mode = ap_mode(am, um, em, l1);
        if (mode == 2)
            ap_items<NI, 0, true>(w, k, true, q.R, q.M, q.Ab, q.Cb, q.ur, q.rows, q.cols, q.score, q.ld, lo, hi);
        else
            ap_items<NI, 0, false>(w, k, mode != 0, q.R, q.M, q.Ab, q.Cb, q.ur, q.rows, q.cols, q.score, q.ld, lo, hi);
    }
}

// ------------------------------------------------------------------ dense all-pairs at the reference's operand width
// The same rectangle with every matrix operand as THREE bf16 planes (x = hi + mid + lo exactly: 24 bits, fp32's range) on
// v_mfma_f32_16x16x32_bf16 - the tail of layers_batch.py:70-83 / sg_net.py:131-136 at fp32's own operand width, as the
// embed kernel's wide-range instance is for dgcnn_conv_pass.  Selected per handle (weights outside the f16 range) or by
// debug bit 13, never by the data (out-of-range data keeps the exact per-pair path of score_all_pairs_kernel).
//   layer 1  six significant cross products (lo.hi, hi.lo, mid.mid, mid.hi, hi.mid in a chain of their own - smallest
//            first, never against the large accumulator -, then hi.hi on u_r), one vector add
//   layer 2  H = relu(.) cut into three planes by truncation (upper halves of x, x - hi, x - hi - mid: same sign, so the
//            ReLU is a signed maximum of the fp32 word before the cut); the eight K slots of a lane group carry two planes
//            per instruction: [Hh | Hm].[W1h | W1h], [Hl | Hh].[W1h | W1m], [Hm | Hh].[W1m | W1l] - again six products
// Everything else (u_r, the folded head, the lane-swap transpose-reduce, 16-byte stores, the work split) is the f16
// instance's.  Nine matrix instructions and ~45 vector instructions per (row, 16 columns) against five and ~24.
typedef short bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x4 mfma_bf16(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
// {upper half of b, upper half of a} -> one dword: K slot 2i in the low half, 2i + 1 in the high half
__device__ __forceinline__ unsigned pack_hi16(unsigned a, unsigned b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }

#ifndef SGPR_APW_OCC
#define SGPR_APW_OCC 3
#endif
#ifndef SGPR_APW_NI
#define SGPR_APW_NI 1         // (2: same time, same box - 226.9 / 230.1 against 226.9 / 226.5 us - and 12 registers more)
#endif
constexpr int APW_NI = SGPR_APW_NI;       // row graphs interleaved in program order (1, 2 or 4)
constexpr int APW_OCC = SGPR_APW_OCC;     // (three row-operand planes of four row graphs: 48 registers; three workgroups per CU)

__global__ __launch_bounds__(256, APW_OCC) void score_all_pairs_wide_kernel(const DevWeights w, int R, int M,
                                                                            const unsigned short* __restrict__ Ab,
                                                                            const unsigned short* __restrict__ Cb,
                                                                            const float* __restrict__ ur,
                                                                            float* __restrict__ score, int64_t ld) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    // the head folded into layer 2 (ap_consts), W1 as three planes in the three slot arrangements
    bf16x8 wa, wb2, wc;
    float4 b1v, side;
    {
        const float s = w.fc2_w[l15];
        const float4 w1 = *reinterpret_cast<const float4*>(w.fc1_w + l15 * T + 4 * g);
        const float v[4] = {s * w1.x, s * w1.y, s * w1.z, s * w1.w};
        unsigned hb[4], mb[4], lb[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) split3_bf16(v[i], hb[i], mb[i], lb[i]);
        const unsigned h01 = pack_hi16(hb[0], hb[1]), h23 = pack_hi16(hb[2], hb[3]);
        const unsigned m01 = pack_hi16(mb[0], mb[1]), m23 = pack_hi16(mb[2], mb[3]);
        const unsigned l01 = pack_hi16(lb[0], lb[1]), l23 = pack_hi16(lb[2], lb[3]);
        wa = __builtin_bit_cast(bf16x8, u32x4{h01, h23, h01, h23});
        wb2 = __builtin_bit_cast(bf16x8, u32x4{h01, h23, m01, m23});
        wc = __builtin_bit_cast(bf16x8, u32x4{m01, m23, l01, l23});
        const float4 b1 = *reinterpret_cast<const float4*>(w.fc1_b + 4 * g);
        const float4 w2 = *reinterpret_cast<const float4*>(w.fc2_w + 4 * g);
        b1v = make_float4(w2.x * b1.x, w2.y * b1.y, w2.z * b1.z, w2.w * b1.w);
        side = make_float4(w2.x < 0.f ? -INFINITY : INFINITY, w2.y < 0.f ? -INFINITY : INFINITY,
                           w2.z < 0.f ? -INFINITY : INFINITY, w2.w < 0.f ? -INFINITY : INFINITY);
    }
    const float kL2E = 1.4426950408889634f;
    const float nb2 = -w.fc2_b[0] * kL2E;
    const int ncc = (M + AP_COLS - 1) / AP_COLS;
    const int64_t items = (int64_t)ncc * ((R + AP_ROWS - 1) / AP_ROWS);
    const unsigned nwg = gridDim.x;
    const unsigned wg = (nwg & 7u) == 0u ? (blockIdx.x & 7u) * (nwg >> 3) + (blockIdx.x >> 3) : blockIdx.x;
    const int it0 = (int)(items * wg / nwg), it1 = (int)(items * (wg + 1) / nwg);
    const int nsb = (M + AP_SB - 1) / AP_SB;
    bf16x8 ah[AP_RW], am[AP_RW], al[AP_RW];
    f32x4 u4[AP_RW];
    int cur_rg = -1, rbase = 0;
    for (int it = it0; it < it1; ++it) {
        const int rg = it / ncc, cc = it - rg * ncc;
        if (rg != cur_rg) {
            cur_rg = rg;
            rbase = rg * AP_ROWS + wave * AP_RW;
#pragma unroll
            for (int rr = 0; rr < AP_RW; ++rr) {
                const int r = min(rbase + rr, R - 1);
                const unsigned short* ap = Ab + ((size_t)r * 3 * 64 + lane) * 8;      // A'_r[t = l15][8g .. 8g+7]
                ah[rr] = *reinterpret_cast<const bf16x8*>(ap);
                am[rr] = *reinterpret_cast<const bf16x8*>(ap + 64 * 8);
                al[rr] = *reinterpret_cast<const bf16x8*>(ap + 2 * 64 * 8);
                const float4 u = *reinterpret_cast<const float4*>(ur + (size_t)r * T + 4 * g);
                u4[rr] = f32x4{u.x, u.y, u.z, u.w};
            }
        }
        const int sb0 = cc * (AP_COLS / AP_SB), sb1 = min(nsb, sb0 + AP_COLS / AP_SB);
        if (rbase >= R) continue;                          // this wave's rows lie past the matrix edge
        const unsigned short* cp = Cb + (size_t)sb0 * (3 * 4 * 64 * 8) + (size_t)lane * 8;
        bf16x8 bh = *reinterpret_cast<const bf16x8*>(cp);
        bf16x8 bm = *reinterpret_cast<const bf16x8*>(cp + 4 * 64 * 8);
        bf16x8 bl = *reinterpret_cast<const bf16x8*>(cp + 2 * 4 * 64 * 8);
        for (int sb = sb0; sb < sb1; ++sb) {
            float zb[4][AP_RW];
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int nb = b + 1 < 4 ? b + 1 : 0;
                const int nsbk = b + 1 < 4 ? sb : min(sb + 1, sb1 - 1);
                const unsigned short* np = Cb + ((size_t)nsbk * 3 * 4 + nb) * (64 * 8) + (size_t)lane * 8;
                const bf16x8 nbh = *reinterpret_cast<const bf16x8*>(np);
                const bf16x8 nbm = *reinterpret_cast<const bf16x8*>(np + 4 * 64 * 8);
                const bf16x8 nbl = *reinterpret_cast<const bf16x8*>(np + 2 * 4 * 64 * 8);
#pragma unroll
                for (int r0 = 0; r0 < AP_RW; r0 += APW_NI) {
                    // APW_NI row graphs interleaved in program order: their dependent MFMA -> vector -> MFMA chains overlap
                    f32x4 c[APW_NI], h[APW_NI], qc[APW_NI], q[APW_NI];
                    unsigned h01[APW_NI], h23[APW_NI], m01[APW_NI], m23[APW_NI], l01[APW_NI], l23[APW_NI];
#pragma unroll
                    for (int i = 0; i < APW_NI; ++i) c[i] = mfma_bf16(al[r0 + i], bh, f32x4{0.f, 0.f, 0.f, 0.f});
#pragma unroll
                    for (int i = 0; i < APW_NI; ++i) h[i] = mfma_bf16(ah[r0 + i], bh, u4[r0 + i]);
#pragma unroll
                    for (int i = 0; i < APW_NI; ++i) c[i] = mfma_bf16(ah[r0 + i], bl, c[i]);
#pragma unroll
                    for (int i = 0; i < APW_NI; ++i) c[i] = mfma_bf16(am[r0 + i], bm, c[i]);
#pragma unroll
                    for (int i = 0; i < APW_NI; ++i) c[i] = mfma_bf16(am[r0 + i], bh, c[i]);
#pragma unroll
                    for (int i = 0; i < APW_NI; ++i) c[i] = mfma_bf16(ah[r0 + i], bm, c[i]);
#pragma unroll
                    for (int i = 0; i < APW_NI; ++i) {
                        h[i] = h[i] + c[i];
                        // relu, then the three planes of each of the lane's four values (t = 4g .. 4g+3)
                        unsigned hb[4], mb[4], lb[4];
#pragma unroll
                        for (int v = 0; v < 4; ++v) split3_bf16(relu(h[i][v]), hb[v], mb[v], lb[v]);
                        h01[i] = pack_hi16(hb[0], hb[1]);
                        h23[i] = pack_hi16(hb[2], hb[3]);
                        m01[i] = pack_hi16(mb[0], mb[1]);
                        m23[i] = pack_hi16(mb[2], mb[3]);
                        l01[i] = pack_hi16(lb[0], lb[1]);
                        l23[i] = pack_hi16(lb[2], lb[3]);
                    }
#pragma unroll
                    for (int i = 0; i < APW_NI; ++i)                                                       // Hm.W1m + Hh.W1l
                        qc[i] = mfma_bf16(wc, __builtin_bit_cast(bf16x8, u32x4{m01[i], m23[i], h01[i], h23[i]}), f32x4{0.f, 0.f, 0.f, 0.f});
#pragma unroll
                    for (int i = 0; i < APW_NI; ++i)                                                       // Hh.W1h + Hm.W1h
                        q[i] = mfma_bf16(wa, __builtin_bit_cast(bf16x8, u32x4{h01[i], h23[i], m01[i], m23[i]}), f32x4{b1v.x, b1v.y, b1v.z, b1v.w});
#pragma unroll
                    for (int i = 0; i < APW_NI; ++i)                                                       // Hl.W1h + Hh.W1m
                        qc[i] = mfma_bf16(wb2, __builtin_bit_cast(bf16x8, u32x4{l01[i], l23[i], h01[i], h23[i]}), qc[i]);
#pragma unroll
                    for (int i = 0; i < APW_NI; ++i) {
                        q[i] = q[i] + qc[i];
                        const float t0 = __builtin_amdgcn_fmed3f(q[i][0], 0.f, side.x), t1 = __builtin_amdgcn_fmed3f(q[i][1], 0.f, side.y);
                        const float t2 = __builtin_amdgcn_fmed3f(q[i][2], 0.f, side.z), t3 = __builtin_amdgcn_fmed3f(q[i][3], 0.f, side.w);
                        zb[b][r0 + i] = (t0 + t1) + (t2 + t3);
                    }
                }
                bh = nbh;
                bm = nbm;
                bl = nbl;
            }
            float sc[4];
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const float p02 = swap32_add(zb[b][0], zb[b][2]);
                const float p13 = swap32_add(zb[b][1], zb[b][3]);
                const float zsel = swap16_add(p02, p13);
                sc[b] = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(fmaf(zsel, -kL2E, nb2)));
            }
            const int r = rbase + g, c0 = sb * AP_SB + 4 * l15;
            if (r < R) {
                float* dst = score + (size_t)r * ld + c0;
                if (c0 + 3 < M) {
                    typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
                    *reinterpret_cast<f32x4u*>(dst) = f32x4u{sc[0], sc[1], sc[2], sc[3]};
                } else {
#pragma unroll
                    for (int b = 0; b < 4; ++b)
                        if (c0 + b < M) dst[b] = sc[b];
                }
            }
        }
    }
}

int launch_score_all_pairs(const sgpr_handle* h, const float* rows, int R, const float* cols, int M, float* score,
                           int64_t ld, void* ws, hipStream_t stream, bool wide) {
    if (R == 0 || M == 0) return SGPR_OK;
    const int ngroups = ap_prep_groups(R, M), nrng = 2 * ngroups;
    const size_t nsb = (size_t)(M + AP_SB - 1) / AP_SB;
    float* ur = static_cast<float*>(ws);
    float* rng = ur + (size_t)R * T;
    unsigned short* Ab = reinterpret_cast<unsigned short*>(rng + (size_t)nrng * 4);
    unsigned short* Cb = Ab + (size_t)R * (wide ? 3 : 2) * 64 * 8;
    (void)nsb;
    const int64_t items = (int64_t)((M + AP_COLS - 1) / AP_COLS) * ((R + AP_ROWS - 1) / AP_ROWS);
    if (wide) {                                            // three bf16 planes: the reference's operand width, fp32's range
        hipLaunchKernelGGL(ntn_prep_wide_kernel, dim3(nrng), dim3(256), 0, stream, h->w, rows, R, cols, M, Ab, ur, rng, Cb);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return hip_fail(e, "ntn_prep_wide_kernel launch");
        const int64_t slots = (int64_t)h->num_cus * APW_OCC;
        const unsigned grid = (unsigned)(items < slots ? items : slots);
        hipLaunchKernelGGL(score_all_pairs_wide_kernel, dim3(grid), dim3(256), 0, stream, h->w, R, M, Ab, Cb, ur, score, ld);
        e = hipGetLastError();
        if (e != hipSuccess) return hip_fail(e, "score_all_pairs_wide_kernel launch");
        return SGPR_OK;
    }
    hipLaunchKernelGGL(ntn_prep_kernel, dim3(nrng), dim3(256), 0, stream, h->w, rows, R, cols, M, Ab, ur, rng, Cb);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "ntn_prep_kernel launch");
    const int64_t slots = (int64_t)h->num_cus * AP_OCC;   // one resident slot per workgroup: a single, full round
    const unsigned grid = (unsigned)(items < slots ? items : slots);
    hipLaunchKernelGGL((score_all_pairs_kernel<AP_OCC, AP_NI, 0>), dim3(grid), dim3(256), 0, stream, h->w, R, M, Ab, Cb, ur, rng, nrng, rows,
                       cols, score, ld);
    e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "score_all_pairs_kernel launch");
    return SGPR_OK;
}

static size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

size_t score_all_pairs_multi_ws_bytes(int n, const sgpr_pairs_job* jobs) {
    size_t total = 0;
    for (int j = 0; j < n; ++j) total += align256(score_all_pairs_ws_bytes(jobs[j].R, jobs[j].M));
    return total;
}

int launch_score_all_pairs_multi(const sgpr_handle* h, int n, const sgpr_pairs_job* jobs, void* ws, hipStream_t stream) {
    ApJobs a;
    memset(&a, 0, sizeof(a));
    unsigned char* base = static_cast<unsigned char*>(ws);
    int64_t items = 0;
    int blocks = 0;
    for (int j = 0; j < n; ++j) {
        const int R = jobs[j].R, M = jobs[j].M;
        if (R == 0 || M == 0) continue;                    // empty rectangles take no work
        ApJob& q = a.job[a.n];
        const int nrng = 2 * ap_prep_groups(R, M);
        q.rows = jobs[j].d_pooled_rows;
        q.cols = jobs[j].d_pooled_cols;
        q.score = jobs[j].d_score;
        q.ld = jobs[j].ld;
        q.R = R;
        q.M = M;
        q.nrng = nrng;
        q.ur = reinterpret_cast<float*>(base);             // the layout of launch_score_all_pairs, per job
        q.rng = q.ur + (size_t)R * T;
        q.Ab = reinterpret_cast<unsigned short*>(q.rng + (size_t)nrng * 4);
        q.Cb = q.Ab + (size_t)R * 2 * 64 * 8;
        base += align256(score_all_pairs_ws_bytes(R, M));
        a.block0[a.n] = blocks;
        a.item0[a.n] = (int)items;
        blocks += nrng;
        items += (int64_t)((M + AP_COLS - 1) / AP_COLS) * ((R + AP_ROWS - 1) / AP_ROWS);
        if (items > 0x7fffffff) {
            set_error("sgpr_score_all_pairs_multi: more than 2^31 work items");
            return SGPR_E_INVALID;
        }
        ++a.n;
    }
    if (a.n == 0) return SGPR_OK;
    a.block0[a.n] = blocks;
    a.item0[a.n] = (int)items;
    hipLaunchKernelGGL(ntn_prep_multi_kernel, dim3(blocks), dim3(256), 0, stream, h->w, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "ntn_prep_multi_kernel launch");
    const int64_t slots = (int64_t)h->num_cus * AP_MULTI_OCC;
    const unsigned grid = (unsigned)(items < slots ? items : slots);
    // (one row graph at a time here: at four workgroups per CU interleaving two cost this instance - two instantiations
    //  of the loop in one job loop - 380 -> ~900 bytes of scratch per lane, 334 -> 420 us on the five KITTI matrices; at
    //  three per CU nothing spills either way and the two are on par, 358.9 / 360.8 us: same-box A/Bs, round 5)
    hipLaunchKernelGGL((score_all_pairs_multi_kernel<AP_MULTI_OCC, 1>), dim3(grid), dim3(256), 0, stream, h->w, a);
    e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "score_all_pairs_multi_kernel launch");
    return SGPR_OK;
}

// ------------------------------------------------------------------ grouped pair list (sgpr_score_pair_list)
// The reference's own evaluation loop walks a pair LIST (eval_batch.py:30-36: ~15 listed pairs per row graph, 10^4-10^5
// pairs over 10^3 graphs).  The list arrives grouped by row graph (sgpr_pair_plan, host): work item = (row graph, up to 16
// of its listed columns).  ntn_prep hoists A'_r / u_r for the DISTINCT row graphs and the two-plane copies of the column
// graphs exactly as for the dense rectangle; a wave then takes FOUR items at a time through the all-pairs inner loop -
// layer 1 (3 MFMAs), split, layer 2 (2 MFMAs), head - with the columns gathered by index, and the lane-swap
// transpose-reduce that gives lane group g the four rows of one item in the dense kernel gives it item g here.  Every
// instruction and operand of a pair is the one score_all_pairs_kernel would use for that (row, column): the scores are
// bit-identical to the dense matrix's entries.  Against score_pairs_kernel (one wave per pair, the 64 KB NTN weight
// re-read per pair): ~430 VALU + 290 VMEM instructions per pair there, ~3 + 0.5 here.
// workspace:  ur [NR][T] f32 | rng [2 groups][4] f32 | Ab [NR][2][64][8] f16 | Cg [M][2][32] f16
static inline int pl_prep_groups(int NR, int M) { return ((NR > M ? NR : M) + 15) / 16; }

size_t score_pair_list_ws_bytes(int NR, int M) {
    return (size_t)NR * T * sizeof(float) + (size_t)2 * pl_prep_groups(NR, M) * 4 * sizeof(float) +
           (size_t)NR * 2 * 64 * 8 * sizeof(unsigned short) + (size_t)M * 2 * F * sizeof(unsigned short);
}

__global__ __launch_bounds__(256) void ntn_prep_list_kernel(const DevWeights w, const float* __restrict__ rows,
                                                            const int32_t* __restrict__ row_ids, int NR,
                                                            const float* __restrict__ cols, int M,
                                                            unsigned short* __restrict__ Ab, float* __restrict__ ur,
                                                            float* __restrict__ rng, unsigned short* __restrict__ Cg) {
    ntn_prep_body<true>(w, rows, NR, cols, M, Ab, ur, rng, Cg, (int)blockIdx.x, row_ids);
}

struct PairPlan {                 // device views into the plan buffer (sgpr.h, sgpr_pair_plan)
    const int32_t* row_ids;       // [NR]          distinct row graphs, ascending
    const int32_t* item_row;      // [NI]          compact row (index into row_ids) of every work item
    const int32_t* item_beg;      // [NI + 1]      first listed pair of the item in cols / pos (an item holds <= 16)
    const int32_t* cols;          // [P]           column graph of every listed pair, grouped by row graph
    const int32_t* pos;           // [P]           position of that pair in the caller's list = where its score goes
    int NR, NI;
};

template <bool CL>
__device__ __forceinline__ void pl_items(const DevWeights& w, const ApConsts& k, const bool fast, const PairPlan pl,
                                         const unsigned short* __restrict__ Ab, const unsigned short* __restrict__ Cg,
                                         const float* __restrict__ ur, const float* __restrict__ prow,
                                         const float* __restrict__ pcol, float* __restrict__ score) {
    const int lane = threadIdx.x & 63, l15 = lane & 15, g = lane >> 4;
    const f16x8 w1hi = k.w1hi, w1lo = k.w1lo;
    const float4 b1v = k.b1v, side = k.side;
    const float kL2E = 1.4426950408889634f;
    const float nb2 = k.nb2;
    const int nquad = (pl.NI + 3) >> 2;
    const int wave0 = (int)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int)gridDim.x * 4;
    for (int q4 = wave0; q4 < nquad; q4 += nwaves) {
        if (!fast) {        // inputs outside the f16 range: exact fp32 arithmetic, one pair per wave step
#pragma unroll 1
            for (int i = 0; i < 4; ++i) {
                const int item = 4 * q4 + i;
                if (item >= pl.NI) break;
                const int beg = pl.item_beg[item], end = pl.item_beg[item + 1];
                const float* e1 = prow + (size_t)pl.row_ids[pl.item_row[item]] * F;
#pragma unroll 1
                for (int p = beg; p < end; ++p) {
                    const float sc = slow_pair(w, e1, pcol + (size_t)pl.cols[p] * F);
                    if (lane == 0) score[pl.pos[p]] = sc;
                }
            }
            continue;
        }
        float zb[4];
        int my_beg = 0, my_cnt = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int item = min(4 * q4 + i, pl.NI - 1);
            const int beg = pl.item_beg[item];
            const int cnt = (4 * q4 + i < pl.NI) ? pl.item_beg[item + 1] - beg : 0;
            if (i == g) {
                my_beg = beg;
                my_cnt = cnt;
            }
            const int r = pl.item_row[item];
            const int c = pl.cols[beg + min(l15, max(cnt, 1) - 1)];             // slots past the item's end repeat its last pair
            const unsigned short* ap = Ab + ((size_t)r * 2 * 64 + lane) * 8;      // A'_r[t = l15][8g .. 8g+7]
            const f16x8 ah = *reinterpret_cast<const f16x8*>(ap);
            const f16x8 al = *reinterpret_cast<const f16x8*>(ap + 64 * 8);
            const float4 u = *reinterpret_cast<const float4*>(ur + (size_t)r * T + 4 * g);
            const unsigned short* cp = Cg + (size_t)c * (2 * F) + 8 * g;          // e2_c[8g .. 8g+7], column = MFMA column l15
            const f16x8 bh = *reinterpret_cast<const f16x8*>(cp);
            const f16x8 bl = *reinterpret_cast<const f16x8*>(cp + F);
            f32x4 h = mfma_f16(al, bh, f32x4{u.x, u.y, u.z, u.w});
            h = mfma_f16(ah, bl, h);
            h = mfma_f16(ah, bh, h);
            const f16x8 hb = split_relu4<CL>(h);
            f32x4 q = mfma_f16(w1lo, hb, f32x4{b1v.x, b1v.y, b1v.z, b1v.w});
            q = mfma_f16(w1hi, hb, q);
            const float t0 = __builtin_amdgcn_fmed3f(q[0], 0.f, side.x), t1 = __builtin_amdgcn_fmed3f(q[1], 0.f, side.y);
            const float t2 = __builtin_amdgcn_fmed3f(q[2], 0.f, side.z), t3 = __builtin_amdgcn_fmed3f(q[3], 0.f, side.w);
            zb[i] = (t0 + t1) + (t2 + t3);
        }
        const float p02 = swap32_add(zb[0], zb[2]);
        const float p13 = swap32_add(zb[1], zb[3]);
        const float zsel = swap16_add(p02, p13);             // lane group g: item g of the quad, its pair l15
        const float sc = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(fmaf(zsel, -kL2E, nb2)));
        if (l15 < my_cnt) score[pl.pos[my_beg + l15]] = sc;
    }
}

__global__ __launch_bounds__(256) void score_pair_list_kernel(const DevWeights w, const PairPlan pl,
                                                              const unsigned short* __restrict__ Ab,
                                                              const unsigned short* __restrict__ Cg,
                                                              const float* __restrict__ ur,
                                                              const float* __restrict__ rng, int nrng,
                                                              const float* __restrict__ prow,
                                                              const float* __restrict__ pcol,
                                                              float* __restrict__ score) {
    const int lane = threadIdx.x & 63;
    float am = 0.f, um = 0.f, em = 0.f, l1 = 0.f;
    ap_range(rng, nrng, am, um, em, l1);
    const int mode = ap_mode(am, um, em, l1);
    const ApConsts k = ap_consts(w, lane & 15, lane >> 4);
    if (mode == 2)
        pl_items<true>(w, k, true, pl, Ab, Cg, ur, prow, pcol, score);
    else
        pl_items<false>(w, k, mode != 0, pl, Ab, Cg, ur, prow, pcol, score);
}

int launch_score_pair_list(const sgpr_handle* h, const float* rows, const float* cols, int M, const int32_t* plan,
                           int NR, int NI, int64_t P, float* score, void* ws, hipStream_t stream) {
    if (P == 0 || NI == 0) return SGPR_OK;
    PairPlan pl;
    pl.row_ids = plan;
    pl.item_row = pl.row_ids + NR;
    pl.item_beg = pl.item_row + NI;
    pl.cols = pl.item_beg + NI + 1;
    pl.pos = pl.cols + P;
    pl.NR = NR;
    pl.NI = NI;
    const int nrng = 2 * pl_prep_groups(NR, M);
    float* ur = static_cast<float*>(ws);
    float* rng = ur + (size_t)NR * T;
    unsigned short* Ab = reinterpret_cast<unsigned short*>(rng + (size_t)nrng * 4);
    unsigned short* Cg = Ab + (size_t)NR * 2 * 64 * 8;
    hipLaunchKernelGGL(ntn_prep_list_kernel, dim3(nrng), dim3(256), 0, stream, h->w, rows, pl.row_ids, NR, cols, M, Ab, ur,
                       rng, Cg);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "ntn_prep_list_kernel launch");
    const int64_t wgs = ((int64_t)(NI + 3) / 4 + 3) / 4;          // one quad of items per wave, four waves per workgroup
    const int64_t slots = (int64_t)h->num_cus * 8;
    const unsigned grid = (unsigned)(wgs < slots ? wgs : slots);
    hipLaunchKernelGGL(score_pair_list_kernel, dim3(grid), dim3(256), 0, stream, h->w, pl, Ab, Cg, ur, rng, nrng, rows, cols,
                       score);
    e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "score_pair_list_kernel launch");
    return SGPR_OK;
}

}  // namespace sgpr
