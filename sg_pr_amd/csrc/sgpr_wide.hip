// Matrix-core embed for architectures MODERATELY larger than the built shape (the reference builds any
// width: sg_net.py:40-76, parser_sg.py:12-18): labels <= 32, filters_1 / filters_2 <= 128, filters_3 <= 64, node_num <= 112,
// K = 10.  SG.dgcnn_conv_pass + AttentionModule (sg_net.py:79-110, dgcnn.py:14-49, layers_batch.py:28-39) in the
// formulation of the tuned kernels - eval BatchNorm folded, W.[x_j - x_i ; x_i] = W1.x_j + (W2 - W1).x_i, the max over the
// neighbours taken on the first term - with every matrix product on v_mfma_f32_16x16x32_f16 and both operands as two f16
// planes (x = hi + lo, 22 bits), and the organisation of embed_big_kernel (sgpr_embed.hip): one workgroup per graph, one
// wave per 16-row tile; a wave takes the Gram tiles of its rows against every candidate tile from the accumulators (at
// most 7 tiles = 28 keys per lane: all in registers, one pass), selects, computes its own row tile of the per-node
// GEMMs (b waits in registers for the barrier behind which nobody reads X any more) and gathers its own rows.  None of the
// tuned path's compressions (no duplicate-slot collapse, no super-nodes): every slot is processed, like the plain-fp32
// any-shape kernel (sgpr_generic.hip), which stays the path for everything outside these limits and for a graph whose
// values leave the f16 range (flagged here, embedded again there in the same call).  Widths are padded to multiples of
// 32 with zero weights at sgpr_create (WideModel): a padded channel is lrelu(0 + 0) = 0 and meets zero weights.
// Coordinate-layer keys restate the reference's fp32 arithmetic operation for operation (bit-identical neighbour sets
// there), as every other kernel of this library does.
#include <math.h>

#include <algorithm>

#include "sgpr_internal.hpp"

namespace sgpr {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// Two instances: CW = channels an X row holds (two f16 planes of CW each) = the widest padded layer, F3W = the widest
// padded filters_3.  <128, 64> serves filters up to 128 / 128 / 64 (155 KB of LDS at node_num 100: one workgroup per CU);
// <64, 32> the built widths under more labels / neurons (80 KB: two per CU).
template <int CW, int F3W>
struct WideCfg {
    static constexpr int XR = 4 * CW + 16;       // bytes per X row: planes + (x, y, z, |x|^2) of the coordinate layer; b (fp32) overlays the planes
    static constexpr int PA = CW + 4;            // floats per row of the gather target A (and of E)
    static constexpr int PP = F3W + 4;           // floats per row of the parked first-branch output
    static constexpr int KS = CW / 32;           // k-steps of the widest layer
};
#ifndef SGPR_WIDE_NARROW_OCC
#define SGPR_WIDE_NARROW_OCC 4                   // waves per SIMD the narrow instance is built for: 4 = two 7-wave workgroups per CU (A/B builds: 2)
#endif
constexpr int WK = 10;                           // K (the reference's): the selection's lists are cut to it at compile time
constexpr int WNP = SGPR_WIDE_MAX_NODES;         // rows (a multiple of 16)
constexpr int WTILES = WNP / 16;                 // 7 row tiles = 7 waves
constexpr float kWideF16Safe = 60000.f;

struct Frag {
    f16x8 h, l;
};

__device__ __forceinline__ f32x4 mfma(f16x8 a, f16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }

// k-step `st` of the row at `row` (byte pointer): channels 32 st + 8 lq .. + 7 of both planes
template <int CW>
__device__ __forceinline__ Frag xfrag(const unsigned char* row, int st, int lq) {
    Frag f;
    f.h = *reinterpret_cast<const f16x8*>(row + 64 * st + 16 * lq);
    f.l = *reinterpret_cast<const f16x8*>(row + 2 * CW + 64 * st + 16 * lq);
    return f;
}

// sum over ks k-steps of a . b with the correction products in a chain of their own (smallest terms never meet the large
// accumulator), like tile16 of sgpr_embed.hip; ks is wave-uniform.  (Each chain split over even / odd k-steps - half the
// dependent depth, three more vector adds - was measured: 476 -> 530 us per 1024 graphs of the 128-wide model; it spills.)
template <int KS>
__device__ __forceinline__ f32x4 dotk(const Frag (&a)[KS], const Frag (&b)[KS], const int ks) {
    f32x4 lo = {0.f, 0.f, 0.f, 0.f}, hi = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int st = 0; st < KS; ++st)
        if (st < ks) {
            lo = mfma(a[st].l, b[st].h, lo);
            lo = mfma(a[st].h, b[st].l, lo);
        }
#pragma unroll
    for (int st = 0; st < KS; ++st)
        if (st < ks) hi = mfma(a[st].h, b[st].h, hi);
    return hi + lo;
}

// four consecutive channels of one row -> the two f16 planes (hi = the value truncated to f16, lo = f16(v - hi): 22 bits);
// vmax tracks the largest magnitude stored (a graph that reaches the f16 range is embedded again in plain fp32)
template <int CW>
__device__ __forceinline__ void xstore(unsigned char* row, int ch, float4 v, float& vmax) {
    const unsigned h01 = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(v.x, v.y));
    const unsigned h23 = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(v.z, v.w));
    unsigned l01, l23;
    asm("v_fma_mixlo_f16 %0, %2, -1.0, %4 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixlo_f16 %1, %3, -1.0, %6 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %0, %2, -1.0, %5 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %1, %3, -1.0, %7 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
        : "=&v"(l01), "=&v"(l23)
        : "v"(h01), "v"(h23), "v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w));
    *reinterpret_cast<uint2*>(row + 2 * ch) = make_uint2(h01, h23);
    *reinterpret_cast<uint2*>(row + 2 * CW + 2 * ch) = make_uint2(l01, l23);
    vmax = fmaxf(vmax, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
}

__device__ __forceinline__ float kmin(float a, float b) {
    float d;
    asm("v_min_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ float kmax(float a, float b) {
    float d;
    asm("v_max_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ void cswap(float& a, float& b) {
    const float lo = kmin(a, b);
    b = kmax(a, b);
    a = lo;
}
// Batcher's odd-even merge sort of N values, ascending (entries >= NV are +inf and never move: their comparators are not generated)
template <int N, int NV>
__device__ __forceinline__ void batcher_sort(float (&v)[N]) {
#pragma unroll
    for (int p = 1; p < N; p *= 2)
#pragma unroll
        for (int k = p; k >= 1; k /= 2)
#pragma unroll
            for (int j = k % p; j <= N - 1 - k; j += 2 * k)
#pragma unroll
                for (int i = 0; i <= (k - 1 < N - j - k - 1 ? k - 1 : N - j - k - 1); ++i)
                    if ((i + j) / (2 * p) == (i + j + k) / (2 * p) && i + j + k < NV) cswap(v[i + j], v[i + j + k]);
}
template <int N>
__device__ __forceinline__ void bitonic_merge(float (&v)[N]) {
#pragma unroll
    for (int j = N / 2; j > 0; j >>= 1)
#pragma unroll
        for (int i = 0; i < N; ++i)
            if ((i & j) == 0) cswap(v[i], v[i | j]);
}

// candidate index of bit u of a lane's mask: tile u >> 2, lane group lq, element u & 3
__device__ __forceinline__ int cand_of(int u, int lq) { return 16 * (u >> 2) + 4 * lq + (u & 3); }
__device__ __forceinline__ void emit(unsigned take, int lq, unsigned char* __restrict__ out, int& pos) {
    while (take) {
        const int u = __ffs(take) - 1;
        take &= take - 1;
        out[pos++] = (unsigned char)cand_of(u, lq);      // (candidate indices < 128)
    }
}
__device__ __forceinline__ int prefix4(int v, int lq) {      // inclusive prefix over the four lanes of a row (16 lanes apart)
    const int t1 = __shfl_up(v, 16);
    v += lq >= 1 ? t1 : 0;
    const int t2 = __shfl_up(v, 32);
    v += lq >= 2 ? t2 : 0;
    return v;
}

// The K nearest of the wave's 16 rows (row tile `wave`) among the graph's n slots, as a SET: nbr[row][0..K) = candidate
// indices.  Lane (l15, lq) holds the keys of row 16 wave + l15 for candidates 16 tj + 4 lq + r (the accumulator layout of
// candidates-as-A, rows-as-B), <= 28 of them: sorted in registers, the K-th smallest key of the row by a butterfly over
// its four lanes, then one mask per lane; ties across the cut take the lowest candidate indices (the rule of every kernel
// of the library and of sgpr_knn).  coord: the coordinate layer - the reference's own fp32 operations on (x, y, z, |x|^2).
template <int CW>
__device__ __forceinline__ void select_wide(const unsigned char* __restrict__ X, const float* __restrict__ xx,
                                            unsigned char* __restrict__ nbr, const int n, const int nrt, const int ks,
                                            const int wave, const bool coord) {
    constexpr int K = WK, KP = 16, WXR = 4 * CW + 16, WKS = CW / 32;
    const int lane = threadIdx.x & 63, l15 = lane & 15, lq = lane >> 4;
    const int i = 16 * wave + l15;
    const bool active = i < n;
    float d[32];
    if (coord) {
        const float4 ci = *reinterpret_cast<const float4*>(X + i * WXR + (WXR - 16));
#pragma unroll
        for (int tj = 0; tj < 8; ++tj) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float key = INFINITY;
                if (tj < nrt) {
                    const float4 cj = *reinterpret_cast<const float4*>(X + (16 * tj + 4 * lq + r) * WXR + (WXR - 16));
                    const float dot = fmaf(ci.z, cj.z, fmaf(ci.y, cj.y, __fmul_rn(ci.x, cj.x)));
                    const float t = fmaf(2.f, dot, -cj.w);
                    key = __fsub_rn(ci.w, t);            // (+inf for a slot the graph does not have: its |x|^2 is)
                }
                d[4 * tj + r] = key;
            }
        }
    } else {
        Frag b[WKS];
#pragma unroll
        for (int st = 0; st < WKS; ++st)
            if (st < ks) b[st] = xfrag<CW>(X + i * WXR, st, lq);
#pragma unroll
        for (int tj = 0; tj < 8; ++tj) {
            float key[4] = {INFINITY, INFINITY, INFINITY, INFINITY};
            if (tj < nrt) {
                Frag a[WKS];
#pragma unroll
                for (int st = 0; st < WKS; ++st)
                    if (st < ks) a[st] = xfrag<CW>(X + (16 * tj + l15) * WXR, st, lq);
                const f32x4 g = dotk<WKS>(a, b, ks);
                __builtin_amdgcn_sched_barrier(0);       // (one candidate tile's operands in registers at a time)
                const float4 xj = *reinterpret_cast<const float4*>(xx + 16 * tj + 4 * lq);   // (+inf beyond the graph's slots)
                key[0] = fmaf(-2.f, g[0], xj.x);
                key[1] = fmaf(-2.f, g[1], xj.y);
                key[2] = fmaf(-2.f, g[2], xj.z);
                key[3] = fmaf(-2.f, g[3], xj.w);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) d[4 * tj + r] = key[r];
        }
    }
    // ---- this lane's K smallest, ascending (two sorted halves, the smaller halves of their bitonic merge)
    float L[KP];
    {
        float lo[16], hi[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            lo[u] = d[u];
            hi[u] = d[16 + u];
        }
        batcher_sort<16, 16>(lo);
        if (nrt > 4) {                                   // (wave-uniform)
            batcher_sort<16, 16>(hi);
#pragma unroll
            for (int u = 0; u < 16; ++u) L[u] = kmin(lo[u], hi[15 - u]);
            bitonic_merge<KP>(L);
        } else {
#pragma unroll
            for (int u = 0; u < 16; ++u) L[u] = lo[u];
        }
    }
    {   // butterfly over the row's four lanes: 16 lanes away the merged list, 32 away only the K-th key
        float o[KP];
#pragma unroll
        for (int s = 0; s < KP; ++s) o[s] = (KP - 1 - s < K) ? __shfl_xor(L[KP - 1 - s], 16) : INFINITY;
#pragma unroll
        for (int s = 0; s < KP; ++s) L[s] = s < K ? ((KP - 1 - s < K) ? kmin(L[s], o[s]) : L[s]) : o[s];
        bitonic_merge<KP>(L);
    }
    float tau = -INFINITY;                               // the K-th smallest key of the row
#pragma unroll
    for (int s = 0; s < K; ++s) tau = kmax(tau, kmin(L[s], __shfl_xor(L[K - 1 - s], 32)));
    unsigned char* out = nbr + i * 16;
    // ---- the common case: exactly K keys at or below tau -> one mask, a prefix over the row's lanes, emission
    unsigned gt = 0u;
#pragma unroll
    for (int u = 31; u >= 0; --u) gt = __builtin_amdgcn_alignbit(gt, __float_as_uint(tau - d[u]), 31);
    const unsigned valid = nrt >= 8 ? ~0u : ((1u << (4 * nrt)) - 1u);
    const unsigned le = ~gt & valid;
    const int n_le = __popc(le);
    const int incl = prefix4(n_le, lq);
    const int total_le = __shfl(incl, 48 + l15);
    if (__ballot(active && total_le != K) == 0ull) {
        if (active) {
            int pos = incl - n_le;
            emit(le, lq, out, pos);
        }
        return;
    }
    // ---- ties across the cut: everything below tau, then the first T candidates AT tau in candidate-index order
    //      = (tile, lane group, element) order
    unsigned lt = 0u;
#pragma unroll
    for (int u = 31; u >= 0; --u) lt = __builtin_amdgcn_alignbit(lt, __float_as_uint(d[u] - tau), 31);
    lt &= valid;
    const unsigned eq = ~(lt | gt) & valid;              // (inf - inf is a positive NaN: a missing slot at an infinite tau is "equal")
    const int n_less = __popc(lt);
    const int less_incl = prefix4(n_less, lq);
    const int total_less = __shfl(less_incl, 48 + l15);
    const int T = K - total_less;                        // ties to accept
    if (active) {
        int pos = less_incl - n_less;
        emit(lt, lq, out, pos);
    }
    int before = 0;                                      // ties in the tiles before the current one
#pragma unroll 1
    for (int tj = 0; tj < nrt; ++tj) {                   // (every lane runs the exchanges; only active rows emit)
        const unsigned bits0 = (eq >> (4 * tj)) & 0xfu;
        const int own = __popc(bits0);
        const int tin = prefix4(own, lq);
        const int all = __shfl(tin, 48 + l15);
        const int start = before + tin - own;
        const int take_n = max(0, min(own, T - start));
        unsigned bits = bits0, keep = 0u;
        for (int c = 0; c < take_n; ++c) {
            const unsigned low = bits & (0u - bits);
            keep |= low;
            bits ^= low;
        }
        if (active) {
            int tpos = total_less + min(start, T);
            emit(keep << (4 * tj), lq, out, tpos);
        }
        before += all;
    }
}

__device__ __forceinline__ float lrelu(float y) { return fmaxf(y, 0.2f * y); }

struct WideArgs {
    WideModel m;
    EmbedArgs a;
    int N, NP, pw;
};

// LDS: X [NP][XR] | A [NP][PA] f32 | park [NP][PP] f32 | xx [NP] f32 | nbr [NP][16] u8 | red [2 * 64 + 16] f32
template <int CW, int F3W>
__global__ __launch_bounds__(64 * WTILES, (CW <= 64 ? SGPR_WIDE_NARROW_OCC : 2)) void wide_embed_kernel(const WideArgs p) {
    constexpr int WXR = WideCfg<CW, F3W>::XR, WPA = WideCfg<CW, F3W>::PA, WPP = WideCfg<CW, F3W>::PP, WKS = WideCfg<CW, F3W>::KS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const WideModel& m = p.m;
    const EmbedArgs& a = p.a;
    const int N = p.N, NP = p.NP, nrt = NP >> 4;
    unsigned char* X = smem;
    float* A = reinterpret_cast<float*>(X + NP * WXR);
    float* park = A + NP * WPA;
    float* xx = park + NP * WPP;
    unsigned char* nbr = reinterpret_cast<unsigned char*>(xx + NP);
    float* red = reinterpret_cast<float*>(nbr + NP * 16);
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, lq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NT = blockDim.x;
    const int slot = blockIdx.x;
    const int g = a.ids ? a.ids[slot] : slot;
    float vmax = 0.f;
    // ---- one slot per thread: xyz and the semantic row (one-hot of the label, or the dense tensor's own values)
    long long rag0 = 0, ragc = 0;
    bool rag_bad = false;
    if (a.rag_off && !a.dense) {
        rag0 = a.rag_off[g];
        ragc = a.rag_off[g + 1] - rag0;
        rag_bad = ragc < 0 || ragc > N;
    }
    if (rag_bad) {                                       // a graph with more nodes than slots: loud, like every other path
        if (tid == 0) {
            atomicOr(a.status, 8);
            a.redo[slot] = 0;
        }
        for (int c = tid; c < p.pw; c += NT) a.pooled[(size_t)g * p.pw + c] = __int_as_float(0x7fc00000);
        return;
    }
    float fx = 0.f, fy = 0.f, fz = 0.f;
    if (tid < NP) {
        unsigned char* xr = X + tid * WXR;
        const bool live = tid < N;
        int lab = -1;                                    // packed input: this slot's label
        const float* dn = nullptr;                       // dense input: this slot's column of the tensor
        if (live) {
            if (a.dense) {
                const bool second = a.dense2 && g >= a.g_split;
                dn = (second ? a.dense2 : a.dense) + (size_t)(second ? g - a.g_split : g) * (3 + m.L) * N + tid;
                fx = dn[0];
                fy = dn[N];
                fz = dn[2 * N];
            } else {
                if (a.rag_off) {
                    if (tid < ragc) {
                        const float* c3 = a.centers + (size_t)(rag0 + tid) * 3;
                        fx = c3[0];
                        fy = c3[1];
                        fz = c3[2];
                        lab = a.rag_lab[rag0 + tid];
                    }
                } else {
                    const float* c3 = a.centers + ((size_t)g * N + tid) * 3;
                    fx = c3[0];
                    fy = c3[1];
                    fz = c3[2];
                    lab = a.labels[(size_t)g * N + tid];
                }
                if (lab < -1 || lab >= m.L) {
                    atomicOr(a.status, 1);               // KeyError in the reference (sg_net.py:277)
                    lab = -1;
                }
            }
        }
        // the semantic branch's input: <= 32 channels = one k-step (one-hot of the label, or the dense tensor's own values)
        float s = 0.f;
#pragma unroll 1
        for (int c = 0; c < SGPR_WIDE_MAX_LABELS; c += 4) {
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = dn ? (c + u < m.L ? dn[(size_t)(3 + c + u) * N] : 0.f) : (lab == c + u ? 1.f : 0.f);
            xstore<CW>(xr, c, make_float4(v[0], v[1], v[2], v[3]), vmax);
#pragma unroll
            for (int u = 0; u < 4; ++u) s = __fadd_rn(s, __fmul_rn(v[u], v[u]));
        }
        xx[tid] = live ? s : INFINITY;
    }
    __syncthreads();

    // ---- the semantic branch (GenericModel layers 3..5) first, parked; then the xyz branch (0..2)
    for (int pass = 0; pass < 6; ++pass) {
        const int L = pass < 3 ? 3 + pass : pass - 3;                 // GenericModel's index
        const bool coord = L == 0;
        const bool last = (L == 2 || L == 5);
        if (pass == 3) {
            // stage the xyz branch's input: (x, y, z) in channels 0..2 of one k-step, (x, y, z, |x|^2) in fp32 behind the planes
            if (tid < NP) {
                unsigned char* xr = X + tid * WXR;
                const bool live = tid < N;
                xstore<CW>(xr, 0, live ? make_float4(fx, fy, fz, 0.f) : make_float4(0.f, 0.f, 0.f, 0.f), vmax);
#pragma unroll
                for (int c = 4; c < 32; c += 4) xstore<CW>(xr, c, make_float4(0.f, 0.f, 0.f, 0.f), vmax);
                const float n2 = __fadd_rn(__fadd_rn(__fmul_rn(fx, fx), __fmul_rn(fy, fy)), __fmul_rn(fz, fz));   // as torch.sum(x ** 2)
                *reinterpret_cast<float4*>(xr + WXR - 16) = live ? make_float4(fx, fy, fz, n2) : make_float4(0.f, 0.f, 0.f, INFINITY);
            }
            __syncthreads();
        }
        const int ks = m.cinP[L] >> 5, coutP = m.coutP[L];
        const int nca = coutP >> 4;                                   // a-type column tiles (= b-type)
        // ---- selection of the wave's rows, then the a half of its own row tile of [a | b] = x . [W1' | (W2 - W1)'] -> A.
        //      Weight tiles stream from L2 / L1 (all waves read the same ones) one tile AHEAD of the matrix instructions
        //      that consume them (two register buffers; the tile count is even: widths are padded to 32)
        const unsigned short* wl = m.wh[L] + (size_t)lane * 8;
        auto load_w = [&](Frag (&w)[WKS], const int tile) {
#pragma unroll
            for (int st = 0; st < WKS; ++st)
                if (st < ks) {
                    const unsigned short* wp = wl + ((size_t)(tile * ks + st) * 2) * 512;
                    w[st].h = *reinterpret_cast<const f16x8*>(wp);
                    w[st].l = *reinterpret_cast<const f16x8*>(wp + 512);
                }
        };
        Frag w0[WKS], w1[WKS];
        select_wide<CW>(X, xx, nbr, N, nrt, ks, wave, coord);
        __builtin_amdgcn_sched_barrier(0);
        load_w(w0, 0);
        {
            Frag xf[WKS];
            const unsigned char* x0 = X + (wave * 16 + l15) * WXR;
#pragma unroll
            for (int st = 0; st < WKS; ++st)
                if (st < ks) xf[st] = xfrag<CW>(x0, st, lq);
            float* a0 = A + (wave * 16 + l15) * WPA + 4 * lq;
#pragma unroll 1
            for (int ct = 0; ct < nca; ct += 2) {
                load_w(w1, ct + 1);
                const f32x4 r = dotk<WKS>(w0, xf, ks);                     // r[c] = a[channel ct*16 + 4 lq + c][node 16 wave + l15]
                *reinterpret_cast<float4*>(a0 + ct * 16) = make_float4(r[0], r[1], r[2], r[3]);
                load_w(w0, ct + 2);                                   // (behind the last a tile: the first b tile, which follows them)
                const f32x4 r1 = dotk<WKS>(w1, xf, ks);
                *reinterpret_cast<float4*>(a0 + (ct + 1) * 16) = make_float4(r1[0], r1[1], r1[2], r1[3]);
            }
        }
        __syncthreads();                                              // A is complete; X has been read by everyone as candidates
        {
            // the b half: the wave's own rows are read as the operand (into registers) before its first tile replaces them -
            // b = x . (W2 - W1)' + t overlays the row's planes in place (fp32), no other wave reads these rows any more
            Frag xf[WKS];
            unsigned char* x0 = X + (wave * 16 + l15) * WXR;
#pragma unroll
            for (int st = 0; st < WKS; ++st)
                if (st < ks) xf[st] = xfrag<CW>(x0, st, lq);
            __builtin_amdgcn_sched_barrier(0);
            const float* tb = m.tbp[L] + 4 * lq;
#pragma unroll 1
            for (int cb = 0; cb < nca; cb += 2) {
                load_w(w1, nca + cb + 1);
                const float4 t4 = *reinterpret_cast<const float4*>(tb + cb * 16);
                const f32x4 r = dotk<WKS>(w0, xf, ks);
                *reinterpret_cast<float4*>(x0 + (cb * 16 + 4 * lq) * 4) = make_float4(r[0] + t4.x, r[1] + t4.y, r[2] + t4.z, r[3] + t4.w);
                if (cb + 2 < nca) load_w(w0, nca + cb + 2);
                const float4 t5 = *reinterpret_cast<const float4*>(tb + (cb + 1) * 16);
                const f32x4 r1 = dotk<WKS>(w1, xf, ks);
                *reinterpret_cast<float4*>(x0 + ((cb + 1) * 16 + 4 * lq) * 4) = make_float4(r1[0] + t5.x, r1[1] + t5.y, r1[2] + t5.z, r1[3] + t5.w);
            }
        }
        // ---- gather-max of the wave's own rows (its lists and its b rows are its own LDS traffic: program order): 16 lanes
        //      per row, 4 channels per lane and pass of 64 channels, four rows at a time
        {
            const int sub = lane >> 4, cl = (lane & 15) * 4;
            const int npass = (coutP + 63) >> 6;
#pragma unroll 1
            for (int r0 = 0; r0 < 16; r0 += 4) {
                const int ia = 16 * wave + r0 + sub;
                const unsigned char* nw = nbr + ia * 16;
                constexpr int NPASS = CW / 64;                        // passes of 64 channels
                float4 y[NPASS];
                float sq = 0.f;
#pragma unroll
                for (int cc = 0; cc < NPASS; ++cc) {
                    y[cc] = make_float4(0.f, 0.f, 0.f, 0.f);
                    const int c4 = cl + 64 * cc;
                    if (cc < npass && c4 < coutP && ia < N) {         // (a row beyond the graph's slots has no list: it stays zero)
                        float4 mx = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
#pragma unroll
                        for (int q = 0; q < WK; ++q) {
                            const float4 v = *reinterpret_cast<const float4*>(A + (int)nw[q] * WPA + c4);
                            mx.x = fmaxf(mx.x, v.x);
                            mx.y = fmaxf(mx.y, v.y);
                            mx.z = fmaxf(mx.z, v.z);
                            mx.w = fmaxf(mx.w, v.w);
                        }
                        const float4 bv = *reinterpret_cast<const float4*>(X + ia * WXR + 4 * c4);
                        y[cc] = make_float4(lrelu(mx.x + bv.x), lrelu(mx.y + bv.y), lrelu(mx.z + bv.z), lrelu(mx.w + bv.w));
                        sq += fmaf(y[cc].x, y[cc].x, fmaf(y[cc].y, y[cc].y, fmaf(y[cc].z, y[cc].z, y[cc].w * y[cc].w)));
                    }
                }
                // (every lane of the row has read its b above: the planes / the parked row may now replace it)
#pragma unroll
                for (int cc = 0; cc < NPASS; ++cc) {
                    const int c4 = cl + 64 * cc;
                    if (cc < npass && c4 < coutP) {
                        if (L == 5)
                            *reinterpret_cast<float4*>(park + ia * WPP + c4) = y[cc];     // sem3 waits for conv_end
                        else
                            xstore<CW>(X + ia * WXR, c4, y[cc], vmax);
                    }
                }
                if (!last) {                                          // squared norms of the next layer's input rows
                    sq += __shfl_xor(sq, 1);
                    sq += __shfl_xor(sq, 2);
                    sq += __shfl_xor(sq, 4);
                    sq += __shfl_xor(sq, 8);
                    if ((lane & 15) == 0) xx[ia] = ia < N ? sq : INFINITY;
                }
            }
        }
        __syncthreads();
    }
    // ---- X channels [0, F3P) hold xyz3; sem3 joins them at [F3P, 2 F3P): X = cat(xyz3, sem3) (sg_net.py:104)
    const int F3P = m.F3P;
    for (int e = tid; e < NP * (F3P >> 2); e += NT) {
        const int i = e / (F3P >> 2), c4 = (e - i * (F3P >> 2)) * 4;
        xstore<CW>(X + i * WXR, F3P + c4, *reinterpret_cast<const float4*>(park + i * WPP + c4), vmax);
    }
    __syncthreads();
    // ---- conv_end: 2 F3P -> F3P, folded BatchNorm, LeakyReLU -> E (in the A region); a wave per row tile
    float* E = A;
    {
        const int ks = (2 * F3P) >> 5, nct = F3P >> 4;
        Frag xf[WKS];
        const unsigned char* x0 = X + (wave * 16 + l15) * WXR;
#pragma unroll
        for (int st = 0; st < WKS; ++st)
            if (st < ks) xf[st] = xfrag<CW>(x0, st, lq);
        const unsigned short* wl = m.wh_end + (size_t)lane * 8;
#pragma unroll 1
        for (int ct = 0; ct < nct; ++ct) {
            Frag w[WKS];
#pragma unroll
            for (int st = 0; st < WKS; ++st)
                if (st < ks) {
                    const unsigned short* wp = wl + ((size_t)(ct * ks + st) * 2) * 512;
                    w[st].h = *reinterpret_cast<const f16x8*>(wp);
                    w[st].l = *reinterpret_cast<const f16x8*>(wp + 512);
                }
            const f32x4 r = dotk<WKS>(w, xf, ks);
            const float4 t4 = *reinterpret_cast<const float4*>(m.tbp_end + ct * 16 + 4 * lq);
            *reinterpret_cast<float4*>(E + (wave * 16 + l15) * WPA + ct * 16 + 4 * lq) =
                make_float4(lrelu(r[0] + t4.x), lrelu(r[1] + t4.y), lrelu(r[2] + t4.z), lrelu(r[3] + t4.w));
        }
    }
    __syncthreads();
    // ---- attention pooling over all N slots (layers_batch.py:28-39: no pad mask, divisor N)
    const int f3 = m.f3;
    float* mean = red;                                   // [64]
    float* ctx = red + 64;                               // [64]
    for (int c = wave; c < f3; c += NT >> 6) {
        float s = 0.f;
        for (int n = lane; n < N; n += 64) s += E[n * WPA + c];
#pragma unroll
        for (int sft = 32; sft > 0; sft >>= 1) s += __shfl_xor(s, sft);
        if (lane == 0) mean[c] = s / (float)N;
    }
    __syncthreads();
    for (int c = tid; c < f3; c += NT) {
        float gsum = 0.f;
        for (int r = 0; r < f3; ++r) gsum = fmaf(mean[r], m.att_w[(size_t)r * f3 + c], gsum);
        ctx[c] = tanhf(gsum);
    }
    __syncthreads();
    float* sig = xx;
    for (int n = tid; n < N; n += NT) {
        float dsum = 0.f;
        for (int c = 0; c < f3; ++c) dsum = fmaf(E[n * WPA + c], ctx[c], dsum);
        sig[n] = 1.f / (1.f + expf(-dsum));
    }
    __syncthreads();
    for (int c = wave; c < p.pw; c += NT >> 6) {
        float s = 0.f;
        if (c < f3)
            for (int n = lane; n < N; n += 64) s = fmaf(sig[n], E[n * WPA + c], s);
#pragma unroll
        for (int sft = 32; sft > 0; sft >>= 1) s += __shfl_xor(s, sft);
        if (lane == 0) a.pooled[(size_t)g * p.pw + c] = s;
    }
    if (a.att)
        for (int n = tid; n < N; n += NT) a.att[(size_t)g * N + n] = sig[n];
    if (a.emb)
        for (int e = tid; e < N * p.pw; e += NT) {
            const int n = e / p.pw, c = e - n * p.pw;
            a.emb[((size_t)g * N + n) * p.pw + c] = c < f3 ? E[n * WPA + c] : 0.f;
        }
    // a graph whose values reached the f16 range (or NaN): flagged for the plain-fp32 kernel
    const unsigned long long bad = __ballot(!(vmax < kWideF16Safe));
    int* flag = reinterpret_cast<int*>(red + 128);
    if (tid == 0) *flag = 0;
    __syncthreads();
    if (bad && lane == 0) atomicOr(flag, 1);
    __syncthreads();
    if (tid == 0) {
        a.redo[slot] = *flag ? 1 : 0;
        // ... and this launch's token in one word: the plain-fp32 pass behind this launch returns at once while nobody stored it
        if (*flag) __hip_atomic_store(a.redo_count, a.sem_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

template <int CW, int F3W>
size_t wide_lds(int NP) {
    typedef WideCfg<CW, F3W> C;
    return (size_t)NP * C::XR + (size_t)NP * C::PA * 4 + (size_t)NP * C::PP * 4 + (size_t)NP * 4 + (size_t)NP * 16 + (2 * 64 + 16) * 4;
}
// the narrow instance: every padded width of the model is at most 64 / 32
bool wide_narrow(const WideModel& m) {
    for (int l = 0; l < 6; ++l)
        if (m.cinP[l] > 64 || m.coutP[l] > 64) return false;
    return m.F3P <= 32;
}

template <int CW, int F3W>
int launch_wide_t(const WideArgs& p, hipStream_t stream) {
    static LdsLimitOnce once;
    if (int rc = raise_lds_limit(&once, reinterpret_cast<const void*>(&wide_embed_kernel<CW, F3W>), 160 * 1024, "wide_embed_kernel")) return rc;
    const size_t lds = wide_lds<CW, F3W>(p.NP);
    hipLaunchKernelGGL((wide_embed_kernel<CW, F3W>), dim3(p.a.G), dim3(64 * (p.NP >> 4)), lds, stream, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "wide_embed_kernel launch");
    return SGPR_OK;
}


// ------------------------------------------------------------------ dense all-pairs tail on the matrix cores
// TenorNetworkModule.forward + FC head (layers_batch.py:70-83, sg_net.py:131-136) for the same moderately larger models:
// pooled width <= 64, tensor neurons <= 32, bottleneck neurons <= 32 (every width padded with zeros: exact).  The tuned
// tail's formulation (sgpr_score.hip): the bilinear form and the column half of the block term hoisted per row graph -
// A'_r[t][j] = sum_i e1[i] W[i][j][t] + Wb[t][F + j], u_r[t] = Wb[t][:F] . e1 + bias[t] -, both dense layers on
// v_mfma_f32_16x16x32_f16 with two f16 planes per operand, layer 2 fed straight from layer 1's accumulator layout, the
// scoring weight folded into layer 2.  At 64 x 64 x 32 a block of 16 pairs costs 12 + 8 matrix instructions (the built
// shape: 3 + 2).  One wave per (row graph, 256 columns) item; inputs outside the f16 range leave the whole rectangle to
// the plain-fp32 kernel (a gate word).
constexpr int TFP = 64, TTP = 32, TBP = 32;
typedef short i16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split2(float a, unsigned short& h, unsigned short& l) {
    const _Float16 hh = (_Float16)a;
    const _Float16 ll = (_Float16)(a - (float)hh);
    h = __builtin_bit_cast(unsigned short, hh);
    l = __builtin_bit_cast(unsigned short, ll);
}
// max |v| over the wave -> one atomic (bit patterns of non-negative floats order like the floats; +inf stands for NaN)
__device__ __forceinline__ void atomic_max_abs(unsigned* dst, float v) {
    unsigned u = __float_as_uint(fabsf(v));
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) u = max(u, (unsigned)__shfl_xor((int)u, m));
    if ((threadIdx.x & 63) == 0 && u != 0u) atomicMax(dst, u);
}

// header words of the workspace: max |A'|, max |u|, max |e2| (bit patterns of non-negative floats), the gate
struct TailHdr {
    unsigned amax, umax, emax, gate;
};

// eight fp32 values -> their hi / lo f16 planes, eight halves (16 bytes) each
__device__ __forceinline__ void split2x8(const float (&v)[8], u32x4& hi, u32x4& lo) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        unsigned short h0, l0, h1, l1;
        split2(v[2 * q], h0, l0);
        split2(v[2 * q + 1], h1, l1);
        hi[q] = (unsigned)h0 | ((unsigned)h1 << 16);
        lo[q] = (unsigned)l0 | ((unsigned)l1 << 16);
    }
}

// blocks [0, 4 nrg): 16 row graphs per FOUR workgroups - A' as a small fp32-matrix-core GEMM ([16 graphs x 64] x [64 x 2048]):
// each of the 16 waves owns eight consecutive columns j of one half of the neurons (eight accumulators), so that a lane's
// eight values of a graph are the 16 contiguous bytes of the operand planes the tail reads; wave i of the 16 also computes
// u_r of graph i.  Blocks [4 nrg, ...): 64 column graphs each - their planes, 16 bytes per thread and plane.
__global__ __launch_bounds__(256) void wide_tail_prep_kernel(const GenericModel m, const float* __restrict__ rows, const int R,
                                                             const float* __restrict__ cols, const int M, const int pw, const int nrg,
                                                             unsigned short* __restrict__ Ab, float* __restrict__ ur,
                                                             unsigned short* __restrict__ Cb, TailHdr* __restrict__ hdr) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l15 = lane & 15, lq = lane >> 4;
    const int F = m.f3, T = m.T;
    if ((int)blockIdx.x >= 4 * nrg) {
        const int c0 = ((int)blockIdx.x - 4 * nrg) * 64;
        float emax = 0.f;
        for (int e = threadIdx.x; e < 64 * (TFP / 8); e += 256) {
            const int c = c0 + (e >> 3), jo = e & 7;
            if (c >= ((M + 15) & ~15)) continue;
            float v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float x = (c < M && 8 * jo + q < F) ? cols[(size_t)c * pw + 8 * jo + q] : 0.f;
                emax = fmaxf(emax, fabsf(x));
                if (x != x) emax = INFINITY;
                v[q] = x;
            }
            u32x4 hi, lo;
            split2x8(v, hi, lo);
            unsigned short* dst = Cb + ((size_t)(c >> 4) * 2 + (jo >> 2)) * 2 * 512 + (16 * (jo & 3) + (c & 15)) * 8;
            *reinterpret_cast<u32x4*>(dst) = hi;
            *reinterpret_cast<u32x4*>(dst + 512) = lo;
        }
        atomic_max_abs(&hdr->emax, emax);
        return;
    }
    const int g0 = ((int)blockIdx.x >> 2) * 16, gid = ((int)blockIdx.x & 3) * 4 + wave;
    const int jo = gid >> 1, th = gid & 1, t = 16 * th + l15;
    float amax = 0.f;
    f32x4 acc[8];
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) acc[jj] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (8 * jo < F && 16 * th < T) {                          // (wave-uniform; the planes of padding columns / neurons: zeros)
        const int g = min(g0 + l15, R - 1);
        const float* er = rows + (size_t)g * pw;
#pragma unroll 4
        for (int st = 0; st < 16; ++st) {
            // A operand of the fp32 16x16x4 matrix instruction: E[g0 + l15][4 st + lq]; B: W[i][j][t] of the eight columns
            const int i = 4 * st + lq;
            const float ea = i < F ? er[i] : 0.f;
            float b[8];
#pragma unroll
            for (int jj = 0; jj < 8; ++jj)
                b[jj] = (i < F && 8 * jo + jj < F && t < T) ? m.ntn_w[((size_t)i * F + 8 * jo + jj) * T + t] : 0.f;
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) acc[jj] = __builtin_amdgcn_mfma_f32_16x16x4f32(ea, b[jj], acc[jj], 0, 0, 0);
        }
    }
    float wbc[8];
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) wbc[jj] = (8 * jo + jj < F && t < T) ? m.ntn_wb[(size_t)t * 2 * F + F + 8 * jo + jj] : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {                            // acc[jj][r] = (e1^T W)_{g0 + 4 lq + r}[8 jo + jj][t]
        const int g = g0 + 4 * lq + r;
        if (g >= R) continue;
        float v[8];
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            v[jj] = acc[jj][r] + wbc[jj];
            amax = fmaxf(amax, fabsf(v[jj]));
            if (v[jj] != v[jj]) amax = INFINITY;
        }
        u32x4 hi, lo;
        split2x8(v, hi, lo);
        unsigned short* dst = Ab + ((((size_t)g * 2 + th) * 2 + (jo >> 2)) * 2) * 512 + (16 * (jo & 3) + l15) * 8;
        *reinterpret_cast<u32x4*>(dst) = hi;
        *reinterpret_cast<u32x4*>(dst + 512) = lo;
    }
    atomic_max_abs(&hdr->amax, amax);
    // u_r[t] = Wb[t][:F] . e1 + bias[t] of graph g0 + gid: lane = (half of the F terms, t)
    const int g = g0 + gid;
    if (g < R) {
        const int tt = lane & 31, hf = lane >> 5;
        float s = 0.f;
        if (tt < T)
            for (int q = 32 * hf; q < 32 * hf + 32; ++q)
                if (q < F) s = fmaf(m.ntn_wb[(size_t)tt * 2 * F + q], rows[(size_t)g * pw + q], s);
        s += __shfl_xor(s, 32);
        if (tt < T) s += m.ntn_bias[tt];
        if (hf == 0) ur[(size_t)g * TTP + tt] = tt < T ? s : 0.f;
        atomic_max_abs(&hdr->umax, (s != s) ? INFINITY : s);
    }
}

// relu(h[0..3]) -> the layer-2 B operand {hi01, hi23, lo01, lo23} (sgpr_score.hip, split_relu4: truncated hi plane, the
// low plane from one mixed-precision FMA per value, the ReLU as a packed signed-integer maximum with 0 per plane)
__device__ __forceinline__ f16x8 split_relu4(f32x4 h) {
    const unsigned h01 = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(h[0], h[1]));
    const unsigned h23 = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(h[2], h[3]));
    unsigned l01, l23;
    const i16x2 z = {0, 0};
    const unsigned a = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(i16x2, h01), z));
    const unsigned b = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(i16x2, h23), z));
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

// can every f16 this rectangle forms be represented?  |A'|, |e2| and |H| <= |u| + 64 max|A'| max|e2|
__device__ __forceinline__ bool tail_in_range(const TailHdr* hdr) {
    const float am = __uint_as_float(hdr->amax), um = __uint_as_float(hdr->umax), em = __uint_as_float(hdr->emax);
    return (am < kWideF16Safe) && (em < kWideF16Safe) && (um + (float)TFP * am * em < kWideF16Safe);
}

__global__ __launch_bounds__(256) void wide_tail_kernel(const GenericModel m, const int R, const int M,
                                                        const unsigned short* __restrict__ Ab, const float* __restrict__ ur,
                                                        const unsigned short* __restrict__ Cb, TailHdr* __restrict__ hdr,
                                                        float* __restrict__ score, const int64_t ld) {
    if (!tail_in_range(hdr)) {                               // the plain-fp32 kernel behind this launch takes the rectangle
        if (threadIdx.x == 0) hdr->gate = 1u;
        return;
    }
    const int lane = threadIdx.x & 63, l15 = lane & 15, g = lane >> 4;
    const int T = m.T, B = m.B;
    // the head folded into layer 2 (sgpr_score.hip, ap_consts), per (o tile, t tile)
    f16x8 w1hi[2][2], w1lo[2][2];
    f32x4 b1v[2], side[2];
#pragma unroll
    for (int ot = 0; ot < 2; ++ot) {
        const int o = 16 * ot + l15;                         // the A operand's row
        const float s = o < B ? m.fc2_w[o] : 0.f;
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            float w[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int t = 16 * tt + 4 * g + i;
                w[i] = (o < B && t < T) ? s * m.fc1_w[(size_t)o * T + t] : 0.f;
            }
            const _Float16 h0 = (_Float16)w[0], h1 = (_Float16)w[1], h2 = (_Float16)w[2], h3 = (_Float16)w[3], z16 = (_Float16)0.f;
            w1hi[ot][tt] = f16x8{h0, h1, h2, h3, h0, h1, h2, h3};        // meets H's hi and lo planes
            w1lo[ot][tt] = f16x8{(_Float16)(w[0] - (float)h0), (_Float16)(w[1] - (float)h1), (_Float16)(w[2] - (float)h2),
                                 (_Float16)(w[3] - (float)h3), z16, z16, z16, z16};   // meets the hi plane only
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {                        // the accumulator's rows are o = 16 ot + 4 g + r
            const int oo = 16 * ot + 4 * g + r;
            const float w2 = oo < B ? m.fc2_w[oo] : 0.f;
            b1v[ot][r] = oo < B ? w2 * m.fc1_b[oo] : 0.f;
            side[ot][r] = w2 < 0.f ? -INFINITY : INFINITY;
        }
    }
    const float kL2E = 1.4426950408889634f;
    const float nb2 = -m.fc2_b[0] * kL2E;
    const int nblk = (M + 15) >> 4, nch = (nblk + 15) >> 4;  // 16-column blocks; chunks of 16 blocks
    const long long items = (long long)R * nch;
    const long long wid = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (long long)gridDim.x * 4;
    for (long long it = wid; it < items; it += nw) {
        const int r = (int)(it / nch), ch = (int)(it - (long long)r * nch);
        f16x8 ah[2][2], al[2][2];
        f32x4 u4[2];
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const unsigned short* ap = Ab + ((((size_t)r * 2 + tt) * 2 + ks) * 2) * 512 + (size_t)lane * 8;
                ah[tt][ks] = *reinterpret_cast<const f16x8*>(ap);
                al[tt][ks] = *reinterpret_cast<const f16x8*>(ap + 512);
            }
            const float4 u = *reinterpret_cast<const float4*>(ur + (size_t)r * TTP + 16 * tt + 4 * g);
            u4[tt] = f32x4{u.x, u.y, u.z, u.w};
        }
        const int b0 = ch * 16, b1 = min(nblk, b0 + 16);
#pragma unroll 1
        for (int blk = b0; blk < b1; ++blk) {
            f16x8 bh[2], bl[2];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const unsigned short* cp = Cb + (((size_t)blk * 2 + ks) * 2) * 512 + (size_t)lane * 8;
                bh[ks] = *reinterpret_cast<const f16x8*>(cp);
                bl[ks] = *reinterpret_cast<const f16x8*>(cp + 512);
            }
            f16x8 hb[2];
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                f32x4 h = u4[tt];
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    h = mfma(al[tt][ks], bh[ks], h);
                    h = mfma(ah[tt][ks], bl[ks], h);
                }
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) h = mfma(ah[tt][ks], bh[ks], h);
                hb[tt] = split_relu4(h);                     // H[t = 16 tt + 4 g + i][column l15]
            }
            float z = 0.f;
#pragma unroll
            for (int ot = 0; ot < 2; ++ot) {
                f32x4 q = b1v[ot];
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
                    q = mfma(w1lo[ot][tt], hb[tt], q);
                    q = mfma(w1hi[ot][tt], hb[tt], q);
                }
                const float t0 = __builtin_amdgcn_fmed3f(q[0], 0.f, side[ot][0]), t1 = __builtin_amdgcn_fmed3f(q[1], 0.f, side[ot][1]);
                const float t2 = __builtin_amdgcn_fmed3f(q[2], 0.f, side[ot][2]), t3 = __builtin_amdgcn_fmed3f(q[3], 0.f, side[ot][3]);
                z += (t0 + t1) + (t2 + t3);                  // partial over o = 16 ot + 4 g .. + 3, column l15
            }
            z += __shfl_xor(z, 16);
            z += __shfl_xor(z, 32);
            const float sc = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(fmaf(z, -kL2E, nb2)));
            const int c = 16 * blk + l15;
            if (g == 0 && c < M) score[(size_t)r * ld + c] = sc;
        }
    }
}

size_t wide_tail_ws(int R, int M) {
    const size_t nblk = ((size_t)M + 15) >> 4;
    return 256 + (size_t)R * 8 * 1024 + (size_t)R * TTP * sizeof(float) + nblk * 4 * 1024;
}

}  // namespace

bool wide_tail_serves(const sgpr_handle* h) {
    return h->generic_only && h->wm.ok && h->gm.f3 <= TFP && h->gm.T <= TTP && h->gm.B <= TBP && !(h->dbg_skip & (1 << 23));
}
size_t wide_tail_ws_bytes(int R, int M) { return wide_tail_ws(R, M); }

// returns through *d_gate the device word the plain-fp32 kernel behind this call must test (non-zero: the rectangle is its)
int launch_score_all_pairs_wide_any(const sgpr_handle* h, const float* rows, int R, const float* cols, int M, float* score,
                                    int64_t ld, void* ws, const unsigned** d_gate, hipStream_t stream) {
    TailHdr* hdr = static_cast<TailHdr*>(ws);
    unsigned short* Ab = reinterpret_cast<unsigned short*>(static_cast<unsigned char*>(ws) + 256);
    float* ur = reinterpret_cast<float*>(Ab + (size_t)R * 8 * 512);
    unsigned short* Cb = reinterpret_cast<unsigned short*>(ur + (size_t)R * TTP);
    hipError_t e = hipMemsetAsync(hdr, 0, sizeof(TailHdr), stream);
    if (e != hipSuccess) return hip_fail(e, "wide tail: header");
    const int nrg = (R + 15) / 16, ncb = (((M + 15) & ~15) + 63) / 64, pw = h->gm.f3;
    hipLaunchKernelGGL(wide_tail_prep_kernel, dim3(4 * nrg + ncb), dim3(256), 0, stream, h->gm, rows, R, cols, M, pw, nrg, Ab, ur, Cb, hdr);
    e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "wide_tail_prep_kernel launch");
    const long long items = (long long)R * ((((M + 15) >> 4) + 15) >> 4);
    const long long slots = (long long)h->num_cus * 8;
    const unsigned grid = (unsigned)std::max<long long>(1, std::min<long long>((items + 3) / 4, slots));
    hipLaunchKernelGGL(wide_tail_kernel, dim3(grid), dim3(256), 0, stream, h->gm, R, M, Ab, ur, Cb, hdr, score, ld);
    e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "wide_tail_kernel launch");
    *d_gate = &hdr->gate;
    return SGPR_OK;
}

size_t wide_embed_lds_bytes(int N) { return wide_lds<SGPR_WIDE_MAX_FILTERS, SGPR_WIDE_MAX_F3>((N + 15) & ~15); }

bool wide_embed_serves(const sgpr_handle* h, const EmbedArgs& a, int N, int k) {
    return h->generic_only && h->wm.ok && k == WK && N >= k && N <= SGPR_WIDE_MAX_NODES && !a.dbg_layers && !a.dbg_knn &&
           !(h->dbg_skip & (1 << 23));                      // (debug bit 23: plain fp32 only - tests compare the two)
}

int launch_embed_wide(const sgpr_handle* h, const EmbedArgs& a, int N, int k, hipStream_t stream) {
    (void)k;
    if (a.G == 0) return SGPR_OK;
    WideArgs p;
    p.m = h->wm;
    p.a = a;
    p.N = N;
    p.NP = (N + 15) & ~15;
    p.pw = h->gm.f3;
    return wide_narrow(h->wm) ? launch_wide_t<64, 32>(p, stream) : launch_wide_t<SGPR_WIDE_MAX_FILTERS, SGPR_WIDE_MAX_F3>(p, stream);
}

}  // namespace sgpr
