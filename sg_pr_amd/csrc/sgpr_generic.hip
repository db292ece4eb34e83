// Any-shape form of the path: SG.dgcnn_conv_pass + AttentionModule (sg_net.py:79-110, dgcnn.py:14-49, layers_batch.py:28-39)
// and the pair-coupled tail (layers_batch.py:70-83, sg_net.py:131-136) for everything the tuned kernels are not built for -
// architectures LARGER than {12 labels, 64, 64, 32, 16, 16} in any hyper-parameter (the reference builds any:
// sg_net.py:40-76, parser_sg.py:12-18), node_num beyond 256, K beyond 32.  Correctness first, plain fp32: the reference's
// formulation with the two algebraic steps that do not change a value's meaning (eval BatchNorm folded into the 1x1
// convolutions; W.[x_j - x_i ; x_i] = W1.x_j + (W2 - W1).x_i with the max over neighbours taken on the first term) -
// no duplicate-slot compression, no super-nodes, no matrix cores.  One 1024-thread workgroup per CU and graph, activations
// channel-major in LDS when the working set fits (global scratch otherwise); the tail hoists the bilinear form per row
// graph.  At the built shape 39 x (embed) and 11 x (all-pairs tail) the tuned kernels' time (profiles/rNN_any_shape.txt);
// the tuned kernels keep every shape they serve (every shipped checkpoint).
#include <math.h>

#include "sgpr_internal.hpp"

namespace sgpr {

constexpr int GEN_THREADS = 256;                                  // the stand-alone module kernels
constexpr int EMB_THREADS = 1024;                                 // the embed kernel: one workgroup per CU (see generic_embed_slots)
constexpr int GEN_MAX_PER_LANE = SGPR_GENERIC_MAX_NODES / 64;     // candidates of a row one lane holds in the selection
constexpr float kSlope = 0.2f;                                    // LeakyReLU(0.2), sg_net.py:53

// floats of working memory one resident workgroup needs for graphs of N slots: two activation buffers, the a term, the
// first branch's output, squared norms, the neighbour lists
static size_t generic_scratch_floats(const GenericModel& m, int N, int k) {
    return (size_t)N * ((size_t)3 * m.cmax + m.f3 + 1 + k) + 64;
}
// ... which live in LDS when they fit (the built shape at node_num 100: 94 KB) and otherwise in a global scratch area
// per resident workgroup.  One 1024-thread workgroup per CU either way: with 256 CUs the scratch areas of a launch stay
// inside the L2s (three 256-thread workgroups per CU of the first version put 92 MB in flight and ran from the
// Infinity Cache: 5 ms per graph).
constexpr size_t kGenericLdsBytes = 156 * 1024;
static bool generic_in_lds(const GenericModel& m, int N, int k) {
    return generic_scratch_floats(m, N, k) * sizeof(float) <= kGenericLdsBytes;
}

size_t generic_embed_lds_bytes(const sgpr_handle* h, int N, int k) {
    return 2 * SGPR_GENERIC_MAX_F3 * sizeof(float) +
           (generic_in_lds(h->gm, N, k) ? generic_scratch_floats(h->gm, N, k) * sizeof(float) : 0);
}

int generic_embed_slots(const sgpr_handle* h, int G) {
    return G < h->num_cus ? G : h->num_cus;
}

size_t generic_embed_ws_bytes(const sgpr_handle* h, int G, int N, int k) {
    if (generic_in_lds(h->gm, N, k)) return 0;
    return (size_t)generic_embed_slots(h, G) * generic_scratch_floats(h->gm, N, k) * sizeof(float);
}

__device__ __forceinline__ float lrelu(float v) { return v > 0.f ? v : kSlope * v; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) v += __shfl_xor(v, s);
    return v;
}

// Neighbour set of row i: keys of candidates lane, lane + 64, ... in registers (NQ blocks: no guards inside the channel
// loop - a candidate beyond N reads slot N - 1 and gets the key of a slot that does not exist afterwards).
// key = -pd,  pd = (-|x_j|^2 + 2 x_i.x_j) - |x_i|^2  (dgcnn.py:15-17)
template <int NQ>
__device__ __forceinline__ void knn_select_row(const float* __restrict__ X, const float* __restrict__ xx, int* __restrict__ idx,
                                               int32_t* __restrict__ dbg_row, const int i, const int N, const int k,
                                               const int cin, const int lane, const int skip) {
    static_assert(NQ <= GEN_MAX_PER_LANE, "candidates per lane");
    int jj[NQ];
    float key[NQ];
    {
        const float xi = X[i];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int j = lane + 64 * q;
            jj[q] = j < N ? j : N - 1;
            key[q] = __fmul_rn(xi, X[jj[q]]);
        }
    }
    constexpr int kChannelsInFlight = NQ <= 2 ? 4 : (NQ <= 4 ? 2 : 1);   // independent loads in flight: 8 .. 16 per lane
#pragma unroll kChannelsInFlight
    for (int c = 1; c < ((skip & 1) ? 1 : cin); ++c) {
        const float* xc = X + (size_t)c * N;
        const float xi = xc[i];
#pragma unroll
        for (int q = 0; q < NQ; ++q) key[q] = fmaf(xi, xc[jj[q]], key[q]);
    }
    // the k nearest as a SET (the maximum over neighbours does not ask for their order): the k-th smallest key by
    // bisection on the keys' bit patterns - 32 steps of compare + ballot + count, no cross-lane traffic - then every
    // candidate below it and, among the candidates equal to it, the lowest indices (the tie rule of the tuned
    // kernels and of sgpr_knn).  The list is written in candidate order.
    unsigned uk[NQ];
    {
        const float xi2 = xx[i];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            // -pd = |x_i|^2 - (-|x_j|^2 - inner), inner = -2 dot
            const float kf = __fsub_rn(xi2, fmaf(2.f, key[q], -xx[jj[q]]));
            const unsigned bits = __float_as_uint(kf);
            unsigned u = (bits & 0x80000000u) ? ~bits : (bits | 0x80000000u);      // monotone in the float order
            if (kf != kf) u = 0xfffffffeu;                                          // NaN: after every number ...
            uk[q] = lane + 64 * q < N ? u : 0xffffffffu;                            // ... and before the slots that do not exist
        }
    }
    unsigned kth = 0u;                                            // largest v with #{u < v} < k  =  the k-th smallest
    for (int bit = 31; bit >= ((skip & 2) ? 31 : 0); --bit) {
        const unsigned trial = kth | (1u << bit);
        int cnt = 0;
#pragma unroll
        for (int q = 0; q < NQ; ++q) cnt += __popcll(__ballot(uk[q] < trial));
        if (cnt < k) kth = trial;
    }
    int below = 0;
#pragma unroll
    for (int q = 0; q < NQ; ++q) below += __popcll(__ballot(uk[q] < kth));
    int need_eq = k - below, pos = 0;                             // candidates equal to the k-th key still to take
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const unsigned long long m_eq = __ballot(uk[q] == kth);
        const bool take = uk[q] < kth || (uk[q] == kth && __popcll(m_eq & lt_mask) < need_eq);
        const unsigned long long m_take = __ballot(take);
        if (take) {
            const int at = pos + __popcll(m_take & lt_mask);
            idx[(size_t)at * N + i] = jj[q];
            if (dbg_row) dbg_row[at] = jj[q];
        }
        pos += __popcll(m_take);
        const int eq_here = __popcll(m_eq);
        need_eq -= eq_here < need_eq ? eq_here : need_eq;
    }
}

// Activations live CHANNEL-MAJOR in the scratch area (X[c * N + n], the reference's own [C, N] layout): the lanes of a
// wave walk neighbouring nodes, so every load of a layer is one or two contiguous lines (node-major rows put each lane
// on a line of its own: measured 6 ms per graph of 100 nodes, 20 x what the arithmetic takes).
//
// One EdgeConv block on X [cin][N] -> Y [cout][N]:  kNN in X's own space (dgcnn.knn), then
//   Y[co][i] = lrelu(max_{j in knn(i)} a[co][j] + b[co][i]),  a = Wa X,  b = Wb X + t   (BatchNorm folded: sgpr_create)
// (b waits in Y until the gather replaces it)
__device__ __forceinline__ void generic_edgeconv(const float* __restrict__ X, float* __restrict__ Y, float* __restrict__ A,
                                 float* __restrict__ xx, int* __restrict__ idx, const int N, const int k, const int cin,
                                 const int cout, const float* __restrict__ wa, const float* __restrict__ wb,
                                 const float* __restrict__ tb, float* __restrict__ dbg_y, int32_t* __restrict__ dbg_idx,
                                 const int skip) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // (a scalar: what depends on it alone loads through the scalar cache)
    // |x|^2 (dgcnn.py:16: torch.sum(x ** 2, dim=1)) as torch evaluates it: squares rounded, summed in channel order with
    // every step rounded; the dot product below is the FMA chain in channel order.  For the coordinate layer (3 channels)
    // these are the reference's keys bit for bit (the operation order the tuned kernel restates too: gram_xyz_direct in
    // sgpr_embed.hip, pinned against torch in tests/test_oracle_golden.py); wider layers agree to fp32 rounding.
    for (int n = tid; n < N; n += EMB_THREADS) {
        float s = __fmul_rn(X[n], X[n]);
#pragma unroll 8
        for (int c = 1; c < cin; ++c) s = __fadd_rn(s, __fmul_rn(X[(size_t)c * N + n], X[(size_t)c * N + n]));
        xx[n] = s;
    }
    __syncthreads();
    // kNN: a wave per row (knn_select_row), instantiated for the number of 64-candidate blocks a lane holds
    const int nq = (N + 63) >> 6;
    for (int i = wave; i < N; i += EMB_THREADS / 64) {
        int32_t* drow = dbg_idx ? dbg_idx + (size_t)i * k : nullptr;
        if (nq <= 1) knn_select_row<1>(X, xx, idx, drow, i, N, k, cin, lane, skip);
        else if (nq <= 2) knn_select_row<2>(X, xx, idx, drow, i, N, k, cin, lane, skip);
        else if (nq <= 4) knn_select_row<4>(X, xx, idx, drow, i, N, k, cin, lane, skip);
        else if (nq <= 8) knn_select_row<8>(X, xx, idx, drow, i, N, k, cin, lane, skip);
        else knn_select_row<16>(X, xx, idx, drow, i, N, k, cin, lane, skip);
    }
    // a = Wa x, b = Wb x + t: a wave takes (block of 8 output channels, 64 nodes) tasks - one load of x feeds 16
    // multiply-adds, the 8 + 8 weights of a channel are wave-uniform and contiguous (wa / wb are [cin][cout8])
    const int cout8 = (cout + 7) & ~7, nchunk = (N + 63) >> 6;
    for (int task = wave; task < (cout8 >> 3) * nchunk; task += EMB_THREADS / 64) {
        const int cb = task / nchunk, n = (task - cb * nchunk) * 64 + lane;
        const bool ok = n < N;
        const float* wra = wa + cb * 8;
        const float* wrb = wb + cb * 8;
        float aa[8], bb[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) aa[u] = bb[u] = 0.f;
#pragma unroll 4
        for (int c = 0; c < ((skip & 4) ? 1 : cin); ++c) {
            const float x = ok ? X[(size_t)c * N + n] : 0.f;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                aa[u] = fmaf(wra[(size_t)c * cout8 + u], x, aa[u]);
                bb[u] = fmaf(wrb[(size_t)c * cout8 + u], x, bb[u]);
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int co = cb * 8 + u;
            if (ok && co < cout) {
                A[(size_t)co * N + n] = aa[u];
                Y[(size_t)co * N + n] = bb[u] + tb[co];
            }
        }
    }
    __syncthreads();
    for (int e = tid; e < N * cout; e += EMB_THREADS) {
        const int co = e / N, n = e - co * N;
        const float* ar = A + (size_t)co * N;
        float mx = -INFINITY;
#pragma unroll 4
        for (int m = 0; m < ((skip & 8) ? 1 : k); ++m) mx = fmaxf(mx, ar[idx[(size_t)m * N + n]]);
        const float y = lrelu(mx + Y[e]);
        Y[e] = y;
        if (dbg_y) dbg_y[(size_t)n * 64 + co] = y;                  // (sgpr_embed_debug's dump rows are 64 floats)
    }
    __syncthreads();
}

// IN_LDS: the working memory is the dynamic LDS block (every pointer into it derives from the LDS symbol, so the compiler
// addresses it with ds_ instructions; one instance that picks LDS or global at run time addresses both through flat_
// instructions and ran 2 x slower)
template <bool IN_LDS>
__global__ __launch_bounds__(EMB_THREADS) void generic_embed_kernel(const GenericModel m, const EmbedArgs a, const int N,
                                                                    const int k, float* __restrict__ scratch,
                                                                    const size_t per_wg, const int pw) {
    extern __shared__ __attribute__((aligned(16))) float emb_smem[];   // the working memory when it fits (scratch == NULL)
    __shared__ float red[SGPR_GENERIC_MAX_F3];
    __shared__ float ctx[SGPR_GENERIC_MAX_F3];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* base = IN_LDS ? emb_smem : scratch + (size_t)blockIdx.x * per_wg;
    float* X0 = base;                                             // every activation block is [channels][N]
    float* X1 = X0 + (size_t)N * m.cmax;
    float* A = X1 + (size_t)N * m.cmax;
    float* Y3 = A + (size_t)N * m.cmax;                           // [f3][N] the xyz branch's output
    float* xx = Y3 + (size_t)N * m.f3;
    int* idx = reinterpret_cast<int*>(xx + N);                    // [k][N]
    // behind the matrix-core embed of sgpr_wide.hip: only the graphs it flagged (values outside the f16 range) - none, as a
    // rule: one load of the word that launch stores its token in, and out
    if (a.auto_over == 7 && *a.redo_count != a.sem_epoch) return;
    for (int slot = blockIdx.x; slot < a.G; slot += gridDim.x) {
        if (a.auto_over == 7 && a.redo[slot] != 1) continue;
        const int g = a.ids ? a.ids[slot] : slot;
        // ---- input (transfer_to_torch's tensor, sg_net.py:250-299): xyz and the semantic rows of every slot
        long long rag0 = 0, ragc = 0;
        bool rag_bad = false;
        if (a.rag_off && !a.dense) {
            rag0 = a.rag_off[g];
            ragc = a.rag_off[g + 1] - rag0;
            rag_bad = ragc < 0 || ragc > N;
        }
        if (rag_bad) {                                           // (a graph with more nodes than slots: loud, like the tuned path)
            if (tid == 0) atomicOr(a.status, 8);
            for (int c = tid; c < pw; c += EMB_THREADS) a.pooled[(size_t)g * pw + c] = __int_as_float(0x7fc00000);
            continue;
        }
        for (int branch = 0; branch < 2; ++branch) {
            const int C0 = branch == 0 ? 3 : m.L;
            for (int e = tid; e < N * C0; e += EMB_THREADS) {
                const int c = e / N, n = e - c * N;
                float v = 0.f;
                if (a.dense) {
                    const bool second = a.dense2 && g >= a.g_split;               // (sgpr_forward_dense: the two sides of a batch)
                    const float* dn = second ? a.dense2 : a.dense;
                    v = dn[((size_t)(second ? g - a.g_split : g) * (3 + m.L) + (branch == 0 ? c : 3 + c)) * N + n];
                } else if (a.rag_off) {
                    if (n < ragc) v = branch == 0 ? a.centers[(size_t)(rag0 + n) * 3 + c] : (a.rag_lab[rag0 + n] == c ? 1.f : 0.f);
                } else if (branch == 0) {
                    v = a.centers[((size_t)g * N + n) * 3 + c];
                } else {
                    const int lab = a.labels[(size_t)g * N + n];
                    if (c == 0 && (lab < -1 || lab >= m.L)) atomicOr(a.status, 1);   // KeyError in the reference (sg_net.py:277)
                    v = lab == c ? 1.f : 0.f;
                }
                X0[e] = v;
            }
            if (branch == 1 && a.rag_off && !a.dense)
                for (int n = tid; n < ragc; n += EMB_THREADS) {
                    const int lab = a.rag_lab[rag0 + n];
                    if (lab < -1 || lab >= m.L) atomicOr(a.status, 1);   // (-1 is padding, as on the padded arrays and in the tuned kernel)
                }
            __syncthreads();
            float* cur = X0;
            float* nxt = X1;
            for (int l = 0; l < 3; ++l) {
                const int L6 = branch * 3 + l;
                float* dst = (branch == 0 && l == 2) ? Y3 : nxt;  // the xyz branch's output waits in a buffer of its own
                generic_edgeconv(cur, dst, A, xx, idx, N, k, m.cin[L6], m.cout[L6], m.wa[L6], m.wb[L6], m.tb[L6],
                                 a.dbg_layers ? a.dbg_layers + ((size_t)g * 6 + L6) * N * 64 : nullptr,
                                 a.dbg_knn ? a.dbg_knn + ((size_t)g * 6 + L6) * N * k : nullptr, a.skip >> 24);
                if (dst == nxt) {
                    nxt = cur;
                    cur = dst;
                }
            }
            if (branch == 1) {
                // `cur` holds sem3 [f3][N]; conv_end on cat(xyz3, sem3) (sg_net.py:104-109) -> E (into A)
                const float* S3 = cur;
                const int f8 = (m.f3 + 7) & ~7, nchunk = (N + 63) >> 6;      // (w_end is [2 f3][f8]: see generic_edgeconv)
                for (int task = wave; task < (f8 >> 3) * nchunk; task += EMB_THREADS / 64) {
                    const int cb = task / nchunk, n = (task - cb * nchunk) * 64 + lane;
                    const bool ok = n < N;
                    const float* wr = m.w_end + cb * 8;
                    float v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) v[u] = 0.f;
                    for (int half = 0; half < 2; ++half) {
                        const float* src = half == 0 ? Y3 : S3;
#pragma unroll 4
                        for (int c = 0; c < m.f3; ++c) {
                            const float x = ok ? src[(size_t)c * N + n] : 0.f;
#pragma unroll
                            for (int u = 0; u < 8; ++u) v[u] = fmaf(wr[(size_t)(half * m.f3 + c) * f8 + u], x, v[u]);
                        }
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int co = cb * 8 + u;
                        if (ok && co < m.f3) A[(size_t)co * N + n] = lrelu(v[u] + m.t_end[co]);
                    }
                }
                __syncthreads();
            }
        }
        const float* E = A;                                       // [f3][N]
        // ---- attention pooling (layers_batch.py:28-39: no pad mask, divisor N); a wave per channel for the sums over nodes
        for (int c = wave; c < m.f3; c += EMB_THREADS / 64) {
            float s = 0.f;
            for (int n = lane; n < N; n += 64) s += E[(size_t)c * N + n];
            s = wave_sum(s);
            if (lane == 0) red[c] = s / (float)N;
        }
        __syncthreads();
        for (int c = tid; c < m.f3; c += EMB_THREADS) {
            float gsum = 0.f;
            for (int r = 0; r < m.f3; ++r) gsum = fmaf(red[r], m.att_w[(size_t)r * m.f3 + c], gsum);
            ctx[c] = tanhf(gsum);
        }
        __syncthreads();
        for (int n = tid; n < N; n += EMB_THREADS) {
            float d = 0.f;
#pragma unroll 8
            for (int c = 0; c < m.f3; ++c) d = fmaf(E[(size_t)c * N + n], ctx[c], d);
            xx[n] = 1.f / (1.f + expf(-d));
        }
        __syncthreads();
        for (int c = wave; c < pw; c += EMB_THREADS / 64) {
            float s = 0.f;
            if (c < m.f3)
                for (int n = lane; n < N; n += 64) s = fmaf(xx[n], E[(size_t)c * N + n], s);
            s = wave_sum(s);
            if (lane == 0) a.pooled[(size_t)g * pw + c] = s;
        }
        if (a.att)
            for (int n = tid; n < N; n += EMB_THREADS) a.att[(size_t)g * N + n] = xx[n];
        if (a.emb)
            for (int e = tid; e < N * pw; e += EMB_THREADS) {
                const int n = e / pw, c = e - n * pw;
                a.emb[((size_t)g * N + n) * pw + c] = c < m.f3 ? E[(size_t)c * N + n] : 0.f;
            }
        __syncthreads();                                          // the scratch is reused by the next graph
    }
}

int launch_embed_generic(const sgpr_handle* h, const EmbedArgs& a, int N, int k, void* ws, hipStream_t stream) {
    if (a.G == 0) return SGPR_OK;
    const int slots = generic_embed_slots(h, a.G);
    const size_t per_wg = generic_scratch_floats(h->gm, N, k);
    const int pw = h->generic_only ? h->gm.f3 : kF3;
    const bool in_lds = generic_in_lds(h->gm, N, k);
    static LdsLimitOnce once;
    if (in_lds) {
        if (int rc = raise_lds_limit(&once, reinterpret_cast<const void*>(&generic_embed_kernel<true>), (int)kGenericLdsBytes,
                                     "generic_embed_kernel"))
            return rc;
        hipLaunchKernelGGL(generic_embed_kernel<true>, dim3(slots), dim3(EMB_THREADS), per_wg * sizeof(float), stream, h->gm, a, N,
                           k, nullptr, per_wg, pw);
    } else {
        hipLaunchKernelGGL(generic_embed_kernel<false>, dim3(slots), dim3(EMB_THREADS), 0, stream, h->gm, a, N, k,
                           static_cast<float*>(ws), per_wg, pw);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "generic_embed_kernel launch");
    return SGPR_OK;
}

// ---- pair-coupled tail: TenorNetworkModule.forward (layers_batch.py:70-83) + FC head (sg_net.py:131-136) at any width
//
// The tensor network of one pair, by a whole workgroup: prod[j * T + t] = (sum_i e1[i] W[i][j][t]) * e2[j] with the
// threads walking the flattened (j, t) index (W's rows of F * T floats are contiguous: coalesced), then thread t adds the
// F products of its neuron in index order, the block term and the bias -> hout[t] (after the ReLU).  Deterministic.
constexpr int TAIL_THREADS = 256;
__device__ void pair_ntn_wg(const float* __restrict__ w, const float* __restrict__ wb, const float* __restrict__ bias,
                            const float* __restrict__ e1, const float* __restrict__ e2, const int F, const int T,
                            float* __restrict__ prod, float* __restrict__ se, float* __restrict__ hout) {
    const int tid = threadIdx.x;
    for (int q = tid; q < 2 * F; q += TAIL_THREADS) se[q] = q < F ? e1[q] : e2[q - F];
    __syncthreads();
    const int FT = F * T;
    for (int q0 = 0; q0 < FT; q0 += 4 * TAIL_THREADS) {
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int i = 0; i < F; ++i) {
            const float x = se[i];
            const float* wr = w + (size_t)i * FT + q0 + tid;
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (q0 + u * TAIL_THREADS + tid < FT) acc[u] = fmaf(x, wr[u * TAIL_THREADS], acc[u]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int q = q0 + u * TAIL_THREADS + tid;
            if (q < FT) prod[q] = acc[u] * se[F + q / T];
        }
    }
    __syncthreads();
    for (int t = tid; t < T; t += TAIL_THREADS) {
        float s = 0.f;
        for (int j = 0; j < F; ++j) s += prod[j * T + t];
        float blk = 0.f;
        for (int q = 0; q < 2 * F; ++q) blk = fmaf(wb[(size_t)t * 2 * F + q], se[q], blk);
        hout[t] = fmaxf(s + blk + bias[t], 0.f);
    }
    __syncthreads();
}

// FC head of one pair by the first wave's lanes (bottleneck neuron o per lane), result in lane 0 of wave 0
__device__ float pair_head_wg(const GenericModel& m, const float* __restrict__ hbuf, float* __restrict__ gbuf) {
    const int tid = threadIdx.x;
    for (int o = tid; o < m.B; o += TAIL_THREADS) {
        float gsum = m.fc1_b[o];
        for (int t = 0; t < m.T; ++t) gsum = fmaf(m.fc1_w[(size_t)o * m.T + t], hbuf[t], gsum);
        gbuf[o] = fmaxf(gsum, 0.f);
    }
    __syncthreads();
    float z = 0.f;
    if (tid == 0) {
        z = m.fc2_b[0];
        for (int o = 0; o < m.B; ++o) z = fmaf(m.fc2_w[o], gbuf[o], z);
        z = 1.f / (1.f + expf(-z));
    }
    __syncthreads();
    return z;
}

// list form: pair p = (i1 ? i1[p] : p, i2 ? i2[p] : p), one workgroup per pair (grid-stride)
__global__ __launch_bounds__(TAIL_THREADS) void generic_score_list_kernel(const GenericModel m, const float* __restrict__ p1,
                                                                          const int32_t* __restrict__ i1, const float* __restrict__ p2,
                                                                          const int32_t* __restrict__ i2, const int64_t P,
                                                                          float* __restrict__ score, const int pw) {
    extern __shared__ __attribute__((aligned(16))) float tail_smem[];
    float* prod = tail_smem;                                      // [F * T]
    float* se = prod + m.f3 * m.T;                                // [2 F]
    float* hbuf = se + 2 * m.f3;                                  // [T]
    float* gbuf = hbuf + m.T;                                     // [B]
    for (int64_t pair = blockIdx.x; pair < P; pair += gridDim.x) {
        const int64_t r1 = i1 ? i1[pair] : pair, r2 = i2 ? i2[pair] : pair;
        pair_ntn_wg(m.ntn_w, m.ntn_wb, m.ntn_bias, p1 + r1 * pw, p2 + r2 * pw, m.f3, m.T, prod, se, hbuf);
        const float z = pair_head_wg(m, hbuf, gbuf);
        if (threadIdx.x == 0) score[pair] = z;
    }
}

// grouped list (sgpr_score_pair_list's plan, sgpr.h: row_ids | item_row | item_beg | cols | pos): a workgroup walks work
// items (<= 16 listed pairs of ONE row graph); every pair through the list kernel's arithmetic, so a grouped list gives the
// bits of sgpr_score_pairs on the same pairs
__global__ __launch_bounds__(TAIL_THREADS) void generic_score_plan_kernel(const GenericModel m, const float* __restrict__ rows,
                                                                          const float* __restrict__ colsv,
                                                                          const int32_t* __restrict__ row_ids,
                                                                          const int32_t* __restrict__ item_row,
                                                                          const int32_t* __restrict__ item_beg,
                                                                          const int32_t* __restrict__ cols,
                                                                          const int32_t* __restrict__ pos, const int NI,
                                                                          float* __restrict__ score, const int pw) {
    extern __shared__ __attribute__((aligned(16))) float tail_smem[];
    float* prod = tail_smem;                                      // [F * T]
    float* se = prod + m.f3 * m.T;                                // [2 F]
    float* hbuf = se + 2 * m.f3;                                  // [T]
    float* gbuf = hbuf + m.T;                                     // [B]
    for (int it = blockIdx.x; it < NI; it += gridDim.x) {
        const int64_t r1 = row_ids[item_row[it]];
        for (int q = item_beg[it]; q < item_beg[it + 1]; ++q) {
            pair_ntn_wg(m.ntn_w, m.ntn_wb, m.ntn_bias, rows + r1 * pw, colsv + (int64_t)cols[q] * pw, m.f3, m.T, prod, se, hbuf);
            const float z = pair_head_wg(m, hbuf, gbuf);
            if (threadIdx.x == 0) score[pos[q]] = z;
        }
    }
}

int launch_score_plan_generic(const sgpr_handle* h, const float* rows, const float* cols, const int32_t* plan, int NR, int NI,
                              int64_t P, float* score, hipStream_t stream) {
    if (P == 0 || NI == 0) return SGPR_OK;
    const GenericModel& m = h->gm;
    const int pw = h->generic_only ? m.f3 : kF3;
    const int32_t* row_ids = plan;
    const int32_t* item_row = row_ids + NR;
    const int32_t* item_beg = item_row + NI;
    const int32_t* pc = item_beg + NI + 1;
    const int32_t* pos = pc + P;
    const int64_t cap = (int64_t)16 * h->num_cus;
    const unsigned blocks = (unsigned)(NI < cap ? NI : cap);
    const size_t lds = ((size_t)m.f3 * m.T + 2 * m.f3 + m.T + m.B) * sizeof(float);
    hipLaunchKernelGGL(generic_score_plan_kernel, dim3(blocks), dim3(TAIL_THREADS), lds, stream, m, rows, cols, row_ids, item_row,
                       item_beg, pc, pos, NI, score, pw);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "generic_score_plan_kernel launch");
    return SGPR_OK;
}

// dense rectangle R x M: a workgroup owns one row graph and a range of column tiles.  The bilinear form is hoisted per
// row (as the tuned tail does): App[j][t] = sum_i e1[i] W[i][j][t] + Wb[t][F + j], u[t] = Wb[t][:F] . e1 + bias[t]; a
// pair is then s_t = sum_j App[j][t] e2[j] + u[t] - F * T multiply-adds per pair instead of F * F * T - and the head.
// A thread owns a column: the tensor neurons in registers (TMAX of them: instances 16 / 32 / 64), App / the head's
// weights broadcast from LDS.
template <int TMAX>
__global__ __launch_bounds__(TAIL_THREADS) void generic_score_rect_kernel(const GenericModel m, const float* __restrict__ rows,
                                                                          const float* __restrict__ cols, const int R, const int M,
                                                                          float* __restrict__ score, const int64_t ld, const int pw,
                                                                          const int tiles_per_wg, const unsigned* __restrict__ gate) {
    extern __shared__ __attribute__((aligned(16))) float tail_smem[];
    if (gate && *gate == 0u) return;      // behind the matrix-core tail (sgpr_wide.hip): only a rectangle it left to this kernel
    const int F = m.f3, T = m.T, B = m.B, tid = threadIdx.x;
    float* App = tail_smem;                                       // [F][TMAX]
    float* u = App + F * TMAX;                                    // [TMAX]
    float* fc1 = u + TMAX;                                        // [B][TMAX]
    float* fcb = fc1 + B * TMAX;                                  // [B] bias, [B] scoring weights
    float* se = fcb + 2 * B;                                      // [F] the row's pooled vector
    const int r = blockIdx.y;
    const float* e1 = rows + (size_t)r * pw;
    for (int q = tid; q < F; q += TAIL_THREADS) se[q] = e1[q];
    for (int q = tid; q < B * TMAX; q += TAIL_THREADS) {
        const int o = q / TMAX, t = q - o * TMAX;
        fc1[q] = t < T ? m.fc1_w[(size_t)o * T + t] : 0.f;
    }
    for (int q = tid; q < B; q += TAIL_THREADS) {
        fcb[q] = m.fc1_b[q];
        fcb[B + q] = m.fc2_w[q];
    }
    __syncthreads();
    for (int q = tid; q < F * TMAX; q += TAIL_THREADS) {
        const int j = q / TMAX, t = q - j * TMAX;
        float acc = 0.f;
        if (t < T) {
            for (int i = 0; i < F; ++i) acc = fmaf(se[i], m.ntn_w[((size_t)i * F + j) * T + t], acc);
            acc += m.ntn_wb[(size_t)t * 2 * F + F + j];
        }
        App[q] = acc;
    }
    for (int t = tid; t < TMAX; t += TAIL_THREADS) {
        float blk = 0.f;
        if (t < T) {
            for (int q = 0; q < F; ++q) blk = fmaf(m.ntn_wb[(size_t)t * 2 * F + q], se[q], blk);
            blk += m.ntn_bias[t];
        }
        u[t] = blk;
    }
    __syncthreads();
    const float fc2b = m.fc2_b[0];
    const int tile0 = blockIdx.x * tiles_per_wg;
    for (int tile = tile0; tile < tile0 + tiles_per_wg; ++tile) {
        const int c = tile * TAIL_THREADS + tid;
        if (c >= M) break;                                        // (no barrier below: a thread leaves alone)
        const float* e2 = cols + (size_t)c * pw;
        float acc[TMAX];
#pragma unroll
        for (int t = 0; t < TMAX; ++t) acc[t] = 0.f;
        for (int j = 0; j < F; ++j) {
            const float x = e2[j];
            const float4* ap = reinterpret_cast<const float4*>(App + j * TMAX);
#pragma unroll
            for (int t4 = 0; t4 < TMAX / 4; ++t4) {
                const float4 wv = ap[t4];
                acc[4 * t4 + 0] = fmaf(wv.x, x, acc[4 * t4 + 0]);
                acc[4 * t4 + 1] = fmaf(wv.y, x, acc[4 * t4 + 1]);
                acc[4 * t4 + 2] = fmaf(wv.z, x, acc[4 * t4 + 2]);
                acc[4 * t4 + 3] = fmaf(wv.w, x, acc[4 * t4 + 3]);
            }
        }
#pragma unroll
        for (int t = 0; t < TMAX; ++t) acc[t] = fmaxf(acc[t] + u[t], 0.f);      // (neurons >= T: 0 + 0)
        float z = fc2b;
        for (int o = 0; o < B; ++o) {
            const float4* fp = reinterpret_cast<const float4*>(fc1 + o * TMAX);
            float gsum = fcb[o];
#pragma unroll
            for (int t4 = 0; t4 < TMAX / 4; ++t4) {
                const float4 wv = fp[t4];
                gsum = fmaf(wv.x, acc[4 * t4 + 0], gsum);
                gsum = fmaf(wv.y, acc[4 * t4 + 1], gsum);
                gsum = fmaf(wv.z, acc[4 * t4 + 2], gsum);
                gsum = fmaf(wv.w, acc[4 * t4 + 3], gsum);
            }
            z = fmaf(fcb[B + o], fmaxf(gsum, 0.f), z);
        }
        score[(size_t)r * ld + c] = 1.f / (1.f + expf(-z));
    }
}

template <int TMAX>
static int launch_rect(const sgpr_handle* h, const float* rows, const float* cols, int R, int M, float* score, int64_t ld,
                       int pw, hipStream_t stream, const unsigned* gate) {
    const GenericModel& m = h->gm;
    const int tiles = (M + TAIL_THREADS - 1) / TAIL_THREADS;
    // enough workgroups to fill the GPU (8 per CU), as few recomputations of a row's hoisted form as that allows
    int split = (8 * h->num_cus + R - 1) / R;
    if (split < 1) split = 1;
    if (split > tiles) split = tiles;
    const int tiles_per_wg = (tiles + split - 1) / split;
    const int gx = (tiles + tiles_per_wg - 1) / tiles_per_wg;
    const size_t lds = ((size_t)m.f3 * TMAX + TMAX + (size_t)m.B * TMAX + 2 * m.B + m.f3) * sizeof(float);
    for (int r0 = 0; r0 < R; r0 += 65535) {                       // (gridDim.y)
        const int rn = R - r0 < 65535 ? R - r0 : 65535;
        hipLaunchKernelGGL(generic_score_rect_kernel<TMAX>, dim3(gx, rn), dim3(TAIL_THREADS), lds, stream, m,
                           rows + (size_t)r0 * pw, cols, rn, M, score + (size_t)r0 * ld, ld, pw, tiles_per_wg, gate);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return hip_fail(e, "generic_score_rect_kernel launch");
    }
    return SGPR_OK;
}

int launch_score_generic(const sgpr_handle* h, const float* p1, const int32_t* i1, const float* p2, const int32_t* i2,
                         int64_t P, int M, float* score, int64_t ld, hipStream_t stream, const unsigned* d_gate) {
    if (P == 0) return SGPR_OK;
    const int pw = h->generic_only ? h->gm.f3 : kF3;
    const GenericModel& m = h->gm;
    if (M > 0) {                                                  // dense rectangle: p1 = rows [P / M], p2 = columns [M]
        const int R = (int)(P / M);
        if (m.T <= 16) return launch_rect<16>(h, p1, p2, R, M, score, ld, pw, stream, d_gate);
        if (m.T <= 32) return launch_rect<32>(h, p1, p2, R, M, score, ld, pw, stream, d_gate);
        return launch_rect<64>(h, p1, p2, R, M, score, ld, pw, stream, d_gate);
    }
    const int64_t cap = (int64_t)16 * h->num_cus;
    const unsigned blocks = (unsigned)(P < cap ? P : cap);
    const size_t lds = ((size_t)m.f3 * m.T + 2 * m.f3 + m.T + m.B) * sizeof(float);
    hipLaunchKernelGGL(generic_score_list_kernel, dim3(blocks), dim3(TAIL_THREADS), lds, stream, m, p1, i1, p2, i2, P, score, pw);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "generic_score_list_kernel launch");
    return SGPR_OK;
}

// ---- stand-alone modules at any width (the fixed-width forms live in sgpr_modules.hip) -------------------------------

// dgcnn.knn (dgcnn.py:14-20) for N <= SGPR_ANY_MAX_NODES, k <= SGPR_ANY_MAX_K: x [B][C][N] -> idx [B][N][k] int64, best
// first, the lower index first among equal keys.  One wave per row; the arithmetic of generic_edgeconv's selection.
__global__ __launch_bounds__(GEN_THREADS) void generic_knn_kernel(const float* __restrict__ x, const int C, const int N,
                                                                  const int k, long long* __restrict__ idx) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.y;
    const int i = blockIdx.x * (GEN_THREADS / 64) + (threadIdx.x >> 6);
    if (i >= N) return;
    const float* xb = x + (size_t)b * C * N;
    float xi2 = __fmul_rn(xb[i], xb[i]);
    for (int c = 1; c < C; ++c) xi2 = __fadd_rn(xi2, __fmul_rn(xb[(size_t)c * N + i], xb[(size_t)c * N + i]));
    float key[GEN_MAX_PER_LANE];
#pragma unroll
    for (int q = 0; q < GEN_MAX_PER_LANE; ++q) {
        const int j = lane + 64 * q;
        key[q] = INFINITY;
        if (j < N) {
            float xj2 = __fmul_rn(xb[j], xb[j]);
            float dot = __fmul_rn(xb[i], xb[j]);
            for (int c = 1; c < C; ++c) {
                const float vi = xb[(size_t)c * N + i], vj = xb[(size_t)c * N + j];
                xj2 = __fadd_rn(xj2, __fmul_rn(vj, vj));
                dot = fmaf(vi, vj, dot);
            }
            key[q] = __fsub_rn(xi2, fmaf(2.f, dot, -xj2));
        }
    }
    // total order (key, index) on the order-preserving image of the key: +inf keys (overflowing features) and NaN keys rank
    // last but ARE ranked - every row gets k distinct indices, as knn_select_row and the tuned kernel give
    auto image = [](float f) {
        if (f != f) return 0xffffffffu;                       // NaN after everything
        f += 0.0f;
        const unsigned u = __float_as_uint(f);
        return u ^ ((unsigned)((int)u >> 31) | 0x80000000u);
    };
    unsigned okey[GEN_MAX_PER_LANE];
    unsigned taken = 0u;                                      // bit q: candidate lane + 64 q is in the list already
#pragma unroll
    for (int q = 0; q < GEN_MAX_PER_LANE; ++q) okey[q] = image(key[q]);
    long long* out = idx + ((size_t)b * N + i) * k;
    for (int m = 0; m < k; ++m) {
        unsigned long long best = ~0ull;                      // (image << 32 | index): one comparison orders both
#pragma unroll
        for (int q = 0; q < GEN_MAX_PER_LANE; ++q) {
            const int j = lane + 64 * q;
            const unsigned long long cand = ((unsigned long long)okey[q] << 32) | (unsigned)j;
            if (j < N && !((taken >> q) & 1u) && cand < best) best = cand;
        }
#pragma unroll
        for (int sft = 1; sft < 64; sft <<= 1) {
            const unsigned long long ob = __shfl_xor(best, sft);
            if (ob < best) best = ob;
        }
        const int bj = (int)(unsigned)best;
        if (lane == 0) out[m] = best == ~0ull ? 0 : bj;
#pragma unroll
        for (int q = 0; q < GEN_MAX_PER_LANE; ++q)
            if (best != ~0ull && lane + 64 * q == bj) taken |= 1u << q;
    }
}

int launch_knn_any(const float* x, int B, int C, int N, int k, int64_t* idx, hipStream_t stream) {
    if (B == 0) return SGPR_OK;
    for (int b0 = 0; b0 < B; b0 += 65535) {                   // (grid.y holds 65 535 graphs)
        const int nb = B - b0 < 65535 ? B - b0 : 65535;
        hipLaunchKernelGGL(generic_knn_kernel, dim3((N + 3) / 4, nb), dim3(GEN_THREADS), 0, stream, x + (size_t)b0 * C * N, C, N, k,
                           reinterpret_cast<long long*>(idx) + (size_t)b0 * N * k);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return hip_fail(e, "generic_knn_kernel launch");
    }
    return SGPR_OK;
}

// AttentionModule.forward (layers_batch.py:28-39) at width F <= SGPR_ANY_MAX_FILTERS_3: one workgroup per graph, nodes in
// chunks of 256 (scores of a chunk in LDS, then the channels' weighted sums)
__global__ __launch_bounds__(GEN_THREADS) void generic_attention_kernel(const float* __restrict__ w, const float* __restrict__ emb,
                                                                        const int N, const int F, float* __restrict__ rep,
                                                                        float* __restrict__ att) {
    __shared__ float red[SGPR_GENERIC_MAX_F3];
    __shared__ float ctx[SGPR_GENERIC_MAX_F3];
    __shared__ float sc[GEN_THREADS];
    const int tid = threadIdx.x;
    const float* E = emb + (size_t)blockIdx.x * N * F;
    for (int c = tid; c < F; c += GEN_THREADS) {
        float s = 0.f;
        for (int n = 0; n < N; ++n) s += E[(size_t)n * F + c];
        red[c] = s / (float)N;
    }
    __syncthreads();
    for (int c = tid; c < F; c += GEN_THREADS) {
        float g = 0.f;
        for (int r = 0; r < F; ++r) g = fmaf(red[r], w[(size_t)r * F + c], g);
        ctx[c] = tanhf(g);
    }
    __syncthreads();
    float acc = 0.f;                                                // channel tid's weighted sum (tid < F)
    for (int n0 = 0; n0 < N; n0 += GEN_THREADS) {
        const int n = n0 + tid;
        if (n < N) {
            float d = 0.f;
            for (int c = 0; c < F; ++c) d = fmaf(E[(size_t)n * F + c], ctx[c], d);
            const float sg = 1.f / (1.f + expf(-d));
            sc[tid] = sg;
            if (att) att[(size_t)blockIdx.x * N + n] = sg;
        }
        __syncthreads();
        const int cnt = N - n0 < GEN_THREADS ? N - n0 : GEN_THREADS;
        if (tid < F)
            for (int q = 0; q < cnt; ++q) acc = fmaf(sc[q], E[(size_t)(n0 + q) * F + tid], acc);
        __syncthreads();
    }
    if (tid < F) rep[(size_t)blockIdx.x * F + tid] = acc;
}

int launch_attention_any(const float* w, const float* emb, int B, int N, int F, float* rep, float* att, hipStream_t stream) {
    if (B == 0) return SGPR_OK;
    hipLaunchKernelGGL(generic_attention_kernel, dim3(B), dim3(GEN_THREADS), 0, stream, w, emb, N, F, rep, att);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "generic_attention_kernel launch");
    return SGPR_OK;
}

// TenorNetworkModule.forward (layers_batch.py:70-83) at F <= SGPR_ANY_MAX_FILTERS_3, T <= SGPR_ANY_MAX_NEURONS: a workgroup per pair
__global__ __launch_bounds__(TAIL_THREADS) void generic_ntn_kernel(const float* __restrict__ w, const float* __restrict__ wb,
                                                                   const float* __restrict__ bias, const float* __restrict__ e1a,
                                                                   const float* __restrict__ e2a, const int64_t P, const int F,
                                                                   const int T, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float tail_smem[];
    float* prod = tail_smem;
    float* se = prod + F * T;
    float* hbuf = se + 2 * F;
    for (int64_t pair = blockIdx.x; pair < P; pair += gridDim.x) {
        pair_ntn_wg(w, wb, bias, e1a + pair * F, e2a + pair * F, F, T, prod, se, hbuf);
        for (int t = threadIdx.x; t < T; t += TAIL_THREADS) out[pair * T + t] = hbuf[t];
        __syncthreads();
    }
}

int launch_ntn_any(const float* w, const float* wb, const float* bias, const float* e1, const float* e2, int64_t P, int F,
                   int T, float* out, hipStream_t stream) {
    if (P == 0) return SGPR_OK;
    const unsigned blocks = (unsigned)(P < 4096 ? P : 4096);
    const size_t lds = ((size_t)F * T + 2 * F + T) * sizeof(float);
    hipLaunchKernelGGL(generic_ntn_kernel, dim3(blocks), dim3(TAIL_THREADS), lds, stream, w, wb, bias, e1, e2, P, F, T, out);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "generic_ntn_kernel launch");
    return SGPR_OK;
}

}  // namespace sgpr
