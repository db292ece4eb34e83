// Any-shape form of the path: SG.dgcnn_conv_pass + AttentionModule (sg_net.py:79-110, dgcnn.py:14-49, layers_batch.py:28-39)
// and the pair-coupled tail (layers_batch.py:70-83, sg_net.py:131-136) for everything the tuned kernels are not built for -
// architectures LARGER than {12 labels, 64, 64, 32, 16, 16} in any hyper-parameter (the reference builds any:
// sg_net.py:40-76, parser_sg.py:12-18), node_num beyond 256, K beyond 32.  Correctness first, plain fp32: the reference's
// formulation with the two algebraic steps that do not change a value's meaning (eval BatchNorm folded into the 1x1
// convolutions; W.[x_j - x_i ; x_i] = W1.x_j + (W2 - W1).x_i with the max over neighbours taken on the first term) -
// no duplicate-slot compression, no super-nodes, no matrix cores, activations in a global scratch area per resident
// workgroup.  Orders of magnitude slower than the tuned path per graph and still thousands of times the reference on a CPU;
// the tuned kernels keep every shape they serve (every shipped checkpoint).
#include <math.h>

#include "sgpr_internal.hpp"

namespace sgpr {

constexpr int GEN_THREADS = 256;
constexpr int GEN_MAX_PER_LANE = SGPR_GENERIC_MAX_NODES / 64;     // candidates of a row one lane holds in the selection
constexpr float kSlope = 0.2f;                                    // LeakyReLU(0.2), sg_net.py:53

// floats of scratch one resident workgroup needs for graphs of N slots: two activation buffers, the a / b terms, the
// first branch's output, squared norms, the neighbour lists
static size_t generic_scratch_floats(const GenericModel& m, int N, int k) {
    return (size_t)N * ((size_t)4 * m.cmax + m.f3 + 1 + k) + 64;
}

int generic_embed_slots(const sgpr_handle* h, int G) {
    const int cap = 2 * h->num_cus;
    return G < cap ? G : cap;
}

size_t generic_embed_ws_bytes(const sgpr_handle* h, int G, int N, int k) {
    return (size_t)generic_embed_slots(h, G) * generic_scratch_floats(h->gm, N, k) * sizeof(float);
}

__device__ __forceinline__ float lrelu(float v) { return v > 0.f ? v : kSlope * v; }

// One EdgeConv block on X [N][cin] -> Y [N][cout]:  kNN in X's own space (dgcnn.knn), then
//   Y[i] = lrelu(max_{j in knn(i)} a[j] + b[i]),  a = Wa X,  b = Wb X + t   (BatchNorm folded: sgpr_create)
__device__ void generic_edgeconv(const float* __restrict__ X, float* __restrict__ Y, float* __restrict__ A, float* __restrict__ Bm,
                                 float* __restrict__ xx, int* __restrict__ idx, const int N, const int k, const int cin,
                                 const int cout, const float* __restrict__ wa, const float* __restrict__ wb,
                                 const float* __restrict__ tb, float* __restrict__ dbg_y, int32_t* __restrict__ dbg_idx) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // |x|^2 (dgcnn.py:16: torch.sum(x ** 2, dim=1)) as torch evaluates it: squares rounded, summed in channel order with
    // every step rounded; the dot product below is the FMA chain in channel order.  For the coordinate layer (3 channels)
    // these are the reference's keys bit for bit (the operation order the tuned kernel restates too: gram_xyz_direct in
    // sgpr_embed.hip, pinned against torch in tests/test_oracle_golden.py); wider layers agree to fp32 rounding.
    for (int n = tid; n < N; n += GEN_THREADS) {
        float s = __fmul_rn(X[(size_t)n * cin], X[(size_t)n * cin]);
        for (int c = 1; c < cin; ++c) s = __fadd_rn(s, __fmul_rn(X[(size_t)n * cin + c], X[(size_t)n * cin + c]));
        xx[n] = s;
    }
    __syncthreads();
    // kNN: a wave per row; a lane keeps the keys of candidates lane, lane + 64, ... in registers; k rounds of a wave-wide
    // arg-min under the total order (key, index).  key = -pd,  pd = (-|x_j|^2 + 2 x_i.x_j) - |x_i|^2  (dgcnn.py:15-17)
    for (int i = wave; i < N; i += GEN_THREADS / 64) {
        float key[GEN_MAX_PER_LANE];
#pragma unroll
        for (int q = 0; q < GEN_MAX_PER_LANE; ++q) {
            const int j = lane + 64 * q;
            key[q] = INFINITY;
            if (j < N) {
                float dot = __fmul_rn(X[(size_t)i * cin], X[(size_t)j * cin]);
                for (int c = 1; c < cin; ++c) dot = fmaf(X[(size_t)i * cin + c], X[(size_t)j * cin + c], dot);
                key[q] = __fsub_rn(xx[i], fmaf(2.f, dot, -xx[j]));       // -pd = |x_i|^2 - (-|x_j|^2 - inner), inner = -2 dot
            }
        }
        for (int m = 0; m < k; ++m) {
            float best = INFINITY;
            int bj = 0x7fffffff;
#pragma unroll
            for (int q = 0; q < GEN_MAX_PER_LANE; ++q) {         // (ascending candidate index: the first minimum wins)
                const int j = lane + 64 * q;
                if (j < N && key[q] < best) {
                    best = key[q];
                    bj = j;
                }
            }
#pragma unroll
            for (int s = 1; s < 64; s <<= 1) {
                const float ob = __shfl_xor(best, s);
                const int oj = __shfl_xor(bj, s);
                if (ob < best || (ob == best && oj < bj)) {
                    best = ob;
                    bj = oj;
                }
            }
            // (finite inputs leave k <= N finite keys; a row of infinities / NaNs falls back to slot 0)
            if (lane == 0) {
                idx[(size_t)i * k + m] = bj == 0x7fffffff ? 0 : bj;
                if (dbg_idx) dbg_idx[(size_t)i * k + m] = bj == 0x7fffffff ? 0 : bj;
            }
#pragma unroll
            for (int q = 0; q < GEN_MAX_PER_LANE; ++q)
                if (lane + 64 * q == bj) key[q] = INFINITY;      // taken
        }
    }
    // a = Wa x, b = Wb x + t per (node, output channel)
    for (int e = tid; e < N * cout; e += GEN_THREADS) {
        const int n = e / cout, co = e - n * cout;
        const float* x = X + (size_t)n * cin;
        float a = 0.f, b = 0.f;
        for (int c = 0; c < cin; ++c) {
            a = fmaf(wa[(size_t)co * cin + c], x[c], a);
            b = fmaf(wb[(size_t)co * cin + c], x[c], b);
        }
        A[e] = a;
        Bm[e] = b + tb[co];
    }
    __syncthreads();
    for (int e = tid; e < N * cout; e += GEN_THREADS) {
        const int n = e / cout, co = e - n * cout;
        float mx = -INFINITY;
        for (int m = 0; m < k; ++m) mx = fmaxf(mx, A[(size_t)idx[(size_t)n * k + m] * cout + co]);
        Y[e] = lrelu(mx + Bm[e]);
        if (dbg_y) dbg_y[(size_t)n * 64 + co] = Y[e];               // (sgpr_embed_debug's dump rows are 64 floats)
    }
    __syncthreads();
}

__global__ __launch_bounds__(GEN_THREADS) void generic_embed_kernel(const GenericModel m, const EmbedArgs a, const int N,
                                                                    const int k, float* __restrict__ scratch,
                                                                    const size_t per_wg, const int pw) {
    __shared__ float red[SGPR_GENERIC_MAX_F3];
    __shared__ float ctx[SGPR_GENERIC_MAX_F3];
    const int tid = threadIdx.x;
    float* base = scratch + (size_t)blockIdx.x * per_wg;
    float* X0 = base;
    float* X1 = X0 + (size_t)N * m.cmax;
    float* A = X1 + (size_t)N * m.cmax;
    float* Bm = A + (size_t)N * m.cmax;
    float* Y3 = Bm + (size_t)N * m.cmax;                          // [N][f3] the xyz branch's output
    float* xx = Y3 + (size_t)N * m.f3;
    int* idx = reinterpret_cast<int*>(xx + N);
    for (int slot = blockIdx.x; slot < a.G; slot += gridDim.x) {
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
            for (int c = tid; c < pw; c += GEN_THREADS) a.pooled[(size_t)g * pw + c] = __int_as_float(0x7fc00000);
            continue;
        }
        for (int branch = 0; branch < 2; ++branch) {
            const int C0 = branch == 0 ? 3 : m.L;
            for (int e = tid; e < N * C0; e += GEN_THREADS) {
                const int n = e / C0, c = e - n * C0;
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
                for (int n = tid; n < ragc; n += GEN_THREADS) {
                    const int lab = a.rag_lab[rag0 + n];
                    if (lab < 0 || lab >= m.L) atomicOr(a.status, 1);
                }
            __syncthreads();
            float* cur = X0;
            float* nxt = X1;
            for (int l = 0; l < 3; ++l) {
                const int L6 = branch * 3 + l;
                float* dst = (branch == 0 && l == 2) ? Y3 : nxt;  // the xyz branch's output waits in a buffer of its own
                generic_edgeconv(cur, dst, A, Bm, xx, idx, N, k, m.cin[L6], m.cout[L6], m.wa[L6], m.wb[L6], m.tb[L6],
                                 a.dbg_layers ? a.dbg_layers + ((size_t)g * 6 + L6) * N * 64 : nullptr,
                                 a.dbg_knn ? a.dbg_knn + ((size_t)g * 6 + L6) * N * k : nullptr);
                if (dst == nxt) {
                    nxt = cur;
                    cur = dst;
                }
            }
            if (branch == 1) {
                // `cur` holds sem3 [N][f3]; conv_end on cat(xyz3, sem3) (sg_net.py:104-109) -> E (into A)
                const float* S3 = cur;
                for (int e = tid; e < N * m.f3; e += GEN_THREADS) {
                    const int n = e / m.f3, co = e - n * m.f3;
                    float v = 0.f;
                    for (int c = 0; c < m.f3; ++c) v = fmaf(m.w_end[(size_t)co * 2 * m.f3 + c], Y3[(size_t)n * m.f3 + c], v);
                    for (int c = 0; c < m.f3; ++c) v = fmaf(m.w_end[(size_t)co * 2 * m.f3 + m.f3 + c], S3[(size_t)n * m.f3 + c], v);
                    A[e] = lrelu(v + m.t_end[co]);
                }
                __syncthreads();
            }
        }
        const float* E = A;                                       // [N][f3]
        // ---- attention pooling (layers_batch.py:28-39: no pad mask, divisor N)
        for (int c = tid; c < m.f3; c += GEN_THREADS) {
            float s = 0.f;
            for (int n = 0; n < N; ++n) s += E[(size_t)n * m.f3 + c];
            red[c] = s / (float)N;
        }
        __syncthreads();
        for (int c = tid; c < m.f3; c += GEN_THREADS) {
            float gsum = 0.f;
            for (int r = 0; r < m.f3; ++r) gsum = fmaf(red[r], m.att_w[(size_t)r * m.f3 + c], gsum);
            ctx[c] = tanhf(gsum);
        }
        __syncthreads();
        for (int n = tid; n < N; n += GEN_THREADS) {
            float d = 0.f;
            for (int c = 0; c < m.f3; ++c) d = fmaf(E[(size_t)n * m.f3 + c], ctx[c], d);
            xx[n] = 1.f / (1.f + expf(-d));
        }
        __syncthreads();
        for (int c = tid; c < pw; c += GEN_THREADS) {
            float s = 0.f;
            if (c < m.f3)
                for (int n = 0; n < N; ++n) s = fmaf(xx[n], E[(size_t)n * m.f3 + c], s);
            a.pooled[(size_t)g * pw + c] = s;
        }
        if (a.att)
            for (int n = tid; n < N; n += GEN_THREADS) a.att[(size_t)g * N + n] = xx[n];
        if (a.emb)
            for (int e = tid; e < N * pw; e += GEN_THREADS) {
                const int n = e / pw, c = e - n * pw;
                a.emb[((size_t)g * N + n) * pw + c] = c < m.f3 ? E[(size_t)n * m.f3 + c] : 0.f;
            }
        __syncthreads();                                          // the scratch is reused by the next graph
    }
}

int launch_embed_generic(const sgpr_handle* h, const EmbedArgs& a, int N, int k, void* ws, hipStream_t stream) {
    if (a.G == 0) return SGPR_OK;
    const int slots = generic_embed_slots(h, a.G);
    const size_t per_wg = generic_scratch_floats(h->gm, N, k);
    const int pw = h->generic_only ? h->gm.f3 : kF3;
    hipLaunchKernelGGL(generic_embed_kernel, dim3(slots), dim3(GEN_THREADS), 0, stream, h->gm, a, N, k, static_cast<float*>(ws),
                       per_wg, pw);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "generic_embed_kernel launch");
    return SGPR_OK;
}

// ---- pair-coupled tail, one wave per pair: TenorNetworkModule.forward (layers_batch.py:70-83) + FC head (sg_net.py:131-136)
//      pair p = (rows[i1 ? i1[p] : p / M_or_1 ...]): list form (M == 0) or dense rectangle R x M (score [R][ld])
__global__ __launch_bounds__(256) void generic_score_kernel(const GenericModel m, const float* __restrict__ p1,
                                                            const int32_t* __restrict__ i1, const float* __restrict__ p2,
                                                            const int32_t* __restrict__ i2, const int64_t P, const int M,
                                                            float* __restrict__ score, const int64_t ld, const int pw) {
    __shared__ float hbuf[4][SGPR_GENERIC_MAX_T];
    __shared__ float gbuf[4][SGPR_GENERIC_MAX_T];   // (bottleneck neurons: same cap)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t pair = (int64_t)blockIdx.x * 4 + wave;
    const bool live = pair < P;
    int64_t r1 = 0, r2 = 0, out = 0;
    if (live) {
        if (M > 0) {                                              // dense rectangle
            r1 = pair / M;
            r2 = pair - r1 * M;
            out = r1 * ld + r2;
        } else {
            r1 = i1 ? i1[pair] : pair;
            r2 = i2 ? i2[pair] : pair;
            out = pair;
        }
    }
    const float* e1 = p1 + r1 * pw;
    const float* e2 = p2 + r2 * pw;
    const int F = m.f3, T = m.T;
    // neuron t (lanes t, t + 64, ...): s_t = sum_ij e1_i W[i][j][t] e2_j + Wb[t] . [e1; e2] + bias[t]
    for (int t = lane; live && t < T; t += 64) {
        float s = 0.f;
        for (int j = 0; j < F; ++j) {
            float v = 0.f;
            for (int i = 0; i < F; ++i) v = fmaf(e1[i], m.ntn_w[((size_t)i * F + j) * T + t], v);
            s = fmaf(v, e2[j], s);
        }
        float blk = 0.f;
        for (int q = 0; q < F; ++q) blk = fmaf(m.ntn_wb[(size_t)t * 2 * F + q], e1[q], blk);
        for (int q = 0; q < F; ++q) blk = fmaf(m.ntn_wb[(size_t)t * 2 * F + F + q], e2[q], blk);
        hbuf[wave][t] = fmaxf(s + blk + m.ntn_bias[t], 0.f);
    }
    __syncthreads();
    for (int o = lane; live && o < m.B; o += 64) {
        float gsum = m.fc1_b[o];
        for (int t = 0; t < T; ++t) gsum = fmaf(m.fc1_w[(size_t)o * T + t], hbuf[wave][t], gsum);
        gbuf[wave][o] = fmaxf(gsum, 0.f);
    }
    __syncthreads();
    if (live && lane == 0) {
        float z = m.fc2_b[0];
        for (int o = 0; o < m.B; ++o) z = fmaf(m.fc2_w[o], gbuf[wave][o], z);
        score[out] = 1.f / (1.f + expf(-z));
    }
}

int launch_score_generic(const sgpr_handle* h, const float* p1, const int32_t* i1, const float* p2, const int32_t* i2,
                         int64_t P, int M, float* score, int64_t ld, hipStream_t stream) {
    if (P == 0) return SGPR_OK;
    const int64_t blocks = (P + 3) / 4;
    if (blocks > 0x7fffffffLL) {
        set_error("generic tail: too many pairs for one launch");
        return SGPR_E_INVALID;
    }
    const int pw = h->generic_only ? h->gm.f3 : kF3;
    hipLaunchKernelGGL(generic_score_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, h->gm, p1, i1, p2, i2, P, M, score, ld, pw);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "generic_score_kernel launch");
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
    long long* out = idx + ((size_t)b * N + i) * k;
    for (int m = 0; m < k; ++m) {
        float best = INFINITY;
        int bj = 0x7fffffff;
#pragma unroll
        for (int q = 0; q < GEN_MAX_PER_LANE; ++q) {
            const int j = lane + 64 * q;
            if (j < N && key[q] < best) {
                best = key[q];
                bj = j;
            }
        }
#pragma unroll
        for (int sft = 1; sft < 64; sft <<= 1) {
            const float ob = __shfl_xor(best, sft);
            const int oj = __shfl_xor(bj, sft);
            if (ob < best || (ob == best && oj < bj)) {
                best = ob;
                bj = oj;
            }
        }
        if (lane == 0) out[m] = bj == 0x7fffffff ? 0 : bj;
#pragma unroll
        for (int q = 0; q < GEN_MAX_PER_LANE; ++q)
            if (lane + 64 * q == bj) key[q] = INFINITY;
    }
}

int launch_knn_any(const float* x, int B, int C, int N, int k, int64_t* idx, hipStream_t stream) {
    if (B == 0) return SGPR_OK;
    hipLaunchKernelGGL(generic_knn_kernel, dim3((N + 3) / 4, B), dim3(GEN_THREADS), 0, stream, x, C, N, k,
                       reinterpret_cast<long long*>(idx));
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "generic_knn_kernel launch");
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

// TenorNetworkModule.forward (layers_batch.py:70-83) at F <= SGPR_ANY_MAX_FILTERS_3, T <= SGPR_ANY_MAX_NEURONS: one wave per pair
__global__ __launch_bounds__(256) void generic_ntn_kernel(const float* __restrict__ w, const float* __restrict__ wb,
                                                          const float* __restrict__ bias, const float* __restrict__ e1a,
                                                          const float* __restrict__ e2a, const int64_t P, const int F,
                                                          const int T, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t pair = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pair >= P) return;
    const float* e1 = e1a + pair * F;
    const float* e2 = e2a + pair * F;
    for (int t = lane; t < T; t += 64) {
        float s = 0.f;
        for (int j = 0; j < F; ++j) {
            float v = 0.f;
            for (int i = 0; i < F; ++i) v = fmaf(e1[i], w[((size_t)i * F + j) * T + t], v);
            s = fmaf(v, e2[j], s);
        }
        float blk = 0.f;
        for (int q = 0; q < F; ++q) blk = fmaf(wb[(size_t)t * 2 * F + q], e1[q], blk);
        for (int q = 0; q < F; ++q) blk = fmaf(wb[(size_t)t * 2 * F + F + q], e2[q], blk);
        out[pair * T + t] = fmaxf(s + blk + bias[t], 0.f);
    }
}

int launch_ntn_any(const float* w, const float* wb, const float* bias, const float* e1, const float* e2, int64_t P, int F,
                   int T, float* out, hipStream_t stream) {
    if (P == 0) return SGPR_OK;
    const int64_t blocks = (P + 3) / 4;
    if (blocks > 0x7fffffffLL) {
        set_error("sgpr_ntn_any: too many pairs for one launch");
        return SGPR_E_INVALID;
    }
    hipLaunchKernelGGL(generic_ntn_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, w, wb, bias, e1, e2, P, F, T, out);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "generic_ntn_kernel launch");
    return SGPR_OK;
}

}  // namespace sgpr
