// Fused per-graph embed kernel for gfx950 (MI355X): one workgroup stages one whole
// semantic graph in LDS and runs
//     2 branches x 3 x { kNN -> EdgeConv (1x1 conv + eval-BN + LeakyReLU + max_k) }
//     -> conv_end -> attention pooling
// i.e. SG.dgcnn_conv_pass (reference sg_net.py:79-110, dgcnn.py:14-49) followed by
// AttentionModule.forward (layers_batch.py:28-39).
//
// MI355X-first formulation:
//  * the [B,2C,N,k] edge tensor of dgcnn.get_graph_feature is never built:
//      W.[x_j - x_i ; x_i] = W1.x_j + (W2-W1).x_i,  BN(eval) folded into W, and since
//      LeakyReLU is monotone   max_m lrelu(a_j + b_i) = lrelu(max_m a_j + b_i)
//    => two per-node GEMMs (a = W1'x, b = (W2-W1)'x + t) on the fp32 matrix cores
//       (v_mfma_f32_16x16x4_f32, exact fp32) and a gather-max over the k neighbours.
//  * kNN: Gram matrix X.X^T on the same MFMA path (upper triangle only when the
//    whole key matrix fits in LDS), ranking key |x_j|^2 - 2 x_i.x_j (= the reference's
//    -pairwise_distance up to the row constant |x_i|^2) stored as order-preserving
//    int32, then an exact k-smallest selection per row: register sorting networks +
//    butterfly merges, deterministic lowest-index tie-break.
//  * 512 threads x 2 workgroups/CU (small graphs) or 1024 threads x 1 (LDS-bound);
//    everything between the input read and the pooled vector lives in LDS/registers.
#include <limits.h>
#include <math.h>

#include "sgpr_internal.hpp"

namespace sgpr {

constexpr int PX = 68;        // floats per row of X   (64 ch + 4: 16-B aligned, rows shift one 16-B slot)
constexpr int PA = 68;        // floats per row of A   (gather target)
constexpr int PE = 36;        // floats per row of E   (final node embedding, 32 ch + 4)
constexpr int PP = 32;        // floats per row of the parked xyz3 block
constexpr int CAP = 16;       // candidates per lane in the selection phase
constexpr int MAXQ = 8;       // row tiles per wave in the GEMM phase
constexpr int kRedBytes = 4608;
constexpr int kLdsLimit = 160 * 1024;

typedef float f32x4 __attribute__((ext_vector_type(4)));

static inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

bool make_embed_plan(int N, int k, EmbedPlan* p) {
    if (N < 1 || N > SGPR_MAX_NODES || k < 1 || k > SGPR_MAX_K || k > N) return false;
    p->N = N;
    p->NP = round_up(N, 16);
    p->k = k;
    p->kp = k <= 16 ? 16 : 32;
    p->kpitch = round_up(k, 4);
    p->pitchD = p->NP + 4;
    p->park_in_lds = N <= 128 ? 1 : 0;
    int off = 0;
    p->offX = off;    off += p->NP * PX * 4;
    p->offA = off;    off += p->NP * PA * 4;
    p->offPark = off; off += p->park_in_lds ? p->NP * PP * 4 : 0;
    p->offXX = off;   off += p->NP * 4;
    p->offIdx = off;  off += round_up(p->NP * p->kpitch, 16);
    p->offD = off;
    p->offRed = off;  // attention scratch aliases the key chunk (disjoint in time)
    const int rowD = p->pitchD * 4;
    // two 512-thread workgroups per CU when a >=64-row key chunk still fits in 80 KB,
    // else one 1024-thread workgroup owning the whole LDS
    const bool small = off + (p->NP < 64 ? p->NP : 64) * rowD <= kLdsLimit / 2;
    p->nt = small ? 512 : 1024;
    const int budget = small ? kLdsLimit / 2 : kLdsLimit;
    int rc = (budget - off) / rowD / 16 * 16;
    if (rc > p->NP) rc = p->NP;
    int P = 1;
    while ((N + P - 1) / P > CAP) P *= 2;
    if (rc * P > p->nt) rc = p->nt / P / 16 * 16;
    if (rc < 16) return false;
    p->RC = rc;
    p->P = P;
    p->seg = round_up((N + P - 1) / P, 4);
    p->lds_bytes = off + (rc * rowD > kRedBytes ? rc * rowD : kRedBytes);
    return true;
}

struct KParams {
    DevWeights w;
    EmbedPlan p;
    EmbedArgs a;
};

// ------------------------------------------------------------------ MFMA helpers
__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
    // D[4*(l>>4)+r][l&15] += sum_{q<4} A[row][q] * B[q][col];  lane l supplies A[l&15][l>>4], B[l>>4][l&15]
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// One 16-wide k-block: lane group q = l>>4 holds k = 4q..4q+3 of both operands (one
// 16-B read each); MFMA step s contracts k = 4q+s over the four lane groups.  The
// order of the k-sum is permuted, which a dot product does not care about.
__device__ __forceinline__ f32x4 mfma_kblock(const float4 a, const float4 b, f32x4 acc) {
    acc = mfma4(a.x, b.x, acc);
    acc = mfma4(a.y, b.y, acc);
    acc = mfma4(a.z, b.z, acc);
    acc = mfma4(a.w, b.w, acc);
    return acc;
}

// 16x16 output tile, K = 16*NKB: operands preloaded, two accumulators break the MFMA dependency chain
template <int NKB>
__device__ __forceinline__ f32x4 tile16(const float4 (&a)[4], const float4 (&b)[4]) {
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f};
    if (NKB == 1) return mfma_kblock(a[0], b[0], acc0);
    f32x4 acc1 = {0.f, 0.f, 0.f, 0.f};
    acc0 = mfma_kblock(a[0], b[0], acc0);
    acc1 = mfma_kblock(a[1], b[1], acc1);
    acc0 = mfma_kblock(a[2], b[2], acc0);
    acc1 = mfma_kblock(a[3], b[3], acc1);
    return acc0 + acc1;
}

template <int NKB>
__device__ __forceinline__ void load_frag(const float* p, float4 (&f)[4]) {
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) f[kb] = *reinterpret_cast<const float4*>(p + kb * 16);
}

// ------------------------------------------------------------------ selection helpers
// order-preserving float -> int32 (so that ranking uses v_min_i32 / v_max_i32, exact ties)
__device__ __forceinline__ int ord_key(float f) {
    f += 0.0f;  // -0 -> +0
    const int b = __float_as_int(f);
    return b ^ ((b >> 31) & 0x7fffffff);
}

__device__ __forceinline__ void cswap(int& a, int& b) {
    const int lo = min(a, b);
    b = max(a, b);
    a = lo;
}

template <int N>
__device__ __forceinline__ void bitonic_merge(int (&v)[N]) {  // bitonic in -> ascending out
#pragma unroll
    for (int j = N / 2; j > 0; j >>= 1)
#pragma unroll
        for (int i = 0; i < N; ++i)
            if ((i & j) == 0) cswap(v[i], v[i | j]);
}

template <int N>
__device__ __forceinline__ void bitonic_sort(int (&v)[N]) {  // any -> ascending
#pragma unroll
    for (int k = 2; k <= N; k <<= 1)
#pragma unroll
        for (int j = k / 2; j > 0; j >>= 1)
#pragma unroll
            for (int i = 0; i < N; ++i)
                if ((i & j) == 0) {
                    if ((i & k) == 0)
                        cswap(v[i], v[i | j]);
                    else
                        cswap(v[i | j], v[i]);
                }
}

// Exact k-smallest selection for the rows of one key chunk.
// P consecutive lanes share a row; lane `part` owns candidates [part*seg, (part+1)*seg).
// Result: idx[i][0..k) = the k nearest candidates of row i under the total order
// (key ascending, index ascending) - written as an unordered set.
template <int KP>
__device__ __forceinline__ void select_phase(const EmbedPlan& p, const int* __restrict__ D, int rc0, int rows_chunk,
                                             unsigned char* __restrict__ idx, int32_t* __restrict__ dbg_knn) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int P = p.P;
    const int rl = tid >> (__ffs(P) - 1), part = tid & (P - 1);
    const int i = rc0 + rl;
    const bool active = (rl < rows_chunk) && (i < p.N);
    const int* drow = D + (active ? rl : 0) * p.pitchD;
    const int j0 = part * p.seg;
    const int k = p.k;

    int d[CAP];
#pragma unroll
    for (int q = 0; q < CAP / 4; ++q) {
        int4 v = make_int4(INT_MAX, INT_MAX, INT_MAX, INT_MAX);
        if (4 * q < p.seg && j0 + 4 * q < p.NP) v = *reinterpret_cast<const int4*>(drow + j0 + 4 * q);
        d[4 * q + 0] = (j0 + 4 * q + 0 < p.N) ? v.x : INT_MAX;
        d[4 * q + 1] = (j0 + 4 * q + 1 < p.N) ? v.y : INT_MAX;
        d[4 * q + 2] = (j0 + 4 * q + 2 < p.N) ? v.z : INT_MAX;
        d[4 * q + 3] = (j0 + 4 * q + 3 < p.N) ? v.w : INT_MAX;
    }
    int L[KP];
    {
        int s[CAP];
#pragma unroll
        for (int u = 0; u < CAP; ++u) s[u] = d[u];
        if (p.seg <= 8) {
            int h[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) h[u] = s[u];
            bitonic_sort<8>(h);
#pragma unroll
            for (int u = 0; u < 8; ++u) s[u] = h[u];
        } else {
            bitonic_sort<CAP>(s);
        }
#pragma unroll
        for (int u = 0; u < KP; ++u) L[u] = u < CAP ? s[u < CAP ? u : 0] : INT_MAX;
    }
    // butterfly merge of the P sorted lists of a row: min(L[s], O[KP-1-s]) is the KP smallest of the union, bitonic
    for (int m = 1; m < P; m <<= 1) {
        int o[KP];
#pragma unroll
        for (int s = 0; s < KP; ++s) o[s] = __shfl_xor(L[s], m);
#pragma unroll
        for (int s = 0; s < KP; ++s) L[s] = min(L[s], o[KP - 1 - s]);
        bitonic_merge<KP>(L);
    }
    int tau = L[0];
#pragma unroll
    for (int s = 1; s < KP; ++s) tau = (s == k - 1) ? L[s] : tau;
    int c_less = 0;
#pragma unroll
    for (int s = 0; s < KP; ++s) c_less += (s < k && L[s] < tau) ? 1 : 0;
    const int T = k - c_less;  // ties at tau to accept, lowest index first

    int n_less = 0, n_eq = 0;
#pragma unroll
    for (int u = 0; u < CAP; ++u) {
        n_less += d[u] < tau ? 1 : 0;
        n_eq += d[u] == tau ? 1 : 0;
    }
    int e_less = 0, e_eq = 0;
    const int base = lane & ~(P - 1);
    for (int q = 0; q < P; ++q) {
        const int vl = __shfl(n_less, base + q);
        const int ve = __shfl(n_eq, base + q);
        if (q < part) {
            e_less += vl;
            e_eq += ve;
        }
    }
    if (active) {
        int pos = e_less + min(e_eq, T);
        const int my_ties = max(0, min(n_eq, T - e_eq));
        int eqc = 0;
        unsigned char* out = idx + i * p.kpitch;
#pragma unroll
        for (int u = 0; u < CAP; ++u) {
            const bool eq = d[u] == tau;
            const bool take = (d[u] < tau) || (eq && eqc < my_ties);
            eqc += eq ? 1 : 0;
            if (take) {
                out[pos] = (unsigned char)(j0 + u);
                if (dbg_knn) dbg_knn[(size_t)i * k + pos] = j0 + u;
                ++pos;
            }
        }
    }
}

// ------------------------------------------------------------------ Gram tile -> ranking keys
template <int NKB>
__device__ __forceinline__ void gram_tile(const float* __restrict__ X, const float* __restrict__ xx, int* __restrict__ D,
                                          int pitchD, int N, int rc0, int ti, int tj, bool mirror, int l15, int lq) {
    // ti indexes 16-row tiles inside the chunk starting at row rc0; tj indexes candidate tiles
    const int i0 = rc0 + ti * 16, j0 = tj * 16;
    float4 a[4], b[4];
    load_frag<NKB>(X + (i0 + l15) * PX + 4 * lq, a);
    load_frag<NKB>(X + (j0 + l15) * PX + 4 * lq, b);
    const f32x4 g = tile16<NKB>(a, b);          // g[r] = <x_{i0+4lq+r}, x_{j0+l15}>
    const int j = j0 + l15;
    const float xj = xx[j];
    const bool jvalid = j < N;
    int* drow = D + (ti * 16 + 4 * lq) * pitchD + j;
#pragma unroll
    for (int r = 0; r < 4; ++r) drow[r * pitchD] = jvalid ? ord_key(fmaf(-2.f, g[r], xj)) : INT_MAX;
    if (mirror) {  // the transposed tile: row j, candidates i0+4lq..+3 (one 16-B store)
        const float4 xi = *reinterpret_cast<const float4*>(xx + i0 + 4 * lq);
        const int ib = i0 + 4 * lq;
        int4 kv;
        kv.x = (ib + 0 < N) ? ord_key(fmaf(-2.f, g[0], xi.x)) : INT_MAX;
        kv.y = (ib + 1 < N) ? ord_key(fmaf(-2.f, g[1], xi.y)) : INT_MAX;
        kv.z = (ib + 2 < N) ? ord_key(fmaf(-2.f, g[2], xi.z)) : INT_MAX;
        kv.w = (ib + 3 < N) ? ord_key(fmaf(-2.f, g[3], xi.w)) : INT_MAX;
        *reinterpret_cast<int4*>(D + (size_t)j * pitchD + ib) = kv;
    }
}

// ------------------------------------------------------------------ GEMM row-tile loop
template <int NKB, int NT>
__device__ __forceinline__ void gemm_phase(const float* __restrict__ X, float* __restrict__ A,
                                           const float* __restrict__ Wf, const float* __restrict__ tb, int cout,
                                           int nrt, f32x4 (&breg)[MAXQ], bool& is_b_out, int& c4_out, int& rs_out,
                                           int& RS_out) {
    constexpr int NW = NT / 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lq = lane >> 4;
    const int nct = (2 * cout) >> 4;   // 16-channel column tiles of [a | b]: 8 or 4
    const int RS = NW / nct;           // waves sharing a column tile, striding over row tiles
    const int ct = wave % nct, rs = wave / nct;
    const bool is_b = ct * 16 >= cout;
    constexpr int Kp = NKB * 16;
    float4 wreg[4];
    load_frag<NKB>(Wf + (size_t)(ct * 16 + l15) * Kp + 4 * lq, wreg);
    const int c4 = ct * 16 + 4 * lq;   // first of this lane's 4 output channels in [a | b]
    float4 tb4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (is_b) tb4 = *reinterpret_cast<const float4*>(tb + (c4 - cout));
#pragma unroll
    for (int q = 0; q < MAXQ; ++q) {
        const int rt = rs + q * RS;
        if (rt < nrt) {
            float4 xf[4];
            load_frag<NKB>(X + (rt * 16 + l15) * PX + 4 * lq, xf);
            const f32x4 acc = tile16<NKB>(wreg, xf);   // acc[r] = out[channel c4 + r][node rt*16 + l15]
            if (!is_b) {
                *reinterpret_cast<float4*>(A + (rt * 16 + l15) * PA + c4) = make_float4(acc[0], acc[1], acc[2], acc[3]);
            } else {
                breg[q][0] = acc[0] + tb4.x;
                breg[q][1] = acc[1] + tb4.y;
                breg[q][2] = acc[2] + tb4.z;
                breg[q][3] = acc[3] + tb4.w;
            }
        }
    }
    is_b_out = is_b;
    c4_out = c4;
    rs_out = rs;
    RS_out = RS;
}

template <int KP, int NT>
__global__ __launch_bounds__(NT, NT == 512 ? 4 : 4) void embed_kernel(const KParams kp) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NW = NT / 64;
    const EmbedPlan& p = kp.p;
    float* X = reinterpret_cast<float*>(smem + p.offX);
    float* A = reinterpret_cast<float*>(smem + p.offA);
    int* D = reinterpret_cast<int*>(smem + p.offD);
    float* xx = reinterpret_cast<float*>(smem + p.offXX);
    float* red = reinterpret_cast<float*>(smem + p.offRed);
    unsigned char* idx = smem + p.offIdx;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lq = lane >> 4;
    const int g = blockIdx.x;
    const int N = p.N, NP = p.NP, k = p.k;
    const int nrt = NP >> 4;
    float* park = p.park_in_lds ? reinterpret_cast<float*>(smem + p.offPark) : kp.a.park_ws + (size_t)g * NP * PP;
    // optional per-phase cycle accounting (thread 0 of every workgroup; phases end at barriers)
    unsigned long long t_prev = 0;
    const bool prof = kp.a.prof != nullptr && tid == 0;
    if (prof) t_prev = clock64();
#define SGPR_PROF(ph)                                            \
    if (prof) {                                                  \
        const unsigned long long t_now = clock64();              \
        atomicAdd(&kp.a.prof[ph], t_now - t_prev);               \
        t_prev = t_now;                                          \
    }

    for (int L = 0; L < 6; ++L) {
        if (L == 0 || L == 3) {
            // ---- stage this branch's input features, zero padded to 16 channels / NP rows
            const int br = L == 0 ? 0 : 1;
            for (int e = tid; e < NP * 16; e += NT) {
                const int i = e >> 4, c = e & 15;
                float v = 0.f;
                if (i < N) {
                    if (kp.a.dense) {
                        const bool second = kp.a.dense2 && g >= kp.a.g_split;
                        const float* dn = second ? kp.a.dense2 : kp.a.dense;
                        const int gg = second ? g - kp.a.g_split : g;
                        const int ch = br == 0 ? c : 3 + c;
                        const bool ok = br == 0 ? c < 3 : c < kLabels;
                        if (ok) v = dn[((size_t)gg * (3 + kLabels) + ch) * N + i];
                    } else if (br == 0) {
                        if (c < 3) v = kp.a.centers[((size_t)g * N + i) * 3 + c];
                    } else {
                        const int lab = kp.a.labels[(size_t)g * N + i];
                        if (c == 0 && (lab < -1 || lab >= kLabels)) atomicOr(kp.a.status, 1);
                        v = (lab == c) ? 1.f : 0.f;
                    }
                }
                X[i * PX + c] = v;
            }
            __syncthreads();
            SGPR_PROF(0)
        }
        const int Kp = kp.w.kp[L], cout = kp.w.cout[L];
        const bool k64 = Kp == 64;
        int32_t* dbg_knn = kp.a.dbg_knn ? kp.a.dbg_knn + ((size_t)g * 6 + L) * N * k : nullptr;

        // ---- squared norms: 4 lanes per row
        for (int e = tid; e < NP * 4; e += NT) {
            const int i = e >> 2, qq = e & 3;
            const float* xr = X + i * PX + qq * (Kp >> 2);
            float s = 0.f;
            for (int c = 0; c < (Kp >> 2); c += 4) {
                const float4 v = *reinterpret_cast<const float4*>(xr + c);
                s = fmaf(v.x, v.x, s);
                s = fmaf(v.y, v.y, s);
                s = fmaf(v.z, v.z, s);
                s = fmaf(v.w, v.w, s);
            }
            s += __shfl_xor(s, 1);
            s += __shfl_xor(s, 2);
            if (qq == 0) xx[i] = s;
        }
        __syncthreads();
        SGPR_PROF(1)

        // ---- kNN: Gram tiles on MFMA -> ranking keys in LDS -> selection
        if (p.RC == NP) {
            // whole key matrix resident: upper-triangular tiles only, each also stores its transpose
            int cnt = 0;
            for (int ti = 0; ti < nrt; ++ti)
                for (int tj = ti; tj < nrt; ++tj, ++cnt)
                    if (cnt % NW == wave) {
                        if (k64)
                            gram_tile<4>(X, xx, D, p.pitchD, N, 0, ti, tj, tj != ti, l15, lq);
                        else
                            gram_tile<1>(X, xx, D, p.pitchD, N, 0, ti, tj, tj != ti, l15, lq);
                    }
            __syncthreads();
            SGPR_PROF(2)
            select_phase<KP>(p, D, 0, NP, idx, dbg_knn);
            if (kp.a.prof) __syncthreads();
            SGPR_PROF(3)
        } else {
            for (int rc0 = 0; rc0 < NP; rc0 += p.RC) {
                const int rows_chunk = min(p.RC, NP - rc0);
                const int nti = rows_chunk >> 4;
                for (int tile = wave; tile < nti * nrt; tile += NW) {
                    const int ti = tile / nrt, tj = tile - ti * nrt;
                    if (k64)
                        gram_tile<4>(X, xx, D, p.pitchD, N, rc0, ti, tj, false, l15, lq);
                    else
                        gram_tile<1>(X, xx, D, p.pitchD, N, rc0, ti, tj, false, l15, lq);
                }
                __syncthreads();
                SGPR_PROF(2)
                select_phase<KP>(p, D, rc0, rows_chunk, idx, dbg_knn);
                __syncthreads();
                SGPR_PROF(3)
            }
        }

        // ---- per-node GEMMs on MFMA:  a = W1'.x -> A (LDS),  b = (W2-W1)'.x + t -> registers
        f32x4 breg[MAXQ];
        bool is_b;
        int c4, rs, RS;
        if (k64)
            gemm_phase<4, NT>(X, A, kp.w.wf[L], kp.w.tb[L], cout, nrt, breg, is_b, c4, rs, RS);
        else
            gemm_phase<1, NT>(X, A, kp.w.wf[L], kp.w.tb[L], cout, nrt, breg, is_b, c4, rs, RS);
        __syncthreads();  // every wave is done reading X (and D): b may now overwrite X in place
        if (is_b) {
#pragma unroll
            for (int q = 0; q < MAXQ; ++q) {
                const int rt = rs + q * RS;
                if (rt < nrt)
                    *reinterpret_cast<float4*>(X + (rt * 16 + l15) * PX + (c4 - cout)) =
                        make_float4(breg[q][0], breg[q][1], breg[q][2], breg[q][3]);
            }
        }
        __syncthreads();
        SGPR_PROF(4)

        // ---- gather-max over the k neighbours (lane == channel: conflict-free LDS rows)
        float* ydst = L == 2 ? park : (L == 5 ? X + 32 : X);
        const int ypitch = L == 2 ? PP : PX;
        const int rpw = 64 / cout;           // rows per wave-iteration (1 or 2)
        const int c = lane & (cout - 1), sub = lane / cout;
        float* dbg = kp.a.dbg_layers ? kp.a.dbg_layers + ((size_t)g * 6 + L) * N * 64 : nullptr;
        for (int i = wave * rpw + sub; i < NP; i += NW * rpw) {
            float y = 0.f;
            if (i < N) {
                const uint32_t* idw = reinterpret_cast<const uint32_t*>(idx + i * p.kpitch);
                float m = -INFINITY;
#pragma unroll
                for (int w = 0; w < KP / 4; ++w) {
                    if (4 * w < k) {
                        const uint32_t word = idw[w];
                        const int ja = word & 255u;
                        const int jb = (4 * w + 1 < k) ? (word >> 8) & 255u : ja;
                        const int jc = (4 * w + 2 < k) ? (word >> 16) & 255u : ja;
                        const int jd = (4 * w + 3 < k) ? (word >> 24) : ja;
                        const float va = A[ja * PA + c], vb = A[jb * PA + c], vc = A[jc * PA + c], vd = A[jd * PA + c];
                        m = fmaxf(m, fmaxf(fmaxf(va, vb), fmaxf(vc, vd)));
                    }
                }
                y = m + X[i * PX + c];
                y = y > 0.f ? y : 0.2f * y;
                if (dbg) {
                    dbg[(size_t)i * 64 + c] = y;
                    if (cout == 32) dbg[(size_t)i * 64 + 32 + c] = 0.f;
                }
            }
            ydst[(size_t)i * ypitch + c] = y;
        }
        __syncthreads();
        SGPR_PROF(5)
    }

    for (int e = tid; e < NP * 32; e += NT) {                     // xyz3 -> channels 0..31: X = cat(xyz3, sem3)
        const int i = e >> 5, c = e & 31;
        X[i * PX + c] = park[(size_t)i * PP + c];
    }
    __syncthreads();

    // ---- conv_end: 64 -> 32, folded BN, LeakyReLU  -> E (in the A region)
    float* E = A;
    {
        const float* __restrict__ Wf = kp.w.wf_end;
        for (int task = wave; task < 2 * nrt; task += NW) {
            const int ct = task & 1, rt = task >> 1;
            float4 wf[4], xf[4];
            load_frag<4>(Wf + (size_t)(ct * 16 + l15) * 64 + 4 * lq, wf);
            load_frag<4>(X + (rt * 16 + l15) * PX + 4 * lq, xf);
            const f32x4 acc = tile16<4>(wf, xf);
            const int c4 = ct * 16 + 4 * lq;
            const float4 t4 = *reinterpret_cast<const float4*>(kp.w.tb_end + c4);
            float4 e4 = make_float4(acc[0] + t4.x, acc[1] + t4.y, acc[2] + t4.z, acc[3] + t4.w);
            e4.x = e4.x > 0.f ? e4.x : 0.2f * e4.x;
            e4.y = e4.y > 0.f ? e4.y : 0.2f * e4.y;
            e4.z = e4.z > 0.f ? e4.z : 0.2f * e4.z;
            e4.w = e4.w > 0.f ? e4.w : 0.2f * e4.w;
            const int node = rt * 16 + l15;
            *reinterpret_cast<float4*>(E + node * PE + c4) = e4;
            if (kp.a.emb && node < N) *reinterpret_cast<float4*>(kp.a.emb + ((size_t)g * N + node) * 32 + c4) = e4;
        }
    }
    __syncthreads();
    SGPR_PROF(6)

    // ---- attention pooling over all N slots (padding is NOT masked, divisor N: layers_batch.py:34-38)
    constexpr int NPART = NT / 32;   // partial sums per channel
    float* mean = red + NPART * 32;
    float* tg = mean + 32;
    float* sig = xx;
    const int c = tid & 31, prt = tid >> 5;
    {
        float s = 0.f;
        for (int n = prt; n < N; n += NPART) s += E[n * PE + c];
        red[prt * 32 + c] = s;
    }
    __syncthreads();
    if (tid < 32) {
        float s = 0.f;
        for (int q = 0; q < NPART; ++q) s += red[q * 32 + tid];
        mean[tid] = s / (float)N;
    }
    __syncthreads();
    if (tid < 32) {
        float gc = 0.f;
        for (int r = 0; r < 32; ++r) gc = fmaf(mean[r], kp.w.att_w[r * 32 + tid], gc);
        tg[tid] = tanhf(gc);
    }
    __syncthreads();
    for (int n = tid; n < N; n += NT) {
        float dsum = 0.f;
        for (int q = 0; q < 32; ++q) dsum = fmaf(E[n * PE + q], tg[q], dsum);
        const float sg = 1.f / (1.f + expf(-dsum));
        sig[n] = sg;
        if (kp.a.att) kp.a.att[(size_t)g * N + n] = sg;
    }
    __syncthreads();
    {
        float s = 0.f;
        for (int n = prt; n < N; n += NPART) s = fmaf(sig[n], E[n * PE + c], s);
        red[prt * 32 + c] = s;
    }
    __syncthreads();
    if (tid < 32) {
        float s = 0.f;
        for (int q = 0; q < NPART; ++q) s += red[q * 32 + tid];
        kp.a.pooled[(size_t)g * 32 + tid] = s;
    }
    SGPR_PROF(7)
#undef SGPR_PROF
}

template <int KP, int NT>
static int launch_t(const KParams& kp, hipStream_t stream) {
    static bool attr_set = false;  // benign race: idempotent
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&embed_kernel<KP, NT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, kLdsLimit);
        if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(embed_kernel)");
        attr_set = true;
    }
    hipLaunchKernelGGL((embed_kernel<KP, NT>), dim3(kp.a.G), dim3(NT), kp.p.lds_bytes, stream, kp);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "embed_kernel launch");
    return SGPR_OK;
}

int launch_embed(const sgpr_handle* h, const EmbedPlan& plan, const EmbedArgs& a, hipStream_t stream) {
    if (a.G == 0) return SGPR_OK;
    KParams kp;
    kp.w = h->w;
    kp.p = plan;
    kp.a = a;
    if (plan.nt == 512) return plan.kp == 16 ? launch_t<16, 512>(kp, stream) : launch_t<32, 512>(kp, stream);
    return plan.kp == 16 ? launch_t<16, 1024>(kp, stream) : launch_t<32, 1024>(kp, stream);
}

}  // namespace sgpr
