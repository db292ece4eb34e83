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
//  * kNN: Gram matrix X.X^T on the same MFMA path, ranking key  |x_j|^2 - 2 x_i.x_j
//    (= the reference's -pairwise_distance up to the row constant |x_i|^2), then an
//    exact k-smallest selection per row with a deterministic lowest-index tie-break.
//  * 512 threads = 8 wave64; everything between the input read and the pooled
//    vector lives in LDS / registers.
#include <math.h>

#include "sgpr_internal.hpp"

namespace sgpr {

constexpr int NT = 512;       // threads per workgroup
constexpr int NW = NT / 64;   // wave64s per workgroup
constexpr int PX = 68;        // floats per row of X   (64 ch + 4: 16-B aligned, rows shift 4 banks)
constexpr int PA = 64;        // floats per row of A   (gather target, read lane==channel)
constexpr int PE = 36;        // floats per row of E   (final node embedding, 32 ch + 4)
constexpr int PP = 32;        // floats per row of the parked xyz3 block
constexpr int kRedBytes = 2560;
constexpr int kLdsLimit = 160 * 1024;

typedef float f32x4 __attribute__((ext_vector_type(4)));

static inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

bool make_embed_plan(int N, int k, EmbedPlan* p) {
    if (N < 1 || N > SGPR_MAX_NODES || k < 1 || k > SGPR_MAX_K || k > N) return false;
    p->N = N;
    p->NP = round_up(N, 16);
    p->k = k;
    p->kmax = k <= 10 ? 10 : (k <= 16 ? 16 : (k <= 20 ? 20 : 32));
    p->kpitch = round_up(k, 4);
    p->pitchD = p->NP + 4;
    p->park_in_lds = N <= 128 ? 1 : 0;
    int off = 0;
    p->offX = off;    off += p->NP * PX * 4;
    p->offA = off;    off += p->NP * PA * 4;
    p->offPark = off; off += p->park_in_lds ? p->NP * PP * 4 : 0;
    p->offXX = off;   off += p->NP * 4;
    p->offRed = off;  off += kRedBytes;
    p->offIdx = off;  off += round_up(p->NP * p->kpitch, 16);
    p->offD = off;
    const int rowD = p->pitchD * 4;
    // two workgroups per CU when a >=64-row distance chunk still fits in 80 KB, else one
    int budget = (off + (p->NP < 64 ? p->NP : 64) * rowD <= kLdsLimit / 2) ? kLdsLimit / 2 : kLdsLimit;
    int rc = (budget - off) / rowD / 16 * 16;
    if (rc > p->NP) rc = p->NP;
    if (rc < 16) return false;
    p->RC = rc;
    int P = 1;
    while (P * 2 <= 16 && P * 2 * rc <= NT) P *= 2;
    while (P > 1 && round_up((N + P - 1) / P, 4) < 8) P /= 2;
    p->P = P;
    p->seg = round_up((N + P - 1) / P, 4);
    p->lds_bytes = off + rc * rowD;
    return true;
}

struct KParams {
    DevWeights w;
    EmbedPlan p;
    EmbedArgs a;
};

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

template <int KMAX>
__device__ __forceinline__ void insert_sorted(float (&L)[KMAX], float d) {
    // L ascending; keeps the KMAX smallest values seen (duplicates kept). Branch-free bubble.
#pragma unroll
    for (int s = 0; s < KMAX; ++s) {
        const float lo = fminf(L[s], d);
        d = fmaxf(L[s], d);
        L[s] = lo;
    }
}

// Exact k-smallest selection for the rows of one distance chunk.
// P consecutive lanes share a row; lane `part` scans candidates [part*seg, (part+1)*seg).
// Result: idx[i][0..k) = the k nearest candidates of row i under the total order
// (key ascending, index ascending) - written as an unordered set.
template <int KMAX>
__device__ __forceinline__ void select_phase(const KParams& kp, const float* __restrict__ D, int rc0, int rows_chunk,
                                             unsigned char* __restrict__ idx, int32_t* __restrict__ dbg_knn) {
    const EmbedPlan& p = kp.p;
    const int tid = threadIdx.x, lane = tid & 63;
    const int P = p.P;
    const int rl = tid / P, part = tid & (P - 1);
    const int i = rc0 + rl;
    const bool active = (rl < rows_chunk) && (i < p.N);
    const float* drow = D + (active ? rl : 0) * p.pitchD;
    const int j0 = part * p.seg;
    const int j1 = min(p.N, j0 + p.seg);
    const int k = p.k;

    float L[KMAX];
#pragma unroll
    for (int s = 0; s < KMAX; ++s) L[s] = INFINITY;
    for (int j = j0; j < j1; j += 4) {
        const float4 d4 = *reinterpret_cast<const float4*>(drow + j);
        insert_sorted<KMAX>(L, d4.x);
        insert_sorted<KMAX>(L, (j + 1 < j1) ? d4.y : INFINITY);
        insert_sorted<KMAX>(L, (j + 2 < j1) ? d4.z : INFINITY);
        insert_sorted<KMAX>(L, (j + 3 < j1) ? d4.w : INFINITY);
    }
    // merge the P partial lists of a row (butterfly; every lane ends with the row's KMAX smallest)
    for (int m = 1; m < P; m <<= 1) {
        float o[KMAX];
#pragma unroll
        for (int s = 0; s < KMAX; ++s) o[s] = __shfl_xor(L[s], m);
#pragma unroll
        for (int s = 0; s < KMAX; ++s) insert_sorted<KMAX>(L, o[s]);
    }
    float tau = L[0];
#pragma unroll
    for (int s = 1; s < KMAX; ++s) tau = (s == k - 1) ? L[s] : tau;
    int c_less = 0;
#pragma unroll
    for (int s = 0; s < KMAX; ++s) c_less += (s < k && L[s] < tau) ? 1 : 0;
    const int T = k - c_less;  // ties at tau to accept, lowest index first

    int n_less = 0, n_eq = 0;
    for (int j = j0; j < j1; ++j) {
        const float d = drow[j];
        n_less += d < tau ? 1 : 0;
        n_eq += d == tau ? 1 : 0;
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
    if (!active) return;
    int pos = e_less + min(e_eq, T);
    const int my_ties = max(0, min(n_eq, T - e_eq));
    int eqc = 0;
    unsigned char* out = idx + i * p.kpitch;
    for (int j = j0; j < j1; ++j) {
        const float d = drow[j];
        const bool eq = d == tau;
        const bool take = (d < tau) || (eq && eqc < my_ties);
        eqc += eq ? 1 : 0;
        if (take) {
            out[pos] = (unsigned char)j;
            if (dbg_knn) dbg_knn[(size_t)i * k + pos] = j;
            ++pos;
        }
    }
}

// One EdgeConv layer on the branch currently staged in X (Kp input channels, zero padded).
//   dst row i, channel c  ->  ydst[i * ypitch + c]
template <int KMAX>
__device__ __forceinline__ void edgeconv_layer(const KParams& kp, unsigned char* smem, int L, float* ydst, int ypitch) {
    const EmbedPlan& p = kp.p;
    float* X = reinterpret_cast<float*>(smem + p.offX);
    float* A = reinterpret_cast<float*>(smem + p.offA);
    float* D = reinterpret_cast<float*>(smem + p.offD);
    float* xx = reinterpret_cast<float*>(smem + p.offXX);
    unsigned char* idx = smem + p.offIdx;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lq = lane >> 4;
    const int N = p.N, NP = p.NP, k = p.k;
    const int Kp = kp.w.kp[L], cout = kp.w.cout[L];
    const int nkb = Kp >> 4;
    const int g = blockIdx.x;
    int32_t* dbg_knn = kp.a.dbg_knn ? kp.a.dbg_knn + ((size_t)g * 6 + L) * N * k : nullptr;

    // ---- squared norms
    for (int i = tid; i < NP; i += NT) {
        const float* xr = X + i * PX;
        float s = 0.f;
        for (int c = 0; c < Kp; c += 4) {
            const float4 v = *reinterpret_cast<const float4*>(xr + c);
            s = fmaf(v.x, v.x, s);
            s = fmaf(v.y, v.y, s);
            s = fmaf(v.z, v.z, s);
            s = fmaf(v.w, v.w, s);
        }
        xx[i] = s;
    }
    __syncthreads();

    // ---- kNN, one chunk of rows at a time: Gram tile on MFMA -> ranking keys in LDS -> selection
    const int ntj = NP >> 4;
    for (int rc0 = 0; rc0 < NP; rc0 += p.RC) {
        const int rows_chunk = min(p.RC, NP - rc0);
        const int nti = rows_chunk >> 4;
        for (int tile = wave; tile < nti * ntj; tile += NW) {
            const int ti = tile / ntj, tj = tile - ti * ntj;
            const float* pa = X + (rc0 + ti * 16 + l15) * PX + 4 * lq;
            const float* pb = X + (tj * 16 + l15) * PX + 4 * lq;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            for (int kb = 0; kb < nkb; ++kb)
                acc = mfma_kblock(*reinterpret_cast<const float4*>(pa + kb * 16),
                                  *reinterpret_cast<const float4*>(pb + kb * 16), acc);
            const int j = tj * 16 + l15;
            const float xj = xx[j];
            const bool valid = j < N;
            float* drow = D + (ti * 16 + 4 * lq) * p.pitchD + j;
#pragma unroll
            for (int r = 0; r < 4; ++r) drow[r * p.pitchD] = valid ? fmaf(-2.f, acc[r], xj) : INFINITY;
        }
        __syncthreads();
        select_phase<KMAX>(kp, D, rc0, rows_chunk, idx, dbg_knn);
        __syncthreads();
    }

    // ---- per-node GEMMs on MFMA:  a = W1'.x -> A (LDS),  b = (W2-W1)'.x + t -> registers
    const float* __restrict__ Wf = kp.w.wf[L];
    const float* __restrict__ tb = kp.w.tb[L];
    const int nct = (2 * cout) >> 4;   // 16-channel column tiles of [a | b]
    const int RS = NW / nct;           // row-tile stride between waves sharing a column tile
    const int ct = wave % nct, rs = wave / nct;
    const int nrt = NP >> 4;
    const bool is_b = ct * 16 >= cout;
    float4 wreg[4];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
        wreg[kb] = (kb < nkb) ? *reinterpret_cast<const float4*>(Wf + (size_t)(ct * 16 + l15) * Kp + kb * 16 + 4 * lq)
                              : make_float4(0.f, 0.f, 0.f, 0.f);
    f32x4 breg[16];
    const int c4 = ct * 16 + 4 * lq;  // first of this lane's 4 output channels in [a | b]
    float4 tb4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (is_b) tb4 = *reinterpret_cast<const float4*>(tb + (c4 - cout));
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int rt = rs + q * RS;
        if (rt < nrt) {
            const float* px = X + (rt * 16 + l15) * PX + 4 * lq;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
                if (kb < nkb) acc = mfma_kblock(wreg[kb], *reinterpret_cast<const float4*>(px + kb * 16), acc);
            // acc[r] = out[channel c4 + r][node rt*16 + l15]
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
    __syncthreads();  // every wave is done reading X: b may now overwrite it in place
    if (is_b) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int rt = rs + q * RS;
            if (rt < nrt)
                *reinterpret_cast<float4*>(X + (rt * 16 + l15) * PX + (c4 - cout)) =
                    make_float4(breg[q][0], breg[q][1], breg[q][2], breg[q][3]);
        }
    }
    __syncthreads();

    // ---- gather-max over the k neighbours (lane == channel: conflict-free LDS rows)
    const int rpw = 64 / cout;           // rows per wave-iteration (1 or 2)
    const int c = lane & (cout - 1), sub = lane / cout;
    float* dbg = kp.a.dbg_layers ? kp.a.dbg_layers + ((size_t)g * 6 + L) * N * 64 : nullptr;
    for (int i = wave * rpw + sub; i < NP; i += NW * rpw) {
        float y = 0.f;
        if (i < N) {
            const unsigned char* id = idx + i * p.kpitch;
            float m = -INFINITY;
            for (int mm = 0; mm < k; ++mm) m = fmaxf(m, A[(int)id[mm] * PA + c]);
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
}

template <int KMAX>
__global__ __launch_bounds__(NT) void embed_kernel(const KParams kp) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const EmbedPlan& p = kp.p;
    float* X = reinterpret_cast<float*>(smem + p.offX);
    float* A = reinterpret_cast<float*>(smem + p.offA);
    float* xx = reinterpret_cast<float*>(smem + p.offXX);
    float* red = reinterpret_cast<float*>(smem + p.offRed);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lq = lane >> 4;
    const int g = blockIdx.x;
    const int N = p.N, NP = p.NP;
    float* park = p.park_in_lds ? reinterpret_cast<float*>(smem + p.offPark) : kp.a.park_ws + (size_t)g * NP * PP;

    for (int br = 0; br < 2; ++br) {
        // ---- stage this branch's input features, zero padded to 16 channels / NP rows
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
        edgeconv_layer<KMAX>(kp, smem, br * 3 + 0, X, PX);
        edgeconv_layer<KMAX>(kp, smem, br * 3 + 1, X, PX);
        if (br == 0)
            edgeconv_layer<KMAX>(kp, smem, 2, park, PP);          // xyz3 parked while the sem branch runs
        else
            edgeconv_layer<KMAX>(kp, smem, 5, X + 32, PX);        // sem3 -> channels 32..63 (b' sits in 0..31)
    }
    if (!p.park_in_lds) __syncthreads();
    for (int e = tid; e < NP * 32; e += NT) {                     // xyz3 -> channels 0..31: X = cat(xyz3, sem3)
        const int i = e >> 5, c = e & 31;
        X[i * PX + c] = park[(size_t)i * PP + c];
    }
    __syncthreads();

    // ---- conv_end: 64 -> 32, folded BN, LeakyReLU  -> E (in the A region)
    float* E = A;
    {
        const float* __restrict__ Wf = kp.w.wf_end;
        const int nrt = NP >> 4;
        for (int task = wave; task < 2 * nrt; task += NW) {
            const int ct = task & 1, rt = task >> 1;
            const float* pw = Wf + (size_t)(ct * 16 + l15) * 64 + 4 * lq;
            const float* px = X + (rt * 16 + l15) * PX + 4 * lq;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
                acc = mfma_kblock(*reinterpret_cast<const float4*>(pw + kb * 16),
                                  *reinterpret_cast<const float4*>(px + kb * 16), acc);
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

    // ---- attention pooling over all N slots (padding is NOT masked, divisor N: layers_batch.py:34-38)
    float* mean = red + 16 * 32;
    float* tg = mean + 32;
    float* sig = xx;
    const int c = tid & 31, prt = tid >> 5;  // 16 partial sums per channel
    {
        float s = 0.f;
        for (int n = prt; n < N; n += NT / 32) s += E[n * PE + c];
        red[prt * 32 + c] = s;
    }
    __syncthreads();
    if (tid < 32) {
        float s = 0.f;
        for (int q = 0; q < NT / 32; ++q) s += red[q * 32 + tid];
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
        float d = 0.f;
        for (int q = 0; q < 32; ++q) d = fmaf(E[n * PE + q], tg[q], d);
        const float sg = 1.f / (1.f + expf(-d));
        sig[n] = sg;
        if (kp.a.att) kp.a.att[(size_t)g * N + n] = sg;
    }
    __syncthreads();
    {
        float s = 0.f;
        for (int n = prt; n < N; n += NT / 32) s = fmaf(sig[n], E[n * PE + c], s);
        red[prt * 32 + c] = s;
    }
    __syncthreads();
    if (tid < 32) {
        float s = 0.f;
        for (int q = 0; q < NT / 32; ++q) s += red[q * 32 + tid];
        kp.a.pooled[(size_t)g * 32 + tid] = s;
    }
}

template <int KMAX>
static int launch_t(const KParams& kp, hipStream_t stream) {
    static bool attr_set = false;  // benign race: idempotent
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&embed_kernel<KMAX>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, kLdsLimit);
        if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(embed_kernel)");
        attr_set = true;
    }
    hipLaunchKernelGGL(embed_kernel<KMAX>, dim3(kp.a.G), dim3(NT), kp.p.lds_bytes, stream, kp);
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
    switch (plan.kmax) {
        case 10: return launch_t<10>(kp, stream);
        case 16: return launch_t<16>(kp, stream);
        case 20: return launch_t<20>(kp, stream);
        default: return launch_t<32>(kp, stream);
    }
}

}  // namespace sgpr
