// Consumers of the all-pairs score matrix that keep it on the device:
//
//  * sgpr_pair_positives + sgpr_pair_threshold_counts - the counting half of eval_batch.py:48-49, 69-87 (sklearn
//        roc_curve / auc, precision_recall_curve + F1 max), ground truth taken from the KITTI poses on the fly
//        (utils.py:36, sg_net.py:302-309) or from explicit labels.  F1 can only peak at the score of a positive pair,
//        and positives are rare (loop closures), so: (1) collect the scores of the positive pairs (no pass over the
//        matrix: only the poses decide which entries are read), (2) the host sorts them and picks up to 8191 of their
//        distinct values as thresholds, (3) ONE streaming pass counts the NEGATIVES between consecutive thresholds
//        (a search tree in LDS, LDS-privatised counters) - exact FP at every threshold, bounds in between - and, in the
//        same pass, ranks every negative among ALL positive values (the values of the bucket the LDS search lands in
//        arrive in one round of loads from an L2-resident table), which is the Mann-Whitney form of the ROC area: exact, no refinement.  (4) a second
//        pass with the thresholds of the few segments that can still hold the F1 maximum settles it exactly
//        (sg_pr_amd/metrics.py).  HBM-bound integer work: coalesced 16-B reads, no global atomics inside the loop.
//  * sgpr_topk_rows       - loop-closure candidates: for every query row the K best-scoring columns outside a temporal
//                           exclusion window, deterministic (score descending, column ascending).
#include <math.h>
#include <string.h>

#include <string>

#include "sgpr_internal.hpp"

namespace sgpr {

constexpr int PC_THREADS = 1024;
constexpr int PC_MAX_THRESHOLDS = 8191;   // + at least one +inf pad = 8192 floats of LDS, 8192 counters beside them

// how a pair (row r of the rectangle, column c) is labelled
struct PairTruth {
    int row0;                 // global index of row 0 (rows are a shard of the square matrix)
    const double* pose;       // [>= row0 + R and >= M][2] planar pose (x, z) or NULL; float64 like the reference
    double d_pos, d_neg;      // positive if distance <= d_pos, negative if >= d_neg, ignored in between
    const signed char* gt;    // explicit labels [R][ldg]: 1 / 0 / negative = ignore (used when pose == NULL)
    int64_t ldg;
};

// 1 positive, 0 negative, -1 ignored
__device__ __forceinline__ int classify_pair(const PairTruth& t, int r, int c, double px, double pz, double lo2, double hi2) {
    if (t.pose) {
        // utils.py:36 in float64, operation by operation (no fused multiply-add)
        const double dx = px - t.pose[2 * c], dz = pz - t.pose[2 * c + 1];
        const double s2 = __dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dz, dz));
        // sqrt is monotone and correctly rounded: away from the two thresholds the squared distance decides; only
        // within a relative 1e-12 of them is the reference's `sqrt(...) <= t` evaluated literally
        if (s2 < lo2 * (1.0 - 1e-12)) return 1;
        if (s2 > lo2 * (1.0 + 1e-12) && s2 < hi2 * (1.0 - 1e-12)) return -1;
        if (s2 > hi2 * (1.0 + 1e-12)) return 0;
        const double d = sqrt(s2);
        return d <= t.d_pos ? 1 : (d >= t.d_neg ? 0 : -1);
    }
    const int g = t.gt[(int64_t)r * t.ldg + c];
    return g < 0 ? -1 : (g != 0);
}

struct PairScan {
    const float* score;
    int R, M;
    int64_t ld;
    PairTruth truth;
};

// ---- (1) the scores of the positive pairs, appended in no particular order (the caller sorts them).
//      out == NULL: count only.  count[0] = positives, count[1] = positives whose score is negative or NaN.
//      Pose mode touches the matrix only where a pair is positive: the scan itself is arithmetic on the poses.
constexpr int PP_BUF = 2048;          // staged positives per workgroup between two flushes (one global atomic each)
__global__ __launch_bounds__(256) void pair_positives_kernel(const PairScan a, float* __restrict__ out, long long cap,
                                                             unsigned long long* __restrict__ count) {
    __shared__ float buf[PP_BUF];
    __shared__ unsigned nbuf;
    __shared__ unsigned long long gbase;
    const int lane = threadIdx.x & 63;
    const double lo2 = a.truth.d_pos * a.truth.d_pos, hi2 = a.truth.d_neg * a.truth.d_neg;
    unsigned nbad = 0;
    if (threadIdx.x == 0) nbuf = 0u;
    __syncthreads();
    // every global atomic on the ONE list counter costs ~10 ns and they serialise: positives are staged in LDS and a
    // workgroup reserves its piece of the list once per flush
    auto flush = [&]() {                                 // (called by the whole workgroup, between barriers)
        const unsigned n = nbuf;
        if (threadIdx.x == 0 && n) gbase = atomicAdd(&count[0], (unsigned long long)n);
        __syncthreads();
        if (out)
            for (unsigned i = threadIdx.x; i < n; i += 256) {
                const long long idx = (long long)gbase + i;
                if (idx < cap) out[idx] = buf[i];
            }
        __syncthreads();
        if (threadIdx.x == 0) nbuf = 0u;
        __syncthreads();
    };
    // rows by workgroup, columns by lane, four column blocks per iteration so that four pose loads are in flight
    for (int r = blockIdx.x; r < a.R; r += gridDim.x) {
        double px = 0.0, pz = 0.0;
        if (a.truth.pose) {
            px = a.truth.pose[2 * (a.truth.row0 + r)];
            pz = a.truth.pose[2 * (a.truth.row0 + r) + 1];
        }
        for (int cb = 0; cb < a.M; cb += 4 * 256) {
            bool pos[4];
            int cc[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c = cb + q * 256 + (int)threadIdx.x;
                cc[q] = min(c, a.M - 1);
                pos[q] = c < a.M;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) pos[q] = pos[q] && classify_pair(a.truth, r, cc[q], px, pz, lo2, hi2) == 1;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float s = 0.f;
                if (pos[q]) {
                    s = a.score[(int64_t)r * a.ld + cc[q]];
                    if (__float_as_uint(s) > 0x7f800000u) {          // negative or NaN: its order is undefined
                        ++nbad;
                        pos[q] = false;
                    }
                }
                const unsigned long long m = __ballot(pos[q]);
                if (m) {
                    const int leader = __ffsll((long long)m) - 1;
                    unsigned at = 0u;
                    if (lane == leader) at = atomicAdd(&nbuf, (unsigned)__popcll(m));
                    at = __shfl(at, leader);
                    if (pos[q]) buf[at + __popcll(m & ((1ull << lane) - 1ull))] = s;
                }
            }
            __syncthreads();
            // room for one more iteration is gone?  The decision is latched by every thread BEFORE anyone can append
            // again (the next iteration's atomicAdd on nbuf): a fast wave must not change what a slow wave still reads,
            // or the waves would disagree on entering flush() and its barriers
            const bool full = nbuf > PP_BUF - 4 * 256;
            __syncthreads();
            if (full) flush();
        }
    }
    __syncthreads();
    flush();
    if (nbad) atomicAdd(&count[1], (unsigned long long)nbad);
}

struct CountArgs {
    PairScan scan;
    const float* thr;                    // [T] ascending thresholds (scores of positive pairs)
    int T, Tp;                           // Tp = the power of two above T (thresholds padded with +inf in LDS)
    // optional exact ranking of every negative among ALL distinct positive values (ROC area): bucket b > 0 of the
    // thresholds holds the values rank[(b - 1) * gpt .. b * gpt), eight per record, thr[b - 1] the first of them
    const sgpr_rank_group* rank;         // [T * gpt]
    const unsigned long long* at_least;  // [T]: positive pairs with a value >= thr[q]
    int gpt;
    unsigned* slabs;                     // [gridDim.x][slab_words]: T + 1 counters (padded to even) | bad (u64) | rank sum (u64)
    int slab_words;
    const int* dT;                       // optional: T lives on the device (sgpr_f1_max: the thresholds are picked by a kernel);
                                         // *dT < 0 = nothing to count (the launch returns without touching the slabs)
};

// ---- (3) negatives by threshold bucket; bucket b = #{q : thr[q] <= s}, so FP(>= thr[q]) = sum of buckets b > q.
//      rank sum = sum over negatives of 2 #{positive pairs > s} + #{positive pairs == s}  (= 2 P N AUC)
__global__ __launch_bounds__(PC_THREADS) void pair_threshold_count_kernel(const CountArgs a_in) {
    extern __shared__ __attribute__((aligned(16))) unsigned char pc_smem[];
    CountArgs a = a_in;
    if (a.dT) {                          // thresholds picked on the device: their number, too
        const int t = *a.dT;
        if (t < 0) return;
        a.T = t;
        a.Tp = 1;
        while (a.Tp <= t) a.Tp <<= 1;
    }
    float* thr = reinterpret_cast<float*>(pc_smem);                       // [Tp]
    unsigned* cnt = reinterpret_cast<unsigned*>(thr + a.Tp);              // [T + 1]
    unsigned long long* at_least_lds = reinterpret_cast<unsigned long long*>(cnt + ((a.T + 2) & ~1));   // [T] (ranking only)
    if (a.rank)
        for (int i = threadIdx.x; i < a.T; i += PC_THREADS) at_least_lds[i] = a.at_least[i];
    // the thresholds as a complete binary search tree in breadth-first order (node 1 = root, children 2k / 2k + 1),
    // padded with +inf: the probes of one level are neighbours in LDS.  In the sorted array the probes of the upper
    // levels sit a power of two >= 32 words apart - all lanes of a wave in ONE bank, a 64-way conflict per step.
    const int lg = 31 - __clz(a.Tp);
    for (int node = threadIdx.x; node < a.Tp; node += PC_THREADS) {
        float v = INFINITY;
        if (node > 0) {
            const int l = 31 - __clz(node), i = node - (1 << l);
            const int idx = (((2 * i + 1) << (lg - 1 - l))) - 1;          // position of the node's key in sorted order
            if (idx < a.T) v = a.thr[idx];
        }
        thr[node] = v;
    }
    for (int i = threadIdx.x; i <= a.T; i += PC_THREADS) cnt[i] = 0u;
    __syncthreads();
    const PairScan& sc = a.scan;
    // items = (row, group of 4 columns), row-major; a workgroup owns a contiguous range of them and its threads walk
    // it with stride PC_THREADS (lanes along the row), carrying (row, group) along instead of dividing per item
    const int gpr = (sc.M + 3) >> 2;
    const int64_t items = (int64_t)sc.R * gpr;
    const int64_t per = (items + gridDim.x - 1) / gridDim.x;
    const int64_t it0 = blockIdx.x * per + threadIdx.x, it1 = min(items, (int64_t)(blockIdx.x + 1) * per);
    int r = (int)(it0 / gpr), g4 = (int)(it0 - (int64_t)r * gpr);
    const int step_r = PC_THREADS / gpr, step_g = PC_THREADS - step_r * gpr;
    unsigned nbad = 0;
    unsigned long long rank2 = 0ull;
    const double lo2 = sc.truth.d_pos * sc.truth.d_pos, hi2 = sc.truth.d_neg * sc.truth.d_neg;
    for (int64_t it = it0; it < it1; it += PC_THREADS, r += step_r, g4 += step_g) {
        if (g4 >= gpr) {
            g4 -= gpr;
            ++r;
        }
        const int c0 = g4 * 4;
        const float* sp = sc.score + (int64_t)r * sc.ld + c0;
        float s[4];
        if ((c0 + 3 < sc.M) && ((reinterpret_cast<uintptr_t>(sp) & 15) == 0)) {
            const float4 v = *reinterpret_cast<const float4*>(sp);
            s[0] = v.x; s[1] = v.y; s[2] = v.z; s[3] = v.w;
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) s[q] = c0 + q < sc.M ? sp[q] : 0.f;
        }
        double px = 0.0, pz = 0.0;
        if (sc.truth.pose) {
            px = sc.truth.pose[2 * (sc.truth.row0 + r)];
            pz = sc.truth.pose[2 * (sc.truth.row0 + r) + 1];
        }
        bool neg[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const bool inb = c0 + q < sc.M;            // (no early exit: the wave counts together below)
            neg[q] = inb && classify_pair(sc.truth, r, inb ? c0 + q : sc.M - 1, px, pz, lo2, hi2) == 0;
            if (neg[q] && __float_as_uint(s[q]) > 0x7f800000u) {          // negative or NaN
                ++nbad;
                neg[q] = false;
            }
        }
        // four independent branch-free descents of the tree (LDS), interleaved: right whenever the key is <= s; the
        // leaf reached is the number of thresholds <= s
        int b[4] = {1, 1, 1, 1};
        for (int lvl = 0; lvl < lg; ++lvl) {
#pragma unroll
            for (int q = 0; q < 4; ++q) b[q] = 2 * b[q] + ((thr[b[q]] <= s[q]) ? 1 : 0);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) b[q] -= a.Tp;
        if (a.rank) {
            // the bucket's values (and how many pairs carry each) arrive in ONE round of independent 16-byte loads from
            // the L2-resident table, two elements at a time: pairs above s = pairs from the bucket's first value on
            // - pairs of the bucket's values <= s
#pragma unroll
            for (int q0 = 0; q0 < 4; q0 += 2) {
                unsigned le[2] = {0u, 0u}, eq[2] = {0u, 0u};
                for (int gi = 0; gi < a.gpt; ++gi) {
                    float4 v[2][2];
                    uint4 m[2][2];
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const int bq = min(b[q0 + e], a.T);
                        const uint4* rec = reinterpret_cast<const uint4*>(a.rank + (size_t)max(bq - 1, 0) * a.gpt + gi);
                        v[e][0] = *reinterpret_cast<const float4*>(rec);
                        v[e][1] = *reinterpret_cast<const float4*>(rec + 1);
                        m[e][0] = rec[2];
                        m[e][1] = rec[3];
                    }
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const float x = s[q0 + e];
                        const float vv[8] = {v[e][0].x, v[e][0].y, v[e][0].z, v[e][0].w, v[e][1].x, v[e][1].y, v[e][1].z, v[e][1].w};
                        const unsigned mm[8] = {m[e][0].x, m[e][0].y, m[e][0].z, m[e][0].w, m[e][1].x, m[e][1].y, m[e][1].z, m[e][1].w};
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            le[e] += vv[i] <= x ? mm[i] : 0u;
                            eq[e] += vv[i] == x ? mm[i] : 0u;
                        }
                    }
                }
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int bq = min(b[q0 + e], a.T);
                    // b == 0: s lies below every positive value, all P = at_least[0] pairs rank above it
                    const unsigned long long gt_s = bq > 0 ? at_least_lds[bq - 1] - le[e] : at_least_lds[0];
                    rank2 += neg[q0 + e] ? 2ull * gt_s + (bq > 0 ? eq[e] : 0u) : 0ull;
                }
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int bq = min(b[q], a.T);
            // wave-aggregated counting: sigmoid scores of negatives pile up in a few buckets, and 64 lanes hitting one
            // LDS counter serialise.  Up to three rounds take the bucket of the first waiting lane with ONE atomic for all
            // the lanes that share it; whoever is left counts on its own.
            int idx = neg[q] ? bq : -1;
            unsigned long long todo = __ballot(idx >= 0);
#pragma unroll 1
            for (int round = 0; round < 3 && todo; ++round) {
                const int leader = __ffsll((long long)todo) - 1;
                const int lidx = __shfl(idx, leader);
                const unsigned long long same = __ballot(idx == lidx);
                if ((int)(threadIdx.x & 63) == leader) atomicAdd(&cnt[lidx], (unsigned)__popcll(same));
                if (idx == lidx) idx = -1;
                todo &= ~same;
                if (__popcll(same) < 4) break;           // scattered buckets: the rounds would only add instructions
            }
            if (idx >= 0) atomicAdd(&cnt[idx], 1u);
        }
    }
    // per-workgroup results go to the workgroup's own slab (no global atomics); slab_sum_kernel adds the slabs up
    __shared__ unsigned long long tail[2];
    if (threadIdx.x == 0) tail[0] = tail[1] = 0ull;
    __syncthreads();
    if (a.rank) {
#pragma unroll
        for (int m = 32; m > 0; m >>= 1) rank2 += __shfl_xor(rank2, m);
        if ((threadIdx.x & 63) == 0 && rank2) atomicAdd(&tail[1], rank2);
    }
    if (nbad) atomicAdd(&tail[0], (unsigned long long)nbad);
    __syncthreads();
    unsigned* slab = a.slabs + (size_t)blockIdx.x * a.slab_words;
    for (int i = threadIdx.x; i <= a.T; i += PC_THREADS) slab[i] = cnt[i];
    if (threadIdx.x < 2) reinterpret_cast<unsigned long long*>(slab + a.slab_words - 4)[threadIdx.x] = tail[threadIdx.x];
}

// out[i] = sum over the slabs: 32 counters per workgroup, 32 threads per counter (each sums every 32nd slab: one round
// of independent loads), 128-B coalesced reads
__global__ __launch_bounds__(1024) void slab_sum_kernel(const unsigned* __restrict__ slabs, int n_slabs, int slab_words, int T,
                                                        unsigned long long* __restrict__ out, const int* __restrict__ dT = nullptr) {
    __shared__ unsigned long long part[32][33];
    if (dT) {
        T = *dT;
        if (T < 0) return;
    }
    const int b = threadIdx.x & 31, grp = threadIdx.x >> 5;
    const int i = blockIdx.x * 32 + b;                  // counters 0..T, then T+1 = bad, T+2 = rank sum
    unsigned long long s = 0ull;
    if (i <= T) {
#pragma unroll 8
        for (int q = grp; q < n_slabs; q += 32) s += slabs[(size_t)q * slab_words + i];
    } else if (i <= T + 2) {
        for (int q = grp; q < n_slabs; q += 32)
            s += reinterpret_cast<const unsigned long long*>(slabs + (size_t)q * slab_words + slab_words - 4)[i - T - 1];
    }
    part[grp][b] = s;
    __syncthreads();
    if (grp == 0 && i <= T + 2) {
#pragma unroll
        for (int q = 1; q < 32; ++q) s += part[q][b];
        out[i] = s;
    }
}

// ------------------------------------------------------------------ F1-max in one call (sgpr_f1_max)
// eval_batch.py:69, 85-87 without a sort of the matrix and without a host round trip between the steps.  F1(t) can only
// peak at the score t of a POSITIVE pair, F1 = f(TP(>= t), FP(>= t)) grows with TP and falls with FP, so:
//   1. pair_positives_kernel            the scores of the positive pairs (a short list: loop closures are rare)
//   2. f1_pick_kernel (one workgroup)   up to 8191 of them, evenly spaced in list order, sorted and de-duplicated in LDS
//                                       = the thresholds; every positive is bucketed among them (b = #{thr <= s})
//   3. pair_threshold_count_kernel      the negatives by bucket: one streaming pass over the matrix
//   4. f1_plan_kernel (one workgroup)   exact F1 at every threshold; for the positives strictly inside bucket b the
//                                       bound F1(#{pos > thr[b-1]}, #{neg >= thr[b]}); buckets whose bound beats the
//                                       best exact value hand their interior positives to a second list
//   5. steps 2-3 on that list (every value a threshold), f1_final_kernel: exact F1 there; the maximum is exact.
// Everything the steps decide lives in a control block on the device; the host reads 8 doubles at the end.  Lists the
// device path is not built for (more than 2^20 positives, more than 8191 values to settle) are reported in the status
// word and the caller falls back to the multi-call path (sg_pr_amd/metrics.py), which handles any size.
constexpr int F1_MAXT = PC_MAX_THRESHOLDS;     // 8191
constexpr int F1_SORT = 8192;
struct F1Ctrl {
    int T1, T2;                  // thresholds of pass 1 / 2 (-1: pass not needed)
    int n2;                      // values collected for pass 2
    int status;                  // 0 ok, 1 fall back (sizes), 2 negative / NaN scores
    double best1;
    unsigned long long P, N;     // positive / negative pairs
};

__device__ __forceinline__ double f1_of(double tp, double fp, double pos) {
    // metrics._f1: p = tp / (tp + fp) (0 without predictions), r = tp / pos, f = 2 p r / (p + r), nan -> 0
    const double p = tp + fp > 0.0 ? tp / (tp + fp) : 0.0;
    const double r = pos > 0.0 ? tp / pos : 1.0;
    const double f = 2.0 * p * r / (p + r);
    return f == f ? f : 0.0;
}

// One workgroup.  cand[0..nc): values to draw the thresholds from (every stride-th, at most F1_PICK of them).
// Out: thr[T] ascending distinct, *dT = T, posc / eqc zeroed for f1_bucket_kernel.
// The bitonic network runs on 16 waves that each own 1/16 of the array: every step whose partner distance stays inside
// a wave's block needs no workgroup barrier (the LDS serves a wave's accesses in order), only the log2(16) + ... steps
// that cross blocks do.
constexpr int F1_PICK = 4095;                  // thresholds of a pass at most (a 4096-entry sort; 12 levels of the count tree)
__global__ __launch_bounds__(1024) void f1_pick_kernel(const float* __restrict__ cand, const unsigned long long* __restrict__ nc_dev,
                                                       const int* __restrict__ nc_int, const unsigned long long* __restrict__ na_dev,
                                                       long long cap, float* __restrict__ thr, int* __restrict__ dT,
                                                       unsigned* __restrict__ posc, unsigned* __restrict__ eqc,
                                                       F1Ctrl* __restrict__ ctrl, int pass) {
    __shared__ float v[F1_PICK + 1];
    __shared__ int scan[1024 / 64 + 1];
    __shared__ int total;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const long long na = (long long)na_dev[0];
    const long long nc = nc_int ? (long long)*nc_int : (long long)nc_dev[0];
    for (int i = tid; i < F1_SORT; i += 1024) {
        posc[i] = 0u;
        eqc[i] = 0u;
    }
    if (pass == 1 && (na > cap || ctrl->status != 0)) {                    // list longer than its buffer: the caller falls back
        if (tid == 0) {
            if (ctrl->status == 0) ctrl->status = 1;
            *dT = -1;
        }
        return;
    }
    if (pass == 2 && (ctrl->status != 0 || nc <= 0)) {                     // nothing left to settle (or a fall-back already)
        if (tid == 0) *dT = -1;
        return;
    }
    // ---- sample: every stride-th candidate, +inf padding to a power of two (>= 1024: one pair per thread and step)
    const long long stride = (nc + F1_PICK - 1) / F1_PICK > 0 ? (nc + F1_PICK - 1) / F1_PICK : 1;
    const int ns = (int)((nc + stride - 1) / stride);                      // <= F1_PICK
    int np2 = 2048;
    while (np2 < ns) np2 <<= 1;
    for (int i = tid; i < np2; i += 1024) v[i] = i < ns ? cand[(long long)i * stride] : INFINITY;
    __syncthreads();
    // ---- bitonic sort: wave w owns pairs [w * np2/32, (w+1) * np2/32), i.e. elements [w * np2/16, (w+1) * np2/16)
    const int ppw = np2 >> 5, blk = np2 >> 4;                              // pairs / elements per wave
    bool crossed = false;                                                  // the previous step crossed wave blocks
    for (int k = 2; k <= np2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            const bool cross = 2 * j > blk;
            if (cross || crossed) __syncthreads();
            crossed = cross;
            for (int u = lane; u < ppw; u += 64) {
                const int t = w * ppw + u;                                 // pair t: elements i and i + j
                const int i = 2 * t - (t & (j - 1));
                const int l = i + j;
                const float a = v[i], b = v[l];
                const bool up = (i & k) == 0;
                if ((a > b) == up) {
                    v[i] = b;
                    v[l] = a;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");         // (program order inside the wave is all it takes)
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    __syncthreads();
    // ---- distinct values: flag, block scan, compact (ascending order is kept)
    constexpr int PER = (F1_PICK + 1) / 1024;                              // 4 consecutive entries per thread
    int keep[PER], mine = 0;
    float vals[PER];
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        const int i = tid * PER + q;
        vals[q] = i < np2 ? v[i] : INFINITY;
        keep[q] = (i < ns && (i == 0 || vals[q] != v[i - 1])) ? 1 : 0;
        mine += keep[q];
    }
    int incl = mine;
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) {
        const int o = __shfl_up(incl, m);
        if (lane >= m) incl += o;
    }
    if (lane == 63) scan[w] = incl;
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        for (int q = 0; q < 1024 / 64; ++q) {
            const int c = scan[q];
            scan[q] = run;
            run += c;
        }
        total = run;
    }
    __syncthreads();
    int at = scan[w] + incl - mine;
#pragma unroll
    for (int q = 0; q < PER; ++q)
        if (keep[q]) thr[at++] = vals[q];
    if (tid == 0) {
        *dT = total;
        if (pass == 1) ctrl->T1 = total; else ctrl->T2 = total;
    }
}

// Every positive among the thresholds of a pass: b = #{thr <= s} (bisection in an LDS copy of the thresholds), bid[i] = b |
// (equal to thr[b-1]) << 15; counters posc[b] / eqc[b] first in LDS (random LDS atomics are cheap; 47 k global atomics on
// 4 k hot addresses were measured at 0.5 ms), then one global atomic per bucket the workgroup touched.
__global__ __launch_bounds__(1024) void f1_bucket_kernel(const float* __restrict__ all, const unsigned long long* __restrict__ na_dev,
                                                         const float* __restrict__ thr, const int* __restrict__ dT,
                                                         unsigned* __restrict__ posc, unsigned* __restrict__ eqc,
                                                         unsigned short* __restrict__ bid) {
    __shared__ float v[F1_PICK + 1];
    __shared__ unsigned pc[F1_PICK + 1], ec[F1_PICK + 1];
    const int T = *dT;
    if (T < 0) return;
    for (int i = threadIdx.x; i <= T; i += 1024) {
        v[i] = i < T ? thr[i] : INFINITY;
        pc[i] = 0u;
        ec[i] = 0u;
    }
    __syncthreads();
    const long long na = (long long)na_dev[0];
    for (long long i = (long long)blockIdx.x * 1024 + threadIdx.x; i < na; i += (long long)gridDim.x * 1024) {
        const float s = all[i];
        int lo = 0, hi = T;                                                // thr[lo-1] <= s < thr[hi]
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (v[mid] <= s) lo = mid + 1; else hi = mid;
        }
        const bool eq = lo > 0 && v[lo - 1] == s;
        atomicAdd(&pc[lo], 1u);
        if (eq) atomicAdd(&ec[lo], 1u);
        bid[i] = (unsigned short)(lo | (eq ? 0x8000 : 0));
    }
    __syncthreads();
    for (int i = threadIdx.x; i <= T; i += 1024) {
        if (pc[i]) atomicAdd(&posc[i], pc[i]);
        if (ec[i]) atomicAdd(&eqc[i], ec[i]);
    }
}

// The interior positives of the marked buckets = the values the second pass settles (f1_plan_kernel decided: ctrl->n2 of
// them, at most F1_PICK)
__global__ __launch_bounds__(256) void f1_collect_kernel(const float* __restrict__ all, const unsigned long long* __restrict__ na_dev,
                                                         const unsigned short* __restrict__ bid, const unsigned char* __restrict__ mark,
                                                         const F1Ctrl* __restrict__ ctrl, float* __restrict__ list2,
                                                         int* __restrict__ cursor) {
    __shared__ unsigned char mk[F1_SORT];
    if (ctrl->status != 0 || ctrl->n2 <= 0) return;
    for (int i = threadIdx.x; i < F1_SORT; i += 256) mk[i] = mark[i];
    __syncthreads();
    const long long na = (long long)na_dev[0];
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < na; i += (long long)gridDim.x * 256) {
        const unsigned b = bid[i];
        if (!(b & 0x8000u) && mk[b]) list2[atomicAdd(cursor, 1)] = all[i];
    }
}

// block-wide sums of two 64-bit values per thread -> exclusive prefix over the threads in DESCENDING thread order (suffix
// sums: thread t gets the total of the threads above it); sh: [2][1024 / 64 + 1] scratch
__device__ __forceinline__ void suffix_scan2(unsigned long long& a, unsigned long long& b, unsigned long long (*sh)[17]) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    unsigned long long ia = a, ib = b;                                     // inclusive over lanes >= this one
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) {
        const unsigned long long oa = __shfl_down(ia, m), ob = __shfl_down(ib, m);
        if (lane + m < 64) {
            ia += oa;
            ib += ob;
        }
    }
    if (lane == 0) {
        sh[0][w] = ia;
        sh[1][w] = ib;
    }
    __syncthreads();
    unsigned long long ua = 0ull, ub = 0ull;                               // waves above this one
    for (int q = w + 1; q < 16; ++q) {
        ua += sh[0][q];
        ub += sh[1][q];
    }
    __syncthreads();
    a = ua + ia - a;                                                       // exclusive: everything above this thread
    b = ub + ib - b;
}

__device__ __forceinline__ double block_max(double x, double* sh) {
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) x = fmax(x, __shfl_xor(x, m));
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = x;
    __syncthreads();
    double r = sh[0];
    for (int q = 1; q < 16; ++q) r = fmax(r, sh[q]);
    __syncthreads();
    return r;
}

// One workgroup.  negc[b] (uint64, from slab_sum_kernel; negc[T+1] = negatives with unusable scores), posc / eqc of pass 1.
__global__ __launch_bounds__(1024) void f1_plan_kernel(const unsigned long long* __restrict__ negc, const unsigned* __restrict__ posc,
                                                       const unsigned* __restrict__ eqc, const unsigned long long* __restrict__ count,
                                                       unsigned char* __restrict__ mark, int* __restrict__ n2_dev,
                                                       F1Ctrl* __restrict__ ctrl) {
    __shared__ unsigned long long sh[2][17];
    __shared__ double shd[16];
    __shared__ unsigned n2s;
    const int tid = threadIdx.x;
    if (ctrl->status != 0) return;
    const int T = ctrl->T1;
    if (tid == 0) n2s = 0u;
    if (count[1] != 0ull || negc[T + 1] != 0ull) {                         // negative / NaN scores have no rank
        if (tid == 0) {
            ctrl->status = 2;
            *n2_dev = 0;
        }
        return;
    }
    constexpr int PER = F1_SORT / 1024;
    unsigned long long ng[PER + 1], ps[PER + 1];                           // [q]: buckets tid*PER+q .. of this thread and above
    unsigned long long sn = 0ull, sp = 0ull;
#pragma unroll
    for (int q = PER - 1; q >= 0; --q) {
        const int b = tid * PER + q;
        sn += b <= T ? negc[b] : 0ull;
        sp += b <= T ? (unsigned long long)posc[b] : 0ull;
        ng[q] = sn;
        ps[q] = sp;
    }
    unsigned long long an = sn, ap = sp;
    suffix_scan2(an, ap, sh);                                              // totals of the buckets of the threads above
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        ng[q] += an;
        ps[q] += ap;
    }
    ng[PER] = an;
    ps[PER] = ap;
    __shared__ unsigned long long tot[2];
    if (tid == 0) {
        tot[0] = ps[0];                                                    // all positives / negatives
        tot[1] = ng[0];
    }
    __syncthreads();
    const double P = (double)tot[0];
    // exact F1 at threshold b - 1 = F1(positives / negatives in buckets >= b)
    double best = 0.0;
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        const int b = tid * PER + q;
        if (b >= 1 && b <= T) best = fmax(best, f1_of((double)ps[q], (double)ng[q], P));
    }
    best = block_max(best, shd);
    // bound for the positives strictly inside bucket b: at most the positives above thr[b-1], at least the negatives >= thr[b]
    unsigned mine2 = 0u;
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        const int b = tid * PER + q;
        unsigned char m = 0;
        if (b <= T) {
            const unsigned inside = posc[b] - eqc[b];
            if (inside > 0u && f1_of((double)(ps[q] - eqc[b]), (double)ng[q + 1], P) > best) {
                m = 1;
                mine2 += inside;
            }
        }
        if (b < F1_SORT) mark[b] = m;
    }
    if (mine2) atomicAdd(&n2s, mine2);
    __threadfence_block();
    __syncthreads();
    const unsigned n2 = n2s;
    if (tid == 0) {
        ctrl->best1 = best;
        ctrl->P = tot[0];
        ctrl->N = tot[1];
        ctrl->n2 = (int)n2;
        if (n2 > (unsigned)F1_PICK) ctrl->status = 1;                      // too many values to settle in one more pass
        *n2_dev = n2 > (unsigned)F1_PICK ? 0 : (int)n2;
    }
}

__global__ __launch_bounds__(1024) void f1_final_kernel(const unsigned long long* __restrict__ negc2, const unsigned* __restrict__ posc2,
                                                        const F1Ctrl* __restrict__ ctrl, double* __restrict__ result) {
    __shared__ unsigned long long sh[2][17];
    __shared__ double shd[16];
    const int tid = threadIdx.x;
    double best = ctrl->best1;
    int passes = 1;
    if (ctrl->status == 0 && ctrl->n2 > 0) {
        const int T = ctrl->T2;
        constexpr int PER = F1_SORT / 1024;
        unsigned long long ng[PER], ps[PER], sn = 0ull, sp = 0ull;
#pragma unroll
        for (int q = PER - 1; q >= 0; --q) {
            const int b = tid * PER + q;
            sn += b <= T ? negc2[b] : 0ull;
            sp += b <= T ? (unsigned long long)posc2[b] : 0ull;
            ng[q] = sn;
            ps[q] = sp;
        }
        unsigned long long an = sn, ap = sp;
        suffix_scan2(an, ap, sh);
        const double P = (double)ctrl->P;
        double b2 = 0.0;
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int b = tid * PER + q;
            if (b >= 1 && b <= T) b2 = fmax(b2, f1_of((double)(ps[q] + ap), (double)(ng[q] + an), P));
        }
        best = fmax(best, block_max(b2, shd));
        passes = 2;
    }
    if (tid == 0) {
        result[0] = ctrl->status == 0 ? fmax(best, 0.0) : 0.0;
        result[1] = (double)ctrl->status;
        result[2] = (double)ctrl->P;
        result[3] = (double)ctrl->N;
        result[4] = (double)passes;
        result[5] = (double)ctrl->T1;
        result[6] = (double)ctrl->n2;
        result[7] = 0.0;
    }
}

// ------------------------------------------------------------------ top-K per row
template <int K>
__global__ __launch_bounds__(256) void topk_rows_kernel(const float* __restrict__ score, int R, int M, int64_t ld,
                                                        int row0, int window, float* __restrict__ out_val,
                                                        int32_t* __restrict__ out_idx) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    // lane-local best K (descending value, ascending column) over columns lane, lane + 64, ...
    float v[K];
    int ix[K];
#pragma unroll
    for (int q = 0; q < K; ++q) {
        v[q] = -INFINITY;
        ix[q] = 0x7fffffff;
    }
    const float* sp = score + (int64_t)r * ld;
    const int self = row0 + r;
    for (int c = lane; c < M; c += 64) {
        const int dc = c - self;
        if ((dc < 0 ? -dc : dc) <= window) continue;
        float x = sp[c];
        if (!(x == x)) x = -INFINITY;                       // NaN ranks last
        int xi = c;
        if (x > v[K - 1]) {                                 // columns arrive in ascending order: ties keep the earlier one
            bool carry = false;                             // once an entry is displaced, everything behind it shifts
#pragma unroll
            for (int q = 0; q < K; ++q) {
                const bool before = carry || x > v[q];
                carry = before;
                const float tv = before ? v[q] : x;
                const int ti = before ? ix[q] : xi;
                v[q] = before ? x : v[q];
                ix[q] = before ? xi : ix[q];
                x = tv;
                xi = ti;
            }
        }
    }
    // K rounds: the wave's best head wins (value descending, column ascending), its lane pops
#pragma unroll 1
    for (int round = 0; round < K; ++round) {
        float bv = v[0];
        int bi = ix[0];
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) {
            const float ov = __shfl_xor(bv, m);
            const int oi = __shfl_xor(bi, m);
            const bool take = ov > bv || (ov == bv && oi < bi);
            bv = take ? ov : bv;
            bi = take ? oi : bi;
        }
        if (lane == 0) {
            out_val[(size_t)r * K + round] = bv;
            out_idx[(size_t)r * K + round] = bi == 0x7fffffff ? -1 : bi;
        }
        if (ix[0] == bi && bi != 0x7fffffff) {              // columns are unique: exactly one lane owns the winner
#pragma unroll
            for (int q = 0; q + 1 < K; ++q) {
                v[q] = v[q + 1];
                ix[q] = ix[q + 1];
            }
            v[K - 1] = -INFINITY;
            ix[K - 1] = 0x7fffffff;
        }
    }
}

}  // namespace sgpr

using namespace sgpr;

static int check_scan(const char* who, const sgpr_handle* h, const float* d_score, int R, int M, int64_t ld,
                      const double* d_pose_xz, const signed char* d_gt, int64_t ldg) {
    if (!h || !d_score || R < 0 || M < 0 || ld < M || (!d_pose_xz && !d_gt) || (d_gt && !d_pose_xz && ldg < M)) {
        set_error(std::string(who) + ": NULL argument, negative size or leading dimension below M");
        return SGPR_E_INVALID;
    }
    return SGPR_OK;
}

static PairScan make_scan(const float* d_score, int R, int M, int64_t ld, int row0, const double* d_pose_xz, double d_pos,
                          double d_neg, const signed char* d_gt, int64_t ldg) {
    PairScan sc;
    memset(&sc, 0, sizeof(sc));
    sc.score = d_score;
    sc.R = R;
    sc.M = M;
    sc.ld = ld;
    sc.truth.row0 = row0;
    sc.truth.pose = d_pose_xz;
    sc.truth.d_pos = d_pos;
    sc.truth.d_neg = d_neg;
    sc.truth.gt = d_gt;
    sc.truth.ldg = ldg;
    return sc;
}

int sgpr_pair_positives(const sgpr_handle* h, const float* d_score, int R, int M, int64_t ld, int row0,
                        const double* d_pose_xz, double d_pos, double d_neg, const signed char* d_gt, int64_t ldg,
                        float* d_out, int64_t capacity, unsigned long long* d_count, void* stream) {
    int rc = check_scan("sgpr_pair_positives", h, d_score, R, M, ld, d_pose_xz, d_gt, ldg);
    if (rc != SGPR_OK) return rc;
    if (!d_count || capacity < 0 || (capacity > 0 && !d_out)) {
        set_error("sgpr_pair_positives: NULL count buffer or capacity without an output buffer");
        return SGPR_E_INVALID;
    }
    DeviceGuard guard(h->device);
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipError_t e = hipMemsetAsync(d_count, 0, 2 * sizeof(unsigned long long), s);
    if (e != hipSuccess) return hip_fail(e, "sgpr_pair_positives: memset");
    if ((int64_t)R * M == 0) return SGPR_OK;
    const PairScan sc = make_scan(d_score, R, M, ld, row0, d_pose_xz, d_pos, d_neg, d_gt, ldg);
    hipLaunchKernelGGL(pair_positives_kernel, dim3(h->num_cus * 8), dim3(256), 0, s, sc, capacity > 0 ? d_out : nullptr,
                       (long long)capacity, d_count);
    e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "pair_positives_kernel launch");
    return SGPR_OK;
}

static int slab_words(int T) { return ((T + 2) & ~1) + 4; }

size_t sgpr_pair_threshold_counts_workspace_bytes(const sgpr_handle* h, int T) {
    if (!h || T < 0 || T > PC_MAX_THRESHOLDS) return 0;
    return (size_t)h->num_cus * slab_words(T) * sizeof(unsigned);
}

int sgpr_pair_threshold_counts(const sgpr_handle* h, const float* d_score, int R, int M, int64_t ld, int row0,
                               const double* d_pose_xz, double d_pos, double d_neg, const signed char* d_gt,
                               int64_t ldg, const float* d_thresholds, int T, const sgpr_rank_group* d_rank,
                               int groups_per_threshold, const unsigned long long* d_at_least, unsigned long long* d_out,
                               void* d_workspace, size_t workspace_bytes, void* stream) {
    int rc = check_scan("sgpr_pair_threshold_counts", h, d_score, R, M, ld, d_pose_xz, d_gt, ldg);
    if (rc != SGPR_OK) return rc;
    if (!d_out || T < 0 || T > PC_MAX_THRESHOLDS || (T > 0 && !d_thresholds)) {
        set_error("sgpr_pair_threshold_counts: 0.." + std::to_string(PC_MAX_THRESHOLDS) + " thresholds and an output buffer");
        return SGPR_E_INVALID;
    }
    if (d_rank && (T < 1 || groups_per_threshold < 1 || !d_at_least)) {
        set_error("sgpr_pair_threshold_counts: the ranking needs thresholds, >= 1 value group per threshold and the pair counts");
        return SGPR_E_INVALID;
    }
    const size_t need = sgpr_pair_threshold_counts_workspace_bytes(h, T);
    if (!d_workspace || workspace_bytes < need) {
        set_error("sgpr_pair_threshold_counts: workspace of " + std::to_string(need) + " bytes required");
        return SGPR_E_WORKSPACE;
    }
    DeviceGuard guard(h->device);
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipError_t e;
    if ((int64_t)R * M == 0) {
        e = hipMemsetAsync(d_out, 0, (size_t)(T + 3) * sizeof(unsigned long long), s);
        if (e != hipSuccess) return hip_fail(e, "sgpr_pair_threshold_counts: memset");
        return SGPR_OK;
    }
    CountArgs a;
    memset(&a, 0, sizeof(a));
    a.scan = make_scan(d_score, R, M, ld, row0, d_pose_xz, d_pos, d_neg, d_gt, ldg);
    a.thr = d_thresholds;
    a.T = T;
    a.Tp = 1;
    while (a.Tp <= T) a.Tp <<= 1;
    a.rank = d_rank;
    a.at_least = d_at_least;
    a.gpt = d_rank ? groups_per_threshold : 0;
    a.slabs = static_cast<unsigned*>(d_workspace);
    a.slab_words = slab_words(T);
    static bool attr_set = false;  // benign race: idempotent
    if (!attr_set) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&pair_threshold_count_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);   // (+ 16 B of static LDS)
        if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(pair_threshold_count_kernel)");
        attr_set = true;
    }
    const size_t lds = (size_t)a.Tp * sizeof(float) + (size_t)((T + 2) & ~1) * sizeof(unsigned) +
                       (d_rank ? (size_t)T * sizeof(unsigned long long) : 0);
    hipLaunchKernelGGL(pair_threshold_count_kernel, dim3(h->num_cus), dim3(PC_THREADS), lds, s, a);
    e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "pair_threshold_count_kernel launch");
    hipLaunchKernelGGL(slab_sum_kernel, dim3((T + 3 + 31) / 32), dim3(1024), 0, s, a.slabs, h->num_cus, a.slab_words, T, d_out);
    e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "slab_sum_kernel launch");
    return SGPR_OK;
}

// ---- sgpr_f1_max: workspace = header (counts, control block, device-side sizes) | thresholds, bucket counters, marks of
//      both passes | negatives by bucket of both passes | second list | positives | their buckets | counter slabs
static size_t a256(size_t v) { return (v + 255) & ~(size_t)255; }
struct F1Layout {
    size_t off_thr1, off_thr2, off_posc1, off_eqc1, off_posc2, off_eqc2, off_mark, off_neg1, off_neg2, off_list2, off_pos, off_bid,
        off_slabs, total;
    long long cap;
};
static F1Layout f1_layout(const sgpr_handle* h, int R, int M) {
    F1Layout L;
    const long long pairs = (long long)R * M;
    L.cap = pairs < (1LL << 20) ? pairs : (1LL << 20);
    if (L.cap < 1) L.cap = 1;
    size_t off = 256;                                                       // header
    L.off_thr1 = off;  off += a256(F1_SORT * sizeof(float));
    L.off_thr2 = off;  off += a256(F1_SORT * sizeof(float));
    L.off_posc1 = off; off += a256(F1_SORT * sizeof(unsigned));
    L.off_eqc1 = off;  off += a256(F1_SORT * sizeof(unsigned));
    L.off_posc2 = off; off += a256(F1_SORT * sizeof(unsigned));
    L.off_eqc2 = off;  off += a256(F1_SORT * sizeof(unsigned));
    L.off_mark = off;  off += a256(F1_SORT);
    L.off_neg1 = off;  off += a256((F1_SORT + 4) * sizeof(unsigned long long));
    L.off_neg2 = off;  off += a256((F1_SORT + 4) * sizeof(unsigned long long));
    L.off_list2 = off; off += a256(F1_SORT * sizeof(float));
    L.off_pos = off;   off += a256((size_t)L.cap * sizeof(float));
    L.off_bid = off;   off += a256((size_t)L.cap * sizeof(unsigned short));
    L.off_slabs = off; off += a256((size_t)h->num_cus * slab_words(F1_MAXT) * sizeof(unsigned));
    L.total = off;
    return L;
}

size_t sgpr_f1_max_workspace_bytes(const sgpr_handle* h, int R, int M) {
    if (!h || R < 0 || M < 0) return 0;
    return f1_layout(h, R, M).total;
}

int sgpr_f1_max(const sgpr_handle* h, const float* d_score, int R, int M, int64_t ld, int row0, const double* d_pose_xz,
                double d_pos, double d_neg, const signed char* d_gt, int64_t ldg, double* d_result, void* d_workspace,
                size_t workspace_bytes, void* stream) {
    int rc = check_scan("sgpr_f1_max", h, d_score, R, M, ld, d_pose_xz, d_gt, ldg);
    if (rc != SGPR_OK) return rc;
    if (!d_result) {
        set_error("sgpr_f1_max: NULL result buffer");
        return SGPR_E_INVALID;
    }
    const F1Layout L = f1_layout(h, R, M);
    if (!d_workspace || workspace_bytes < L.total) {
        set_error("sgpr_f1_max: workspace of " + std::to_string(L.total) + " bytes required");
        return SGPR_E_WORKSPACE;
    }
    DeviceGuard guard(h->device);
    hipStream_t s = static_cast<hipStream_t>(stream);
    unsigned char* ws = static_cast<unsigned char*>(d_workspace);
    unsigned long long* count = reinterpret_cast<unsigned long long*>(ws);            // [2]
    F1Ctrl* ctrl = reinterpret_cast<F1Ctrl*>(ws + 64);
    int* dT1 = reinterpret_cast<int*>(ws + 192);
    int* dT2 = dT1 + 1;
    int* n2 = dT1 + 2;
    float* thr1 = reinterpret_cast<float*>(ws + L.off_thr1);
    float* thr2 = reinterpret_cast<float*>(ws + L.off_thr2);
    unsigned* posc1 = reinterpret_cast<unsigned*>(ws + L.off_posc1);
    unsigned* eqc1 = reinterpret_cast<unsigned*>(ws + L.off_eqc1);
    unsigned* posc2 = reinterpret_cast<unsigned*>(ws + L.off_posc2);
    unsigned* eqc2 = reinterpret_cast<unsigned*>(ws + L.off_eqc2);
    unsigned char* mark = ws + L.off_mark;
    unsigned long long* neg1 = reinterpret_cast<unsigned long long*>(ws + L.off_neg1);
    unsigned long long* neg2 = reinterpret_cast<unsigned long long*>(ws + L.off_neg2);
    float* list2 = reinterpret_cast<float*>(ws + L.off_list2);
    float* pos = reinterpret_cast<float*>(ws + L.off_pos);
    unsigned short* bid = reinterpret_cast<unsigned short*>(ws + L.off_bid);
    unsigned* slabs = reinterpret_cast<unsigned*>(ws + L.off_slabs);
    static_assert(sizeof(F1Ctrl) <= 128, "control block");
    hipError_t e = hipMemsetAsync(ws, 0, 256, s);                                      // counts, control block, sizes
    if (e != hipSuccess) return hip_fail(e, "sgpr_f1_max: memset");
    static bool attr_set = false;  // benign race: idempotent
    if (!attr_set) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&pair_threshold_count_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
        if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(f1 kernels)");
        attr_set = true;
    }
    const PairScan sc = make_scan(d_score, R, M, ld, row0, d_pose_xz, d_pos, d_neg, d_gt, ldg);
    if ((int64_t)R * M > 0)
        hipLaunchKernelGGL(pair_positives_kernel, dim3(h->num_cus * 8), dim3(256), 0, s, sc, pos, L.cap, count);
    const size_t count_lds = (size_t)F1_SORT * sizeof(float) + (size_t)F1_SORT * sizeof(unsigned);
    CountArgs a;
    memset(&a, 0, sizeof(a));
    a.scan = sc;
    a.slabs = slabs;
    a.slab_words = slab_words(F1_MAXT);
    int* cursor = n2 + 1;                                                              // append cursor of the second list
    for (int pass = 1; pass <= 2; ++pass) {
        float* thr = pass == 1 ? thr1 : thr2;
        int* dT = pass == 1 ? dT1 : dT2;
        unsigned* posc = pass == 1 ? posc1 : posc2;
        unsigned* eqc = pass == 1 ? eqc1 : eqc2;
        hipLaunchKernelGGL(f1_pick_kernel, dim3(1), dim3(1024), 0, s, pass == 1 ? pos : list2, count,
                           pass == 1 ? (const int*)nullptr : n2, count, L.cap, thr, dT, posc, eqc, ctrl, pass);
        hipLaunchKernelGGL(f1_bucket_kernel, dim3(32), dim3(1024), 0, s, pos, count, thr, dT, posc, eqc, bid);
        a.thr = thr;
        a.dT = dT;
        if ((int64_t)R * M > 0) {
            hipLaunchKernelGGL(pair_threshold_count_kernel, dim3(h->num_cus), dim3(PC_THREADS), count_lds, s, a);
            hipLaunchKernelGGL(slab_sum_kernel, dim3((F1_MAXT + 3 + 31) / 32), dim3(1024), 0, s, slabs, h->num_cus, a.slab_words,
                               0, pass == 1 ? neg1 : neg2, dT);
        }
        if (pass == 1) {
            hipLaunchKernelGGL(f1_plan_kernel, dim3(1), dim3(1024), 0, s, neg1, posc1, eqc1, count, mark, n2, ctrl);
            hipLaunchKernelGGL(f1_collect_kernel, dim3(64), dim3(256), 0, s, pos, count, bid, mark, ctrl, list2, cursor);
        }
    }
    hipLaunchKernelGGL(f1_final_kernel, dim3(1), dim3(1024), 0, s, neg2, posc2, ctrl, d_result);
    e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "sgpr_f1_max launches");
    return SGPR_OK;
}

int sgpr_topk_rows(const sgpr_handle* h, const float* d_score, int R, int M, int64_t ld, int row0, int window, int k,
                   float* d_values, int32_t* d_indices, void* stream) {
    if (!h || !d_score || !d_values || !d_indices || R < 0 || M < 0 || ld < M || window < -1) {
        set_error("sgpr_topk_rows: NULL argument, negative size or leading dimension below M");
        return SGPR_E_INVALID;
    }
    if (k != 1 && k != 4 && k != 8 && k != 16) {
        set_error("sgpr_topk_rows: k must be 1, 4, 8 or 16");
        return SGPR_E_K;
    }
    if (R == 0) return SGPR_OK;
    DeviceGuard guard(h->device);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const dim3 grid((R + 3) / 4), block(256);
    switch (k) {
        case 1: hipLaunchKernelGGL(topk_rows_kernel<1>, grid, block, 0, s, d_score, R, M, ld, row0, window, d_values, d_indices); break;
        case 4: hipLaunchKernelGGL(topk_rows_kernel<4>, grid, block, 0, s, d_score, R, M, ld, row0, window, d_values, d_indices); break;
        case 8: hipLaunchKernelGGL(topk_rows_kernel<8>, grid, block, 0, s, d_score, R, M, ld, row0, window, d_values, d_indices); break;
        default: hipLaunchKernelGGL(topk_rows_kernel<16>, grid, block, 0, s, d_score, R, M, ld, row0, window, d_values, d_indices); break;
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "topk_rows_kernel launch");
    return SGPR_OK;
}
