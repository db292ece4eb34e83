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
// out_t (optional): the sums once more in the order the F1 plan kernel reads them - counter i = t * per + q of thread t
// at [q * 1024 + t] (one coalesced load per q instead of 192-byte strides between the lanes)
__global__ __launch_bounds__(1024) void slab_sum_kernel(const unsigned* __restrict__ slabs, int n_slabs, int slab_words, int T,
                                                        unsigned long long* __restrict__ out, const int* __restrict__ dT = nullptr,
                                                        unsigned long long* __restrict__ out_t = nullptr, int per = 1) {
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
        if (out_t && i <= T) out_t[(size_t)(i % per) * 1024 + i / per] = s;
    }
}

// ------------------------------------------------------------------ F1-max in one call (sgpr_f1_max)
// eval_batch.py:69, 85-87 without a sort of the matrix and without a host round trip between the steps.  F1(t) can only
// peak at the score t of a POSITIVE pair, and F1 = f(TP(>= t), FP(>= t)) grows with TP and falls with FP.  Round 4: a radix
// histogram on the score's bit pattern replaces the threshold search tree (13 dependent LDS reads per score, two passes
// of ~85 us each) and the separate scan for the positives:
//   A  f1_scan_kernel      ONE streaming pass: every pair is classified once; a negative costs one shift and one LDS
//                          atomic (bin = f1_key(score): a monotone map of the bit pattern with 5 mantissa bits of
//                          resolution in s below 1/2 and in 1 - s above - sigmoid scores crowd towards 1), a positive is
//                          appended to the (short) list of positive scores
//   .  slab_sum_kernel     the workgroups' histograms added up
//   P  f1_plan_kernel      (one workgroup) positives by bin; suffix sums give exact (TP, FP) at every bin edge - each an
//                          attained point of the curve - and for the positives INSIDE bin b the bound F1(TP(>= edge b),
//                          FP(>= edge b+1)); the bins whose bound beats the best edge value are the candidates: their
//                          positives (at most F1_PICK) are sorted and de-duplicated in LDS = the thresholds of pass B,
//                          with TP exact for each of them
//   B  f1_refine_kernel    second streaming pass: a negative outside the candidate bins costs one bit test; inside, a
//                          bisection among the thresholds and one LDS atomic
//   .  slab_sum_kernel
//   F  f1_final            (the workgroup of pass B that finishes last) exact FP, hence exact F1, at every threshold; the
//                          maximum with the edge values is exact.
// Everything the steps decide lives in a control block on the device; the host reads 8 doubles at the end.  Rectangles
// the device path is not built for (more than 2^20 positives, more than F1_PICK values to settle - a flat curve) are
// reported in the status word and the caller falls back to the multi-call path (sg_pr_amd/metrics.py).
#ifndef SGPR_F1_SH
#define SGPR_F1_SH 18          // (17, twice the bins: 126.6 against 120.2 us per KITTI-00 matrix - the plan kernel walks them,
#endif                         //  the histograms are written and added up; 389 against 506 values left to pass B)
constexpr int F1_SH = SGPR_F1_SH;                                // key = bit pattern >> 18: sign, exponent, 5 mantissa bits
constexpr int F1_HALF = 0x3F000000 >> F1_SH;                     // 4032 bins for s in [0, 1/2)
constexpr int F1_ONE = 2 * F1_HALF;                              // the bin of s == 1
constexpr int F1_NB = F1_ONE + 1 + ((0x7F800000 - 0x3F800000) >> F1_SH) + 1;   // 12 162 bins: ... and s in (1, +inf]
constexpr int F1_NBP = (F1_NB + 3) & ~3;
constexpr int F1_PICK = 4095;                                    // values pass B settles at most (a 4096-entry sort)
constexpr int F1_SORT = 4096;
constexpr int F1_PBUF = 8192;                                    // staged positives per workgroup between two flushes
constexpr int F1_THREADS = 1024;
constexpr int F1_PER = (F1_NBP + F1_THREADS - 1) / F1_THREADS;   // 12 consecutive bins per thread of the plan kernel
// where the plan kernel's thread bin / F1_PER finds the count of `bin` in the transposed per-bin arrays
__device__ __forceinline__ int f1_tslot(int bin) { return (bin % F1_PER) * F1_THREADS + bin / F1_PER; }

// monotone: s1 < s2 => key(s1) <= key(s2), for every s >= 0 (not NaN).  1 - s is exact for s in [1/2, 1] (Sterbenz).
__device__ __forceinline__ int f1_key(float s) {
    // branch-free: three shifts and two selects (the three-way branch cost ~27 scalar instructions per score in pass A)
    const unsigned b = __float_as_uint(s);
    const int lo = (int)(b >> F1_SH);
    const int mid = F1_ONE - (int)(__float_as_uint(1.0f - s) >> F1_SH);
    const int hi = F1_ONE + 1 + (int)((b - 0x3F800000u) >> F1_SH);
    return b < 0x3F000000u ? lo : (b <= 0x3F800000u ? mid : hi);
}

// The first bit pattern (a non-negative float's) whose key is >= want; 0x7f800001 (past +inf) when there is none.  Called
// by a whole wave: sixty-four probes per round, six rounds (one thread bisecting took 31 dependent rounds = 3 us with a
// workgroup waiting at the next barrier).
__device__ __forceinline__ unsigned f1_first_pattern(int want) {
    if (f1_key(0.f) >= want) return 0u;
    const unsigned lane = threadIdx.x & 63;
    unsigned lo = 0u, hi = 0x7f800001u;                    // key(lo) < want; hi: key >= want or the sentinel
    while (hi - lo > 1u) {
        const unsigned step = (hi - lo + 63u) >> 6;
        const unsigned long long c64 = (unsigned long long)lo + (unsigned long long)(lane + 1u) * step;
        const unsigned cand = c64 >= hi ? hi : (unsigned)c64;
        const bool ok = cand == 0x7f800001u || f1_key(__uint_as_float(cand)) >= want;
        const int first = __builtin_ctzll(__ballot(ok));   // (lane 63 probes hi: the vote is never empty)
        const unsigned below = first > 0 ? (unsigned)__shfl((int)cand, first - 1) : lo;
        hi = (unsigned)__shfl((int)cand, first);
        lo = below;
    }
    return hi;
}

struct F1Ctrl {
    int T2;                      // thresholds of pass B (-1: pass not needed)
    int n2;                      // positives inside the candidate bins
    int status;                  // 0 ok, 1 fall back (sizes), 2 negative / NaN scores
    int pad;
    double best1;                // best F1 at a bin edge
    unsigned long long P, N;     // positive / negative pairs
};

__device__ __forceinline__ double f1_of(double tp, double fp, double pos) {
    // metrics._f1: p = tp / (tp + fp) (0 without predictions), r = tp / pos, f = 2 p r / (p + r), nan -> 0
    const double p = tp + fp > 0.0 ? tp / (tp + fp) : 0.0;
    const double r = pos > 0.0 ? tp / pos : 1.0;
    const double f = 2.0 * p * r / (p + r);
    return f == f ? f : 0.0;
}

// Streaming layout of passes A and B: a WAVE owns a strip of 256 columns (four per lane) and walks a chunk of rows down
// it - the column poses are loaded once per task, the row pose is wave-uniform (scalar loads), a row of the strip is one
// contiguous kilobyte (16 bytes per lane, dword-aligned: rows of an M x M float matrix are not 16-byte aligned), and four
// rows are in flight per wave.  Tasks (row chunk, strip) are dealt to the waves of the grid strip-first, so that
// neighbouring waves read neighbouring kilobytes.
#ifndef SGPR_F1_ROWS
#define SGPR_F1_ROWS 4
#endif
#ifndef SGPR_F1_CULL
#define SGPR_F1_CULL 1         // strips culled by bounding box (pass A) / by the thresholds' range (pass B); 0: A/B builds
#endif
#ifndef SGPR_F1_WGB
#define SGPR_F1_WGB 1          // workgroups per CU of pass B (2: 27.4 against 24.1 us - twice the closing atomics)
#endif
constexpr int F1_ROWS = SGPR_F1_ROWS;                                       // rows per task = rows in flight per wave (5 tasks per wave on a KITTI-00 matrix)
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
typedef double f64x2u __attribute__((ext_vector_type(2), aligned(8)));

__device__ __forceinline__ void load_row4(const PairScan& sc, int r, int c0, float (&s)[4]) {
    const float* sp = sc.score + (int64_t)r * sc.ld + c0;
    if (c0 + 3 < sc.M) {
        const f32x4u v = *reinterpret_cast<const f32x4u*>(sp);
        s[0] = v[0]; s[1] = v[1]; s[2] = v[2]; s[3] = v[3];
    } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) s[q] = c0 + q < sc.M ? sp[q] : 0.f;
    }
}

// the planar poses of a lane's four columns (pose mode), loaded once per task
struct ColPoses {
    double x[4], z[4];
    __device__ __forceinline__ void load(const PairTruth& t, int c0, int M) {
        if (!t.pose) return;
        if (c0 + 3 < M) {
            const f64x2u* cp = reinterpret_cast<const f64x2u*>(t.pose + 2 * c0);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f64x2u v = cp[q];
                x[q] = v[0];
                z[q] = v[1];
            }
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c = min(c0 + q, M - 1);
                x[q] = t.pose[2 * c];
                z[q] = t.pose[2 * c + 1];
            }
        }
    }
};

// A wave's place in a streaming pass: virtual wave v of strips * groups owns strip v % strips and the row quads
// v / strips, + groups, + 2 groups, ... of it (neighbouring waves: neighbouring kilobytes of the same rows).  With at
// least as many waves as strips (always, short of a million columns) every wave has one strip for the whole pass.
struct StripWalk {
    int strips, groups, nquads;
    long long nvirt;
    __device__ __forceinline__ StripWalk(const PairScan& sc, long long nwaves) {
        strips = (sc.M + 255) >> 8;
        nquads = (sc.R + F1_ROWS - 1) / F1_ROWS;
        groups = (int)max(1ll, min((long long)nquads, nwaves / strips));
        nvirt = (long long)strips * groups;
    }
};

// four rows of a strip: 16 bytes per lane and row, requested together (with the rows' poses in pose mode: scalar loads)
template <typename PT>          // PT: float (the bounding-box test), double (the pair-by-pair arithmetic)
struct RowQuad {
    float s[F1_ROWS][4];
    PT px[F1_ROWS], pz[F1_ROWS];
    int r0, r1;
    __device__ __forceinline__ void fetch(const PairScan& sc, int quad, int nquads, int c0, float fill, bool poses) {
        r0 = r1 = 0;
#pragma unroll
        for (int u = 0; u < F1_ROWS; ++u) {
            px[u] = pz[u] = (PT)0;
#pragma unroll
            for (int q = 0; q < 4; ++q) s[u][q] = fill;
        }
        if (quad >= nquads) return;                       // (wave-uniform)
        r0 = quad * F1_ROWS;
        r1 = min(sc.R, r0 + F1_ROWS);
        if (poses) {
#pragma unroll
            for (int u = 0; u < F1_ROWS; ++u) {
                const f64x2u v = *reinterpret_cast<const f64x2u*>(sc.truth.pose + 2 * (size_t)(sc.truth.row0 + min(r0 + u, r1 - 1)));
                px[u] = (PT)v[0];
                pz[u] = (PT)v[1];
            }
        }
        if (c0 < sc.M) {
#pragma unroll
            for (int u = 0; u < F1_ROWS; ++u)
                if (r0 + u < r1) load_row4(sc, r0 + u, c0, s[u]);
        }
    }
};

// classes (1 positive, 0 negative, -1 ignored / outside the matrix) of the pairs (r, c0 .. c0 + 3); classify_pair's
// arithmetic (utils.py:36 in float64, operation by operation) on the preloaded column poses
__device__ __forceinline__ void classify_row4(const PairTruth& t, const ColPoses& cp, const double px, const double pz, int r, int c0,
                                              int M, double lo2, double hi2, int (&cls)[4]) {
    if (t.pose) {                                        // (px, pz: the row's pose, wave-uniform, requested with its scores)
        // Almost every pair lies far from both thresholds: the float64 differences, squared and summed in fp32 (relative error
        // below 4 x 2^-24 = 2.4e-7), decide it when they clear a threshold by a relative 4e-6.  The float64 arithmetic of the
        // reference (utils.py:36) runs for a wave only when one of its pairs is closer to a threshold than that (or not
        // finite: every fp32 comparison is false) - the pass was bound by its float64 instructions (14.5 M vector
        // instructions per KITTI-00 matrix), not by HBM.
        const float plo = (float)(lo2 * (1.0 - 4e-6)), phi = (float)(lo2 * (1.0 + 4e-6));
        const float nlo = (float)(hi2 * (1.0 - 4e-6)), nhi = (float)(hi2 * (1.0 + 4e-6));
        double dxs[4], dzs[4];
        bool open = false;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            dxs[q] = px - cp.x[q];
            dzs[q] = pz - cp.z[q];
            const float fx = (float)dxs[q], fz = (float)dzs[q];
            const float s2f = fmaf(fx, fx, fz * fz);
            const int c = s2f < plo ? 1 : (s2f > nhi ? 0 : ((s2f > phi && s2f < nlo) ? -1 : 2));
            open = open || (c == 2 && c0 + q < M);
            cls[q] = c0 + q < M ? c : -1;
        }
        if (__ballot(open) != 0ull) {                                                      // (rare; wave-uniform branch)
            const double pos_lo = lo2 * (1.0 - 1e-12), pos_hi = lo2 * (1.0 + 1e-12), neg_lo = hi2 * (1.0 - 1e-12), neg_hi = hi2 * (1.0 + 1e-12);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (cls[q] != 2) continue;
                const double s2 = __dadd_rn(__dmul_rn(dxs[q], dxs[q]), __dmul_rn(dzs[q], dzs[q]));
                // sqrt is monotone and correctly rounded: away from the two thresholds the squared distance decides; only
                // within a relative 1e-12 of them is the reference's `sqrt(...) <= t` evaluated literally
                int c = s2 < pos_lo ? 1 : (s2 > neg_hi ? 0 : -1);
                if (!(s2 < pos_lo) && !(s2 > neg_hi) && !(s2 > pos_hi && s2 < neg_lo)) {
                    const double d = sqrt(s2);
                    c = d <= t.d_pos ? 1 : (d >= t.d_neg ? 0 : -1);
                }
                cls[q] = c;
            }
        }
    } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int g = c0 + q < M ? t.gt[(int64_t)r * t.ldg + c0 + q] : -1;
            cls[q] = g < 0 ? -1 : (g != 0);
        }
    }
}

// f1_key for a score in [0, 1] (bit pattern <= 1.0f's), times four: the byte offset of its bin
__device__ __forceinline__ unsigned f1_key01_x4(float s) {
    const bool low = __float_as_uint(s) < 0x3F000000u;
    const unsigned k4 = (__float_as_uint(low ? s : 1.0f - s) >> (F1_SH - 2)) & ~3u;
    return low ? k4 : 4u * F1_ONE - k4;
}

// min / max over the 16 lanes of a row of the wave (every lane gets the result): quad permutes, then the two mirrors
template <bool MAX, int CTRL>
__device__ __forceinline__ float dpp_minmax(float v) {
    const float o = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
    return MAX ? fmaxf(v, o) : fminf(v, o);
}
template <bool MAX>
__device__ __forceinline__ float row16_reduce(float v) {
    v = dpp_minmax<MAX, 0xB1>(v);        // quad_perm [1, 0, 3, 2]
    v = dpp_minmax<MAX, 0x4E>(v);        // quad_perm [2, 3, 0, 1]
    v = dpp_minmax<MAX, 0x141>(v);       // row_half_mirror
    return dpp_minmax<MAX, 0x140>(v);    // row_mirror
}

// The planar bounding box (fp32) of the 64 columns of a lane's row of the wave, and whether a row pose is so far from it
// that every pair (row, one of those columns) is a NEGATIVE without looking at it: almost every strip of a trajectory's
// matrix.  The box and the row pose are rounded to fp32 (relative 2^-24 each) and the gaps computed in fp32: the gap is
// cut by 2^-21 of the largest magnitude involved and the squared distance must clear max(d_neg, d_pos)^2 by 0.1 % - far beyond those
// roundings, so a strip passes only if the reference's float64 distance (utils.py:36) is >= d_neg for each of its pairs.
struct StripBox {
    float xlo, xhi, zlo, zhi, amax;
    __device__ __forceinline__ void init(const ColPoses& cp) {
        float a = (float)cp.x[0], b = a, c = (float)cp.z[0], d = c;
        bool nan = a != a || c != c;
#pragma unroll
        for (int q = 1; q < 4; ++q) {
            const float x = (float)cp.x[q], z = (float)cp.z[q];
            nan = nan || x != x || z != z;
            a = fminf(a, x);
            b = fmaxf(b, x);
            c = fminf(c, z);
            d = fmaxf(d, z);
        }
        if (nan) {                                        // (fmin / fmax drop a NaN: a column without a pose is IGNORED by the
            a = c = -INFINITY;                            //  reference, never a negative - its group's box covers the plane,
            b = d = INFINITY;                             //  no row is far from it)
        }
        xlo = row16_reduce<false>(a);
        xhi = row16_reduce<true>(b);
        zlo = row16_reduce<false>(c);
        zhi = row16_reduce<true>(d);
        amax = fmaxf(fmaxf(fabsf(xlo), fabsf(xhi)), fmaxf(fabsf(zlo), fabsf(zhi)));
    }
    // (NaN / infinite poses: a comparison with NaN is false -> not far -> the per-pair arithmetic decides)
    __device__ __forceinline__ bool far(float px, float pz, float cut2) const {
        const float e = fmaxf(fmaxf(amax, fabsf(px)), fabsf(pz)) * 4.76837158e-7f;          // 2^-21
        const float gx = fmaxf(fmaxf(xlo - px, px - xhi) - e, 0.f);
        const float gz = fmaxf(fmaxf(zlo - pz, pz - zhi) - e, 0.f);
        return fmaf(gx, gx, gz * gz) > cut2;
    }
};

// ---- A: negatives by key bin (LDS histogram of the workgroup -> its slab), positives appended to `pos` (count[0] of
//      them; count[1] = positives with a negative / NaN score); slab tail word 0 = negatives with such a score.
//      No workgroup barrier inside the loop: the positives are staged per WAVE (512 floats each, appended with a ballot
//      prefix, flushed with one global atomic by the wave).
constexpr int F1_WBUF = F1_PBUF / (F1_THREADS / 64);          // 512 staged positives per wave
#ifndef SGPR_F1_SLOWQ
#define SGPR_F1_SLOWQ 1024     // (variant builds with a tiny list exercise the several-rounds path: tools/gpu_r6t.sh TEST_VARIANT)
#endif
constexpr int F1_SLOWQ = SGPR_F1_SLOWQ;                                // a workgroup's list of quads left to the pair-by-pair code
//      The classes of a lane's four pairs leave as one byte (2 bits each: 0 negative, 1 positive, 3 ignored) into
//      cls_out [R][(M + 3) / 4]: pass B reads that byte instead of repeating the pose arithmetic.
#ifndef SGPR_F1_SCAN_STAMPS
#define SGPR_F1_SCAN_STAMPS 0  // 1: every wave of pass A leaves six time stamps (tools/exp/f1_scan_timeline.py; variant builds only)
#endif
#if SGPR_F1_SCAN_STAMPS
__device__ unsigned long long f1_scan_stamps[8192 * 8];
#define F1A_STAMP(i) if ((threadIdx.x & 63) == 0) f1_scan_stamps[((size_t)blockIdx.x * (F1_THREADS / 64) + (threadIdx.x >> 6)) * 8 + (i)] = wall_clock64();
#else
#define F1A_STAMP(i)
#endif
__global__ __launch_bounds__(F1_THREADS) void f1_scan_kernel(const PairScan sc, unsigned* __restrict__ slabs, int slab_words,
                                                             float* __restrict__ pos, long long cap,
                                                             unsigned long long* __restrict__ count,
                                                             unsigned char* __restrict__ cls_out, unsigned* __restrict__ posb_g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char f1_smem[];
    unsigned* hist = reinterpret_cast<unsigned*>(f1_smem);                  // [F1_NBP]
    float* buf = reinterpret_cast<float*>(hist + F1_NBP) + (threadIdx.x >> 6) * F1_WBUF;   // this wave's staging area
    __shared__ unsigned nbad_neg, nslow, ntaken;
    __shared__ int wave_nst[F1_THREADS / 64];
    __shared__ unsigned long long wg_base;
    __shared__ int slow_strip[F1_SLOWQ], slow_quad[F1_SLOWQ];
    const int tid = threadIdx.x, lane = tid & 63;
    F1A_STAMP(0)
    for (int i = tid; i < F1_NBP; i += F1_THREADS) hist[i] = 0u;
    if (tid == 0) nbad_neg = nslow = ntaken = 0u;
    __syncthreads();
    F1A_STAMP(1)
    int nst = 0;                                          // staged positives of this wave (wave-uniform)
    auto flush = [&]() {
        unsigned long long g = 0ull;
        if (lane == 0) g = atomicAdd(&count[0], (unsigned long long)nst);
        g = __shfl(g, 0);
        for (int i = lane; i < nst; i += 64) {
            const long long idx = (long long)g + i;
            if (idx < cap) pos[idx] = buf[i];
            atomicAdd(&posb_g[f1_tslot(f1_key(buf[i]))], 1u);   // positives by bin (rare: ~1 pair in 400), for the plan kernel
        }
        nst = 0;
    };
    const double lo2 = sc.truth.d_pos * sc.truth.d_pos, hi2 = sc.truth.d_neg * sc.truth.d_neg;
    const size_t cw = (size_t)(sc.M + 3) >> 2;            // class bytes per row
    unsigned bad_pos = 0u, bad_neg = 0u;
    const float cut2 = (float)fmax(hi2, lo2) * 1.001f;    // (a positive comes first: with d_neg < d_pos "far" means beyond d_pos)
    const long long wave0 = (long long)blockIdx.x * (F1_THREADS / 64) + (tid >> 6), nwaves = (long long)gridDim.x * (F1_THREADS / 64);
    const StripWalk walk(sc, nwaves);
    // the four rows of a quad pair by pair (float64 pose arithmetic near the thresholds, positives staged)
    auto classic = [&](const RowQuad<double>& cur, const int c0, const ColPoses& cp) {
        const bool inb = c0 < sc.M;
#pragma unroll
        for (int u = 0; u < F1_ROWS; ++u) {
            if (cur.r0 + u >= cur.r1) break;              // (wave-uniform)
            int cls[4] = {-1, -1, -1, -1};
            if (inb) {
                classify_row4(sc.truth, cp, cur.px[u], cur.pz[u], cur.r0 + u, c0, sc.M, lo2, hi2, cls);
                cls_out[(size_t)(cur.r0 + u) * cw + (c0 >> 2)] =
                    (unsigned char)((cls[0] & 3) | ((cls[1] & 3) << 2) | ((cls[2] & 3) << 4) | ((cls[3] & 3) << 6));
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float x = cur.s[u][q];
                const bool usable = __float_as_uint(x) <= 0x7f800000u;      // not negative, not NaN
                if (cls[q] == 0) {
                    if (usable) atomicAdd(&hist[f1_key(x)], 1u);
                    else ++bad_neg;
                }
                bool p = cls[q] == 1;
                if (p && !usable) {
                    ++bad_pos;
                    p = false;
                }
                const unsigned long long m = __ballot(p);
                if (m) {
                    if (p) buf[nst + __popcll(m & ((1ull << lane) - 1ull))] = x;
                    nst += __popcll(m);
                }
            }
            if (nst > F1_WBUF - 256) flush();             // (wave-uniform) room for one more row is gone
        }
    };
    // A strip that lies inside the matrix gets its columns' bounding boxes (one per 64 columns): a quad of rows far from
    // all four is 1024 negatives - no per-pair pose arithmetic, and scores in [0, 1] take the short form of the key.  The
    // wave streams down its strip counting such quads (the next quad's kilobytes on their way meanwhile) and leaves the
    // others - the neighbourhood of the diagonal and of every revisit, about a tenth; every quad of a strip without boxes
    // - in the WORKGROUP's list, which its waves share out afterwards (a wave working off its own strip's took up to 30 us
    // after a 17 us stream: those quads cluster by strip).  Kept apart so that the streaming loop stays small in
    // registers; a full list ends the round for the wave that meets it (it resumes at that quad in the next one).
    auto push = [&](const int strip, const int quad) {
        unsigned at = 0u;
        if (lane == 0) at = atomicAdd(&nslow, 1u);
        at = __builtin_amdgcn_readfirstlane(at);
        if (at >= (unsigned)F1_SLOWQ) return false;
        if (lane == 0) {
            slow_strip[at] = strip;
            slow_quad[at] = quad;
        }
        return true;
    };
    long long vw = wave0;
    int quad = -1;                                        // (-1: the strip of vw is not begun)
    int more;
    do {
        bool full = false;
        while (vw < walk.nvirt && !full) {
            const int strip = __builtin_amdgcn_readfirstlane((int)(vw % walk.strips));
            if (quad < 0) quad = __builtin_amdgcn_readfirstlane((int)(vw / walk.strips));
            const int c0 = strip * 256 + 4 * lane;
            if (!(SGPR_F1_CULL && sc.truth.pose && strip * 256 + 256 <= sc.M)) {
                for (; quad < walk.nquads; quad += walk.groups)
                    if (!push(strip, quad)) {
                        full = true;
                        break;
                    }
            } else {
                StripBox box;
                {
                    ColPoses cp;
                    cp.load(sc.truth, c0, sc.M);
                    box.init(cp);
                }
                RowQuad<float> cur;
                cur.fetch(sc, quad, walk.nquads, c0, 0.f, true);
                while (quad < walk.nquads) {
                    RowQuad<float> nxt;
                    nxt.fetch(sc, quad + walk.groups, walk.nquads, c0, 0.f, true);
                    bool ok = true;
#pragma unroll
                    for (int u = 0; u < F1_ROWS; ++u) {
                        const unsigned mx = max(max(__float_as_uint(cur.s[u][0]), __float_as_uint(cur.s[u][1])),
                                                max(__float_as_uint(cur.s[u][2]), __float_as_uint(cur.s[u][3])));
                        // (the rows past the matrix' last one repeat its pose and carry the fill value 0)
                        ok = ok && box.far(cur.px[u], cur.pz[u], cut2) && mx <= 0x3F800000u;
                    }
                    if (__ballot(!ok) == 0ull) {
#pragma unroll
                        for (int u = 0; u < F1_ROWS; ++u) {
                            if (cur.r0 + u >= cur.r1) break;  // (wave-uniform)
                            cls_out[(size_t)(cur.r0 + u) * cw + (c0 >> 2)] = 0;
#pragma unroll
                            for (int q = 0; q < 4; ++q)
                                atomicAdd(reinterpret_cast<unsigned*>(reinterpret_cast<unsigned char*>(hist) + f1_key01_x4(cur.s[u][q])), 1u);
                        }
                    } else if (!push(strip, quad)) {
                        full = true;
                        break;
                    }
                    quad += walk.groups;
                    cur = nxt;
                }
            }
            if (!full) {
                vw += nwaves;
                quad = -1;
            }
        }
        F1A_STAMP(2)
        __syncthreads();                                  // the round's list is complete
        {
            const int n = (int)min(nslow, (unsigned)F1_SLOWQ);
            auto take = [&]() {
                unsigned i = 0u;
                if (lane == 0) i = atomicAdd(&ntaken, 1u);
                return (int)__builtin_amdgcn_readfirstlane(i);
            };
            int e = take();
            while (e < n) {
                const int strip = slow_strip[e], sq_quad = slow_quad[e];
                const int c0 = strip * 256 + 4 * lane;
                ColPoses cp;
                if (c0 < sc.M) cp.load(sc.truth, c0, sc.M);
                RowQuad<double> sq;
                sq.fetch(sc, sq_quad, walk.nquads, c0, 0.f, sc.truth.pose != nullptr);
                e = take();                               // (the next entry's number is on its way meanwhile)
                classic(sq, c0, cp);
            }
        }
        __syncthreads();                                  // the list is worked off
        if (tid == 0) nslow = ntaken = 0u;
        more = __syncthreads_or(vw < walk.nvirt ? 1 : 0);
    } while (more);
    F1A_STAMP(3)
    // what the waves still hold joins the list with ONE reservation per workgroup: a returning atomic per wave on the one
    // counter - 4096 of them, ~10 ns each, one after the other - was the last 20 us of this kernel
    if (lane == 0) wave_nst[tid >> 6] = nst;
    if (bad_pos) atomicAdd(&count[1], (unsigned long long)bad_pos);
    if (bad_neg) atomicAdd(&nbad_neg, bad_neg);
    __syncthreads();
    if (tid == 0) {
        unsigned total = 0u;
        for (int w = 0; w < F1_THREADS / 64; ++w) total += (unsigned)wave_nst[w];
        wg_base = total ? atomicAdd(&count[0], (unsigned long long)total) : 0ull;
    }
    __syncthreads();
    {
        long long at = (long long)wg_base;
        for (int w = 0; w < (tid >> 6); ++w) at += wave_nst[w];
        for (int i = lane; i < nst; i += 64)
            if (at + i < cap) pos[at + i] = buf[i];
    }
    F1A_STAMP(4)
    // (one 64-bit atomic per occupied bin straight into the sums instead of the slab: 54 -> 81 us for this kernel - a few
    // thousand device-scope atomics per workgroup cost more than the 25 MB of slabs and the kernel that adds them up)
    unsigned* slab = slabs + (size_t)blockIdx.x * slab_words;
    for (int i = tid; i < F1_NBP; i += F1_THREADS) {
        slab[i] = hist[i];
        hist[i] = 0u;
    }
    if (tid == 0) {
        reinterpret_cast<unsigned long long*>(slab + slab_words - 4)[0] = nbad_neg;
        reinterpret_cast<unsigned long long*>(slab + slab_words - 4)[1] = 0ull;
    }
    F1A_STAMP(5)
    // The positives by bin, for the plan kernel: the workgroup's are counted in the (now free) histogram first and leave
    // as one device-scope atomic per occupied bin - a sigmoid's positives crowd into a few bins, and one atomic per
    // positive pair on those few addresses was the last 10 - 15 us of this kernel.
    __syncthreads();
    for (int i = lane; i < nst; i += 64) atomicAdd(&hist[f1_key(buf[i])], 1u);
    __syncthreads();
    for (int i = tid; i < F1_NBP; i += F1_THREADS) {
        const unsigned n = hist[i];
        if (n) atomicAdd(&posb_g[f1_tslot(i)], n);
    }
    F1A_STAMP(6)
}
#if SGPR_F1_SCAN_STAMPS
extern "C" int sgpr_debug_f1_scan_stamps(unsigned long long* host_out) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(f1_scan_stamps), sizeof(f1_scan_stamps)) == hipSuccess ? 0 : -1;
}
#endif

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

// ascending bitonic sort of v[0 .. np2) in LDS by 16 waves that each own 1/16 of the array: every step whose partner
// distance stays inside a wave's block needs no workgroup barrier (the LDS serves a wave's accesses in order), only
// the steps that cross blocks do.  np2 = a power of two >= 64.
__device__ __forceinline__ void lds_bitonic_sort(float* v, int np2) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int ppw = np2 >> 5, blk = np2 >> 4;                              // pairs / elements per wave
    bool crossed = true;                                                   // (the caller's fill crossed the blocks)
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
}

// what pass B and the final step need per threshold q (ascending distinct positive values of the candidate bins)
struct F1Thr {
    unsigned long long tp;       // positive pairs with a score >= thr[q]            (exact)
    unsigned long long fp_above; // negative pairs in the bins above the threshold's bin
    int seg_end;                 // thresholds in bins <= the threshold's bin (= the largest bucket a negative of that bin can get)
    int pad;
};

constexpr int F1_HASH = 8192;                                    // slots of the de-duplication table (> F1_PICK + F1_THREADS)
// the plan kernel's first LDS area: the positives' histogram, then the edge values [F1_PER][F1_THREADS], then the hash table
// and the suffix sums
constexpr int f1_max3(int a, int b, int c) { return a > b ? (a > c ? a : c) : (b > c ? b : c); }
constexpr int F1_PLAN_WORDS = (f1_max3(F1_NBP, F1_PER * F1_THREADS, 2 * F1_HASH + F1_SORT + 1) + 3) & ~3;   // (16-byte reads of what follows)
__device__ __forceinline__ unsigned f1_hash(unsigned bits) { return (bits * 2654435761u) >> 19; }

// ---- P: one workgroup.  negb [F1_NBP + 2] (uint64, from slab_sum_kernel; [F1_NBP] = negatives with unusable scores).
//      Out: ctrl, mark bits [F1_NBP / 32 + 1], thr [T2], info [T2], dT2; tpge / fpge [F1_NB + 1] = pairs in bins >= b.
//      A single workgroup lives on latency: the bins' sums are scanned in registers (thread t owns 24 consecutive bins),
//      every pass over the positives keeps eight independent loads in flight.
__global__ __launch_bounds__(F1_THREADS) void f1_plan_kernel(const unsigned long long* __restrict__ negb, const float* __restrict__ pos,
                                                             const unsigned long long* __restrict__ count, long long cap,
                                                             unsigned long long* __restrict__ tpge, unsigned long long* __restrict__ fpge,
                                                             unsigned* __restrict__ mark_out, float* __restrict__ thr,
                                                             F1Thr* __restrict__ info, int* __restrict__ dT2, F1Ctrl* __restrict__ ctrl,
                                                             unsigned long long* __restrict__ stamps, const unsigned* __restrict__ posb_g,
                                                             const unsigned long long* __restrict__ negb_t) {
    extern __shared__ __attribute__((aligned(16))) unsigned char f1_smem[];
#define F1_STAMP(i) if (stamps && threadIdx.x == 0) stamps[i] = wall_clock64();
    F1_STAMP(0)
    unsigned* posb = reinterpret_cast<unsigned*>(f1_smem);                  // [F1_NBP] positives by bin; later:
    unsigned* hkey = posb;                                                  //   [F1_HASH] bit patterns of the distinct values
    unsigned* hcnt = posb + F1_HASH;                                        //   [F1_HASH] positive pairs that carry each
    unsigned* ssum = posb + 2 * F1_HASH;                                    //   [F1_SORT + 1] pairs of the sorted entries >= i
    float* v = reinterpret_cast<float*>(posb + F1_PLAN_WORDS);              // [F1_SORT] the values pass B settles
    unsigned* mark = reinterpret_cast<unsigned*>(v + F1_SORT);              // [F1_NBP / 32 + 1]
    __shared__ unsigned long long sh[2][17];
    __shared__ double shd[16];
    __shared__ unsigned ndist, nv, bspan[2];
    __shared__ int kspan[2];
    __shared__ unsigned long long tot[2];
    const int tid = threadIdx.x;
    const long long na = (long long)count[0];
    if (tid == 0) *dT2 = -1;
    if (na > cap) {                                                         // list longer than its buffer: the caller falls back
        if (tid == 0) ctrl->status = 1;
        return;
    }
    if (count[1] != 0ull || negb[F1_NBP] != 0ull) {                         // negative / NaN scores have no rank
        if (tid == 0) ctrl->status = 2;
        return;
    }
    constexpr int PER = F1_PER;                                             // 24 bins per thread
    static_assert(PER % 2 == 0, "bins per thread");
    for (int i = tid; i <= F1_NBP / 32; i += F1_THREADS) mark[i] = 0u;
    if (tid == 0) {
        ndist = nv = 0u;
        kspan[0] = F1_NBP;
        kspan[1] = -1;
    }
    // positives by bin: counted by pass A where it found them (one global atomic per positive pair; round 4 spent 14 us
    // of this single workgroup on a pass over the list with LDS atomics); negatives by bin: slab_sum_kernel's sums.  Both
    // arrive transposed ([q][thread]): 48 coalesced loads per thread (in the bins' own order - 96 / 192 bytes between
    // neighbouring lanes - the same loads took 8 us)
    unsigned ps[PER];
    unsigned long long ng[PER];
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        const int b = tid * PER + q;
        ps[q] = posb_g[q * F1_THREADS + tid];
        const unsigned long long x = negb_t[q * F1_THREADS + tid];
        ng[q] = b < F1_NB ? x : 0ull;
    }
    __syncthreads();
    F1_STAMP(1)
    unsigned long long sn = 0ull, sp = 0ull;
    bool any = false;
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        const int b = tid * PER + q;
        if (b >= F1_NB) ps[q] = 0u;
        sn += ng[q];
        sp += ps[q];
        any = any || ps[q] != 0u || ng[q] != 0ull;
    }
    unsigned long long an = sn, ap = sp;
    suffix_scan2(an, ap, sh);                                              // pairs in the bins of the threads above
    if (tid == 0) {
        tot[0] = ap + sp;                                                  // all positives / negatives
        tot[1] = an + sn;
    }
    __syncthreads();
    const double P = (double)tot[0];
    // F1 = 2 TP / (TP + FP + P) is monotone in g = TP / (TP + FP + P).  Which edge is best and which bins can beat it is
    // decided on g in fp32 (one division per occupied bin; f1_of's three float64 divisions per bin and test were 40 us of
    // this single workgroup) with a margin of 1e-5 - far above fp32 rounding, so the exact maximum is never excluded -,
    // and the exact float64 F1 is evaluated only at the few edges inside the margin.
    const float Pf = (float)tot[0];
    auto g32 = [&](unsigned long long tp, unsigned long long fp) {
        const float a = (float)tp;
        return a > 0.f ? a * __builtin_amdgcn_rcpf(a + (float)fp + Pf) : 0.f;     // (1 ulp: the margin below is 1e-5)
    };
    float gbest = 0.f;
    float* gkeep = reinterpret_cast<float*>(posb);       // [PER][F1_THREADS] edge values of the first sweep (this thread's own)
    {
        unsigned long long rn = an, rp = ap;
#pragma unroll
        for (int q = PER - 1; q >= 0; --q) {
            rn += ng[q];
            rp += ps[q];
            if (ps[q] != 0u || ng[q] != 0ull) {           // (an empty bin repeats the edge above it)
                const float gq = g32(rp, rn);
                gkeep[q * F1_THREADS + tid] = gq;         // (kept for the second sweep: the positives' histogram area is idle)
                gbest = fmaxf(gbest, gq);
            }
        }
    }
    gbest = (float)block_max((double)gbest, shd);
    const float gcut = gbest * (1.f - 1e-5f);
    // exact F1 at the edges that can be the best one = the curve's point at the smallest score >= the edge
    double best = 0.0;
    {
        unsigned long long rn = an, rp = ap;
#pragma unroll
        for (int q = PER - 1; q >= 0; --q) {
            rn += ng[q];
            rp += ps[q];
            if (ps[q] != 0u || ng[q] != 0ull) {
                const float gq = gkeep[q * F1_THREADS + tid];
                if (gq > 0.f && gq >= gcut) best = fmax(best, f1_of((double)rp, (double)rn, P));
            }
        }
    }
    best = block_max(best, shd);
    // candidate bins: some positive inside could beat the best edge value (at most every positive from the bin's lower
    // edge on, at least the negatives above its upper edge); pairs from every bin edge on -> tpge / fpge
    {
        unsigned long long rn = an, rp = ap;
#pragma unroll
        for (int q = PER - 1; q >= 0; --q) {
            const int b = tid * PER + q;
            const unsigned long long fp_above = rn, tp_above = rp;
            rn += ng[q];
            rp += ps[q];
            if (ps[q] != 0u && g32(rp, fp_above) >= gcut) {
                atomicOr(&mark[b >> 5], 1u << (b & 31));
                atomicMin(&kspan[0], b);                 // (a few dozen bins at most)
                atomicMax(&kspan[1], b);
                tpge[b + 1] = tp_above;                  // (only what the thresholds of this bin will need: a store per bin
                fpge[b + 1] = fp_above;                  //  from 1024 threads in 192-byte strides was 20 us of this kernel)
            }
        }
    }
    (void)any;
    __syncthreads();                                     // marks complete; posb is dead from here on (-> hash table)
    F1_STAMP(2)
    for (int i = tid; i <= F1_NBP / 32; i += F1_THREADS) mark_out[i] = mark[i];
    for (int i = tid; i < 2 * F1_HASH; i += F1_THREADS) posb[i] = i < F1_HASH ? 0xffffffffu : 0u;   // empty = a NaN pattern
    // the candidate bins' span as bit patterns [first pattern of the lowest bin, first pattern past the highest): one
    // positive in a hundred lies inside, the others are turned away by two comparisons instead of key + bit test
    if (tid < 128) {                                                       // (waves 0 and 1)
        const unsigned edge = f1_first_pattern(tid < 64 ? kspan[0] : kspan[1] + 1);
        if ((tid & 63) == 0) bspan[tid >> 6] = edge;
    }
    __syncthreads();
    const unsigned b_lo = bspan[0], b_hi = bspan[1], b_width = b_hi > b_lo ? b_hi - b_lo : 0u;   // (no candidate bin: empty)
    // ---- the positives of the candidate bins: distinct values (and the pairs that carry each) through a hash table
    auto insert = [&](float x) {
        const unsigned bits = __float_as_uint(x);
        {
            const int b = f1_key(x);                                           // (inside the span; in a candidate bin?)
            if (!((mark[b >> 5] >> (b & 31)) & 1u)) return;
        }
        unsigned hslot = f1_hash(bits);
        // (more distinct values than pass B settles: stop - the table never fills, at most one more value per thread)
        while (__hip_atomic_load(&ndist, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) <= (unsigned)F1_PICK) {
            const unsigned old = atomicCAS(&hkey[hslot], 0xffffffffu, bits);
            if (old == 0xffffffffu) atomicAdd(&ndist, 1u);
            if (old == 0xffffffffu || old == bits) {
                atomicAdd(&hcnt[hslot], 1u);
                break;
            }
            hslot = (hslot + 1) & (F1_HASH - 1);
        }
    };
    // A few percent of the positives sit in candidate bins - about one lane in every other wave step, and an insertion
    // (two dependent LDS atomics) issued for one lane costs what it costs for sixty-four.  So a wave queues its candidates
    // (128 slots of the idle sort buffer per wave) and inserts them sixty-four at a time: 17 - 22 us of this phase were
    // ~800 nearly empty insertion rounds.
    {
        const int lane = tid & 63;
        float* wq = v + (tid >> 6) * 128;
        const unsigned long long lt_mask = (1ull << lane) - 1ull;
        int nq = 0;                                                            // queued, wave-uniform
        for (long long i0 = 0; i0 < na; i0 += 16 * F1_THREADS) {
            float x[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const long long i = i0 + u * F1_THREADS + tid;
                x[u] = i < na ? pos[i] : -1.f;
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                // (queued on the span alone - one subtraction, one comparison; key and candidate-bin test wait for the
                //  insertion, sixty-four values at a time: this loop is one CU's vector pipe, 35 instructions per value
                //  and wave were 7 us of it.  -1.f, the filler: beyond b_hi)
                const bool hit = __float_as_uint(x[u]) - b_lo < b_width;
                const unsigned long long m = __ballot(hit);
                if (m) {                                                       // (wave-uniform)
                    if (hit) wq[nq + __popcll(m & lt_mask)] = x[u];
                    nq += __popcll(m);
                    if (nq >= 64) {
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");     // (program order inside the wave is all it takes)
                        __builtin_amdgcn_wave_barrier();
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                        insert(wq[nq - 64 + lane]);                            // the last 64 queued
                        nq -= 64;
                    }
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (lane < nq) insert(wq[lane]);
    }
    __syncthreads();
    F1_STAMP(3)
    const unsigned n2 = ndist;
    if (tid == 0) {
        ctrl->best1 = best;
        ctrl->P = tot[0];
        ctrl->N = tot[1];
        ctrl->n2 = (int)n2;
        if (n2 > (unsigned)F1_PICK) ctrl->status = 1;                      // a flat curve: too many values to settle in one pass
    }
    if (n2 == 0u || n2 > (unsigned)F1_PICK) return;
    int np2 = 64;
    while (np2 < (int)n2) np2 <<= 1;
    for (int i = tid; i < np2; i += F1_THREADS) v[i] = INFINITY;
    __syncthreads();
    {
        // compaction of the table's occupied slots: one LDS atomic per WAVE (a returning atomic on one address per value
        // serialises the whole workgroup behind the LDS atomic unit: ~400 of them were a third of this phase)
        constexpr int SLOTS = F1_HASH / F1_THREADS;                            // 8 slots per thread
        const int lane = tid & 63;
        const unsigned long long lt_mask = (1ull << lane) - 1ull;
        unsigned keys[SLOTS];
        int total = 0;
#pragma unroll
        for (int q = 0; q < SLOTS; ++q) {
            keys[q] = hkey[q * F1_THREADS + tid];
            total += __popcll(__ballot(keys[q] != 0xffffffffu));
        }
        unsigned base = 0u;
        if (lane == 0 && total) base = atomicAdd(&nv, (unsigned)total);
        base = __shfl(base, 0);
#pragma unroll
        for (int q = 0; q < SLOTS; ++q) {
            const bool have = keys[q] != 0xffffffffu;
            const unsigned long long m = __ballot(have);
            if (have) v[base + __popcll(m & lt_mask)] = __uint_as_float(keys[q]);
            base += __popcll(m);
        }
    }
    if (np2 <= F1_THREADS) {
        // up to 1024 values (the usual case: a few hundred): every value is distinct, so its rank - the number of smaller
        // ones, counted against the whole array with broadcast 16-byte LDS reads - is its place.  Two barriers instead of
        // the bitonic network's 45 dependent steps (12.5 -> ~4 us of this phase).
        __syncthreads();
        // (up to 512 values: two threads per value, half the array each)
        const int per = np2 <= F1_THREADS / 2 ? 2 : 1, e = per == 2 ? tid >> 1 : tid, part = per == 2 ? tid & 1 : 0;
        const int half = per == 2 ? (((int)n2 + 7) >> 3) << 2 : (int)n2;   // a multiple of 4
        const float mine = e < (int)n2 ? v[e] : INFINITY;
        int rank = 0;
        for (int j = part * half; j < min((int)n2, (part + 1) * half); j += 4) {   // (v is padded with +inf up to np2 >= n2 rounded up to 4)
            const float4 w = *reinterpret_cast<const float4*>(v + j);
            rank += (w.x < mine ? 1 : 0) + (w.y < mine ? 1 : 0) + (w.z < mine ? 1 : 0) + (w.w < mine ? 1 : 0);
        }
        if (per == 2) rank += __shfl_xor(rank, 1);
        __syncthreads();
        if (e < (int)n2 && part == 0) v[rank] = mine;
        __syncthreads();
    } else {
        lds_bitonic_sort(v, np2);
    }
    F1_STAMP(4)
    // ---- pairs of the sorted entries >= i (suffix sums of the multiplicities)
    constexpr int PT = F1_SORT / F1_THREADS;                               // 4 consecutive entries per thread
    unsigned mult[PT];
    unsigned long long run = 0ull, dummy = 0ull;
#pragma unroll
    for (int q = PT - 1; q >= 0; --q) {
        const int i = tid * PT + q;
        unsigned c = 0u;
        if (i < (int)n2) {
            const unsigned bits = __float_as_uint(v[i]);
            unsigned hslot = f1_hash(bits);
            while (hkey[hslot] != bits) hslot = (hslot + 1) & (F1_HASH - 1);
            c = hcnt[hslot];
        }
        run += c;
        mult[q] = (unsigned)run;                                           // entries i .. end of this thread's four
    }
    unsigned long long above = run;
    suffix_scan2(above, dummy, sh);
#pragma unroll
    for (int q = 0; q < PT; ++q) ssum[tid * PT + q] = mult[q] + (unsigned)above;
    if (tid == 0) ssum[F1_SORT] = 0u;
    __syncthreads();
    // (entry i = tid, tid + 1024, ...: the few hundred entries of a typical matrix spread over as many threads, so that
    // the two global reads per entry - written by this workgroup a few microseconds ago - are ONE round trip, requested
    // before the bisection, not four in a row inside the first hundred threads: 8.0 -> 4.5 us of this phase)
#pragma unroll 1
    for (int i = tid; i < (int)n2; i += F1_THREADS) {
        const float s = v[i];
        const int b = f1_key(s);
        const unsigned long long tp_bin = tpge[b + 1], fp_bin = fpge[b + 1];
        // the end of this bin's segment of the sorted list: first entry of a later bin (bisection on the monotone key)
        int lo = i + 1, hi = (int)n2;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (f1_key(v[mid]) <= b) lo = mid + 1; else hi = mid;
        }
        thr[i] = s;
        F1Thr e;
        e.tp = tp_bin + (unsigned long long)(ssum[i] - ssum[lo]);
        e.fp_above = fp_bin;
        e.seg_end = lo;
        e.pad = 0;
        info[i] = e;
    }
    if (tid == 0) {
        ctrl->T2 = (int)n2;
        *dT2 = (int)n2;
    }
    F1_STAMP(5)
#undef F1_STAMP
}

// ---- F: the closing step, by the workgroup of pass B that finishes last (a launch of its own: 5.4 us).
//      negc2[b] (uint64) = negatives of the candidate bins with exactly b thresholds <= their score - sums that device-scope
//      atomics of this launch built: read past this XCD's L2.  G [F1_SORT + 2], sh [2][17], shd [16]: LDS of the caller.
__device__ __forceinline__ void f1_final(const unsigned long long* __restrict__ negc2, const F1Thr* __restrict__ info,
                                         const F1Ctrl* __restrict__ ctrl, double* __restrict__ result, unsigned long long* G,
                                         unsigned long long (*sh)[17], double* shd) {
    const int tid = threadIdx.x;
    double best = ctrl->best1;
    int passes = 1;
    if (ctrl->status == 0 && ctrl->n2 > 0) {
        const int T = ctrl->T2;
        constexpr int PER = (F1_SORT + 1 + F1_THREADS - 1) / F1_THREADS;   // 5
        unsigned long long part[PER], sn = 0ull, dummy = 0ull;
        for (int q = PER - 1; q >= 0; --q) {
            const int b = tid * PER + q;
            sn += b <= T ? __hip_atomic_load(&negc2[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
            part[q] = sn;
        }
        unsigned long long an = sn;
        suffix_scan2(an, dummy, sh);
        for (int q = 0; q < PER; ++q) {
            const int b = tid * PER + q;
            if (b <= F1_SORT + 1) G[b] = b <= T ? part[q] + an : 0ull;
        }
        __syncthreads();
        const double P = (double)ctrl->P;
        double b2 = 0.0;
        for (int q = tid; q < T; q += F1_THREADS) {
            const F1Thr e = info[q];
            // negatives >= thr[q]: those above the bin + those of the bin with a bucket in (q, seg_end]
            const unsigned long long fp = e.fp_above + (G[q + 1] - G[e.seg_end + 1]);
            b2 = fmax(b2, f1_of((double)e.tp, (double)fp, P));
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
        result[5] = (double)F1_NB;
        result[6] = (double)ctrl->n2;
        result[7] = 0.0;
    }
}

// ---- B: negatives of the candidate bins by threshold bucket b = #{thr <= s}; everything else costs one bit test.
//      Same strips as pass A; a pair's class comes from the byte pass A stored (no pose arithmetic here).
__global__ __launch_bounds__(F1_THREADS) void f1_refine_kernel(const PairScan sc, const unsigned* __restrict__ mark_in,
                                                               const float* __restrict__ thr_in, const int* __restrict__ dT2,
                                                               unsigned long long* __restrict__ neg2,
                                                               const unsigned char* __restrict__ cls_in, const F1Thr* __restrict__ info,
                                                               const F1Ctrl* __restrict__ ctrl, unsigned* __restrict__ done,
                                                               double* __restrict__ result) {
    __shared__ unsigned mark[F1_NBP / 32 + 1];
    __shared__ float thr[F1_SORT];
    // (one area: the counters and the waves' queues during the pass, the closing step's suffix sums after it)
    constexpr int QW = (F1_THREADS / 64) * 128;
    __shared__ __attribute__((aligned(16))) unsigned area[F1_SORT + 4 + 3 * QW];
    static_assert(sizeof(unsigned) * (F1_SORT + 4 + 3 * QW) >= sizeof(unsigned long long) * (F1_SORT + 2), "the closing step's sums");
    unsigned* cnt = area;                                                   // [F1_SORT + 1]
    float* queue_x = reinterpret_cast<float*>(area + F1_SORT + 4);
    int* queue_r = reinterpret_cast<int*>(area + F1_SORT + 4 + QW);
    int* queue_c = reinterpret_cast<int*>(area + F1_SORT + 4 + 2 * QW);
    __shared__ unsigned long long fsh[2][17];
    __shared__ double fshd[16];
    __shared__ unsigned last;
    const int T = *dT2;
    if (T < 0) {                                                            // no second pass: the closing step alone
        if (blockIdx.x == 0) f1_final(neg2, info, ctrl, result, reinterpret_cast<unsigned long long*>(area), fsh, fshd);
        return;
    }
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i <= F1_NBP / 32; i += F1_THREADS) mark[i] = mark_in[i];
    for (int i = tid; i < F1_SORT; i += F1_THREADS) thr[i] = i < T ? thr_in[i] : INFINITY;
    for (int i = tid; i <= F1_SORT; i += F1_THREADS) cnt[i] = 0u;
    // Only a score in [thr[0], the upper edge of thr[T - 1]'s bin) can be counted below (bit patterns of non-negative
    // floats order like the floats; the key is monotone: the edge by bisection on the pattern): a row of a strip with no
    // such score - almost every one when the candidate bins are few - costs two 3-input extrema and two comparisons.
    __shared__ unsigned range[2];
    if (tid < 64) {
        const unsigned edge = f1_first_pattern(f1_key(thr_in[T - 1]) + 1);   // first pattern with a larger key
        if (tid == 0) {
            range[0] = __float_as_uint(thr_in[0]);
            range[1] = edge;
        }
    }
    __syncthreads();
    const unsigned b_lo = SGPR_F1_CULL ? range[0] : 0u, b_hi = SGPR_F1_CULL ? range[1] : 0xffffffffu;
    const size_t cw = (size_t)(sc.M + 3) >> 2;            // class bytes per row
    const long long wave0 = (long long)blockIdx.x * (F1_THREADS / 64) + (tid >> 6), nwaves = (long long)gridDim.x * (F1_THREADS / 64);
    const StripWalk walk(sc, nwaves);
    // A pair inside the range (half a percent of a KITTI-00 matrix, yet some lane of a wave holds one in most rows) is
    // QUEUED by its wave - score, row, column - and the queue is settled sixty-four entries at a time with every lane
    // busy: class byte, key, candidate-bin test, bisection.  A pair outside costs two comparisons.
    float* qx = queue_x + (tid >> 6) * 128;
    int* qr = queue_r + (tid >> 6) * 128;
    int* qc = queue_c + (tid >> 6) * 128;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    int nq = 0;                                           // queued, wave-uniform
    auto wave_sync = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");     // (program order inside the wave is all it takes)
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    auto settle = [&](const int e) {
        const float x = qx[e];
        const int r = qr[e], c = qc[e];
        const unsigned cb = cls_in[(size_t)r * cw + (c >> 2)];
        if (((cb >> (2 * (c & 3))) & 3u) != 0u) return;                // pass A's class: not a negative
        if (__float_as_uint(x) > 0x7f800000u) return;
        const int b = f1_key(x);
        if (!((mark[b >> 5] >> (b & 31)) & 1u)) return;                // not in a candidate bin: nothing to settle
        int lo = 0, hi = T;                                            // thr[lo-1] <= x < thr[hi]
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (thr[mid] <= x) lo = mid + 1; else hi = mid;
        }
        // counted only when it is >= a threshold OF ITS OWN BIN: a negative below every threshold of its bin belongs to
        // no threshold there (the bins above a threshold's bin are accounted through fp_above), and bucket lo then names
        // the bin of thr[lo - 1] unambiguously
        // (one device-scope atomic per candidate straight into the sums instead - ~100 000 of them on a few hundred
        // addresses - took this kernel from 27 to 60 us)
        if (lo > 0 && f1_key(thr[lo - 1]) == b) atomicAdd(&cnt[lo], 1u);
    };
    for (long long vw = wave0; vw < walk.nvirt; vw += nwaves) {
        const int strip = __builtin_amdgcn_readfirstlane((int)(vw % walk.strips)), g0 = __builtin_amdgcn_readfirstlane((int)(vw / walk.strips));
        const int c0 = strip * 256 + 4 * lane;
        RowQuad<float> cur;
        cur.fetch(sc, g0, walk.nquads, c0, -1.f, false);
        for (int quad = g0; quad < walk.nquads; quad += walk.groups) {
            RowQuad<float> nxt;
            nxt.fetch(sc, quad + walk.groups, walk.nquads, c0, -1.f, false);
#pragma unroll
            for (int u = 0; u < F1_ROWS; ++u) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const unsigned b = __float_as_uint(cur.s[u][q]);   // (the fill value -1.f: its pattern is >= b_hi)
                    const bool in = b >= b_lo && b < b_hi && c0 + q < sc.M;
                    const unsigned long long m = __ballot(in);
                    if (m) {                                           // (wave-uniform)
                        if (in) {
                            const int at = nq + __popcll(m & lt_mask);
                            qx[at] = cur.s[u][q];
                            qr[at] = cur.r0 + u;
                            qc[at] = c0 + q;
                        }
                        nq += __popcll(m);
                        if (nq >= 64) {
                            wave_sync();
                            settle(nq - 64 + lane);                    // the last 64 queued
                            wave_sync();
                            nq -= 64;
                        }
                    }
                }
            }
            cur = nxt;
        }
    }
    wave_sync();
    if (lane < nq) settle(lane);
    __syncthreads();
    // the few hundred occupied buckets of the workgroup: one 64-bit atomic each straight into the sums (slabs and the
    // kernel that added them up: 21.4 + 10 us against 26.9)
    for (int i = tid; i <= T; i += F1_THREADS) {
        const unsigned n = cnt[i];
        if (n) atomicAdd(&neg2[i], (unsigned long long)n);
    }
    // This workgroup's sums are out before it is counted as done: they are device-scope atomics (performed past the L2),
    // so waiting for their acknowledgements is all it takes - an agent-scope release fence would also write this XCD's
    // L2 back (24 -> 81 us for this kernel).
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (tid == 0) last = atomicAdd(done, 1u) == gridDim.x - 1 ? 1u : 0u;
    __syncthreads();
    if (last) f1_final(neg2, info, ctrl, result, reinterpret_cast<unsigned long long*>(area), fsh, fshd);
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
    static LdsLimitOnce once;
    if (int rc = raise_lds_limit(&once, reinterpret_cast<const void*>(&pair_threshold_count_kernel), 144 * 1024,   // (+ 16 B of static LDS)
                                 "pair_threshold_count_kernel"))
        return rc;
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

// ---- sgpr_f1_max: workspace = header (counts, control block, device-side sizes) | negatives by bin | pairs from a bin on
//      (two arrays) | candidate-bin marks | thresholds, their info, negatives by bucket of pass B | positives | counter slabs
static size_t a256(size_t v) { return (v + 255) & ~(size_t)255; }
struct F1Layout {
    size_t off_posb, off_negb, off_negbt, off_tpge, off_fpge, off_mark, off_thr, off_info, off_neg2, off_pos, off_slabs, off_cls, total;
    long long cap;
    int slabs_a, slabs_b, words_a, words_b;
};
static F1Layout f1_layout(const sgpr_handle* h, int R, int M) {
    F1Layout L;
    const long long pairs = (long long)R * M;
    L.cap = pairs < (1LL << 20) ? pairs : (1LL << 20);
    if (L.cap < 1) L.cap = 1;
    L.slabs_a = h->num_cus;                       // pass A: one 1024-thread workgroup (97 KB histogram) per CU
    L.slabs_b = h->num_cus * SGPR_F1_WGB;                       // pass B: likewise (4096 waves, ~5 strip tasks each on a KITTI-00 matrix)
    L.words_a = F1_NBP + 4;
    L.words_b = slab_words(F1_SORT);
    size_t off = 256;                                                       // header
    L.off_posb = off;  off += a256((size_t)F1_THREADS * ((F1_NBP + F1_THREADS - 1) / F1_THREADS) * sizeof(unsigned));   // (zeroed with the header)
    L.off_negb = off;  off += a256((F1_NBP + 4) * sizeof(unsigned long long));
    L.off_negbt = off; off += a256((size_t)F1_THREADS * F1_PER * sizeof(unsigned long long));     // negatives by bin, transposed
    L.off_neg2 = off;  off += a256((F1_SORT + 4) * sizeof(unsigned long long));                   // (zeroed up to here)
    L.off_tpge = off;  off += a256((F1_NBP + 4) * sizeof(unsigned long long));
    L.off_fpge = off;  off += a256((F1_NBP + 4) * sizeof(unsigned long long));
    L.off_mark = off;  off += a256((F1_NBP / 32 + 1) * sizeof(unsigned));
    L.off_thr = off;   off += a256(F1_SORT * sizeof(float));
    L.off_info = off;  off += a256(F1_SORT * sizeof(F1Thr));
    L.off_pos = off;   off += a256((size_t)L.cap * sizeof(float));
    L.off_slabs = off; off += a256((size_t)L.slabs_a * L.words_a * sizeof(unsigned));   // pass A's histograms (pass B: atomics)
    L.off_cls = off;   off += a256((size_t)R * (((size_t)M + 3) >> 2));     // one class byte per four pairs (pass A -> pass B)
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
    int* dT2 = reinterpret_cast<int*>(ws + 192);
    unsigned* posb = reinterpret_cast<unsigned*>(ws + L.off_posb);
    unsigned long long* negb = reinterpret_cast<unsigned long long*>(ws + L.off_negb);
    unsigned long long* negb_t = reinterpret_cast<unsigned long long*>(ws + L.off_negbt);
    unsigned long long* tpge = reinterpret_cast<unsigned long long*>(ws + L.off_tpge);
    unsigned long long* fpge = reinterpret_cast<unsigned long long*>(ws + L.off_fpge);
    unsigned* mark = reinterpret_cast<unsigned*>(ws + L.off_mark);
    float* thr = reinterpret_cast<float*>(ws + L.off_thr);
    F1Thr* info = reinterpret_cast<F1Thr*>(ws + L.off_info);
    unsigned long long* neg2 = reinterpret_cast<unsigned long long*>(ws + L.off_neg2);
    float* pos = reinterpret_cast<float*>(ws + L.off_pos);
    unsigned* slabs = reinterpret_cast<unsigned*>(ws + L.off_slabs);
    unsigned char* cls = ws + L.off_cls;
    static_assert(sizeof(F1Ctrl) <= 128, "control block");
    const size_t lds_scan = (size_t)F1_NBP * sizeof(unsigned) + (size_t)F1_PBUF * sizeof(float);
    const size_t lds_plan = (size_t)F1_PLAN_WORDS * sizeof(unsigned) + (size_t)F1_SORT * sizeof(float) +
                            (size_t)(F1_NBP / 32 + 1) * sizeof(unsigned);
    static LdsLimitOnce once_scan, once_plan;
    if (int rc = raise_lds_limit(&once_scan, reinterpret_cast<const void*>(&f1_scan_kernel), (int)lds_scan, "sgpr_f1_max (f1_scan_kernel)"))
        return rc;
    if (int rc = raise_lds_limit(&once_plan, reinterpret_cast<const void*>(&f1_plan_kernel), (int)lds_plan, "sgpr_f1_max (f1_plan_kernel)"))
        return rc;
    hipError_t e = hipMemsetAsync(ws, 0, L.off_tpge, s);       // counts, control block, sizes, every sum the passes add to
    if (e != hipSuccess) return hip_fail(e, "sgpr_f1_max: memset");
    if ((int64_t)R * M == 0) {                       // an empty rectangle: F1-max 0 over 0 positive / 0 negative pairs, status 0
        e = hipMemsetAsync(d_result, 0, 8 * sizeof(double), s);
        if (e != hipSuccess) return hip_fail(e, "sgpr_f1_max: memset");
        return SGPR_OK;
    }
    const PairScan sc = make_scan(d_score, R, M, ld, row0, d_pose_xz, d_pos, d_neg, d_gt, ldg);
    hipLaunchKernelGGL(f1_scan_kernel, dim3(L.slabs_a), dim3(F1_THREADS), lds_scan, s, sc, slabs, L.words_a, pos, L.cap, count, cls, posb);
    hipLaunchKernelGGL(slab_sum_kernel, dim3((F1_NBP + 2 + 31) / 32), dim3(1024), 0, s, slabs, L.slabs_a, L.words_a, F1_NBP - 1, negb, nullptr, negb_t, F1_PER);
    hipLaunchKernelGGL(f1_plan_kernel, dim3(1), dim3(F1_THREADS), lds_plan, s, negb, pos, count, L.cap, tpge, fpge, mark, thr, info, dT2,
                       ctrl, reinterpret_cast<unsigned long long*>(ws + 128), posb, negb_t);
    hipLaunchKernelGGL(f1_refine_kernel, dim3(L.slabs_b), dim3(F1_THREADS), 0, s, sc, mark, thr, dT2, neg2, cls, info, ctrl,
                       reinterpret_cast<unsigned*>(ws + 200), d_result);
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
