// Consumers of the all-pairs score matrix that keep it on the device:
//
//  * sgpr_pair_histogram  - the counting half of eval_batch.py:69-87 (sklearn precision_recall_curve + F1 max):
//                           class-wise radix histograms of the scores, ground truth taken from the KITTI poses on
//                           the fly (utils.py:36, sg_net.py:302-309) or from explicit labels.  The host refines the
//                           few bins that can still hold the F1 maximum (sg_pr_amd/metrics.py:f1_max_device), so the
//                           82 MB matrix is streamed two or three times at HBM speed instead of being copied to the
//                           host and sorted there.  HBM-bound integer work: coalesced 16-B reads, LDS-privatised
//                           counters, one slab per workgroup, a second kernel sums the slabs - no global atomics.
//  * sgpr_topk_rows       - loop-closure candidates: for every query row the K best-scoring columns outside a temporal
//                           exclusion window, deterministic (score descending, column ascending).
#include <math.h>
#include <string.h>

#include <string>

#include "sgpr_internal.hpp"

namespace sgpr {

constexpr int HB_THREADS = 1024;
constexpr int HB_MAX_PREFIX = 4;          // candidate prefixes per pass
constexpr int HB_MAX_BITS = 12;           // bins per prefix = 2^bits <= 4096  ->  4 x 4096 x 2 x 4 B = 128 KB of LDS

struct HistArgs {
    const float* score;
    int R, M;
    int64_t ld;
    int row0;                 // global index of row 0 (rows are a shard of the square matrix)
    const double* pose;       // [>= row0 + R and >= M][2] planar pose (x, z) or NULL; float64 like the reference
    double d_pos, d_neg;      // positive if distance <= d_pos, negative if >= d_neg, ignored in between
    const signed char* gt;    // optional explicit labels [R][ldg]: 1 / 0 / negative = ignore (used when pose == NULL)
    int64_t ldg;
    int n_prefix, prefix_bits, bits;
    unsigned prefix[HB_MAX_PREFIX];
    unsigned* slabs;          // [gridDim.x][n_prefix << bits][2]
    unsigned* bad;            // counts scores that are negative or NaN (their order is undefined)
};

// key of a non-negative float = its bit pattern (monotone); the pass looks at `bits` bits below `prefix_bits`
__global__ __launch_bounds__(HB_THREADS) void pair_histogram_kernel(const HistArgs a) {
    extern __shared__ unsigned hist[];                  // [n_prefix << bits][2]
    const int nb = (a.n_prefix << a.bits) * 2;
    for (int i = threadIdx.x; i < nb; i += HB_THREADS) hist[i] = 0u;
    __syncthreads();
    const int shift_p = 32 - a.prefix_bits, shift_b = 32 - a.prefix_bits - a.bits;
    const unsigned bmask = (1u << a.bits) - 1u;
    // items = (row, group of 4 columns); a workgroup walks them with a grid stride, lanes along the row
    const int gpr = (a.M + 3) >> 2;
    const int64_t items = (int64_t)a.R * gpr;
    unsigned nbad = 0;
    const double lo2 = a.d_pos * a.d_pos, hi2 = a.d_neg * a.d_neg;
    for (int64_t it = (int64_t)blockIdx.x * HB_THREADS + threadIdx.x; it < items; it += (int64_t)gridDim.x * HB_THREADS) {
        const int r = (int)(it / gpr), c0 = (int)(it - (int64_t)r * gpr) * 4;
        const float* sp = a.score + (int64_t)r * a.ld + c0;
        float s[4];
        const bool vec = (c0 + 3 < a.M) && ((reinterpret_cast<uintptr_t>(sp) & 15) == 0);
        if (vec) {
            const float4 v = *reinterpret_cast<const float4*>(sp);
            s[0] = v.x; s[1] = v.y; s[2] = v.z; s[3] = v.w;
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) s[q] = c0 + q < a.M ? sp[q] : 0.f;
        }
        double px = 0.0, pz = 0.0;
        if (a.pose) {
            px = a.pose[2 * (a.row0 + r)];
            pz = a.pose[2 * (a.row0 + r) + 1];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const bool inb = c0 + q < a.M;             // (no early exit: the wave counts together below)
            const int c = inb ? c0 + q : a.M - 1;
            int cls;                                   // 1 positive, 0 negative, -1 ignored
            if (a.pose) {
                // utils.py:36 in float64, operation by operation (no fused multiply-add)
                const double dx = px - a.pose[2 * c], dz = pz - a.pose[2 * c + 1];
                const double s2 = __dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dz, dz));
                // sqrt is monotone and correctly rounded: away from the two thresholds the squared distance decides;
                // only within a relative 1e-12 of them is the reference's `sqrt(...) <= t` evaluated literally
                if (s2 < lo2 * (1.0 - 1e-12)) cls = 1;
                else if (s2 > lo2 * (1.0 + 1e-12) && s2 < hi2 * (1.0 - 1e-12)) cls = -1;
                else if (s2 > hi2 * (1.0 + 1e-12)) cls = 0;
                else {
                    const double d = sqrt(s2);
                    cls = d <= a.d_pos ? 1 : (d >= a.d_neg ? 0 : -1);
                }
            } else {
                const int g = a.gt[(int64_t)r * a.ldg + c];
                cls = g < 0 ? -1 : (g != 0);
            }
            if (!inb) cls = -1;
            int idx = -1;                              // the counter this element increments (-1: none)
            if (cls >= 0) {
                const unsigned key = __float_as_uint(s[q]);
                if (key > 0x7f800000u) {               // negative or NaN
                    ++nbad;
                } else {
                    const unsigned pre = a.prefix_bits ? key >> shift_p : 0u;
                    int slot = -1;
#pragma unroll
                    for (int p = 0; p < HB_MAX_PREFIX; ++p)
                        if (p < a.n_prefix && pre == a.prefix[p]) slot = p;
                    if (slot >= 0) idx = (int)(((((unsigned)slot << a.bits) + ((key >> shift_b) & bmask)) * 2) + cls);
                }
            }
            // wave-aggregated counting: sigmoid scores pile up in a few bins (next to 0 and 1), and 64 lanes hitting one
            // LDS counter serialise.  Up to three rounds take the most common counter of the wave with ONE atomic
            // (leader = the first lane still waiting); whoever is left counts on its own.
            unsigned long long todo = __ballot(idx >= 0);
#pragma unroll 1
            for (int round = 0; round < 3 && todo; ++round) {
                const int leader = __ffsll((long long)todo) - 1;
                const int lidx = __shfl(idx, leader);
                const unsigned long long same = __ballot(idx == lidx);
                if ((int)(threadIdx.x & 63) == leader) atomicAdd(&hist[lidx], (unsigned)__popcll(same));
                if (idx == lidx) idx = -1;
                todo &= ~same;
            }
            if (idx >= 0) atomicAdd(&hist[idx], 1u);
        }
    }
    if (nbad) atomicAdd(a.bad, nbad);
    __syncthreads();
    unsigned* slab = a.slabs + (size_t)blockIdx.x * nb;
    for (int i = threadIdx.x; i < nb; i += HB_THREADS) slab[i] = hist[i];
}

// 64 counters per workgroup, 4 threads per counter (each sums every 4th slab), 256-B coalesced reads
__global__ __launch_bounds__(256) void slab_sum_kernel(const unsigned* __restrict__ slabs, int n_slabs, int nb,
                                                       const unsigned* __restrict__ bad,
                                                       unsigned long long* __restrict__ out) {
    __shared__ unsigned long long part[4][64];
    const int b = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + b;
    unsigned long long s = 0ull;
    if (i < nb)
        for (int q = grp; q < n_slabs; q += 4) s += slabs[(size_t)q * nb + i];
    part[grp][b] = s;
    __syncthreads();
    if (grp == 0 && i < nb) out[i] = part[0][b] + part[1][b] + part[2][b] + part[3][b];
    if (blockIdx.x == 0 && threadIdx.x == 0) out[nb] = *bad;
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

size_t sgpr_pair_histogram_workspace_bytes(const sgpr_handle* h, int n_prefix, int bits) {
    if (!h || n_prefix < 1 || n_prefix > HB_MAX_PREFIX || bits < 1 || bits > HB_MAX_BITS) return 0;
    return (size_t)h->num_cus * ((size_t)n_prefix << bits) * 2 * sizeof(unsigned) + 16;
}

int sgpr_pair_histogram(const sgpr_handle* h, const float* d_score, int R, int M, int64_t ld, int row0,
                        const double* d_pose_xz, double d_pos, double d_neg, const signed char* d_gt, int64_t ldg,
                        int n_prefix, int prefix_bits, int bits, const uint32_t* prefixes,
                        unsigned long long* d_hist, void* d_workspace, size_t workspace_bytes, void* stream) {
    if (!h || !d_score || !d_hist || R < 0 || M < 0 || ld < M || (!d_pose_xz && !d_gt) || (d_gt && !d_pose_xz && ldg < M)) {
        set_error("sgpr_pair_histogram: NULL argument, negative size or leading dimension below M");
        return SGPR_E_INVALID;
    }
    if (n_prefix < 1 || n_prefix > HB_MAX_PREFIX || bits < 1 || bits > HB_MAX_BITS || prefix_bits < 0 ||
        prefix_bits + bits > 32 || (prefix_bits > 0 && !prefixes)) {
        set_error("sgpr_pair_histogram: 1..4 prefixes, 1..12 bits per pass, prefix_bits + bits <= 32");
        return SGPR_E_INVALID;
    }
    const size_t need = sgpr_pair_histogram_workspace_bytes(h, n_prefix, bits);
    if (!d_workspace || workspace_bytes < need) {
        set_error("sgpr_pair_histogram: workspace of " + std::to_string(need) + " bytes required");
        return SGPR_E_WORKSPACE;
    }
    DeviceGuard guard(h->device);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int nb = (n_prefix << bits) * 2;
    HistArgs a;
    memset(&a, 0, sizeof(a));
    a.score = d_score;
    a.R = R;
    a.M = M;
    a.ld = ld;
    a.row0 = row0;
    a.pose = d_pose_xz;
    a.d_pos = d_pos;
    a.d_neg = d_neg;
    a.gt = d_gt;
    a.ldg = ldg;
    a.n_prefix = n_prefix;
    a.prefix_bits = prefix_bits;
    a.bits = bits;
    for (int p = 0; p < n_prefix; ++p) a.prefix[p] = prefix_bits ? prefixes[p] : 0u;
    a.bad = reinterpret_cast<unsigned*>(d_workspace);
    a.slabs = a.bad + 4;
    hipError_t e = hipMemsetAsync(a.bad, 0, 16, s);
    if (e != hipSuccess) return hip_fail(e, "sgpr_pair_histogram: memset");
    static bool attr_set = false;  // benign race: idempotent
    if (!attr_set) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&pair_histogram_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(pair_histogram_kernel)");
        attr_set = true;
    }
    const int grid = h->num_cus;
    hipLaunchKernelGGL(pair_histogram_kernel, dim3(grid), dim3(HB_THREADS), nb * sizeof(unsigned), s, a);
    e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "pair_histogram_kernel launch");
    hipLaunchKernelGGL(slab_sum_kernel, dim3((nb + 63) / 64), dim3(256), 0, s, a.slabs, grid, nb, a.bad, d_hist);
    e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "slab_sum_kernel launch");
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
