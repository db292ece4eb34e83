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
//    => two per-node GEMMs (a = W1'x, b = (W2-W1)'x + t) and a gather-max over the k neighbours.
//  * every matrix product (Gram, per-node GEMMs, conv_end) runs on v_mfma_f32_16x16x32_f16 with both operands split
//    into two f16 planes (x = hi + lo, 22 bits; three cross products in two fp32 chains, see tile16); graphs whose
//    values leave the f16 range are embedded again by the wide-range instance (three bf16 planes = 24 bits on
//    v_mfma_f32_16x16x32_bf16, or fp32 rows split on load).  The fp32 MFMA blocks the VALU of its SIMD for 32 cycles
//    per instruction and is several times slower per product.
//  * kNN: Gram matrix X.X^T (upper triangle only when the whole key matrix fits in LDS), ranking key
//    |x_j|^2 - 2 x_i.x_j (= the reference's -pairwise_distance up to the row constant |x_i|^2) - except in the
//    coordinate layer, whose keys restate the reference's fp32 arithmetic operation for operation on the vector ALU
//    (gram_xyz_direct: bit-identical neighbour sets there) -, then an exact k-smallest selection per row: register
//    sorting networks + butterfly merges, deterministic lowest-index tie-break.
//  * trailing duplicate (padding) slots collapse to one representative; with packed input the whole semantic branch
//    runs on 13 label super-nodes (embed_kernel, "semantic branch on label super-nodes").
//  * one workgroup per graph, 128..512 threads chosen by the host plan (make_embed_plan): graphs of <= 64 processed
//    slots run on the lean instance - one fixed 39 424-byte LDS layout (30 208 bytes for <= 48 slots), 91 (79) VGPRs,
//    no scratch: four (five) 4-wave workgroups per CU, limited by LDS; everything between the input read and the
//    pooled vector lives in LDS / registers.
//  * input: padded arrays, the reference's dense tensors, or a ragged store whose padding is made in registers
//    (sgpr_embed_ragged).
#include <math.h>

#include <atomic>
#include <type_traits>

#include "sgpr_internal.hpp"
#ifndef SGPR_EXP_BARRIERS
#define SGPR_EXP_BARRIERS 1      // timing experiment only (tools/build_variant.sh): every workgroup barrier of this file N times -
#endif                           // the launch's growth per extra copy is what its barriers cost (results stay valid)
#if SGPR_EXP_BARRIERS > 1
namespace sgpr {
__device__ __forceinline__ void sync_n() {
    __syncthreads();
#pragma unroll
    for (int i = 1; i < SGPR_EXP_BARRIERS; ++i) __builtin_amdgcn_s_barrier();
}
}
#define __syncthreads() ::sgpr::sync_n()     // (after sync_n's own use of the real one)
#endif

namespace sgpr {

constexpr int PXB = 400;      // BYTES per row of X as three bf16 planes of 64 channels (128 B each; x = hi + mid + lo,
                              // exact to 24 bits) + 16 B so that rows shift one 16-B slot; the per-node term b
                              // (64 fp32 = 256 B) later overlays the row in place
constexpr int PXF = 272;      // bytes per row of X in the fp32 layout (64 ch + 4 floats)
// bytes per row of X as two f16 planes (x = hi + lo, 22 bits; 2 x 128 B + padding).  The rows are read as MFMA operands
// with ds_read_b128, whose lane groups are {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... (MI355X_MICROARCH.md, LDS): at
// a pitch of 17 16-byte slots lane (row r, k-slice q) starts at slot r + q (mod 16), and every group holds one pair of
// lanes on the same slot - a 2-way conflict in all four groups of every operand read (8 LDS cycles instead of 4).  At 18
// slots the start is 2 r + q: the k-slices of even / odd q sit on even / odd slots and all 16 lanes of a group differ.
// Measured (round 3, same box, 288-byte rows against 272): KITTI-00 launch 186.8 vs 185.2 us, stress 866.2 vs 865.8,
// pairs128 33.3 vs 33.1 - the conflicts are real (SQ_LDS_BANK_CONFLICT: 39 % of them sit in the GEMM phase, 25 % in the
// Gram phase) but the LDS array is only ~38 % busy and nothing waits on it: the smaller rows stay.
constexpr int PXH = 272;
#ifndef SGPR_LEAN_WAVES
#define SGPR_LEAN_WAVES 4
#endif
constexpr int kLeanNT = 64 * SGPR_LEAN_WAVES;   // threads of a lean workgroup (4 waves = one per 16-row tile; experiments: 5)
// X layout of a kernel instance (EmbedPlan::fmt)
constexpr int FMT_F32 = 0;    // fp32 rows, split into three bf16 planes when loaded (fallback of plans too large for FMT_BF3)
constexpr int FMT_BF3 = 1;    // three bf16 planes, written once per layer by the gather epilogue (fp32 range: the fallback
                              // for graphs whose activations leave the f16 range)
constexpr int FMT_H2 = 2;     // two f16 planes (the default): half the matrix instructions and a third of the split work
constexpr float kF16Safe = 60000.f;
constexpr int kSmallParkBytes = 16 * 32 * 4 + 256;   // 16 parked super-node rows + the branch's scratch (cnt[16], rowlab[NP <= 64])
constexpr int kHybridParkBytes = 16 * 32 * 4 + 64 + 256;   // the same for the large plans (rowlab[NP <= 256])
constexpr int PE = 36;        // floats per row of E   (final node embedding, 32 ch + 4)
constexpr int PP = 32;        // floats per row of the parked sem3 block (output of the first branch)
constexpr int NT_MAX = 512;   // threads per workgroup: 256 or 512 (blockDim.x), up to 256 VGPRs per lane either way
constexpr int CAP = 64;       // candidates per lane in the selection phase
constexpr int kRedBytes = 1536;   // attention partial sums (8 x 32) + mean + tanh vector; duplicate-run scratch
constexpr int kLdsLimit = 160 * 1024;

typedef float f32x4 __attribute__((ext_vector_type(4)));

// threadIdx.x behind an opaque copy: every lane-derived address is then computed inside the phase that uses it.
// Without it the optimizer hoists dozens of them out of the six-layer loop, where they live in - and spill from -
// registers across all phases (the scratch traffic showed up as 440 MB of HBM writes per launch).
__device__ __forceinline__ int phase_tid() {
    int t = threadIdx.x;
    asm volatile("" : "+v"(t));
    return t;
}

static inline int round_up(int v, int m) { return (v + m - 1) / m * m; }
#ifndef SGPR_OVER_IN_REDO
#define SGPR_OVER_IN_REDO 0   // "auto" launches, node_num <= 128, K = 10: 1 = the second pass serves the oversize graphs instead of a launch
#endif                        // of their own.  Measured (tools/run_auto.py, same box): KITTI-00 shape, no oversize graph: 147.7 us per
                              // call against 149 + 5 (the empty launch) - nothing; 693 / 2180 oversize graphs of 4541: 501 / 791 us
                              // against 307 / 443 (one workgroup per CU at the second pass's LDS size instead of two): off
#ifndef SGPR_BIG_SEM_WAVE
#define SGPR_BIG_SEM_WAVE 0   // 1 (A/B builds): owned-rows instance, the super-node branch on one otherwise idle wave beside the first
#endif                        // xyz layer.  Measured (same box, twice): config 5 548.6 / 551.0 -> 572.2 / 575.6 us - bit-identical, SLOWER:
                              // one wave needs longer for the branch (its 256-thread loops four times over) than the other waves for
                              // their selection + GEMMs, so they wait for it at their first barrier instead of it hiding behind them
constexpr int kSemWaveBytes = 16 * PXH + 32 * 68 * 4 + 64 + 64;   // X rows | A rows + keys | norms | label sets
#ifndef SGPR_BIG_OWNED
#define SGPR_BIG_OWNED 1      // production plans beyond 64 rows on the owned-rows instance (0: A/B builds on the chunked plans)
#endif

// The LDS layout of every lean plan (node_cap <= 64, f16 planes) is ONE fixed layout - 64 rows, 16 neighbour slots per
// row - so that the lean kernel instance sees its offsets and pitches as compile-time constants (no scalar registers
// for them: the runtime plan cost that instance 160 spilled SGPRs) while the host and the instrumented instances read
// the same numbers from the plan.  small_park: the park holds the 16 super-node rows only (production launches).
__host__ __device__ constexpr void lean_fixed_layout(EmbedPlan& p, bool small_park, int rows) {
    p.NP = rows;                 // 64, or 48 (30 208 bytes: five workgroups per CU) when no graph needs more slots
    p.pitchD = rows + 4;
    p.pitchA = 68;
    p.kpitch = 16;
    p.RC = rows;
    p.P = 1;
    p.overlap = 1;
    p.park_in_lds = 1;
    p.park_hybrid = 0;
    p.small_park = small_park ? 1 : 0;
    p.alias_da = 1;
    p.lean = rows;
    p.big = 0;
    p.sem_wave = 0;
    p.offSem = 0;
    p.xplanes = 1;
    p.fmt = FMT_H2;
    p.rowb = PXH;
    p.nt = kLeanNT;
    p.offX = 0;
    p.offRed = 0;
    p.offPark = rows * PXH;
    p.offXX = p.offPark + (small_park ? kSmallParkBytes : rows * PP * 4);
    p.offIdx = p.offXX + 64 * 4;
    p.offA = p.offIdx + rows * 16 * 2;
    p.offD = p.offA;
    p.lds_bytes = p.offA + rows * 68 * 4;
#if SGPR_EXP_LDS32      // timing experiment only (results invalid): the 64-row layout cut to 32 KB = five workgroups per CU
    if (p.lds_bytes > 32768) p.lds_bytes = 32768;
#endif
}

static bool plan_layout(int N, int NC, int k, int fmt, EmbedPlan* p, bool small_park = false, int min_nt = 0) {
    const bool planes = fmt != FMT_F32;
    p->N = N;
    p->NC = NC;
    p->NP = round_up(NC, 16);
    p->k = k;
    p->kp = k <= 16 ? 16 : 32;
    p->kpitch = round_up(k, 2);           // u16 entries per row of the neighbour list
    p->pitchD = p->NP + 4;
    p->park_in_lds = NC <= 128 ? 1 : 0;
    p->pitchA = NC <= 192 ? 68 : 64;      // 64 only when LDS is otherwise exhausted (bank-conflicted stores)
    p->xplanes = planes ? 1 : 0;
    p->fmt = fmt;
    p->rowb = fmt == FMT_BF3 ? PXB : (fmt == FMT_H2 ? PXH : PXF);
    int off = 0;
    p->offX = off;    off += p->NP * p->rowb;
    p->offRed = p->offX;
    p->small_park = small_park ? 1 : 0;
    p->park_hybrid = p->park_in_lds ? 0 : 1;
    p->offPark = off; off += p->park_in_lds ? (small_park ? kSmallParkBytes : p->NP * PP * 4) : kHybridParkBytes;
    p->offXX = off;   off += p->NP * 4;
    p->offIdx = off;  off += round_up(p->NP * p->kpitch * 2, 16);
    p->offA = off;
    p->offD = off;
    p->alias_da = 1;
    const int rowD = p->pitchD * 4, bytesA = p->NP * p->pitchA * 4;
    const int region = kLdsLimit - off;
    if (region < bytesA) return false;
    int rc = region / rowD / 16 * 16;
    if (rc > p->NP) rc = p->NP;
    if (rc < 16) return false;
    p->lds_bytes = off + (rc * rowD > bytesA ? rc * rowD : bytesA);
    // small graphs: 256-thread workgroups, two of them per CU, overlap each other's barriers
    p->nt = p->lds_bytes <= kLdsLimit / 2 ? 256 : 512;
    if (p->nt < min_nt) p->nt = min_nt;
    // resident mode: the whole key matrix fits in LDS -> upper-triangular Gram tiles, mirrored
    int P = 1;
    while ((NC + P - 1) / P > CAP) P *= 2;
    p->overlap = (rc == p->NP && NC <= 128 && p->NP * P <= p->nt) ? 1 : 0;
    if (!p->overlap && rc * P > p->nt) {
        rc = p->nt / P / 16 * 16;
        if (rc < 16) return false;
        p->lds_bytes = off + (rc * rowD > bytesA ? rc * rowD : bytesA);
    }
    p->P = P;                                   // lower bound; the kernel widens it per graph
    p->seg = round_up((NC + P - 1) / P, 4);
    p->RC = rc;
    // lean plans (NP <= 64, plane layouts): one wave per 16-row tile, 12..16 waves per CU on the <= 128-VGPR kernel instance
    p->lean = 0;
    p->big = 0;
    p->sem_wave = 0;
    p->offSem = 0;
    if (planes && p->overlap && p->NP <= 64 && min_nt == 0) {
        int nt3 = 64 * (p->NP / 16);
        if (nt3 < round_up(N, 64)) nt3 = round_up(N, 64);     // one thread per input slot
        if (nt3 < 128) nt3 = 128;                             // gemm_cols needs two waves
        if (nt3 <= 256 && (12 / (nt3 / 64)) * p->lds_bytes <= kLdsLimit && fmt == FMT_H2 && p->kpitch <= 16) {
            // one of two fixed layouts (see above); the 48-row one only for production launches (no timers / dumps)
            lean_fixed_layout(*p, small_park, (small_park && NC <= 48) ? 48 : 64);
            p->seg = round_up(NC, 4);
        }
    }
    if (small_park && !p->lean) return false;   // the 16-row park only serves the lean instance's super-node branch
    return true;
}

// Production plans beyond 64 processed slots, K = 10 or 20 (embed_big_kernel): one wave per 16-row tile (up to 16 waves),
// at most 128 registers per lane - sixteen waves per CU where the chunked plans run eight.  No key matrix: a wave takes
// the Gram tiles of its rows against every candidate tile from the accumulators (select_owned_big), so LDS holds X, the
// neighbour lists, the gather target and the 16 super-node rows of the semantic branch (a graph that needs the generic
// branch goes to the second pass, as on the lean plans).
static bool plan_big(int N, int NC, int k, EmbedPlan* p) {
    if ((k != 10 && k != 20) || NC <= 64) return false;
    EmbedPlan full;
    if (!plan_layout(N, NC, k, FMT_H2, &full)) return false;
    *p = full;
    p->NP = round_up(NC, 16);
    p->pitchD = 20;                       // (only the 16 x 16 key tile of the super-node branch)
    p->park_in_lds = 0;
    p->park_hybrid = 1;
    p->small_park = 1;
    p->alias_da = 1;
    p->lean = 0;
    p->big = full.overlap ? 2 : 1;
    p->overlap = 0;
    p->xplanes = 1;
    p->fmt = FMT_H2;
    p->rowb = PXH;
    int off = 0;
    p->offX = off;    off += p->NP * PXH;
    p->offRed = p->offX;
    p->offPark = off; off += kHybridParkBytes;
    p->offXX = off;   off += p->NP * 4;
    p->offIdx = off;  off += round_up(p->NP * p->kpitch * 2, 16);
    p->offA = off;
    p->offD = off;
    p->pitchA = off + p->NP * 68 * 4 <= kLdsLimit ? 68 : 64;
    p->lds_bytes = off + p->NP * p->pitchA * 4;
    if (p->lds_bytes > kLdsLimit) return false;
    p->nt = 64 * (p->NP / 16);
    if (p->nt < round_up(N, 64)) p->nt = round_up(N, 64);       // one thread per input slot
    p->RC = p->NP;
    if (p->nt > 1024) return false;
#if SGPR_BIG_SEM_WAVE
    // the super-node branch on a wave of its own, beside the first xyz layer (embed_graph): 16 rows of X, 32 of A / keys,
    // norms, label sets = kSemWaveBytes of LDS - and one wave more than the rows need, while the block size allows
    if (p->lds_bytes + kSemWaveBytes <= kLdsLimit) {
        p->sem_wave = 1;
        p->offSem = p->lds_bytes;
        p->lds_bytes += kSemWaveBytes;
        if (p->nt == 64 * (p->NP / 16) && p->nt + 64 <= 1024) p->nt += 64;
    }
#endif
    return true;
}

bool make_embed_plan(int N, int node_cap, int k, EmbedPlan* p, bool wide_range, bool small_park, int min_nt) {
    if (N < 1 || N > SGPR_MAX_NODES || k < 1 || k > SGPR_MAX_K || k > N) return false;
    const int NC = (node_cap <= 0 || node_cap > N) ? N : node_cap;
    // default: two f16 planes (fit every node_num).  wide_range (the fallback for graphs whose activations leave the
    // f16 range): bf16 planes whenever they fit (up to 208 processed slots), fp32 rows beyond
    if (wide_range) return plan_layout(N, NC, k, FMT_BF3, p, false, min_nt) || plan_layout(N, NC, k, FMT_F32, p, false, min_nt);
    // production launches of lean plans park only the 16 super-node rows of the first branch (four workgroups per CU
    // instead of three); a graph that needs the generic branch is handed to the second pass (embed_redo_kernel)
    if (small_park && plan_layout(N, NC, k, FMT_H2, p, true, 0)) return true;
#if SGPR_BIG_OWNED
    if (small_park && plan_big(N, NC, k, p)) return true;
#endif
    return plan_layout(N, NC, k, FMT_H2, p, false, min_nt);
}

struct KParams {
    DevWeights w;
    EmbedPlan p;
    EmbedPlan p2;     // embed_redo_kernel only: the full f16 plan beside the wide-range plan in p
    EmbedPlan p3;     // embed_redo_kernel only: the owned-rows plan for node_num slots (big != 0: the kernel serves flag 3 itself)
    EmbedArgs a;
    int num_cus;      // (host side: grid of the persistent launches)
};

// ------------------------------------------------------------------ MFMA helpers
// Every matrix product of the kernel (Gram, per-node GEMMs, conv_end) runs on the 16x16x32 half-precision matrix
// instructions with both operands split into planes: two f16 planes (x = hi + lo, 22 bits, three significant cross
// products) in the default instance, three bf16 planes (x = hi + mid + lo, exact to 24 bits, six products) in the
// wide-range instance; products are accumulated in fp32, the correction terms in a chain of their own (tile16).
// fp32 MFMA (v_mfma_f32_16x16x4_f32) blocks the VALU of its SIMD for 32 cycles per instruction
// (tools/probes/coexec_probe.hip); the half-precision instructions do not, and three of them cost 48 cycles per
// 16x16x32 block against 256 in fp32.
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16v2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// one 32-wide k-step of one operand: lane (l15, lq) holds k slots 8*lq .. 8*lq+7 of row l15, one register quad per plane
template <int FMT>
struct FragT {          // FMT_BF3 / FMT_F32: three bf16 planes
    bf16x8 h, m, l;
};
template <>
struct FragT<FMT_H2> {  // two f16 planes
    f16x8 h, l;
};

__device__ __forceinline__ f32x4 mfma_b(bf16x8 a, bf16x8 b, f32x4 c) {
    // D[4*(l>>4)+r][l&15] += sum_k A[row][k] * B[k][col];  lane l supplies A[l&15][8*(l>>4)..+7], B[8*(l>>4)..+7][l&15]
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma_h(f16x8 a, f16x8 b, f32x4 c) {   // same operand / result layout
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

// 16x16 output tile, K = 16*NKB (NKB = 4: two k-steps; NKB = 1: one half-filled k-step).
// The matrix core aligns the 32 products of an instruction and its C operand to the largest exponent and drops what
// falls below fp32 precision, so the correction terms must never meet a large accumulator: they get a chain of their
// own (smallest terms first), the hi.hi products another, and the two sums meet in ONE fp32 add.
// bf16 x 3 planes: six significant cross products, error ~ 4 ulp of the result - tighter than a sequential fp32 dot
// product.  f16 x 2 planes: three products (lo.hi, hi.lo, hi.hi), operands exact to 22 bits.  Dependent back-to-back
// MFMAs issue every 18 cycles; more, shorter chains would be slower (tools/probes/mfma_chain_probe.hip).
template <int NKB, int FMT>
__device__ __forceinline__ f32x4 tile16(const FragT<FMT> (&a)[2], const FragT<FMT> (&b)[2]) {
    constexpr int NS = NKB == 1 ? 1 : 2;
    f32x4 lo = {0.f, 0.f, 0.f, 0.f}, hi = {0.f, 0.f, 0.f, 0.f};
    if constexpr (FMT == FMT_H2) {
#pragma unroll
        for (int st = 0; st < NS; ++st) {
            lo = mfma_h(a[st].l, b[st].h, lo);
            lo = mfma_h(a[st].h, b[st].l, lo);
        }
#pragma unroll
        for (int st = 0; st < NS; ++st) hi = mfma_h(a[st].h, b[st].h, hi);
    } else {
#pragma unroll
        for (int st = 0; st < NS; ++st) {
            lo = mfma_b(a[st].l, b[st].h, lo);
            lo = mfma_b(a[st].h, b[st].l, lo);
            lo = mfma_b(a[st].m, b[st].m, lo);
        }
#pragma unroll
        for (int st = 0; st < NS; ++st) {
            lo = mfma_b(a[st].m, b[st].h, lo);
            lo = mfma_b(a[st].h, b[st].m, lo);
        }
#pragma unroll
        for (int st = 0; st < NS; ++st) hi = mfma_b(a[st].h, b[st].h, hi);
    }
    return hi + lo;
}

// X operand of row `row` (byte pointer to the row) in a plane layout: channels 32*step + 8*lq + 0..7 (NKB = 4), or
// channels 4*lq + 0..3 followed by four zero slots (NKB = 1: 16 input channels); planes are 128 bytes apart
template <int NKB, int FMT>
__device__ __forceinline__ void load_xfrag(const unsigned char* row, int lq, FragT<FMT> (&f)[2]) {
    if constexpr (FMT == FMT_H2) {
        if (NKB == 1) {
            const f16x4 h = *reinterpret_cast<const f16x4*>(row + 8 * lq);
            const f16x4 l = *reinterpret_cast<const f16x4*>(row + 128 + 8 * lq);
            const _Float16 z = (_Float16)0.f;
            f[0].h = f16x8{h[0], h[1], h[2], h[3], z, z, z, z};
            f[0].l = f16x8{l[0], l[1], l[2], l[3], z, z, z, z};
        } else {
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                f[st].h = *reinterpret_cast<const f16x8*>(row + 64 * st + 16 * lq);
                f[st].l = *reinterpret_cast<const f16x8*>(row + 128 + 64 * st + 16 * lq);
            }
        }
    } else {
        if (NKB == 1) {
            const bf16x4 h = *reinterpret_cast<const bf16x4*>(row + 8 * lq);
            const bf16x4 m = *reinterpret_cast<const bf16x4*>(row + 128 + 8 * lq);
            const bf16x4 l = *reinterpret_cast<const bf16x4*>(row + 256 + 8 * lq);
            f[0].h = bf16x8{h[0], h[1], h[2], h[3], 0, 0, 0, 0};
            f[0].m = bf16x8{m[0], m[1], m[2], m[3], 0, 0, 0, 0};
            f[0].l = bf16x8{l[0], l[1], l[2], l[3], 0, 0, 0, 0};
        } else {
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                f[st].h = *reinterpret_cast<const bf16x8*>(row + 64 * st + 16 * lq);
                f[st].m = *reinterpret_cast<const bf16x8*>(row + 128 + 64 * st + 16 * lq);
                f[st].l = *reinterpret_cast<const bf16x8*>(row + 256 + 64 * st + 16 * lq);
            }
        }
    }
}

// weight operand of one 16-row column tile (DevWeights::wb / wh layout: [k-step][plane][lane][8]);
// tile = weights + ct * wtile<NKB, FMT>
template <int NKB, int FMT>
__device__ __forceinline__ constexpr int wtile() { return (NKB == 1 ? 1 : 2) * (FMT == FMT_H2 ? 2 : 3) * 512; }
template <int NKB, int FMT>
__device__ __forceinline__ void load_wfrag(const unsigned short* __restrict__ tile, FragT<FMT> (&f)[2], const bool constw = false) {
    if (constw) {
        // ablation (mask bit 9): constant weights, no loads
        if constexpr (FMT == FMT_H2) {
            const _Float16 c = (_Float16)0.01f;
            f[0].h = f[0].l = f[1].h = f[1].l = f16x8{c, c, c, c, c, c, c, c};
        } else {
            f[0].h = f[0].m = f[0].l = f[1].h = f[1].m = f[1].l = bf16x8{1, 2, 3, 4, 5, 6, 7, 8};
        }
        return;
    }
    const unsigned short* p = tile + (phase_tid() & 63) * 8;
#pragma unroll
    for (int st = 0; st < (NKB == 1 ? 1 : 2); ++st) {
        if constexpr (FMT == FMT_H2) {
            f[st].h = *reinterpret_cast<const f16x8*>(p + (st * 2 + 0) * 512);
            f[st].l = *reinterpret_cast<const f16x8*>(p + (st * 2 + 1) * 512);
        } else {
            f[st].h = *reinterpret_cast<const bf16x8*>(p + (st * 3 + 0) * 512);
            f[st].m = *reinterpret_cast<const bf16x8*>(p + (st * 3 + 1) * 512);
            f[st].l = *reinterpret_cast<const bf16x8*>(p + (st * 3 + 2) * 512);
        }
    }
}

template <int NKB, int FMT>
__device__ __forceinline__ void copy_frag(FragT<FMT> (&d)[2], const FragT<FMT> (&s)[2]) {
    d[0] = s[0];
    if (NKB != 1) d[1] = s[1];
}

// four consecutive fp32 values -> three bf16 planes, two packed values per dword (v_cvt_pk_bf16_f32, round to nearest
// even; x - hi and (x - hi) - mid are exact in fp32)
__device__ __forceinline__ void split4(float4 v, uint2& h, uint2& m, uint2& l) {
    const f32x2 a = {v.x, v.y}, b = {v.z, v.w};
    const bf16v2 ha = __builtin_convertvector(a, bf16v2), hb = __builtin_convertvector(b, bf16v2);
    const f32x2 ra = a - __builtin_convertvector(ha, f32x2), rb = b - __builtin_convertvector(hb, f32x2);
    const bf16v2 ma = __builtin_convertvector(ra, bf16v2), mb = __builtin_convertvector(rb, bf16v2);
    const f32x2 sa = ra - __builtin_convertvector(ma, f32x2), sb = rb - __builtin_convertvector(mb, f32x2);
    const bf16v2 la = __builtin_convertvector(sa, bf16v2), lb = __builtin_convertvector(sb, bf16v2);
    h = make_uint2(__builtin_bit_cast(unsigned, ha), __builtin_bit_cast(unsigned, hb));
    m = make_uint2(__builtin_bit_cast(unsigned, ma), __builtin_bit_cast(unsigned, mb));
    l = make_uint2(__builtin_bit_cast(unsigned, la), __builtin_bit_cast(unsigned, lb));
}

// four consecutive fp32 values -> two f16 planes: hi = the value truncated to f16 (v_cvt_pkrtz_f16_f32), lo = f16(v - hi)
// straight out of one mixed-precision FMA per value (v_fma_mixlo / mixhi_f16: f16 * f32 + f32 -> f16 half of the
// destination); hi + lo = v to 22 bits (|v| < 2^-14: to 2^-25 absolute - the matrix core honours f16 denormals).
// The operands come from LDS reads / vector arithmetic, never straight out of an MFMA (whose wait states the compiler
// would not insert in front of inline asm).
__device__ __forceinline__ void split4_h2(float4 v, uint2& h, uint2& l) {
    const unsigned h01 = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(v.x, v.y));
    const unsigned h23 = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(v.z, v.w));
    unsigned l01, l23;
    asm("v_fma_mixlo_f16 %0, %2, -1.0, %4 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixlo_f16 %1, %3, -1.0, %6 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %0, %2, -1.0, %5 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %1, %3, -1.0, %7 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
        : "=&v"(l01), "=&v"(l23)
        : "v"(h01), "v"(h23), "v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w));
    h = make_uint2(h01, h23);
    l = make_uint2(l01, l23);
}

__device__ __forceinline__ bf16x8 pack8(uint2 a, uint2 b) {
    return __builtin_bit_cast(bf16x8, u32x4{a.x, a.y, b.x, b.y});
}

// X layout policy (FMT_*).  Plane layouts are written once per layer by the gather epilogue; FMT_F32 keeps fp32 rows
// and splits when loading.  Either way the per-node term b (fp32) overlays bytes 0..255 of the row in place.
template <int FMT>
__device__ __forceinline__ constexpr int xrow() { return FMT == FMT_BF3 ? PXB : (FMT == FMT_H2 ? PXH : PXF); }

template <int NKB, int FMT>
__device__ __forceinline__ void xload(const unsigned char* row, int lq, FragT<FMT> (&f)[2]) {
    if constexpr (FMT != FMT_F32) {
        load_xfrag<NKB, FMT>(row, lq, f);
    } else if (NKB == 1) {
        uint2 h, m, l;
        split4(*reinterpret_cast<const float4*>(row + 16 * lq), h, m, l);
        const uint2 z = make_uint2(0u, 0u);
        f[0].h = pack8(h, z);
        f[0].m = pack8(m, z);
        f[0].l = pack8(l, z);
    } else {
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            uint2 h0, m0, l0, h1, m1, l1;
            split4(*reinterpret_cast<const float4*>(row + 128 * st + 32 * lq), h0, m0, l0);
            split4(*reinterpret_cast<const float4*>(row + 128 * st + 32 * lq + 16), h1, m1, l1);
            f[st].h = pack8(h0, h1);
            f[st].m = pack8(m0, m1);
            f[st].l = pack8(l0, l1);
        }
    }
}

// four consecutive channels ch..ch+3 of one row.  FMT_H2 also tracks the largest magnitude it has stored (`vmax`): a
// graph whose activations reach the f16 range is re-run on the wide-range instance (see embed_kernel)
template <int FMT>
__device__ __forceinline__ void xstore(unsigned char* row, int ch, float4 v, float& vmax) {
    if constexpr (FMT == FMT_BF3) {
        uint2 h, m, l;
        split4(v, h, m, l);
        *reinterpret_cast<uint2*>(row + 2 * ch) = h;
        *reinterpret_cast<uint2*>(row + 128 + 2 * ch) = m;
        *reinterpret_cast<uint2*>(row + 256 + 2 * ch) = l;
    } else if constexpr (FMT == FMT_H2) {
        uint2 h, l;
        split4_h2(v, h, l);
        *reinterpret_cast<uint2*>(row + 2 * ch) = h;
        *reinterpret_cast<uint2*>(row + 128 + 2 * ch) = l;
        float m2;
        asm("v_max3_f32 %0, |%1|, |%2|, %3" : "=v"(m2) : "v"(v.x), "v"(v.y), "v"(vmax));
        asm("v_max3_f32 %0, |%1|, |%2|, %3" : "=v"(vmax) : "v"(v.z), "v"(v.w), "v"(m2));
    } else {
        *reinterpret_cast<float4*>(row + 4 * ch) = v;
    }
}

// four zero channels ch..ch+3 of one row: plain stores, no split arithmetic
template <int FMT>
__device__ __forceinline__ void xzero(unsigned char* row, int ch) {
    const uint2 z = make_uint2(0u, 0u);
    if constexpr (FMT == FMT_BF3) {
        *reinterpret_cast<uint2*>(row + 2 * ch) = z;
        *reinterpret_cast<uint2*>(row + 128 + 2 * ch) = z;
        *reinterpret_cast<uint2*>(row + 256 + 2 * ch) = z;
    } else if constexpr (FMT == FMT_H2) {
        *reinterpret_cast<uint2*>(row + 2 * ch) = z;
        *reinterpret_cast<uint2*>(row + 128 + 2 * ch) = z;
    } else {
        *reinterpret_cast<float4*>(row + 4 * ch) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

// ------------------------------------------------------------------ selection helpers
__device__ __forceinline__ float max3(float a, float b, float c);   // (v_max3_f32, defined with the gather)
// Exact min / max of two keys (never NaN) as ONE instruction each.  The builtins (fminf, fmed3 with an infinite bound,
// which the optimizer folds back into it) first canonicalise every operand that does not come out of an arithmetic
// instruction - a v_max x, x behind each LDS read and each DPP move: 78 such instructions in the lean instance, most of
// them in the selection's merges.  Operands come from LDS reads, DPP moves or other minima / maxima, never straight
// from an MFMA (whose wait states the compiler would not insert in front of inline asm).
#ifndef SGPR_EXP_DOUBLE
#define SGPR_EXP_DOUBLE 0     // timing experiments only (tools/build_variant.sh): run an idempotent phase TWICE - the launch's time
#endif                        // difference is what the phase costs, with valid results (1 counting selection, 2 selection, 4 Gram
                              // tiles, 8 super-node branch, 16 coordinate-layer keys; big instance: 32 selection, 64 per-node GEMMs)
#ifndef SGPR_ASM_MINMAX
#define SGPR_ASM_MINMAX 1
#endif
#ifndef SGPR_IN_PREFETCH
#define SGPR_IN_PREFETCH 0    // launch slots ahead whose input lines a workgroup pulls into its XCD's L2 (A/B builds; measured: 1024 / 512
                              // slots ahead 128.4 -> 130.0 / 130.2 us per KITTI-00 launch - the opening round trip is not what a workgroup waits for)
#endif
#ifndef SGPR_WPREFETCH
#define SGPR_WPREFETCH 1      // lean 64-row production instance: a GEMM phase's first weight fragment requested before the selection
#endif
#ifndef SGPR_ATT_PREFETCH
#define SGPR_ATT_PREFETCH 1   // attention: the column of att_w a lane needs two barriers later is requested ahead of the partial sums
#endif
#ifndef SGPR_SEM_STAGE
#define SGPR_SEM_STAGE 1      // super-node branch, tabled layer 2: this graph's table rows staged in LDS / registers beside the keys
#endif
#if SGPR_ASM_MINMAX
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
#else
__device__ __forceinline__ float kmin(float a, float b) { return __builtin_amdgcn_fmed3f(a, b, -INFINITY); }
__device__ __forceinline__ float kmax(float a, float b) { return __builtin_amdgcn_fmed3f(a, b, INFINITY); }
#endif

__device__ __forceinline__ void cswap(float& a, float& b) {
    const float lo = kmin(a, b);
    b = kmax(a, b);
    a = lo;
}

template <int N>
__device__ __forceinline__ void bitonic_merge(float (&v)[N]) {  // bitonic in -> ascending out
#pragma unroll
    for (int j = N / 2; j > 0; j >>= 1)
#pragma unroll
        for (int i = 0; i < N; ++i)
            if ((i & j) == 0) cswap(v[i], v[i | j]);
}

template <int N>
__device__ __forceinline__ void bitonic_sort(float (&v)[N]) {  // any -> ascending
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

// Batcher's odd-even merge sort (63 comparators for 16 values against the bitonic sorter's 80) of N values of which
// the entries >= NV are +inf: such an entry never leaves its place in an ascending network, so comparators that touch
// one are no-ops and are not generated (NV = 12: 42 comparators).  Outputs the caller does not use cost nothing
// either: the comparators that feed only them are dead code.
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

// value of lane (l ^ m) for m = 1, 2 (DPP quad permute), 4 (ds_swizzle), else ds_bpermute
__device__ __forceinline__ float lane_xor(float v, int m) {
    const int iv = __float_as_int(v);
    int r;
    if (m == 1)
        r = __builtin_amdgcn_update_dpp(iv, iv, 0xB1, 0xF, 0xF, false);   // quad_perm [1,0,3,2]
    else if (m == 2)
        r = __builtin_amdgcn_update_dpp(iv, iv, 0x4E, 0xF, 0xF, false);   // quad_perm [2,3,0,1]
    else if (m == 4)
        r = __builtin_amdgcn_ds_swizzle(iv, 0x101F);                      // bit mode: xor 4
    else
        r = __shfl_xor(iv, m);
    return __int_as_float(r);
}

// The KEEP smallest (ascending; entries >= KEEP are +inf) of the first `cnt` (wave-uniform) of the 32 candidates
// d[OFF..OFF+32)
template <int KP, int KEEP, int OFF, int CAPX>
__device__ __forceinline__ void list_from_32(const float (&d)[CAPX], int cnt, float (&L)[KP]) {
    if (cnt <= 8) {
        float lo[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) lo[u] = d[OFF + u];
        batcher_sort<8, 8>(lo);
#pragma unroll
        for (int u = 0; u < KP; ++u) L[u] = (u < 8 && u < KEEP) ? lo[u & 7] : INFINITY;
        return;
    }
    float lo[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) lo[u] = d[OFF + u];
    if (cnt <= 12) {
#pragma unroll
        for (int u = 12; u < 16; ++u) lo[u] = INFINITY;      // (they are +inf at run time already: now the compiler knows)
        batcher_sort<16, 12>(lo);
    } else {
        batcher_sort<16, 16>(lo);
    }
    if (CAPX >= OFF + 32 && cnt > 16) {
        float hi[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) hi[u] = d[CAPX >= OFF + 32 ? OFF + 16 + u : 0];
        batcher_sort<16, 16>(hi);
        if (KP == 16) {
#pragma unroll
            for (int u = 0; u < 16; ++u) L[u] = kmin(lo[u], hi[15 - u]);
        } else {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                L[u] = lo[u];
                L[(KP - 16) + u] = hi[15 - u];
            }
        }
        bitonic_merge<KP>(L);
#pragma unroll
        for (int u = KEEP; u < KP; ++u) L[u] = INFINITY;
    } else {
#pragma unroll
        for (int u = 0; u < KP; ++u) L[u] = (u < 16 && u < KEEP) ? lo[u & 15] : INFINITY;
    }
}

// value of lane (l - sh) within the same 16-lane row (0 where that lane does not exist)
template <int SH>
__device__ __forceinline__ int row_shr(int v) {
    return __builtin_amdgcn_update_dpp(0, v, 0x110 + SH, 0xF, 0xF, true);   // row_shr:SH, bound_ctrl: zero fill
}

// order-preserving float -> uint32
__device__ __forceinline__ unsigned ord_u32(float f) {
    f += 0.0f;  // -0 -> +0
    const unsigned b = __float_as_uint(f);
    return b ^ ((unsigned)((int)b >> 31) | 0x80000000u);
}

// the value of another lane of the same 16-lane row: DPP row rotation by S (which of the two neighbours at distance S
// it is does not matter to the caller below, which visits all fifteen distances and carries the lane's identity along)
template <int S>
__device__ __forceinline__ int row_rot(int v) {
    return __builtin_amdgcn_update_dpp(v, v, 0x120 + S, 0xF, 0xF, false);   // row_ror:S
}
// super-node counting selection: nodes of the labels ranked ahead of label j = sum of the counts of the other 15
// candidates of the row whose key is smaller (or equal with a smaller label).  packed = count | label << 16
// One 64-bit unsigned comparison per step: (order-preserving image of the key, label << 16 | count) against (own image,
// own label << 16) is "smaller key, or the same key and a smaller label" - no lane-mask logic on the scalar unit.
template <int S>
__device__ __forceinline__ int count_before(unsigned okey, unsigned packed, unsigned long long mine, int before) {
    if constexpr (S < 16) {
        const unsigned ko = (unsigned)row_rot<S>((int)okey);
        const unsigned po = (unsigned)row_rot<S>((int)packed);
        before += (((unsigned long long)ko << 32) | po) < mine ? (int)(po & 0xffffu) : 0;
        return count_before<S + 1>(okey, packed, mine, before);
    } else {
        return before;
    }
}

// emit the candidates whose bits are set in `take` (32 candidate slots starting at index jbase)
__device__ __forceinline__ void emit_bits(unsigned take, int jbase, int pitchA, unsigned short* __restrict__ out,
                                          int32_t* __restrict__ dbg_row, int& pos) {
    while (take) {
        const int u = __ffs(take) - 1;
        take &= take - 1;
        out[pos] = (unsigned short)((jbase + u) * pitchA);
        if (dbg_row) dbg_row[pos] = jbase + u;
        ++pos;
    }
}

// Exact k-smallest selection for the rows of one key chunk.
// P consecutive lanes share a row; lane `part` owns candidates [part*seg, (part+1)*seg), seg <= CAP.
// Result: nbr[i][0..k) = (float offset of the A row of) the k nearest candidates of row i under the
// total order (key ascending, index ascending) - written as an unordered set.
// CAPX = candidates a lane may own: CAP in general, 16 in lean plans (NP <= 64 and >= 4 lanes per row whenever a row
// has more than 16 candidates), which drops the 32- and 64-candidate paths and their registers from that instance.
// KEEP = list entries that matter (k <= KEEP <= KP): the sorted lists carry +inf beyond it, and every comparator,
// lane exchange and minimum that would only feed those entries is never generated (KEEP = 10 for the reference's K).
template <int KP, int CAPX, int KEEP, int PC = 0>   // PC: lanes per row when it is a compile-time constant
__device__ __forceinline__ void select_phase(const EmbedPlan& p, int n, int np, int P_, int seg, int k, bool one_rep,
                                             const float* __restrict__ D, int rc0, int rows_chunk,
                                             unsigned short* __restrict__ nbr, int32_t* __restrict__ dbg_knn,
                                             unsigned long long* __restrict__ prof8) {
    const int P = PC ? PC : P_;
    const int tid = phase_tid(), lane = tid & 63;
    unsigned long long ts = (prof8 && tid == 0) ? clock64() : 0ull;
#define SEL_STAMP(i)                                                   \
    if (prof8 && tid == 0) {                                           \
        const unsigned long long tn_ = clock64();                      \
        atomicAdd(&prof8[i], tn_ - ts);                                \
        ts = tn_;                                                      \
    }
    const int rl = tid >> (__ffs(P) - 1), part = tid & (P - 1);
    const int i = rc0 + rl;
    // a wave whose rows all lie past the graph's last slot has nothing to select (no barrier in this phase)
    if (rc0 + (((tid & ~63)) >> (__ffs(P) - 1)) >= n) return;
    const bool active = (rl < rows_chunk) && (i < n);
    const float* drow = D + (active ? rl : 0) * p.pitchD;
    const int j0 = part * seg;
    // keys of candidates j >= n are +inf already (written so by the Gram phase); only the row end (np) needs a guard
    const int nq = min(seg, max(np - j0, 0)) >> 2;   // 16-B groups this lane really owns

    float d[CAPX];
#pragma unroll
    for (int c = 0; c < CAPX / 16; ++c) {
        if (16 * c < seg) {                          // wave-uniform
#pragma unroll
            for (int q = 4 * c; q < 4 * c + 4; ++q) {
                const float4 v = *reinterpret_cast<const float4*>(drow + j0 + 4 * min(q, max(nq - 1, 0)));
                const bool ok = q < nq;
                d[4 * q + 0] = ok ? v.x : INFINITY;
                d[4 * q + 1] = ok ? v.y : INFINITY;
                d[4 * q + 2] = ok ? v.z : INFINITY;
                d[4 * q + 3] = ok ? v.w : INFINITY;
            }
        }   // slots >= seg are never read: every later use is guarded by the same (wave-uniform) test
    }
    SEL_STAMP(0)
    // this lane's KP smallest, ascending
    float L[KP];
    list_from_32<KP, KEEP, 0, CAPX>(d, seg, L);
    if (CAPX > 32 && seg > 32) {
        float L2[KP];
        list_from_32<KP, KP, (CAPX > 32 ? 32 : 0), CAPX>(d, seg - 32, L2);
#pragma unroll
        for (int u = 0; u < KP; ++u) L[u] = kmin(L[u], L2[KP - 1 - u]);
        bitonic_merge<KP>(L);
#pragma unroll
        for (int u = KEEP; u < KP; ++u) L[u] = INFINITY;
    }
    SEL_STAMP(1)
    // butterfly merge of the P sorted lists of a row: min(L[s], O[KP-1-s]) = the KP smallest of the union, bitonic
    // (own and partner entries >= KEEP are +inf: no exchange for them, no minimum against them)
#define SGPR_MERGE_ROUND(M)                                              \
    if (P > (M)) {                                                       \
        float o[KP];                                                     \
        _Pragma("unroll") for (int s = 0; s < KP; ++s) o[s] = (KP - 1 - s < KEEP) ? lane_xor(L[KP - 1 - s], (M)) : INFINITY; \
        _Pragma("unroll") for (int s = 0; s < KP; ++s)                   \
            L[s] = s < KEEP ? ((KP - 1 - s < KEEP) ? kmin(L[s], o[s]) : L[s]) : o[s]; \
        bitonic_merge<KP>(L);                                            \
        _Pragma("unroll") for (int s = KEEP; s < KP; ++s) L[s] = INFINITY; \
    }
    float tau;                                       // the k-th smallest key of the row
    bool have_tau = false;
    SGPR_MERGE_ROUND(1)
    if constexpr (PC == 4 && KEEP == 10) {
        // LAST round of the lean instances (four lanes per row): only the K-th smallest key is needed, not the merged list.
        // min(L[s], O[K-1-s]), s < K, ARE the K smallest of the two lists (as a set): their maximum is that key - five
        // v_max3 instead of the ~15 comparators of the merge network that feed entry K-1
        float mn[10];
#pragma unroll
        for (int s = 0; s < 10; ++s) mn[s] = kmin(L[s], lane_xor(L[9 - s], 2));
        tau = max3(max3(max3(mn[0], mn[1], mn[2]), max3(mn[3], mn[4], mn[5]), max3(mn[6], mn[7], mn[8])), mn[9], mn[9]);
        have_tau = true;
    } else {
        SGPR_MERGE_ROUND(2)
        SGPR_MERGE_ROUND(4)
        SGPR_MERGE_ROUND(8)
        SGPR_MERGE_ROUND(16)
        SGPR_MERGE_ROUND(32)
    }
#undef SGPR_MERGE_ROUND
    SEL_STAMP(2)
    if (have_tau) {
    } else if (KEEP == 10 || k == 10) {
        tau = L[9];
    } else if (KP == 32 && (KEEP == 20 || k == 20)) {
        tau = L[KP == 32 ? 19 : 0];
    } else {
        tau = L[0];
#pragma unroll
        for (int s = 1; s < KP; ++s) tau = (s == k - 1) ? L[s] : tau;
    }
    // Last slot = representative of >= k identical slots (padding): if its key lies below the k-th key, every
    // neighbour slot after the candidates at or below that key goes to copies of it, so the neighbour SET is
    // {key <= key_rep}; the unused list entries are pre-filled with the representative itself.
    bool dup_cut = false;
    if (one_rep) {
        const float krep = drow[n - 1];
        if (krep < tau) {
            tau = krep;
            dup_cut = true;
        }
        if (active) {
            unsigned short* out0 = nbr + i * p.kpitch;
            const unsigned short rep = (unsigned short)((n - 1) * p.pitchA);
            for (int q = part; q < k; q += P) out0[q] = rep;
            if (dbg_knn)
                for (int q = part; q < k; q += P) dbg_knn[(size_t)i * p.k + q] = n - 1;
        }
    }
    if constexpr (PC == 4 && CAPX == 16) {
        // Lean instances, the common case first: when exactly k keys of the row lie at or below tau (no ties across the
        // cut) - or the representative cuts the list - the neighbour set is {key <= tau}: ONE mask (the sign bits of
        // tau - key, complemented), a prefix over the four lanes of the row, emission.  Rows with ties across the cut
        // (identical nodes; kept padding copies) take the general path below, the whole wave together.
        unsigned gt = 0u;
#pragma unroll
        for (int u = 15; u >= 0; --u) gt = __builtin_amdgcn_alignbit(gt, __float_as_uint(tau - d[u]), 31);
        const unsigned le = ~gt & 0xffffu;
        const int n_le = __popc(le);
        int incl = n_le;
        {
            const int t = row_shr<1>(incl);
            incl += (part >= 1) ? t : 0;
        }
        {
            const int t = row_shr<2>(incl);
            incl += (part >= 2) ? t : 0;
        }
        const int total_le = __builtin_amdgcn_update_dpp(0, incl, 0xFF, 0xF, 0xF, false);   // quad_perm [3,3,3,3]
        if (__ballot(active && !(dup_cut || total_le == k)) == 0ull) {
            if (active) {
                int pos = incl - n_le;
                emit_bits(le, j0, p.pitchA, nbr + i * p.kpitch, dbg_knn ? dbg_knn + (size_t)i * p.k : nullptr, pos);
            }
            SEL_STAMP(5)
            return;
        }
    }
    if constexpr (PC == 0) {
        // the same shortcut for the general instances (up to 64 candidates per lane, 2..16 lanes per row)
        if (P <= 16) {
            unsigned g0 = 0u, g1 = 0u;
#pragma unroll
            for (int c = 1; c >= 0; --c)
                if (16 * c < CAPX && 16 * c < seg) {
#pragma unroll
                    for (int u = 16 * c + 15; u >= 16 * c; --u)
                        g0 = __builtin_amdgcn_alignbit(g0, __float_as_uint(tau - d[u < CAPX ? u : 0]), 31);
                }
#pragma unroll
            for (int c = 1; c >= 0; --c)
                if (32 + 16 * c < CAPX && 32 + 16 * c < seg) {
#pragma unroll
                    for (int u = 16 * c + 15; u >= 16 * c; --u)
                        g1 = __builtin_amdgcn_alignbit(g1, __float_as_uint(tau - d[32 + u < CAPX ? 32 + u : 0]), 31);
                }
            const unsigned v0 = (CAPX > 16 && seg > 16) ? 0xffffffffu : 0xffffu;
            const unsigned v1 = (CAPX > 32 && seg > 32) ? ((CAPX > 48 && seg > 48) ? 0xffffffffu : 0xffffu) : 0u;
            const unsigned le0 = ~g0 & v0, le1 = ~g1 & v1;
            const int n_le = __popc(le0) + __popc(le1);
            int incl = n_le;
            if (P > 1) {
                const int t = row_shr<1>(incl);
                incl += (part >= 1) ? t : 0;
            }
            if (P > 2) {
                const int t = row_shr<2>(incl);
                incl += (part >= 2) ? t : 0;
            }
            if (P > 4) {
                const int t = row_shr<4>(incl);
                incl += (part >= 4) ? t : 0;
            }
            if (P > 8) {
                const int t = row_shr<8>(incl);
                incl += (part >= 8) ? t : 0;
            }
            const int total_le = __shfl(incl, (lane & ~(P - 1)) + P - 1);
            if (__ballot(active && !(dup_cut || total_le == k)) == 0ull) {
                if (active) {
                    int pos = incl - n_le;
                    unsigned short* out = nbr + i * p.kpitch;
                    int32_t* dbg_row = dbg_knn ? dbg_knn + (size_t)i * p.k : nullptr;
                    emit_bits(le0, j0, p.pitchA, out, dbg_row, pos);
                    if (seg > 32) emit_bits(le1, j0 + 32, p.pitchA, out, dbg_row, pos);
                }
                SEL_STAMP(5)
                return;
            }
        }
    }
    // per-lane bit sets of the candidates below / at the threshold: the sign bit of (key - tau) and of (tau - key),
    // shifted into the masks with one v_alignbit each (highest candidate first, so candidate u ends up in bit u);
    // equal keys (and inf - inf, whose NaN is positive) set neither -> the "equal" set is what remains
    unsigned lt0 = 0u, lt1 = 0u, gt0 = 0u, gt1 = 0u;
#pragma unroll
    for (int c = 1; c >= 0; --c)
        if (16 * c < CAPX && 16 * c < seg) {
#pragma unroll
            for (int u = 16 * c + 15; u >= 16 * c; --u) {
                const float dv = d[u < CAPX ? u : 0];
                lt0 = __builtin_amdgcn_alignbit(lt0, __float_as_uint(dv - tau), 31);
                gt0 = __builtin_amdgcn_alignbit(gt0, __float_as_uint(tau - dv), 31);
            }
        }
#pragma unroll
    for (int c = 1; c >= 0; --c)
        if (32 + 16 * c < CAPX && 32 + 16 * c < seg) {
#pragma unroll
            for (int u = 16 * c + 15; u >= 16 * c; --u) {
                const float dv = d[32 + u < CAPX ? 32 + u : 0];
                lt1 = __builtin_amdgcn_alignbit(lt1, __float_as_uint(dv - tau), 31);
                gt1 = __builtin_amdgcn_alignbit(gt1, __float_as_uint(tau - dv), 31);
            }
        }
    // slots are processed in whole blocks of 16: the valid width of a word is 16 x the blocks processed for it
    const unsigned w0 = (CAPX > 16 && seg > 16) ? 0xffffffffu : 0xffffu;
    const unsigned w1 = (CAPX > 32 && seg > 32) ? ((CAPX > 48 && seg > 48) ? 0xffffffffu : 0xffffu) : 0u;
    unsigned eq0 = ~(lt0 | gt0) & w0, eq1 = ~(lt1 | gt1) & w1;
    if (dup_cut) {           // take every candidate at or below the representative's key, no tie limit
        lt0 |= eq0;
        lt1 |= eq1;
        eq0 = eq1 = 0u;
    }
    SEL_STAMP(3)
    const int n_less = __popc(lt0) + __popc(lt1), n_eq = __popc(eq0) + __popc(eq1);
    const int packed = n_less | (n_eq << 16);
    // inclusive prefix over the P lanes of the row (P <= 16: DPP row shifts; else ds_bpermute), and the row total
    int incl = packed, total;
    if (P <= 16) {
        // (the DPP reads must execute with every lane enabled: keep them outside the per-lane selects)
        if (P > 1) {
            const int t = row_shr<1>(incl);
            incl += (part >= 1) ? t : 0;
        }
        if (P > 2) {
            const int t = row_shr<2>(incl);
            incl += (part >= 2) ? t : 0;
        }
        if (P > 4) {
            const int t = row_shr<4>(incl);
            incl += (part >= 4) ? t : 0;
        }
        if (P > 8) {
            const int t = row_shr<8>(incl);
            incl += (part >= 8) ? t : 0;
        }
        total = __shfl(incl, (lane & ~(P - 1)) + P - 1);
    } else {
        int e = 0;
        const int base = lane & ~(P - 1);
        for (int q = 0; q < P; ++q) {
            const int v = __shfl(packed, base + q);
            if (q <= part) e += v;
        }
        incl = e;
        total = __shfl(incl, (lane & ~(P - 1)) + P - 1);
    }
    const int e_pack = incl - packed;                       // exclusive
    const int e_less = e_pack & 0xffff, e_eq = e_pack >> 16;
    const int T = dup_cut ? 0 : k - (total & 0xffff);       // ties at tau to accept, lowest index first
    SEL_STAMP(4)
    if (active) {
        int pos = e_less + min(e_eq, T);
        int my_ties = max(0, min(n_eq, T - e_eq));
        // keep only the first my_ties tie bits (index order: low word first)
        unsigned t0 = eq0, t1 = eq1;
        if (my_ties < n_eq) {
            unsigned keep0 = 0u, keep1 = 0u;
            while (my_ties > 0 && t0) {
                const unsigned low = t0 & (0u - t0);
                keep0 |= low;
                t0 ^= low;
                --my_ties;
            }
            while (my_ties > 0 && t1) {
                const unsigned low = t1 & (0u - t1);
                keep1 |= low;
                t1 ^= low;
                --my_ties;
            }
            t0 = keep0;
            t1 = keep1;
        }
        unsigned short* out = nbr + i * p.kpitch;
        int32_t* dbg_row = dbg_knn ? dbg_knn + (size_t)i * p.k : nullptr;
        emit_bits(lt0 | t0, j0, p.pitchA, out, dbg_row, pos);
        if (seg > 32) emit_bits(lt1 | t1, j0 + 32, p.pitchA, out, dbg_row, pos);
    }
    SEL_STAMP(5)
#undef SEL_STAMP
}

// ------------------------------------------------------------------ lean production instances: keys straight from the accumulators
// "Owned rows": wave w owns the 16 rows of row tile w.  It computes the Gram tiles of its rows against EVERY candidate
// tile with the candidates as the A operand and its rows as the B operand, so that the accumulator of lane (l15, lq)
// holds the keys of row 16 w + l15 for candidates 16 tj + 4 lq + r (r = 0..3): after nrt tiles the lane has the <= 16
// candidates the selection works on IN REGISTERS - no key matrix in LDS (8 stores + 4 loads per lane and layer), no
// barrier between the Gram phase and the selection, no upper-triangle bookkeeping; the matrix cores (9 % busy) do
// nrt^2 tiles instead of nrt (nrt + 1) / 2.  The four lanes of a row sit 16 lanes apart, so the butterfly merges
// exchange through ds_bpermute (the LDS pipe, not the vector ALU this kernel is short of) instead of DPP.
// Bit-identical keys: a key of the resident path comes from tile16(a = lower tile, b = higher tile), whose correction
// chain adds lo(a).hi(b) before hi(a).lo(b); here the operands are (candidates, rows), so a candidate tile at or above
// the row tile takes the chain in the other order (tile16_swapped), one below it the standard one.
#ifndef SGPR_OWNED_SELECT
#define SGPR_OWNED_SELECT 1
#endif

template <int FMT>
__device__ __forceinline__ f32x4 tile16_swapped(const FragT<FMT> (&a)[2], const FragT<FMT> (&b)[2]) {
    static_assert(FMT == FMT_H2, "lean instances run on the f16 planes");
    f32x4 lo = {0.f, 0.f, 0.f, 0.f}, hi = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int st = 0; st < 2; ++st) {
        lo = mfma_h(a[st].h, b[st].l, lo);           // = lo(row) . hi(candidate) of tile16(rows, candidates) ...
        lo = mfma_h(a[st].l, b[st].h, lo);           // ... then hi(row) . lo(candidate)
    }
#pragma unroll
    for (int st = 0; st < 2; ++st) hi = mfma_h(a[st].h, b[st].h, hi);
    return hi + lo;
}

// candidate index of bit u of a lane's mask: tile u >> 2, lane group lq, element u & 3
__device__ __forceinline__ int owned_cand(int u, int lq) { return 16 * (u >> 2) + 4 * lq + (u & 3); }

__device__ __forceinline__ void emit_owned(unsigned take, int lq, int pitchA, unsigned short* __restrict__ out, int& pos) {
    while (take) {
        const int u = __ffs(take) - 1;
        take &= take - 1;
        out[pos++] = (unsigned short)(owned_cand(u, lq) * pitchA);
    }
}

// coord: the coordinate layer (keys = the reference's fp32 arithmetic, gram_xyz_direct's operations on the same operands);
// a run-time flag, so that the sorting networks behind the keys exist once in the instance (instruction cache)
template <int FMT>
__device__ __forceinline__ void select_owned(const EmbedPlan& p, const int n, const int nrt, const bool one_rep,
                                             const unsigned char* __restrict__ X, const float* __restrict__ xx,
                                             unsigned short* __restrict__ nbr, const int wave, const bool coord) {
    constexpr int K = 10, KP = 16, XR = xrow<FMT>();
    if (wave >= nrt) return;                         // no rows of the graph in this wave's tile
    const int lane = phase_tid() & 63, l15 = lane & 15, lq = lane >> 4;
    const int i = 16 * wave + l15;
    const bool active = i < n;
    // the representative of >= K identical trailing slots (select_phase): its key, picked up while its tile goes by
    const int jr = n - 1, tr = jr >> 4, rr = jr & 3, lqr = (jr >> 2) & 3;
    float d[16], krep_part = 0.f;
    if (coord) {
        const float4 ci = *reinterpret_cast<const float4*>(X + i * XR + (XR - 16));
#pragma unroll
        for (int tj = 0; tj < 4; ++tj) {
            float key[4] = {INFINITY, INFINITY, INFINITY, INFINITY};
            if (tj < nrt) {                          // wave-uniform
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float4 cj = *reinterpret_cast<const float4*>(X + (16 * tj + 4 * lq + r) * XR + (XR - 16));
                    const float dot = fmaf(ci.z, cj.z, fmaf(ci.y, cj.y, __fmul_rn(ci.x, cj.x)));
                    const float t = fmaf(2.f, dot, -cj.w);
                    key[r] = __fsub_rn(ci.w, t);     // (+inf for an empty slot: its |x|^2 is)
                }
                if (tj == tr) krep_part = rr == 0 ? key[0] : (rr == 1 ? key[1] : (rr == 2 ? key[2] : key[3]));
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) d[4 * tj + r] = key[r];
        }
    } else {
        FragT<FMT> b[2];
        xload<4, FMT>(X + i * XR, lq, b);
#pragma unroll
        for (int tj = 0; tj < 4; ++tj) {
            float key[4] = {INFINITY, INFINITY, INFINITY, INFINITY};
            if (tj < nrt) {                          // wave-uniform
                FragT<FMT> a[2];
                xload<4, FMT>(X + (16 * tj + l15) * XR, lq, a);
                const f32x4 g = tj >= wave ? tile16_swapped<FMT>(a, b) : tile16<4, FMT>(a, b);
                const int jb = 16 * tj + 4 * lq;
                const float4 xj = *reinterpret_cast<const float4*>(xx + jb);
                key[0] = fmaf(-2.f, g[0], jb + 0 < n ? xj.x : INFINITY);
                key[1] = fmaf(-2.f, g[1], jb + 1 < n ? xj.y : INFINITY);
                key[2] = fmaf(-2.f, g[2], jb + 2 < n ? xj.z : INFINITY);
                key[3] = fmaf(-2.f, g[3], jb + 3 < n ? xj.w : INFINITY);
                if (tj == tr) krep_part = rr == 0 ? key[0] : (rr == 1 ? key[1] : (rr == 2 ? key[2] : key[3]));
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) d[4 * tj + r] = key[r];
        }
    }
    // ---- this lane's K smallest (ascending), then the butterfly over the row's four lanes (16 and 32 lanes away)
    float L[KP];
    list_from_32<KP, K, 0, 16>(d, 4 * nrt, L);
    {
        float o[KP];
#pragma unroll
        for (int s = 0; s < KP; ++s) o[s] = (KP - 1 - s < K) ? __shfl_xor(L[KP - 1 - s], 16) : INFINITY;
#pragma unroll
        for (int s = 0; s < KP; ++s) L[s] = s < K ? ((KP - 1 - s < K) ? kmin(L[s], o[s]) : L[s]) : o[s];
        bitonic_merge<KP>(L);
    }
    float tau;                                       // the K-th smallest key of the row
    {
        float mn[K];
#pragma unroll
        for (int s = 0; s < K; ++s) mn[s] = kmin(L[s], __shfl_xor(L[K - 1 - s], 32));
        tau = max3(max3(max3(mn[0], mn[1], mn[2]), max3(mn[3], mn[4], mn[5]), max3(mn[6], mn[7], mn[8])), mn[9], mn[9]);
    }
    unsigned short* out = nbr + i * p.kpitch;
    bool dup_cut = false;
    if (one_rep) {
        const float krep = __shfl(krep_part, l15 + 16 * lqr);
        if (krep < tau) {
            tau = krep;
            dup_cut = true;
        }
        if (active) {
            const unsigned short rep = (unsigned short)(jr * p.pitchA);
            for (int q = lq; q < K; q += 4) out[q] = rep;
        }
    }
    // ---- the common case: exactly K keys at or below tau (or the representative's cut) -> one mask, a prefix, emission
    unsigned gt = 0u;
#pragma unroll
    for (int u = 15; u >= 0; --u) gt = __builtin_amdgcn_alignbit(gt, __float_as_uint(tau - d[u]), 31);
    const unsigned le = ~gt & 0xffffu;
    const int n_le = __popc(le);
    int incl = n_le;                                 // inclusive prefix over the row's lanes (lq order)
    {
        const int t1 = __shfl_up(incl, 16);
        incl += lq >= 1 ? t1 : 0;
        const int t2 = __shfl_up(incl, 32);
        incl += lq >= 2 ? t2 : 0;
    }
    const int total_le = __shfl(incl, 48 + l15);
    if (__ballot(active && !(dup_cut || total_le == K)) == 0ull) {
        if (active) {
            int pos = incl - n_le;
            emit_owned(le, lq, p.pitchA, out, pos);
        }
        return;
    }
    // ---- ties across the cut (identical nodes, kept padding copies): the whole wave takes the general path.  Everything
    //      below tau, then the first T candidates AT tau in candidate-index order = (tile, lane group, element) order
    unsigned lt = 0u, gtm = 0u;
#pragma unroll
    for (int u = 15; u >= 0; --u) {
        lt = __builtin_amdgcn_alignbit(lt, __float_as_uint(d[u] - tau), 31);
        gtm = __builtin_amdgcn_alignbit(gtm, __float_as_uint(tau - d[u]), 31);
    }
    lt &= 0xffffu;
    unsigned eq = ~(lt | gtm) & 0xffffu;             // (inf - inf is a positive NaN: an empty slot at an infinite tau is "equal")
    if (dup_cut) {                                   // everything at or below the representative's key, no tie limit
        lt |= eq;
        eq = 0u;
    }
    const int n_less = __popc(lt);
    int less_incl = n_less;
    {
        const int t1 = __shfl_up(less_incl, 16);
        less_incl += lq >= 1 ? t1 : 0;
        const int t2 = __shfl_up(less_incl, 32);
        less_incl += lq >= 2 ? t2 : 0;
    }
    const int total_less = __shfl(less_incl, 48 + l15);
    const int T = dup_cut ? 0 : K - total_less;      // ties to accept
    // ties per tile of every lane of the row: one byte per tile
    const unsigned mine = (unsigned)__popc(eq & 0xfu) | ((unsigned)__popc(eq & 0xf0u) << 8) |
                          ((unsigned)__popc(eq & 0xf00u) << 16) | ((unsigned)__popc(eq & 0xf000u) << 24);
    unsigned pk[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) pk[q] = (unsigned)__shfl((int)mine, 16 * q + l15);
    if (active) {
        int pos = less_incl - n_less;
        emit_owned(lt, lq, p.pitchA, out, pos);
        int before = 0;                              // ties in the tiles before the current one
#pragma unroll 1
        for (int tj = 0; tj < 4; ++tj) {             // (rolled: this path is rare, the instance's code size is not)
            int ahead = 0, own = 0, all = 0;         // ties of this tile in the lane groups before mine / in mine / in all four
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int e = (int)((pk[q] >> (8 * tj)) & 0xffu);
                all += e;
                ahead += q < lq ? e : 0;
                own = q == lq ? e : own;
            }
            const int start = before + ahead;
            const int take_n = max(0, min(own, T - start));
            unsigned bits = (eq >> (4 * tj)) & 0xfu, keep = 0u;
            for (int c = 0; c < take_n; ++c) {
                const unsigned low = bits & (0u - bits);
                keep |= low;
                bits ^= low;
            }
            int tpos = total_less + min(start, T);
            emit_owned(keep << (4 * tj), lq, p.pitchA, out, tpos);
            before += all;
        }
    }
}

// ------------------------------------------------------------------ owned rows beyond 64 slots (embed_big_kernel)
// L, O ascending with +inf from entry K on -> L = the K smallest of the union, ascending (the butterfly step of
// select_phase as a function: min(L[s], O[KP-1-s]) is a bitonic sequence holding the KP smallest)
template <int KP, int K>
__device__ __forceinline__ void merge_keep(float (&L)[KP], const float (&O)[KP]) {
    float o[KP];
#pragma unroll
    for (int s = 0; s < KP; ++s) o[s] = (KP - 1 - s < K) ? O[KP - 1 - s] : INFINITY;
#pragma unroll
    for (int s = 0; s < KP; ++s) L[s] = s < K ? ((KP - 1 - s < K) ? kmin(L[s], o[s]) : L[s]) : o[s];
    bitonic_merge<KP>(L);
#pragma unroll
    for (int s = K; s < KP; ++s) L[s] = INFINITY;
}

// candidate index of bit u of a 64-bit owned mask (16 tiles x 4 elements)
__device__ __forceinline__ void emit_owned64(unsigned long long take, int lq, int pitchA, unsigned short* __restrict__ out, int& pos) {
    while (take) {
        const int u = __ffsll((long long)take) - 1;
        take &= take - 1ull;
        out[pos++] = (unsigned short)(owned_cand(u, lq) * pitchA);
    }
}

// Wave w owns the 16 rows of row tile w; up to 16 candidate tiles = 64 candidates per lane, which do not fit the 128
// registers of a sixteen-wave workgroup beside the sorting networks.  So the candidates STREAM: four tiles at a time the
// lane's 16 keys are sorted and merged into its running list of smallest keys (pass 1; the butterfly over the row's four
// lanes then yields the K-th smallest key of the row, as in select_owned), and the Gram tiles are computed AGAIN to mark
// the candidates at or below that key (pass 2: the matrix cores are ~10 % busy, the same instructions on the same operands
// give the same bits).  Keys: `sym` != 0 - the resident plans' operation order (select_owned), else the chunked plans'
// (gram_tile: rows as the A operand for every tile) - whatever the full plan of the same launch produces, so that a
// capped and an uncapped launch rank near-ties alike.
// The keys of candidate tile tj for this lane (candidates 16 tj + 4 lq + r).  xx[j] is +inf for every slot the graph does
// not have (embed_graph's gather epilogue writes it so), the coordinate layer's |x|^2 likewise: no index test per key
template <int FMT>
struct OwnedKeys {
    const unsigned char* X;
    const float* xx;
    FragT<FMT> b[2];
    float4 ci;
    int wave, l15, lq;
    bool coord, sym;
    __device__ __forceinline__ void operator()(const int tj, float (&key)[4]) const {
        constexpr int XR = xrow<FMT>();
        if (coord) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float4 cj = *reinterpret_cast<const float4*>(X + (16 * tj + 4 * lq + r) * XR + (XR - 16));
                const float dot = fmaf(ci.z, cj.z, fmaf(ci.y, cj.y, __fmul_rn(ci.x, cj.x)));
                const float t = fmaf(2.f, dot, -cj.w);
                key[r] = __fsub_rn(ci.w, t);         // (+inf for an empty slot: its |x|^2 is)
            }
        } else {
            FragT<FMT> a[2];
            xload<4, FMT>(X + (16 * tj + l15) * XR, lq, a);
            const f32x4 g = (sym && tj < wave) ? tile16<4, FMT>(a, b) : tile16_swapped<FMT>(a, b);
            const float4 xj = *reinterpret_cast<const float4*>(xx + 16 * tj + 4 * lq);
            key[0] = fmaf(-2.f, g[0], xj.x);
            key[1] = fmaf(-2.f, g[1], xj.y);
            key[2] = fmaf(-2.f, g[2], xj.z);
            key[3] = fmaf(-2.f, g[3], xj.w);
        }
    }
};

// Pass 1 and the butterflies: the K-th smallest key of the lane's row.  KL = entries a lane keeps of its OWN quarter of
// the candidates.  KL == K is exact by construction.  KL < K (2 KL >= K) is exact unless some lane holds more than KL of
// the row's K smallest - then its list was cut short, which shows as last_kept (the lane's KL-th smallest) <= the result:
// the caller repeats the pass with KL = K (a quarter of the candidates holding 13 of the 20 nearest: ~2e-4 per lane).
// krep_part: the key of candidate n - 1 (tile tr, element rr), picked up while its tile goes by.
template <int FMT, int K, int KP, int KL, int KPL>
__device__ __forceinline__ float owned_kth(const OwnedKeys<FMT>& keys, const int nrt, const int tr, const int rr,
                                           float& krep_part, float& last_kept) {
    static_assert(2 * KL >= K && KL <= K && KPL <= KP, "owned_kth: list sizes");
    float L[KPL];
#pragma unroll
    for (int s = 0; s < KPL; ++s) L[s] = INFINITY;
#pragma unroll 1
    for (int g0 = 0; g0 < nrt; g0 += 4) {            // wave-uniform
        float d[16];
#pragma unroll
        for (int tq = 0; tq < 4; ++tq) {
            float key[4] = {INFINITY, INFINITY, INFINITY, INFINITY};
            if (g0 + tq < nrt) {
                keys(g0 + tq, key);
                if (g0 + tq == tr) krep_part = rr == 0 ? key[0] : (rr == 1 ? key[1] : (rr == 2 ? key[2] : key[3]));
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) d[4 * tq + r] = key[r];
        }
        float S[KPL];
        list_from_32<KPL, KL, 0, 16>(d, 4 * min(4, nrt - g0), S);
        merge_keep<KPL, KL>(L, S);
    }
    last_kept = L[KL - 1];
    // ---- the butterfly over the row's four lanes (16 and 32 lanes away); the last round only needs the K-th key
    float V[KP];
    if constexpr (KL == K) {
#pragma unroll
        for (int s = 0; s < KP; ++s) V[s] = L[s < KPL ? s : 0];
        float O[KP];
#pragma unroll
        for (int s = 0; s < KP; ++s) O[s] = s < K ? __shfl_xor(V[s], 16) : INFINITY;
        merge_keep<KP, K>(V, O);
    } else {
        // two lists of KL: own ascending, then +inf, then the partner's descending - a bitonic sequence of KP
#pragma unroll
        for (int s = 0; s < KP; ++s) V[s] = INFINITY;
#pragma unroll
        for (int s = 0; s < KL; ++s) {
            V[s] = L[s];
            V[KP - 1 - s] = __shfl_xor(L[s], 16);
        }
        bitonic_merge<KP>(V);
    }
    float mn[K];
#pragma unroll
    for (int s = 0; s < K; ++s) mn[s] = kmin(V[s], __shfl_xor(V[K - 1 - s], 32));
    constexpr int G3 = (K + 2) / 3;
    float m3[G3];
#pragma unroll
    for (int s = 0; s < G3; ++s)
        m3[s] = max3(mn[3 * s], mn[3 * s + 1 < K ? 3 * s + 1 : 3 * s], mn[3 * s + 2 < K ? 3 * s + 2 : 3 * s]);
    float tau = m3[0];
#pragma unroll
    for (int s = 1; s + 1 < G3; s += 2) tau = max3(tau, m3[s], m3[s + 1]);
    if (G3 % 2 == 0) tau = kmax(tau, m3[G3 - 1]);
    return tau;
}

#ifndef SGPR_BIG_KL20
#define SGPR_BIG_KL20 12      // K = 20: entries a lane keeps of its own candidates in the first attempt (20: always exact, no retry)
#endif
template <int FMT, int K, int KP>
__device__ __forceinline__ void select_owned_big(const EmbedPlan& p, const int n, const int nrt, const bool one_rep,
                                                 const unsigned char* __restrict__ X, const float* __restrict__ xx,
                                                 unsigned short* __restrict__ nbr, const int wave, const bool coord,
                                                 const bool sym) {
    constexpr int XR = xrow<FMT>();
    if (wave >= nrt) return;                         // no rows of the graph in this wave's tile
    const int lane = phase_tid() & 63, l15 = lane & 15, lq = lane >> 4;
    const int i = 16 * wave + l15;
    const bool active = i < n;
    const int jr = n - 1, tr = jr >> 4, rr = jr & 3, lqr = (jr >> 2) & 3;
    OwnedKeys<FMT> keys;
    keys.X = X;
    keys.xx = xx;
    keys.wave = wave;
    keys.l15 = l15;
    keys.lq = lq;
    keys.coord = coord;
    keys.sym = sym;
    keys.ci = make_float4(0.f, 0.f, 0.f, 0.f);
    if (coord)
        keys.ci = *reinterpret_cast<const float4*>(X + i * XR + (XR - 16));
    else
        xload<4, FMT>(X + i * XR, lq, keys.b);
    // ---- pass 1: the K-th smallest key of the row
    constexpr int KL = (K == 20 && SGPR_BIG_KL20 < 20) ? SGPR_BIG_KL20 : K, KPL = KL <= 16 ? 16 : 32;
    float krep_part = 0.f, last_kept = 0.f;
    float tau = owned_kth<FMT, K, KP, KL, KPL>(keys, nrt, tr, rr, krep_part, last_kept);
    if constexpr (KL < K) {
        if (__ballot(active && !(last_kept > tau)) != 0ull)      // some lane's list may have been cut short: the exact pass
            tau = owned_kth<FMT, K, KP, K, KP>(keys, nrt, tr, rr, krep_part, last_kept);
    }
    unsigned short* out = nbr + i * p.kpitch;
    bool dup_cut = false;
    if (one_rep) {
        const float krep = __shfl(krep_part, l15 + 16 * lqr);
        if (krep < tau) {
            tau = krep;
            dup_cut = true;
        }
        if (active) {
            const unsigned short rep = (unsigned short)(jr * p.pitchA);
            for (int q = lq; q < K; q += 4) out[q] = rep;
        }
    }
    // ---- pass 2: the candidates at or below tau (bit 4 tj + r of the lane's mask)
    unsigned long long gt = 0ull;
#pragma unroll 2
    for (int tj = 0; tj < nrt; ++tj) {
        float key[4];
        keys(tj, key);
        unsigned nib = 0u;
#pragma unroll
        for (int r = 3; r >= 0; --r) nib = __builtin_amdgcn_alignbit(nib, __float_as_uint(tau - key[r]), 31);
        gt |= (unsigned long long)nib << (4 * tj);
    }
    const unsigned long long valid = nrt >= 16 ? ~0ull : ((1ull << (4 * nrt)) - 1ull);
    const unsigned long long le = ~gt & valid;
    const int n_le = __popcll(le);
    int incl = n_le;                                 // inclusive prefix over the row's lanes (lq order)
    {
        const int t1 = __shfl_up(incl, 16);
        incl += lq >= 1 ? t1 : 0;
        const int t2 = __shfl_up(incl, 32);
        incl += lq >= 2 ? t2 : 0;
    }
    const int total_le = __shfl(incl, 48 + l15);
    if (__ballot(active && !(dup_cut || total_le == K)) == 0ull) {
        if (active) {
            int pos = incl - n_le;
            emit_owned64(le, lq, p.pitchA, out, pos);
        }
        return;
    }
    // ---- ties across the cut (identical nodes, kept padding copies): a third pass marks the candidates below tau; then
    //      everything below, and the first T candidates AT tau in candidate-index order = (tile, lane group, element)
    unsigned long long ltm = 0ull;
#pragma unroll 1
    for (int tj = 0; tj < nrt; ++tj) {
        float key[4];
        keys(tj, key);
        unsigned nib = 0u;
#pragma unroll
        for (int r = 3; r >= 0; --r) nib = __builtin_amdgcn_alignbit(nib, __float_as_uint(key[r] - tau), 31);
        ltm |= (unsigned long long)nib << (4 * tj);
    }
    ltm &= valid;
    unsigned long long eq = ~(ltm | gt) & valid;     // (inf - inf is a positive NaN: an empty slot at an infinite tau is "equal")
    if (dup_cut) {                                   // everything at or below the representative's key, no tie limit
        ltm |= eq;
        eq = 0ull;
    }
    const int n_less = __popcll(ltm);
    int less_incl = n_less;
    {
        const int t1 = __shfl_up(less_incl, 16);
        less_incl += lq >= 1 ? t1 : 0;
        const int t2 = __shfl_up(less_incl, 32);
        less_incl += lq >= 2 ? t2 : 0;
    }
    const int total_less = __shfl(less_incl, 48 + l15);
    const int T = dup_cut ? 0 : K - total_less;      // ties to accept
    if (active) {
        int pos = less_incl - n_less;
        emit_owned64(ltm, lq, p.pitchA, out, pos);
    }
    int before = 0;                                  // ties in the tiles before the current one
#pragma unroll 1
    for (int tj = 0; tj < nrt; ++tj) {               // (every lane runs the exchanges; only active rows emit)
        const unsigned bits0 = (unsigned)(eq >> (4 * tj)) & 0xfu;
        const int own = __popc(bits0);
        int tin = own;
        {
            const int t1 = __shfl_up(tin, 16);
            tin += lq >= 1 ? t1 : 0;
            const int t2 = __shfl_up(tin, 32);
            tin += lq >= 2 ? t2 : 0;
        }
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
            emit_owned64((unsigned long long)keep << (4 * tj), lq, p.pitchA, out, tpos);
        }
        before += all;
    }
}

// ------------------------------------------------------------------ selection by value bisection (wave per row)
// wave64 min / max of a u32 (result uniform)
__device__ __forceinline__ unsigned wave_min_u32(unsigned v) {
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x111, 0xF, 0xF, false));   // row_shr:1
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x112, 0xF, 0xF, false));   // row_shr:2
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x114, 0xF, 0xF, false));   // row_shr:4
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x118, 0xF, 0xF, false));   // row_shr:8
    // lane 15 of each row holds the row minimum; combine the four rows
    const unsigned r0 = __builtin_amdgcn_readlane(v, 15), r1 = __builtin_amdgcn_readlane(v, 31);
    const unsigned r2 = __builtin_amdgcn_readlane(v, 47), r3 = __builtin_amdgcn_readlane(v, 63);
    return min(min(r0, r1), min(r2, r3));
}
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x111, 0xF, 0xF, false));
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x112, 0xF, 0xF, false));
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x114, 0xF, 0xF, false));
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x118, 0xF, 0xF, false));
    const unsigned r0 = __builtin_amdgcn_readlane(v, 15), r1 = __builtin_amdgcn_readlane(v, 31);
    const unsigned r2 = __builtin_amdgcn_readlane(v, 47), r3 = __builtin_amdgcn_readlane(v, 63);
    return max(max(r0, r1), max(r2, r3));
}

// One wave per row; lane l holds candidates l, l+64, ... (C of them).  The k-th smallest key is bracketed by
// bisection on the key VALUE: one v_cmp (= a 64-lane ballot) + scalar popcount per candidate slot and step,
// stopping as soon as exactly k keys lie below the pivot.  Ties at the k-th key (identical padding slots) run
// the interval down to a single value and are then taken in candidate-index order.  Emission needs no loop:
// a taken candidate's output position is the number of taken candidates before it (v_mbcnt).
template <int C>
__device__ __forceinline__ void select_bisect(const EmbedPlan& p, int n, int np, int k, bool one_rep,
                                              const float* __restrict__ D, int rc0, int rows_chunk,
                                              unsigned short* __restrict__ nbr,
                                              int32_t* __restrict__ dbg_knn) {
    const int tid_ = phase_tid();
    const int lane = tid_ & 63, wave = tid_ >> 6;
    const int NW = blockDim.x >> 6;
    for (int rl = wave; rl < rows_chunk; rl += NW) {
        const int i = rc0 + rl;
        if (i >= n) break;
        const float* drow = D + rl * p.pitchD;
        unsigned key[C];
        unsigned kmin_ = 0xffffffffu, kmax_ = 0u;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const int j = lane + 64 * c;
            const bool valid = j < n;                       // keys of [n, np) are +inf, beyond np unwritten
            const float v = drow[min(j, np - 1)];
            key[c] = valid ? ord_u32(v) : 0xffffffffu;
            kmin_ = min(kmin_, key[c]);
            kmax_ = max(kmax_, valid ? key[c] : 0u);
        }
        unsigned lo = wave_min_u32(kmin_), hi = wave_max_u32(kmax_);
        // invariant: #(key <= hi) >= k, #(key < lo) < k
        unsigned pivot = hi;
        bool exact = false;
        while (lo < hi) {
            const unsigned mid = lo + ((hi - lo) >> 1);
            int cnt = 0;
#pragma unroll
            for (int c = 0; c < C; ++c) cnt += __popcll(__ballot(key[c] <= mid));
            if (cnt == k) {
                pivot = mid;
                exact = true;
                break;
            }
            if (cnt > k)
                hi = mid;
            else
                lo = mid + 1;
        }
        if (!exact) pivot = lo;                              // the k-th smallest key itself; ties possible
        unsigned short* out = nbr + i * p.kpitch;
        if (one_rep) {       // see select_phase: the representative of >= k identical slots cuts the list
            const unsigned krep = ord_u32(drow[n - 1]);
            if (krep < pivot || (krep == pivot && exact)) {
                pivot = krep;
                exact = true;                                // take {key <= krep}
            }
            if (lane < k) {
                out[lane] = (unsigned short)((n - 1) * p.pitchA);
                if (dbg_knn) dbg_knn[(size_t)i * p.k + lane] = n - 1;
            }
        }
        // taken = {key < pivot} (or <= pivot when exact) + the first T ties in candidate order
        bool is_lt[C], is_eq[C];
        unsigned long long eq[C];
        int n_lt = 0;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            is_lt[c] = exact ? key[c] <= pivot : key[c] < pivot;
            is_eq[c] = !exact && key[c] == pivot;
            eq[c] = __ballot(is_eq[c]);
            n_lt += __popcll(__ballot(is_lt[c]));
        }
        int T = exact ? 0 : k - n_lt;                        // ties still to accept
        int base = 0;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const int tie_rank = __builtin_amdgcn_mbcnt_hi((unsigned)(eq[c] >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)eq[c], 0));
            const bool take = is_lt[c] || (is_eq[c] && tie_rank < T);
            const unsigned long long tk = __ballot(take);
            const int pos = base + __builtin_amdgcn_mbcnt_hi((unsigned)(tk >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)tk, 0));
            if (take) {
                out[pos] = (unsigned short)((lane + 64 * c) * p.pitchA);
                if (dbg_knn) dbg_knn[(size_t)i * p.k + pos] = lane + 64 * c;
            }
            base += __popcll(tk);
            T -= min(T, __popcll(eq[c]));
        }
    }
}

// ------------------------------------------------------------------ Gram tile -> ranking keys
template <int NKB, int FMT>
__device__ __forceinline__ void gram_tile(const unsigned char* __restrict__ X, const float* __restrict__ xx, float* __restrict__ D,
                                          int pitchD, int N, int rc0, int ti, int tj, bool mirror, int l15, int lq) {
    // ti indexes 16-row tiles inside the chunk starting at row rc0; tj indexes candidate tiles
    const int i0 = rc0 + ti * 16, j0 = tj * 16;
    FragT<FMT> a[2], b[2];
    xload<NKB, FMT>(X + (i0 + l15) * xrow<FMT>(), lq, a);
    xload<NKB, FMT>(X + (j0 + l15) * xrow<FMT>(), lq, b);
    const f32x4 g = tile16<NKB>(a, b);          // g[r] = <x_{i0+4lq+r}, x_{j0+l15}>
    const int j = j0 + l15;
    const float xj = j < N ? xx[j] : INFINITY;  // invalid candidates rank last
    float* drow = D + (ti * 16 + 4 * lq) * pitchD + j;
#pragma unroll
    for (int r = 0; r < 4; ++r) drow[r * pitchD] = fmaf(-2.f, g[r], xj);
    if (mirror) {  // the transposed tile: row j, candidates i0+4lq..+3 (one 16-B store)
        const int ib = i0 + 4 * lq;
        float4 xi = *reinterpret_cast<const float4*>(xx + ib);
        xi.x = (ib + 0 < N) ? xi.x : INFINITY;
        xi.y = (ib + 1 < N) ? xi.y : INFINITY;
        xi.z = (ib + 2 < N) ? xi.z : INFINITY;
        xi.w = (ib + 3 < N) ? xi.w : INFINITY;
        *reinterpret_cast<float4*>(D + (size_t)j * pitchD + ib) =
            make_float4(fmaf(-2.f, g[0], xi.x), fmaf(-2.f, g[1], xi.y), fmaf(-2.f, g[2], xi.z), fmaf(-2.f, g[3], xi.w));
    }
}

// ------------------------------------------------------------------ per-node GEMMs, gemm-wave gw of GW
// A wave owns up to two 16-node row tiles (rt = gw, gw + GW) and computes every 16-channel column
// tile of [a | b] for them:  a = x.W1'  -> A (LDS, the gather target);  b = x.(W2-W1)' + t replaces
// the wave's own rows of X in place (fp32, bytes 0..255 of the row) once all its column tiles are done (no other
// wave reads those rows in this phase, so no barrier is needed).  Weight fragments stream from L1/L2.
template <int NKB, int COUT, int FMT>
__device__ __forceinline__ void gemm_rows(unsigned char* __restrict__ X, float* __restrict__ A, int pitchA,
                                          const unsigned short* __restrict__ Wb, const float* __restrict__ tb, int nrt,
                                          int gw, int GW) {
    const int lane = phase_tid() & 63;
    const int l15 = lane & 15, lq = lane >> 4;
    constexpr int NCA = COUT / 16;                   // a-type column tiles (= b-type column tiles)
    constexpr int NCT = 2 * NCA;
    const int rt0 = gw, rt1 = gw + GW;
    if (rt0 >= nrt) return;
    const bool two = rt1 < nrt;
    unsigned char* x0 = X + (rt0 * 16 + l15) * xrow<FMT>();
    unsigned char* x1 = X + ((two ? rt1 : rt0) * 16 + l15) * xrow<FMT>();
    FragT<FMT> w[2], wn[2];                                // weight fragments, fetched one column tile ahead (L1/L2)
    load_wfrag<NKB>(Wb, w);
    FragT<FMT> xf0[2], xf1[2];
    xload<NKB, FMT>(x0, lq, xf0);
    xload<NKB, FMT>(x1, lq, xf1);
    float4 tv[NCA];
#pragma unroll
    for (int cb = 0; cb < NCA; ++cb) tv[cb] = *reinterpret_cast<const float4*>(tb + cb * 16 + 4 * lq);
    float* a0 = A + (rt0 * 16 + l15) * pitchA + 4 * lq;
    float* a1 = A + ((two ? rt1 : rt0) * 16 + l15) * pitchA + 4 * lq;
    f32x4 b0[NCA], b1[NCA];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
        if (ct + 1 < NCT) load_wfrag<NKB>(Wb + (size_t)(ct + 1) * wtile<NKB, FMT>(), wn);
        // r[c] = out[channel ct*16 + 4lq + c][node rt*16 + l15]
        const f32x4 r0 = tile16<NKB>(w, xf0);
        f32x4 r1 = r0;
        if (two) r1 = tile16<NKB>(w, xf1);
        if (ct < NCA) {
            *reinterpret_cast<float4*>(a0 + ct * 16) = make_float4(r0[0], r0[1], r0[2], r0[3]);
            if (two) *reinterpret_cast<float4*>(a1 + ct * 16) = make_float4(r1[0], r1[1], r1[2], r1[3]);
        } else {
            const float4 t4 = tv[ct < NCA ? 0 : ct - NCA];
            const f32x4 t = {t4.x, t4.y, t4.z, t4.w};
            b0[ct < NCA ? 0 : ct - NCA] = r0 + t;
            b1[ct < NCA ? 0 : ct - NCA] = r1 + t;
        }
        if (ct + 1 < NCT) copy_frag<NKB>(w, wn);
    }
#pragma unroll
    for (int cb = 0; cb < NCA; ++cb) {
        *reinterpret_cast<float4*>(x0 + (cb * 16 + 4 * lq) * 4) = make_float4(b0[cb][0], b0[cb][1], b0[cb][2], b0[cb][3]);
        if (two) *reinterpret_cast<float4*>(x1 + (cb * 16 + 4 * lq) * 4) = make_float4(b1[cb][0], b1[cb][1], b1[cb][2], b1[cb][3]);
    }
}

// ------------------------------------------------------------------ per-node GEMMs of a wave's OWN row tile (embed_big_kernel)
// gemm_rows for one row tile: a = x.W1' -> A (LDS, the gather target); b = x.(W2-W1)' + t stays in registers (bout) -
// the caller stores it over the wave's rows of X once every wave has read them as candidates / operands (a barrier).
template <int NKB, int COUT, int FMT>
__device__ __forceinline__ void gemm_own(const unsigned char* __restrict__ X, float* __restrict__ A, int pitchA,
                                         const unsigned short* __restrict__ Wb, const float* __restrict__ tb, int rt,
                                         f32x4 (&bout)[4]) {
    const int lane = phase_tid() & 63;
    const int l15 = lane & 15, lq = lane >> 4;
    constexpr int NCA = COUT / 16, NCT = 2 * NCA;
    const unsigned char* x0 = X + (rt * 16 + l15) * xrow<FMT>();
    FragT<FMT> w[2], wn[2], xf[2];
    load_wfrag<NKB>(Wb, w);
    xload<NKB, FMT>(x0, lq, xf);
    float* a0 = A + (rt * 16 + l15) * pitchA + 4 * lq;
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
        if (ct + 1 < NCT) load_wfrag<NKB>(Wb + (size_t)(ct + 1) * wtile<NKB, FMT>(), wn);
        const f32x4 r0 = tile16<NKB>(w, xf);        // r0[c] = out[channel ct*16 + 4lq + c][node rt*16 + l15]
        if (ct < NCA) {
            *reinterpret_cast<float4*>(a0 + ct * 16) = make_float4(r0[0], r0[1], r0[2], r0[3]);
        } else {
            const float4 t4 = *reinterpret_cast<const float4*>(tb + (ct - NCA) * 16 + 4 * lq);
            const f32x4 t = {t4.x, t4.y, t4.z, t4.w};
            bout[ct < NCA ? 0 : ct - NCA] = r0 + t;
        }
        if (ct + 1 < NCT) copy_frag<NKB>(w, wn);
    }
}

// ------------------------------------------------------------------ per-node GEMMs, weight-stationary (capped plans)
// Small graphs (nrt <= 4 row tiles, 2..4 waves): a wave owns COLUMN tiles ct = wave, wave + NW, ... of [a | b], keeps
// the weight fragment of the tile in registers and walks the row tiles, so the weights cross L2 -> CU once per
// workgroup instead of once per row tile and the work divides evenly whatever nrt is.  a-tiles go straight to A;
// the (at most two) b-tiles of a wave wait in registers until every wave has read its X operands, then replace X.
// pre: the weight fragment of this wave's FIRST column tile (tile index `wave`: NW = 4, NCT = 4 or 8), requested by the
// caller ahead of the phase (embed_graph asks for it before the selection, whose ~2 us hide the L2 round trip that
// otherwise opens every GEMM phase); nullptr: loaded here
template <int NKB, int COUT, int FMT, int NWC = 0>   // NWC: the number of waves when it is a compile-time constant (lean instances)
__device__ __forceinline__ void gemm_cols(unsigned char* __restrict__ X, float* __restrict__ A, int pitchA,
                                          const unsigned short* __restrict__ Wb, const float* __restrict__ tb, int nrt,
                                          int wave, int NW_, int ex = 0, const FragT<FMT>* pre = nullptr) {
    const int NW = NWC ? NWC : NW_;
    const bool constw = (ex & 512) != 0;       // ablation bit 9: constant weights instead of loads (a compile-time 0 in production)
    const int lane = phase_tid() & 63;
    const int l15 = lane & 15, lq = lane >> 4;
    constexpr int NCA = COUT / 16, NCT = 2 * NCA;
    constexpr int XR = xrow<FMT>();
    const unsigned char* xp = X + l15 * XR;
    // this wave's b-tiles: the first ct >= NCA congruent to wave (mod NW), and the one after it
    int ctb0 = wave;
    while (ctb0 < NCA) ctb0 += NW;
    const int ctb1 = ctb0 + NW;
    FragT<FMT> w[2], wn[2], xf[2];      // no operand double-buffering for X: registers are the scarce resource here
    f32x4 k0[4], k1[4];
    // ---- a-tiles
    int ct = wave;
    if (pre) {                                       // (NW = 4: the wave's first tile is tile `wave`, a- or b-type)
        w[0] = pre[0];
        if (NKB != 1) w[1] = pre[1];
    } else if (ct < NCA) load_wfrag<NKB>(Wb + (size_t)ct * wtile<NKB, FMT>(), w, constw);
    else if (ctb0 < NCT) load_wfrag<NKB>(Wb + (size_t)ctb0 * wtile<NKB, FMT>(), w, constw);
    for (; ct < NCA; ct += NW) {
        const int cn = ct + NW < NCA ? ct + NW : ctb0;          // next tile of this wave (a-type, else its first b-tile)
        if (cn < NCT) load_wfrag<NKB>(Wb + (size_t)cn * wtile<NKB, FMT>(), wn, constw);
        float* ap = A + l15 * pitchA + ct * 16 + 4 * lq;
        for (int rt = 0; rt < nrt; ++rt) {
            load_xfrag<NKB>(xp + rt * 16 * XR, lq, xf);
            const f32x4 r = (ex & 1024) ? f32x4{1.f, 2.f, 3.f, 4.f} : tile16<NKB>(w, xf);              // r[c] = a[channel ct*16 + 4lq + c][node rt*16 + l15]
            *reinterpret_cast<float4*>(ap + rt * 16 * pitchA) = make_float4(r[0], r[1], r[2], r[3]);
        }
        copy_frag<NKB>(w, wn);
    }
    // ---- b-tiles: results stay in registers (row-tile loop unrolled: register arrays need static indices)
    const bool has0 = ctb0 < NCT, has1 = ctb1 < NCT;
    if (has0) {
        if (has1) load_wfrag<NKB>(Wb + (size_t)ctb1 * wtile<NKB, FMT>(), wn, constw);
        const float4 t4 = *reinterpret_cast<const float4*>(tb + (ctb0 - NCA) * 16 + 4 * lq);
        const f32x4 t = {t4.x, t4.y, t4.z, t4.w};
#pragma unroll
        for (int rt = 0; rt < 4; ++rt)
            if (rt < nrt) {
                load_xfrag<NKB>(xp + rt * 16 * XR, lq, xf);
                k0[rt] = ((ex & 1024) ? f32x4{1.f, 2.f, 3.f, 4.f} : tile16<NKB>(w, xf)) + t;
            }
    }
    if (has1) {
        const float4 t4 = *reinterpret_cast<const float4*>(tb + (ctb1 - NCA) * 16 + 4 * lq);
        const f32x4 t = {t4.x, t4.y, t4.z, t4.w};
#pragma unroll
        for (int rt = 0; rt < 4; ++rt)
            if (rt < nrt) {
                load_xfrag<NKB>(xp + rt * 16 * XR, lq, xf);
                k1[rt] = ((ex & 1024) ? f32x4{1.f, 2.f, 3.f, 4.f} : tile16<NKB>(wn, xf)) + t;
            }
    }
    if (!(ex & 2048)) __syncthreads();                                             // every wave is done reading X
    unsigned char* bp = X + l15 * XR + 16 * lq;
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
        if (rt < nrt) {
            if (has0) *reinterpret_cast<float4*>(bp + rt * 16 * XR + (ctb0 - NCA) * 64) = make_float4(k0[rt][0], k0[rt][1], k0[rt][2], k0[rt][3]);
            if (has1) *reinterpret_cast<float4*>(bp + rt * 16 * XR + (ctb1 - NCA) * 64) = make_float4(k1[rt][0], k1[rt][1], k1[rt][2], k1[rt][3]);
        }
}

template <bool COLS, int FMT, int NWC = 0>
__device__ __forceinline__ void gemm_layer(unsigned char* X, float* A, int pitchA, const unsigned short* Wb,
                                           const float* tb, int Kp, int cout, int nrt, int gw, int GW, int ex = 0,
                                           const FragT<FMT>* pre = nullptr) {
    if (COLS) {   // capped plans: nrt <= 4, GW >= 2 (make_embed_plan); contains a barrier - every wave calls it
        if (Kp != 64)
            gemm_cols<1, 64, FMT, NWC>(X, A, pitchA, Wb, tb, nrt, gw, GW, ex, pre);
        else if (cout == 64)
            gemm_cols<4, 64, FMT, NWC>(X, A, pitchA, Wb, tb, nrt, gw, GW, ex, pre);
        else
            gemm_cols<4, 32, FMT, NWC>(X, A, pitchA, Wb, tb, nrt, gw, GW, ex, pre);
        return;
    }
    if (Kp != 64)
        gemm_rows<1, 64, FMT>(X, A, pitchA, Wb, tb, nrt, gw, GW);   // first layer of a branch: 3 / 12 -> 64 channels
    else if (cout == 64)
        gemm_rows<4, 64, FMT>(X, A, pitchA, Wb, tb, nrt, gw, GW);
    else
        gemm_rows<4, 32, FMT>(X, A, pitchA, Wb, tb, nrt, gw, GW);
}

// ------------------------------------------------------------------ coordinate-space keys (first layer of the xyz branch)
// Node centres reach +-50 m, so the expansion |x_j|^2 - 2<x_i, x_j> cancels numbers near 5000 to rank distances near
// 300: on two f16 planes (22 bits) its error is about 1e-3, in the reference's fp32 about 5e-4, and candidates closer
// than that at the k-th place do occur (1.1e-4 apart in a fuzzed graph of 135 nodes, where the planes picked the other
// one).  With three coordinates the reference's own arithmetic is five fp32 operations per pair, so this layer restates
// it operation for operation on the vector ALU (model/layers_batch.py:8-12 / dgcnn.py:14-20 as torch evaluates them:
// the K=3 matmul is the FMA chain x, y, z; |x|^2 = (x*x + y*y) + z*z with every step rounded; pd = (-|x_j|^2 -
// inner) - |x_i|^2) and ranks by -pd: bit for bit the reference's keys, hence its neighbour sets even between
// near-ties (tools/exp/fuzz_oracle.py; the operation order is pinned against torch in tests/test_oracle_golden.py::test_coordinate_keys_operation_order).  (x, y, z, |x|^2)
// of each slot sit in the 16 pad bytes at the end of its X row (written by the staging of this layer; an empty slot
// carries |x|^2 = +inf, which makes its key +inf in every row: invalid candidates rank last); four lanes per row, four
// candidates per 16-byte store.
template <int FMT>
__device__ __forceinline__ void gram_xyz_direct(const unsigned char* __restrict__ X, float* __restrict__ D, int pitchD, int N,
                                                int NP, int rc0, int rows_chunk, int tid, int NT) {
    constexpr int XR = xrow<FMT>(), OFF = XR - 16;
    const int part = tid & 3;
    for (int r = tid >> 2; r < rows_chunk; r += NT >> 2) {
        const float4 ci = *reinterpret_cast<const float4*>(X + (rc0 + r) * XR + OFF);
        float* drow = D + r * pitchD;
        for (int j0 = 4 * part; j0 < NP; j0 += 16) {
            float d[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float4 cj = *reinterpret_cast<const float4*>(X + (j0 + u) * XR + OFF);
                const float dot = fmaf(ci.z, cj.z, fmaf(ci.y, cj.y, __fmul_rn(ci.x, cj.x)));
                const float t = fmaf(2.f, dot, -cj.w);                       // -|x_j|^2 - inner,  inner = -2 dot (exact)
                d[u] = __fsub_rn(ci.w, t);                                   // -pd (+inf for an invalid candidate: its |x|^2 is)
            }
            *reinterpret_cast<float4*>(drow + j0) = make_float4(d[0], d[1], d[2], d[3]);
        }
    }
}

// ------------------------------------------------------------------ Gram phase (whole key matrix resident)
// upper-triangular tile t (row-major) -> (ti, tj), wave-uniform
__device__ __forceinline__ void tri_decode(int t, int n, int& ti, int& tj) {
    int r = 0;
    while (t >= n - r) {
        t -= n - r;
        ++r;
    }
    ti = r;
    tj = r + t;
}

template <int NKB, int FMT, bool PF, int NWC = 0>   // PF: fetch the next tile's operands during this tile's MFMAs (48 more VGPRs)
__device__ __forceinline__ void gram_tiles_sym(const unsigned char* __restrict__ X, const float* __restrict__ xx,
                                               float* __restrict__ D, int pitchD, int N, int nrt, int wave) {
    const int lane = phase_tid() & 63;
    const int l15 = lane & 15, lq = lane >> 4;
    const int ntiles = nrt * (nrt + 1) / 2;
    int t = wave;
    if (t >= ntiles) return;
    int ti, tj;
    tri_decode(t, nrt, ti, tj);
    FragT<FMT> a[2], b[2];
    xload<NKB, FMT>(X + (ti * 16 + l15) * xrow<FMT>(), lq, a);
    xload<NKB, FMT>(X + (tj * 16 + l15) * xrow<FMT>(), lq, b);
    while (true) {
        const int tn = t + (NWC ? NWC : (int)(blockDim.x >> 6));
        const bool more = tn < ntiles;
        int tin = 0, tjn = 0;
        FragT<FMT> an[2], bn[2];
        if (more) {                                   // operands of the next tile in flight during this tile's MFMAs
            tri_decode(tn, nrt, tin, tjn);
            if (PF) {
                xload<NKB, FMT>(X + (tin * 16 + l15) * xrow<FMT>(), lq, an);
                xload<NKB, FMT>(X + (tjn * 16 + l15) * xrow<FMT>(), lq, bn);
            }
        }
        const f32x4 g = tile16<NKB>(a, b);            // g[r] = <x_{i0+4lq+r}, x_{j0+l15}>
        const int i0 = ti * 16, j = tj * 16 + l15;
        const float xj = j < N ? xx[j] : INFINITY;    // invalid candidates rank last
        float* drow = D + (i0 + 4 * lq) * pitchD + j;
#pragma unroll
        for (int r = 0; r < 4; ++r) drow[r * pitchD] = fmaf(-2.f, g[r], xj);
        if (tj != ti) {                               // the transposed tile: row j, candidates i0+4lq..+3 (one 16-B store)
            const int ib = i0 + 4 * lq;
            float4 xi = *reinterpret_cast<const float4*>(xx + ib);
            xi.x = (ib + 0 < N) ? xi.x : INFINITY;
            xi.y = (ib + 1 < N) ? xi.y : INFINITY;
            xi.z = (ib + 2 < N) ? xi.z : INFINITY;
            xi.w = (ib + 3 < N) ? xi.w : INFINITY;
            *reinterpret_cast<float4*>(D + (size_t)j * pitchD + ib) = make_float4(
                fmaf(-2.f, g[0], xi.x), fmaf(-2.f, g[1], xi.y), fmaf(-2.f, g[2], xi.z), fmaf(-2.f, g[3], xi.w));
        }
        if (!more) break;
        if (PF) {
            copy_frag<NKB>(a, an);
            copy_frag<NKB>(b, bn);
        } else {
            xload<NKB, FMT>(X + (tin * 16 + l15) * xrow<FMT>(), lq, a);
            xload<NKB, FMT>(X + (tjn * 16 + l15) * xrow<FMT>(), lq, b);
        }
        t = tn;
        ti = tin;
        tj = tjn;
    }
}

// ------------------------------------------------------------------ gather-max over the k neighbours
// cout/4 lanes own one row (4 channels each, 16-B LDS reads); nw = the row's u16 offsets, two per word.
// Two rows per call so that twice as many independent LDS reads are in flight.
// two maxima in one instruction; inline asm because fmaxf(fmaxf()) first canonicalises every LDS-loaded operand
// (operands come from ds_read, never from an MFMA: no hazard the assembler-level scheduler would have to know about)
__device__ __forceinline__ float max3(float a, float b, float c) {
    float d;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}

#ifndef SGPR_GATHER_GROUPS
#define SGPR_GATHER_GROUPS 1  // a row's 16 gather lanes = one lane group of the LDS's ds_read_b128 schedule (0: 16 consecutive lanes;
                              // same-box A/B, twice: config 5 560.6 / 556.1 -> 549.9 / 546.3 us, KITTI-00 147.1 / 145.6 -> 147.4 / 146.6)
#endif
#ifndef SGPR_GATHER_U16
#define SGPR_GATHER_U16 0     // 1: neighbour offsets as sixteen-bit LDS reads - the load-store vectorizer fuses the pair back into a 32-bit read + unpack (measured: no ds_read_u16 is emitted), so the plain form stays
#endif
__device__ __forceinline__ void nbr_pair(const unsigned short* __restrict__ nw, int q, bool odd, int& a0, int& a1) {
#if SGPR_GATHER_U16
    // the offsets come out of the LDS pipe ready to use: the vector ALU is what this kernel is short of
    // (the second pointer goes through an opaque copy: the load-store vectorizer would otherwise fuse the pair back into one
    // 32-bit read + v_and / v_bfe)
    const unsigned short *p0 = nw + 2 * q, *p1 = nw + 2 * q + 1;
    asm("" : "+v"(p0));
    asm("" : "+v"(p1));
    a0 = *p0;
    a1 = odd ? (int)*p1 : a0;
#else
    const uint32_t w = reinterpret_cast<const uint32_t*>(nw)[q];
    a0 = w & 0xffffu;
    a1 = odd ? (int)(w >> 16) : a0;
#endif
}

__device__ __forceinline__ void gather_max2(const float* __restrict__ A4, const unsigned short* __restrict__ nwa,
                                            const unsigned short* __restrict__ nwb, int k, float4& ma, float4& mb) {
    ma = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    mb = ma;
    const int kw = (k + 1) >> 1;
#pragma unroll 5
    for (int q = 0; q < kw; ++q) {
        const bool odd = 2 * q + 1 < k;
        int a0, a1, b0, b1;
        nbr_pair(nwa, q, odd, a0, a1);
        nbr_pair(nwb, q, odd, b0, b1);
        const float4 va0 = *reinterpret_cast<const float4*>(A4 + a0);
        const float4 va1 = *reinterpret_cast<const float4*>(A4 + a1);
        const float4 vb0 = *reinterpret_cast<const float4*>(A4 + b0);
        const float4 vb1 = *reinterpret_cast<const float4*>(A4 + b1);
        ma.x = max3(ma.x, va0.x, va1.x);
        ma.y = max3(ma.y, va0.y, va1.y);
        ma.z = max3(ma.z, va0.z, va1.z);
        ma.w = max3(ma.w, va0.w, va1.w);
        mb.x = max3(mb.x, vb0.x, vb1.x);
        mb.y = max3(mb.y, vb0.y, vb1.y);
        mb.z = max3(mb.z, vb0.z, vb1.z);
        mb.w = max3(mb.w, vb0.w, vb1.w);
    }
}

// one row (the last, unpaired group of a wave)
__device__ __forceinline__ void gather_max1(const float* __restrict__ A4, const unsigned short* __restrict__ nwa, int k, float4& ma) {
    ma = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    const int kw = (k + 1) >> 1;
#pragma unroll 5
    for (int q = 0; q < kw; ++q) {
        const bool odd = 2 * q + 1 < k;
        int a0, a1;
        nbr_pair(nwa, q, odd, a0, a1);
        const float4 va0 = *reinterpret_cast<const float4*>(A4 + a0);
        const float4 va1 = *reinterpret_cast<const float4*>(A4 + a1);
        ma.x = max3(ma.x, va0.x, va1.x);
        ma.y = max3(ma.y, va0.y, va1.y);
        ma.z = max3(ma.z, va0.z, va1.z);
        ma.w = max3(ma.w, va0.w, va1.w);
    }
}

__device__ __forceinline__ float4 add_lrelu(float4 m, float4 b, bool live) {
    float4 y = make_float4(m.x + b.x, m.y + b.y, m.z + b.z, m.w + b.w);
    // LeakyReLU(0.2) = max(y, 0.2 y): two instructions per channel instead of compare / multiply / select
    y.x = fmaxf(y.x, 0.2f * y.x);
    y.y = fmaxf(y.y, 0.2f * y.y);
    y.z = fmaxf(y.z, 0.2f * y.z);
    y.w = fmaxf(y.w, 0.2f * y.w);
    return live ? y : make_float4(0.f, 0.f, 0.f, 0.f);
}

// ------------------------------------------------------------------ the semantic branch on label super-nodes
// workgroup barrier, or - for a group that is ONE wave - nothing but a compiler fence: the LDS serves the accesses of a
// wave in program order, so a lane's read sees every earlier write of any lane of its own wave
template <bool WAVE>
__device__ __forceinline__ void group_sync() {
    if constexpr (WAVE) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else {
        __syncthreads();
    }
}

// Layers 1-3 of the semantic branch on the 13 label super-nodes (see embed_graph): cnt[0..11] = nodes per label,
// cnt[12] = K.  Runs on a thread group of NT threads = NW waves that owns X rows 0..15, A / D (16 rows each; D may alias
// A unless LEAN), xx[16], vmask[16]: either the whole workgroup of a graph (group_sync = barrier) or a single wave
// (WAVE: NT = 64, NW = 1, the "semantic waves" of the split launch).  Result: sem3 rows 0..15 -> out[16][PP] (LDS or
// global).  Every value comes from the instructions of the generic path on the same operands.
template <int FMT, int LEAN, bool WAVE>
__device__ __forceinline__ void supernode_branch(const DevWeights& w, const int k0, const int skip, unsigned char* X,
                                                 float* xx, float* A, float* D, const int pitchA, const int pitchD,
                                                 const int* cnt, int* vmask, float* out, const int tid, const int wave,
                                                 const int NT, const int NW, float& vmax) {
    constexpr int XROW = xrow<FMT>();
    const int lane = tid & 63;
    // Layer 1 and the matrix products of layer 2 see the graph only through the bits "fewer than K nodes carry label v":
    // with the tables of sem_table_kernel (same instructions, run once at sgpr_create) layer 1 costs nothing per graph
    // and layer 2 is its selection and its gather.
    const bool tabled = FMT == FMT_H2 && w.sem_g != nullptr && !(skip & (65536 | 262144));
    // lean instance: layer 3's matrix products run on waves 1 .. 3 (wave 0 takes the Gram tile); the weight fragment of a
    // wave's first column tile is requested HERE, two phases ahead (its L2 round trip otherwise opens the GEMM phase)
    constexpr bool kPre3 = SGPR_WPREFETCH != 0 && LEAN == 64 && FMT == FMT_H2 && !WAVE && SGPR_LEAN_WAVES == 4;
    FragT<FMT> wpre3[2];
    if constexpr (kPre3) {
        if (wave >= 1 && !(skip & 16384)) load_wfrag<4>(w.wh[2] + (size_t)(wave - 1) * wtile<4, FMT>(), wpre3);
    }
    // layer 1: the table, straight into X rows 0..15 (rows 13..15 zero)
    const float* wf0 = w.wf[0];                                         // [2 * 64][16] folded fp32 weights
    const float* tb0 = w.tb[0];
    for (int t = ((skip & 65536) || tabled) ? 256 : tid; t < 16 * 16; t += NT) {   // (ablation bit 16: no layer-1 table)
        const int v = t >> 4, c4 = (t & 15) * 4;
        float4 a4 = make_float4(0.f, 0.f, 0.f, 0.f), b4 = *reinterpret_cast<const float4*>(tb0 + c4);
        bool with_rep = true;
        if (v < kLabels) {
            // the weight as the generic path's matrix product sees it: exact in three bf16 planes, hi + lo of
            // the two f16 planes (22 bits) in FMT_H2
            auto wq = [](float x) {
                if (FMT != FMT_H2) return x;
                const _Float16 h = (_Float16)x;
                return (float)h + (float)(_Float16)(x - (float)h);
            };
            a4 = make_float4(wq(wf0[(c4 + 0) * 16 + v]), wq(wf0[(c4 + 1) * 16 + v]), wq(wf0[(c4 + 2) * 16 + v]),
                             wq(wf0[(c4 + 3) * 16 + v]));
            b4 = make_float4(wq(wf0[(64 + c4 + 0) * 16 + v]) + b4.x, wq(wf0[(64 + c4 + 1) * 16 + v]) + b4.y,
                             wq(wf0[(64 + c4 + 2) * 16 + v]) + b4.z, wq(wf0[(64 + c4 + 3) * 16 + v]) + b4.w);
            with_rep = cnt[v] < k0;
        }
        const float z = with_rep ? 0.f : -INFINITY;                        // the representative's a is 0
        const float4 m4 = make_float4(max3(-INFINITY, a4.x, z), max3(-INFINITY, a4.y, z),
                                      max3(-INFINITY, a4.z, z), max3(-INFINITY, a4.w, z));
        const float4 y = add_lrelu(m4, b4, v <= kLabels);
        xstore<FMT>(X + v * XROW, c4, y, vmax);
        float sa = fmaf(y.x, y.x, fmaf(y.y, y.y, fmaf(y.z, y.z, y.w * y.w)));
        sa += lane_xor(sa, 1);
        sa += lane_xor(sa, 2);
        sa += lane_xor(sa, 4);
        sa += lane_xor(sa, 8);
        if ((t & 15) == 0) xx[v] = sa;
    }
    group_sync<WAVE>();
    // layers 2 and 3 on the 13 virtual rows
    for (int Lv = (skip & 16384) ? 3 : 1; Lv < 3; ++Lv) {      // (ablation bit 14: super-node layers 2 and 3 off)
        const int cout = w.cout[Lv];
        const unsigned short* wl = FMT == FMT_H2 ? w.wh[Lv] : w.wb[Lv];
        // the keys of the 16 virtual rows sit beside the 16 rows of A, so that in the lean instance the Gram tile
        // (wave 0) and the a / b column tiles (the other waves) run side by side: three barriers per layer
        float* Dv = LEAN != 0 ? A + 16 * pitchA : D;
        const bool tab2 = tabled && Lv == 1;                           // layer 2: keys, a and b come from the tables
        if (tab2) {
        } else if (LEAN != 0) {
            if (wave == 0) {
                gram_tiles_sym<4, FMT, false>(X, xx, Dv, pitchD, kLabels + 1, 1, 0);
                group_sync<WAVE>();                                           // = the barrier inside gemm_cols
            } else {
                gemm_layer<true, FMT, (LEAN > 0 ? SGPR_LEAN_WAVES - 1 : 0)>(X, A, pitchA, wl, w.tb[Lv], 64, cout, 1, wave - 1, NW - 1, 0,
                                                           (kPre3 && Lv == 2) ? wpre3 : nullptr);   // (lean: four waves; LEAN < 0: the big instance's)
            }
        } else {
            gram_tiles_sym<4, FMT, false>(X, xx, Dv, pitchD, kLabels + 1, 1, wave);
            group_sync<WAVE>();
        }
        float4 b2row = make_float4(0.f, 0.f, 0.f, 0.f);                  // tab2: this thread's piece of its row's b term
        for (int t = tid; t < 16 * 16; t += NT) {                          // row l = t >> 4, candidate j = t & 15
            const int l = t >> 4, j = t & 15;
            const int cj = j <= kLabels ? cnt[j] : 0;
            float key = INFINITY;
            if (tab2) {
                const int bl = (l >= kLabels || cnt[l] < k0) ? 1 : 0, bj = (j >= kLabels || cj < k0) ? 1 : 0;
                if (cj > 0) key = fmaf(-2.f, w.sem_g[((bl * 2 + bj) * 16 + l) * 16 + j], w.sem_xx[bj * 16 + j]);
                if constexpr (SGPR_SEM_STAGE) {
                    // this graph's version of every table row, fetched in the same round trip as the keys: the a rows go to
                    // A rows 0..15 (free until layer 3), where the gather below finds them like the generic path's - a
                    // read of the table inside its data-dependent loop would be one L2 round trip per neighbour label
                    const int c4 = j * 4;
                    *reinterpret_cast<float4*>(A + l * pitchA + c4) =
                        *reinterpret_cast<const float4*>(w.sem_a2 + ((bl ? 16 : 0) + l) * 64 + c4);
                    if (NT >= 256) b2row = *reinterpret_cast<const float4*>(w.sem_b2 + ((bl ? 16 : 0) + l) * 64 + c4);
                }
            } else if (cj > 0) {
                key = Dv[l * pitchD + j];
            }
            int before = 0;                                                // nodes ranked ahead of label j
            // the 16 candidates of a row sit in one 16-lane DPP row (t is a multiple of NT >= 64 away from the
            // lane id): candidate (j + s) & 15 arrives by a row rotation - two v_mov_dpp instead of two
            // ds_bpermute round trips per step; every lane of the wave is active here (256 % 64 == 0)
            const unsigned okey = ord_u32(key);                            // (-0 == +0; +inf for labels the graph lacks)
            before = count_before<1>(okey, (unsigned)cj | ((unsigned)j << 16),
                                     ((unsigned long long)okey << 32) | ((unsigned)j << 16), before);
#if SGPR_EXP_DOUBLE & 1
            {
                unsigned ok2 = okey;
                asm volatile("" : "+v"(ok2));
                const int b2 = count_before<1>(ok2, (unsigned)cj | ((unsigned)j << 16),
                                               ((unsigned long long)ok2 << 32) | ((unsigned)j << 16), 0);
                before = min(before, b2);
            }
#endif
            const unsigned long long inc = __ballot(cj > 0 && before < k0);
            if (j == 0) vmask[l] = (int)((inc >> (lane & 48)) & 0xffffull);
        }
        group_sync<WAVE>();                                                   // keys consumed: A may overwrite D
        if (LEAN == 0 && !tab2) {
            if constexpr (WAVE) {
                // (one wave, one row tile: the one-tile form - this instance shares the owned-rows kernel's 128 registers)
                f32x4 bo[4];
                if (cout == 64)
                    gemm_own<4, 64, FMT>(X, A, pitchA, wl, w.tb[Lv], 0, bo);
                else
                    gemm_own<4, 32, FMT>(X, A, pitchA, wl, w.tb[Lv], 0, bo);
                unsigned char* x0 = X + (lane & 15) * XROW;
#pragma unroll
                for (int cb = 0; cb < 4; ++cb)
                    if (cb < (cout >> 4))
                        *reinterpret_cast<float4*>(x0 + (cb * 16 + 4 * (lane >> 4)) * 4) = make_float4(bo[cb][0], bo[cb][1], bo[cb][2], bo[cb][3]);
            } else {
                gemm_layer<false, FMT>(X, A, pitchA, wl, w.tb[Lv], 64, cout, 1, wave, NW, 0);
            }
            group_sync<WAVE>();
        }
        const int lpr = cout >> 2;                                         // lanes per row: 16 or 8
        const int lsh = cout == 64 ? 4 : 3;                                // (a shift, not an integer division by lpr)
        for (int t = tid; t < 16 * lpr; t += NT) {
            const int l = t >> lsh, c4 = (t & (lpr - 1)) * 4;
            int mask = vmask[l];                                           // a few labels per row: walk the set bits
            float4 m4 = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
            while (mask) {
                const int j = __builtin_ctz(mask);
                mask &= mask - 1;
                const float* arow = (tab2 && !SGPR_SEM_STAGE) ? w.sem_a2 + (((j >= kLabels || cnt[j] < k0) ? 16 : 0) + j) * 64
                                                              : A + j * pitchA;
                const float4 v = *reinterpret_cast<const float4*>(arow + c4);
                m4.x = kmax(m4.x, v.x);
                m4.y = kmax(m4.y, v.y);
                m4.z = kmax(m4.z, v.z);
                m4.w = kmax(m4.w, v.w);
            }
            const float* brow = tab2 ? w.sem_b2 + (((l >= kLabels || cnt[l] < k0) ? 16 : 0) + l) * 64
                                     : reinterpret_cast<const float*>(X + l * XROW);
            // (tab2 on >= 256 threads: the piece was fetched with the keys - same (row, channels) per thread in both loops)
            const float4 bv = (tab2 && SGPR_SEM_STAGE && NT >= 256) ? b2row : *reinterpret_cast<const float4*>(brow + c4);
            const float4 y = add_lrelu(m4, bv, l <= kLabels);
            if (Lv == 1) {
                xstore<FMT>(X + l * XROW, c4, y, vmax);
                float sa = fmaf(y.x, y.x, fmaf(y.y, y.y, fmaf(y.z, y.z, y.w * y.w)));
                sa += lane_xor(sa, 1);
                sa += lane_xor(sa, 2);
                sa += lane_xor(sa, 4);
                sa += lane_xor(sa, 8);
                if ((t & 15) == 0) xx[l] = sa;
            } else {
                *reinterpret_cast<float4*>(out + (size_t)l * PP + c4) = y;
            }
        }
        group_sync<WAVE>();
    }
}

// ------------------------------------------------------------------ split launch (few graphs: the latency regime)
// The two branches of dgcnn_conv_pass are independent until conv_end (sg_net.py:81-104).  When a launch has at most half
// as many graphs as the GPU has CUs, every graph gets TWO workgroups, each on a CU of its own: workgroup s < G runs the
// input phase and the semantic branch (role 1) and publishes the 16 sem3 rows through global memory (sem_tab[s] +
// sem_flag[s] = token(s)); workgroup G + s runs the input phase and the xyz branch, meets the rows of the other half
// before conv_end and finishes the graph (role 2).  The producers own the lower block indices, so a waiting workgroup is
// never dispatched ahead of its producer; on an idle GPU all 2 G workgroups are resident at once.  Should a flag
// nevertheless not arrive in time (another stream or process holding CUs), the waiting workgroup hands its graph to the
// second pass (request_redo) instead of hanging or failing.  With more graphs the halves would share CUs, which costs each more than the
// split saves (launch_embed), and in the throughput regime the kernel's time is its instruction count: every workgroup
// does both branches (role 0).  Measured and dropped on the way here: the semantic branch of EVERY launch on single
// waves ("semantic waves", 4 graphs per leading workgroup): the same instructions at lower parallelism, 197 -> 303 us.
// A launch slot's second-pass request (EmbedArgs::redo: 1 wide-range instance, 2 full f16 plan) - and, with any request,
// this launch's token in redo_count: the second pass reads that ONE word and returns when no workgroup stored it, so
// the (normal) empty pass costs a launch and a load instead of a scan of every flag.  The word is never reset: a token
// belongs to one launch (launch_embed's counter), whatever the workspace held before is some other value.
__device__ __forceinline__ void request_redo(const EmbedArgs& a, int slot, unsigned char why) {
    a.redo[slot] = why;
    if (why && a.redo_count) __hip_atomic_store(a.redo_count, a.sem_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__host__ __device__ __forceinline__ unsigned long long sem_token(unsigned epoch, int slot) {
    return ((unsigned long long)epoch << 32) | (unsigned)(slot + 1);   // (bit 31: an f16 overflow in the branch)
}

// ------------------------------------------------------------------ graph-independent part of the super-node branch
// One workgroup of 256 threads, once per handle (sgpr_create).  Row set p (p = 1: "with the representative", p = 0:
// without) of layer 1 -> X rows 16 p + v by the code of supernode_branch's first loop; layer 2's Gram tile for the four
// combinations of row sets and its a / b terms for both sets by tile16 / gemm_rows - the instructions the per-graph path
// would execute on the same operands, so that a table entry carries the bits the per-graph path would produce.
__global__ __launch_bounds__(256) void sem_table_kernel(const DevWeights w, float* __restrict__ table, float* __restrict__ vmax_out) {
    constexpr int FMT = FMT_H2;
    constexpr int XROW = xrow<FMT>();
    __shared__ __attribute__((aligned(16))) unsigned char X[32 * XROW];
    __shared__ __attribute__((aligned(16))) float A[32 * 68];
    __shared__ float xx[32];
    __shared__ float vm[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, lq = lane >> 4;
    float vmax = 0.f;
    const float* wf0 = w.wf[0];
    const float* tb0 = w.tb[0];
    for (int t2 = tid; t2 < 2 * 256; t2 += 256) {
        const int p = t2 >> 8, t = t2 & 255;
        const int v = t >> 4, c4 = (t & 15) * 4;
        float4 a4 = make_float4(0.f, 0.f, 0.f, 0.f), b4 = *reinterpret_cast<const float4*>(tb0 + c4);
        bool with_rep = true;
        if (v < kLabels) {
            auto wq = [](float x) {
                const _Float16 h = (_Float16)x;
                return (float)h + (float)(_Float16)(x - (float)h);
            };
            a4 = make_float4(wq(wf0[(c4 + 0) * 16 + v]), wq(wf0[(c4 + 1) * 16 + v]), wq(wf0[(c4 + 2) * 16 + v]),
                             wq(wf0[(c4 + 3) * 16 + v]));
            b4 = make_float4(wq(wf0[(64 + c4 + 0) * 16 + v]) + b4.x, wq(wf0[(64 + c4 + 1) * 16 + v]) + b4.y,
                             wq(wf0[(64 + c4 + 2) * 16 + v]) + b4.z, wq(wf0[(64 + c4 + 3) * 16 + v]) + b4.w);
            with_rep = p == 1;
        }
        const float z = with_rep ? 0.f : -INFINITY;
        const float4 m4 = make_float4(max3(-INFINITY, a4.x, z), max3(-INFINITY, a4.y, z), max3(-INFINITY, a4.z, z),
                                      max3(-INFINITY, a4.w, z));
        const float4 y = add_lrelu(m4, b4, v <= kLabels);
        xstore<FMT>(X + (16 * p + v) * XROW, c4, y, vmax);
        float sa = fmaf(y.x, y.x, fmaf(y.y, y.y, fmaf(y.z, y.z, y.w * y.w)));
        sa += lane_xor(sa, 1);
        sa += lane_xor(sa, 2);
        sa += lane_xor(sa, 4);
        sa += lane_xor(sa, 8);
        if ((t & 15) == 0) xx[16 * p + v] = sa;
    }
    __syncthreads();
    float* tg = table;                 // [2][2][16][16]
    float* txx = tg + 4 * 256;         // [2][16]
    float* ta = txx + 32;              // [2][16][64]
    float* tb = ta + 2 * 16 * 64;      // [2][16][64]
    {   // Gram: wave = (row set of the A operand, row set of the B operand)
        const int pa = wave >> 1, pb = wave & 1;
        FragT<FMT> a[2], b[2];
        xload<4, FMT>(X + (16 * pa + l15) * XROW, lq, a);
        xload<4, FMT>(X + (16 * pb + l15) * XROW, lq, b);
        const f32x4 g = tile16<4>(a, b);              // g[r] = <x_{4 lq + r} (set pa), x_{l15} (set pb)>
#pragma unroll
        for (int r = 0; r < 4; ++r) tg[((pa * 2 + pb) * 16 + 4 * lq + r) * 16 + l15] = g[r];
    }
    if (tid < 32) txx[tid] = xx[tid];
    __syncthreads();
    gemm_rows<4, 64, FMT>(X, A, 68, w.wh[1], w.tb[1], 2, wave, 4);   // row tile = row set (waves 0 and 1)
    __syncthreads();
    for (int e = tid; e < 32 * 16; e += 256) {
        const int r = e >> 4, c4 = (e & 15) * 4;
        *reinterpret_cast<float4*>(ta + r * 64 + c4) = *reinterpret_cast<const float4*>(A + r * 68 + c4);
        *reinterpret_cast<float4*>(tb + r * 64 + c4) = *reinterpret_cast<const float4*>(X + r * XROW + 4 * c4);
    }
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, m));
    if (lane == 0) vm[wave] = vmax;
    __syncthreads();
    if (tid == 0) *vmax_out = fmaxf(fmaxf(vm[0], vm[1]), fmaxf(vm[2], vm[3]));
}

int launch_sem_tables(const DevWeights& w, float* d_table, float* d_vmax, hipStream_t stream) {
    hipLaunchKernelGGL(sem_table_kernel, dim3(1), dim3(256), 0, stream, w, d_table, d_vmax);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "sem_table_kernel launch");
    return SGPR_OK;
}

// DBG = true: the instrumented build used by sgpr_embed_debug / the profiling and ablation hooks; the
// production instance carries none of that code.
// LEAN (64 / 48): the instance of the fixed lean layouts - kLeanNT = 256 threads, four / five workgroups per CU (91 / 79 VGPRs)
// KC: K as a compile-time constant (10, the reference's, in the lean production instances; 0 = read from the plan): the
// K-derived loop bounds and predicates of every phase fold away
// BIG: the owned-rows instance for plans beyond 64 rows (embed_big_kernel; production only: DBG = 0, LEAN = 0, FMT_H2, KC = 10 / 20)
template <int KP, int DBG, int LEAN, int FMT, int KC = 0, bool BIG = false>   // DBG: 0 production, 1 phase timers + ablation mask, 2 + layer / kNN dumps
__device__ __forceinline__ void embed_graph(const KParams& kp, const EmbedPlan& plan_in, const int g, const int launch_slot,
                                            const int role = 0,     // role: 0 whole graph, 1 / 2 the halves of a split launch
                                            const int g_ahead = -1) {   // packed input: the graph whose lines are pulled into L2 (see below)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // lean instance: the layout fields are the constants of lean_fixed_layout (the host built the plan from the same
    // function); N, NC, k stay run-time values
    EmbedPlan plan_local = plan_in;
    if constexpr (LEAN != 0) lean_fixed_layout(plan_local, DBG == 0, LEAN);
    const EmbedPlan& p = plan_local;
    float vmax = 0.f;                                    // FMT_H2: largest magnitude stored into the f16 planes
    const int NT = LEAN != 0 ? kLeanNT : (int)blockDim.x, NW = NT >> 6;   // 64 .. 512 threads (EmbedPlan::nt); lean: a constant
    unsigned char* X = smem + p.offX;                    // [NP][XROW]: bf16 planes (or fp32 rows) / in-place fp32 b
    constexpr int XROW = xrow<FMT>();
    float* A = reinterpret_cast<float*>(smem + p.offA);
    float* D = reinterpret_cast<float*>(smem + p.offD);
    float* xx = reinterpret_cast<float*>(smem + p.offXX);
    float* red = reinterpret_cast<float*>(smem + p.offRed);
    unsigned short* nbr = reinterpret_cast<unsigned short*>(smem + p.offIdx);
    const int tid0 = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);   // scalar: wave-uniform task loops and branches
    int tid = tid0, lane = tid & 63, l15 = lane & 15, lq = lane >> 4;
    const int NS = p.N;                                   // slots per graph in global memory
    // the first branch's output: LDS, or (large plans) this graph's rows of the global workspace - unless the branch runs
    // on the 16 super-node rows, which every plan holds in LDS (re-pointed below once that is known)
    float* park = p.park_in_lds ? reinterpret_cast<float*>(smem + p.offPark) : kp.a.park_ws + (size_t)g * p.NP * PP;
    // optional per-phase cycle accounting (thread 0 of every workgroup; phases end at barriers)
    unsigned long long t_prev = 0;
    unsigned long long* const prof_buf = DBG ? kp.a.prof : nullptr;
#ifndef SGPR_EXP_SKIP
#define SGPR_EXP_SKIP 0       // timing experiments only (tools/build_variant.sh): an ablation mask compiled into the production instance
#endif
    const int skip = DBG ? kp.a.skip : SGPR_EXP_SKIP;
    float* const dbg_layers = DBG == 2 ? kp.a.dbg_layers : nullptr;
    int32_t* const dbg_knn_all = DBG == 2 ? kp.a.dbg_knn : nullptr;
    // timers live in scalar registers of wave 0 and reach memory once, at the end of the kernel: per-phase global
    // atomics from 4541 workgroups onto 8 addresses would triple the kernel time
    const bool prof = prof_buf != nullptr && __builtin_amdgcn_readfirstlane(wave) == 0;
    unsigned long long pacc[8] = {0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull};
    if (prof) t_prev = clock64();
#define SGPR_PROF(ph)                                            \
    if (prof) {                                                  \
        const unsigned long long t_now = clock64();              \
        pacc[ph] += t_now - t_prev;                              \
        t_prev = t_now;                                          \
    }

    if (skip & 16) return;   // ablation: pure dispatch cost
    // ---- L2 prefetch for the workgroup that will take this one's place: workgroups reach the XCDs round-robin by block
    //      index, so launch slot s + kAhead (a multiple of 8, one round of resident workgroups later) runs on THIS XCD about
    //      one workgroup lifetime from now; one load per 128-byte line of its graph, issued first and never waited for
    //      until the end, turns that workgroup's opening HBM / MALL round trip into an L2 hit
    float pf_val = 0.f;
    if (SGPR_IN_PREFETCH && g_ahead >= 0 && kp.a.centers && kp.a.labels && !kp.a.rag_off) {
        const int NSx = plan_in.N;
        const int lc = (NSx * 12 + 127) / 128 + 1, ll = (NSx * 4 + 127) / 128 + 1;       // lines (+1: arbitrary alignment)
        const int t = threadIdx.x;
        if (t < lc)
            pf_val = kp.a.centers[(size_t)g_ahead * NSx * 3 + min(t * 32, NSx * 3 - 1)];
        else if (t < lc + ll)
            pf_val = __int_as_float(kp.a.labels[(size_t)g_ahead * NSx + min((t - lc) * 32, NSx - 1)]);
    }
    // ---- one slot per thread, fetched once for both branches: xyz + 12 semantic channels
    float fx = 0.f, fy = 0.f, fz = 0.f;
    int mylab = -2;                                       // packed input: this slot's label (-1 = pad)
    float sem[kLabels];
#pragma unroll
    for (int c = 0; c < kLabels; ++c) sem[c] = 0.f;
    // ragged store: a graph with more nodes than slots is an error of the caller, reported like a broken promise
    const bool rag_bad = kp.a.rag_off && !kp.a.dense &&
                         (kp.a.rag_off[g + 1] - kp.a.rag_off[g] < 0 || kp.a.rag_off[g + 1] - kp.a.rag_off[g] > NS);
    if (tid < NS) {
        if (kp.a.dense) {
            const bool second = kp.a.dense2 && g >= kp.a.g_split;
            const int nl = kp.a.num_labels;               // (a smaller architecture's tensors have fewer label channels)
            const float* dn = (second ? kp.a.dense2 : kp.a.dense) +
                              (size_t)(second ? g - kp.a.g_split : g) * (3 + nl) * NS + tid;
            fx = dn[0];
            fy = dn[NS];
            fz = dn[2 * NS];
#pragma unroll
            for (int c = 0; c < kLabels; ++c) sem[c] = c < nl ? dn[(size_t)(3 + c) * NS] : 0.f;
            // the reference's own tensors (transfer_to_torch, sg_net.py:274-298) are one-hot rows / all-zero padding:
            // recover the label so that the super-node branch below applies; anything else runs the generic branch
            int ones = 0, zeros = 0, which = -1;
#pragma unroll
            for (int c = 0; c < kLabels; ++c) {
                ones += sem[c] == 1.f;
                zeros += sem[c] == 0.f;
                which = sem[c] == 1.f ? c : which;
            }
            mylab = (ones == 1 && zeros == kLabels - 1) ? which : (zeros == kLabels ? -1 : -3);
        } else {
            int lab = -1;
            if (kp.a.rag_off) {
                // ragged store: only the graph's real nodes are in memory; the zero padding of transfer_to_torch
                // (sg_net.py:258-272) up to node_num slots is made here
                const long long o0 = kp.a.rag_off[g];
                const long long cnt = kp.a.rag_off[g + 1] - o0;
                if (!rag_bad && tid < cnt) {
                    const float* c3 = kp.a.centers + (size_t)(o0 + tid) * 3;
                    fx = c3[0];
                    fy = c3[1];
                    fz = c3[2];
                    lab = kp.a.rag_lab[o0 + tid];
                }
            } else {
                const float* c3 = kp.a.centers + ((size_t)g * NS + tid) * 3;
                fx = c3[0];
                fy = c3[1];
                fz = c3[2];
                lab = kp.a.labels[(size_t)g * NS + tid];
            }
            if (lab < -1 || lab >= kp.a.num_labels) {
                atomicOr(kp.a.status, 1);
                lab = -1;                                 // (flagged; embedded like padding, never as another class)
            }
            mylab = lab;
#pragma unroll
            for (int c = 0; c < kLabels; ++c) sem[c] = (lab == c) ? 1.f : 0.f;
        }
    }
    // ---- trailing duplicate slots (zero padding: sg_net.py:258-272): slots identical to the last one keep
    // identical features in every layer and any of them is an equally valid neighbour, so only min(m, k) of
    // the m copies are processed; the attention pool re-weights them by m / min(m, k)  
    int N, nd;            // slots processed; slots before the trailing run of duplicates
    float wdup;           // weight of each kept duplicate in the attention pool
    bool one_rep;         // the last processed slot stands for >= k identical slots
    // Everything the prologue decides - the trailing run, "does the super-node branch apply", the nodes per label - is a
    // function of per-wave ballots: each wave publishes its ballots once, ONE barrier later every thread derives what it
    // needs (three barriers from the input to the first layer instead of eight, no LDS atomics).
    bool any_bad = false;     // a node without a label, a stray all-zero row among the nodes, no padding: the generic branch
    int my_label_count = 0;   // lanes 0..11 of every wave: nodes that carry label `lane`
    {
        // (owned-rows instance: in the A region, which nothing touches before the first GEMM - X is then free for the xyz
        //  staging ahead of the prologue's last barrier, see the super-node wave below)
        float* pro = (BIG && SGPR_BIG_SEM_WAVE != 0) ? A : red;
        float* ref = pro;                       // 16 floats: the last slot
        int* wmax = reinterpret_cast<int*>(pro + 16);
        unsigned long long* negm = reinterpret_cast<unsigned long long*>(pro + 32);   // [8 waves] slots with a label < 0
        unsigned long long* nm1 = negm + 8;                                           // [8] slots whose label is not -1
        unsigned long long* labm = nm1 + 8;                                           // [8][12] slots per label
        if (tid == NS - 1) {
            ref[0] = fx;
            ref[1] = fy;
            ref[2] = fz;
#pragma unroll
            for (int c = 0; c < kLabels; ++c) ref[3 + c] = sem[c];
        }
        __syncthreads();
        bool same = tid < NS && fx == ref[0] && fy == ref[1] && fz == ref[2];
#pragma unroll
        for (int c = 0; c < kLabels; ++c) same = same && (sem[c] == ref[3 + c]);
        const unsigned long long differs = __ballot(tid < NS && !same);
        const unsigned long long neg = __ballot(tid < NS && mylab < 0), not_pad = __ballot(tid < NS && mylab != -1);
        unsigned long long lab_mine = 0ull;                 // lane c < 12 keeps the ballot of label c
#pragma unroll
        for (int c = 0; c < kLabels; ++c) {
            const unsigned long long mk = __ballot(tid < NS && mylab == c);
            lab_mine = lane == c ? mk : lab_mine;
        }
        // (slots sit in the first NS / 64 <= 8 waves; the big instance runs up to 16)
        const int NWI = NW < 8 ? NW : 8;
        if (lane == 0 && wave < NWI) {
            wmax[wave] = differs ? wave * 64 + 63 - __clzll((long long)differs) : -1;
            negm[wave] = neg;
            nm1[wave] = not_pad;
        }
        if (lane < kLabels && wave < NWI) labm[wave * kLabels + lane] = lab_mine;
        __syncthreads();
        int last = -1;
#pragma unroll
        for (int q = 0; q < NWI; ++q) last = max(last, wmax[q]);
        nd = last + 1;
        // (nd <= NS - 1: the last slot equals itself)  slots before the trailing run, per wave, as lane masks
        any_bad = ((nm1[nd >> 6] >> (nd & 63)) & 1ull) != 0ull;       // the run's first slot is not padding
#pragma unroll
        for (int q = 0; q < NWI; ++q) {
            const int nb = min(max(nd - 64 * q, 0), 64);
            const unsigned long long below = nb >= 64 ? ~0ull : ((1ull << nb) - 1ull);
            any_bad = any_bad || (negm[q] & below) != 0ull;
            if (lane < kLabels) my_label_count += __popcll(labm[q * kLabels + lane] & below);
        }
        const int m = NS - nd;
        // m >= k copies: ONE representative slot is enough (a row can take at most k copies, and once a row
        // reaches the duplicate key every remaining neighbour slot is filled by copies: see select_phase);
        // fewer copies: keep them all as ordinary slots
        one_rep = m >= p.k && m > 1;
        const int c = one_rep ? 1 : m;
        N = nd + c;
        wdup = (float)m / (float)c;
        // (the scratch above sits in the X region: the barrier before X is written again follows below)
    }
    if (N > p.NC && N <= kp.a.promise && !rag_bad && (kp.a.auto_over == 1 || kp.a.auto_over == 3) && kp.a.redo) {
        // no promise was made and this launch's plan is the lean one: the graph goes to the owned-rows instance for
        // node_num slots - inside the second pass (auto_over 3: flag 3 + the second pass's word) or in a launch of its own
        // (1: flag 3 + over_count) - or, where that instance does not exist for this K, to the second pass's full plan
        if (role == 1) return;
        if (tid == 0) {
            if (kp.a.auto_over == 3) {
                request_redo(kp.a, launch_slot, 3);
            } else if (kp.a.over_count) {
                kp.a.redo[launch_slot] = 3;
                __hip_atomic_store(kp.a.over_count, kp.a.sem_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                request_redo(kp.a, launch_slot, 2);
            }
        }
        return;
    }
    if (N > p.NC || N > kp.a.promise || rag_bad) {   // more slots to process than the caller's node_cap promised: fail loudly
        if (role == 1) return;                   // (reported by the graph's other workgroup)
        if (tid == 0) atomicOr(kp.a.status, rag_bad ? 8 : 2);
        if (tid < 32) kp.a.pooled[(size_t)g * 32 + tid] = __int_as_float(0x7fc00000);
        if (FMT == FMT_H2 && kp.a.redo && tid == 0) kp.a.redo[launch_slot] = 0;
        return;
    }
    if (skip & 32) return;   // ablation: input fetch + duplicate detection only
    const int NP = (N + 15) & ~15;
    const int nrt = NP >> 4;
    // lanes per row in the selection: as many as the workgroup has (two VALU waves per SIMD are needed
    // to keep the vector pipe busy), but at least 8 candidates per lane
    // (P is a power of two: shifts, not integer divisions - the scalar division sequences of this handful of lines were
    // several hundred instructions per wave and graph)
    int lp = 31 - __clz(p.P);
    if (LEAN != 0 || BIG) {
        lp = 2;        // lean instances: always four lanes per row (<= 64 rows x 4 = the workgroup), <= 16 candidates per lane
    } else {
        const int rows = p.overlap ? NP : p.RC;
        while (((2 * rows) << lp) <= NT && ((N + (2 << lp) - 1) >> (lp + 1)) >= 8) ++lp;
    }
    const int P = 1 << lp;
    const int seg = (((N + P - 1) >> lp) + 3) & ~3;

    // ---- the semantic branch on label super-nodes.  Packed input = one-hot rows, so nodes with the same label have
    //      identical features - and, by the argument that lets duplicate slots be dropped, identical
    //      features in EVERY layer of this branch: its whole output is a function of the label (12 labels + the
    //      zero-feature padding representative = 13 distinct rows) and of how many nodes carry each label.
    //      Layer 1: the ranking key of candidate j for row i is 1 - 2[l_i == l_j] (0 for the representative): the
    //      neighbour set is "label-mates, plus the representative when there are fewer than K of them", the a / b
    //      terms are columns of the folded weights -> a 13-row table.  Layers 2 and 3 run on those 13 rows: one Gram
    //      tile, a counting selection (labels in key order, each standing for its node count, the representative for
    //      >= K copies), one row tile of GEMMs, a 13-row gather.  Every value is produced by the instructions of the
    //      generic path on the same operands, so the result is bit-identical to it (tests); the generic path still
    //      runs for semantic rows that are not one-hot, debug dumps, graphs without >= K padding slots, stray
    //      all-zero rows among the nodes, < 17 slots.
    int L0 = 0;
    bool sem_conc = false;                                   // owned-rows instance: the super-node branch runs beside the xyz layers
    const signed char* rowlab = nullptr;                     // fast path: table row of every slot (13 = zero row)
    const bool split = LEAN != 0 && DBG == 0 && role == 2;   // this graph's semantic branch runs in another workgroup
    {
        const int k0 = p.k;
        const bool fast = DBG != 2 && !(skip & 4096) && one_rep && !any_bad && NP >= 32;
        if (!fast && (p.small_park || role == 1)) {
            // this plan parks only the super-node rows: the graph takes the generic branch in the second pass
            // (split launch: decided and reported by the graph's xyz workgroup, which takes the same decision)
            if (tid == 0 && kp.a.redo && role != 1) request_redo(kp.a, launch_slot, 2);
            return;
        }
        if (fast && !p.park_in_lds && p.park_hybrid) park = reinterpret_cast<float*>(smem + p.offPark);
        // scratch that must survive the branch sits behind the 16 virtual rows of the parked block
        int* cnt = reinterpret_cast<int*>(park + 16 * PP);                     // [16] nodes per label, [12] = K
        signed char* rl = reinterpret_cast<signed char*>(cnt + 16);            // [NP]
        int* vmask = reinterpret_cast<int*>(nbr);                              // [16] neighbour label sets
        if (fast) {
            rowlab = rl;
            if (tid < NP) rl[tid] = (signed char)(tid < N ? (tid == nd ? kLabels : mylab) : kLabels + 1);
            // (split launch: the graph's other workgroup computes the branch and needs no counts here)
            if (!split && tid < 16) cnt[tid] = tid == kLabels ? k0 : (tid < kLabels ? my_label_count : 0);
        }
        if constexpr (BIG && SGPR_BIG_SEM_WAVE != 0) {
            // ---- the super-node branch on ONE wave (the first one the graph's row tiles leave idle), in LDS of its own,
            //      while the other waves run the coordinate layer's selection and GEMMs: with one workgroup per CU nothing
            //      else would hide its barriers and round trips (31 of 521 us per config-5 launch).  The xyz input is
            //      staged HERE, ahead of the barrier, so that the xyz waves meet no barrier before the one behind their
            //      first GEMMs - by which time the branch is done (its wave then only keeps the barrier count).
            sem_conc = fast && p.sem_wave && nrt < NW;
            if (sem_conc && tid < NP) {
                const bool live = tid < N;
                unsigned char* xr = X + tid * XROW;
                xstore<FMT>(xr, 0, live ? make_float4(fx, fy, fz, 0.f) : make_float4(0.f, 0.f, 0.f, 0.f), vmax);
                xzero<FMT>(xr, 4);
                xzero<FMT>(xr, 8);
                xzero<FMT>(xr, 12);
                const float n2 = __fadd_rn(__fadd_rn(__fmul_rn(fx, fx), __fmul_rn(fy, fy)), __fmul_rn(fz, fz));   // as torch.sum(x ** 2)
                *reinterpret_cast<float4*>(xr + XROW - 16) = live ? make_float4(fx, fy, fz, n2) : make_float4(0.f, 0.f, 0.f, INFINITY);
            }
        }
        __syncthreads();       // the prologue's scratch (in the X region) is dead; counts and row labels are visible
        if (fast) {
            if ((BIG && SGPR_BIG_SEM_WAVE != 0) && sem_conc) {
                if constexpr (BIG && SGPR_BIG_SEM_WAVE != 0) {
                    if (wave == nrt) {
                        unsigned char* sX = smem + p.offSem;
                        float* sA = reinterpret_cast<float*>(sX + 16 * XROW);
                        float* sxx = sA + 32 * 68;
                        int* svm = reinterpret_cast<int*>(sxx + 16);
                        supernode_branch<FMT, 0, true>(kp.w, k0, skip, sX, sxx, sA, sA + 16 * 68, 68, 20, cnt, svm, park, lane, 0, 64, 1, vmax);
                    }
                }
            } else if (!split) {
                supernode_branch<FMT, (BIG ? -1 : LEAN), false>(kp.w, k0, skip, X, xx, A, D, p.pitchA, p.pitchD, cnt, vmask, park, tid, wave, NT, NW, vmax);
#if SGPR_EXP_DOUBLE & 8
                __syncthreads();
                supernode_branch<FMT, LEAN, false>(kp.w, k0, skip, X, xx, A, D, p.pitchA, p.pitchD, cnt, vmask, park, tid, wave, NT, NW, vmax);
#endif
            }
            L0 = 3;
            if (LEAN != 0 && DBG == 0 && role == 1) {
                // ---- split launch, semantic half: publish the 16 rows (release at agent scope), then the flag
                int* ovf = reinterpret_cast<int*>(xx);      // (the squared norms are dead)
                if (tid == 0) *ovf = 0;
                __syncthreads();
                for (int e = tid; e < 16 * 8; e += NT)
                    *reinterpret_cast<float4*>(kp.a.sem_tab + ((size_t)launch_slot * 16 + (e >> 3)) * PP + (e & 7) * 4) =
                        *reinterpret_cast<const float4*>(park + (size_t)(e >> 3) * PP + (e & 7) * 4);
                if (FMT == FMT_H2 && __ballot(!(vmax < kF16Safe)) != 0ull && lane == 0) atomicOr(ovf, 1);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                __syncthreads();
                // (debug mask bit 20, tests only: the producers of odd launch slots withhold their flag - their consumers
                //  must time out and hand the graph to the second pass)
                if (tid == 0 && !((kp.a.skip & (1 << 20)) && (launch_slot & 1)))
                    __hip_atomic_store(kp.a.sem_flag + launch_slot,
                                       sem_token(kp.a.sem_epoch, launch_slot) | (*ovf ? 0x80000000ull : 0ull), __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
                return;
            }
        } else if (tid < NP) {
            // ---- generic: stage the first branch's input (12 semantic channels, zero padded to 16 / NP rows) +
            //      squared norms, ahead of the layer loop so that the 12 input registers die here
            const bool live = tid < N;
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < kLabels; ++c) s = fmaf(sem[c], sem[c], s);
            const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
            unsigned char* xr = X + tid * XROW;
            xstore<FMT>(xr, 0, live ? make_float4(sem[0], sem[1], sem[2], sem[3]) : z4, vmax);
            xstore<FMT>(xr, 4, live ? make_float4(sem[4], sem[5], sem[6], sem[7]) : z4, vmax);
            xstore<FMT>(xr, 8, live ? make_float4(sem[8], sem[9], sem[10], sem[11]) : z4, vmax);
            xzero<FMT>(xr, 12);
            xx[tid] = live ? s : 0.f;
        }
    }
    __syncthreads();
    SGPR_PROF(0)

    for (int L = L0; L < 6; ++L) {
        // per-iteration opaque copy: keeps the compiler from hoisting (and then spilling) dozens of
        // k-derived predicates out of the layer loop
        int k = KC ? KC : p.k;
        if (!KC) asm volatile("" : "+s"(k));
        // same for the lane-derived addresses: re-derived per layer (a few VALU ops) instead of living in - and
        // spilling from - dozens of registers across all phases
        tid = tid0;
        asm volatile("" : "+v"(tid));
        lane = tid & 63;
        l15 = lane & 15;
        lq = lane >> 4;
        if (L == 3 && !((BIG && SGPR_BIG_SEM_WAVE != 0) && sem_conc)) {
            // ---- stage the second branch's input (xyz, zero padded to 16 channels / NP rows) + (x, y, z, |x|^2) in fp32
            if (tid < NP) {
                const bool live = tid < N;
                unsigned char* xr = X + tid * XROW;
                xstore<FMT>(xr, 0, live ? make_float4(fx, fy, fz, 0.f) : make_float4(0.f, 0.f, 0.f, 0.f), vmax);
                xzero<FMT>(xr, 4);
                xzero<FMT>(xr, 8);
                xzero<FMT>(xr, 12);
                const float n2 = __fadd_rn(__fadd_rn(__fmul_rn(fx, fx), __fmul_rn(fy, fy)), __fmul_rn(fz, fz));   // as torch.sum(x ** 2)
                *reinterpret_cast<float4*>(xr + XROW - 16) = live ? make_float4(fx, fy, fz, n2) : make_float4(0.f, 0.f, 0.f, INFINITY);
            }
            __syncthreads();
            SGPR_PROF(0)
        }
        const int Kp = kp.w.kp[L], cout = kp.w.cout[L];
        const bool k64 = Kp == 64;
        const int Ldump = L < 3 ? L + 3 : L - 3;     // dumps keep the reference's order: xyz1..3, sem1..3
        int32_t* dbg_knn = dbg_knn_all ? dbg_knn_all + ((size_t)g * 6 + Ldump) * NS * p.k : nullptr;
        // ---- kNN keys (Gram on MFMA) -> selection, one chunk of rows at a time (a single chunk, upper-triangular
        //      tiles mirrored, when the whole key matrix is resident)
        constexpr bool kOwned = SGPR_OWNED_SELECT != 0 && LEAN != 0 && DBG == 0 && FMT == FMT_H2 && KC == 10;
        // the 64-row production instance (128 registers per lane at four workgroups per CU, ~90 in use): the weights of
        // this wave's first column tile of the layer's GEMMs are requested BEFORE the selection
        constexpr bool kWPre = SGPR_WPREFETCH != 0 && kOwned && LEAN == 64 && SGPR_LEAN_WAVES == 4;
        FragT<FMT> wpre[2];
        if constexpr (kWPre) {
            const unsigned short* wl = kp.w.wh[L] + (size_t)wave * (Kp == 64 ? wtile<4, FMT>() : wtile<1, FMT>());
            if (Kp == 64)
                load_wfrag<4>(wl, wpre);
            else
                load_wfrag<1>(wl, wpre);
        }
        if constexpr (BIG) {
            // ---- owned rows beyond 64 slots (only the xyz layers 3..5 get here: the semantic branch ran on the super-nodes):
            //      selection from the accumulators, then the wave's own row tile of the per-node GEMMs; b waits in registers
            //      until every wave has read X (as candidates and as operands)
            f32x4 bown[4];
            const bool mine = wave < nrt;
            select_owned_big<FMT, KC, KP>(p, N, nrt, one_rep, X, xx, nbr, wave, L == 3, p.big == 2);
#if SGPR_EXP_DOUBLE & 32
            select_owned_big<FMT, KC, KP>(p, N, nrt, one_rep, X, xx, nbr, wave, L == 3, p.big == 2);
#endif
#if SGPR_EXP_DOUBLE & 64
            if (mine) {
                if (Kp != 64)
                    gemm_own<1, 64, FMT>(X, A, p.pitchA, kp.w.wh[L], kp.w.tb[L], wave, bown);
                else if (cout == 64)
                    gemm_own<4, 64, FMT>(X, A, p.pitchA, kp.w.wh[L], kp.w.tb[L], wave, bown);
                else
                    gemm_own<4, 32, FMT>(X, A, p.pitchA, kp.w.wh[L], kp.w.tb[L], wave, bown);
                asm volatile("" : "+v"(bown[0]), "+v"(bown[1]), "+v"(bown[2]), "+v"(bown[3]));
            }
#endif
            if (mine) {
                if (Kp != 64)
                    gemm_own<1, 64, FMT>(X, A, p.pitchA, kp.w.wh[L], kp.w.tb[L], wave, bown);
                else if (cout == 64)
                    gemm_own<4, 64, FMT>(X, A, p.pitchA, kp.w.wh[L], kp.w.tb[L], wave, bown);
                else
                    gemm_own<4, 32, FMT>(X, A, p.pitchA, kp.w.wh[L], kp.w.tb[L], wave, bown);
            }
            __syncthreads();                          // A is complete; X has been read by everyone
            if (mine) {
                unsigned char* x0 = X + (wave * 16 + l15) * XROW;
#pragma unroll
                for (int cb = 0; cb < 4; ++cb)
                    if (cb < (cout >> 4))
                        *reinterpret_cast<float4*>(x0 + (cb * 16 + 4 * lq) * 4) = make_float4(bown[cb][0], bown[cb][1], bown[cb][2], bown[cb][3]);
            }
        }
        if constexpr (kOwned) {
            // lean production instances: keys in registers, selection right behind them, no barrier until the GEMMs' own
            // (A is written only by the GEMM phase and nothing reads it here; b replaces X behind gemm_cols' barrier, which
            // every wave reaches after its selection)
            select_owned<FMT>(p, N, nrt, one_rep, X, xx, nbr, wave, L == 3);
        }
        for (int rc0 = 0; !kOwned && !BIG && rc0 < NP; rc0 += p.RC) {
            const int rows_chunk = min(p.RC, NP - rc0);
            if (skip & 4) {
            } else if (L == 3) {
                gram_xyz_direct<FMT>(X, D, p.pitchD, N, NP, rc0, rows_chunk, tid, NT);
#if SGPR_EXP_DOUBLE & 16
                __syncthreads();
                gram_xyz_direct<FMT>(X, D, p.pitchD, N, NP, rc0, rows_chunk, tid, NT);
#endif
            } else if (p.overlap) {
                if (k64) {
                    gram_tiles_sym<4, FMT, !LEAN, (LEAN != 0 ? SGPR_LEAN_WAVES : 0)>(X, xx, D, p.pitchD, N, nrt, wave);
#if SGPR_EXP_DOUBLE & 4
                    __syncthreads();
                    gram_tiles_sym<4, FMT, !LEAN, (LEAN != 0 ? SGPR_LEAN_WAVES : 0)>(X, xx, D, p.pitchD, N, nrt, wave);
#endif
                } else
                    gram_tiles_sym<1, FMT, !LEAN, (LEAN != 0 ? SGPR_LEAN_WAVES : 0)>(X, xx, D, p.pitchD, N, nrt, wave);
            } else {
                const int nti = rows_chunk >> 4;
                for (int tile = wave; tile < nti * nrt; tile += NW) {
                    const int ti = tile / nrt, tj = tile - ti * nrt;
                    if (k64)
                        gram_tile<4, FMT>(X, xx, D, p.pitchD, N, rc0, ti, tj, false, l15, lq);
                    else
                        gram_tile<1, FMT>(X, xx, D, p.pitchD, N, rc0, ti, tj, false, l15, lq);
                }
            }
            __syncthreads();
            SGPR_PROF(2)
            if (!(skip & 1)) {
                // register sorting networks whenever the chunk's rows x lanes-per-row fit the workgroup (always for the
                // resident key matrix; for chunked keys too: 2x faster than the value bisection at node_num 256, which
                // stays as the fallback), with lists cut to the reference's K (10) or the stress configuration's (20)
                if (p.overlap || (!LEAN && rows_chunk * P <= NT && seg <= CAP)) {
                    unsigned long long* const sp = (DBG == 2 && prof_buf && (skip & 128)) ? prof_buf + 8 : nullptr;
                    if (KP == 16 && k == 10)
                        select_phase<KP, LEAN ? 16 : CAP, (KP == 16 ? 10 : KP), (LEAN != 0 ? 4 : 0)>(p, N, NP, P, seg, k, one_rep, D, rc0, rows_chunk, nbr, dbg_knn, sp);
#if SGPR_EXP_DOUBLE & 2
                    if (KP == 16 && k == 10) {
                        __syncthreads();
                        select_phase<KP, LEAN ? 16 : CAP, (KP == 16 ? 10 : KP), (LEAN != 0 ? 4 : 0)>(p, N, NP, P, seg, k, one_rep, D, rc0, rows_chunk, nbr, dbg_knn, sp);
                    }
#endif
                    else if (KP == 32 && k == 20)
                        select_phase<KP, LEAN ? 16 : CAP, (KP == 32 ? 20 : KP), (LEAN != 0 ? 4 : 0)>(p, N, NP, P, seg, k, one_rep, D, rc0, rows_chunk, nbr, dbg_knn, sp);
                    else
                        select_phase<KP, LEAN ? 16 : CAP, KP, (LEAN != 0 ? 4 : 0)>(p, N, NP, P, seg, k, one_rep, D, rc0, rows_chunk, nbr, dbg_knn, sp);
                } else {              // one wave per row, bisection on the key value
                    select_bisect<4>(p, N, NP, k, one_rep, D, rc0, rows_chunk, nbr, dbg_knn);
                }
            }
            __syncthreads();                          // the key matrix / chunk is reused (next chunk, or A)
            SGPR_PROF(1)
        }
        // per-node GEMMs (MFMA); A overwrites the key matrix
        if constexpr (!BIG) {
        if (!(skip & 2)) gemm_layer<(LEAN != 0), FMT, (LEAN != 0 ? SGPR_LEAN_WAVES : 0)>(X, A, p.pitchA, (FMT == FMT_H2 ? kp.w.wh[L] : kp.w.wb[L]), kp.w.tb[L], Kp, cout, nrt, wave, NW, skip, kWPre ? wpre : nullptr);
        __syncthreads();  // neighbour lists, A and b (in X) are complete
        SGPR_PROF(3)
        }

        // ---- gather-max over the k neighbours: cout/4 lanes per row, 4 channels (16 B) per lane
        {
            const int lpr = cout >> 2;                   // lanes per row: 16 or 8
            const int lsh = cout == 64 ? 4 : 3;          // log2(lpr): shifts, not integer divisions (a division by a run-time
            const int rpw = 64 >> lsh;                   //   value is a ~30-instruction sequence: -2 % of the launch)
            int c4 = (lane & (lpr - 1)) * 4, sub = lane >> lsh;
#if SGPR_GATHER_GROUPS
            // 64-channel layers: the 16 lanes of a row = one of the LDS's four lane groups of a ds_read_b128
            // ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, the same + 32: MI355X_MICROARCH.md), so that a group reads 256
            // contiguous bytes of ONE neighbour row - no bank conflict whatever rows its neighbours are; with 16 consecutive
            // lanes per row every group holds pieces of two rows, which collide unless the rows are 16 apart
            const bool ggrp = cout == 64;
            if (ggrp) {
                const int hl = lane & 31, q = hl >> 2;
                sub = ((0x96 >> q) & 1) + 2 * (lane >> 5);
                c4 = (4 * (q >> 1) + (hl & 3)) * 4;
            }
#else
            const bool ggrp = false;
#endif
            const bool want_norm = (L != 2 && L != 5);
            float* dbg = dbg_layers ? dbg_layers + ((size_t)g * 6 + Ldump) * NS * 64 : nullptr;
            const int rstep = NW * rpw;
            // a wave-iteration covers two groups of rpw consecutive rows; groups that lie entirely in the padding
            // [N, NP) are skipped (their rows are zero-filled below: the matrix phases still read them as operands)
            // (a wave's last group may have no partner: that iteration runs the one-row form instead of computing - and
            // discarding - a second copy of the same row)
            auto rows = [&](auto two_tag, const int ia, const int ib) {
                constexpr bool TWO = decltype(two_tag)::value;
                const int ra = min(ia, N - 1), rb = TWO ? min(ib, N - 1) : ra;   // padded rows: compute a real row
                float4 ma, mb;
                if constexpr (TWO) {
                    gather_max2(A + c4, nbr + ra * p.kpitch, nbr + rb * p.kpitch, k, ma, mb);
                } else {
                    gather_max1(A + c4, nbr + ra * p.kpitch, k, ma);
                    mb = ma;
                }
                // padded rows (>= N) inside a partly real group simply keep a copy of row N-1: they are never candidates
                // (their key is +inf) and nothing reads them as rows
                const float4 ya = add_lrelu(ma, *reinterpret_cast<const float4*>(X + ra * XROW + 4 * c4), true);
                const float4 yb = TWO ? add_lrelu(mb, *reinterpret_cast<const float4*>(X + rb * XROW + 4 * c4), true) : ya;
                if (dbg) {
                    if (ia < N) {
                        *reinterpret_cast<float4*>(dbg + (size_t)ia * 64 + c4) = ya;
                        if (cout == 32) *reinterpret_cast<float4*>(dbg + (size_t)ia * 64 + 32 + c4) = make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                    if (TWO && ib < N) {
                        *reinterpret_cast<float4*>(dbg + (size_t)ib * 64 + c4) = yb;
                        if (cout == 32) *reinterpret_cast<float4*>(dbg + (size_t)ib * 64 + 32 + c4) = make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                }
                // the next layer's operand: planes over the row's own (already consumed) b; all lanes of a row sit in
                // one wave, whose LDS reads above precede these writes
                if (L == 2) {
                    *reinterpret_cast<float4*>(park + (size_t)ia * PP + c4) = ya;
                    if (TWO) *reinterpret_cast<float4*>(park + (size_t)ib * PP + c4) = yb;
                } else {
                    xstore<FMT>(X + ia * XROW, c4, ya, vmax);
                    if (TWO) xstore<FMT>(X + ib * XROW, c4, yb, vmax);
                }
                if (want_norm) {                          // squared norms of the next layer's input rows (cout == 64: 16 lanes/row)
                    float sa = fmaf(ya.x, ya.x, fmaf(ya.y, ya.y, fmaf(ya.z, ya.z, ya.w * ya.w)));
                    float sb = TWO ? fmaf(yb.x, yb.x, fmaf(yb.y, yb.y, fmaf(yb.z, yb.z, yb.w * yb.w))) : 0.f;
                    sa += lane_xor(sa, 1);
                    if (TWO) sb += lane_xor(sb, 1);
                    sa += lane_xor(sa, 2);
                    if (TWO) sb += lane_xor(sb, 2);
                    if (ggrp) {
                        // the row's four quads: two meet by a row mirror (lanes 0-3 <-> 15-12, 4-7 <-> 11-8), the pairs 20 lanes apart
                        sa += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sa), 0x140, 0xF, 0xF, false));
                        if (TWO) sb += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sb), 0x140, 0xF, 0xF, false));
                        sa += __shfl_xor(sa, 20);
                        if (TWO) sb += __shfl_xor(sb, 20);
                    } else {
                        sa += lane_xor(sa, 4);
                        if (TWO) sb += lane_xor(sb, 4);
                        sa += lane_xor(sa, 8);
                        if (TWO) sb += lane_xor(sb, 8);
                    }
                    if (ggrp ? c4 == 0 : (lane & 15) == 0) {
                        // (big instance: +inf for the slots the graph does not have - OwnedKeys ranks them last without an index test)
                        xx[ia] = (!BIG || ia < N) ? sa : INFINITY;
                        if (TWO) xx[ib] = (!BIG || ib < N) ? sb : INFINITY;
                    }
                }
            };
            if constexpr (BIG) {
                // the wave's own 16 rows (its neighbour lists and its b rows are its own LDS traffic: program order), two
                // groups of rpw rows at a time; all-padding groups of the graph's last tile get zero planes
                if (wave < nrt) {
                    constexpr int QW = 256 / 16;
                    for (int r0 = 0; r0 < 16; r0 += 2 * rpw) {
                        const int ga = 16 * wave + r0;      // (wave-uniform)
                        if (ga < N) {
                            if (ga + rpw < N)
                                rows(std::true_type{}, ga + sub, ga + rpw + sub);
                            else
                                rows(std::false_type{}, ga + sub, ga + sub);
                        }
                        const int z0 = ga >= N ? ga : (ga + rpw >= N ? ga + rpw : ga + 2 * rpw), z1 = ga + 2 * rpw;
                        for (int e = lane; e < (z1 - z0) * QW; e += 64) {
                            *reinterpret_cast<uint4*>(X + (z0 + e / QW) * XROW + (e % QW) * 16) = make_uint4(0u, 0u, 0u, 0u);
                            if (e % QW == 0) xx[z0 + e / QW] = INFINITY;
                        }
                    }
                }
            } else {
            for (int ia = wave * rpw + sub; ia - sub < ((skip & 8) ? 0 : N); ia += 2 * rstep) {
                if (ia + rstep - sub < N)                  // wave-uniform
                    rows(std::true_type{}, ia, ia + rstep);
                else
                    rows(std::false_type{}, ia, ia);
            }
            }
            // rows of the skipped all-padding groups: zero planes (finite operands for the next layer's matrix phases)
            if (!BIG && L != 2 && !(skip & 8)) {
                constexpr int QW = (FMT == FMT_BF3 ? 384 : 256) / 16;        // 16-byte pieces per row
                const int nq = (N + rpw - 1) & ~(rpw - 1);
                for (int e = tid; e < (NP - nq) * QW; e += NT)
                    *reinterpret_cast<uint4*>(X + (nq + e / QW) * XROW + (e % QW) * 16) = make_uint4(0u, 0u, 0u, 0u);
            }
        }
        __syncthreads();
        SGPR_PROF(5)
    }

    if (skip & 131072) return;   // ablation bit 17: nothing after the layer loop
    const float* sem3 = park;                                // the first branch's output: [row][PP]
    if (split && rowlab) {
        // ---- meet the semantic half: its 16 rows are in global memory once sem_flag[slot] carries this launch's token
        int* met = reinterpret_cast<int*>(xx);               // (the squared norms are dead: no Gram phase is left)
        if (tid == 0) {
            const unsigned long long want = sem_token(kp.a.sem_epoch, launch_slot);
            unsigned long long got = 0ull;
            int spins = 0;
            while (true) {
                got = __hip_atomic_load(kp.a.sem_flag + launch_slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((got & ~0x80000000ull) == want || ++spins > (1 << 20)) break;
                __builtin_amdgcn_s_sleep(8);
            }
            *met = (got & ~0x80000000ull) == want ? ((got & 0x80000000ull) ? 2 : 1) : 0;
        }
        __syncthreads();
        const int st = *met;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // the rows were published before the flag
        if (st == 0) {
            // The producer did not arrive in time: HIP promises neither co-residency nor forward progress between the
            // workgroups of a launch, and with another stream or process on the device the producers may still be
            // waiting for a CU.  Not an error: the graph is handed to the second pass (full plan, both branches in one
            // workgroup); without a second pass (no workspace flag) it stays a loud failure.
            if (kp.a.redo) {
                if (tid == 0) request_redo(kp.a, launch_slot, 2);
            } else {
                if (tid == 0) atomicOr(kp.a.status, 4);
                if (tid < 32) kp.a.pooled[(size_t)g * 32 + tid] = __int_as_float(0x7fc00000);
            }
            return;
        }
        if (st == 2 && tid == 0) vmax = INFINITY;            // the branch left the f16 range: this graph goes to the second pass
        sem3 = kp.a.sem_tab + (size_t)launch_slot * 16 * PP;
        __syncthreads();                                     // `met` is read before xx is used again
    }
    // conv_end weights of this wave's first tile: in flight while sem3 is moved back
    FragT<FMT> wf_end[2];
    int ct_end = wave & 1;
    load_wfrag<4>((FMT == FMT_H2 ? kp.w.wh_end : kp.w.wb_end) + (size_t)ct_end * wtile<4, FMT>(), wf_end);
    float4 t4_end = *reinterpret_cast<const float4*>(kp.w.tb_end + ct_end * 16 + 4 * lq);
    for (int e = tid; e < NP * 8; e += NT) {                      // sem3 -> channels 32..63: X = cat(xyz3, sem3)
        const int i = e >> 3, c4 = (e & 7) * 4;
        const int pr = rowlab ? (int)rowlab[i] : i;              // fast path: the row of slot i's label
        const float4 v = i < N ? *reinterpret_cast<const float4*>(sem3 + (size_t)pr * PP + c4) : make_float4(0.f, 0.f, 0.f, 0.f);
        xstore<FMT>(X + i * XROW, 32 + c4, v, vmax);
    }
    __syncthreads();

    // ---- conv_end: 64 -> 32, folded BN, LeakyReLU  -> E (in the A region)
    float* E = A;
    {
        for (int task = wave; task < 2 * nrt; task += NW) {
            const int ct = task & 1, rt = task >> 1;      // NW even: ct == wave & 1 for every task of this wave
            if (ct != ct_end) {                           // odd wave counts (192-thread workgroups) alternate
                ct_end = ct;
                load_wfrag<4>((FMT == FMT_H2 ? kp.w.wh_end : kp.w.wb_end) + (size_t)ct * wtile<4, FMT>(), wf_end);
                t4_end = *reinterpret_cast<const float4*>(kp.w.tb_end + ct * 16 + 4 * lq);
            }
            FragT<FMT> xf[2];
            xload<4, FMT>(X + (rt * 16 + l15) * XROW, lq, xf);
            const f32x4 acc = tile16<4>(wf_end, xf);
            const int c4 = ct * 16 + 4 * lq;
            float4 e4 = make_float4(acc[0] + t4_end.x, acc[1] + t4_end.y, acc[2] + t4_end.z, acc[3] + t4_end.w);
            e4.x = e4.x > 0.f ? e4.x : 0.2f * e4.x;
            e4.y = e4.y > 0.f ? e4.y : 0.2f * e4.y;
            e4.z = e4.z > 0.f ? e4.z : 0.2f * e4.z;
            e4.w = e4.w > 0.f ? e4.w : 0.2f * e4.w;
            *reinterpret_cast<float4*>(E + (rt * 16 + l15) * PE + c4) = e4;
        }
    }
    __syncthreads();
    SGPR_PROF(6)
    if (kp.a.emb)                                                  // dropped duplicates replicate the last kept row
        for (int e = tid; e < NS * 8; e += NT) {
            const int i = e >> 3, c4 = (e & 7) * 4;
            *reinterpret_cast<float4*>(kp.a.emb + ((size_t)g * NS + i) * 32 + c4) =
                *reinterpret_cast<const float4*>(E + min(i, N - 1) * PE + c4);
        }
    if (dbg_layers)
        for (int e = tid; e < 6 * (NS - N) * 16; e += NT) {
            const int c4 = (e & 15) * 4, r = e >> 4;
            const int Ld = r / (NS - N), i = N + r - Ld * (NS - N);
            float* base = dbg_layers + ((size_t)g * 6 + Ld) * NS * 64;
            *reinterpret_cast<float4*>(base + (size_t)i * 64 + c4) = *reinterpret_cast<const float4*>(base + (size_t)(N - 1) * 64 + c4);
        }
    if (dbg_knn_all)
        for (int e = tid; e < 6 * (NS - N) * p.k; e += NT) {
            const int q = e % p.k, r = e / p.k;
            const int Ld = r / (NS - N), i = N + r - Ld * (NS - N);
            int32_t* base = dbg_knn_all + ((size_t)g * 6 + Ld) * NS * p.k;
            base[(size_t)i * p.k + q] = base[(size_t)(N - 1) * p.k + q];
        }

    if (skip & 32768) return;   // ablation bit 15: no attention pooling
    // ---- attention pooling over all NS slots (padding is NOT masked, divisor NS: layers_batch.py:34-38);
    //      each kept duplicate row stands for wdup slots
    constexpr int NPART = 8;         // partial sums per channel: fixed, so results do not depend on the block size
    float* mean = red + NPART * 32;
    float* tg = mean + 32;
    int* ovflag = reinterpret_cast<int*>(tg + 32);          // FMT_H2: some value stored into the f16 planes was out of range
    float* sig = xx;
    const int c = tid & 31;
    if (FMT == FMT_H2 && tid == 0) *ovflag = 0;
    // column tid of the attention matrix (lanes 0..31): needed two barriers from here - requested now, so that its L2
    // round trip runs under the partial sums instead of opening the tanh step
    float attw[32];
    if (SGPR_ATT_PREFETCH && tid < 32) {
#pragma unroll
        for (int r = 0; r < 32; ++r) attw[r] = kp.w.att_w[r * 32 + tid];
    }
    for (int prt = tid >> 5; prt < NPART; prt += NT >> 5) {
        float s = 0.f;
        for (int n = prt; n < N; n += NPART) s = fmaf(n >= nd ? wdup : 1.f, E[n * PE + c], s);
        red[prt * 32 + c] = s;
    }
    __syncthreads();
    if (tid < 32) {
        float s = 0.f;
        for (int q = 0; q < NPART; ++q) s += red[q * 32 + tid];
        mean[tid] = s / (float)NS;
    }
    __syncthreads();
    if (tid < 32) {
        float gc = 0.f;
        if (SGPR_ATT_PREFETCH) {
#pragma unroll
            for (int r = 0; r < 32; ++r) gc = fmaf(mean[r], attw[r], gc);
        } else {
            for (int r = 0; r < 32; ++r) gc = fmaf(mean[r], kp.w.att_w[r * 32 + tid], gc);
        }
        tg[tid] = tanhf(gc);
    }
    __syncthreads();
    for (int n = tid; n < N; n += NT) {
        float dsum = 0.f;
        for (int q = 0; q < 32; ++q) dsum = fmaf(E[n * PE + q], tg[q], dsum);
        sig[n] = 1.f / (1.f + expf(-dsum));
    }
    __syncthreads();
    if (kp.a.att)
        for (int n = tid; n < NS; n += NT) kp.a.att[(size_t)g * NS + n] = sig[min(n, N - 1)];
    for (int prt = tid >> 5; prt < NPART; prt += NT >> 5) {
        float s = 0.f;
        for (int n = prt; n < N; n += NPART) s = fmaf((n >= nd ? wdup : 1.f) * sig[n], E[n * PE + c], s);
        red[prt * 32 + c] = s;
    }
    if (FMT == FMT_H2) {                                     // !(vmax < safe): NaN inputs count as out of range too
        const unsigned long long bad = __ballot(!(vmax < kF16Safe));
        if (bad && lane == 0) atomicOr(ovflag, 1);
    }
    __syncthreads();
    if (tid < 32) {
        float s = 0.f;
        for (int q = 0; q < NPART; ++q) s += red[q * 32 + tid];
        kp.a.pooled[(size_t)g * 32 + tid] = s;
    }
    // a graph whose activations left the f16 range is embedded again by the wide-range instance (embed_redo_kernel)
    if (FMT == FMT_H2 && kp.a.redo && tid == 0) request_redo(kp.a, launch_slot, *ovflag ? 1 : 0);
    asm volatile("" ::"v"(pf_val));                          // (the prefetch loads above end here)
    SGPR_PROF(7)
#undef SGPR_PROF
    if (prof && lane == 0) {
#pragma unroll
        for (int q = 0; q < 8; ++q) atomicAdd(&prof_buf[q], pacc[q]);
    }
}

#ifndef SGPR_EMBED_STAMPS
#define SGPR_EMBED_STAMPS 0    // 1 (variant builds): every workgroup of embed_kernel leaves its start / end time and where it ran
#endif                         // (tools/exp/embed_timeline.py)
#if SGPR_EMBED_STAMPS
__device__ unsigned long long embed_stamps[32768 * 4];
extern "C" int sgpr_debug_embed_stamps(unsigned long long* host_out) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(embed_stamps), sizeof(embed_stamps)) == hipSuccess ? 0 : -1;
}
#endif
template <int KP, int DBG, int LEAN, int FMT, int KC = 0>
#ifndef SGPR_EXP_LDS32
#define SGPR_EXP_LDS32 0
#endif
__global__ __launch_bounds__(LEAN ? kLeanNT : NT_MAX, LEAN ? (DBG == 0 ? ((LEAN == 48 || SGPR_EXP_LDS32) ? 5 : 4) : 3) : 1) void embed_kernel(const KParams kp) {
#if SGPR_EMBED_STAMPS
    if (threadIdx.x == 0 && blockIdx.x < 32768) {
        embed_stamps[blockIdx.x * 4 + 0] = wall_clock64();
        embed_stamps[blockIdx.x * 4 + 1] = ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) |
                                           (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4);      // XCC_ID | HW_ID
    }
#endif
    // (ONE call site: two inlined copies of embed_graph would double the kernel's footprint in the instruction cache)
    int slot = (int)blockIdx.x, role = 0;
    if constexpr (LEAN != 0 && DBG == 0) {
        if (kp.a.sem_tab) {            // split launch: workgroups [0, G) semantic halves - the PRODUCERS are dispatched
            role = slot >= kp.a.G ? 2 : 1;                        // first, a waiting xyz half [G, 2 G) never holds a CU its
            slot -= role == 2 ? kp.a.G : 0;                       // own producer still needs
        }
    }
    // (the prefetch partner: one round of resident workgroups ahead, same XCD; fetched beside this slot's own index)
    int g_ahead = -1;
    if constexpr (LEAN != 0 && DBG == 0 && SGPR_IN_PREFETCH != 0) {
        const int s2 = slot + SGPR_IN_PREFETCH;
        if (role == 0 && s2 < kp.a.G) g_ahead = kp.a.ids ? kp.a.ids[s2] : s2;
    }
    embed_graph<KP, DBG, LEAN, FMT, KC>(kp, kp.p, kp.a.ids ? kp.a.ids[slot] : slot, slot, role, g_ahead);
#if SGPR_EMBED_STAMPS
    __syncthreads();
    if (threadIdx.x == 0 && blockIdx.x < 32768) embed_stamps[blockIdx.x * 4 + 2] = wall_clock64();
#endif
}

// Owned rows beyond 64 slots: up to sixteen waves (one per 16-row tile) under a 128-register budget
template <int KP, int KC>
__global__ __launch_bounds__(1024) void embed_big_kernel(const KParams kp) {
    if (kp.a.auto_over == 2) {
        // the launch behind an "auto" lean launch: the slots that launch flagged 3 (more processed slots than the lean
        // plan holds), workgroup b those congruent to b.  Nothing flagged (KITTI-like data): one load, done - a PLAIN
        // load, the word was stored by the previous kernel of the stream (see embed_redo_kernel)
        if (*kp.a.over_count != kp.a.sem_epoch) return;
        for (int slot = (int)blockIdx.x; slot < kp.a.G; slot += (int)gridDim.x) {      // (workgroup-uniform)
            if (kp.a.redo[slot] != 3) continue;
            embed_graph<KP, 0, 0, FMT_H2, KC, true>(kp, kp.p, kp.a.ids ? kp.a.ids[slot] : slot, slot, 0, -1);
            __syncthreads();                             // LDS is reused by the next graph
        }
        return;
    }
    const int slot = (int)blockIdx.x;
#if SGPR_EMBED_STAMPS
    if (threadIdx.x == 0 && blockIdx.x < 32768) {
        embed_stamps[blockIdx.x * 4 + 0] = wall_clock64();
        embed_stamps[blockIdx.x * 4 + 1] = ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) |
                                           (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4);      // XCC_ID | HW_ID
    }
#endif
    embed_graph<KP, 0, 0, FMT_H2, KC, true>(kp, kp.p, kp.a.ids ? kp.a.ids[slot] : slot, slot, 0, -1);
#if SGPR_EMBED_STAMPS
    __syncthreads();
    if (threadIdx.x == 0 && blockIdx.x < 32768) embed_stamps[blockIdx.x * 4 + 2] = wall_clock64();
#endif
}

// Second pass over the launch slots the f16 instance flagged (kp.a.redo):
//   1  a coordinate or an activation reached the f16 range -> bf16 planes / fp32 rows (plan kp.p: fp32's range)
//   2  the graph needs the generic semantic branch, which the lean plan's 16-row park cannot hold -> the full f16 plan kp.p2
// A handful of persistent workgroups scan the flags; on real data there is nothing to do and the launch costs ~4 us.
template <int KP, int FMTW>
__global__ __launch_bounds__(NT_MAX, 1) void embed_redo_kernel(const KParams kp) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // nothing requested by the first pass (every launch on real data): one load, done.  A PLAIN load: the word was
    // stored by the previous kernel of the stream (the kernel boundary makes it visible); an agent-coherent load would
    // cost this empty pass a trip past the L2 (the pass is pure latency: 4.6 us by rocprof with the coherent load)
    if (kp.a.redo_count && *kp.a.redo_count != kp.a.sem_epoch) return;
    // this workgroup's contiguous range of launch slots, 64 flags at a time: wave 0 reads them with one load and hands
    // the ballots to the other waves through the first bytes of LDS (one dependent load per slot made the empty pass
    // cost 23 us); the masks then live in registers, so embed_graph is free to overwrite LDS
    const int per = (kp.a.G + gridDim.x - 1) / gridDim.x;
    const int b0 = blockIdx.x * per, b1 = min(kp.a.G, b0 + per);
    for (int base = b0; base < b1; base += 64) {
        if (threadIdx.x < 64) {
            const int slot = base + (int)threadIdx.x;
            const int f = slot < b1 ? kp.a.redo[slot] : 0;
            // (3 without kp.p3.big belongs to embed_big_kernel's own launch, which ran ahead of this one)
            const unsigned long long m1 = __ballot(f == 1), m2 = __ballot(f == 2), m3 = __ballot(f == 3);
            if (threadIdx.x == 0) {
                reinterpret_cast<unsigned long long*>(smem)[0] = m1;
                reinterpret_cast<unsigned long long*>(smem)[1] = m2;
                reinterpret_cast<unsigned long long*>(smem)[2] = m3;
            }
        }
        __syncthreads();
        unsigned long long wide = reinterpret_cast<const unsigned long long*>(smem)[0];
        unsigned long long full = reinterpret_cast<const unsigned long long*>(smem)[1];
        unsigned long long over = reinterpret_cast<const unsigned long long*>(smem)[2];
        __syncthreads();
        // an "auto" lean launch's oversize graphs (more processed slots than its 64 rows): the owned-rows code for node_num
        // slots (K = 10, node_num <= 128: it fits this kernel's 512 threads); it resets the flag and may raise 2 (the graph
        // needs the generic semantic branch) or 1 (f16 range)
#if SGPR_OVER_IN_REDO
        if constexpr (KP == 16) {
            while (over && kp.p3.big) {                      // workgroup-uniform
                const int bit = __ffsll((long long)over) - 1;
                const int slot = base + bit;
                over &= over - 1;
                embed_graph<KP, 0, 0, FMT_H2, 10, true>(kp, kp.p3, kp.a.ids ? kp.a.ids[slot] : slot, slot);
                __threadfence();
                __syncthreads();
                const int f = __hip_atomic_load(&kp.a.redo[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (f == 2) full |= 1ull << bit;
                if (f == 1) wide |= 1ull << bit;
            }
        }
#else
        (void)over;
#endif
        // the graphs that need the generic semantic branch first: the full f16 plan resets the slot's flag and raises it
        // again (1) when an activation or a coordinate leaves the f16 range - such a slot joins the wide-range list
        while (full) {                                       // workgroup-uniform
            const int bit = __ffsll((long long)full) - 1;
            const int slot = base + bit;
            full &= full - 1;
            embed_graph<KP, 0, 0, FMT_H2>(kp, kp.p2, kp.a.ids ? kp.a.ids[slot] : slot, slot);
            __threadfence();
            __syncthreads();                                 // LDS is reused by the next graph; thread 0's flag is visible
            if (__hip_atomic_load(&kp.a.redo[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 1) wide |= 1ull << bit;
        }
        while (wide) {
            const int slot = base + __ffsll((long long)wide) - 1;
            wide &= wide - 1;
            embed_graph<KP, 0, 0, FMTW>(kp, kp.p, kp.a.ids ? kp.a.ids[slot] : slot, slot);
            __syncthreads();
        }
    }
}

template <int KP, int DBG, int LEAN, int FMT, int KC = 0>
static int launch_t(const KParams& kp, hipStream_t stream) {
    static LdsLimitOnce once;
    int rc = raise_lds_limit(&once, reinterpret_cast<const void*>(&embed_kernel<KP, DBG, LEAN, FMT, KC>), kLdsLimit, "embed_kernel");
    if (rc != SGPR_OK) return rc;
    const int grid = kp.a.G * ((LEAN != 0 && DBG == 0 && kp.a.sem_tab) ? 2 : 1);
    hipLaunchKernelGGL((embed_kernel<KP, DBG, LEAN, FMT, KC>), dim3(grid), dim3(kp.p.nt), kp.p.lds_bytes, stream, kp);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "embed_kernel launch");
    return SGPR_OK;
}

template <int KP, int KC>
static int launch_big_t(const KParams& kp, hipStream_t stream) {
    static LdsLimitOnce once;
    int rc = raise_lds_limit(&once, reinterpret_cast<const void*>(&embed_big_kernel<KP, KC>), kLdsLimit, "embed_big_kernel");
    if (rc != SGPR_OK) return rc;
    int grid = kp.a.G;
    if (kp.a.auto_over == 2) {                        // persistent: as many workgroups as the device holds at once
        const int per_cu = kp.p.lds_bytes <= kLdsLimit / 2 && kp.p.nt <= 512 ? 2 : 1;
        if (grid > per_cu * kp.num_cus) grid = per_cu * kp.num_cus;
    }
    hipLaunchKernelGGL((embed_big_kernel<KP, KC>), dim3(grid), dim3(kp.p.nt), kp.p.lds_bytes, stream, kp);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "embed_big_kernel launch");
    return SGPR_OK;
}

template <int KP, int FMT>
static int launch_redo_t(const KParams& kp, int blocks, hipStream_t stream) {
    static LdsLimitOnce once;
    int rc = raise_lds_limit(&once, reinterpret_cast<const void*>(&embed_redo_kernel<KP, FMT>), kLdsLimit, "embed_redo_kernel");
    if (rc != SGPR_OK) return rc;
    int lds = kp.p.lds_bytes > kp.p2.lds_bytes ? kp.p.lds_bytes : kp.p2.lds_bytes;
    if (kp.p3.big && kp.p3.lds_bytes > lds) lds = kp.p3.lds_bytes;
    hipLaunchKernelGGL((embed_redo_kernel<KP, FMT>), dim3(blocks), dim3(kp.p.nt), lds, stream, kp);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "embed_redo_kernel launch");
    return SGPR_OK;
}

template <int KP, int DBG>
static int launch_layout(const EmbedPlan& plan, const KParams& kp, hipStream_t stream) {
    if (plan.fmt == FMT_H2) {
        if (DBG == 0 && plan.big)                                  // owned rows beyond 64 slots (K = 10 / 20 only: plan_big)
            return KP == 16 ? launch_big_t<16, 10>(kp, stream) : launch_big_t<32, 20>(kp, stream);
        if (KP == 16 && DBG == 0 && plan.lean && plan.k == 10)     // the reference's K: compile-time constant
            return plan.lean == 48 ? launch_t<16, 0, 48, FMT_H2, 10>(kp, stream) : launch_t<16, 0, 64, FMT_H2, 10>(kp, stream);
        if (KP == 32 && DBG == 0 && !plan.lean && plan.k == 20)    // the stress configuration's K: compile-time constant, too
            return launch_t<32, 0, 0, FMT_H2, 20>(kp, stream);
        if (plan.lean == 48 && DBG == 0) return launch_t<KP, 0, 48, FMT_H2>(kp, stream);
        if (plan.lean && DBG != 2) return launch_t<KP, DBG == 2 ? 0 : DBG, 64, FMT_H2>(kp, stream);
        return launch_t<KP, DBG, 0, FMT_H2>(kp, stream);
    }
    // wide-range layouts: forced by the debug mask (bit 13) - timers / dumps are not instantiated for them
    if (plan.fmt == FMT_BF3) return launch_t<KP, 0, 0, FMT_BF3>(kp, stream);
    return launch_t<KP, 0, 0, FMT_F32>(kp, stream);
}

int launch_embed(const sgpr_handle* h, const EmbedPlan& plan, const EmbedArgs& a, hipStream_t stream) {
    if (a.G == 0) return SGPR_OK;
    KParams kp;
    kp.w = h->w;
    kp.p = plan;
    kp.p2 = plan;
    kp.a = a;
    kp.num_cus = h->num_cus;
    if (kp.a.promise <= 0 || kp.a.promise > plan.N) kp.a.promise = plan.N;   // no promise made
    // "auto" launch (embed_common): a lean plan although nothing was promised - oversize graphs are handed on
    EmbedPlan over_plan;
    bool over_launch = false, over_in_redo = false;
    if (kp.a.auto_over == 1) {
        if (!(plan.fmt == FMT_H2 && plan.lean && a.redo && a.over_count)) {
            kp.a.auto_over = 0;
        } else {
            const int ocap = (kp.a.over_cap > 64 && kp.a.over_cap < plan.N) ? kp.a.over_cap : 0;
            over_launch = make_embed_plan(plan.N, ocap, plan.k, &over_plan, false, true) && over_plan.big;
            if (!over_launch) {
                kp.a.over_count = nullptr;                        // (no owned-rows instance for this K: the second pass's full plan)
            } else if (over_plan.nt <= 512 && plan.k == 10 && SGPR_OVER_IN_REDO) {
                kp.a.auto_over = 3;                               // node_num <= 128: the second pass runs the owned-rows code itself
                over_launch = false;
                over_in_redo = true;
            }
        }
    }
    // split launch: lean production plans on packed input when every workgroup of both halves gets a CU of its own
    // (measured: two workgroups sharing a CU cost each other more than the split saves - 64 graphs 38.0 -> 31.7 us per
    // call, 128 graphs 38.1 -> 34.6, 256 graphs 37.9 -> 44.2); the caller reserved sem_tab / sem_flag in the workspace
    const bool can_split = plan.fmt == FMT_H2 && plan.lean && !a.dense && !a.dbg_layers && !a.dbg_knn && !a.prof &&
                           !(a.skip & ~(8192 | (1 << 20))) && a.sem_tab && a.sem_flag && 2 * a.G <= h->num_cus;
    {
        // this launch's token (split-launch flags, redo_count, over_count): a counter spread over 32 bits - a fresh workspace
        // holds stale small integers (labels, flags), which a bare counter meets by chance in a process's first launches; a
        // match costs a second pass its scan of the flags, never a result
        static std::atomic<unsigned> epoch{0u};
        unsigned e = (++epoch) * 0x9E3779B1u;
        if (e == 0u) e = 0x9E3779B1u;
        kp.a.sem_epoch = e;
    }
    if (!can_split) {
        kp.a.sem_tab = nullptr;
        kp.a.sem_flag = nullptr;
    }
    // layer / kNN dumps run on the roomy instance; timers and ablation keep the production occupancy
    const int mode = (a.dbg_layers || a.dbg_knn) ? 2 : ((a.prof || (a.skip & ~(8192 | (1 << 20)))) ? 1 : 0);
    int rc;
    if (plan.kp == 16)
        rc = mode == 2 ? launch_layout<16, 2>(plan, kp, stream)
                       : (mode == 1 ? launch_layout<16, 1>(plan, kp, stream) : launch_layout<16, 0>(plan, kp, stream));
    else
        rc = mode == 2 ? launch_layout<32, 2>(plan, kp, stream)
                       : (mode == 1 ? launch_layout<32, 1>(plan, kp, stream) : launch_layout<32, 0>(plan, kp, stream));
    if (rc == SGPR_OK && over_launch && mode == 0) {
        // the graphs the lean launch could not hold, on the owned-rows instance sized for node_num (persistent workgroups)
        KParams ko = kp;
        ko.p = over_plan;
        ko.p2 = over_plan;
        ko.a.auto_over = 2;
        ko.a.sem_tab = nullptr;
        ko.a.sem_flag = nullptr;
        rc = over_plan.kp == 16 ? launch_big_t<16, 10>(ko, stream) : launch_big_t<32, 20>(ko, stream);
    }
    if (rc != SGPR_OK || plan.fmt != FMT_H2 || !a.redo || mode == 1) return rc;
    // second pass over the graphs the f16 instance flagged, on the wide-range plan (no node_cap: any graph fits)
    KParams kr = kp;
    const int nt_floor = over_in_redo ? (over_plan.nt > 256 ? 512 : 256) : 0;
    bool ok = make_embed_plan(plan.N, 0, plan.k, &kr.p, true, false, nt_floor) && make_embed_plan(plan.N, 0, plan.k, &kr.p2, false, false, kr.p.nt);
    if (ok && kr.p2.nt != kr.p.nt) ok = make_embed_plan(plan.N, 0, plan.k, &kr.p, true, false, kr.p2.nt);   // one block size for both
    kr.p3 = plan;
    kr.p3.big = 0;
    if (over_in_redo) kr.p3 = over_plan;
    if (!ok || kr.p.nt != kr.p2.nt) {
        set_error("no second-pass LDS plan for node_num " + std::to_string(plan.N));
        return SGPR_E_NODES;
    }
    kr.a.dbg_layers = nullptr;
    kr.a.dbg_knn = nullptr;
    kr.a.prof = nullptr;
    kr.a.skip = 0;
    // (an "auto" launch may hand this pass many graphs: a workgroup per CU)
#ifndef SGPR_REDO_BLOCKS
#define SGPR_REDO_BLOCKS 64   // persistent workgroups of the second pass (A/B builds: what the EMPTY pass costs against its grid)
#endif
    const int nblk = over_in_redo ? h->num_cus : SGPR_REDO_BLOCKS;
    const int blocks = a.G < nblk ? a.G : nblk;
    if (plan.kp == 16)
        return kr.p.fmt == FMT_BF3 ? launch_redo_t<16, FMT_BF3>(kr, blocks, stream) : launch_redo_t<16, FMT_F32>(kr, blocks, stream);
    return kr.p.fmt == FMT_BF3 ? launch_redo_t<32, FMT_BF3>(kr, blocks, stream) : launch_redo_t<32, FMT_F32>(kr, blocks, stream);
}

}  // namespace sgpr
