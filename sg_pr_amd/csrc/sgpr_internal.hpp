// Internal declarations shared by the translation units of libsgpr_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include <string>

#include "sgpr.h"

namespace sgpr {

// Architecture the kernels are written for (every shipped checkpoint; sgpr.h).
constexpr int kLabels = 12;
constexpr int kF1 = 64, kF2 = 64, kF3 = 32;
constexpr int kT = 16;   // tensor_neurons
constexpr int kB = 16;   // bottle_neck_neurons
constexpr int kKPad = 16;  // layer-1 inputs (3 / 12 channels) are zero-padded to one MFMA k-block

// Device-resident, kernel-ready weights (all pointers into one allocation).
//   EdgeConv layer l (order: s_conv1, s_conv2, s_conv3, f_conv1, f_conv2, f_conv3):
//     wf[l] : [2*cout][kp] row-major.  rows [0,cout)      = s * W[:, :C]          (acts on x_j)
//                                      rows [cout,2cout)  = s * (W[:, C:] - W[:, :C])  (acts on x_i)
//     tb[l] : [cout] = beta - mean * s,  s = gamma / sqrt(var + 1e-5)   (eval BatchNorm folded)
struct DevWeights {
    const float* wf[6];             // folded fp32 weights [2*cout][kp] (a rows, then b rows); host-side source of wb
    // the same weights as three bf16 planes (w = hi + mid + lo, exact to 24 bits) in MFMA operand order:
    // [column tile][k-step][plane][lane][8]  (lane = 16*lq + l15 holds row ct*16 + l15, k = 32*step + 8*lq + 0..7;
    // first layers, K = 16: one step, k = 4*lq + 0..3 then four zeros)
    const unsigned short* wb[6];
    // ... and as two f16 planes (w = hi + lo, 22 bits), same operand order with two planes per k-step: the default
    const unsigned short* wh[6];
    const float* tb[6];
    int kp[6];
    int cout[6];
    const float* wf_end;  // [32][64] folded
    const unsigned short* wb_end;   // conv_end in the wb layout (2 column tiles, 2 k-steps)
    const unsigned short* wh_end;   // conv_end in the wh layout
    const float* tb_end;  // [32]
    const float* att_w;   // [32][32]
    const float* ntn_w;   // [32][32*16]   (weight_matrix.view(F3,-1), col = j*16 + t)
    const float* ntn_wt;  // the same tensor as [i][t][j] (ntn_prep_kernel reads 16 consecutive j per lane group)
    const float* ntn_wb;  // [16][64]
    const float* ntn_bias;  // [16]
    const float* fc1_w;   // [16][16]
    const float* fc1_b;   // [16]
    const float* fc2_w;   // [16]
    const float* fc2_b;   // [1]
    // Semantic branch on label super-nodes, what does not depend on the graph (sgpr_embed.hip, sem_table_kernel; filled on
    // the device at sgpr_create by the instructions of the per-graph path; NULL: the per-graph path computes it).  A label
    // row of layer 1 has two versions - "fewer than K nodes carry the label" (bit 1: the padding representative is among
    // its neighbours) or not (bit 0) -, rows 12..15 (representative, unused) have one:
    const float* sem_g;   // [2][2][16][16] <x1_l (version p), x1_j (version q)> of the layer-1 rows (layer 2's Gram tile)
    const float* sem_xx;  // [2][16]        |x1_j|^2
    const float* sem_a2;  // [2][16][64]    layer 2's per-node term a = W1' x1
    const float* sem_b2;  // [2][16][64]    ... and b = (W2 - W1)' x1 + t
};
constexpr int kSemTableFloats = 4 * 256 + 32 + 2 * 2 * 16 * 64;

// Any-shape model (sgpr_generic.hip): the folded fp32 weights at the model's own dimensions, for what the tuned kernels
// are not built for - architectures beyond the built shape, node_num > SGPR_MAX_NODES, K > SGPR_MAX_K.
#define SGPR_GENERIC_MAX_LABELS SGPR_ANY_MAX_LABELS
#define SGPR_GENERIC_MAX_FILTERS SGPR_ANY_MAX_FILTERS
#define SGPR_GENERIC_MAX_F3 SGPR_ANY_MAX_FILTERS_3
#define SGPR_GENERIC_MAX_T SGPR_ANY_MAX_NEURONS          // tensor neurons and bottleneck neurons
#define SGPR_GENERIC_MAX_NODES SGPR_ANY_MAX_NODES
#define SGPR_GENERIC_MAX_K SGPR_ANY_MAX_K
struct GenericModel {
    int L, f1, f2, f3, T, B, cmax;      // cmax: widest activation row (max of 3, L, f1, f2, f3)
    int cin[6], cout[6];                // EdgeConv blocks: xyz branch (s_conv1..3), then the semantic branch (f_conv1..3)
    // (the 1x1 convolutions are stored input-channel-major with the output channels padded to a multiple of 8, cout8:
    //  the 8 weights a wave needs for one input channel are contiguous and wave-uniform)
    const float* wa[6];                 // [cin][cout8]  s * W[:, :cin]            (acts on x_j)
    const float* wb[6];                 // [cin][cout8]  s * (W[:, cin:] - W[:, :cin])   (acts on x_i)
    const float* tb[6];                 // [cout]        beta - mean * s,  s = gamma / sqrt(var + 1e-5)
    const float* w_end;                 // [2 f3][f3 rounded up to 8]   conv_end, BatchNorm folded
    const float* t_end;                 // [f3]
    const float* att_w;                 // [f3][f3]
    const float* ntn_w;                 // [f3][f3][T]
    const float* ntn_wb;                // [T][2 f3]
    const float* ntn_bias;              // [T]
    const float* fc1_w;                 // [B][T]
    const float* fc1_b;                 // [B]
    const float* fc2_w;                 // [B]
    const float* fc2_b;                 // [1]
};

// Matrix-core form of a GenericModel for MODERATELY larger architectures (sgpr_wide.hip): labels <= 32, filters_1 / 2 <= 128,
// filters_3 <= 64; every width padded to a multiple of 32 with zero weights (exact: a padded channel is lrelu(0 + 0) = 0
// and meets zero weights in the next layer).  Weights as two f16 planes in MFMA operand order, like DevWeights::wh.
#define SGPR_WIDE_MAX_LABELS 32
#define SGPR_WIDE_MAX_FILTERS 128
#define SGPR_WIDE_MAX_F3 64
#define SGPR_WIDE_MAX_NODES 112
struct WideModel {
    int ok;                                // 0: this architecture / these weights are not served (limits, f16 range)
    int L, f3, F3P;                        // F3P = filters_3 padded
    int cinP[6], coutP[6];                 // GenericModel's layer order: xyz layers 0..2, semantic 3..5
    const unsigned short* wh[6];           // [2 coutP / 16 column tiles: a rows, then b rows][cinP / 32 k-steps][2 planes][64 lanes][8]
    const float* tbp[6];                   // [coutP]
    const unsigned short* wh_end;          // [F3P / 16][2 F3P / 32][2][64][8]: conv_end on cat(xyz3 [F3P], sem3 [F3P])
    const float* tbp_end;                  // [F3P]
    const float* att_w;                    // [f3][f3] (the GenericModel's)
};

}  // namespace sgpr

struct sgpr_handle {
    int device;
    int num_cus;
    sgpr_dims dims;
    float* d_blob;       // owns the packed weights
    size_t blob_floats;
    int32_t* d_status;   // label-error flag
    sgpr::DevWeights w;
    // debug / ablation hooks (sgpr_debug_set_*): per handle, off by default; the only mutable state of a handle
    int dbg_skip;
    unsigned long long* dbg_prof;
    int f16_weights;     // every folded weight fits the f16 range (else the wide-range layouts are used throughout)
    int generic_only;    // the architecture is larger than the built shape: every call runs on the any-shape kernels
    float* d_gblob;      // owns the any-shape model's weights
    sgpr::GenericModel gm;
    void* d_wblob;       // owns the matrix-core form of the any-shape model (wm.ok != 0)
    sgpr::WideModel wm;
};

namespace sgpr {

// LDS plan of the embed kernel for one (N, k); computed on the host, passed by value.
struct EmbedPlan {
    int N;           // slots per graph in global memory (node_num)
    int NC;          // upper bound of the slots PROCESSED per graph (node_cap <= N): sizes every LDS buffer
    int NP, k, kp;   // NP = NC rounded up to 16
    int nt;          // threads per workgroup: 256 (LDS <= 80 KB: two or more workgroups per CU) or 512
    int pitchD;      // ints per row of the ranking-key chunk
    int RC;          // rows per key chunk (multiple of 16); RC == NP => symmetric Gram
    int P;           // lanes per row in the selection phase (power of two)
    int seg;         // candidates per lane (multiple of 4, <= 32)
    int kpitch;      // u16 entries per row of the neighbour list
    int pitchA;      // floats per row of the gather target A
    int overlap;     // 1: key matrix resident, selection (half the waves) overlaps the GEMMs (other half)
    int park_in_lds; // xyz3 parked in LDS (1) or in the global workspace (0)
    int park_hybrid; // park_in_lds == 0 only: the 16 super-node rows of the first branch (+ its scratch) still sit in LDS
                     // (offPark); only a graph that takes the generic semantic branch uses its rows of the global workspace
    int small_park;  // 1: the LDS park holds only the 16 super-node rows (lean production launches): a graph that needs the
                     // generic semantic branch is re-embedded by the second pass
    int alias_da;    // 1: the key matrix / chunk D shares the A region (a barrier separates selection and GEMMs)
    int lean;        // 1: NP <= 64 - one wave per 16-row tile, weight-stationary GEMMs, <= 168-VGPR kernel instance
    int big;         // owned-rows instance for plans beyond 64 rows (sgpr_embed.hip, embed_big_kernel): 0 no; 1 keys in the
                     // chunked plans' operation order, 2 in the resident plans' (what the full plan of the same launch uses)
    int xplanes;     // 1: X rows are bf16 / f16 planes; 0: fp32 rows (272 B) split when loaded
    int fmt;         // X layout: 2 = two f16 planes (272 B rows, the default), 1 = three bf16 planes (400 B), 0 = fp32 rows
    int rowb;        // bytes per X row
    int offX, offA, offD, offPark, offXX, offRed, offIdx;  // byte offsets into dynamic LDS
    int sem_wave;    // big != 0 only: the super-node branch has LDS of its own (offSem) and runs on ONE otherwise idle wave beside
    int offSem;      // the first xyz layer's selection (embed_graph); 0: on the whole workgroup, ahead of the xyz layers
    int lds_bytes;
};
bool make_embed_plan(int N, int node_cap, int k, EmbedPlan* plan, bool wide_range = false, bool small_park = false,
                     int min_nt = 0);

struct EmbedArgs {
    const float* centers;   // packed input, or
    const int32_t* labels;
    const long long* rag_off;       // ragged packed input (sgpr_embed_ragged): graph g owns nodes [rag_off[g], rag_off[g+1]) of
    const signed char* rag_lab;     // centers [S][3] / rag_lab [S]; the N - count padding slots exist only in registers
    const float* dense;     // dense [G][3+L][N] input (centers/labels NULL)
    const int32_t* ids;     // optional [G] graph indices: workgroup b embeds graph ids[b] (sgpr_embed_ordered)
    const float* dense2;    // optional second dense tensor: graphs g >= g_split read dense2[g - g_split]
    int g_split;
    int num_labels;         // label classes of the loaded checkpoint (<= kLabels): dense tensors have 3 + num_labels channels,
                            // a packed label outside [-1, num_labels) is an error (the reference raises KeyError, sg_net.py:277)
    int G;
    float* pooled;
    float* att;
    float* emb;
    float* dbg_layers;
    int32_t* dbg_knn;
    float* park_ws;         // [G][NP][32] when !park_in_lds
    unsigned char* redo;    // [launch slots] written by the f16 instance: 1 = the graph left the f16 range -> embed_redo_kernel
    unsigned* redo_count;   // one word: receives sem_epoch when any slot asks for the second pass (else untouched)
    // no promise from the caller and a plan for fewer slots than node_num (launch_embed, "auto" launches): a graph that
    // needs more processed slots than the plan holds is handed on instead of being an error - redo[slot] = 3 and the
    // launch's token in over_count (the owned-rows instance for node_num slots takes those graphs in a launch of its own),
    // or a plain second-pass request when over_count is NULL
    unsigned* over_count;
    int auto_over;          // 1: the hand-over above applies; 2 (embed_big_kernel only): persistent launch over the slots flagged 3
    int over_cap;           // slots the hand-over launch is sized for: the caller's node_cap when one above 64 was promised, else node_num
    // split launch (sgpr_embed.hip): workgroup s < G (a PRODUCER: the semantic half of launch slot s) publishes the 16
    // sem3 rows of the slot in sem_tab[s][16][32] and sem_flag[s] = token(s); workgroup G + s (the CONSUMER: the xyz half)
    // picks them up before conv_end.  Producers own the lower block indices and are therefore dispatched first - the
    // forward-progress argument of the hand-over (a waiting consumer never holds a CU its own producer still needs)
    // depends on that order.  sem_tab == NULL: the unsplit launch
    float* sem_tab;
    unsigned long long* sem_flag;
    unsigned sem_epoch;         // this launch's token: distinguishes its flags / redo_count from whatever the workspace held
    int32_t* status;
    unsigned long long* prof;   // optional [8] per-phase cycle counters (sgpr_debug_set_profile_buffer)
    int promise;                // the caller's node_cap (or N): checked even when the plan ignores it
    int skip;                   // debug/ablation only (sgpr_debug_set_profile_buffer's companion): phases to skip
};

void set_error(const std::string& msg);
int hip_fail(hipError_t e, const char* what);

int launch_embed(const sgpr_handle* h, const EmbedPlan& plan, const EmbedArgs& a, hipStream_t stream);
// fills table[kSemTableFloats] (layout: sem_g | sem_xx | sem_a2 | sem_b2) and *vmax_out (largest |layer-1 output|)
int launch_sem_tables(const DevWeights& w, float* d_table, float* d_vmax, hipStream_t stream);
int launch_score_pairs(const sgpr_handle* h, const float* p1, const int32_t* i1, const float* p2, const int32_t* i2,
                       int64_t P, float* score, hipStream_t stream);
size_t score_all_pairs_ws_bytes(int R, int M);
int launch_score_all_pairs(const sgpr_handle* h, const float* rows, int R, const float* cols, int M, float* score,
                           int64_t ld, void* ws, hipStream_t stream, bool wide = false);
size_t score_all_pairs_multi_ws_bytes(int n, const sgpr_pairs_job* jobs);
int launch_score_all_pairs_multi(const sgpr_handle* h, int n, const sgpr_pairs_job* jobs, void* ws, hipStream_t stream);
size_t score_pair_list_ws_bytes(int NR, int M);
int launch_score_pair_list(const sgpr_handle* h, const float* rows, const float* cols, int M, const int32_t* plan,
                           int NR, int NI, int64_t P, float* score, void* ws, hipStream_t stream);
int generic_embed_slots(const sgpr_handle* h, int G);
size_t generic_embed_ws_bytes(const sgpr_handle* h, int G, int N, int k);   // 0: the working memory fits LDS
size_t generic_embed_lds_bytes(const sgpr_handle* h, int N, int k);
int launch_embed_generic(const sgpr_handle* h, const EmbedArgs& a, int N, int k, void* ws, hipStream_t stream);
// the matrix-core any-shape embed: does this handle / launch take it, its LDS, the launch (flags the graphs whose values
// leave the f16 range in a.redo - the plain-fp32 kernel then embeds those, launch_embed_generic with a.auto_over == 7)
bool wide_embed_serves(const sgpr_handle* h, const EmbedArgs& a, int N, int k);
size_t wide_embed_lds_bytes(int N);
int launch_embed_wide(const sgpr_handle* h, const EmbedArgs& a, int N, int k, hipStream_t stream);
// ... and its dense all-pairs tail (pooled width <= 64, tensor / bottleneck neurons <= 32): *d_gate = the device word the
// plain-fp32 kernel behind the call tests (non-zero: inputs outside the f16 range, the rectangle is its)
bool wide_tail_serves(const sgpr_handle* h);
size_t wide_tail_ws_bytes(int R, int M);
int launch_score_all_pairs_wide_any(const sgpr_handle* h, const float* rows, int R, const float* cols, int M, float* score,
                                    int64_t ld, void* ws, const unsigned** d_gate, hipStream_t stream);
// list form (M == 0: pair p = (i1 ? i1[p] : p, i2 ? i2[p] : p) -> score[p]) or dense rectangle (M > 0: P = R * M pairs -> score[r * ld + c])
int launch_knn_any(const float* x, int B, int C, int N, int k, int64_t* idx, hipStream_t stream);
int launch_attention_any(const float* w, const float* emb, int B, int N, int F, float* rep, float* att, hipStream_t stream);
int launch_ntn_any(const float* w, const float* wb, const float* bias, const float* e1, const float* e2, int64_t P, int F,
                   int T, float* out, hipStream_t stream);
int launch_score_generic(const sgpr_handle* h, const float* p1, const int32_t* i1, const float* p2, const int32_t* i2,
                         int64_t P, int M, float* score, int64_t ld, hipStream_t stream, const unsigned* d_gate = nullptr);
int launch_score_plan_generic(const sgpr_handle* h, const float* rows, const float* cols, const int32_t* plan, int NR, int NI,
                              int64_t P, float* score, hipStream_t stream);
int launch_ntn(const float* w, const float* wb, const float* bias, const float* e1, const float* e2, int64_t B,
               float* out, hipStream_t stream);
int launch_knn(const float* x, int B, int C, int N, int k, int64_t* idx, hipStream_t stream);
int launch_graph_feature(const float* x, const int64_t* idx, int B, int C, int N, int k, float* out, hipStream_t stream);
int launch_attention_pool(const float* w, const float* emb, int B, int N, float* rep, float* att, hipStream_t stream);

size_t size_order_ws_bytes(int G);
int launch_size_order(const float* centers, const int32_t* labels, const long long* rag_off, int G, int N, int k,
                      int num_labels, int32_t* order, int32_t* info, void* ws, hipStream_t stream);

size_t cluster_ws_bytes(int P);
int launch_cluster_scan(const float* pts, int stride, const uint32_t* label, int P, int max_nodes, double* centers,
                        int32_t* node_labels, int32_t* node_sizes, int32_t* point_node, int32_t* num_nodes, void* ws,
                        hipStream_t stream);

int launch_graph_edges(const float* pts, int stride, const int32_t* point_node, int P, int n, const double* centers,
                       double* min_dis, void* ws, hipStream_t stream);

// Raising a kernel's dynamic-LDS limit (hipFuncSetAttribute) applies to the CURRENT device only: remembered per device
// (one bit each; a process that drives several GPUs raises it once on each), with a message that says what did not fit -
// a part with 64 KB of LDS per workgroup cannot run the kernels that stage a whole graph / histogram in 130 - 160 KB.
struct LdsLimitOnce {
    std::atomic<unsigned long long> done{0ull};   // (threads that drive different GPUs share it; setting the attribute twice is idempotent)
};
int raise_lds_limit(LdsLimitOnce* once, const void* kernel, int bytes, const char* what);

// makes `device` current for the lifetime of the object and restores the caller's device afterwards
struct DeviceGuard {
    int prev;
    bool switched;
    explicit DeviceGuard(int device) : prev(-1), switched(false) {
        if (hipGetDevice(&prev) == hipSuccess && prev != device) switched = hipSetDevice(device) == hipSuccess;
    }
    ~DeviceGuard() {
        if (switched) (void)hipSetDevice(prev);
    }
};

}  // namespace sgpr
