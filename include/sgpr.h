/*
 * sgpr.h - C ABI of the MI355X-native SG_PR pair-scoring engine (libsgpr_hip.so).
 *
 * The reference (kxhit/SG_PR) is pure Python/PyTorch and has no FFI layer; its
 * boundary for the hot path is the Python API of `sg_net.SG` plus the
 * `model.pth` state-dict layout.  This header is the C-ABI a
 * maintainer binds *below* that API: plain pointers and sizes, no torch types.
 * Each entry point names the reference code it replaces.
 *
 * Conventions
 *   - every `d_` pointer is DEVICE memory owned by the caller (fp32 / int32,
 *     dense row-major); `stream` is a hipStream_t passed as void* (NULL = the
 *     null stream).  Launches are asynchronous on that stream.
 *   - the engine owns only its packed-weights copy inside the handle; no
 *     allocation happens inside launch calls (workspaces are passed in).
 *   - return value 0 = SGPR_OK, negative = error; sgpr_last_error() gives the
 *     message of the last failing call on the calling thread.
 *   - a handle is immutable after create => safe to use from several
 *     streams/threads (the sgpr_debug_* hooks are the one exception: they are
 *     per handle, off by default and not meant for production).  One process
 *     per GPU.
 *   - every entry point that takes a handle runs on the handle's device and
 *     restores the caller's current device before it returns; the handle-free
 *     entry points (sgpr_knn, sgpr_graph_feature, sgpr_attention_pool,
 *     sgpr_ntn) run on the caller's current device.
 */
#ifndef SGPR_H
#define SGPR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SGPR_ABI_VERSION 8

enum {
    SGPR_OK = 0,
    SGPR_E_INVALID = -1,   /* NULL pointer / negative count                                   */
    SGPR_E_DIMS = -2,      /* architecture beyond the any-shape limits, or an entry point the tuned kernels alone serve */
    SGPR_E_NODES = -3,     /* node_num outside [k, SGPR_ANY_MAX_NODES], or a graph exceeded node_cap  */
    SGPR_E_K = -4,         /* K outside [1, SGPR_ANY_MAX_K] or K > node_num                    */
    SGPR_E_LABEL = -5,     /* a label outside [-1, num_labels) was seen (sgpr_check_status)    */
    SGPR_E_HIP = -6,       /* HIP runtime error (message has hipGetErrorString)                */
    SGPR_E_WORKSPACE = -7, /* workspace missing or too small                                   */
    SGPR_E_BLOB = -8       /* weights blob has the wrong number of floats                      */
};

/* the tuned kernels (matrix cores, one workgroup stages a whole graph in LDS): every shipped checkpoint and config */
#define SGPR_MAX_NODES 256
#define SGPR_MAX_K 32
/* the any-shape kernels (plain fp32, activations in a global scratch area): what the reference can be configured to
 * beyond that (parser_sg.py:12-22 takes any filters_* / tensor_neurons / bottle_neck_neurons / node_num / K) */
#define SGPR_ANY_MAX_LABELS 64
#define SGPR_ANY_MAX_FILTERS 256  /* filters_1, filters_2 */
#define SGPR_ANY_MAX_FILTERS_3 128
#define SGPR_ANY_MAX_NEURONS 64   /* tensor_neurons, bottle_neck_neurons */
#define SGPR_ANY_MAX_NODES 1024
#define SGPR_ANY_MAX_K 64

typedef struct sgpr_handle sgpr_handle;

/* Architecture hyper-parameters = the `arch:` block of the reference's
 * config.yml (parser_sg.py:12-18) + number_of_labels (sg_net.py:201-203).
 * The HIP kernels are written for the architecture of every shipped
 * checkpoint, {12, 64, 64, 32, 16, 16}, and serve every architecture that is
 * no larger in any of the six: sgpr_create embeds its tensors into the built
 * shapes with zero weights for the channels it does not have, which is exact
 * (device buffers keep the built widths: pooled [G, 32], emb [G, N, 32], the
 * missing channels are 0; dense features are [G, 3 + num_labels, N]).
 * A LARGER architecture (any of the six, up to the SGPR_ANY_MAX_* limits) gets
 * an "any-shape" handle: the same entry points on plain-fp32 kernels
 * (sgpr_generic.hip) - correct against the same oracle, not tuned; the embed of
 * a model with <= 32 labels and filters <= 128 / 128 / 64 runs on the matrix
 * cores for node_num <= 112, K = 10, and so does sgpr_score_all_pairs(_multi)
 * when filters_3 <= 64 and tensor / bottleneck neurons <= 32 and the caller
 * passes the workspace the _workspace_bytes call asks for (sgpr_wide.hip;
 * without a workspace: the plain-fp32 kernel) - with device
 * buffers of the model's own width (pooled [G, filters_3], emb [G, N, filters_3]:
 * sgpr_pooled_width).  Not served on such a handle: sgpr_embed_debug's dumps
 * (-> SGPR_E_DIMS; sgpr_score_pair_list walks its plan pair by pair there: the
 * bits of sgpr_score_pairs, no workspace).  Beyond the
 * SGPR_ANY_MAX_* limits -> SGPR_E_DIMS at sgpr_create.
 * node_num in (SGPR_MAX_NODES, SGPR_ANY_MAX_NODES] or K in (SGPR_MAX_K,
 * SGPR_ANY_MAX_K] run on the any-shape embed kernel on every handle (the pooled
 * width stays the handle's). */
typedef struct sgpr_dims {
    int32_t num_labels;
    int32_t filters_1;
    int32_t filters_2;
    int32_t filters_3;
    int32_t tensor_neurons;
    int32_t bottle_neck_neurons;
} sgpr_dims;

/* Number of floats sgpr_create expects in `weights` for `dims`.
 * Blob layout = the fp32 tensors of the reference state dict
 * (sg_net.py:45-76; SURVEY.md row a-ckpt), each flattened row-major, in this order:
 *   for blk in [dgcnn_s_conv1, dgcnn_f_conv1, dgcnn_s_conv2, dgcnn_f_conv2,
 *               dgcnn_s_conv3, dgcnn_f_conv3, dgcnn_conv_end]:
 *       blk.0.weight [Cout, Cin2], blk.1.weight (gamma), blk.1.bias (beta),
 *       blk.1.running_mean, blk.1.running_var            (each [Cout])
 *   attention.weight_matrix [F3,F3]
 *   tensor_network.weight_matrix [F3,F3,T], .weight_matrix_block [T,2*F3], .bias [T]
 *   fully_connected_first.weight [B,T], .bias [B]
 *   scoring_layer.weight [1,B], .bias [1]
 * (the seven int64 num_batches_tracked scalars are not part of the blob). */
size_t sgpr_weights_count(const sgpr_dims* dims);

/* Replaces SGTrainer.setup_model's load_state_dict + .cuda() (sg_net.py:158-176):
 * folds eval-mode BatchNorm into the 1x1 convs, re-lays the weights out for the
 * kernels and uploads them to `device`.  `weights` is HOST memory. */
int sgpr_create(const float* weights, size_t n_floats, const sgpr_dims* dims, int device, sgpr_handle** out);
void sgpr_destroy(sgpr_handle* h);

/* Floats per row of the handle's pooled / emb buffers: 32 (the built width) for every architecture the tuned kernels
 * serve, filters_3 on an any-shape handle.  sgpr_is_any_shape: 1 for a handle of a larger architecture. */
int sgpr_pooled_width(const sgpr_handle* h);
int sgpr_is_any_shape(const sgpr_handle* h);

/* Bytes of device workspace sgpr_embed* needs for (G graphs, N slots): one flag byte per launch slot (written by the
 * f16-plane kernel instance, read by its wide-range / generic-branch second pass in the same call) plus, for N > 128,
 * the parked output of the first EdgeConv branch.  Never 0: always pass the workspace. */
size_t sgpr_embed_workspace_bytes(const sgpr_handle* h, int G, int N, int k);

/* Per-graph half of SG.forward: SG.dgcnn_conv_pass (sg_net.py:79-110 ->
 * dgcnn.knn / get_graph_feature dgcnn.py:14-49, the six EdgeConv blocks
 * sg_net.py:50-73, conv_end sg_net.py:74-76) + AttentionModule.forward
 * (layers_batch.py:28-39), fused in one kernel, one workgroup per graph.
 *   d_centers [G,N,3] f32 (0 for padded slots), d_labels [G,N] i32 (-1 = pad)
 *     = the packed form of transfer_to_torch's output (sg_net.py:250-299)
 *   d_pooled [G,F3] (required); d_att [G,N] and d_emb [G,N,F3] may be NULL. */
int sgpr_embed(const sgpr_handle* h, const float* d_centers, const int32_t* d_labels, int G, int N, int k,
               float* d_pooled, float* d_att, float* d_emb, void* d_workspace, size_t workspace_bytes,
               void* stream);
/* Without a node_cap promise (this call; node_cap 0 below) a launch of more graphs than the device has CUs still runs on
 * the 64-row layout (K <= 16): a graph that needs more processed slots is embedded in the same call by the kernel
 * instance sized for N (same bits; no error) - data whose graphs mostly exceed 64 processed slots is served faster
 * with its node_cap (sgpr_size_order), which sizes one launch for all of them. */

/* sgpr_embed with a promise about the input: no graph of the batch needs more than `node_cap` PROCESSED slots
 * (= slots before the trailing run of m identical padding slots, + 1 when m >= k, + m otherwise; 0 = no promise).
 * The kernel sizes its LDS for node_cap instead of N: caps <= 64 (<= 48) select a fixed 64-row (48-row) layout that
 * runs four (five) workgroups per CU; caps of 65 .. 96 (K <= 16, more graphs than CUs) keep the graphs of up to 64 slots
 * on that layout and run only the others on the instance sized for node_cap; beyond, one launch sized for node_cap.  A graph that breaks the promise gets a NaN pooled vector and
 * sgpr_check_status returns SGPR_E_NODES.  Results are otherwise identical to sgpr_embed.
 * On the any-shape kernels (an architecture beyond the built shape, N > SGPR_MAX_NODES, K > SGPR_MAX_K) node_cap is
 * advisory: they size nothing by it and do not check it. */
int sgpr_embed_capped(const sgpr_handle* h, const float* d_centers, const int32_t* d_labels, int G, int N, int node_cap,
                      int k, float* d_pooled, float* d_att, float* d_emb, void* d_workspace, size_t workspace_bytes,
                      void* stream);

/* sgpr_embed_capped over an explicit launch order: workgroup b embeds graph d_order[b] (device i32 [n_order],
 * distinct indices in [0, G), n_order <= G).  A packed graph store lists its graphs largest-first (longest
 * workgroups start first, the last round of the launch is filled by the smallest graphs - KITTI-00: -7 %), or lists
 * only the graphs that changed.  Outputs are indexed by graph id exactly as in sgpr_embed and are bit-identical to
 * it; a graph listed nowhere is not embedded and its output rows are left untouched. */
int sgpr_embed_ordered(const sgpr_handle* h, const float* d_centers, const int32_t* d_labels, int G, int N, int node_cap,
                       int k, const int32_t* d_order, int n_order, float* d_pooled, float* d_att, float* d_emb,
                       void* d_workspace, size_t workspace_bytes, void* stream);

/* The data-set properties sgpr_embed_capped / _ordered / _ragged take, computed on the device (two small launches, no
 * host pass): d_order [G] i32 = graph indices by PROCESSED slots, largest first, stable; d_info [2] i32 = { node_cap =
 * the largest processed-slot count of the batch, graphs beyond 64 processed slots }.  Padded arrays (d_centers, d_labels;
 * d_offsets NULL) or a ragged store (d_offsets [G+1] i64; d_centers / d_labels unused, may be NULL).  The caller reads
 * d_info[0] back once per data set (4 bytes) and passes it as node_cap.  The reference has no counterpart: it pads
 * every graph to node_num (sg_net.py:258-272) and pays for node_num slots.  Workspace: sgpr_size_order_workspace_bytes(G). */
size_t sgpr_size_order_workspace_bytes(int G);
int sgpr_size_order(const sgpr_handle* h, const float* d_centers, const int32_t* d_labels, const int64_t* d_offsets, int G,
                    int N, int k, int32_t* d_order, int32_t* d_info, void* d_workspace, size_t workspace_bytes,
                    void* stream);

/* sgpr_embed_ordered over a RAGGED graph store: only the real nodes of a graph are in memory -
 *   d_centers [S,3] f32, d_labels [S] i8 (0 .. num_labels-1), d_offsets [G+1] i64: graph g owns nodes
 *   [d_offsets[g], d_offsets[g+1]) -
 * and the zero padding up to N = node_num slots that transfer_to_torch appends (sg_net.py:258-272) is made in
 * registers by the kernel.  13 bytes per real node instead of 16 per slot (KITTI-like graphs: 2.8x fewer bytes to hold
 * and to move across PCIe); results are bit-identical to sgpr_embed on the padded arrays.  d_order may be NULL
 * (n_order ignored: all G graphs in index order).  A graph with more than N nodes gets a NaN pooled vector and
 * sgpr_check_status returns SGPR_E_NODES.  Outputs (d_att [G,N], d_emb [G,N,F3]) keep the padded shape. */
int sgpr_embed_ragged(const sgpr_handle* h, const float* d_centers, const int8_t* d_labels, const int64_t* d_offsets,
                      int G, int N, int node_cap, int k, const int32_t* d_order, int n_order, float* d_pooled,
                      float* d_att, float* d_emb, void* d_workspace, size_t workspace_bytes, void* stream);

/* Same as sgpr_embed, taking the reference's dense tensor `features` [G, 3+L, N] f32
 * (data["features_1"], sg_net.py:119) - the sem block may hold any values. */
int sgpr_embed_dense(const sgpr_handle* h, const float* d_features, int G, int N, int k, float* d_pooled,
                     float* d_att, float* d_emb, void* d_workspace, size_t workspace_bytes, void* stream);

/* sgpr_embed + dumps of every intermediate for parity tests:
 *   d_layers [G,6,N,64] f32  outputs of s_conv1..3, f_conv1..3 after max-k (channels >= Cout are 0)
 *   d_knn    [G,6,N,k]  i32  neighbour lists chosen at the input of those six layers (unordered sets). */
int sgpr_embed_debug(const sgpr_handle* h, const float* d_centers, const int32_t* d_labels, int G, int N, int k,
                     float* d_pooled, float* d_att, float* d_emb, float* d_layers, int32_t* d_knn,
                     void* d_workspace, size_t workspace_bytes, void* stream);

/* Pair-coupled half of SG.forward: TenorNetworkModule.forward
 * (layers_batch.py:70-83) + fully_connected_first/ReLU + scoring_layer/sigmoid
 * (sg_net.py:131-136) for P pairs.  Pair p scores
 *   (d_pooled1[idx1 ? idx1[p] : p], d_pooled2[idx2 ? idx2[p] : p]); idx may be NULL. */
int sgpr_score_pairs(const sgpr_handle* h, const float* d_pooled1, const int32_t* d_idx1,
                     const float* d_pooled2, const int32_t* d_idx2, int64_t P, float* d_score, void* stream);

/* The same tail for a pair LIST grouped by row graph - the shape of the reference's own evaluation loop
 * (eval_batch.py:30-36 walks `<seq>.txt`, utils.py:61-70: 10^4 - 10^5 listed pairs over 10^3 graphs, ~15 per row graph).
 * sgpr_score_pairs spends a whole wave and a 64 KB weight read on every pair; here the bilinear form is hoisted per
 * DISTINCT row graph and the listed columns of a row go through the matrix cores 16 at a time, exactly like a row of
 * the dense rectangle: the scores are bit-identical to sgpr_score_all_pairs' entries at the listed (row, column).
 *
 * sgpr_pair_plan (HOST memory in and out, no GPU work): groups pair p = (idx1[p], idx2[p]), 0 <= idx1 < R, 0 <= idx2 < M,
 * P < 2^31, by row graph (stable: a row's pairs keep their list order) and cuts every row's pairs into work items of
 * <= 16.  h_plan receives int32 words
 *     row_ids [n_rows] | item_row [n_items] | item_begin [n_items + 1] | cols [P] | pos [P]
 * (distinct row graphs ascending; per item its row as an index into row_ids and its first pair in cols / pos; per pair
 * its column graph and its position in the caller's list).  sgpr_pair_plan_ints(P, R) bounds the words needed;
 * *plan_ints returns the words written.  An index out of range -> SGPR_E_INVALID.  Build it once per list (like a launch
 * order), copy it to the device, reuse it for every call.
 * sgpr_score_pair_list: d_score[p] = SG-tail(d_pooled_rows[idx1[p]], d_pooled_cols[idx2[p]]) for the P pairs of the plan
 * (d_plan: the plan words in DEVICE memory; R, M, n_rows, n_items and P are the values sgpr_pair_plan was given and
 * returned - the kernels trust the plan's indices, as sgpr_score_pairs trusts idx1 / idx2).  Workspace:
 * sgpr_score_pair_list_workspace_bytes(h, n_rows, M). */
size_t sgpr_pair_plan_ints(int64_t P, int R);
int sgpr_pair_plan(const int32_t* h_idx1, const int32_t* h_idx2, int64_t P, int R, int M, int32_t* h_plan,
                   size_t plan_capacity_ints, size_t* plan_ints, int32_t* n_rows, int32_t* n_items);
size_t sgpr_score_pair_list_workspace_bytes(const sgpr_handle* h, int n_rows, int M);
int sgpr_score_pair_list(const sgpr_handle* h, const float* d_pooled_rows, int R, const float* d_pooled_cols, int M,
                         const int32_t* d_plan, int n_rows, int n_items, int64_t P, float* d_score, void* d_workspace,
                         size_t workspace_bytes, void* stream);

/* Dense all-pairs form of the same tail: score[r, c] = SG-tail(rows[r], cols[c])
 * (the NTN is asymmetric, layers_batch.py:77-83, so the full rectangle is computed).
 * d_score is [R, ld] with ld >= M.  Workspace: sgpr_score_all_pairs_workspace_bytes. */
size_t sgpr_score_all_pairs_workspace_bytes(const sgpr_handle* h, int R, int M);
int sgpr_score_all_pairs(const sgpr_handle* h, const float* d_pooled_rows, int R, const float* d_pooled_cols,
                         int M, float* d_score, int64_t ld, void* d_workspace, size_t workspace_bytes,
                         void* stream);

/* Several independent rectangles with ONE pair of launches - the matrices of the sequences of an evaluation job
 * (eval_batch.py:26-36 loops over `eva_batch.sequences`): the work items of all jobs form one list that the workgroups
 * split evenly, so small matrices do not leave the GPU half empty and the launch gaps between them disappear.  `jobs` is
 * a HOST array (its device pointers are read at launch); results are bit-identical to one sgpr_score_all_pairs per job. */
#define SGPR_MAX_PAIR_JOBS 8
typedef struct sgpr_pairs_job {
    const float* d_pooled_rows;   /* [R][32] */
    int R;
    const float* d_pooled_cols;   /* [M][32] */
    int M;
    float* d_score;               /* [R][ld] */
    int64_t ld;
} sgpr_pairs_job;
size_t sgpr_score_all_pairs_multi_workspace_bytes(const sgpr_handle* h, int n_jobs, const sgpr_pairs_job* jobs);
int sgpr_score_all_pairs_multi(const sgpr_handle* h, int n_jobs, const sgpr_pairs_job* jobs, void* d_workspace,
                               size_t workspace_bytes, void* stream);

/* Drop-in SG.forward (sg_net.py:112-138): dense features of both sides in,
 * (score [B], att1 [B,N], att2 [B,N]) out; att pointers may be NULL. */
size_t sgpr_forward_workspace_bytes(const sgpr_handle* h, int B, int N, int k);
int sgpr_forward_dense(const sgpr_handle* h, const float* d_features_1, const float* d_features_2, int B, int N,
                       int k, float* d_score, float* d_att1, float* d_att2, void* d_workspace,
                       size_t workspace_bytes, void* stream);

/* Synchronises `stream` and returns SGPR_E_LABEL if any launch on this handle
 * saw a label outside [-1, num_labels) since the last check (the reference
 * raises KeyError at sg_net.py:277), else SGPR_OK.  Clears the flag. */
int sgpr_check_status(const sgpr_handle* h, void* stream);

/* ---- consumers of the score matrix that keep it on the device ----------------------------------------
 *
 * sgpr_pair_positives + sgpr_pair_threshold_counts: the counting half of eval_batch.py:48-49 and 69-87 (sklearn roc_curve
 * / auc and precision_recall_curve -> F1 max) on an R x M score rectangle (rows row0 .. row0+R-1 of the square matrix).
 * Ground truth comes from the planar poses d_pose_xz [.][2] (float64 x, z of the KITTI pose; the distance is evaluated
 * in float64 operation by operation like utils.py:36): distance <= d_pos positive, >= d_neg negative, in between
 * ignored (the pairs the reference refuses, sg_net.py:302-309) - or, when d_pose_xz is NULL, from explicit labels
 * d_gt [R][ldg] (1 / 0 / negative = ignore).
 *
 * sgpr_pair_positives appends the scores of the positive pairs to d_out (at most `capacity` of them, in no particular
 * order; capacity 0 / d_out NULL = count only) and sets d_count[0] = number of positive pairs with a usable score,
 * d_count[1] = positive pairs whose score is negative or NaN (no defined rank; they are skipped).
 *
 * sgpr_pair_threshold_counts streams the rectangle once and counts the NEGATIVE pairs by threshold bucket: for T <=
 * 8191 ascending thresholds, d_out[b] (uint64, b = 0..T) = negatives with exactly b thresholds <= their score, so
 * FP(score >= threshold q) = sum of d_out[b], b > q.  d_out[T+1] = negatives skipped for a negative / NaN score.
 * With d_rank every negative is also ranked among ALL distinct scores of positive pairs: the thresholds are every S-th
 * of those ascending values, and d_rank holds, for threshold q, the S values from it up to the next threshold in
 * groups_per_threshold records of eight (padded with +inf / 0 pairs) together with the number of pairs that carry each;
 * d_at_least[q] = positive pairs with a score >= threshold q.  d_out[T+2] = sum over negatives of 2 #{positive pairs
 * > s} + #{positive pairs == s} = 2 P N AUC (the Mann-Whitney form of sklearn's trapezoid area): the ROC area exactly,
 * in the same pass.
 * F1 peaks at the score of a positive pair, so the host (sg_pr_amd/metrics.py) takes every S-th distinct positive value
 * as thresholds, reads exact F1 there and bounds in between, and settles the few segments that can still hold the
 * maximum with a second call - exact, no sort of the matrix, which never leaves the GPU.  The workspace holds one
 * slab of counters per workgroup (no global atomics). */
typedef struct sgpr_rank_group {
    float value[8];             /* ascending distinct scores of positive pairs (+inf padding) */
    uint32_t pairs[8];          /* positive pairs with exactly that score (0 for padding) */
} sgpr_rank_group;
int sgpr_pair_positives(const sgpr_handle* h, const float* d_score, int R, int M, int64_t ld, int row0,
                        const double* d_pose_xz, double d_pos, double d_neg, const signed char* d_gt, int64_t ldg,
                        float* d_out, int64_t capacity, unsigned long long* d_count, void* stream);
size_t sgpr_pair_threshold_counts_workspace_bytes(const sgpr_handle* h, int T);
int sgpr_pair_threshold_counts(const sgpr_handle* h, const float* d_score, int R, int M, int64_t ld, int row0,
                               const double* d_pose_xz, double d_pos, double d_neg, const signed char* d_gt,
                               int64_t ldg, const float* d_thresholds, int T, const sgpr_rank_group* d_rank,
                               int groups_per_threshold, const unsigned long long* d_at_least, unsigned long long* d_out,
                               void* d_workspace, size_t workspace_bytes, void* stream);

/* F1-max of a score rectangle (eval_batch.py:69, 85-87) in ONE call, no host round trip between its steps: one
 * streaming pass classifies every pair once - negatives into a histogram over the score's bit pattern (one shift, one LDS
 * atomic), the scores of the positive pairs into a list -, exact F1 at every bin edge and bounds for the positives
 * inside the bins follow from suffix sums, and a second pass settles the few bins that can still hold the maximum
 * (their positives sorted and de-duplicated in LDS; a negative outside them costs one bit test).  Same ground truth
 * arguments as sgpr_pair_positives.  d_result (device, 8 doubles): [0] F1-max (exact), [1] status - 0 ok, 1 the
 * rectangle needs the multi-call path (more than 2^20 positive pairs, or more than 4095 positive scores left to settle:
 * a flat curve), 2 negative / NaN scores among the labelled pairs -, [2] positive pairs, [3] negative pairs, [4] passes
 * over the matrix, [5] bins of the first pass, [6] values settled by the second.  Asynchronous on `stream`; the caller
 * copies d_result when it needs it. */
size_t sgpr_f1_max_workspace_bytes(const sgpr_handle* h, int R, int M);
int sgpr_f1_max(const sgpr_handle* h, const float* d_score, int R, int M, int64_t ld, int row0, const double* d_pose_xz,
                double d_pos, double d_neg, const signed char* d_gt, int64_t ldg, double* d_result, void* d_workspace,
                size_t workspace_bytes, void* stream);

/* Loop-closure candidates (the use the reference makes of a sequence's similarity matrix, README.md:92-97): for every
 * row r the k (1, 4, 8 or 16) best-scoring columns c with |c - (row0 + r)| > window (window = -1 keeps every column),
 * ordered by (score descending, column ascending); d_values / d_indices [R][k], index -1 where fewer than k columns
 * qualify.  One wave per row, one pass over the row. */
int sgpr_topk_rows(const sgpr_handle* h, const float* d_score, int R, int M, int64_t ld, int row0, int window, int k,
                   float* d_values, int32_t* d_indices, void* stream);

/* LDS bytes per workgroup the embed kernel uses for (N, k) on this handle; 0 if unsupported. */
size_t sgpr_embed_lds_bytes(const sgpr_handle* h, int N, int k);

/* ---- stand-alone forms of the reference's building blocks (SURVEY.md 8b "signatures to keep") -------------------
 * Inside sgpr_embed / sgpr_forward_dense these run fused and never materialise their outputs; the entry points below
 * serve callers of the individual symbols.  They need no handle: a stand-alone module owns its own parameters, which
 * are passed as device pointers.
 *
 * sgpr_knn replaces dgcnn.knn (dgcnn.py:14-20): d_x [B,C,N] f32 -> d_idx [B,N,k] int64 (torch.topk's index type), the
 * k nearest candidates of every node under pd[i][j] = -|x_j|^2 + 2 x_i.x_j - |x_i|^2, best first; equal distances keep
 * the lower candidate index first (torch.topk's tie order is implementation-defined).  N <= SGPR_ANY_MAX_NODES, k <= N,
 * k <= SGPR_ANY_MAX_K (beyond SGPR_MAX_NODES / SGPR_MAX_K: one wave per row instead of the LDS-resident kernel). */
int sgpr_knn(const float* d_x, int B, int C, int N, int k, int64_t* d_idx, void* stream);

/* Replaces dgcnn.get_graph_feature (dgcnn.py:23-49) for given neighbour lists d_idx [B,N,k] (int64, from sgpr_knn or
 * the caller): d_out [B,2C,N,k] f32 = cat(x_j - x_i, x_i) in the reference's channel order (dgcnn.py:47). */
int sgpr_graph_feature(const float* d_x, const int64_t* d_idx, int B, int C, int N, int k, float* d_out, void* stream);

/* Replaces AttentionModule.forward (layers_batch.py:28-39): d_weight [F3,F3] (attention.weight_matrix),
 * d_emb [B,N,F3] -> d_rep [B,F3] (the graph-level representation) and d_att [B,N] (sigmoid scores; may be NULL).
 * No padding mask, divisor N - exactly like the reference.  F3 = 32. */
int sgpr_attention_pool(const float* d_weight, const float* d_emb, int B, int N, float* d_rep, float* d_att,
                        void* stream);

/* The same module at any width F <= SGPR_ANY_MAX_FILTERS_3 (d_weight [F,F], d_emb [B,N,F], d_rep [B,F]); plain fp32. */
int sgpr_attention_pool_any(const float* d_weight, const float* d_emb, int B, int N, int F, float* d_rep, float* d_att,
                            void* stream);

/* Replaces TenorNetworkModule.forward (layers_batch.py:70-83): d_weight [F3,F3,T], d_weight_block [T,2*F3],
 * d_bias [T], d_e1 / d_e2 [B,F3] -> d_out [B,T] = relu(e1^T W e2 + Wb [e1;e2] + bias).  F3 = 32, T = 16. */
int sgpr_ntn(const float* d_weight, const float* d_weight_block, const float* d_bias, const float* d_e1,
             const float* d_e2, int64_t B, float* d_out, void* stream);

/* The same module at any width F <= SGPR_ANY_MAX_FILTERS_3, T <= SGPR_ANY_MAX_NEURONS tensor neurons (d_weight [F,F,T],
 * d_weight_block [T,2F], d_bias [T], d_e1 / d_e2 [B,F] -> d_out [B,T]); plain fp32. */
int sgpr_ntn_any(const float* d_weight, const float* d_weight_block, const float* d_bias, const float* d_e1,
                 const float* d_e2, int64_t B, int F, int T, float* d_out, void* stream);

/* ---- upstream of the path: labelled LiDAR scan -> semantic-graph nodes -------------------------
 * Replaces, for one scan, gen_labels + the node half of gen_graphs (data_process/gen_label_graph.py:196-365):
 * raw SemanticKITTI labels are remapped (learning_map, :23-58); road / parking and the discarded classes produce no
 * node; a class that carries instance ids is grouped by instance (groups of <= 20 points dropped, :274); every other
 * class is clustered like PCL's EuclideanClusterExtraction (connected components of "squared distance < tolerance^2",
 * tolerance 0.2 / 0.5 / 2 m and minimum size 50..300 by class, maximum 50 000, :283-305); each surviving cluster whose
 * class is in node_map (:64-77) becomes a node: label = node_map[class], centre = mean of its points.
 *   d_points  [P, point_stride] f32 (x, y, z first; point_stride >= 3, 4 for KITTI .bin scans)
 *   d_labels  [P] u32 = semantic id | instance id << 16 (the .label file format)
 *   outputs   d_centers [max_nodes,3] f64, d_node_labels / d_node_sizes [max_nodes] i32, in the reference's node order
 *             (class ascending; instance id ascending / cluster size descending, lowest point index first among equal
 *             sizes - PCL leaves that last order unspecified); d_point_node [P] i32 = node of each point or -1 (may be
 *             NULL); d_num_nodes [1] i32 = number of nodes found (may exceed max_nodes: then only the first max_nodes are
 *             written; -1 if more than 8192 clusters qualified).
 * Results do not depend on execution order (integer fixed-point centroid sums, lowest-index roots).  Handle-free; runs
 * on the caller's current device. */
size_t sgpr_cluster_workspace_bytes(int P);
int sgpr_cluster_scan(const float* d_points, int point_stride, const uint32_t* d_labels, int P, int max_nodes,
                      double* d_centers, int32_t* d_node_labels, int32_t* d_node_sizes, int32_t* d_point_node,
                      int32_t* d_num_nodes, void* d_workspace, size_t workspace_bytes, void* stream);

/* The edge rule of gen_graphs (gen_label_graph.py:367-385) for the n nodes of sgpr_cluster_scan: d_min_dis [n,n] f64 =
 * for i != j the distance between the point of cluster i and the point of cluster j that lie nearest to the midpoint of
 * the two centres (0 on the diagonal); the caller keeps the pairs i < j with distance <= 5 m as edges of weight
 * 1 - d/5.  The scorer never reads edges (utils.py:21-38 loads nodes, centers and pose only); this serves writers of
 * the reference's graph JSON.  d_workspace: n*n*4 bytes. */
int sgpr_graph_edges(const float* d_points, int point_stride, const int32_t* d_point_node, int P, int n,
                     const double* d_centers, double* d_min_dis, void* d_workspace, size_t workspace_bytes, void* stream);

/* Debug (per handle; not thread-safe; never set in production): a device array of 16 uint64 counters makes
 * sgpr_embed* run its profiling instance and add the
 * shader cycles wave 0 of every workgroup spends in each phase, barrier to barrier (0 stage, 1 select, 2 Gram,
 * 3 GEMM, 5 gather-max, 6 conv_end, 7 attention; 8..13 = sub-phases of the selection in sgpr_embed_debug with mask
 * bit 7).  The timers perturb the kernel (~1.6x); use the ablation mask for magnitudes.  NULL (default) disables. */
void sgpr_debug_set_profile_buffer(sgpr_handle* h, void* d_counters);

/* Debug / ablation timing only (results become invalid): bit 0 skips the kNN selection, bit 1 the per-node GEMMs,
 * bit 2 the Gram phase, bit 3 the gather-max; bit 4 returns right after dispatch, bit 5 after the input fetch;
 * bit 8 = nothing skipped (just selects the profiling instance); bits 9/10/11 keep the GEMM phase but drop its weight
 * loads / its MFMAs / its inner barrier; bit 12 runs the generic first semantic layer instead of the label lookup (valid
 * results); bit 13 forces the wide-range instance and bit 20 makes the producers of the split launch's odd slots
 * withhold their flag, so that their graphs reach the second pass through the late-producer path (both: valid
 * results, production kernels).  0 (default) = normal. */
void sgpr_debug_set_skip_mask(sgpr_handle* h, int mask);

/* 1: the handle's weights run on the default datapath (two f16 planes per matrix operand); 0: the checkpoint's folded
 * weights (or the super-node tables made from them) leave the f16 range, and every launch uses the wide-range
 * instance (three bf16 planes / fp32 rows).  Decided once, at sgpr_create.  (Weights BELOW f16's normal range keep an
 * absolute 2^-25 in the planes: harmless - the shipped checkpoints hold whole channels of 1e-30 .. 1e-7 behind dead
 * BatchNorm scales.) */
int sgpr_debug_uses_f16_planes(const sgpr_handle* h);

const char* sgpr_last_error(void);
int sgpr_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* SGPR_H */
