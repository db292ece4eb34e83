/* Stand-alone C caller of libsgpr_hip.so - no Python, no PyTorch: the C-ABI of include/sgpr.h is the whole boundary.
 *
 *   sgpr_demo <weights.f32> <graphs.bin> <scores.f32>
 *
 * weights.f32 : the fp32 weights blob (sgpr_weights_count floats, order documented in sgpr.h)
 * graphs.bin  : int32 G, int32 N, int32 K, then centers f32 [G][N][3], then labels i32 [G][N]  (-1 = padding)
 * scores.f32  : output, the dense G x G similarity matrix (row-major fp32)
 *
 * Build:  gcc -O2 -std=c11 -D__HIP_PLATFORM_AMD__ examples/sgpr_demo.c -Iinclude -I/opt/rocm/include -Lsg_pr_amd/lib -lsgpr_hip \
 *             -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,'$ORIGIN/../sg_pr_amd/lib' -Wl,-rpath,/opt/rocm/lib -o examples/sgpr_demo
 */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "sgpr.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define CHECK_SGPR(x) do { int r_ = (x); if (r_ != SGPR_OK) { fprintf(stderr, "%s -> %d: %s\n", #x, r_, sgpr_last_error()); return 3; } } while (0)

static void* read_file(const char* path, size_t* bytes) {
    FILE* f = fopen(path, "rb");
    if (!f) return NULL;
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    void* p = malloc((size_t)n);
    if (p && fread(p, 1, (size_t)n, f) != (size_t)n) { free(p); p = NULL; }
    fclose(f);
    *bytes = (size_t)n;
    return p;
}

int main(int argc, char** argv) {
    if (argc != 4) { fprintf(stderr, "usage: %s weights.f32 graphs.bin scores.f32\n", argv[0]); return 1; }
    size_t wbytes = 0, gbytes = 0;
    float* weights = (float*)read_file(argv[1], &wbytes);
    int32_t* graphs = (int32_t*)read_file(argv[2], &gbytes);
    if (!weights || !graphs || gbytes < 12) { fprintf(stderr, "cannot read the inputs\n"); return 1; }
    const int G = graphs[0], N = graphs[1], K = graphs[2];
    const size_t nc = (size_t)G * N * 3, nl = (size_t)G * N;
    if (gbytes != 12 + (nc + nl) * 4) { fprintf(stderr, "graphs.bin has the wrong size\n"); return 1; }

    sgpr_dims dims = {12, 64, 64, 32, 16, 16};
    if (wbytes != sgpr_weights_count(&dims) * sizeof(float)) { fprintf(stderr, "weights blob has the wrong size\n"); return 1; }
    sgpr_handle* h = NULL;
    CHECK_SGPR(sgpr_create(weights, wbytes / sizeof(float), &dims, 0, &h));

    float *d_centers, *d_pooled, *d_score;
    int32_t* d_labels;
    void *d_ws1 = NULL, *d_ws2 = NULL;
    CHECK_HIP(hipMalloc((void**)&d_centers, nc * 4));
    CHECK_HIP(hipMalloc((void**)&d_labels, nl * 4));
    CHECK_HIP(hipMalloc((void**)&d_pooled, (size_t)G * 32 * 4));
    CHECK_HIP(hipMalloc((void**)&d_score, (size_t)G * G * 4));
    CHECK_HIP(hipMemcpy(d_centers, graphs + 3, nc * 4, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(d_labels, graphs + 3 + nc, nl * 4, hipMemcpyHostToDevice));
    const size_t ws1 = sgpr_embed_workspace_bytes(h, G, N, K), ws2 = sgpr_score_all_pairs_workspace_bytes(h, G, G);
    if (ws1) CHECK_HIP(hipMalloc(&d_ws1, ws1));
    if (ws2) CHECK_HIP(hipMalloc(&d_ws2, ws2));

    /* SG.dgcnn_conv_pass + attention per graph (sg_net.py:79-127), then NTN + head for every ordered pair */
    CHECK_SGPR(sgpr_embed(h, d_centers, d_labels, G, N, K, d_pooled, NULL, NULL, d_ws1, ws1, NULL));
    CHECK_SGPR(sgpr_score_all_pairs(h, d_pooled, G, d_pooled, G, d_score, G, d_ws2, ws2, NULL));
    CHECK_SGPR(sgpr_check_status(h, NULL));
    CHECK_HIP(hipDeviceSynchronize());

    float* score = (float*)malloc((size_t)G * G * 4);
    CHECK_HIP(hipMemcpy(score, d_score, (size_t)G * G * 4, hipMemcpyDeviceToHost));

    /* the reference's own loop shape (eval_batch.py:30-36): a pair LIST, grouped by row graph once on the host
     * (sgpr_pair_plan) and scored by sgpr_score_pair_list - every listed score must be the dense matrix's entry, bit for bit */
    const int64_t P = (int64_t)G * 5;
    int32_t* i1 = (int32_t*)malloc((size_t)P * 4);
    int32_t* i2 = (int32_t*)malloc((size_t)P * 4);
    uint32_t lcg = 12345u;
    for (int64_t p = 0; p < P; ++p) {
        lcg = lcg * 1664525u + 1013904223u;
        i1[p] = (int32_t)((lcg >> 8) % (uint32_t)G);
        lcg = lcg * 1664525u + 1013904223u;
        i2[p] = (int32_t)((lcg >> 8) % (uint32_t)G);
    }
    const size_t cap = sgpr_pair_plan_ints(P, G);
    int32_t* plan = (int32_t*)malloc(cap * 4 + 4);
    size_t used = 0;
    int32_t n_rows = 0, n_items = 0;
    CHECK_SGPR(sgpr_pair_plan(i1, i2, P, G, G, plan, cap, &used, &n_rows, &n_items));
    int32_t* d_plan;
    float* d_list;
    void* d_ws3 = NULL;
    const size_t ws3 = sgpr_score_pair_list_workspace_bytes(h, n_rows, G);
    CHECK_HIP(hipMalloc((void**)&d_plan, used * 4 + 4));
    CHECK_HIP(hipMalloc((void**)&d_list, (size_t)P * 4 + 4));
    if (ws3) CHECK_HIP(hipMalloc(&d_ws3, ws3));
    CHECK_HIP(hipMemcpy(d_plan, plan, used * 4, hipMemcpyHostToDevice));
    CHECK_SGPR(sgpr_score_pair_list(h, d_pooled, G, d_pooled, G, d_plan, n_rows, n_items, P, d_list, d_ws3, ws3, NULL));
    CHECK_HIP(hipDeviceSynchronize());
    float* listed = (float*)malloc((size_t)P * 4 + 4);
    CHECK_HIP(hipMemcpy(listed, d_list, (size_t)P * 4, hipMemcpyDeviceToHost));
    int64_t differ = 0;
    for (int64_t p = 0; p < P; ++p) differ += listed[p] != score[(size_t)i1[p] * G + i2[p]];
    printf("pair list: %lld pairs over %d row graphs in %d work items, %lld differ from the dense matrix\n", (long long)P, n_rows,
           n_items, (long long)differ);
    if (differ) return 4;
    FILE* out = fopen(argv[3], "wb");
    if (!out || fwrite(score, 4, (size_t)G * G, out) != (size_t)G * G) { fprintf(stderr, "cannot write %s\n", argv[3]); return 1; }
    fclose(out);
    printf("scored %d x %d graph pairs (node_num %d, K %d); score[0][0] = %.6f\n", G, G, N, K, score[0]);
    sgpr_destroy(h);
    return 0;
}
