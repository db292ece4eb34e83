// Timing harness for variants of score_all_pairs_kernel (sg_pr_amd/csrc/sgpr_score.hip is compiled INTO this program):
// random pooled vectors and weights, every variant timed over 20 launches and compared with the first one.
#include "../../sg_pr_amd/csrc/sgpr_score.hip"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
namespace sgpr {
void set_error(const std::string& m) { fprintf(stderr, "%s\n", m.c_str()); }
int hip_fail(hipError_t e, const char* what) { fprintf(stderr, "%s: %s\n", what, hipGetErrorString(e)); return -6; }
}
using namespace sgpr;
static float* dev(const std::vector<float>& v) { float* d; hipMalloc(&d, v.size() * 4); hipMemcpy(d, v.data(), v.size() * 4, hipMemcpyHostToDevice); return d; }
static std::vector<float> rnd(size_t n, float s) { std::vector<float> v(n); for (auto& x : v) x = s * (2.f * rand() / RAND_MAX - 1.f); return v; }

template <int OCC, int NI, int VAR>
static void run(const char* name, const DevWeights& w, int R, int M, unsigned short* Ab, unsigned short* Cb, float* ur, float* rng, int nrng,
                float* rows, float* cols, float* score, std::vector<float>& ref, int cus) {
    const int64_t items = (int64_t)((M + AP_COLS - 1) / AP_COLS) * ((R + AP_ROWS - 1) / AP_ROWS);
    const int64_t slots = (int64_t)cus * OCC;
    const unsigned grid = (unsigned)(items < slots ? items : slots);
    hipMemset(score, 0, (size_t)R * M * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((score_all_pairs_kernel<OCC, NI, VAR>), dim3(grid), dim3(256), 0, 0, w, R, M, Ab, Cb, ur, rng, nrng, rows, cols, score, (int64_t)M);
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((score_all_pairs_kernel<OCC, NI, VAR>), dim3(grid), dim3(256), 0, 0, w, R, M, Ab, Cb, ur, rng, nrng, rows, cols, score, (int64_t)M);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<float> out((size_t)R * M);
    hipMemcpy(out.data(), score, out.size() * 4, hipMemcpyDeviceToHost);
    double md = 0, mean = 0;
    if (ref.empty()) ref = out;
    for (size_t i = 0; i < out.size(); ++i) { md = fmax(md, fabs((double)out[i] - ref[i])); mean += out[i]; }
    printf("%-28s occ %d ni %d var %d : %8.1f us   max|d vs first| %.3g  mean %.4f\n", name, OCC, NI, VAR, ms / 20 * 1e3, md, mean / out.size());
}

int main(int argc, char** argv) {
    const int R = argc > 1 ? atoi(argv[1]) : 4541, M = argc > 2 ? atoi(argv[2]) : 4541;
    srand(1);
    DevWeights w; memset(&w, 0, sizeof(w));
    auto ntn = rnd(32 * 32 * 16, 0.12f);
    std::vector<float> ntnt(ntn.size());
    for (int i = 0; i < 32; ++i) for (int t = 0; t < 16; ++t) for (int j = 0; j < 32; ++j) ntnt[(i * 16 + t) * 32 + j] = ntn[i * 512 + j * 16 + t];
    w.ntn_w = dev(ntn); w.ntn_wt = dev(ntnt); w.ntn_wb = dev(rnd(16 * 64, 0.25f)); w.ntn_bias = dev(rnd(16, 0.3f));
    w.fc1_w = dev(rnd(256, 0.4f)); w.fc1_b = dev(rnd(16, 0.3f)); w.fc2_w = dev(rnd(16, 0.5f)); w.fc2_b = dev(rnd(1, 0.1f));
    float* rows = dev(rnd((size_t)R * 32, 3.f)); float* cols = dev(rnd((size_t)M * 32, 3.f));
    void* ws; hipMalloc(&ws, score_all_pairs_ws_bytes(R, M));
    float* score; hipMalloc(&score, (size_t)R * M * 4);
    const int ngroups = ap_prep_groups(R, M), nrng = 2 * ngroups;
    float* ur = static_cast<float*>(ws); float* rng = ur + (size_t)R * T;
    unsigned short* Ab = reinterpret_cast<unsigned short*>(rng + (size_t)nrng * 4); unsigned short* Cb = Ab + (size_t)R * 2 * 64 * 8;
    hipLaunchKernelGGL(ntn_prep_kernel, dim3(nrng), dim3(256), 0, 0, w, rows, R, cols, M, Ab, ur, rng, Cb);
    hipDeviceSynchronize();
    int cus = 256; hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    // warm the clocks (a fresh process measures the ramp), then the pair-list kernel as the correctness reference
    for (int i = 0; i < 3000; ++i) hipLaunchKernelGGL((score_all_pairs_kernel<4, 1, 0>), dim3(1024), dim3(256), 0, 0, w, R, M, Ab, Cb, ur, rng, nrng, rows, cols, score, (int64_t)M);
    hipDeviceSynchronize();
    std::vector<float> ref((size_t)R * M);
    {
        std::vector<int> i1((size_t)R * M), i2((size_t)R * M);
        for (int r = 0; r < R; ++r) for (int c = 0; c < M; ++c) { i1[(size_t)r * M + c] = r; i2[(size_t)r * M + c] = c; }
        int *d1, *d2; hipMalloc(&d1, i1.size() * 4); hipMalloc(&d2, i2.size() * 4);
        hipMemcpy(d1, i1.data(), i1.size() * 4, hipMemcpyHostToDevice); hipMemcpy(d2, i2.data(), i2.size() * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(score_pairs_kernel, dim3((unsigned)(((size_t)R * M + 3) / 4)), dim3(256), 0, 0, w, rows, d1, cols, d2, (int64_t)R * M, score);
        hipMemcpy(ref.data(), score, ref.size() * 4, hipMemcpyDeviceToHost);
    }
#define RUN(O, N, V) run<O, N, V>(#O "," #N "," #V, w, R, M, Ab, Cb, ur, rng, nrng, rows, cols, score, ref, cus)
    RUN(4, 1, 0); RUN(3, 1, 0); RUN(2, 1, 0); RUN(1, 1, 0); RUN(4, 1, 16); RUN(2, 1, 16); RUN(1, 1, 16); RUN(4, 1, 32); RUN(2, 1, 32); RUN(1, 1, 32);
    return 0;
}
