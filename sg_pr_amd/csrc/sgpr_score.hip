// Pair-coupled tail of SG.forward on gfx950: Neural Tensor Network
// (TenorNetworkModule.forward, reference layers_batch.py:70-83) + fully_connected_first
// / ReLU + scoring_layer / sigmoid (sg_net.py:131-136).
//
//  * score_pairs_kernel      one wave64 per pair (pair-list mode: eval_batch.py:30-36,
//                            SG.forward's per-pair tail).
//  * ntn_prep_kernel + score_all_pairs_kernel
//                            dense R x M rectangle.  The bilinear form is hoisted per row
//                            graph (A_r = e1^T W, 16x32) so a pair costs 512+256+16 FMA;
//                            one thread owns one column graph (its pooled vector stays in
//                            registers), rows are wave-uniform so A_r / FC weights arrive
//                            through the scalar cache; the score row is written coalesced.
#include <math.h>

#include "sgpr_internal.hpp"

namespace sgpr {

constexpr int F = kF3;   // 32 pooled features
constexpr int T = kT;    // 16 tensor neurons
constexpr int BN_ = kB;  // 16 bottleneck neurons

// ------------------------------------------------------------------ per-pair list
__global__ __launch_bounds__(256) void score_pairs_kernel(const DevWeights w, const float* __restrict__ p1,
                                                          const int32_t* __restrict__ i1,
                                                          const float* __restrict__ p2,
                                                          const int32_t* __restrict__ i2, int64_t P,
                                                          float* __restrict__ score) {
    const int lane = threadIdx.x & 63;
    const int64_t pair = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pair >= P) return;
    const int64_t r1 = i1 ? i1[pair] : pair;
    const int64_t r2 = i2 ? i2[pair] : pair;
    const float* e1 = p1 + r1 * F;
    const float* e2 = p2 + r2 * F;
    const int t = lane & 15, q = lane >> 4;

    // v[r] = sum_i e1[i] * W[i][col_r],  col_r = lane + 64 r  ->  j = q + 4r, same t for every r
    float v[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) v[r] = 0.f;
    for (int i = 0; i < F; ++i) {
        const float a = e1[i];
        const float* wr = w.ntn_w + i * (F * T) + lane;
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = fmaf(a, wr[64 * r], v[r]);
    }
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r) s = fmaf(v[r], e2[q + 4 * r], s);
    // block term  Wb[t][:] . [e1; e2], 16 of the 64 products per lane group
    for (int m = 0; m < 16; ++m) {
        const int mm = q * 16 + m;
        const float x = mm < F ? e1[mm] : e2[mm - F];
        s = fmaf(w.ntn_wb[t * 2 * F + mm], x, s);
    }
    s += __shfl_xor(s, 16);
    s += __shfl_xor(s, 32);
    const float h = fmaxf(s + w.ntn_bias[t], 0.f);   // lanes with equal t agree
    // fully_connected_first + ReLU: lane t computes output neuron t
    float gacc = w.fc1_b[t];
    for (int tt = 0; tt < T; ++tt) gacc = fmaf(w.fc1_w[t * T + tt], __shfl(h, tt), gacc);
    float z = fmaxf(gacc, 0.f) * w.fc2_w[t];
    z += __shfl_xor(z, 1);
    z += __shfl_xor(z, 2);
    z += __shfl_xor(z, 4);
    z += __shfl_xor(z, 8);
    if (lane == 0) score[pair] = 1.f / (1.f + expf(-(z + w.fc2_b[0])));
}

int launch_score_pairs(const sgpr_handle* h, const float* p1, const int32_t* i1, const float* p2, const int32_t* i2,
                       int64_t P, float* score, hipStream_t stream) {
    if (P == 0) return SGPR_OK;
    const int64_t blocks = (P + 3) / 4;
    if (blocks > 0x7fffffffLL) {
        set_error("sgpr_score_pairs: too many pairs for one launch");
        return SGPR_E_INVALID;
    }
    hipLaunchKernelGGL(score_pairs_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, h->w, p1, i1, p2, i2, P, score);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "score_pairs_kernel launch");
    return SGPR_OK;
}

// ------------------------------------------------------------------ dense all-pairs
// workspace layout (floats):  Ar [R][T][F] | ur [R][T] | vc [M][T]
size_t score_all_pairs_ws_bytes(int R, int M) {
    return ((size_t)R * (T * F + T) + (size_t)M * T) * sizeof(float);
}

// one wave per graph: rows get A_r and u_r = Wb[:, :F] e1 + bias, columns get v_c = Wb[:, F:] e2
__global__ __launch_bounds__(256) void ntn_prep_kernel(const DevWeights w, const float* __restrict__ rows, int R,
                                                       const float* __restrict__ cols, int M,
                                                       float* __restrict__ Ar, float* __restrict__ ur,
                                                       float* __restrict__ vc) {
    const int lane = threadIdx.x & 63;
    const int gidx = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (gidx >= R + M) return;
    const int t = lane & 15, q = lane >> 4;
    if (gidx < R) {
        const float* e1 = rows + (size_t)gidx * F;
        float v[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = 0.f;
        for (int i = 0; i < F; ++i) {
            const float a = e1[i];
            const float* wr = w.ntn_w + i * (F * T) + lane;
#pragma unroll
            for (int r = 0; r < 8; ++r) v[r] = fmaf(a, wr[64 * r], v[r]);
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) Ar[(size_t)gidx * (T * F) + t * F + (q + 4 * r)] = v[r];
        float s = 0.f;
        for (int m = 0; m < 8; ++m) s = fmaf(w.ntn_wb[t * 2 * F + q * 8 + m], e1[q * 8 + m], s);
        s += __shfl_xor(s, 16);
        s += __shfl_xor(s, 32);
        if (q == 0) ur[(size_t)gidx * T + t] = s + w.ntn_bias[t];
    } else {
        const int c = gidx - R;
        const float* e2 = cols + (size_t)c * F;
        float s = 0.f;
        for (int m = 0; m < 8; ++m) s = fmaf(w.ntn_wb[t * 2 * F + F + q * 8 + m], e2[q * 8 + m], s);
        s += __shfl_xor(s, 16);
        s += __shfl_xor(s, 32);
        if (q == 0) vc[(size_t)c * T + t] = s;
    }
}

constexpr int AP_ROWS = 16;  // row graphs per workgroup (column vectors are reused from registers)

__global__ __launch_bounds__(256) void score_all_pairs_kernel(const DevWeights w, const float* __restrict__ cols,
                                                              int R, int M, const float* __restrict__ Ar,
                                                              const float* __restrict__ ur,
                                                              const float* __restrict__ vc,
                                                              float* __restrict__ score, int64_t ld) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    const bool live = c < M;
    const int cc = live ? c : M - 1;
    float e2[F], v[T];
    {
        const float4* src = reinterpret_cast<const float4*>(cols + (size_t)cc * F);
#pragma unroll
        for (int j = 0; j < F / 4; ++j) {
            const float4 x = src[j];
            e2[4 * j] = x.x; e2[4 * j + 1] = x.y; e2[4 * j + 2] = x.z; e2[4 * j + 3] = x.w;
        }
        const float4* vs = reinterpret_cast<const float4*>(vc + (size_t)cc * T);
#pragma unroll
        for (int j = 0; j < T / 4; ++j) {
            const float4 x = vs[j];
            v[4 * j] = x.x; v[4 * j + 1] = x.y; v[4 * j + 2] = x.z; v[4 * j + 3] = x.w;
        }
    }
    const int r0 = blockIdx.y * AP_ROWS;
    const int r1 = min(R, r0 + AP_ROWS);
    for (int r = r0; r < r1; ++r) {                 // r is wave-uniform: A_r, u_r, FC weights -> scalar loads
        const float* __restrict__ a = Ar + (size_t)r * (T * F);
        const float* __restrict__ u = ur + (size_t)r * T;
        float h[T];
#pragma unroll
        for (int t = 0; t < T; ++t) {
            float s = u[t] + v[t];
#pragma unroll
            for (int j = 0; j < F; ++j) s = fmaf(a[t * F + j], e2[j], s);
            h[t] = fmaxf(s, 0.f);
        }
        float z = w.fc2_b[0];
#pragma unroll
        for (int o = 0; o < BN_; ++o) {
            float gacc = w.fc1_b[o];
#pragma unroll
            for (int t = 0; t < T; ++t) gacc = fmaf(w.fc1_w[o * T + t], h[t], gacc);
            z = fmaf(w.fc2_w[o], fmaxf(gacc, 0.f), z);
        }
        if (live) score[(size_t)r * ld + c] = 1.f / (1.f + expf(-z));
    }
}

int launch_score_all_pairs(const sgpr_handle* h, const float* rows, int R, const float* cols, int M, float* score,
                           int64_t ld, void* ws, hipStream_t stream) {
    if (R == 0 || M == 0) return SGPR_OK;
    float* Ar = static_cast<float*>(ws);
    float* ur = Ar + (size_t)R * T * F;
    float* vc = ur + (size_t)R * T;
    hipLaunchKernelGGL(ntn_prep_kernel, dim3((R + M + 3) / 4), dim3(256), 0, stream, h->w, rows, R, cols, M, Ar, ur, vc);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "ntn_prep_kernel launch");
    dim3 grid((M + 255) / 256, (R + AP_ROWS - 1) / AP_ROWS);
    hipLaunchKernelGGL(score_all_pairs_kernel, grid, dim3(256), 0, stream, h->w, cols, R, M, Ar, ur, vc, score, ld);
    e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "score_all_pairs_kernel launch");
    return SGPR_OK;
}

}  // namespace sgpr
