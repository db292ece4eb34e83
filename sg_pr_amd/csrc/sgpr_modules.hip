// Stand-alone forms of the building blocks the reference exposes as callable symbols ("signatures to
// keep"): dgcnn.knn / dgcnn.get_graph_feature (dgcnn.py:14-49) and AttentionModule.forward (layers_batch.py:28-39).
// Inside SG.forward these run fused in embed_kernel and never materialise their outputs; these kernels serve callers
// that use the pieces on their own tensors.  They take weights as explicit device pointers (a stand-alone module owns
// its own parameter), so they need no engine handle.
#include <math.h>

#include "sgpr_internal.hpp"

namespace sgpr {

constexpr int KNN_THREADS = 256;

// dgcnn.knn: x [B][C][N] -> idx [B][N][k] (int64 like torch.topk's indices).
//   pd[i][j] = -xx[j] - inner[i][j] - xx[i],  inner = -2 <x_i, x_j>        (dgcnn.py:15-17, same operation order)
//   idx[i][:] = the k largest pd[i][j], best first; equal values keep the lower candidate index first (torch.topk's
//   tie order is implementation-defined - CPU and CUDA already disagree; this is the engine's deterministic rule).
// One workgroup per graph: x is staged in LDS, thread i owns row i (N <= 256) and keeps its running top-k list in LDS
// ([slot][thread] layout: conflict-free), inserting a candidate only when it beats the current k-th.
__global__ __launch_bounds__(KNN_THREADS) void knn_kernel(const float* __restrict__ x, int C, int N, int k,
                                                          long long* __restrict__ idx) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* xs = reinterpret_cast<float*>(smem);                 // [C][N]
    float* xx = xs + (size_t)C * N;                             // [N]
    float* lv = xx + N;                                         // [k][KNN_THREADS] values, descending
    int* li = reinterpret_cast<int*>(lv + (size_t)k * KNN_THREADS);   // [k][KNN_THREADS]
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* xb = x + (size_t)b * C * N;
    for (int e = tid; e < C * N; e += KNN_THREADS) xs[e] = xb[e];
    __syncthreads();
    if (tid < N) {
        float s = 0.f;
        for (int c = 0; c < C; ++c) s = fmaf(xs[c * N + tid], xs[c * N + tid], s);
        xx[tid] = s;
    }
    __syncthreads();
    if (tid >= N) return;
    const int i = tid;
    for (int s = 0; s < k; ++s) {
        lv[s * KNN_THREADS + tid] = -INFINITY;
        li[s * KNN_THREADS + tid] = N;                          // sentinel: loses every tie
    }
    const float xi = xx[i];
    int filled = 0;
    for (int j = 0; j < N; ++j) {
        float dot = 0.f;
        for (int c = 0; c < C; ++c) dot = fmaf(xs[c * N + i], xs[c * N + j], dot);
        const float inner = -2.f * dot;
        const float pd = (-xx[j] - inner) - xi;
        // candidates arrive in ascending j: a strict comparison keeps the lower index ahead among equal values
        if (filled < k || pd > lv[(k - 1) * KNN_THREADS + tid]) {
            int s = filled < k ? filled : k - 1;
            while (s > 0 && pd > lv[(s - 1) * KNN_THREADS + tid]) {
                lv[s * KNN_THREADS + tid] = lv[(s - 1) * KNN_THREADS + tid];
                li[s * KNN_THREADS + tid] = li[(s - 1) * KNN_THREADS + tid];
                --s;
            }
            lv[s * KNN_THREADS + tid] = pd;
            li[s * KNN_THREADS + tid] = j;
            if (filled < k) ++filled;
        }
    }
    long long* out = idx + ((size_t)b * N + i) * k;
    for (int s = 0; s < k; ++s) out[s] = li[s * KNN_THREADS + tid];
}

int launch_knn(const float* x, int B, int C, int N, int k, int64_t* idx, hipStream_t stream) {
    if (B == 0) return SGPR_OK;
    const size_t lds = ((size_t)C * N + N) * sizeof(float) + (size_t)k * KNN_THREADS * (sizeof(float) + sizeof(int));
    if (lds > 160 * 1024) {
        set_error("sgpr_knn: C * N too large for one workgroup's LDS (" + std::to_string(lds) + " bytes)");
        return SGPR_E_NODES;
    }
    static LdsLimitOnce once;
    if (int rc = raise_lds_limit(&once, reinterpret_cast<const void*>(&knn_kernel), 160 * 1024, "sgpr_knn")) return rc;
    hipLaunchKernelGGL(knn_kernel, dim3(B), dim3(KNN_THREADS), lds, stream, x, C, N, k,
                       reinterpret_cast<long long*>(idx));
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "knn_kernel launch");
    return SGPR_OK;
}

// dgcnn.get_graph_feature (dgcnn.py:23-49) for given neighbour lists:
//   out [B][2C][N][k]:  out[b][c][n][m] = x[b][c][idx[b][n][m]] - x[b][c][n]   (c < C),   out[b][C + c][n][m] = x[b][c][n]
// HBM-bound gather: one thread per output element, m fastest (the k outputs of a node are contiguous).
__global__ __launch_bounds__(256) void graph_feature_kernel(const float* __restrict__ x,
                                                            const long long* __restrict__ idx, int C, int N, int k,
                                                            long long total, float* __restrict__ out) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const int m = (int)(e % k);
        long long r = e / k;
        const int n = (int)(r % N);
        r /= N;
        const int c2 = (int)(r % (2 * C));
        const long long b = r / (2 * C);
        const int c = c2 < C ? c2 : c2 - C;
        const float* xr = x + ((size_t)b * C + c) * N;
        const float xi = xr[n];
        float v = xi;
        if (c2 < C) {
            long long j = idx[((size_t)b * N + n) * k + m];
            j = j < 0 ? 0 : (j >= N ? N - 1 : j);
            v = xr[j] - xi;
        }
        out[e] = v;
    }
}

int launch_graph_feature(const float* x, const int64_t* idx, int B, int C, int N, int k, float* out,
                         hipStream_t stream) {
    const long long total = (long long)B * 2 * C * N * k;
    if (total == 0) return SGPR_OK;
    long long blocks = (total + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(graph_feature_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, x,
                       reinterpret_cast<const long long*>(idx), C, N, k, total, out);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "graph_feature_kernel launch");
    return SGPR_OK;
}

// AttentionModule.forward (layers_batch.py:28-39) on given node embeddings E [B][N][32]:
//   ctx = tanh(mean_n(E W)),  att[n] = sigmoid(E[n] . ctx),  rep = E^T att          (no padding mask, divisor N)
// One workgroup per graph; the same summation scheme as the fused tail of embed_kernel (8 fixed partial sums per
// channel, mean taken before the product with W).
__global__ __launch_bounds__(256) void attention_pool_kernel(const float* __restrict__ w, const float* __restrict__ emb,
                                                             int N, float* __restrict__ rep, float* __restrict__ att) {
    constexpr int F = kF3, NPART = 8;
    __shared__ float red[NPART * F];
    __shared__ float mean[F];
    __shared__ float tg[F];
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* sig = reinterpret_cast<float*>(smem);            // [N]
    const int b = blockIdx.x, tid = threadIdx.x, c = tid & 31, prt = tid >> 5;
    const float* E = emb + (size_t)b * N * F;
    {
        float s = 0.f;
        for (int n = prt; n < N; n += NPART) s += E[(size_t)n * F + c];
        red[prt * F + c] = s;
    }
    __syncthreads();
    if (tid < F) {
        float s = 0.f;
        for (int q = 0; q < NPART; ++q) s += red[q * F + tid];
        mean[tid] = s / (float)N;
    }
    __syncthreads();
    if (tid < F) {
        float gc = 0.f;
        for (int r = 0; r < F; ++r) gc = fmaf(mean[r], w[r * F + tid], gc);
        tg[tid] = tanhf(gc);
    }
    __syncthreads();
    for (int n = tid; n < N; n += 256) {
        float d = 0.f;
        for (int q = 0; q < F; ++q) d = fmaf(E[(size_t)n * F + q], tg[q], d);
        const float sg = 1.f / (1.f + expf(-d));
        sig[n] = sg;
        if (att) att[(size_t)b * N + n] = sg;
    }
    __syncthreads();
    {
        float s = 0.f;
        for (int n = prt; n < N; n += NPART) s = fmaf(sig[n], E[(size_t)n * F + c], s);
        red[prt * F + c] = s;
    }
    __syncthreads();
    if (tid < F) {
        float s = 0.f;
        for (int q = 0; q < NPART; ++q) s += red[q * F + tid];
        rep[(size_t)b * F + tid] = s;
    }
}

int launch_attention_pool(const float* w, const float* emb, int B, int N, float* rep, float* att, hipStream_t stream) {
    if (B == 0) return SGPR_OK;
    hipLaunchKernelGGL(attention_pool_kernel, dim3(B), dim3(256), (size_t)N * sizeof(float), stream, w, emb, N, rep, att);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "attention_pool_kernel launch");
    return SGPR_OK;
}

}  // namespace sgpr
