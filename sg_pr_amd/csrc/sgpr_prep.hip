// Data-set properties of a packed batch, on the device: the slots the embed kernel will PROCESS per graph (trailing
// duplicate slots collapse, sgpr_embed.hip / DESIGN.md 2.5), the node_cap they justify and the largest-first launch
// order of sgpr_embed_ordered.  The reference has no counterpart (it pads every graph to node_num and pays for it,
// sg_net.py:258-272); this replaces the host pass of Engine.size_order (16 ms for a KITTI-00-sized set) by two small
// launches, so that a data set that is already on the device never comes back to the host for it.
#include "sgpr_internal.hpp"

namespace sgpr {

constexpr int kBins = SGPR_MAX_NODES + 2;        // processed slots 0 .. SGPR_MAX_NODES (+1: a broken ragged graph)

// one wave per graph: slots[g] = nd + (m >= k && m > 1 ? 1 : m), nd = slots before the trailing run of slots identical
// to the last one (the embed kernel's own rule, embed_graph's prologue: same centre bits by ==, same effective label -
// a label outside [-1, num_labels) counts as padding there, and is reported by the embed launch, not here)
__global__ __launch_bounds__(256) void slots_kernel(const float* __restrict__ centers, const int32_t* __restrict__ labels,
                                                    const long long* __restrict__ rag_off, int G, int N, int k,
                                                    int num_labels, int32_t* __restrict__ slots) {
    const int lane = threadIdx.x & 63;
    const int g = (int)((blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6);
    if (g >= G) return;                                // (wave-uniform)
    int nd = 0, m;
    if (rag_off) {
        // ragged store: the padding exists only in the kernel's registers - count of real nodes, as Engine.ragged_order
        const long long cnt = rag_off[g + 1] - rag_off[g];
        nd = cnt < 0 ? N : (cnt > N ? N : (int)cnt);
        m = N - nd;
    } else {
        const float* c = centers + (size_t)g * N * 3;
        const int32_t* l = labels + (size_t)g * N;
        auto eff = [num_labels](int v) { return (v < -1 || v >= num_labels) ? -1 : v; };
        const float rx = c[(N - 1) * 3], ry = c[(N - 1) * 3 + 1], rz = c[(N - 1) * 3 + 2];
        const int rl = eff(l[N - 1]);
        for (int base = ((N - 1) >> 6) << 6; base >= 0; base -= 64) {
            const int s = base + lane;
            bool differs = false;
            if (s < N) differs = !(c[s * 3] == rx && c[s * 3 + 1] == ry && c[s * 3 + 2] == rz && eff(l[s]) == rl);
            const unsigned long long d = __ballot(differs);
            if (d) {
                nd = base + 64 - __clzll((long long)d);
                break;
            }
        }
        m = N - nd;
    }
    if (lane == 0) slots[g] = nd + ((m >= k && m > 1) ? 1 : m);
}

// one workgroup: counting sort of the graphs by processed slots, largest first, STABLE (equal graphs keep their index
// order, as torch.argsort(descending=True, stable=True) gives) - the launch order never changes a result, a stable one
// also never changes a timing from run to run.  info[0] = node_cap (largest count), info[1] = graphs beyond 64 slots.
__global__ __launch_bounds__(1024) void order_kernel(const int32_t* __restrict__ slots, int G, int32_t* __restrict__ order,
                                                    int32_t* __restrict__ info) {
    constexpr int NW = 16;
    __shared__ int start[kBins];                    // first output position of a bin
    __shared__ int cnt[NW][kBins];                  // per wave: graphs of its index range per bin, then its cursor
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int e = tid; e < NW * kBins; e += 1024) (&cnt[0][0])[e] = 0;
    __syncthreads();
    const int per = (((G + NW - 1) / NW) + 63) & ~63;             // graphs per wave, a multiple of 64
    const int g0 = wave * per, g1 = min(G, g0 + per);
    for (int g = g0 + lane; g < g1; g += 64) atomicAdd(&cnt[wave][min(max(slots[g], 0), kBins - 1)], 1);
    __syncthreads();
    if (tid < kBins) {
        int tot = 0;
        for (int w = 0; w < NW; ++w) tot += cnt[w][tid];
        start[tid] = tot;                           // (the bin's size for now)
    }
    __syncthreads();
    if (tid == 0) {
        int run = 0, cap = 0, over = 0;
        for (int v = kBins - 1; v >= 0; --v) {
            const int c = start[v];
            if (c > 0 && cap == 0) cap = v;
            if (v > 64) over += c;
            start[v] = run;
            run += c;
        }
        info[0] = cap;
        info[1] = over;
    }
    __syncthreads();
    if (tid < kBins) {
        int run = start[tid];
        for (int w = 0; w < NW; ++w) {
            const int c = cnt[w][tid];
            cnt[w][tid] = run;
            run += c;
        }
    }
    __syncthreads();
    volatile int* cur = cnt[wave];
    for (int gb = g0; gb < g1; gb += 64) {
        const int g = gb + lane;
        const bool valid = g < g1;
        const int v = valid ? min(max(slots[g], 0), kBins - 1) : -1;
        unsigned long long todo = __ballot(valid);
        while (todo) {                              // one distinct count of this group of 64 per step (wave-uniform)
            const int src = __ffsll((long long)todo) - 1;
            const int v0 = __shfl(v, src);
            const unsigned long long mk = __ballot(v == v0);
            const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(mk >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mk, 0));
            const int at = cur[v0];
            if (v == v0) order[at + rank] = g;
            __builtin_amdgcn_wave_barrier();
            if (lane == src) cur[v0] = at + __popcll(mk);
            __builtin_amdgcn_wave_barrier();
            todo &= ~mk;
        }
    }
}

size_t size_order_ws_bytes(int G) { return ((size_t)(G > 0 ? G : 1) * sizeof(int32_t) + 255) & ~(size_t)255; }

int launch_size_order(const float* centers, const int32_t* labels, const long long* rag_off, int G, int N, int k,
                      int num_labels, int32_t* order, int32_t* info, void* ws, hipStream_t stream) {
    int32_t* slots = static_cast<int32_t*>(ws);
    if (G > 0) {
        const int waves_per_block = 4;
        hipLaunchKernelGGL(slots_kernel, dim3((G + waves_per_block - 1) / waves_per_block), dim3(64 * waves_per_block), 0,
                           stream, centers, labels, rag_off, G, N, k, num_labels, slots);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return hip_fail(e, "slots_kernel launch");
    }
    hipLaunchKernelGGL(order_kernel, dim3(1), dim3(1024), 0, stream, slots, G, order, info);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "order_kernel launch");
    return SGPR_OK;
}

}  // namespace sgpr
