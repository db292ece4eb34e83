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
    constexpr int NW = 16, NC = 8;                  // NC: groups of 64 graphs whose counts a wave keeps in registers between its passes
    __shared__ int s_cap, s_over;
    __shared__ int wsum[8];
    __shared__ int cnt[NW][kBins];                  // per wave: graphs of its index range per bin, then its cursor
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int e = tid; e < NW * kBins; e += 1024) (&cnt[0][0])[e] = 0;
    const int per = (((G + NW - 1) / NW) + 63) & ~63;             // graphs per wave, a multiple of 64
    const int g0 = wave * per, g1 = min(G, g0 + per);
    int vc[NC];                                     // this lane's graphs of the wave's first NC groups (kBins: none)
#pragma unroll
    for (int q = 0; q < NC; ++q) {
        const int g = g0 + 64 * q + lane;
        vc[q] = g < g1 ? min(max(slots[g], 0), kBins - 1) : kBins;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NC; ++q)
        if (vc[q] < kBins) atomicAdd(&cnt[wave][vc[q]], 1);
    for (int g = g0 + 64 * NC + lane; g < g1; g += 64) atomicAdd(&cnt[wave][min(max(slots[g], 0), kBins - 1)], 1);
    __syncthreads();
    // bins in DESCENDING order of the count (largest graphs first): thread t owns bin kBins - 1 - t; an exclusive scan over t
    int mine = 0;
    if (tid < kBins) {
        for (int w = 0; w < NW; ++w) mine += cnt[w][kBins - 1 - tid];
    }
    if (tid == 0) s_cap = s_over = 0;
    int incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(incl, d);
        incl += lane >= d ? t : 0;
    }
    if (lane == 63 && wave < 8) wsum[wave] = incl;
    __syncthreads();
    if (tid < kBins) {
        int before = incl - mine;
        for (int w = 0; w < wave; ++w) before += wsum[w];
        const int bin = kBins - 1 - tid;
        int run = before;                           // ... then the waves' cursors inside the bin, in index order (stable)
        for (int w = 0; w < NW; ++w) {
            const int c = cnt[w][bin];
            cnt[w][bin] = run;
            run += c;
        }
    }
    if (tid < kBins && mine > 0) {                  // (a one-thread walk over the 258 bins was 11 us of this launch)
        atomicMax(&s_cap, kBins - 1 - tid);
        if (kBins - 1 - tid > 64) atomicAdd(&s_over, mine);
    }
    __syncthreads();
    if (tid == 0) {
        info[0] = s_cap;
        info[1] = s_over;
    }
    // the scatter, 64 graphs of a wave's range at a time: a lane's rank among the lanes of its group with the SAME count comes
    // from nine ballots (one per bit of the count: the lanes that agree with mine in every bit), no loop over the distinct
    // counts of the group; the first lane of each set of equal lanes advances the wave's cursor of that bin
    volatile int* cur = cnt[wave];
    auto place = [&](const int g, const int v) {
        const bool valid = v < kBins;
        unsigned long long same = ~0ull;
#pragma unroll
        for (int b = 0; b < 9; ++b) {
            const unsigned long long has = __ballot((v >> b) & 1);
            same &= ((v >> b) & 1) ? has : ~has;
        }
        const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(same >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)same, 0));
        const int at = valid ? cur[v] : 0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();             // every lane has read its cursor before a leader moves it
        if (valid) {
            order[at + rank] = g;
            if (rank == 0) cur[v] = at + __popcll(same);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
#pragma unroll
    for (int q = 0; q < NC; ++q)
        if (g0 + 64 * q < g1) place(g0 + 64 * q + lane, vc[q]);              // (wave-uniform)
    for (int gb = g0 + 64 * NC; gb < g1; gb += 64) {
        const int g = gb + lane;
        place(g, g < g1 ? min(max(slots[g], 0), kBins - 1) : kBins);
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
