// Upstream semantic-graph generation on gfx950: one labelled LiDAR scan -> graph nodes.
// Replaces, for one scan, `gen_labels` (data_process/gen_label_graph.py:196-326: label remapping, per-class instance
// grouping or PCL EuclideanClusterExtraction with class-dependent tolerance / minimum size) and the node half of
// `gen_graphs` (:336-365: one node per cluster whose class is in node_map, centre = mean of the cluster's points).
//
// MI355X-first formulation - HBM-bound integer / pointer work, no matrix cores:
//   * Euclidean clustering = connected components of "squared distance < tolerance^2" (what PCL's FLANN radius search
//     accepts): points are hashed into cells of one tolerance (open-addressing table keyed by (class, ix, iy, iz),
//     a linked list per cell), every point tests the points of its 27 neighbouring cells with a lower index and unites
//     with those in range in a lock-free union-find that always hooks the larger root under the smaller, so a
//     component's root is its LOWEST POINT INDEX whatever the execution order.
//   * instance-labelled classes: all points of a (class, instance) pair point at the pair's lowest point index
//     (one atomicMin per point on a second hash table).
//   * sizes and centroids by atomics on the root: coordinates are accumulated as 2^-24 m fixed point in 64-bit integers
//     (exact for |x| >= 0.5 m, 3e-8 m otherwise), so sums - and therefore every output - are independent of the order
//     the atomics land in.
//   * clusters that pass the size rules and map to a node class are ranked in the reference's order (class ascending;
//     instance id ascending / size descending, lowest point index first among equal sizes) by counting, one thread each.
// The per-point cluster index is returned as well (the reference's intermediate "labels" output, :323-324).
#include <math.h>

#include "sgpr_internal.hpp"

namespace sgpr {

constexpr int CL_THREADS = 256;
constexpr int CL_MAX_CAND = 8192;          // clusters that may qualify as nodes in one scan
constexpr int CL_INST_SLOTS = 1 << 16;     // (class, instance) hash table
constexpr unsigned long long CL_EMPTY = ~0ull;
constexpr int CL_NCLASS = 20;
constexpr int CL_MAX_CLUSTER = 50000;      // gen_label_graph.py:305

// gen_label_graph.py:23-58: raw SemanticKITTI id -> training id (0 for every id not listed)
__device__ __forceinline__ int remap_label(unsigned raw) {
    switch (raw) {
        case 10: case 252: return 1;
        case 11: return 2;
        case 13: case 16: case 20: case 256: case 257: case 259: return 5;
        case 15: return 3;
        case 18: case 258: return 4;
        case 30: case 254: return 6;
        case 31: case 253: return 7;
        case 32: case 255: return 8;
        case 40: case 60: return 9;
        case 44: return 10;
        case 48: return 11;
        case 49: return 12;
        case 50: return 13;
        case 51: return 14;
        case 70: return 15;
        case 71: return 16;
        case 72: return 17;
        case 80: return 18;
        case 81: return 19;
        default: return 0;
    }
}
// per training class: 0 = discarded (:268), 1 = one cluster, never a node (road / parking, :259), 2 = clustered
__device__ __forceinline__ int class_mode(int c) {
    if (c == 9 || c == 10) return 1;
    if (c == 0 || c == 2 || c == 3 || c == 6 || c == 7 || c == 8) return 0;
    return 2;
}
__device__ __forceinline__ float class_tolerance(int c) {      // :283-288
    if (c == 1 || c == 4 || c == 5 || c == 14) return 0.5f;
    if (c == 11 || c == 12 || c == 13 || c == 15 || c == 17) return 2.0f;
    return 0.2f;
}
__device__ __forceinline__ int class_min_size(int c) {         // :290-297
    if (c == 16 || c == 19) return 50;
    if (c == 15) return 200;
    if (c == 11 || c == 12 || c == 13 || c == 17) return 300;
    return 100;
}
__device__ __forceinline__ int node_class(int c) {             // node_map, :64-77 (-1: not a node class)
    switch (c) {
        case 1: return 0;
        case 4: return 1;
        case 5: return 2;
        case 11: return 3;
        case 12: return 4;
        case 13: return 5;
        case 14: return 6;
        case 15: return 7;
        case 16: return 8;
        case 17: return 9;
        case 18: return 10;
        case 19: return 11;
        default: return -1;
    }
}

struct ClusterWs {
    int* cls;                      // [P] training class
    int* parent;                   // [P] union-find forest; after flatten: the root (lowest index) of the point's cluster
    int* next;                     // [P] linked list of the point's cell
    int* size;                     // [P] points per root
    long long* sum;                // [P][3] fixed-point coordinate sums per root
    int* node_of_root;             // [P] node index of a selected root, else -1
    unsigned long long* cell_key;  // [H] cell hash table
    int* cell_head;                // [H]
    unsigned long long* inst_key;  // [CL_INST_SLOTS]
    int* inst_min;                 // [CL_INST_SLOTS] lowest point index of the (class, instance) pair
    int* class_inst;               // [32] bit 0: the class carries instance labels in this scan
    int* count;                    // [1] qualifying clusters; [1] overflow flag
    int4* cand;                    // [CL_MAX_CAND] (class, order key, root, size)
    unsigned H;                    // cell table slots (power of two)
};

__device__ __forceinline__ unsigned long long mix64(unsigned long long k) {
    k ^= k >> 33;
    k *= 0xff51afd7ed558ccdull;
    k ^= k >> 33;
    k *= 0xc4ceb9fe1a85ec53ull;
    k ^= k >> 33;
    return k;
}
__device__ __forceinline__ unsigned long long cell_key_of(int c, int ix, int iy, int iz) {
    // 19 bits per axis (offset binary), 5 bits of class
    return ((unsigned long long)c << 57) | ((unsigned long long)((ix + (1 << 18)) & 0x7ffff) << 38) |
           ((unsigned long long)((iy + (1 << 18)) & 0x7ffff) << 19) | (unsigned long long)((iz + (1 << 18)) & 0x7ffff);
}
// slot of `key`, inserting it when absent
__device__ __forceinline__ unsigned table_insert(unsigned long long* keys, unsigned mask, unsigned long long key) {
    unsigned s = (unsigned)mix64(key) & mask;
    while (true) {
        const unsigned long long old = atomicCAS(&keys[s], CL_EMPTY, key);
        if (old == CL_EMPTY || old == key) return s;
        s = (s + 1) & mask;
    }
}
// slot of `key`, or ~0u when absent (table complete: built by an earlier kernel)
__device__ __forceinline__ unsigned table_find(const unsigned long long* keys, unsigned mask, unsigned long long key) {
    unsigned s = (unsigned)mix64(key) & mask;
    while (true) {
        const unsigned long long k = keys[s];
        if (k == key) return s;
        if (k == CL_EMPTY) return ~0u;
        s = (s + 1) & mask;
    }
}

// agent-scope relaxed accesses: the forest is rewritten by other CUs while it is read (a CU's L1 is never refreshed by
// other CUs' stores), so its loads must be served by L2
__device__ __forceinline__ int ld_parent(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_parent(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ int uf_find(int* parent, int x) {
    int p = ld_parent(parent + x);
    while (p != x) {                       // path halving: only ever replaces a pointer by an ancestor
        const int gp = ld_parent(parent + p);
        if (gp != p) st_parent(parent + x, gp);
        x = p;
        p = gp;
    }
    return x;
}
__device__ __forceinline__ void uf_unite(int* parent, int a, int b) {
    while (true) {
        a = uf_find(parent, a);
        b = uf_find(parent, b);
        if (a == b) return;
        const int hi = a > b ? a : b, lo = a > b ? b : a;
        if (atomicCAS(&parent[hi], hi, lo) == hi) return;      // hook the larger root under the smaller
    }
}

// cells are a hair wider than the tolerance, so that two points in range can never end up two cells apart through the
// rounding of the division
__device__ __forceinline__ int cell_coord(float v, float tol) { return (int)floorf(v / (tol * 1.0001f)); }

__global__ __launch_bounds__(CL_THREADS) void cluster_init_kernel(ClusterWs w, const unsigned* __restrict__ label, int P) {
    const int p = blockIdx.x * CL_THREADS + threadIdx.x;
    if (p >= P) return;
    const unsigned l = label[p];
    const int c = remap_label(l & 0xffffu);
    w.cls[p] = c;
    w.parent[p] = p;
    w.size[p] = 0;
    w.sum[3 * (size_t)p + 0] = 0;
    w.sum[3 * (size_t)p + 1] = 0;
    w.sum[3 * (size_t)p + 2] = 0;
    if (class_mode(c) == 2 && (l >> 16) != 0) atomicOr(&w.class_inst[c], 1);   // :270: the class has instance labels
}

__global__ __launch_bounds__(CL_THREADS) void cluster_link_kernel(ClusterWs w, const float* __restrict__ pts, int stride,
                                                                  const unsigned* __restrict__ label, int P) {
    const int p = blockIdx.x * CL_THREADS + threadIdx.x;
    if (p >= P) return;
    const int c = w.cls[p];
    if (class_mode(c) != 2) return;
    if (w.class_inst[c]) {
        const unsigned s = table_insert(w.inst_key, CL_INST_SLOTS - 1, ((unsigned long long)c << 16) | (label[p] >> 16));
        atomicMin(&w.inst_min[s], p);
    } else {
        const float tol = class_tolerance(c);
        const float* q = pts + (size_t)p * stride;
        const unsigned s = table_insert(w.cell_key, w.H - 1, cell_key_of(c, cell_coord(q[0], tol), cell_coord(q[1], tol),
                                                                         cell_coord(q[2], tol)));
        w.next[p] = atomicExch(&w.cell_head[s], p);
    }
}

__global__ __launch_bounds__(CL_THREADS) void cluster_union_kernel(ClusterWs w, const float* __restrict__ pts, int stride,
                                                                   const unsigned* __restrict__ label, int P) {
    const int p = blockIdx.x * CL_THREADS + threadIdx.x;
    if (p >= P) return;
    const int c = w.cls[p];
    if (class_mode(c) != 2) return;
    if (w.class_inst[c]) {
        const unsigned s = table_find(w.inst_key, CL_INST_SLOTS - 1, ((unsigned long long)c << 16) | (label[p] >> 16));
        w.parent[p] = w.inst_min[s];       // no other thread touches this class's forest entries
        return;
    }
    const float tol = class_tolerance(c);
    const float tol2 = tol * tol;
    const float* q = pts + (size_t)p * stride;
    const float x = q[0], y = q[1], z = q[2];
    const int ix = cell_coord(x, tol), iy = cell_coord(y, tol), iz = cell_coord(z, tol);
    for (int dz = -1; dz <= 1; ++dz)
        for (int dy = -1; dy <= 1; ++dy)
            for (int dx = -1; dx <= 1; ++dx) {
                const unsigned s = table_find(w.cell_key, w.H - 1, cell_key_of(c, ix + dx, iy + dy, iz + dz));
                if (s == ~0u) continue;
                for (int o = w.cell_head[s]; o >= 0; o = w.next[o]) {
                    if (o >= p) continue;                       // every unordered pair once
                    const float* r = pts + (size_t)o * stride;
                    const float ex = x - r[0], ey = y - r[1], ez = z - r[2];
                    // (dx^2 + dy^2) + dz^2 with individually rounded operations (no contraction): the oracle's order
                    const float d2 = __fadd_rn(__fadd_rn(__fmul_rn(ex, ex), __fmul_rn(ey, ey)), __fmul_rn(ez, ez));
                    if (d2 < tol2) uf_unite(w.parent, p, o);    // FLANN's radius search: strictly below
                }
            }
}

__global__ __launch_bounds__(CL_THREADS) void cluster_flatten_kernel(ClusterWs w, const float* __restrict__ pts, int stride, int P) {
    const int p = blockIdx.x * CL_THREADS + threadIdx.x;
    if (p >= P) return;
    if (class_mode(w.cls[p]) != 2) return;
    int r = p;
    while (true) {                                              // the forest is final: plain reads through L2
        const int up = ld_parent(w.parent + r);
        if (up == r) break;
        r = up;
    }
    w.next[p] = r;                                              // (the cell lists are dead) the point's root
    atomicAdd(&w.size[r], 1);
    const float* q = pts + (size_t)p * stride;
    const double k = 16777216.0;                                // 2^24: centimetre-free fixed point, order-independent sums
    atomicAdd(reinterpret_cast<unsigned long long*>(&w.sum[3 * (size_t)r + 0]), (unsigned long long)llrint((double)q[0] * k));
    atomicAdd(reinterpret_cast<unsigned long long*>(&w.sum[3 * (size_t)r + 1]), (unsigned long long)llrint((double)q[1] * k));
    atomicAdd(reinterpret_cast<unsigned long long*>(&w.sum[3 * (size_t)r + 2]), (unsigned long long)llrint((double)q[2] * k));
}

__global__ __launch_bounds__(CL_THREADS) void cluster_collect_kernel(ClusterWs w, const unsigned* __restrict__ label, int P) {
    const int p = blockIdx.x * CL_THREADS + threadIdx.x;
    if (p >= P) return;
    const int c = w.cls[p];
    if (class_mode(c) != 2 || w.next[p] != p) return;           // roots only
    const int n = w.size[p];
    const bool inst = w.class_inst[c] != 0;
    const bool keep = inst ? n > 20 : (n >= class_min_size(c) && n <= CL_MAX_CLUSTER);     // :274, :303-305
    if (!keep || node_class(c) < 0) return;
    const int i = atomicAdd(&w.count[0], 1);
    if (i >= CL_MAX_CAND) {
        w.count[1] = 1;
        return;
    }
    // order key inside a class: instance id ascending (:272), or size descending with the lowest point index first
    w.cand[i] = make_int4(c, inst ? (int)(label[p] >> 16) : -n, p, n);
}

__global__ __launch_bounds__(CL_THREADS) void cluster_rank_kernel(ClusterWs w, int max_nodes, double* __restrict__ centers,
                                                                  int32_t* __restrict__ node_labels,
                                                                  int32_t* __restrict__ node_sizes, int32_t* __restrict__ num_nodes) {
    const int n = min(w.count[0], CL_MAX_CAND);
    const int i = blockIdx.x * CL_THREADS + threadIdx.x;
    if (i == 0) num_nodes[0] = w.count[1] ? -1 : n;             // -1: more clusters than CL_MAX_CAND
    if (i >= n) return;
    const int4 me = w.cand[i];
    int rank = 0;
    for (int j = 0; j < n; ++j) {
        const int4 o = w.cand[j];
        const bool before = o.x < me.x || (o.x == me.x && (o.y < me.y || (o.y == me.y && o.z < me.z)));
        rank += before ? 1 : 0;
    }
    if (rank >= max_nodes) return;
    const double inv = 1.0 / (16777216.0 * (double)me.w);
    centers[3 * rank + 0] = (double)w.sum[3 * (size_t)me.z + 0] * inv;
    centers[3 * rank + 1] = (double)w.sum[3 * (size_t)me.z + 1] * inv;
    centers[3 * rank + 2] = (double)w.sum[3 * (size_t)me.z + 2] * inv;
    node_labels[rank] = node_class(me.x);
    node_sizes[rank] = me.w;
    w.node_of_root[me.z] = rank;
}

__global__ __launch_bounds__(CL_THREADS) void cluster_assign_kernel(ClusterWs w, int P, int32_t* __restrict__ point_node) {
    const int p = blockIdx.x * CL_THREADS + threadIdx.x;
    if (p >= P) return;
    point_node[p] = class_mode(w.cls[p]) == 2 ? w.node_of_root[w.next[p]] : -1;
}

// gen_graphs' edge rule (gen_label_graph.py:367-385): for clusters i < j the distance between the point of i and the
// point of j that lie nearest to the midpoint of the two centres.  One workgroup per ORDERED pair (i, j) finds the
// point of cluster i nearest to mid(i, j) - first point in scan order among equals, like np.argmin - in float64.
__global__ __launch_bounds__(CL_THREADS) void graph_nearest_kernel(const float* __restrict__ pts, int stride,
                                                                   const int32_t* __restrict__ point_node, int P, int n,
                                                                   const double* __restrict__ centers, int* __restrict__ near) {
    __shared__ double bd[CL_THREADS];
    __shared__ int bi[CL_THREADS];
    const int i = blockIdx.x / n, j = blockIdx.x - i * n;
    if (i == j) return;
    const double mx = (centers[3 * i] + centers[3 * j]) * 0.5, my = (centers[3 * i + 1] + centers[3 * j + 1]) * 0.5,
                 mz = (centers[3 * i + 2] + centers[3 * j + 2]) * 0.5;
    double best = INFINITY;
    int bidx = 0x7fffffff;
    for (int p = threadIdx.x; p < P; p += CL_THREADS) {
        if (point_node[p] != i) continue;
        const float* q = pts + (size_t)p * stride;
        const double dx = mx - (double)q[0], dy = my - (double)q[1], dz = mz - (double)q[2];
        const double d = (dx * dx + dy * dy) + dz * dz;
        if (d < best) {                          // ascending p per thread: strict keeps the first among equals
            best = d;
            bidx = p;
        }
    }
    bd[threadIdx.x] = best;
    bi[threadIdx.x] = bidx;
    __syncthreads();
    for (int s = CL_THREADS / 2; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
            const double od = bd[threadIdx.x + s];
            const int oi = bi[threadIdx.x + s];
            if (od < bd[threadIdx.x] || (od == bd[threadIdx.x] && oi < bi[threadIdx.x])) {
                bd[threadIdx.x] = od;
                bi[threadIdx.x] = oi;
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) near[(size_t)i * n + j] = bi[0];
}

__global__ __launch_bounds__(CL_THREADS) void graph_min_dis_kernel(const float* __restrict__ pts, int stride, int n,
                                                                   const int* __restrict__ near, double* __restrict__ min_dis) {
    const int e = blockIdx.x * CL_THREADS + threadIdx.x;
    if (e >= n * n) return;
    const int i = e / n, j = e - i * n;
    double d = 0.0;
    if (i != j) {
        const float* a = pts + (size_t)near[(size_t)i * n + j] * stride;
        const float* b = pts + (size_t)near[(size_t)j * n + i] * stride;
        const double dx = (double)a[0] - (double)b[0], dy = (double)a[1] - (double)b[1], dz = (double)a[2] - (double)b[2];
        d = sqrt((dx * dx + dy * dy) + dz * dz);
    }
    min_dis[e] = d;
}

int launch_graph_edges(const float* pts, int stride, const int32_t* point_node, int P, int n, const double* centers,
                       double* min_dis, void* ws, hipStream_t stream) {
    if (n == 0) return SGPR_OK;
    int* near = static_cast<int*>(ws);
    hipLaunchKernelGGL(graph_nearest_kernel, dim3(n * n), dim3(CL_THREADS), 0, stream, pts, stride, point_node, P, n, centers, near);
    hipLaunchKernelGGL(graph_min_dis_kernel, dim3((n * n + CL_THREADS - 1) / CL_THREADS), dim3(CL_THREADS), 0, stream, pts,
                       stride, n, near, min_dis);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "sgpr_graph_edges: launch");
    return SGPR_OK;
}

static unsigned table_slots(int P) {
    unsigned h = 1024;
    while (h < 2u * (unsigned)P) h <<= 1;
    return h;
}

static size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

size_t cluster_ws_bytes(int P) {
    const size_t p = (size_t)P, h = table_slots(P);
    return align256(p * 4) * 5 + align256(p * 24) + align256(h * 8) + align256(h * 4) + align256((size_t)CL_INST_SLOTS * 8) +
           align256((size_t)CL_INST_SLOTS * 4) + 256 + 256 + align256((size_t)CL_MAX_CAND * sizeof(int4));
}

int launch_cluster_scan(const float* pts, int stride, const uint32_t* label, int P, int max_nodes, double* centers,
                        int32_t* node_labels, int32_t* node_sizes, int32_t* point_node, int32_t* num_nodes, void* ws,
                        hipStream_t stream) {
    unsigned char* b = static_cast<unsigned char*>(ws);
    auto take = [&](size_t bytes) {
        unsigned char* r = b;
        b += align256(bytes);
        return r;
    };
    ClusterWs w;
    const size_t p = (size_t)P;
    w.H = table_slots(P);
    w.cls = reinterpret_cast<int*>(take(p * 4));
    w.parent = reinterpret_cast<int*>(take(p * 4));
    w.next = reinterpret_cast<int*>(take(p * 4));
    w.size = reinterpret_cast<int*>(take(p * 4));
    w.node_of_root = reinterpret_cast<int*>(take(p * 4));
    w.sum = reinterpret_cast<long long*>(take(p * 24));
    w.cell_key = reinterpret_cast<unsigned long long*>(take((size_t)w.H * 8));
    w.cell_head = reinterpret_cast<int*>(take((size_t)w.H * 4));
    w.inst_key = reinterpret_cast<unsigned long long*>(take((size_t)CL_INST_SLOTS * 8));
    w.inst_min = reinterpret_cast<int*>(take((size_t)CL_INST_SLOTS * 4));
    w.class_inst = reinterpret_cast<int*>(take(256));
    w.count = reinterpret_cast<int*>(take(256));
    w.cand = reinterpret_cast<int4*>(take((size_t)CL_MAX_CAND * sizeof(int4)));
    hipError_t e = hipMemsetAsync(w.node_of_root, 0xff, p * 4, stream);
    // cell_key | cell_head: empty keys, list ends (-1)
    if (e == hipSuccess) e = hipMemsetAsync(w.cell_key, 0xff, align256((size_t)w.H * 8) + align256((size_t)w.H * 4), stream);
    // inst_key empty (all ones); inst_min = INT_MAX-ish (0x7f7f7f7f)
    if (e == hipSuccess) e = hipMemsetAsync(w.inst_key, 0xff, align256((size_t)CL_INST_SLOTS * 8), stream);
    if (e == hipSuccess) e = hipMemsetAsync(w.inst_min, 0x7f, align256((size_t)CL_INST_SLOTS * 4), stream);
    if (e == hipSuccess) e = hipMemsetAsync(w.class_inst, 0, 512, stream);     // class flags + counters
    if (e != hipSuccess) return hip_fail(e, "sgpr_cluster_scan: memset");
    if (P > 0) {
        const dim3 grid((P + CL_THREADS - 1) / CL_THREADS), block(CL_THREADS);
        hipLaunchKernelGGL(cluster_init_kernel, grid, block, 0, stream, w, label, P);
        hipLaunchKernelGGL(cluster_link_kernel, grid, block, 0, stream, w, pts, stride, label, P);
        hipLaunchKernelGGL(cluster_union_kernel, grid, block, 0, stream, w, pts, stride, label, P);
        hipLaunchKernelGGL(cluster_flatten_kernel, grid, block, 0, stream, w, pts, stride, P);
        hipLaunchKernelGGL(cluster_collect_kernel, grid, block, 0, stream, w, label, P);
    }
    hipLaunchKernelGGL(cluster_rank_kernel, dim3(CL_MAX_CAND / CL_THREADS), dim3(CL_THREADS), 0, stream, w, max_nodes, centers,
                       node_labels, node_sizes, num_nodes);
    if (P > 0 && point_node)
        hipLaunchKernelGGL(cluster_assign_kernel, dim3((P + CL_THREADS - 1) / CL_THREADS), dim3(CL_THREADS), 0, stream, w, P, point_node);
    e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "sgpr_cluster_scan: launch");
    return SGPR_OK;
}

}  // namespace sgpr
