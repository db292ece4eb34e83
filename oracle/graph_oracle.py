"""CPU ORACLE for the upstream semantic-graph generation.  TEST INFRASTRUCTURE ONLY.

A numpy / scipy restatement of the reference's `gen_labels` + `gen_graphs`
(data_process/gen_label_graph.py:196-334 and :336-398) without ROS, open3d or PCL:
only `tests/` may import it.

PARITY UNPINNED against PCL: the reference clusters with
`pcl.EuclideanClusterExtraction` (gen_label_graph.py:300-307); neither PCL nor python-pcl
is installed here and the reference ships no scan, label file or expected graph for this
step, so this file restates PCL 1.x's published algorithm instead:
  * radius search = FLANN `RadiusResultSet::addPoint`: neighbours are the points with
    squared L2 distance STRICTLY below tolerance^2 (float arithmetic);
  * clusters = connected components of that neighbourhood graph whose size lies in
    [min_size, max_size] (`extractEuclideanClusters`);
  * `EuclideanClusterExtraction::extract` returns them largest first (`std::sort` on
    sizes - the order of equal-sized clusters is unspecified there; here: the cluster
    holding the lowest point index first).
Everything else (label remapping, per-class modes, size rules, node_map, centroids,
edges and weights) follows the reference line by line.
"""
import numpy as np
from scipy.sparse import coo_matrix
from scipy.sparse.csgraph import connected_components
from scipy.spatial import cKDTree

# gen_label_graph.py:23-58 (SemanticKITTI raw label -> training id) and :64-77 (training id -> graph node class)
LEARNING_MAP = {0: 0, 1: 0, 10: 1, 11: 2, 13: 5, 15: 3, 16: 5, 18: 4, 20: 5, 30: 6, 31: 7, 32: 8, 40: 9, 44: 10,
                48: 11, 49: 12, 50: 13, 51: 14, 52: 0, 60: 9, 70: 15, 71: 16, 72: 17, 80: 18, 81: 19, 99: 0, 252: 1,
                253: 7, 254: 6, 255: 8, 256: 5, 257: 5, 258: 4, 259: 5}
NODE_MAP = {1: 0, 4: 1, 5: 2, 11: 3, 12: 4, 13: 5, 14: 6, 15: 7, 16: 8, 17: 9, 18: 10, 19: 11}
MAX_CLUSTER = 50000          # gen_label_graph.py:305


def remap_lut():
    """gen_label_graph.py:60-62."""
    lut = np.zeros(max(LEARNING_MAP) + 100, dtype=np.int32)
    lut[list(LEARNING_MAP.keys())] = list(LEARNING_MAP.values())
    return lut


def cluster_params(label_i):
    """gen_label_graph.py:283-297: (tolerance [m], minimum size) of the Euclidean clustering of class label_i."""
    if label_i in (1, 4, 5, 14):
        tol = 0.5
    elif label_i in (11, 12, 13, 15, 17):
        tol = 2.0
    else:
        tol = 0.2
    if label_i in (16, 19):
        mn = 50
    elif label_i == 15:
        mn = 200
    elif label_i in (11, 12, 13, 17):
        mn = 300
    else:
        mn = 100
    return tol, mn


def euclidean_clusters(xyz, tol, min_size, max_size=MAX_CLUSTER):
    """PCL EuclideanClusterExtraction restated (see the module docstring) -> list of index arrays, largest first."""
    n = xyz.shape[0]
    if n == 0:
        return []
    x = np.ascontiguousarray(xyz, dtype=np.float32)
    pairs = cKDTree(x.astype(np.float64)).query_pairs(float(tol) * 1.0001, output_type="ndarray")
    if pairs.size:
        d = x[pairs[:, 0]] - x[pairs[:, 1]]                      # float32 arithmetic, fixed order: (dx^2 + dy^2) + dz^2
        d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
        pairs = pairs[d2 < np.float32(tol) * np.float32(tol)]
    g = coo_matrix((np.ones(len(pairs), dtype=np.int8), (pairs[:, 0], pairs[:, 1])), shape=(n, n))
    _, comp = connected_components(g, directed=False)
    order = np.argsort(comp, kind="stable")
    bounds = np.flatnonzero(np.diff(comp[order])) + 1
    groups = np.split(order, bounds)
    groups = [gidx for gidx in groups if min_size <= len(gidx) <= max_size]
    groups.sort(key=lambda gidx: (-len(gidx), int(gidx.min())))
    return groups


def gen_labels(points, label):
    """gen_label_graph.py:196-326 (no demolition): points [P,4] float32 (x, y, z, remission), label uint32 [P]
    (semantic id | instance id << 16) -> cluster array [N,6] float64: x, y, z, r, training label, instance id."""
    points = np.asarray(points, dtype=np.float32)
    label = np.asarray(label, dtype=np.uint32).reshape(-1)
    if label.shape[0] != points.shape[0]:
        raise ValueError("Scan and Label don't contain same number of points")
    sem_label = remap_lut()[label & 0xFFFF]
    inst_label = label >> 16
    cluster, inst_id = [], 0
    for label_i in sorted(set(sem_label.tolist())):
        index = np.flatnonzero(sem_label == label_i)
        sem_cluster = points[index]
        tmp_inst_label = inst_label[index]
        tmp_inst_set = sorted(set(tmp_inst_label.tolist()))

        def emit(rows):
            nonlocal inst_id
            block = np.concatenate((rows.astype(np.float64), np.full((len(rows), 1), label_i, dtype=np.float64),
                                    np.full((len(rows), 1), inst_id, dtype=np.float64)), axis=1)
            inst_id += 1
            cluster.append(block)

        if label_i in (9, 10):                                   # road / parking: one cluster, never a node
            emit(sem_cluster)
        elif label_i in (0, 2, 3, 6, 7, 8):                      # discarded classes
            continue
        elif len(tmp_inst_set) > 1 or (len(tmp_inst_set) == 1 and tmp_inst_set[0] != 0):   # instance labels present
            for label_j in tmp_inst_set:
                points_index = np.flatnonzero(tmp_inst_label == label_j)
                if len(points_index) <= 20:
                    continue
                emit(sem_cluster[points_index])
        else:
            tol, mn = cluster_params(label_i)
            for indices in euclidean_clusters(sem_cluster[:, :3], tol, mn):
                emit(sem_cluster[indices, 0:4])
    return np.concatenate(cluster, axis=0) if cluster else np.zeros((0, 6))


def gen_graphs(scan, dist_thresh=5.0, with_edges=True):
    """gen_label_graph.py:336-398: cluster array -> {"nodes", "edges", "weights", "centers"}."""
    inst = scan[:, -1]
    nodes, centers, clusters = [], [], []
    for inst_id in sorted(set(inst.tolist())):
        inst_cluster = scan[inst == inst_id]
        sem = set(inst_cluster[:, -2].tolist())
        assert len(sem) == 1
        sem = int(sem.pop())
        if sem in NODE_MAP:
            clusters.append(inst_cluster[:, :3])
            nodes.append(int(NODE_MAP[sem]))
            centers.append(np.mean(inst_cluster[:, :3], axis=0).tolist())
        elif sem in (9, 10):
            continue
        else:
            raise ValueError("wrong semantic label: %r" % sem)
    edges, weights = [], []
    if with_edges:
        for i in range(len(clusters) - 1):
            for j in range(i + 1, len(clusters)):
                pc_i, pc_j = clusters[i], clusters[j]
                center = np.mean([np.mean(pc_i, axis=0), np.mean(pc_j, axis=0)], axis=0)
                index1 = np.argmin(np.linalg.norm(center - pc_i, axis=-1))
                index2 = np.argmin(np.linalg.norm(center - pc_j, axis=-1))
                min_dis = np.linalg.norm(pc_i[index1] - pc_j[index2], axis=-1)
                if min_dis <= dist_thresh:
                    edges.append([i, j])
                    weights.append(float(1 - min_dis / dist_thresh))
    return {"nodes": nodes, "edges": edges, "weights": weights, "centers": centers}
