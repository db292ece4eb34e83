"""Upstream of the hot path: labelled LiDAR scan -> semantic graph, the counterpart of the reference's
`data_process/gen_label_graph.py` without ROS, open3d or PCL.

    python -m sg_pr_amd.gen_label_graph <scan.bin> <scan.label> <out.json> [pose as 12 numbers]

`gen_labels` + `gen_graphs` (gen_label_graph.py:196-398) run on the MI355X (`sgpr_cluster_scan`, `sgpr_graph_edges`,
sg_pr_amd/csrc/sgpr_cluster.hip): label remapping, per-class instance grouping / Euclidean clustering, centroids, node
labels and the edge distances; the host only formats the result.  The clustering restates what PCL's
EuclideanClusterExtraction computes (connected components of "squared distance < tolerance^2", sizes within
[min, 50000], largest first); PCL is not available to this build, so that step is parity-unpinned against PCL itself
(oracle/graph_oracle.py states the same).  GPU only: there is no CPU fallback.
"""
import json
import sys

import numpy as np
import torch

from . import engine as _engine

# gen_label_graph.py:23-58 and :64-77, names kept
learning_map = {0: 0, 1: 0, 10: 1, 11: 2, 13: 5, 15: 3, 16: 5, 18: 4, 20: 5, 30: 6, 31: 7, 32: 8, 40: 9, 44: 10, 48: 11,
                49: 12, 50: 13, 51: 14, 52: 0, 60: 9, 70: 15, 71: 16, 72: 17, 80: 18, 81: 19, 99: 0, 252: 1, 253: 7,
                254: 6, 255: 8, 256: 5, 257: 5, 258: 4, 259: 5}
max_key = max(learning_map.keys())
remap_lut = np.zeros((max_key + 100), dtype=np.int32)
remap_lut[list(learning_map.keys())] = list(learning_map.values())
node_map = {1: 0, 4: 1, 5: 2, 11: 3, 12: 4, 13: 5, 14: 6, 15: 7, 16: 8, 17: 9, 18: 10, 19: 11}
_node_map_inv = {v: k for k, v in node_map.items()}


class ScanClusters:
    """Device result of one scan: node centres / labels / sizes, point -> node, and the inputs they refer to."""

    def __init__(self, points, label, centers, node_labels, node_sizes, point_node):
        self.points, self.label = points, label
        self.centers, self.node_labels, self.node_sizes, self.point_node = centers, node_labels, node_sizes, point_node

    def __len__(self):
        return int(self.node_labels.shape[0])


def cluster_scan(points, label, max_nodes=1024, device=0):
    """points [P,4] float32 (x, y, z, remission) and label uint32 [P] (semantic | instance << 16) -> ScanClusters."""
    dev = torch.device("cuda", device)
    pts = torch.as_tensor(np.ascontiguousarray(points, dtype=np.float32)).to(dev)
    lab = np.ascontiguousarray(label).reshape(-1)
    if lab.shape[0] != pts.shape[0]:
        raise ValueError("Scan and Label don't contain same number of points")      # gen_label_graph.py:232
    lab_t = torch.as_tensor(lab.astype(np.uint32).view(np.int32)).to(dev)
    centers, nlab, nsize, pnode = _engine.cluster_scan(pts, lab_t, max_nodes=max_nodes, want_point_node=True)
    return ScanClusters(pts, lab_t, centers, nlab, nsize, pnode)


def gen_labels(points, label, device=0):
    """gen_label_graph.py:196-326 (no demolition, no ROS): -> the `cluster` array [N,6] float64 with rows
    (x, y, z, remission, training label, instance id): clusters in class order, ids counted like the reference's
    `inst_id`; road / parking appear as one cluster each.  Rows inside a cluster are in scan order."""
    sc = cluster_scan(points, label, device=device)
    pts = np.asarray(points, dtype=np.float32)
    sem = remap_lut[np.asarray(label).reshape(-1).astype(np.uint32) & 0xFFFF]
    pnode = sc.point_node.cpu().numpy()
    node_class = np.array([_node_map_inv[int(v)] for v in sc.node_labels.cpu().numpy()], dtype=np.int64)
    blocks, inst_id = [], 0
    for label_i in sorted(set(sem.tolist())):
        if label_i in (9, 10):
            groups = [np.flatnonzero(sem == label_i)]
        else:
            groups = [np.flatnonzero(pnode == k) for k in np.flatnonzero(node_class == label_i)]
        for g in groups:
            blocks.append(np.concatenate((pts[g].astype(np.float64), np.full((len(g), 1), float(label_i)),
                                          np.full((len(g), 1), float(inst_id))), axis=1))
            inst_id += 1
    return np.concatenate(blocks, axis=0) if blocks else np.zeros((0, 6))


def gen_graphs(clusters, dist_thresh=5.0, with_edges=True):
    """gen_label_graph.py:336-398 for a ScanClusters: {"nodes", "edges", "weights", "centers"} - nodes and centres as
    computed on the device, edges from the device's pairwise closest-point distances (<= dist_thresh, weight 1 - d/5)."""
    nodes = [int(v) for v in clusters.node_labels.cpu().numpy()]
    centers = clusters.centers.cpu().numpy().tolist()
    edges, weights = [], []
    if with_edges and len(nodes) > 1:
        d = _engine.graph_edges(clusters.points, clusters.point_node, clusters.centers).cpu().numpy()
        for i in range(len(nodes) - 1):
            for j in range(i + 1, len(nodes)):
                if d[i, j] <= dist_thresh:
                    edges.append([i, j])
                    weights.append(float(1 - d[i, j] / dist_thresh))
    return {"nodes": nodes, "edges": edges, "weights": weights, "centers": centers}


def scan_to_graph(points, label, pose=None, with_edges=False, device=0):
    """One scan -> the graph dictionary the scorer's loaders read (utils.py:21-38: nodes, centers, pose)."""
    g = gen_graphs(cluster_scan(points, label, device=device), with_edges=with_edges)
    if pose is not None:
        g["pose"] = [float(v) for v in np.asarray(pose).reshape(-1)[:12]]
    return g


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    scan = np.fromfile(argv[0], dtype=np.float32).reshape((-1, 4))                  # gen_label_graph.py:200-201
    label = np.fromfile(argv[1], dtype=np.uint32).reshape((-1))                     # :206-207
    pose = [float(v) for v in argv[3:15]] if len(argv) >= 15 else None
    graph = scan_to_graph(scan, label, pose=pose, with_edges=True)
    with open(argv[2], "w", encoding="utf-8") as f:
        json.dump(graph, f)
    print("nodes", len(graph["nodes"]), "edges", len(graph["edges"]))


if __name__ == "__main__":
    main()
