"""Seeded synthetic semantic graphs for the bench / parity configurations.

The reference ships only three real graphs (data/{0,3,250}.json); KITTI graph
JSONs live off-repo (reference README.md:54).  These generators follow the
specification in SURVEY.md §8(d) (configs 2, 3 and 5 of BASELINE.json): the
packed wire format is what `SGTrainer.transfer_to_torch` (reference
sg_net.py:241-310) produces before the one-hot expansion, i.e.

    centers  float32 [G, N, 3]   zero for padded slots        (sg_net.py:262)
    labels   int32   [G, N]      -1   for padded slots        (sg_net.py:260)

Every graph keeps >= k padded slots so that kNN ties are only ever between
feature-identical nodes (SURVEY.md §7 finding 3) - the regime in which the
reference itself is deterministic across torch.topk implementations.

numpy only; no torch import here.
"""
import numpy as np

NUM_LABELS = 12

# label histogram of the three shipped graphs (classes 0,3,5..11 dominant)
_KITTI_LABEL_P = np.array([28, 0, 0, 10, 0, 16, 7, 25, 12, 4, 7, 3], dtype=np.float64)
_KITTI_LABEL_P /= _KITTI_LABEL_P.sum()


def make_graphs(num_graphs, node_num, n_real_lo, n_real_hi, seed, kitti_like=False):
    """Return (centers f32 [G,N,3], labels i32 [G,N], n_real i32 [G])."""
    rng = np.random.default_rng(seed)
    centers = np.zeros((num_graphs, node_num, 3), dtype=np.float32)
    labels = -np.ones((num_graphs, node_num), dtype=np.int32)
    n_real = rng.integers(n_real_lo, n_real_hi + 1, size=num_graphs).astype(np.int32)
    for g in range(num_graphs):
        n = int(n_real[g])
        xy = rng.uniform(-50.0, 50.0, size=(n, 2))
        z = rng.uniform(-2.0, 1.0, size=(n, 1))
        if kitti_like:
            lab = rng.choice(NUM_LABELS, size=n, p=_KITTI_LABEL_P)
        else:
            lab = rng.integers(0, NUM_LABELS, size=n)
        lab.sort()
        centers[g, :n, :2] = xy
        centers[g, :n, 2:] = z
        labels[g, :n] = lab
    return centers, labels, n_real


def config2_pairs(seed=0, batch=128, node_num=64):
    """BASELINE config 2: `batch` pairs, N=64 slots, k=10, n_real ~ U{20..54}.

    Graph 2b is side 1 and graph 2b+1 is side 2 of pair b."""
    return make_graphs(2 * batch, node_num, 20, node_num - 10, seed)


def config5_pairs(seed=0, batch=1024, node_num=256):
    """BASELINE config 5 (stress): N=256, k=20, n_real ~ U{100..236}."""
    return make_graphs(2 * batch, node_num, 100, node_num - 20, seed)


def kitti_like_sequence(num_graphs=4541, node_num=100, seed=0):
    """BASELINE config 3 stand-in: KITTI-00-sized sequence (4541 frames).

    n_real ~ U{25..60}, shipped-graph label histogram, poses = planar random
    walk with revisits so that d<=3 m and d>=20 m pairs both exist.
    Returns (centers, labels, n_real, poses f64 [G,12])."""
    centers, labels, n_real = make_graphs(num_graphs, node_num, 25, 60, seed, kitti_like=True)
    rng = np.random.default_rng(seed + 7919)
    heading = np.cumsum(rng.normal(0.0, 0.05, size=num_graphs))
    step = np.stack([np.cos(heading), np.sin(heading)], axis=1)  # 1 m per frame
    xz = np.cumsum(step, axis=0)
    # revisit: the last third of the sequence re-drives the first third (+noise)
    third = num_graphs // 3
    if third > 0:
        xz[-third:] = xz[:third] + rng.normal(0.0, 0.5, size=(third, 2))
    poses = np.zeros((num_graphs, 12), dtype=np.float64)
    poses[:, 0] = poses[:, 5] = poses[:, 10] = 1.0
    poses[:, 3] = xz[:, 0]   # x   (utils.py:36 uses pose[3], pose[11])
    poses[:, 11] = xz[:, 1]  # z
    return centers, labels, n_real, poses


def world_sequence(num_graphs=4541, node_num=100, seed=0, sensor_range=50.0, center_noise=0.15, dropout=0.12,
                   density=0.0062, region=60.0, mix_alpha=3.0):
    """A KITTI-00-sized sequence whose graphs come from ONE world, so that revisits look alike and the
    precision-recall curve of the score matrix means something (`kitti_like_sequence` draws every graph
    independently of its pose: its F1-max is chance).

    The world is a fixed set of landmarks (label from the shipped-graph histogram, position uniform over the
    trajectory's bounding box + sensor_range, height in [-2, 1] like the shipped graphs).  The vehicle drives a
    planar random walk at 1 m per frame; the last third re-drives the first third (lateral offset ~0.5 m, heading
    within a few degrees).  Frame t sees the landmarks within `sensor_range` of its pose, in the sensor frame
    (x forward, y left, z up: the frame the shipped graphs' centres are in), each one dropped with probability
    `dropout` and its centre jittered by N(0, center_noise) m - what a per-frame clustering does to a real
    object.  At most node_num - 40 nodes are kept (nearest first), i.e. >= 40 padded slots: the tie regime stays
    the reference's deterministic one.  Nodes are listed label-ascending like the shipped graphs.
    Returns (centers f32 [G,N,3], labels i32 [G,N], n_real i32 [G], poses f64 [G,12]) - the KITTI 3x4 pose with
    the heading as a rotation about the camera's y axis; utils.py:36 reads entries 3 (x) and 11 (z)."""
    rng = np.random.default_rng(seed + 104729)
    G = int(num_graphs)
    heading = np.cumsum(rng.normal(0.0, 0.05, size=G))
    xz = np.cumsum(np.stack([np.cos(heading), np.sin(heading)], axis=1), axis=0)
    third = G // 3
    if third > 0:
        xz[-third:] = xz[:third] + rng.normal(0.0, 0.5, size=(third, 2))
        heading[-third:] = heading[:third] + rng.normal(0.0, 0.03, size=third)
    lo, hi = xz.min(0) - sensor_range, xz.max(0) + sensor_range
    n_land = max(int(density * float(np.prod(hi - lo))), 1)
    land_xz = rng.uniform(lo, hi, size=(n_land, 2))
    land_h = rng.uniform(-2.0, 1.0, size=n_land)
    # places differ: every `region` x `region` m cell of the world has its own label mix (a Dirichlet draw around the
    # shipped-graph histogram) - a street of buildings and fences here, vegetation and trunks there
    reg = np.floor((land_xz - lo) / region).astype(np.int64)
    nreg = reg.max(0) + 1
    mix = rng.dirichlet(_KITTI_LABEL_P * mix_alpha + 1e-3, size=int(nreg[0] * nreg[1]))
    cum = np.cumsum(mix[reg[:, 0] * nreg[1] + reg[:, 1]], axis=1)
    land_lab = np.minimum((rng.random(n_land)[:, None] > cum).sum(1), NUM_LABELS - 1)
    # landmarks nowhere near the trajectory are never seen: drop them before the per-frame search
    cell = np.floor(xz / sensor_range).astype(np.int64)
    near = set()
    for dx in (-1, 0, 1):
        for dz in (-1, 0, 1):
            near.update(map(tuple, cell + (dx, dz)))
    keep = np.fromiter((tuple(c) in near for c in np.floor(land_xz / sensor_range).astype(np.int64)), bool, n_land)
    land_xz, land_h, land_lab = land_xz[keep], land_h[keep], land_lab[keep]
    max_real = max(node_num - 40, 1)
    centers = np.zeros((G, node_num, 3), dtype=np.float32)
    labels = -np.ones((G, node_num), dtype=np.int32)
    n_real = np.zeros(G, dtype=np.int32)
    for t in range(G):
        d = land_xz - xz[t]
        r2 = (d * d).sum(1)
        seen = np.flatnonzero(r2 <= sensor_range * sensor_range)
        seen = seen[rng.random(seen.size) >= dropout]
        if seen.size > max_real:
            seen = seen[np.argsort(r2[seen], kind="stable")[:max_real]]
        c, s = np.cos(heading[t]), np.sin(heading[t])
        fwd = d[seen, 0] * c + d[seen, 1] * s
        left = -d[seen, 0] * s + d[seen, 1] * c
        pts = np.stack([fwd, left, land_h[seen]], axis=1) + rng.normal(0.0, center_noise, size=(seen.size, 3))
        order = np.argsort(land_lab[seen], kind="stable")
        n = seen.size
        centers[t, :n] = pts[order]
        labels[t, :n] = land_lab[seen][order]
        n_real[t] = n
    poses = np.zeros((G, 12), dtype=np.float64)
    poses[:, 0] = poses[:, 10] = np.cos(heading)
    poses[:, 2] = np.sin(heading)
    poses[:, 8] = -np.sin(heading)
    poses[:, 5] = 1.0
    poses[:, 3] = xz[:, 0]
    poses[:, 11] = xz[:, 1]
    return centers, labels, n_real, poses


def dense_features(centers, labels, num_labels=NUM_LABELS):
    """Packed (centers, labels) -> the reference's dense `B x (3+L) x N` float32
    tensor (sg_net.py:274-298): xyz rows then a one-hot block, all-zero for -1."""
    g, n, _ = centers.shape
    out = np.zeros((g, 3 + num_labels, n), dtype=np.float32)
    out[:, :3, :] = centers.transpose(0, 2, 1)
    gi, ni = np.nonzero(labels >= 0)
    out[gi, 3 + labels[gi, ni], ni] = 1.0
    return out


def effective_nodes(centers, labels, k):
    """Slots the engine actually processes per graph: of the m trailing slots identical to the last one
    (zero padding) a single representative is kept when m >= k, all m otherwise ("duplicate
    slots").  Returns int32 [G]."""
    g, n = labels.shape
    same = (labels == labels[:, -1:]) & (centers == centers[:, -1:, :]).all(-1)     # [G, N]
    differs = ~same
    last = np.where(differs.any(1), n - 1 - np.argmax(differs[:, ::-1], axis=1), -1)
    nd = last + 1
    m = n - nd
    return (nd + np.where((m >= k) & (m > 1), 1, m)).astype(np.int32)


def labelled_scan(seed=0, scale=1.0):
    """A SemanticKITTI-like labelled LiDAR scan for the upstream graph generator (the reference ships
    no scans): points float32 [P,4] (x, y, z, remission) and labels uint32 [P] (raw semantic id | instance id << 16,
    the .label format gen_label_graph.py:204-227 reads).  Contains every per-class mode of gen_labels: road / parking
    (one cluster, never a node), discarded classes, instance-labelled vehicles (incl. an instance of <= 20 points and
    points with instance id 0), and Euclidean classes at all three tolerances with objects above and below their
    minimum sizes.  `scale` multiplies the point counts (1.0 -> ~70 k points).  Point order is shuffled."""
    rng = np.random.default_rng(seed)
    pts, lab = [], []

    def add(xyz, sem, inst=0):
        pts.append(np.asarray(xyz, dtype=np.float64))
        lab.append(np.full(len(xyz), sem | (inst << 16), dtype=np.uint32))

    def n(x):
        return max(1, int(round(x * scale)))

    def box(c, size, count):
        return np.asarray(c) + (rng.random((count, 3)) - 0.5) * np.asarray(size)

    add(box((0, 0, -1.7), (100, 8, 0.05), n(15000)), 40)                       # road
    add(box((20, 9, -1.7), (12, 6, 0.05), n(1500)), 44)                        # parking
    add(box((0, 6.5, -1.6), (100, 3, 0.05), n(5000)), 48)                      # sidewalk: two strips > 2 m apart
    add(box((0, -6.5, -1.6), (100, 3, 0.05), n(5000)), 48)
    add(box((60, 20, -1.6), (2, 2, 0.05), n(120)), 48)                         # ... and a patch below min size 300
    for b in range(6):                                                         # buildings, > 2 m apart
        add(box((-45 + 18 * b, 18 + (b % 2) * 4, 2), (10, 0.3, 8), n(1500 + 300 * b)), 50)
    add(box((48, -18, 1), (1.5, 0.3, 4), n(200)), 50)                          # too small (min 300)
    for v in range(10):                                                        # vegetation blobs (tol 2, min 200)
        add(box((-40 + 9 * v, -14 - (v % 3) * 5, 0.5), (4, 4, 3), n(150 + 180 * v)), 70)
    for t in range(8):                                                         # trunks: dense vertical lines (tol 0.2, min 50)
        h = np.sort(rng.random(n(40 + 15 * t))) * 3.0 - 1.5
        add(np.stack((np.full_like(h, -40 + 9 * t) + rng.normal(0, 0.01, h.size),
                      np.full_like(h, -11.0) + rng.normal(0, 0.01, h.size), h), axis=1), 71)
    for t in range(5):                                                         # poles (tol 0.2, min 100)
        h = np.sort(rng.random(n(80 + 20 * t))) * 5.0 - 1.5
        add(np.stack((np.full_like(h, -30 + 15 * t) + rng.normal(0, 0.01, h.size),
                      np.full_like(h, 8.2) + rng.normal(0, 0.01, h.size), h), axis=1), 80)
    for t in range(4):                                                         # traffic signs (tol 0.2, min 50)
        add(box((-25 + 15 * t, 8.2, 3.2), (0.6, 0.05, 0.6), n(40 + 15 * t)), 81)
    add(box((0, 12, -0.5), (60, 0.1, 1.2), n(2500)), 51)                       # fence (tol 0.5, min 100)
    add(box((45, 12, -0.5), (3, 0.1, 1.2), n(60)), 51)                         # ... short piece below min size
    add(box((0, -30, -1.5), (100, 12, 0.3), n(9000)), 72)                      # terrain (tol 2, min 300)
    add(box((30, 30, -1.5), (20, 10, 0.3), n(800)), 49)                        # other-ground
    for c in range(6):                                                         # cars with instance ids
        add(box((-35 + 12 * c, 2.0 * (-1) ** c, -0.9), (4, 1.8, 1.5), n(150 + 120 * c)), 10, inst=c + 1)
    add(box((40, 2, -0.9), (4, 1.8, 1.5), 15), 10, inst=9)                     # an instance of <= 20 points: dropped
    add(box((46, -2, -0.9), (4, 1.8, 1.5), n(60)), 10, inst=0)                 # car points without an instance id
    add(box((-20, 26, 0), (8, 2.5, 3), n(900)), 18, inst=1)                    # truck with an instance id
    add(box((10, 27, 0), (10, 2.5, 3), n(700)), 20)                            # other-vehicle WITHOUT instance ids: Euclidean
    add(box((26, 27, 0), (1, 1, 1), n(40)), 20)                                # ... and a fragment below min size 100
    add(box((5, 6, -0.7), (0.5, 0.5, 1.7), n(120)), 30, inst=3)                # person: discarded class
    add(box((0, 0, 5), (120, 80, 1), n(1500)), 0)                              # unlabeled
    add(box((0, 0, 6), (120, 80, 1), n(300)), 1)                               # outlier -> unlabeled
    add(box((-10, 9, -1.2), (1.5, 0.6, 1.0), n(150)), 11, inst=2)              # bicycle: discarded class
    xyz = np.concatenate(pts)
    labels = np.concatenate(lab)
    perm = rng.permutation(len(labels))
    points = np.concatenate((xyz, rng.random((len(labels), 1))), axis=1).astype(np.float32)[perm]
    return np.ascontiguousarray(points), np.ascontiguousarray(labels[perm])
