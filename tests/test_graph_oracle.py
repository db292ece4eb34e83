"""CPU checks of the graph-generation oracle (oracle/graph_oracle.py) and of the synthetic labelled scan:
the oracle is parity-unpinned against PCL (not installed, no reference fixtures), so it is held to the published
definition of Euclidean cluster extraction by brute force and to the per-class rules of gen_label_graph.py:259-305."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def go():
    from oracle import graph_oracle
    return graph_oracle


def _brute_components(x, tol):
    n = len(x)
    parent = list(range(n))

    def find(a):
        while parent[a] != a:
            parent[a] = parent[parent[a]]
            a = parent[a]
        return a
    x = x.astype(np.float32)
    for i in range(n):
        d = x[i] - x[:i]
        d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
        for j in np.flatnonzero(d2 < np.float32(tol) * np.float32(tol)):
            a, b = find(i), find(int(j))
            if a != b:
                parent[max(a, b)] = min(a, b)
    return np.array([find(i) for i in range(n)])


def test_euclidean_clusters_is_connected_components(go):
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.normal(c, 0.25, size=(m, 3)) for c, m in ((0, 80), (3, 40), (6, 9), (9, 120))]).astype(np.float32)
    roots = _brute_components(x, 0.5)
    want = [np.flatnonzero(roots == r) for r in np.unique(roots)]
    want = sorted((g for g in want if 10 <= len(g) <= 100), key=lambda g: (-len(g), g.min()))
    got = go.euclidean_clusters(x, 0.5, 10, 100)
    assert len(got) == len(want) and len(got) >= 2
    for a, b in zip(got, want):
        np.testing.assert_array_equal(np.sort(a), b)
    assert go.euclidean_clusters(x[:0], 0.5, 1) == []


def test_euclidean_clusters_against_sklearn_dbscan(go):
    """An independent, published implementation of the same definition: with min_samples = 1 every point is a core
    point and DBSCAN's clusters are exactly the connected components of the radius graph that PCL's
    EuclideanClusterExtraction (gen_label_graph.py:196-243) grows - size limits applied afterwards."""
    cluster = pytest.importorskip("sklearn.cluster")
    rng = np.random.default_rng(5)
    for tol, lo, hi in ((0.2, 50, 10000), (0.5, 100, 10000), (2.0, 200, 10000)):
        blobs = [rng.normal(rng.uniform(-20, 20, 3), rng.uniform(0.05, 0.6) * tol * 2, size=(int(rng.integers(5, 400)), 3))
                 for _ in range(25)]
        x = np.concatenate(blobs).astype(np.float32)
        lab = cluster.DBSCAN(eps=tol, min_samples=1).fit(x.astype(np.float64)).labels_
        want = [np.flatnonzero(lab == c) for c in np.unique(lab)]
        want = sorted((g for g in want if lo <= len(g) <= hi), key=lambda g: (-len(g), g.min()))
        got = go.euclidean_clusters(x, tol, lo, hi)
        assert len(got) == len(want) and len(got) >= 1, (tol, len(got), len(want))
        for a, b in zip(got, want):
            np.testing.assert_array_equal(np.sort(a), b)


def test_lut_and_class_rules(go):
    from sg_pr_amd import gen_label_graph as product
    np.testing.assert_array_equal(go.remap_lut(), product.remap_lut)
    assert go.NODE_MAP == product.node_map and go.LEARNING_MAP == product.learning_map
    assert go.cluster_params(1) == (0.5, 100) and go.cluster_params(15) == (2.0, 200) and go.cluster_params(13) == (2.0, 300)
    assert go.cluster_params(16) == (0.2, 50) and go.cluster_params(19) == (0.2, 50) and go.cluster_params(18) == (0.2, 100)
    assert go.cluster_params(14) == (0.5, 100) and go.cluster_params(12) == (2.0, 300)


def test_synthetic_scan_exercises_every_mode(go):
    from sg_pr_amd import synth
    pts, lab = synth.labelled_scan(seed=3, scale=0.4)
    assert pts.dtype == np.float32 and pts.shape[1] == 4 and lab.dtype == np.uint32 and len(lab) == len(pts)
    cl = go.gen_labels(pts, lab)
    sem = cl[:, 4].astype(int)
    assert {9, 10} <= set(sem) and not ({0, 2, 3, 6, 7, 8} & set(sem))          # road / parking kept, discarded classes gone
    g = go.gen_graphs(cl)
    assert len(g["nodes"]) == len(g["centers"]) >= 20 and set(g["nodes"]) <= set(range(12))
    # instance mode: cars = six instances + the "instance 0" group (> 20 points); the 15-point instance is dropped
    assert g["nodes"].count(0) == 7 and g["nodes"].count(1) == 1
    # every Euclidean cluster respects its class minimum; the undersized objects produced nothing
    inst = cl[:, 5].astype(int)
    for i in np.unique(inst):
        c, n = int(sem[inst == i][0]), int((inst == i).sum())
        if c in (1, 4):
            assert n > 20
        elif c not in (9, 10):
            assert n >= go.cluster_params(c)[1]
    assert all(0.0 <= w <= 1.0 for w in g["weights"]) and all(i < j for i, j in g["edges"])
    # the node set does not depend on the order of the points in the scan
    perm = np.random.default_rng(1).permutation(len(lab))
    g2 = go.gen_graphs(go.gen_labels(pts[perm], lab[perm]), with_edges=False)
    key = lambda gr: sorted((n, tuple(np.round(c, 9))) for n, c in zip(gr["nodes"], gr["centers"]))   # noqa: E731
    assert key(g) == key(g2)
