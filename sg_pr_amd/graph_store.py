"""Packed graph store + whole-sequence evaluation (the caller side of the hot path).

The reference re-reads and re-pads both JSON graphs of every pair (utils.py:27-28, sg_net.py:250-299): ~0.7 ms of
host work per pair, hours for the 20 M pairs of KITTI-00.  Here every graph of a sequence is parsed ONCE into the
engine's wire format

    centers  float32 [M, node_num, 3]   zero for padded slots         (sg_net.py:262)
    labels   int32   [M, node_num]      -1   for padded slots         (sg_net.py:260)
    poses    float64 [M, 12]            the 3x4 KITTI pose rows       (utils.py:36 uses [3] and [11])

kept as one `.npz` per sequence, together with what the engine's launch plan needs (processed slots per graph ->
node_cap and the largest-first order).  `evaluate_all_pairs` then runs the sequence end to end on the GPU: embed
once per graph, dense M x M scores, F1-max from threshold counts and the loop-closure candidates - the score
matrix never leaves the device.
"""
import os
import re

import numpy as np
import torch

from . import metrics
from .sg_net import pack_graph
from .utils import read_graph


class PackedSequence:
    def __init__(self, centers, labels, poses, names):
        self.centers = np.ascontiguousarray(centers, dtype=np.float32)
        self.labels = np.ascontiguousarray(labels, dtype=np.int32)
        self.poses = np.ascontiguousarray(poses, dtype=np.float64)
        self.names = list(names)
        m = self.labels.shape[0]
        if self.centers.shape != (m, self.labels.shape[1], 3) or self.poses.shape != (m, 12) or len(self.names) != m:
            raise ValueError("inconsistent packed sequence")

    def __len__(self):
        return self.labels.shape[0]

    @property
    def node_num(self):
        return self.labels.shape[1]

    def ragged(self, device=None):
        """The sequence as the ragged store of sgpr_embed_ragged: an `allpairs.RaggedGraphs` (centers f32 [S,3], labels
        i8 [S], host offsets i64 [M+1]) - only the real nodes, 13 bytes each instead of 16 per slot: what to keep
        resident for a whole data set and what to move across PCIe.  Accepted wherever graphs are
        (`SG.embed(rag, None)`, `AllPairsScorer.run(rag, None)`, `SequenceSet`)."""
        from .allpairs import RaggedGraphs
        return RaggedGraphs.from_padded(self.centers, self.labels, device=device)

    def save(self, path):
        np.savez_compressed(path, centers=self.centers, labels=self.labels, poses=self.poses,
                            names=np.array(self.names))

    @classmethod
    def load(cls, path):
        with np.load(path, allow_pickle=False) as z:
            return cls(z["centers"], z["labels"], z["poses"], [str(n) for n in z["names"]])


def _natural_key(name):
    return [int(t) if t.isdigit() else t for t in re.split(r"(\d+)", name)]


def pack_directory(graph_dir, node_num, number_of_labels=12, names=None):
    """Parse `<graph_dir>/*.json` (or the listed file names) once -> PackedSequence, frames in natural order."""
    if names is None:
        names = sorted((f for f in os.listdir(graph_dir) if f.endswith(".json")), key=_natural_key)
    m = len(names)
    centers = np.zeros((m, node_num, 3), dtype=np.float32)
    labels = -np.ones((m, node_num), dtype=np.int32)
    poses = np.zeros((m, 12), dtype=np.float64)
    for g, name in enumerate(names):
        d = read_graph(os.path.join(graph_dir, name))
        centers[g], labels[g] = pack_graph(d["centers"], d["nodes"], node_num, number_of_labels)
        poses[g] = np.asarray(d["pose"], dtype=np.float64).reshape(-1)[:12]
    return PackedSequence(centers, labels, poses, names)


def evaluate_all_pairs(model, seq, p_thresh=3.0, n_thresh=20.0, top_k=1, window=50, scorer=None):
    """Whole-sequence evaluation on the device.  Returns {"f1_max", "roc_auc", "closure_scores" [M,k], "closure_frames"
    [M,k], "matrix" (device tensor: this rank's row block)}.  `scorer`: an AllPairsScorer for multi-GPU runs."""
    from . import allpairs
    if scorer is None:
        scorer = allpairs.AllPairsScorer(model=model)
    eng = model.engine()
    k = int(model.args.K)
    world, rank = scorer._world()
    lo, hi = allpairs.shard_bounds(len(seq), world, rank)
    order, cap = eng.size_order(seq.centers[lo:hi], seq.labels[lo:hi], k)
    scorer.embed_fn = lambda c, l: eng.embed(c, l, k, node_cap=cap, order=order)[0]     # noqa: E731
    pooled = scorer.pooled_all(seq.centers, seq.labels)
    block = scorer.score_rows(pooled)
    err = None
    try:
        eng.check_status()                 # bad labels / broken node_cap promises are errors, not silent NaNs
    except RuntimeError as e:              # ... on EVERY rank: agreed on before the collectives of pr_roc
        err = e
    allpairs.agree_on_error(err, like=block, group=scorer.group)
    f1, auc = scorer.pr_roc(block, seq.poses, p_thresh=p_thresh, n_thresh=n_thresh)
    vals, idx = scorer.loop_closures(block, k=top_k, window=window)
    return {"f1_max": f1, "roc_auc": auc, "closure_scores": vals, "closure_frames": idx, "matrix": block}


def main(argv=None):
    """python -m sg_pr_amd.graph_store config.yml   - all-pairs evaluation of every `eva_batch.sequences` entry:
    packs `<graph_pairs_dir>/<seq>/` once (cached as `<output_path>/<seq>_packed.npz`), writes
    `<seq>_allpairs_F1_max.txt` and `<seq>_loop_closures.npy` (frame, best match, score)."""
    import sys
    from .parser_sg import sgpr_args
    from .sg_net import SGTrainer
    argv = sys.argv[1:] if argv is None else argv
    args = sgpr_args()
    args.load(argv[0] if argv else "./config/config.yml")
    trainer = SGTrainer(args, False)
    trainer.model.eval()
    os.makedirs(args.output_path, exist_ok=True)
    results = {}
    for sequence in args.sequences:
        cache = os.path.join(args.output_path, sequence + "_packed.npz")
        if os.path.exists(cache):
            seq = PackedSequence.load(cache)
        else:
            seq = pack_directory(os.path.join(args.graph_pairs_dir, sequence), int(args.node_num),
                                 trainer.number_of_labels)
            seq.save(cache)
        r = evaluate_all_pairs(trainer.model, seq, p_thresh=float(args.p_thresh))
        with open(os.path.join(args.output_path, sequence + "_allpairs_F1_max.txt"), "w") as f:
            f.write(str(r["f1_max"]))
        m = len(seq)
        np.save(os.path.join(args.output_path, sequence + "_loop_closures.npy"),
                np.stack((np.arange(m), r["closure_frames"][:, 0].cpu().numpy(),
                          r["closure_scores"][:, 0].cpu().numpy()), axis=1))
        print("sequence", sequence, "frames", m, "roc_auc: ", r["roc_auc"], "F1 max score", r["f1_max"])
        results[sequence] = r["f1_max"]
    return results


if __name__ == "__main__":
    main()
