"""Counterpart of the reference's eval_batch.py (eval_batch.py:14-91).

    python -m sg_pr_amd.eval_batch [config.yml]

Per sequence in `eva_batch.sequences`: reads `<pair_list_dir>/<seq>.txt`, scores every
listed pair, and writes the same artefacts as the reference into `output_path`:
`<seq>_gt_db.npy` (float64), `<seq>_DL_db.npy` (float32), `<seq>_DL_F1_max.txt`
`<seq>_DL_roc_curve.png`, `<seq>_DL_pr_curve.png`, the fpr / tpr / thresholds / roc_auc prints, `plt.show()` when
`eva_batch.show` is set.

Unlike the reference (which re-reads, re-pads and re-embeds both graphs of every pair,
utils.py:27-28 / sg_net.py:503-520) each distinct graph is parsed and embedded ONCE on the
GPU; only the NTN + head tail runs per listed pair (sgpr_score_pairs with index lists).
Scores are identical because eval-mode batch elements are independent.
"""
import os
import sys

import numpy as np
import torch

from . import metrics
from .parser_sg import sgpr_args
from .sg_net import SGTrainer
from .utils import load_paires, pose_distance, tab_printer


def score_pair_list(trainer, graph_pairs, group=None):
    """[[path_a, path_b], ...] -> (pred float32 [P], gt float64 [P]); embeds each graph once.

    Under torch.distributed (one process per GPU) the GRAPHS are sharded, then the list (the reference's
    loop eval_batch.py:30-36 walks the list in order, utils.py:61-70 names the files): every rank builds the same table
    of the distinct graphs the list names (first-appearance order: deterministic, no communication), rank r parses and
    embeds graphs [glo_r, ghi_r) of that table - G / world of them, where a contiguous split of a shuffled LIST would
    make every rank embed ~85 % of all graphs -, ONE all_gather_into_tensor brings every rank all pooled vectors
    (G x 128 bytes) and one the poses (the ground truth's x, z), and rank r scores pairs [lo_r, hi_r) of the list against
    the full table; the per-rank `float32[P_r]` / `float64[P_r]` vectors are all-gathered in rank order
    (allpairs.all_gather_varlen).  Every rank returns the full vectors, identical to the single-process ones: a score
    depends on its two graphs only, and the tail kernel is chosen by the length of the whole list, not of a shard (the
    grouped kernel's one launch-wide decision - f16 planes or the exact fp32 path - differs between shards only when a
    pooled vector leaves the f16 range, |pooled| > 6e4: never on real data, ~1e-7 apart when it does).
    `score_pair_list.last_embedded` = the number of graphs THIS rank embedded in the last call (tests)."""
    from . import allpairs
    world, rank = allpairs._world_of(group)
    index, paths = {}, []
    ia = np.empty(len(graph_pairs), dtype=np.int32)
    ib = np.empty(len(graph_pairs), dtype=np.int32)
    for p, (a, b) in enumerate(graph_pairs):
        for path in (a, b):
            if path not in index:
                index[path] = len(paths)
                paths.append(path)
        ia[p], ib[p] = index[a], index[b]
    G = len(paths)
    glo, ghi = allpairs.shard_bounds(G, world, rank)          # the graphs this rank parses and embeds
    lo, hi = allpairs.shard_bounds(len(graph_pairs), world, rank)   # the pairs this rank scores
    n = int(trainer.args.node_num)
    centers = np.empty((ghi - glo, n, 3), dtype=np.float32)
    labels = np.empty((ghi - glo, n), dtype=np.int32)
    poses = np.zeros((ghi - glo, 12), dtype=np.float64)
    model = trainer.model
    dev = getattr(model.engine(), "device", None) if world > 1 else None    # (the CPU tests' stand-in has none)
    dev = dev if isinstance(dev, torch.device) else torch.device("cpu")
    err, pooled = None, None
    score_pair_list.last_embedded = ghi - glo
    try:
        for g in range(glo, ghi):
            c, l, pose = trainer._load_graph(paths[g])
            centers[g - glo], labels[g - glo] = c, l
            poses[g - glo] = np.asarray(pose, dtype=np.float64).reshape(-1)[:12]
        chunk = max(1, int(trainer.args.batch_size)) * 64
        if ghi > glo:
            pooled = torch.cat([model.embed(centers[s:s + chunk], labels[s:s + chunk])[0]
                                for s in range(0, ghi - glo, chunk)])
        model.engine().check_status()      # bad labels / broken node_cap promises are errors, not silent NaNs
    except Exception as e:       # any rank-local failure must reach the agreement below, or the other ranks hang in it
        if world == 1:
            raise
        err = e
    if world > 1:
        # a rank-local failure (missing file, label outside 0..11) is raised on every rank instead of stranding the others
        allpairs.agree_on_error(err, like=torch.empty(0, device=dev) if dev.type == "cuda" else None, group=group)
        width = int(getattr(model.engine(), "pw", 0)) or (int(pooled.shape[1]) if pooled is not None else 32)
        if pooled is None:
            pooled = torch.zeros((0, width), dtype=torch.float32, device=dev)
        pooled = allpairs.all_gather_rows(pooled.to(torch.float32), G, group)
        poses_t = torch.from_numpy(poses)
        poses = allpairs.all_gather_rows(poses_t.to(pooled.device) if pooled.is_cuda else poses_t, G, group).cpu().numpy()
    gt = np.array([trainer.target_from_distance(pose_distance(poses[i], poses[j])) for i, j in zip(ia[lo:hi], ib[lo:hi])],
                  dtype=np.float64)
    pred = torch.empty(0, dtype=torch.float32, device=pooled.device if pooled is not None else dev)
    if hi > lo:
        # which tail kernel: decided on the WHOLE list's length, not on this rank's shard of it
        grouped = len(graph_pairs) >= getattr(model, "GROUPED_MIN_PAIRS", 1 << 62)
        pred = model.score_pooled(pooled, pooled, torch.from_numpy(ia[lo:hi].copy()), torch.from_numpy(ib[lo:hi].copy()),
                                  grouped=grouped).reshape(-1)
    if world == 1:
        return pred.cpu().numpy().reshape(-1), gt
    pred_all, lens = allpairs.all_gather_varlen(pred.to(torch.float32), group)
    gt_t = torch.from_numpy(gt)
    gt_all, _ = allpairs.all_gather_varlen(gt_t.to(pred.device) if pred.is_cuda else gt_t, group)
    assert lens == [b - a for a, b in (allpairs.shard_bounds(len(graph_pairs), world, r) for r in range(world))]
    return pred_all.cpu().numpy().reshape(-1), gt_all.cpu().numpy().reshape(-1)


def plot_curves(args, sequence, fpr, tpr, roc_auc, precision, recall):
    """eval_batch.py:55-82: `<seq>_DL_roc_curve.png`, `<seq>_DL_pr_curve.png`, and `plt.show()` when `args.show`.
    Headless unless `show` is set; skipped with a note when matplotlib is not installed."""
    try:
        import matplotlib
        if not getattr(args, "show", False):
            matplotlib.use("Agg")
        from matplotlib import pyplot as plt
    except ImportError:
        print("matplotlib is not installed: ROC / P-R plots skipped")
        return
    lw = 2
    plt.figure(0)
    plt.plot(fpr, tpr, color='darkorange', lw=lw, label='ROC curve (area = %0.2f)' % roc_auc)
    plt.plot([0, 1], [0, 1], color='navy', lw=lw, linestyle='--')
    plt.xlabel('False Positive Rate')
    plt.ylabel('True Positive Rate')
    plt.title('DL ROC Curve')
    plt.legend(loc="lower right")
    plt.savefig(os.path.join(args.output_path, sequence + "_DL_roc_curve.png"))
    plt.figure(1)
    plt.plot(recall, precision, color='darkorange', lw=lw, label='P-R curve')
    plt.axis([0, 1, 0, 1])
    plt.xlabel('Recall')
    plt.ylabel('Precision')
    plt.title('DL Precision-Recall Curve')
    plt.legend(loc="lower right")
    plt.savefig(os.path.join(args.output_path, sequence + "_DL_pr_curve.png"))
    if getattr(args, "show", False):
        plt.show()
    plt.close(0)
    plt.close(1)


def evaluate_sequence(trainer, sequence, args, plots=True, group=None):
    """One sequence of eval_batch.py:26-91.  Under torch.distributed the pair list is scored in shards
    (score_pair_list); every rank gets the full vectors and computes the same metrics, rank 0 alone writes files."""
    from . import allpairs
    graph_pairs = load_paires(os.path.join(args.pair_list_dir, sequence + ".txt"), args.graph_pairs_dir)
    pred_db, gt_db = score_pair_list(trainer, graph_pairs, group=group)
    assert len(pred_db) == len(gt_db)
    assert np.sum(gt_db) > 0  # gt_db should have positive samples   (eval_batch.py:38)
    if allpairs._world_of(group)[1] != 0:
        return metrics.f1_max(gt_db, pred_db)
    np.save(os.path.join(args.output_path, sequence + "_gt_db.npy"), gt_db)
    np.save(os.path.join(args.output_path, sequence + "_DL_db.npy"), pred_db)
    # ROC (eval_batch.py:48-53)
    fpr, tpr, roc_thresholds = metrics.roc_curve(gt_db, pred_db)
    roc_auc = metrics.auc(fpr, tpr)
    print("fpr: ", fpr)
    print("tpr: ", tpr)
    print("thresholds: ", roc_thresholds)
    print("roc_auc: ", roc_auc)
    precision, recall, pr_thresholds = metrics.precision_recall_curve(gt_db, pred_db)
    if plots:
        plot_curves(args, sequence, fpr, tpr, roc_auc, precision, recall)
    f1_max = metrics.f1_max(gt_db, pred_db)
    print('F1 max score', f1_max)
    with open(os.path.join(args.output_path, sequence + "_DL_F1_max.txt"), "w") as out:
        out.write(str(f1_max))
    return f1_max


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = sgpr_args()
    args.load(argv[0] if argv else './config/config.yml')
    tab_printer(args)
    # launched as `python -m torch.distributed.run --nproc-per-node N -m sg_pr_amd.eval_batch cfg`: one process per GPU
    # (backend nccl = RCCL), every pair list scored in contiguous shards (score_pair_list)
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and not dist.is_initialized():
        local = int(os.environ.get("LOCAL_RANK", "0"))
        args.gpu = local
        args.cuda = str(local)
        torch.cuda.set_device(local)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=os.environ.get("SGPR_BACKEND", "nccl"), device_id=torch.device("cuda", local))
    trainer = SGTrainer(args, False)
    trainer.model.eval()
    os.makedirs(args.output_path, exist_ok=True)
    results = {}
    for sequence in args.sequences:
        print("sequence: ", sequence)
        results[sequence] = evaluate_sequence(trainer, sequence, args)
    if world > 1:
        dist.barrier()
    return results


if __name__ == "__main__":
    main()
