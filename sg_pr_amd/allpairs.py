"""Dense all-pairs evaluation of one sequence, sharded over the GPUs of a node.

The reference's eval loop (eval_batch.py:26-36) walks a pair list and embeds both
graphs of every pair from scratch.  `SG.dgcnn_conv_pass` + `attention` depend on
ONE graph only (sg_net.py:123-127), so for the M x M similarity matrix of a
sequence each graph is embedded once and only the NTN + head tail runs per pair.

Multi-GPU (one process per GPU, torch.distributed; backend "nccl" = RCCL over xGMI):
    rank r embeds graphs [lo_r, hi_r)                      (no communication)
    all_gather of the pooled vectors   M x 32 fp32         (0.58 MB for KITTI-00)
    rank r scores rows [lo_r, hi_r) against all M columns  (no communication)
    row blocks sent point-to-point into rank 0's matrix   (M*M*4 bytes in total, 7 xGMI links in parallel)
Every score depends on its two graphs only, so the matrix is bit-identical for any
world size.

Consumers that never move the matrix: `pr_roc` / `f1_max` (eval_batch.py:48-49, 69-87: the positives'
scores all-gathered, per-rank counts of the negatives summed with one small all_reduce per pass) and `loop_closures` (best matches per query row).
"""
import torch
import torch.distributed as dist


def shard_bounds(total, world_size, rank):
    """Contiguous balanced split of range(total): the first total % world ranks get one extra."""
    base, extra = divmod(int(total), int(world_size))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def _world_of(group=None):
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(group), dist.get_rank(group)
    return 1, 0


def _host_staged(t, group=None):
    """gloo (the CPU tests, and the debugging aid that runs several ranks on one GPU) moves host memory only:
    device tensors are staged through the host for it.  RCCL takes the device buffers directly."""
    return t.is_cuda and dist.get_backend(group) == "gloo"


def all_gather_varlen(t, group=None):
    """Concatenation over ranks (rank order) of 1-D tensors of different lengths, on every rank: one all_gather of the
    lengths + one of the buffers padded to the longest (plain tensor collectives - nothing is pickled, nothing
    crosses the host under RCCL).  -> (concatenated tensor on t's device, per-rank lengths list)."""
    world, _ = _world_of(group)
    t = t.reshape(-1).contiguous()
    if world == 1:
        return t, [int(t.numel())]
    staged = _host_staged(t, group)
    dev = t.device
    n = torch.tensor([t.numel()], dtype=torch.int64, device=torch.device("cpu") if staged else dev)
    lens = [torch.empty_like(n) for _ in range(world)]
    dist.all_gather(lens, n, group=group)
    lens = [int(v.item()) for v in lens]
    cap = max(max(lens), 1)
    buf = t.new_zeros(cap)
    buf[: t.numel()] = t
    if staged:
        buf = buf.cpu()
    parts = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(parts, buf, group=group)
    out = torch.cat([p[:l] for p, l in zip(parts, lens)])
    return (out.to(dev) if staged else out), lens


def all_gather_rows(local, total, group=None):
    """The rows [lo_r, hi_r) = shard_bounds(total, world, r) of a 2-D tensor, one shard per rank -> all `total` rows on every
    rank: ONE collective on equal-sized (padded) shards, straight into one tensor (all_gather_into_tensor: no list of
    per-rank buffers, no concatenation); when the shards are equal - total divisible by the world size - the gathered
    tensor IS the result, otherwise the padding rows of the shorter shards are dropped by one index copy."""
    world, rank = _world_of(group)
    lo, hi = shard_bounds(total, world, rank)
    assert local.shape[0] == hi - lo
    if world == 1:
        return local
    cap = shard_bounds(total, world, 0)[1]               # largest shard
    if hi - lo == cap:
        buf = local.contiguous()
    else:
        buf = local.new_zeros((cap,) + tuple(local.shape[1:]))
        buf[: hi - lo] = local
    staged = _host_staged(buf, group)
    if staged:
        buf = buf.cpu()
    gathered = buf.new_empty((world * cap,) + tuple(buf.shape[1:]))
    dist.all_gather_into_tensor(gathered, buf, group=group)
    if world * cap != total:
        keep = torch.cat([torch.arange(r * cap, r * cap + shard_bounds(total, world, r)[1] - shard_bounds(total, world, r)[0])
                          for r in range(world)]).to(gathered.device)
        gathered = gathered.index_select(0, keep)
    return gathered.to(local.device) if staged else gathered


def agree_on_error(err, like=None, group=None):
    """Rank-local failures ahead of a collective would leave the other ranks blocked in it: every rank contributes
    its exception (or None) to one MAX all_reduce of a flag and ALL ranks raise when any of them failed."""
    world, rank = _world_of(group)
    if world > 1:
        on_dev = like is not None and like.is_cuda and not _host_staged(like, group)
        flag = torch.tensor([1 if err is not None else 0], dtype=torch.int32, device=like.device if on_dev else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=group)
        if int(flag.item()) and err is None:
            err = RuntimeError("another rank of the job failed (rank %d is fine): see its error" % rank)
    if err is not None:
        raise err


class RaggedGraphs:
    """A sequence of graphs as the ragged store of `sgpr_embed_ragged` (include/sgpr.h): centers f32 [S,3] and labels
    i8 [S] wherever the caller keeps them (device tensors for the GPU path), offsets i64 [G+1] on the HOST - slicing by
    graph range, which the sharding below does, needs them there.  Takes the place of the (centers [G,N,3], labels
    [G,N]) pair wherever this module and `SG.embed` accept graphs: pass it as `centers` with `labels=None`."""

    def __init__(self, centers, labels, offsets, node_num):
        self.centers, self.labels = centers, labels
        self.offsets = torch.as_tensor(offsets).to(torch.int64).cpu()
        self.node_num = int(node_num)
        if self.offsets.numel() < 1 or int(self.offsets[-1] - self.offsets[0]) != int(labels.shape[0]) or \
                centers.shape[0] != labels.shape[0]:
            raise ValueError("RaggedGraphs: offsets [G+1] must span the %d stored nodes" % int(labels.shape[0]))

    @classmethod
    def from_padded(cls, centers, labels, device=None, num_labels=None):
        """(centers [G,N,3], labels [G,N], -1 = trailing pad) -> RaggedGraphs (tensors moved to `device` when given).
        num_labels: the model's label count when it is not the shipped 12 (labels outside [0, num_labels) are refused)."""
        from .engine import Engine
        import numpy as np
        c = centers.cpu().numpy() if isinstance(centers, torch.Tensor) else np.asarray(centers)
        l = labels.cpu().numpy() if isinstance(labels, torch.Tensor) else np.asarray(labels)
        rc, rl, off = Engine.to_ragged(c, l) if num_labels is None else Engine.to_ragged(c, l, num_labels)
        tc, tl = torch.from_numpy(rc), torch.from_numpy(rl)
        if device is not None:
            tc, tl = tc.to(device), tl.to(device)
        return cls(tc, tl, off, l.shape[1])

    def __len__(self):
        return self.offsets.numel() - 1

    @property
    def shape(self):                                   # (graphs, slots): what `labels.shape` is for padded arrays
        return (len(self), self.node_num)

    def __getitem__(self, sl):
        if not isinstance(sl, slice) or sl.step not in (None, 1):
            raise TypeError("RaggedGraphs are sliced by contiguous graph ranges")
        lo, hi, _ = sl.indices(len(self))
        hi = max(hi, lo)
        a, b = int(self.offsets[lo]), int(self.offsets[hi])
        base = int(self.offsets[0])
        return RaggedGraphs(self.centers[a - base:b - base], self.labels[a - base:b - base], self.offsets[lo:hi + 1],
                            self.node_num)

    @staticmethod
    def cat(parts):
        """Concatenation in graph order (SequenceSet: this rank's shards of several sequences as one launch)."""
        parts = list(parts)
        if len({p.node_num for p in parts}) != 1:
            raise ValueError("RaggedGraphs.cat: sequences of different node_num")
        offs, at = [torch.zeros(1, dtype=torch.int64)], 0
        for p in parts:
            rel = p.offsets - p.offsets[0]
            offs.append(rel[1:] + at)
            at += int(rel[-1])
        return RaggedGraphs(torch.cat([p.centers for p in parts]), torch.cat([p.labels for p in parts]), torch.cat(offs),
                            parts[0].node_num)


def _graph_count(centers, labels):
    return len(centers) if isinstance(centers, RaggedGraphs) else labels.shape[0]


def _graph_slice(centers, labels, lo, hi):
    return (centers[lo:hi], None) if isinstance(centers, RaggedGraphs) else (centers[lo:hi], labels[lo:hi])


class AllPairsScorer:
    """embed_fn(centers[g0:g1], labels[g0:g1]) -> pooled [g, F];  score_fn(rows, cols) -> [R, M].

    With an `sg_net.SG` model the two callables are its HIP paths; tests inject CPU
    stand-ins to exercise the sharding / collective logic under gloo."""

    def __init__(self, model=None, embed_fn=None, score_fn=None, group=None):
        self._engine = None
        if model is not None:
            embed_fn = lambda c, l: model.embed(c, l)[0]   # noqa: E731
            score_fn = model.score_all_pairs
            self._engine = model.engine()

        if embed_fn is None or score_fn is None:
            raise ValueError("need a model or both embed_fn and score_fn")
        self.embed_fn = embed_fn
        self.score_fn = score_fn
        self.group = group

    def _world(self):
        return _world_of(self.group)

    def _host_staged(self, t):
        return _host_staged(t, self.group)

    def _irecv_ops(self, views, srcs):
        """P2POps receiving into `views` (+ the copies to run after the wait when staged)."""
        ops, after = [], []
        for v, r in zip(views, srcs):
            if self._host_staged(v):
                h = torch.empty(v.shape, dtype=v.dtype)
                ops.append(dist.P2POp(dist.irecv, h, r, self.group))
                after.append((v, h))
            else:
                ops.append(dist.P2POp(dist.irecv, v, r, self.group))
        return ops, after

    def _isend(self, t, dst):
        buf = t.cpu() if self._host_staged(t) else t.contiguous()
        return buf, dist.batch_isend_irecv([dist.P2POp(dist.isend, buf, dst, self.group)])

    def pooled_all(self, centers, labels, local=None):
        """Embed this rank's shard and all_gather: every rank returns pooled [M, F].  `local`: the shard's pooled vectors
        when they were embedded already (SequenceSet embeds the shards of several sequences with one launch)."""
        world, rank = self._world()
        m = _graph_count(centers, labels)
        lo, hi = shard_bounds(m, world, rank)
        if local is None:
            local = self.embed_fn(*_graph_slice(centers, labels, lo, hi))
        return all_gather_rows(local, m, self.group)

    def score_rows(self, pooled):
        """This rank's row block of the score matrix: [hi-lo, M]."""
        world, rank = self._world()
        lo, hi = shard_bounds(pooled.shape[0], world, rank)
        return self.score_fn(pooled[lo:hi].contiguous(), pooled)

    def gather_matrix(self, block, m, dst=0, out=None):
        """Collect the row blocks on rank `dst` -> [M, M] there (written in place into `out` when given:
        every peer's block is received straight into its rows of the matrix - no padding, no extra copy),
        None elsewhere."""
        world, rank = self._world()
        if world == 1:
            return block
        if rank == dst:
            full = out if out is not None else block.new_empty((m, m))
            lo, hi = shard_bounds(m, world, rank)
            spans = [(r,) + shard_bounds(m, world, r) for r in range(world) if r != rank]
            spans = [(r, l, h) for r, l, h in spans if h > l]
            ops, after = self._irecv_ops([full[l:h] for _, l, h in spans], [r for r, _, _ in spans])
            reqs = dist.batch_isend_irecv(ops) if ops else []
            full[lo:hi].copy_(block)
            for q in reqs:
                q.wait()
            for v, h in after:
                v.copy_(h)
            return full
        if block.shape[0] > 0:
            _keep, reqs = self._isend(block, dst)
            for q in reqs:
                q.wait()
        return None

    def pr_roc(self, block, poses, p_thresh=3.0, n_thresh=20.0, fns=None, want_auc=True):
        """(F1-max of eval_batch.py:85-87, ROC area of eval_batch.py:48-49) over this job's matrix WITHOUT gathering it:
        `block` is this rank's row block (score_rows), ground truth comes from the poses ([M,12] KITTI rows or [M,2]
        x/z).  Every rank returns the same values: the scores of the positive pairs are all-gathered (they are few),
        the per-threshold counts of the negatives all-reduced (sg_pr_amd/metrics.py:pr_roc_from_counts).
        fns(block, row0, pose_xz) -> (positive scores of the block, count_fn) overrides the engine (CPU tests)."""
        import numpy as np
        from . import metrics
        world, rank = self._world()
        xz = pose_xz(poses)
        lo, _ = shard_bounds(xz.shape[0], world, rank)
        # a rank-local failure (negative / NaN scores in this rank's block, a bad label seen by the embed launch) must
        # not strand the other ranks in the collectives below: the error is agreed on first, every rank raises
        err, pos, local_count = None, None, None
        try:
            if fns is None:
                pos, local_count = metrics._device_fns(self._engine, block, xz.to(block.device), p_thresh, n_thresh, None,
                                                       lo, distinct=False, to_host=False)
            else:
                pos, local_count = fns(block, lo, xz)
        except (ValueError, RuntimeError) as e:
            err = e
        agree_on_error(err, like=block, group=self.group)
        if world > 1:
            # the positives' scores of all row blocks: a padded tensor all_gather (device buffers under RCCL)
            pt = pos if isinstance(pos, torch.Tensor) else torch.from_numpy(np.asarray(pos, dtype=np.float32))
            if block.is_cuda and not pt.is_cuda:
                pt = pt.to(block.device)
            pos = all_gather_varlen(pt.to(torch.float32), self.group)[0]
        if isinstance(pos, torch.Tensor):
            pos = pos.cpu().numpy()

        def count_fn(thresholds, ranking):
            bad = 0
            try:
                counts, rank_sum = local_count(thresholds, ranking)
            except ValueError as e:                              # agreed on below, together with the counts
                counts, rank_sum, bad, err_c = np.zeros(len(thresholds) + 1, dtype=np.int64), 0, 1, e
            if world > 1:
                t = torch.from_numpy(np.concatenate((np.asarray(counts, dtype=np.int64), [rank_sum or 0, bad])))
                if block.is_cuda and not self._host_staged(block):
                    t = t.to(block.device)
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
                t = t.cpu().numpy()
                counts, rank_sum = t[:-2], (int(t[-2]) if ranking is not None else None)
                if int(t[-1]):
                    raise (err_c if bad else RuntimeError("another rank saw negative or NaN scores in its row block"))
            elif bad:
                raise err_c
            return counts, rank_sum
        f1, auc, _ = metrics.pr_roc_from_counts(pos, count_fn, want_auc=want_auc)
        return f1, auc

    def f1_max(self, block, poses, p_thresh=3.0, n_thresh=20.0, fns=None):
        """F1-max of pr_roc alone (no ranking of the negatives)."""
        return self.pr_roc(block, poses, p_thresh, n_thresh, fns=fns, want_auc=False)[0]

    def loop_closures(self, block, k=1, window=50):
        """Per query row of this rank's block: the k best-scoring frames at least `window` frames away
        -> (scores [R,k], frame indices [R,k]) on the device."""
        world, rank = self._world()
        lo, _ = shard_bounds(block.shape[1], world, rank)
        return self._engine.topk_rows(block, k=k, row0=lo, window=window)

    def run(self, centers, labels, gather=True, out=None, chunks=4, local_pooled=None):
        """Whole job: returns the [M, M] matrix on rank 0 (row block elsewhere / if gather=False).
        `out` (rank 0): preallocated [M, M] buffer that receives the matrix.
        chunks > 1 (multi-rank gather): every rank scores its row block in `chunks` pieces and ships each piece to
        rank 0 as soon as it is computed, so the xGMI transfer of piece i overlaps the scoring of piece i+1; rank 0
        posts all its receives up front (one group per piece index, one receive per peer in it), straight into the
        rows of the matrix.  chunks = 1 is the plain form: score the block, then one gather (gather_matrix).
        local_pooled: see pooled_all."""
        pooled = self.pooled_all(centers, labels, local=local_pooled)
        world, rank = self._world()
        if not gather or world == 1 or chunks <= 1:
            block = self.score_rows(pooled)
            if not gather:
                return block
            full = self.gather_matrix(block, pooled.shape[0], out=out)
            return full if full is not None else block
        m = pooled.shape[0]
        dst = 0

        def pieces(lo, hi):
            n = max(1, min(chunks, hi - lo))
            return [(lo + (hi - lo) * i // n, lo + (hi - lo) * (i + 1) // n) for i in range(n)] if hi > lo else []

        lo, hi = shard_bounds(m, world, rank)
        if rank == dst:
            full = out if out is not None else pooled.new_empty((m, m))
            # one receive group PER PIECE INDEX, holding one receive from every peer (the shape of a plain gather:
            # a root group of world-1 receives against one single-send group on each peer); peer r's c-th send group
            # meets rank 0's c-th receive group, in order, so both sides agree on the sequence of groups per pair
            peer_pieces = {r: pieces(*shard_bounds(m, world, r)) for r in range(world) if r != rank}
            reqs, after = [], []
            for c in range(chunks):
                srcs = [r for r, pc in peer_pieces.items() if c < len(pc)]
                ops, aft = self._irecv_ops([full[peer_pieces[r][c][0]:peer_pieces[r][c][1]] for r in srcs], srcs)
                after += aft
                if ops:
                    reqs += dist.batch_isend_irecv(ops)
            for a, b in pieces(lo, hi):                      # own rows: scored straight into the matrix
                if self._engine is not None:
                    self.score_fn(pooled[a:b].contiguous(), pooled, out=full[a:b])
                else:
                    full[a:b].copy_(self.score_fn(pooled[a:b].contiguous(), pooled))
            for q in reqs:
                q.wait()
            for v, h in after:
                v.copy_(h)
            return full
        reqs, keep, sent = [], [], []
        for a, b in pieces(lo, hi):
            blk = self.score_fn(pooled[a:b].contiguous(), pooled)
            keep.append(blk)
            buf, rq = self._isend(blk, dst)
            sent.append(buf)                                 # the buffer must outlive the asynchronous send
            reqs += rq
        for q in reqs:
            q.wait()
        return torch.cat(keep, dim=0) if keep else pooled.new_empty((0, m))


class SequenceSet:
    """Several independent sequences evaluated as one job - the reference's loop over `eva_batch.sequences`
    (eval_batch.py:26-36).  Graphs are independent, so this rank's shards of ALL sequences are embedded by ONE launch
    (a KITTI sequence sharded over 8 ranks leaves 140..580 graphs per rank and sequence: less than one round of the
    1024 workgroup slots of a GPU); the matrices are then scored with one pair of launches when nothing has to be shipped
    between them (one process, or the row blocks stay sharded), else scored and gathered sequence by sequence.  Results are
    bit-identical to per-sequence runs."""

    def __init__(self, scorer, sequences, batch_tails=True):
        """sequences: list of (centers [M,N,3], labels [M,N]) tensors, or of (RaggedGraphs, None) - one kind per set."""
        self.scorer = scorer
        self.batch_tails = batch_tails
        self.sequences = list(sequences)
        world, rank = scorer._world()
        self.bounds = [shard_bounds(_graph_count(c, l), world, rank) for c, l in self.sequences]
        if self.sequences and isinstance(self.sequences[0][0], RaggedGraphs):
            self.centers = RaggedGraphs.cat([c[lo:hi] for (c, _), (lo, hi) in zip(self.sequences, self.bounds)])
            self.labels = None
        else:
            self.centers = torch.cat([c[lo:hi] for (c, _), (lo, hi) in zip(self.sequences, self.bounds)], dim=0)
            self.labels = torch.cat([l[lo:hi] for (_, l), (lo, hi) in zip(self.sequences, self.bounds)], dim=0)

    def run(self, embed_fn=None, gather=True, outs=None, chunks=4):
        """embed_fn(centers, labels) -> pooled of the concatenated shards (default: the scorer's).  Returns one result
        per sequence, each what AllPairsScorer.run returns."""
        pooled = (embed_fn or self.scorer.embed_fn)(self.centers, self.labels)
        world, _ = self.scorer._world()
        locals_, at = [], 0
        for lo, hi in self.bounds:
            locals_.append(pooled[at:at + hi - lo])
            at += hi - lo
        eng = self.scorer._engine
        if eng is not None and self.batch_tails and (world == 1 or not gather):
            # nothing to ship between the tails: all row blocks with one pair of launches (sgpr_score_all_pairs_multi)
            jobs = []
            for i, ((c, l), local) in enumerate(zip(self.sequences, locals_)):
                full = self.scorer.pooled_all(c, l, local=local)
                lo, hi = self.bounds[i]
                jobs.append((full[lo:hi].contiguous(), full, outs[i] if (outs is not None and world == 1) else None))
            return eng.score_all_pairs_multi(jobs)
        return [self.scorer.run(c, l, gather=gather, out=outs[i] if outs is not None else None, chunks=chunks,
                                local_pooled=local) for i, ((c, l), local) in enumerate(zip(self.sequences, locals_))]


def pose_xz(poses):
    """[M,12] KITTI pose rows (or [M,2]) -> float64 [M,2] planar position (x, z): the two numbers utils.py:36 uses."""
    p = torch.as_tensor(poses)
    if p.shape[1] != 2:
        p = torch.stack((p[:, 3], p[:, 11]), dim=1)
    return p.to(torch.float64).contiguous()


def pose_distance_matrix(poses):
    """Planar pose distances of a sequence (utils.py:36 for every pair): [M, M] float64 tensor."""
    p = torch.as_tensor(poses, dtype=torch.float64)
    xz = torch.stack((p[:, 3], p[:, 11]), dim=1)
    return torch.cdist(xz, xz)


def ground_truth_mask(dist_matrix, p_thresh):
    """(gt, valid): gt = 1 where d <= p_thresh, 0 where d >= 20; pairs in between are the
    ones the reference refuses (`exit(-1)`, sg_net.py:302-309) and are masked out."""
    gt = (dist_matrix <= p_thresh).to(torch.float64)
    valid = (dist_matrix <= p_thresh) | (dist_matrix >= 20)
    return gt, valid
