#!/usr/bin/env python3
"""bench.py - graph-pairs/sec of the SG_PR hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload kitti00|kitti5seq|pairlist|pairs128|stress]

`--gpus N` with N > 1 and no torch.distributed environment: this script re-launches itself as
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...` (one rank per GPU, backend
nccl = RCCL) and rank 0 prints the line; under a launcher (RANK / WORLD_SIZE set, the driver's form) it joins as a
rank.  Asking for more ranks than visible GPUs is an error, never a silent 1-GPU run
(SGPR_BENCH_BACKEND=gloo is the debugging aid that lets several ranks share one GPU).

Default workload (`kitti00`): the KITTI-00 all-pairs similarity matrix - M = 4541 graphs (KITTI-00 frame count),
node_num = 100, K = 10 -> 4541^2 = 20 620 681 ordered pairs per step.  KITTI graphs are not in the reference tree
(README.md:54), so the sequence is the seeded KITTI-like synthetic generator of sg_pr_amd.synth (`"data":
"synthetic"`); weights are the shipped checkpoint tests/golden/model.pth.
`kitti5seq` (BASELINE config 4): sequences 00+02+05+06+08 (M = 4541, 4661, 2761, 1101, 4071 -> 67.75 M pairs),
evaluated one after another per step like the reference's loop over `eva_batch.sequences` (eval_batch.py:26-36); the
graphs of all five are embedded by one launch and, where nothing has to be gathered in between, their matrices scored
by one pair of launches (`allpairs.SequenceSet`).

One step = one pass of the hot path over the whole job with inputs already resident in HBM: embed every graph of
this rank's shard (fused kNN/EdgeConv/attention kernel), exchange the pooled vectors, score this rank's row block of
the matrix (NTN + head), gather the matrix on rank 0.  N > 1: the M graphs / M rows are sharded across ranks; the
matrix is fixed, so `scaling` is "strong".  `end_to_end` adds what `value` excludes by contract: the H2D copy of the
packed graphs and either the D2H copy of the matrix or its device-side consumer (F1-max from threshold counts).

`pairlist`: the reference's OWN loop shape (eval_batch.py:30-36 walks a pair list, utils.py:61-70) on its own evaluation
lists for KITTI 02 / 05 / 06 / 08 (tests/golden/pair_lists_3_20.npz: the index pairs of the reference's
data_process/pair_list/pair_list_3_20_*.npy - 184 133 listed pairs over 12 584 graphs, ~15 per row graph), in a seeded
shuffled order like the reference's files; graphs = synthetic KITTI-like sequences of those lengths.  One step = every
graph embedded once + the listed pairs scored by sgpr_score_pair_list (grouped by row graph, matrix cores).

`value` keeps inputs and outputs resident in HBM (`value_definition` says so in the line); `roofline.wide_range` repeats
the step after the timed region with the embed's wide-range instance forced (three bf16 planes = the reference's 24-bit
operand width instead of the default two f16 planes) and prices the all-fp32 variant from a sample of the exact tail.

Other workloads (parity-test shapes, not the headline): `pairs128` (config 2: 128 pairs, N=64, k=10, faithful
per-pair forward) and `stress` (config 5: 1024 pairs, N=256, k=20).
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8 TB/s HBM3E
FP32_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: fp32 vector peak (= fp32 matrix peak)
SHADER_CLOCK_HZ = 2.4e9      # MI355X_MICROARCH.md: peak engine clock (used only when the --pmc passes carry no measured one)
KITTI_FRAMES = {"00": 4541, "02": 4661, "05": 2761, "06": 1101, "08": 4071}


def embed_flops_per_graph(n, k):
    """Algorithmic FLOPs of one graph through the embed kernel, factored EdgeConv formulation
    (SURVEY.md 8d / DESIGN.md): kNN Gram 2*N^2*sum(Cin), per-node GEMMs 2*N*sum(Cin*2*Cout),
    gather-max N*k*sum(Cout), conv_end 2*N*64*32, attention 2*N*(32*2) + 2*32*32."""
    cin = [3, 64, 64, 12, 64, 64]
    cout = [64, 64, 32, 64, 64, 32]
    gram = 2 * n * n * sum(cin)
    gemm = 2 * n * sum(ci * 2 * co for ci, co in zip(cin, cout))
    gmax = n * k * sum(cout)
    end = 2 * n * 64 * 32
    att = 2 * n * 64 + 2 * 32 * 32
    return gram + gemm + gmax + end + att


def embed_bytes_per_graph(n):
    """Algorithmic HBM bytes of one graph through the embed kernel: packed input
    (3 fp32 centre + 1 int32 label per slot) + the pooled vector out."""
    return 16 * n + 32 * 4


def source_hash():
    """sha256 over the HIP sources: ties a committed PMC profile to the code it was taken from."""
    h = hashlib.sha256()
    d = os.path.join(REPO, "sg_pr_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".hpp")):
            h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--prewarm", type=float, default=1.5,
                    help="seconds of untimed steps before the warmup (GPU clock ramp), 0 disables")
    ap.add_argument("--workload", default="kitti00", choices=["kitti00", "kitti5seq", "pairlist", "pairs128", "stress"])
    ap.add_argument("--graphs", type=int, default=4541, help="M for the kitti00 workload")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU time of the baseline sample")
    ap.add_argument("--cpu-pairs", type=int, default=0,
                    help="size of the CPU baseline's sample in pairs (0: bounded by --cpu-seconds; SURVEY.md 8d's calibration "
                         "sample is 100000 - minutes of CPU time, profiles/rNN_cpu_baseline_100k.json)")
    ap.add_argument("--embed-mode", default="ordered", choices=["ordered", "capped"],
                    help="launch the graphs largest-first (default) or in storage order")
    ap.add_argument("--no-gather", action="store_true", help="leave the score matrix sharded (skip the gather)")
    ap.add_argument("--chunks", type=int, default=4, help="pieces per rank of the overlapped gather (1 = plain gather)")
    ap.add_argument("--no-end-to-end", action="store_true", help="skip the transfer-inclusive measurements")
    ap.add_argument("--per-sequence-tails", action="store_true",
                    help="kitti5seq: one prep + tail launch per sequence instead of one pair for all matrices")
    ap.add_argument("--per-sequence-embed", action="store_true",
                    help="kitti5seq: one embed launch per sequence instead of one for the shards of all sequences")
    ap.add_argument("--data-dir", default=os.environ.get("SG_PR_DATA_DIR", ""),
                    help="kitti00: a directory of real graph JSONs (e.g. $SG_PR_DATA/graphs_sk/00) packed once through "
                         "sg_pr_amd.graph_store.pack_directory instead of the synthetic sequence; the line then says "
                         "\"data\": \"real\" and M is the number of graphs found")
    ap.add_argument("--kernel-reps", type=int, default=32,
                    help="launches per kernel of the untimed duration pass that follows the timed region (HIP events "
                         "around that many back-to-back embed calls, then tail calls; the timed region carries no events)")
    ap.add_argument("--no-wide-range", action="store_true",
                    help="skip the untimed pass at the reference's operand width (wide-range embed instance forced)")
    ap.add_argument("--d2h-pieces", type=int, default=4,
                    help="end_to_end.d2h: row blocks whose device-to-host copy overlaps the scoring of the next one")
    return ap.parse_args()


def respawn_under_launcher(a):
    """`python bench.py --gpus N` without a launcher: start N ranks of this script."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    return subprocess.call(cmd, env=env)


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(respawn_under_launcher(a))

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(a.gpus, 1):
        sys.exit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (a.gpus, world))
    ndev = torch.cuda.device_count()
    backend = os.environ.get("SGPR_BENCH_BACKEND", "nccl")   # gloo: debugging aid, several ranks on ONE GPU
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl" and ndev < world:
            sys.exit("bench.py: --gpus %d but only %d GPU(s) are visible (RCCL needs one GPU per rank)" % (world, ndev))
        local_rank %= max(ndev, 1)
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)
    else:
        local_rank = 0
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local_rank)

    from sg_pr_amd import sg_net, synth, allpairs
    from sg_pr_amd.parser_sg import sgpr_args

    args = sgpr_args()
    args.model = os.path.join(REPO, "tests", "golden", "model.pth")
    args.gpu = dev.index
    allpairs_job = a.workload in ("kitti00", "kitti5seq")
    data_kind = "synthetic"
    if a.workload == "kitti00":
        n, k = 100, 10
        seqs = [("00", a.graphs)]
        wl_name = "KITTI-00-sized all-pairs matrix (synthetic KITTI-like graphs), M=%d, node_num=100, K=10" % a.graphs
    elif a.workload == "kitti5seq":
        n, k = 100, 10
        seqs = list(KITTI_FRAMES.items())
        wl_name = ("config 4: KITTI 00+02+05+06+08-sized all-pairs matrices back to back (synthetic KITTI-like graphs), "
                   "M=" + "/".join(str(m) for _, m in seqs) + ", node_num=100, K=10")
    elif a.workload == "pairlist":
        n, k = 100, 10
        wl_name = ("the reference's evaluation pair lists for KITTI 02+05+06+08 (pair_list_3_20_*.npy index pairs, seeded "
                   "shuffle) over synthetic KITTI-like graphs: embed every graph once + sgpr_score_pair_list, node_num=100, K=10")
    elif a.workload == "pairs128":
        n, k = 64, 10
        wl_name = "config 2: 128 synthetic pairs per GPU, node_num=64, K=10, faithful per-pair forward"
    else:
        n, k = 256, 20
        wl_name = "config 5: 1024 synthetic pairs per GPU, node_num=256, K=20, faithful per-pair forward"
    args.node_num, args.K = n, k
    import contextlib
    with contextlib.redirect_stdout(sys.stderr):
        trainer = sg_net.SGTrainer(args, False)
    model = trainer.model
    eng = model.engine()
    eng_cus = torch.cuda.get_device_properties(dev).multi_processor_count

    # Kernel durations for the roofline objects come from an UNTIMED pass after the timed region (kernel_ms below): an
    # event record between two launches idles the stream for ~5 us, so the timed region carries none.
    calls = {"embed": 0, "tail": 0}      # engine calls per kind during the timed region (launches per step)
    dur_calls = {}                       # kind -> zero-argument callable that enqueues ONE such call (duration pass)
    graphs_per_step = 0             # graphs this rank embeds per step
    n_eff_all = []                  # processed slots of those graphs (algorithmic FLOPs)
    host_inputs = []                # (centers, labels) numpy, for the transfer-inclusive runs and the CPU baseline
    host_pairs, plan, plan_ms = None, None, None
    prep_ms = 0.0                   # one-off cost of node_cap / launch order on the DEVICE (sgpr_size_order on the resident
                                    # arrays + a 4-byte read-back): data-set properties

    def counted(kind, fn):
        def wrapped(*x, **kw):
            calls[kind] += 1
            return fn(*x, **kw)
        return wrapped

    if allpairs_job:
        jobs = []
        units = 0
        if a.workload == "kitti00" and a.data_dir:
            # real graphs (README.md:54, 92-97: <graph_pairs_dir>/<seq>/*.json), parsed and packed once
            from sg_pr_amd import graph_store
            real = graph_store.pack_directory(a.data_dir, n)
            if len(real) < 2:
                sys.exit("bench.py: --data-dir %s holds %d graph JSONs" % (a.data_dir, len(real)))
            seqs = [(os.path.basename(os.path.normpath(a.data_dir)) or "real", len(real))]
            data_kind = "real"
            wl_name = "all-pairs matrix of the %d graphs in %s, node_num=100, K=10" % (len(real), a.data_dir)
        for si, (name, m) in enumerate(seqs):
            if data_kind == "real":
                centers, labels, poses = real.centers, real.labels, real.poses
            else:
                centers, labels, _, poses = synth.kitti_like_sequence(num_graphs=m, node_num=n, seed=si)
            host_inputs.append((centers, labels, poses))
            lo, hi = allpairs.shard_bounds(m, world, rank)
            d_c, d_l = torch.from_numpy(centers).to(dev), torch.from_numpy(labels).to(dev)
            eng.size_order(d_c[lo:hi], d_l[lo:hi], k)        # (first use: module load)
            torch.cuda.synchronize()
            t_h = time.perf_counter()
            order, node_cap = eng.size_order(d_c[lo:hi], d_l[lo:hi], k)   # on the device, from the resident arrays
            torch.cuda.synchronize()
            prep_ms += (time.perf_counter() - t_h) * 1e3
            if a.embed_mode != "ordered":
                order = None
            scorer = allpairs.AllPairsScorer(model=model)
            scorer.embed_fn = counted("embed", lambda c, l, cap=node_cap, o=order: eng.embed(c, l, k, node_cap=cap, order=o)[0])
            scorer.score_fn = counted("tail", model.score_all_pairs)
            full_out = (torch.empty(m, m, dtype=torch.float32, device=dev)
                        if (world > 1 and rank == 0 and not a.no_gather) else None)
            jobs.append({"name": name, "m": m, "scorer": scorer, "out": full_out, "node_cap": node_cap,
                         "d_centers": d_c, "d_labels": d_l})
            units += m * m
            graphs_per_step += hi - lo
            n_eff_all.append(synth.effective_nodes(centers[lo:hi], labels[lo:hi], k))
        node_cap_report = max(j["node_cap"] for j in jobs)
        seqset = None
        if len(jobs) > 1 and not a.per_sequence_embed:
            # several sequences: this rank's shards of all of them are embedded by ONE launch (allpairs.SequenceSet)
            seqset = allpairs.SequenceSet(jobs[0]["scorer"], [(j["d_centers"], j["d_labels"]) for j in jobs],
                                          batch_tails=not a.per_sequence_tails)
            eng.score_all_pairs_multi = counted("tail", eng.score_all_pairs_multi)   # (the batched tails' launches)
            torch.cuda.synchronize()
            t_h = time.perf_counter()
            set_order, node_cap_report = eng.size_order(seqset.centers, seqset.labels, k)
            torch.cuda.synchronize()
            prep_ms = (time.perf_counter() - t_h) * 1e3          # (the set's order replaces the per-sequence ones)
            if a.embed_mode != "ordered":
                set_order = None
            set_embed = counted("embed", lambda c, l: eng.embed(c, l, k, node_cap=node_cap_report, order=set_order)[0])

        def step(gather=not a.no_gather):
            if seqset is not None:
                return seqset.run(embed_fn=set_embed, gather=gather, outs=[j["out"] for j in jobs], chunks=a.chunks)[-1]
            out = None
            for j in jobs:
                out = j["scorer"].run(j["d_centers"], j["d_labels"], gather=gather, out=j["out"], chunks=a.chunks)
            return out

        # one embed call / one tail call exactly as the step issues them (this rank's shard), for the duration pass
        if seqset is not None:
            dur_calls["embed"] = lambda: set_embed(seqset.centers, seqset.labels)
        else:
            j0 = jobs[0]
            lo0, hi0 = allpairs.shard_bounds(j0["m"], world, rank)
            dur_calls["embed"] = lambda: j0["scorer"].embed_fn(j0["d_centers"][lo0:hi0], j0["d_labels"][lo0:hi0])
            _c0, _l0 = j0["d_centers"][lo0:hi0].contiguous(), j0["d_labels"][lo0:hi0].contiguous()
            _ord0 = eng.size_order_device(_c0, _l0, None, n, k)[0]
            dur_calls["embed_plain"] = lambda: eng.embed(_c0, _l0, k, auto_order=False)[0]    # the plain C-ABI call
            dur_calls["embed_engine_default"] = lambda: eng.embed(_c0, _l0, k)[0]           # + the binding's cached device order
            dur_calls["embed_order_only"] = lambda: eng.embed(_c0, _l0, k, order=_ord0)[0]
            dur_calls["size_order"] = lambda: eng.size_order_device(_c0, _l0, None, n, k)[0]
        if seqset is not None and not a.per_sequence_tails and world == 1:
            _pl = [eng.embed(j["d_centers"], j["d_labels"], k)[0] for j in jobs]
            _outs = [torch.empty(j["m"], j["m"], dtype=torch.float32, device=dev) for j in jobs]
            dur_calls["tail"] = lambda: eng.score_all_pairs_multi([(p_, p_, o_) for p_, o_ in zip(_pl, _outs)])
        else:
            j0 = jobs[0]
            lo0, hi0 = allpairs.shard_bounds(j0["m"], world, rank)
            _p0 = eng.embed(j0["d_centers"], j0["d_labels"], k)[0]
            _o0 = torch.empty(hi0 - lo0, j0["m"], dtype=torch.float32, device=dev)
            _r0 = _p0[lo0:hi0].contiguous()
            dur_calls["tail"] = lambda: j0["scorer"].score_fn(_r0, _p0, out=_o0)
    elif a.workload == "pairlist":
        fx = np.load(os.path.join(REPO, "tests", "golden", "pair_lists_3_20.npz"))
        seqs, parts_c, parts_l, parts_i, parts_j, base = [], [], [], [], [], 0
        for si, name in enumerate(("02", "05", "06", "08")):
            ij = fx["seq_" + name].astype(np.int64)
            m = int(ij.max()) + 1
            c, l, _, _ = synth.kitti_like_sequence(num_graphs=m, node_num=n, seed=10 + si)
            parts_c.append(c)
            parts_l.append(l)
            parts_i.append(ij[:, 0] + base)
            parts_j.append(ij[:, 1] + base)
            seqs.append((name, m))
            base += m
        centers, labels = np.concatenate(parts_c), np.concatenate(parts_l)
        li, lj = np.concatenate(parts_i), np.concatenate(parts_j)
        perm = np.random.default_rng(2024).permutation(li.size)       # the reference's files are in shuffled order
        li, lj = li[perm].astype(np.int32), lj[perm].astype(np.int32)
        # pair-list mode (sg_pr_amd/eval_batch.py:score_pair_list): the GRAPHS are sharded - rank r embeds graphs
        # [glo_r, ghi_r) of the table, one all_gather_into_tensor of pooled (M x 128 bytes) - then the list: rank r scores
        # pairs [lo_r, hi_r) against the full table (a contiguous split of the shuffled LIST alone would make every rank
        # embed ~85 % of all graphs)
        m = centers.shape[0]
        lo, hi = allpairs.shard_bounds(li.size, world, rank)
        glo, ghi = allpairs.shard_bounds(m, world, rank)
        li, lj = li[lo:hi], lj[lo:hi]
        host_inputs.append((centers, labels, None))
        host_pairs = (li, lj)
        units = int(sum(fx["seq_" + nm].shape[0] for nm, _ in seqs))
        used = np.arange(glo, ghi)                                    # the graphs THIS rank embeds
        remap = np.arange(m, dtype=np.int32)                          # (pooled holds the whole table after the all-gather)
        d_centers, d_labels = torch.from_numpy(centers[used]).to(dev), torch.from_numpy(labels[used]).to(dev)
        eng.size_order(d_centers, d_labels, k)
        torch.cuda.synchronize()
        t_h = time.perf_counter()
        order, node_cap_report = eng.size_order(d_centers, d_labels, k)
        torch.cuda.synchronize()
        prep_ms += (time.perf_counter() - t_h) * 1e3
        if a.embed_mode != "ordered":
            order = None
        t_h = time.perf_counter()
        plan = eng.pair_plan(remap[li], remap[lj], m, m)
        plan_ms = (time.perf_counter() - t_h) * 1e3
        graphs_per_step = int(used.size)
        n_eff_all.append(synth.effective_nodes(centers[used], labels[used], k))
        embed_local = counted("embed", lambda: eng.embed(d_centers, d_labels, k, node_cap=node_cap_report, order=order)[0])
        embed_t = (lambda: allpairs.all_gather_rows(embed_local(), m)) if world > 1 else embed_local
        _pooled0 = embed_t()
        _score0 = torch.empty(plan.P, dtype=torch.float32, device=dev)
        tail_t = counted("tail", lambda pooled: eng.score_pair_list(pooled, pooled, plan, out=_score0))
        dur_calls["embed"] = embed_local
        dur_calls["tail"] = lambda: tail_t(_pooled0)
        _d_i, _d_j = torch.from_numpy(remap[li]).to(dev), torch.from_numpy(remap[lj]).to(dev)
        dur_calls["tail_one_wave_per_pair"] = lambda: eng.score_pairs(_pooled0, _pooled0, _d_i, _d_j, out=_score0)
        if world == 1:
            # sequence 02 on its own (71 226 listed pairs over 4 661 graphs: the list VERDICT r3 sized its estimate on)
            _m02 = seqs[0][1]
            _ij02 = fx["seq_02"].astype(np.int32)
            _c02, _l02 = torch.from_numpy(parts_c[0]).to(dev), torch.from_numpy(parts_l[0]).to(dev)
            _o02, _cap02 = eng.size_order(parts_c[0], parts_l[0], k)
            _plan02 = eng.pair_plan(_ij02[:, 0], _ij02[:, 1], _m02, _m02)
            _p02 = eng.embed(_c02, _l02, k, node_cap=_cap02, order=_o02)[0]
            _s02 = torch.empty(_plan02.P, dtype=torch.float32, device=dev)
            _i02, _j02 = torch.from_numpy(_ij02[:, 0].copy()).to(dev), torch.from_numpy(_ij02[:, 1].copy()).to(dev)
            dur_calls["seq02_embed"] = lambda: eng.embed(_c02, _l02, k, node_cap=_cap02, order=_o02)[0]
            dur_calls["seq02_tail"] = lambda: eng.score_pair_list(_p02, _p02, _plan02, out=_s02)
            dur_calls["seq02_tail_one_wave_per_pair"] = lambda: eng.score_pairs(_p02, _p02, _i02, _j02, out=_s02)

        def step(gather=True):
            sc = tail_t(embed_t())
            if world > 1 and gather:
                sc = allpairs.all_gather_varlen(sc, None)[0]
            return sc
    else:
        centers, labels, _ = (synth.config2_pairs(seed=0) if a.workload == "pairs128" else synth.config5_pairs(seed=0))
        host_inputs.append((centers, labels, None))
        m = centers.shape[0]
        b = m // 2
        units = b * world
        d_centers, d_labels = torch.from_numpy(centers).to(dev), torch.from_numpy(labels).to(dev)
        cc = torch.cat((d_centers[0::2], d_centers[1::2])).contiguous()
        ll = torch.cat((d_labels[0::2], d_labels[1::2])).contiguous()
        eng.size_order(cc, ll, k)
        torch.cuda.synchronize()
        t_h = time.perf_counter()
        order, node_cap_report = eng.size_order(cc, ll, k)
        torch.cuda.synchronize()
        prep_ms += (time.perf_counter() - t_h) * 1e3
        if a.embed_mode != "ordered":
            order = None
        graphs_per_step = m
        n_eff_all.append(synth.effective_nodes(centers, labels, k))
        embed_t = counted("embed", lambda: eng.embed(cc, ll, k, node_cap=node_cap_report, order=order)[0])
        dur_calls["embed"] = embed_t

        def step(gather=True):
            pooled = embed_t()
            return eng.score_pairs(pooled[:b], pooled[b:])

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def timed_steps(count, fn):
        """Barrier + synchronize on both sides, MAX over ranks."""
        sync_all()
        t0 = time.perf_counter()
        for _ in range(count):
            out = fn()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        del out
        return dt

    # The GPU leaves its idle power state only after some tens of milliseconds of sustained work: a fresh process that
    # times 15 ms of kernels right away measures the clock ramp (observed: 7x slower kernels).  Run the same step,
    # untimed, until the device has been busy for --prewarm seconds; the W warmup steps of the contract follow.
    t_pre = time.perf_counter()
    while a.prewarm > 0:
        for _ in range(20 if a.workload == "kitti5seq" else 50):
            step()
        torch.cuda.synchronize()
        more = torch.tensor([1 if time.perf_counter() - t_pre < a.prewarm else 0], device=dev)
        if world > 1:
            dist.broadcast(more, src=0)          # every rank runs the same number of (collective) steps
        if int(more.item()) == 0:
            break
    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    calls["embed"] = calls["tail"] = 0
    dt = timed_steps(a.steps, step)
    launches_per_step = max(1, round(calls["embed"] / max(a.steps, 1)))
    tail_calls_per_step = calls["tail"] / max(a.steps, 1)

    def kernel_ms(fn, reps):
        """Average duration of one engine call from HIP events around `reps` back-to-back calls on the launch stream
        (torch's current stream is the stream the engine launches on): no event between the calls, no idle gaps."""
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    launches_timed = max(16, a.kernel_reps)
    embed_ms = kernel_ms(dur_calls["embed"], launches_timed)
    # the same graphs through the calls that need no data-set property: no node_cap promise (the 64-row layout + hand-over of
    # larger graphs), with and without the device-made launch order
    embed_forms = None
    if world == 1 and "embed_plain" in dur_calls:
        embed_forms = {"ordered_with_node_cap": embed_ms,
                       "plain_no_promise": kernel_ms(dur_calls["embed_plain"], launches_timed),
                       "engine_embed_default": kernel_ms(dur_calls["embed_engine_default"], launches_timed),
                       "ordered_no_promise": kernel_ms(dur_calls["embed_order_only"], launches_timed),
                       "size_order_launches": kernel_ms(dur_calls["size_order"], launches_timed)}
    tail_ms = kernel_ms(dur_calls["tail"], launches_timed) if "tail" in dur_calls else None
    one_wave_ms = kernel_ms(dur_calls["tail_one_wave_per_pair"], 16) if "tail_one_wave_per_pair" in dur_calls else None
    step_ms = dt / a.steps * 1e3
    kernels_ms = embed_ms * launches_per_step + (tail_ms or 0.0) * tail_calls_per_step
    # the kernels of a step cannot take longer than the step (back-to-back launches of ONE kernel lack only the
    # dependencies between different kernels, which cost time, never save it); 5 % for clock / box noise
    durations_consistent = bool(world > 1 or kernels_ms <= 1.05 * step_ms)
    if not durations_consistent:
        embed_ms = kernel_ms(dur_calls["embed"], launches_timed)          # once more before saying so
        tail_ms = kernel_ms(dur_calls["tail"], launches_timed) if "tail" in dur_calls else None
        kernels_ms = embed_ms * launches_per_step + (tail_ms or 0.0) * tail_calls_per_step
        durations_consistent = bool(kernels_ms <= 1.05 * step_ms)
    if world == 1 and kernels_ms > 1.25 * step_ms:        # reported (`kernel_durations.consistent`), never fatal: the line
        print("bench.py: kernel durations (%.4f ms) exceed the timed step (%.4f ms) - clock change between the two passes?"
              % (kernels_ms, step_ms), file=sys.stderr)    # with `value` must come out whatever the duration pass saw

    # N > 1, extra information (not `value`): the same K steps with the matrix left sharded by rows - what the
    # device-side consumers (F1-max counts, top-k retrieval) work on; isolates the cost of the gather to rank 0
    sharded = None
    if world > 1 and allpairs_job and not a.no_gather:
        for _ in range(a.warmup):
            step(gather=False)
        ts = timed_steps(a.steps, lambda: step(gather=False))
        sharded = {"ms_per_step": ts / a.steps * 1e3, "value": units * a.steps / ts,
                   "note": "matrix left sharded by rows (no gather to rank 0)"}

    # The same step at the reference's own operand width (extra information, after the timed region): debug bit 13 sends
    # every graph through the wide-range embed instance (three bf16 planes = 24-bit operands, fp32's range) instead of the
    # default two f16 planes (22 bits), and the dense all-pairs tail through its three-plane instance as well
    # (score_all_pairs_wide_kernel; the multi-rectangle and pair-list tails have no such instance and keep f16 planes).
    wide = None
    if world == 1 and not a.no_wide_range:
        try:
            eng.set_skip_mask(8192)
            for _ in range(3):
                step()
            wsteps = max(5, min(20, a.steps))
            tw = timed_steps(wsteps, step)
            wide_embed_ms = kernel_ms(dur_calls["embed"], 8)
            wide_tail_ms = kernel_ms(dur_calls["tail"], 8) if "tail" in dur_calls else None
            tail_is_wide = a.workload == "kitti00" or (a.workload == "kitti5seq" and a.per_sequence_tails)
            wide = {"ms_per_step": tw / wsteps * 1e3, "value_at_24bit_operands": units * wsteps / tw,
                    "wide_range_launch_ms": wide_embed_ms, "wide_tail_call_ms": wide_tail_ms, "tail_at_24bit_operands": tail_is_wide,
                    "steps": wsteps,
                    "note": "debug bit 13: embed on the wide-range instance and the dense all-pairs tail on its own (3 x bf16 planes = "
                            "24-bit matrix operands = the reference's fp32 operand width, fp32 accumulate)" if tail_is_wide else
                            "debug bit 13: embed on the wide-range instance (3 x bf16 planes = 24-bit matrix operands); this workload's "
                            "tail kernel has no three-plane instance and keeps 2 x f16 planes"}
        except Exception as e:
            wide = {"error": "%s: %s" % (type(e).__name__, e)}
        finally:
            eng.set_skip_mask(0)
        if allpairs_job and isinstance(wide, dict) and "error" not in wide:
            try:
                # the tail in exact fp32 (sgpr_score_pairs: one wave per pair, fp32 FMAs, no f16 planes) on a 2^20-pair sample
                _pw = eng.embed(jobs[0]["d_centers"], jobs[0]["d_labels"], k)[0]
                _gi = torch.randint(0, jobs[0]["m"], (2, 1 << 20), dtype=torch.int32, device=dev)
                _so = torch.empty(1 << 20, dtype=torch.float32, device=dev)
                t_fp32 = kernel_ms(lambda: eng.score_pairs(_pw, _pw, _gi[0], _gi[1], out=_so), 4)
                wide["tail_exact_fp32_pairs_per_s"] = (1 << 20) / (t_fp32 * 1e-3)
                wide["footnote_fp32_vector_tail"] = ("the tail WITHOUT matrix cores (sgpr_score_pairs: one wave per pair, fp32 FMAs) runs "
                                                     "%.3g pairs/s on a 2^20-pair sample - context only, not a datapath of the step"
                                                     % wide["tail_exact_fp32_pairs_per_s"])
            except Exception as e:
                wide["tail_exact_fp32_error"] = "%s: %s" % (type(e).__name__, e)

    # what `value` excludes by contract: host <-> device transfers (SURVEY.md 8d counts them in its metric)
    end_to_end = None
    if allpairs_job and world == 1 and not a.no_end_to_end:
        try:
            # the graphs cross PCIe as the ragged store (sgpr_embed_ragged: 13 bytes per real node, no padding slots); what a
            # FRESH data set pays is inside the step: the launch order is computed on the device from the offsets that just
            # arrived (sgpr_size_order, no read-back) and the launch makes no node_cap promise (the 64-row layout + the
            # hand-over of larger graphs)
            ragged = [eng.to_ragged(c, l) for c, l, _ in host_inputs]
            # (one pinned buffer per job - offsets | centers | labels -, one H2D copy: Engine.ragged_blob)
            pinned = [eng.ragged_blob(*r) for r in ragged]
            h2d_bytes = sum(x.nbytes for r in ragged for x in r)
            host_out = [torch.empty(j["m"], j["m"], dtype=torch.float32).pin_memory() for j in jobs]
            xz = [allpairs.pose_xz(p).to(dev) for _, _, p in host_inputs]
            reps = max(3, min(20, a.steps))

            copy_stream = torch.cuda.Stream(device=dev)
            pieces = max(1, a.d2h_pieces)
            dev_out = [torch.empty(j["m"], j["m"], dtype=torch.float32, device=dev) for j in jobs]

            def e2e(consumer):
                for j, (blob, layout), ho, do, pz in zip(jobs, pinned, host_out, dev_out, xz):
                    dc, dl, do_ = eng.ragged_views(blob.to(dev, non_blocking=True), layout)
                    order_r = eng.size_order_device(None, None, do_, n, k)[0] if a.embed_mode == "ordered" else None
                    pooled = eng.embed_ragged(dc, dl, do_, n, k, node_cap=0, order=order_r)[0]
                    if consumer == "d2h":
                        # the matrix leaves in row blocks: block i crosses PCIe on the copy stream while block i + 1 is scored
                        m = j["m"]
                        for q in range(pieces):
                            r0, r1 = m * q // pieces, m * (q + 1) // pieces
                            model.score_all_pairs(pooled[r0:r1].contiguous(), pooled, out=do[r0:r1])
                            ready = torch.cuda.Event()
                            ready.record()
                            copy_stream.wait_event(ready)
                            with torch.cuda.stream(copy_stream):
                                ho[r0:r1].copy_(do[r0:r1], non_blocking=True)
                    else:
                        from sg_pr_amd import metrics
                        mat = model.score_all_pairs(pooled, pooled, out=do)
                        metrics.f1_max_device(eng, mat, pose_xz=pz)
                torch.cuda.synchronize()

            end_to_end = {}
            for consumer in ("d2h", "device_f1"):
                e2e(consumer)
                t0 = time.perf_counter()
                for _ in range(reps):
                    e2e(consumer)
                te = (time.perf_counter() - t0) / reps
                end_to_end[consumer] = {"ms_per_step": te * 1e3, "value": units / te}
            end_to_end["note"] = ("per step: H2D of the graphs as a ragged store from pinned host memory (%.1f MB, one copy) + the launch order on "
                                  "the device from the offsets that arrived + the step without a node_cap promise + either "
                                  "the D2H copy of the score matrices into pinned memory (%.1f MB, in %d row blocks whose copies "
                                  "overlap the scoring of the next block; `d2h`) or the device-side F1-max over them in one "
                                  "engine call with 64 bytes crossing PCIe (`device_f1`); %d repetitions"
                                  % (h2d_bytes / 1e6, units * 4 / 1e6, pieces, reps))
        except Exception as e:     # extra legs beside `value`: a failure here is reported, the contract line still comes out
            end_to_end = {"error": "%s: %s" % (type(e).__name__, e)}

    if rank == 0:
        value = units * a.steps / dt
        g = graphs_per_step
        # ALGORITHMIC FLOPs of the embed launches of one step: the factored formulation evaluated at each
        # graph's processed slot count (surplus padding slots are dropped).  The kernel executes fewer still: the
        # semantic branch runs on 13 label super-nodes instead of the graph's nodes.
        n_eff = np.concatenate(n_eff_all)
        flops = float(sum(embed_flops_per_graph(int(v), k) for v in n_eff)) / launches_per_step   # per launch
        flops_dense = embed_flops_per_graph(n, k) * g / launches_per_step
        bytes_ = embed_bytes_per_graph(n) * g / launches_per_step
        ach_tflops = flops / (embed_ms * 1e-3) / 1e12
        ach_gbs = bytes_ / (embed_ms * 1e-3) / 1e9
        # HBM traffic of the dominant kernel: PMC counters cannot be read from inside this process; the value comes from
        # the committed rocprofv3 --pmc passes of the same workload (profiles/pmc_hbm_latest.json) and is reported only
        # while that profile was taken from exactly these kernel sources
        traffic, traffic_tail, issue, issue_tail = None, None, None, None
        try:
            with open(os.path.join(REPO, "profiles", "pmc_hbm_latest.json")) as f:
                pmc = json.load(f)
            if pmc.get("source_hash") == source_hash() and a.workload == "kitti00" and world == 1:
                pe = pmc["sgpr::embed_kernel"]
                if pe["graphs_per_launch"] == g and pe["node_num"] == n:
                    traffic = (2 * pe["FETCH_SIZE_KiB"] + pe["WRITE_SIZE_KiB"]) * 1024.0   # gfx950 FETCH_SIZE correction
                    pc = pmc.get("embed_kernel_counters")
                    if pc and pc.get("SQ_INSTS_VALU"):
                        # the resource that binds the kernel: instruction issue.  Busy cycles of the vector / matrix pipe
                        # = 4 x SQ_ACTIVE_INST_VALU (the counter ticks once per 4 cycles an instruction occupies the
                        # pipe, matrix instructions included) against the SIMD-cycles of the launch (4 SIMDs per CU at
                        # the shader clock); the kernel's launch time tracks its instruction count
                        # the shader clock is the one MEASURED under this kernel (tools/collect_profiles.py: SQ_BUSY_CYCLES over
                        # the 32 shader engines / the kernel's duration in the same counter pass - a lower bound of the
                        # clock, so an upper bound of the fraction; GRBM_GUI_ACTIVE gives the other side)
                        clk_lo, clk_hi = pc.get("clock_ghz_from_sq_busy"), pc.get("clock_ghz_from_grbm")
                        clock_hz = clk_lo * 1e9 if clk_lo else SHADER_CLOCK_HZ
                        simd_cycles = embed_ms * 1e-3 * clock_hz * 4 * eng_cus
                        busy = 4.0 * pc.get("SQ_ACTIVE_INST_VALU", 0)
                        issue = {"bound": "valu_issue", "achieved": busy, "peak": simd_cycles, "unit": "SIMD-cycles",
                                 "frac": busy / simd_cycles,
                                 "shader_clock_ghz": {"used": clock_hz / 1e9, "from_sq_busy_cycles": clk_lo, "from_grbm_gui_active": clk_hi,
                                                      "measured": bool(clk_lo)},
                                 "frac_at_grbm_clock": (busy / (embed_ms * 1e-3 * clk_hi * 1e9 * 4 * eng_cus)) if clk_hi else None,
                                 "instructions_per_launch": {k_: pc.get(k_) for k_ in (
                                     "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_MFMA", "SQ_INSTS_VMEM")},
                                 "mfma_busy_cycles": pc.get("SQ_VALU_MFMA_BUSY_CYCLES"),
                                 "wave_cycles": pc.get("SQ_WAVE_CYCLES"), "wait_cycles": pc.get("SQ_WAIT_ANY"),
                                 "note": "counters from the committed rocprofv3 --pmc passes of these sources; "
                                         "launch time from this run; shader clock %.2f GHz %s" % (
                                             clock_hz / 1e9, "measured in the counter pass" if clk_lo else "assumed")}
                pt = pmc.get("sgpr::score_all_pairs_kernel")
                if pt and a.graphs == 4541:
                    traffic_tail = (2 * pt["FETCH_SIZE_KiB"] + pt["WRITE_SIZE_KiB"]) * 1024.0
                    tc = pmc.get("tail_kernel_counters")
                    if tc and tc.get("SQ_ACTIVE_INST_VALU") and tail_ms:
                        # what binds the tail is instruction issue too (vector and f16 matrix instructions do not overlap
                        # at throughput); the HBM figure above is what SURVEY 8d prices it against.  The
                        # event-timed call holds the prep kernel as well: the SIMD-cycles are an upper bound, the
                        # fraction a lower one (the kernel alone: profiles/rNN_kernel_stats.txt)
                        simd_cycles_t = tail_ms * 1e-3 * (pmc.get("embed_kernel_counters", {}).get("clock_ghz_from_sq_busy") or SHADER_CLOCK_HZ / 1e9) * 1e9 * 4 * eng_cus
                        busy_t = 4.0 * tc["SQ_ACTIVE_INST_VALU"]
                        issue_tail = {"bound": "valu_issue", "achieved": busy_t, "peak": simd_cycles_t, "unit": "SIMD-cycles",
                                      "frac": busy_t / simd_cycles_t,
                                      "instructions_per_launch": {k_: tc.get(k_) for k_ in (
                                          "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_MFMA", "SQ_INSTS_VMEM")},
                                      "mfma_busy_cycles": tc.get("SQ_VALU_MFMA_BUSY_CYCLES"),
                                      "note": "score_all_pairs_kernel counters from the committed --pmc passes of these "
                                              "sources; time = the event-timed tail call (prep + tail kernels) of this run"}
        except (OSError, KeyError, ValueError):
            pass
        res = {
            "metric": "graph-pairs/sec", "value": value, "unit": "graph-pairs/s", "n_gpus": world,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True, "scaling": "strong" if (allpairs_job or a.workload == "pairlist") else "weak",
            "value_definition": "inputs and outputs resident in HBM (the bench contract); SURVEY.md 8d's transfer-inclusive "
                                "figures are end_to_end.d2h (scores copied to the host) and end_to_end.device_f1 (F1-max on the "
                                "device; they include the launch order computed on the device per step and make no node_cap promise)",
            "vs_baseline": None, "dtype": "f32 (2xf16-plane operands on the matrix cores, fp32 accumulate; fp32 vector math)",
            "data": data_kind,
            "config": {"workload": wl_name, "graphs": int(sum(mm for _, mm in seqs)) if allpairs_job else int(m),
                       "node_num": n, "K": k, "pairs_per_step": int(units), "node_cap": int(node_cap_report),
                       "embed_launch_order": "largest graph first" if a.embed_mode == "ordered" else "as stored",
                       "parallelism": ("graphs + list sharded x%d" if a.workload == "pairlist" else "row-sharded x%d") % world, "backend": backend if world > 1 else None,
                       "gather_to_rank0": (not a.no_gather) if allpairs_job else None,
                       "gather_chunks": a.chunks if (allpairs_job and world > 1) else None,
                       "checkpoint": "tests/golden/model.pth",
                       "device_prep_ms_once": prep_ms,
                       "device_prep_note": "node_cap + largest-first launch order of the resident arrays: sgpr_size_order on the "
                                           "device + a 4-byte read-back (wall time incl. the synchronisation), once per data set, "
                                           "outside the timed step; end_to_end recomputes the order per step and promises no node_cap",
                       "embed_calls_ms": embed_forms},
            "kernel_durations": {"embed_call_ms": embed_ms, "tail_call_ms": tail_ms, "sum_per_step_ms": kernels_ms,
                                 "ms_per_step": step_ms, "consistent": durations_consistent,
                                 "launches_timed": launches_timed},
            "sharded_output": sharded,
            "end_to_end": end_to_end,
            "roofline": {"kernel": "sgpr::embed_big_kernel" if a.workload == "stress" else "sgpr::embed_kernel", "bound": "valu",
                         "bound_note": "frac = useful-work-equivalent throughput (algorithmic FLOPs of the factored formulation at "
                                       "each graph's processed slots / launch time) against the fp32 VECTOR peak - not a pipe "
                                       "utilisation; what binds the kernel is vector issue + dependent latency (selection "
                                       "networks, gather-max, epilogues): roofline.issue.frac; the matrix pipe is ~10 % busy, "
                                       "HBM ~0.8 % (roofline.hbm)",
                         "wide_range": wide,
                         "achieved": ach_tflops, "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": ach_tflops / FP32_PEAK_TFLOPS, "traffic": traffic, "issue": issue, "launch_ms": embed_ms, "launches_timed": launches_timed,
                         "timing": "HIP events around %d back-to-back calls after the timed region (embed_kernel + its "
                                   "second-pass dispatcher)" % launches_timed,
                         "graphs_per_launch": g / launches_per_step, "launches_per_step": launches_per_step,
                         "flops_per_launch_algorithmic": flops, "mean_nodes_processed": float(np.mean(n_eff)),
                         "dense_equivalent_tflops": flops_dense / (embed_ms * 1e-3) / 1e12,
                         "frac_8d_dense": flops_dense / (embed_ms * 1e-3) / 1e12 / FP32_PEAK_TFLOPS,
                         "frac_8d_dense_note": "SURVEY.md 8d's algorithmic FLOPs (factored formulation at ALL node_num slots) / launch "
                                               "time / fp32 peak: above 1 because the kernel - exactly - does not do that work "
                                               "(trailing duplicate slots collapse to one, the semantic branch runs on 13 label "
                                               "super-nodes); not a utilisation, see issue.frac",
                         "flops_per_graph_dense": embed_flops_per_graph(n, k),
                         "hbm": {"achieved": ach_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": ach_gbs / HBM_PEAK_GBS, "bytes_per_graph": embed_bytes_per_graph(n)}},
        }
        if a.workload == "pairlist":
            # the grouped pair-list tail (ntn_prep_list + score_pair_list kernels of one call): algorithmic bytes = the
            # pooled vectors of the graphs read once + per listed pair its two indices, its output position and its score
            tb = graphs_per_step * 128 * 2 + plan.P * 16
            res["pairlist"] = {"pairs": int(plan.P), "graphs_embedded": int(graphs_per_step), "distinct_row_graphs": plan.n_rows,
                               "work_items": plan.n_items, "embed_call_ms": embed_ms, "tail_call_ms": tail_ms,
                               "tail_one_wave_per_pair_ms": one_wave_ms,
                               "tail_below_embed": bool(tail_ms < embed_ms),
                               "plan_build_ms_once": plan_ms,
                               "seq02_alone": ({"pairs": 71226, "graphs": 4661,
                                                "embed_call_ms": kernel_ms(dur_calls["seq02_embed"], 16),
                                                "tail_call_ms": kernel_ms(dur_calls["seq02_tail"], 16),
                                                "tail_one_wave_per_pair_ms": kernel_ms(dur_calls["seq02_tail_one_wave_per_pair"], 16)}
                                               if "seq02_tail" in dur_calls else None),
                               "hbm": {"achieved": tb / (tail_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                       "frac": tb / (tail_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "bytes_per_call_algorithmic": tb},
                               "note": "kernels: sgpr::ntn_prep_list_kernel + sgpr::score_pair_list_kernel (grouped by row "
                                       "graph, 16 listed columns per MFMA group) against sgpr::score_pairs_kernel (one wave "
                                       "per pair) on the same list; the call is launch-latency bound at this size"}
        elif tail_ms is not None:
            # all-pairs tail (ntn_prep + score_all_pairs kernels of one call): the HBM-write-bound piece -
            # algorithmic bytes = the scores written + the pooled vectors read
            rows_per_call = sum(allpairs.shard_bounds(mm, world, 0)[1] for _, mm in seqs) / max(tail_calls_per_step, 1)
            cols = float(np.mean([mm for _, mm in seqs]))
            tb = sum(allpairs.shard_bounds(mm, world, 0)[1] * mm * 4 + (allpairs.shard_bounds(mm, world, 0)[1] + mm) * 128
                     for _, mm in seqs) / max(tail_calls_per_step, 1)
            gbs = tb / (tail_ms * 1e-3) / 1e9
            res["roofline_tail"] = {"kernel": "sgpr::ntn_prep_kernel + sgpr::score_all_pairs_kernel", "bound": "hbm",
                                    "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                                    "traffic": traffic_tail, "launch_ms": tail_ms, "rows_per_call": rows_per_call,
                                    "mean_cols": cols, "bytes_per_call_algorithmic": tb,
                                    "calls_per_step": tail_calls_per_step, "issue": issue_tail}
        if not a.no_cpu_baseline and world == 1:
            try:
                res["cpu_baseline"] = cpu_baseline(args.model, host_inputs[0][0], host_inputs[0][1], n, k, a.cpu_seconds,
                                                   allpairs_job, pairs=host_pairs if a.workload == "pairlist" else None,
                                                   sample_pairs=a.cpu_pairs)
            except Exception as e:      # the GPU measurement above stands whatever happens to the host-side comparison leg
                res["cpu_baseline"] = {"value": None, "unit": "graph-pairs/s", "cores": 0, "kind": "port", "sample": "failed",
                                       "error": "%s: %s" % (type(e).__name__, e)}
        print(json.dumps(res))
        sys.stdout.flush()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(ckpt, centers, labels, n, k, target_s, allpairs_job, pairs=None, sample_pairs=0):
    """The oracle (faithful torch-CPU restatement of the reference forward: both graphs of
    every pair embedded, materialised edge tensors) timed on the host cores over a bounded
    sample of pairs drawn from the same workload."""
    import numpy as np
    import torch
    from oracle import sgpr_oracle as oracle   # checker / baseline only
    from sg_pr_amd import synth
    sd = oracle.load_checkpoint(ckpt)
    rng = np.random.default_rng(0)
    bsz = 128
    m = centers.shape[0]

    def batch():
        if pairs is not None:                 # the listed pairs themselves, in list order (eval_batch.py:30-36)
            at = int(rng.integers(0, max(len(pairs[0]) - bsz, 1)))
            i, j = pairs[0][at:at + bsz], pairs[1][at:at + bsz]
        else:
            i = rng.integers(0, m, size=bsz)
            j = rng.integers(0, m, size=bsz)
        f1 = torch.from_numpy(synth.dense_features(centers[i], labels[i]))
        f2 = torch.from_numpy(synth.dense_features(centers[j], labels[j]))
        return f1, f2

    # pick the intra-op thread count that serves this shape best (all cores oversubscribe it)
    ncpu = os.cpu_count() or 1
    f1, f2 = batch()
    best = None
    for th in sorted({t for t in (8, 16, 32, 64, 128) if t <= ncpu} | {min(ncpu, 8)}):
        torch.set_num_threads(th)
        oracle.forward(sd, f1[:32], f2[:32], k)
        t0 = time.perf_counter()
        oracle.forward(sd, f1, f2, k)
        one = time.perf_counter() - t0
        if best is None or one < best[0]:
            best = (one, th)
    one, cores = best
    torch.set_num_threads(cores)
    reps = int(max(1, min(64, round(target_s / max(one, 1e-3)))))
    if sample_pairs > 0:                      # an explicit sample size (100 000 pairs) instead of the time bound
        reps = (int(sample_pairs) + bsz - 1) // bsz
    t0 = time.perf_counter()
    for _ in range(reps):
        f1, f2 = batch()
        oracle.forward(sd, f1, f2, k)
    dt = time.perf_counter() - t0
    res = {"value": reps * bsz / dt, "unit": "graph-pairs/s", "cores": cores, "kind": "port",
           "sample": "%d random pairs of the same workload in batches of %d (node_num=%d, K=%d), faithful per-pair "
                     "forward incl. dense feature assembly; SURVEY.md 8d asks for a 100 k-pair sample (~%d s at this "
                     "rate) - bounded to ~%d s of CPU work by the bench contract"
                     % (reps * bsz, bsz, n, k, int(100000 / (reps * bsz / dt)), int(target_s)),
           "host_cpus": ncpu, "cpu_model": cpu_model()}
    if allpairs_job:
        # context, not the baseline: the SAME algorithm as the GPU path on the CPU (embed every graph once, then only
        # the NTN + head per pair) - separates "better algorithm" from "faster hardware"
        ge = min(m, 256)
        fe = torch.from_numpy(synth.dense_features(centers[:ge], labels[:ge]))
        oracle.embed(sd, fe[:32], k)
        t0 = time.perf_counter()
        pooled = oracle.embed(sd, fe, k)[0]
        te = (time.perf_counter() - t0) / ge
        rows = pooled[:64].contiguous()
        oracle.score_all_pairs(sd, rows[:8], pooled)
        t0 = time.perf_counter()
        oracle.score_all_pairs(sd, rows, pooled)
        tp = (time.perf_counter() - t0) / (rows.shape[0] * pooled.shape[0])
        res["same_algorithm_cpu"] = {"value": (m * m) / (m * te + m * m * tp), "unit": "graph-pairs/s",
                                     "embed_s_per_graph": te, "tail_s_per_pair": tp,
                                     "sample": "%d graphs embedded once + a %d x %d tail block, extrapolated to the "
                                               "%d x %d matrix" % (ge, rows.shape[0], pooled.shape[0], m, m)}
    return res


if __name__ == "__main__":
    main()
