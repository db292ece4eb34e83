#!/usr/bin/env python3
"""Whole-sequence parity census with proofs (tests/tie_proof.py): every graph of a KITTI-00-sized world-consistent
sequence through the engine and the CPU oracle; every graph whose embedding differs is traced to the kNN row that chose
another of two near-tied candidates and the tie is proven in float64; both 4541 x 4541 score matrices, the scores beyond
1e-4 and the F1-max of both.  Output: profiles/r04_seq_parity.txt.
usage: seq_parity.py [num_graphs=4541] [out.txt]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import torch
from sg_pr_amd import engine, synth
from oracle import sgpr_oracle as oracle      # checker only
import tie_proof
G = int(sys.argv[1]) if len(sys.argv) > 1 else 4541
out_path = sys.argv[2] if len(sys.argv) > 2 else None
ckpt = os.path.join(REPO, "tests", "golden", "model.pth")
sd = oracle.load_checkpoint(ckpt)
eng = engine.Engine(sd)
torch.set_num_threads(min(64, os.cpu_count() or 8))
lines = []
def log(s):
    print(s, flush=True)
    lines.append(s)
for name, gen in (("world_sequence (one world, revisits alike)", synth.world_sequence),
                  ("kitti_like_sequence (independent graphs, the bench's default)", synth.kitti_like_sequence)):
    log("== %s, %d graphs, node_num 100, K 10, seed 0" % (name, G))
    c, l, _, poses = gen(G, 100, seed=0)
    r = tie_proof.census(eng, oracle, sd, c, l, poses, log=log)
    log("")
if out_path:
    with open(out_path, "w") as f:
        f.write("\n".join(lines) + "\n")
