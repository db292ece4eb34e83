#!/usr/bin/env python3
"""Generates tests/golden/pair_lists_3_20.npz from the reference's evaluation pair lists
(/root/reference/data_process/pair_list/pair_list_3_20_{02,05,06,08}.npy: arrays of strings "<i>_<j>.json", the pairs
eval_batch.py:30-36 walks for KITTI sequences 02 / 05 / 06 / 08 - SURVEY.md row 14 / 8d "a real pair-index
distribution").  Only the index pairs are kept - data, no code: per sequence a uint16 [P, 2] array sorted by (i, j)
(the reference's files are in shuffled order; consumers that want a list order shuffle with their own seed).
The graphs and poses the indices refer to are not in the reference tree (README.md:54).

    python tests/golden/make_pair_lists.py [/root/reference]
"""
import os
import sys

import numpy as np

ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
out = {}
for seq in ("02", "05", "06", "08"):
    names = np.load(os.path.join(ref, "data_process", "pair_list", "pair_list_3_20_%s.npy" % seq))
    ij = np.array([[int(x) for x in str(v).split(".")[0].split("_")] for v in names], dtype=np.int64)
    assert ij.min() >= 0 and ij.max() < 65536
    ij = ij[np.lexsort((ij[:, 1], ij[:, 0]))]
    out["seq_" + seq] = ij.astype(np.uint16)
    print(seq, ij.shape[0], "pairs over", np.unique(ij).size, "graphs; max index", ij.max())
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "pair_lists_3_20.npz"), **out)
