"""Host I/O helpers with the reference's names (utils.py:10-38, 61-70)."""
import json
import math
import os


def tab_printer(args):
    """utils.py:10-19 - print the parameters as a table (texttable is optional here)."""
    rows = [[k.replace("_", " ").capitalize(), v] for k, v in sorted(vars(args).items())]
    try:
        from texttable import Texttable
        t = Texttable()
        t.add_rows([["Parameter", "Value"]] + rows)
        print(t.draw())
    except ImportError:
        width = max(len(r[0]) for r in rows) if rows else 0
        for name, val in rows:
            print("%-*s  %s" % (width, name, val))


def read_graph(path):
    """One graph JSON: {"centers": [[x,y,z]...], "nodes": [label...], "pose": [12 floats]}."""
    with open(path) as f:
        return json.load(f)


def pose_distance(pose1, pose2):
    """Planar distance between two 3x4 KITTI poses (x = pose[3], z = pose[11]); utils.py:36."""
    return math.sqrt((pose1[3] - pose2[3]) ** 2 + (pose1[11] - pose2[11]) ** 2)


def process_pair(path):
    """utils.py:21-38 - `path` is [json_1, json_2]; returns the pair dictionary."""
    data1 = read_graph(path[0])
    data2 = read_graph(path[1])
    return {
        "centers_1": data1["centers"],
        "nodes_1": data1["nodes"],
        "centers_2": data2["centers"],
        "nodes_2": data2["nodes"],
        "distance": pose_distance(data1["pose"], data2["pose"]),
    }


def load_paires(file, graph_pairs_dir):
    """utils.py:61-70 - pair list file with lines `a.json b.json` -> [[path_a, path_b], ...]."""
    paires = []
    with open(file) as f:
        for line in f:
            if not line:
                break
            line = line.strip().split(" ")
            paires.append([os.path.join(graph_pairs_dir, line[0]), os.path.join(graph_pairs_dir, line[1])])
    return paires
