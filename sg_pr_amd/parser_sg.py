"""Configuration bag with the reference's attribute names (parser_sg.py:3-67)."""
import os

import yaml


class sgpr_args():
    """Same attributes / defaults as the reference `sgpr_args`; `load(path)` reads the
    same YAML sections (common / arch / train / eva_batch / eva_pair)."""

    def __init__(self):
        # common
        self.model = ""
        self.graph_pairs_dir = "/dir_of_graph_pairs"
        self.p_thresh = 3
        self.batch_size = 128
        self.pair_list_dir = ''
        self.cuda = "0"
        # arch
        self.keep_node = 1
        self.filters_1 = 64
        self.filters_2 = 64
        self.filters_3 = 32
        self.tensor_neurons = 16
        self.bottle_neck_neurons = 16
        self.K = 10
        # train (kept for config compatibility; training is out of scope here)
        self.epochs = 500
        self.train_sequences = []
        self.eval_sequences = []
        self.dropout = 0
        self.learning_rate = 1e-3
        self.weight_decay = 5e-4
        self.gpu = 0
        self.logdir = "./logs"
        self.node_num = 100
        # eva_batch
        self.sequences = []
        self.output_path = "./eva"
        self.show = False
        # eva_pair
        self.pair_file = ""

    _SECTIONS = {
        "arch": ["keep_node", "filters_1", "filters_2", "filters_3", "tensor_neurons", "bottle_neck_neurons", "K"],
        "train": ["epochs", "train_sequences", "eval_sequences", "dropout", "learning_rate", "weight_decay", "gpu",
                  "logdir", "node_num"],
        "common": ["model", "cuda", "batch_size", "p_thresh", "graph_pairs_dir", "pair_list_dir"],
        "eva_batch": ["sequences", "output_path", "show"],
        "eva_pair": ["pair_file"],
    }

    def load(self, config_file):
        with open(os.path.abspath(config_file)) as f:
            cfg = yaml.load(f, Loader=yaml.FullLoader)  # the reference omits Loader (breaks on PyYAML>=6)
        for section, keys in self._SECTIONS.items():
            for key in keys:
                setattr(self, key, cfg[section][key])  # KeyError on a missing entry, like the reference
