"""Counterpart of the reference's eval_pair.py (eval_pair.py:6-13): score one pair of graphs.

    python -m sg_pr_amd.eval_pair [config.yml]

The config has the reference's layout (config/config.yml); `eva_pair.pair_file`
is the two-element list of graph JSONs.  Prints `Score: <float>`."""
import sys

from .parser_sg import sgpr_args
from .sg_net import SGTrainer
from .utils import tab_printer


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = sgpr_args()
    args.load(argv[0] if argv else './config/config.yml')
    tab_printer(args)
    trainer = SGTrainer(args, False)
    trainer.model.eval()
    pred, gt = trainer.eval_batch_pair([args.pair_file, ])
    print("Score:", pred[0])
    return pred[0]


if __name__ == "__main__":
    main()
