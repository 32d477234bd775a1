"""Command-line flags -- the same names, defaults and help semantics as
/root/reference/conv_gp/arguments.py:9-43 (it is the reference's whole config system)."""
import argparse
import math


def train_steps(flags):
    # roughly until the learning rate becomes 1e-5 (conv_gp/arguments.py:4-7)
    decay_count = math.log(5e-5 / flags.lr, 0.1)
    return math.ceil(flags.lr_decay_steps * decay_count / flags.test_every)


def default_parser():
    p = argparse.ArgumentParser()
    p.add_argument('--name', type=str, required=True, default='experiment')
    p.add_argument('--lr-decay-steps', type=int, default=100000)
    p.add_argument('--test-every', type=int, default=50000)
    p.add_argument('--test-size', type=int, default=10000)
    p.add_argument('--num-samples', type=int, default=10)
    p.add_argument('--log-dir', type=str, default='results')
    p.add_argument('--lr', type=float, default=0.01)
    p.add_argument('--batch-size', type=int, default=32)
    p.add_argument('--optimizer', type=str, default='Adam')
    p.add_argument('-M', type=str, default='384,384')
    p.add_argument('--feature-maps', type=str, default='10')
    p.add_argument('--filter-sizes', type=str, default='5,5')
    p.add_argument('--strides', type=str, default='2,1')
    p.add_argument('--base-kernel', type=str, default='rbf')
    p.add_argument('--white', action='store_true', default=False)
    p.add_argument('--last-kernel', type=str, default='conv')
    p.add_argument('--gamma', type=float, default=0.001)
    p.add_argument('--identity-mean', action='store_true')
    p.add_argument('--load-model', type=str, default=None)
    return p
