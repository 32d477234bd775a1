"""Command-line flags of an experiment, as a table.

The flag NAMES and DEFAULTS are the reference's whole configuration system (/root/reference/conv_gp/arguments.py:9-43) and are
what its option files (results/*/options.toml) record, so they are kept to the letter; everything else here is this
repository's: one table, one loop, and a note per flag on what the device path does with it."""
import argparse
import math

# (flag, type or None for a switch, default, what it selects)
FLAGS = (
    ("--name", str, "experiment", "run name (log directory, checkpoint file name)"),
    ("--lr-decay-steps", int, 100000, "staircase period of the learning-rate decay (x 0.1 per period)"),
    ("--test-every", int, 50000, "optimisation steps between two test-set evaluations"),
    ("--test-size", int, 10000, "test images used per evaluation"),
    ("--num-samples", int, 10, "S: Monte-Carlo samples per image in the doubly-stochastic ELBO"),
    ("--log-dir", str, "results", "where logs and checkpoints go"),
    ("--lr", float, 0.01, "Adam / SGD learning rate (device optimiser: dcgp_model_adam_step / _sgd_step)"),
    ("--batch-size", int, 32, "minibatch size (images)"),
    ("--optimizer", str, "Adam", "Adam | SGD | NatGrad (NatGrad: variational parameters by natural gradient, rest by Adam)"),
    ("-M", str, "384,384", "inducing patches per layer, comma separated; one value = SVGP head only"),
    ("--feature-maps", str, "10", "outputs of each conv layer, comma separated ('' with a head-only model)"),
    ("--filter-sizes", str, "5,5", "patch size of each layer incl. the head"),
    ("--strides", str, "2,1", "patch stride of each layer incl. the head"),
    ("--base-kernel", str, "rbf", "base kernel of the conv layers: rbf | acos"),
    ("--white", None, False, "whitened variational parameters"),
    ("--last-kernel", str, "conv", "head kernel: conv | add | rbf"),
    ("--gamma", float, 0.001, "NatGrad step size"),
    ("--identity-mean", None, False, "Conv2dMean identity mean function on the conv layers"),
    ("--load-model", str, None, "checkpoint (.npy, reference key set) to start from"),
)
REQUIRED = ("--name",)


def default_parser():
    """argparse parser over FLAGS (same namespace attributes as the reference's parser)."""
    parser = argparse.ArgumentParser(description="Deep convolutional GP experiment flags")
    for flag, kind, default, doc in FLAGS:
        if kind is None:
            parser.add_argument(flag, action="store_true", default=default, help=doc)
        else:
            parser.add_argument(flag, type=kind, default=default, required=flag in REQUIRED, help=doc)
    return parser


def train_steps(flags):
    """Number of test periods until the decayed learning rate reaches about 1e-5 (conv_gp/arguments.py:4-7)."""
    decades = math.log(5e-5 / flags.lr, 0.1)
    return math.ceil(decades * flags.lr_decay_steps / flags.test_every)
