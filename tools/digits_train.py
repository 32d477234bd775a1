#!/usr/bin/env python
"""usage (GPU box): python tools/digits_train.py [steps] [variant ...]

End-to-end learning on REAL images that ship offline: sklearn.datasets.load_digits (1797 8 x 8 grey-level digits, 10 classes),
standardised like the reference standardises MNIST (conv_gp/mnist.py:40-45), 1500 / 297 train / test split, trained with the
reference's own flags through ModelBuilder + models.train (Adam branch of conv_gp/experiment.py:84-108) and scored the way its
log does (AccuracyLogger, conv_gp/utils/log.py:50-67).  Prints ELBO / accuracy over the run for each variant:
  head   -M 32 --feature-maps '' --filter-sizes 3 --strides 1      (the paper's "1-layer": SVGP head with the ConvKernel)
  conv   -M 32,32 --feature-maps 4 --filter-sizes 3,3 --strides 1,1 (one ConvLayer + head)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepcgp_amd.arguments import default_parser                   # noqa: E402
from deepcgp_amd.models import ModelBuilder, train, AccuracyLogger  # noqa: E402

VARIANTS = {"head": ["-M", "32", "--feature-maps", "", "--filter-sizes", "3", "--strides", "1"],
            "conv": ["-M", "32,32", "--feature-maps", "4", "--filter-sizes", "3,3", "--strides", "1,1"]}


def digits(seed=0):
    from sklearn.datasets import load_digits
    d = load_digits()
    X = d.images.astype(np.float64)
    X = (X - X.mean()) / X.std()
    idx = np.random.default_rng(seed).permutation(X.shape[0])
    X, Y = X[idx][..., None], d.target[idx].astype(np.int64)
    return X[:1500], Y[:1500], X[1500:], Y[1500:]


def run(variant, steps, lr=0.01, seed=0):
    Xtr, Ytr, Xte, Yte = digits()
    flags = default_parser().parse_args(["--name", "digits", "--batch-size", "64", "--lr", str(lr), "--num-samples", "5"] + VARIANTS[variant])
    np.random.seed(seed)
    model = ModelBuilder(flags, Xtr, Ytr.reshape(-1, 1)).build()
    acc = AccuracyLogger(Xte.reshape(len(Xte), -1), Yte)
    out = [(0, None, acc(model))]
    t0 = time.time()
    done = 0
    while done < steps:
        n = min(250, steps - done)
        hist = train(model, n, lr=lr, lr_decay_steps=10 ** 9, global_step=done, seed=seed)
        done += n
        out.append((done, float(np.mean(hist)), acc(model)))
    model.close()
    return out, time.time() - t0


if __name__ == "__main__":
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
    for v in (sys.argv[2:] or list(VARIANTS)):
        out, dt = run(v, steps)
        print(v, "%.1f s" % dt, " ".join("%d:%s/%.3f" % (s, "-" if e is None else "%.0f" % e, a) for s, e, a in out))
