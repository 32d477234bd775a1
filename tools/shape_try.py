#!/usr/bin/env python
"""usage (GPU box): python tools/shape_try.py   -- the one-launch conv layer's strip shapes (csrc/conv_fused.hip kShapes, forced through the ctx
option fused_shape, and the sharing of the last round's strips, option fused_split) on a rank's shard of the headline batch: ms per synchronous step and the ELBO (identical for every shape)."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
from deepcgp_amd import device as dev, synthetic as syn
from deepcgp_amd.models import build_from_spec
spec, X, Y = syn.make_config("cfg2_mnist_CH_M256")
scale = float(spec["num_data"]) / X.shape[0]
ctx = dev.get_context()
combos = [(-1, -1), (-1, 0), (2, 0), (7, 0), (3, 0), (0, 0), (0, -1), (0, 2), (0, 3), (0, 4), (7, -1)]
for b in (4, 8, 16, 32):
    model = build_from_spec(spec, X[:b], Y[:b])
    dX, dY = ctx.to_device(X[:b]), ctx.to_device(Y[:b], np.int32)
    ref = None
    for shape, split in combos:
        with ctx.options(fused_shape=shape, fused_split=split):
            for i in range(40):
                e = model.compute_log_likelihood(dX, dY, seed=i, scale=scale)
            ctx.sync()
            t0 = time.perf_counter()
            for i in range(200):
                e = model.compute_log_likelihood(dX, dY, seed=7, scale=scale)
            ctx.sync()
            dt = (time.perf_counter() - t0) / 200
        if ref is None:
            ref = e
        print("batch %2d shape %2d split %2d: %.4f ms/step  elbo %.12g  rel diff %.2e" % (b, shape, split, 1e3 * dt, e, abs(e - ref) / abs(ref)))
    model.close()
