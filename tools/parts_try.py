#!/usr/bin/env python
"""usage (GPU box): python tools/parts_try.py   -- the layer kernel on a rank's shard of the headline batch (4 / 8 / 16 images): strips handed over and their outputs
dealt as parts (csrc/conv_fused.hip: plan_parts; ctx option fused_parts) against the whole-strip / shared-last-round launches, per strip shape: ms per synchronous
step, the layer kernel's own launch (HIP events) and the ELBO (identical for every row of a batch)."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
from deepcgp_amd import device as dev, synthetic as syn
from deepcgp_amd.models import build_from_spec
spec, X, Y = syn.make_config("cfg2_mnist_CH_M256")
scale = float(spec["num_data"]) / X.shape[0]
ctx = dev.get_context()
combos = [dict(fused_parts=0), dict(fused_parts=-2), dict(fused_shape=0, fused_parts=0), dict(fused_shape=0, fused_parts=-2), dict(fused_shape=0, fused_parts=2),
          dict(fused_shape=0, fused_parts=5), dict(fused_shape=0, fused_parts=10), dict(fused_shape=7, fused_parts=0), dict(fused_shape=7, fused_parts=-2),
          dict(fused_shape=7, fused_parts=2), dict(fused_shape=7, fused_parts=5), dict(fused_shape=7, fused_parts=10), dict(fused_parts=-2)]
for b in [int(a) for a in sys.argv[1:]] or (4, 8, 16):
    model = build_from_spec(spec, X[:b], Y[:b])
    dX, dY = ctx.to_device(X[:b]), ctx.to_device(Y[:b], np.int32)
    ref = None
    for kw in combos:
        with ctx.options(**kw):
            for i in range(40):
                e = model.compute_log_likelihood(dX, dY, seed=i, scale=scale)
            ctx.sync()
            t0 = time.perf_counter()
            for i in range(200):
                e = model.compute_log_likelihood(dX, dY, seed=7, scale=scale)
            ctx.sync()
            dt = (time.perf_counter() - t0) / 200
            ctx.timing_enable(3)
            ctx.timing_reset()
            for i in range(40):
                model.compute_log_likelihood(dX, dY, seed=7, scale=scale)
            ctx.sync()
            tim = ctx.timing().get("conv_fused", (0, 0.0))
            ctx.timing_enable(0)
        if ref is None:
            ref = e
        print("batch %2d %-36s %.4f ms/step  conv_fused %6.1f us  elbo %.12g  rel diff %.1e" % (
            b, ",".join("%s=%d" % kv for kv in kw.items()), 1e3 * dt, 1e3 * tim[1] / max(tim[0], 1), e, abs(e - ref) / abs(ref)), flush=True)
    model.close()
