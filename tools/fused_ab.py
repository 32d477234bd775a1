#!/usr/bin/env python
"""usage (GPU box): python tools/fused_ab.py [config] [--libs a.so,b.so]  -- the one-launch conv layer (csrc/conv_fused.hip) under its launch options on a
BASELINE config: per option set the synchronous step (ms, wall clock of 200 steps) and the layer kernel's own launch (us, HIP events around every launch,
timing mode 3), with the ELBO (identical for every set).  Option sets: name=value[,name=value...] separated by spaces in FUSED_AB_SETS, default below."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepcgp_amd import device as dev, synthetic as syn          # noqa: E402
from deepcgp_amd.models import build_from_spec                   # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "cfg2_mnist_CH_M256"
DEFAULT = ("fused_persist=0 fused_persist=-1 fused_persist=2 fused_persist=1,fused_pre=0 fused_pre=0 fused_pre=-1 fused_pre=3 fused_pre=9 fused_shape=2,fused_persist=0 "
           "fused_shape=2,fused_persist=1,fused_stagger=0 fused_shape=2,fused_persist=1,fused_stagger=40 fused_persist=0 fused_pre=0 fused_pre=-1")
sets = os.environ.get("FUSED_AB_SETS", DEFAULT).split()
spec, X, Y = syn.make_config(name)
scale = float(spec["num_data"]) / X.shape[0]
ctx = dev.get_context()
model = build_from_spec(spec, X, Y)
dX, dY = ctx.to_device(X), ctx.to_device(Y, np.int32)
ref = None
for st in sets:
    kw = {k: int(v) for k, v in (kv.split("=") for kv in st.split(","))}
    with ctx.options(**kw):
        for i in range(60):
            e = model.compute_log_likelihood(dX, dY, seed=i, scale=scale)
        ctx.sync()
        t0 = time.perf_counter()
        for i in range(200):
            e = model.compute_log_likelihood(dX, dY, seed=7, scale=scale)
        ctx.sync()
        dt = (time.perf_counter() - t0) / 200
        ctx.timing_enable(3)
        ctx.timing_reset()
        for i in range(60):
            model.compute_log_likelihood(dX, dY, seed=7, scale=scale)
        ctx.sync()
        tim = ctx.timing()
        ctx.timing_enable(0)
    t = tim.get("conv_fused", (0, 0.0))
    if ref is None:
        ref = e
    print("%-58s step %.4f ms  conv_fused %7.1f us x %d   elbo %.12g  rel diff %.1e" % (
        st, 1e3 * dt, 1e3 * t[1] / max(t[0], 1), t[0], e, abs(e - ref) / abs(ref)), flush=True)
model.close()
