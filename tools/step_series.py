#!/usr/bin/env python
"""Wall time of every one of 600 consecutive synchronous forward steps of the headline configuration, in windows of 25: how long the part
takes to reach its steady state after start-up (and after any idle stretch: see tools/bench_overhead.py).  usage (GPU box): python tools/step_series.py"""
import sys, time, numpy as np
sys.path.insert(0, ".")
from deepcgp_amd import device as dev, synthetic as syn
from deepcgp_amd.models import build_from_spec
spec, X, Y = syn.make_config("cfg2_mnist_CH_M256")
ctx = dev.get_context()
model = build_from_spec(spec, X, Y)
dX, dY = ctx.to_device(X), ctx.to_device(Y, np.int32)
ts = []
for i in range(600):
    t0 = time.perf_counter()
    model.compute_log_likelihood(dX, dY, seed=i)
    ts.append(time.perf_counter() - t0)
ts = np.array(ts) * 1e3
for k in range(0, 600, 25):
    print("steps %3d-%3d  mean %.4f ms  min %.4f max %.4f" % (k, k + 24, ts[k:k+25].mean(), ts[k:k+25].min(), ts[k:k+25].max()))
