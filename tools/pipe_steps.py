#!/usr/bin/env python
"""N forward ELBO steps with `depth` steps in flight (dcgp_elbo_forward_enqueue / _collect) -- the workload of the pipelined
timeline profiles: tools/prof_pipe.sh.  usage: python tools/pipe_steps.py [config] [steps] [depth] [images: the shard of a rank]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepcgp_amd import device as dev, synthetic as syn          # noqa: E402
from deepcgp_amd.models import build_from_spec                   # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "cfg2_mnist_CH_M256"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
depth = int(sys.argv[3]) if len(sys.argv) > 3 else 2
spec, X, Y = syn.make_config(name)
scale = float(spec["num_data"]) / X.shape[0]          # a shard keeps the global batch's scale
if len(sys.argv) > 4:
    X, Y = X[:int(sys.argv[4])], Y[:int(sys.argv[4])]
ctx = dev.get_context()
model = build_from_spec(spec, X, Y)
dX, dY = ctx.to_device(X), ctx.to_device(Y, np.int32)
for i in range(30):
    model.compute_log_likelihood(dX, dY, seed=i, scale=scale)
ctx.sync()
t0 = time.perf_counter()
tickets = []
for i in range(steps):
    tickets.append(model.enqueue_log_likelihood(dX, dY, seed=i, scale=scale))
    if len(tickets) >= depth:
        model.collect_log_likelihood(tickets.pop(0))
while tickets:
    model.collect_log_likelihood(tickets.pop(0))
dt = time.perf_counter() - t0
print("%s: %d steps, %d in flight: %.4f ms/step = %.1f steps/s" % (name, steps, depth, 1e3 * dt / steps, steps / dt))
