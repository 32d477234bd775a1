#!/usr/bin/env python
"""What the brackets of bench.py's timed region cost on their own (device sync + rank barrier on an idle device), and the per-step
series of a short timed region -- why `--steps 20` reads a few percent under `--steps 200`.  usage (GPU box): python tools/bench_overhead.py"""
import gc
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                     # noqa: E402
from deepcgp_amd import device as dev, dist, synthetic as syn   # noqa: E402

ctx = dev.get_context()
grp = dist.HostGroup(0, 1)
cfg = syn.CONFIGS["cfg2_mnist_CH_M256"]
leg = bench.Leg(ctx, grp, "rccl", "cfg2_mnist_CH_M256", 10, cfg["batch"], 0, cfg["batch"], False)
for i in range(50):
    leg.step(i)
gc.collect()
gc.disable()
for rep in range(3):
    t0 = time.perf_counter(); leg.barrier(); t1 = time.perf_counter()
    print("barrier on an idle device: %.1f us" % ((t1 - t0) * 1e6))
for rep in range(3):
    leg.barrier()
    ts = []
    t0 = time.perf_counter()
    for i in range(20):
        a = time.perf_counter(); leg.step(5 + i); ts.append(time.perf_counter() - a)
    t1 = time.perf_counter(); leg.barrier(); t2 = time.perf_counter()
    print("20 steps: %.1f us in the steps (first %.1f, median %.1f, max %.1f) + closing barrier %.1f us -> %.4f ms/step as bench.py counts it" % (
        (t1 - t0) * 1e6, ts[0] * 1e6, np.median(ts) * 1e6, max(ts) * 1e6, (t2 - t1) * 1e6, (t2 - t0) * 1e3 / 20))
