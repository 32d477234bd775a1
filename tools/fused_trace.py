#!/usr/bin/env python
"""Phase timeline of the one-launch conv layer kernel (csrc/conv_fused.hip) inside a forward step of a BASELINE config:
shader-clock stamps of 8 sampled workgroups x their waves.  usage: python tools/fused_trace.py [config] (GPU box)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepcgp_amd import device as dev, synthetic as syn          # noqa: E402
from deepcgp_amd.models import build_from_spec                   # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "cfg2_mnist_CH_M256"
spec, X, Y = syn.make_config(name)
ctx = dev.get_context()
model = build_from_spec(spec, X, Y)
dX, dY = ctx.to_device(X), ctx.to_device(Y, np.int32)
for i in range(5):
    model.compute_log_likelihood(dX, dY, seed=i)
buf = ctx.to_device(np.zeros((32, 16, 16), np.int64), np.int64)   # [8 sampled workgroups x 4 strips of a persistent one][wave][stamp]
dev.lib().dcgp_debug_set_fused_trace(ctx.handle, buf.ptr)
model.compute_log_likelihood(dX, dY, seed=9)
dev.lib().dcgp_debug_set_fused_trace(ctx.handle, None)
t = buf.numpy().astype(np.float64)
g0 = t[:, :, 0][t[:, :, 0] > 0].min()
GHZ = 2.4
print("summary (us at %.1f GHz): sampled workgroup . strip, start relative to the earliest stamp, duration, second-product span" % GHZ)
for b in range(32):
    if not t[b, 0, 0]:
        continue
    live = t[b, :, 0] > 0
    wall_us = (t[b, 0, 11] - t[b, 0, 10]) / 100.0           # wall_clock64: 100 MHz
    # (prologues ahead, conv_fused.hip: an item that runs phases 0 - 2 only leaves no stamp behind "A1 published", one that fetches A1 none in front of it)
    kind = "prologue only" if not t[b, 0, 9] else ("A1 fetched" if not t[b, 0, 1] else "whole strip")
    last = t[b, live, :10].max()
    print("  wg %d.%d  start %8.2f  duration %7.2f  stage3 %7.2f   wall %7.2f us -> shader clock %.3f GHz  %s" % (
        b // 4, b % 4, (t[b, live, 0].min() - g0) / GHZ / 1e3, (last - t[b, live, 0].min()) / GHZ / 1e3,
        (t[b, live, 6].max() - t[b, live, 5].min()) / GHZ / 1e3 if t[b, 0, 9] else 0.0, wall_us, (t[b, 0, :10].max() - t[b, 0, 0]) / wall_us / 1e3, kind))
names = ["start", "images+xn", "kuf done", "kuf barrier", "stage1 done", "A1 published", "stage3 done", "mean done", "partials", "end"]
if "--summary" in sys.argv:
    sys.exit(0)
for b in range(32):
    if not t[b, 0, 0] or (b // 4) % 4:
        continue
    print("workgroup %d strip %d (us at %.1f GHz after the earliest stamp of the launch):" % (b // 4, b % 4, GHZ))
    for w in range(16):
        if not t[b, w, 0]:
            continue
        print("  wave %2d " % w + " ".join("%s %7.2f" % (n, (t[b, w, k] - g0) / GHZ / 1e3) for k, n in enumerate(names)))
