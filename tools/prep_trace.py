#!/usr/bin/env python
"""Which task of the one-launch prepare_all kernel (csrc/prep.hip) is the long one: first start / last end per (layer, task).
Needs the trace build:  tools/ab_build.sh ptrace prep.hip 's/^\\/\\/ PREP_TRACE_SWITCH/#define DCGP_PREP_TRACE 1/'
usage (GPU box): DCGP_LIB=deepcgp_amd/ab/libdcgp_ptrace.so python tools/prep_trace.py [config]"""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepcgp_amd import device as dev, synthetic as syn          # noqa: E402
from deepcgp_amd.models import build_from_spec                   # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "cfg2_mnist_CH_M256"
spec, X, Y = syn.make_config(name)
ctx = dev.get_context()
model = build_from_spec(spec, X, Y)
dX, dY = ctx.to_device(X), ctx.to_device(Y, np.int32)
for i in range(5):
    model.compute_log_likelihood(dX, dY, seed=i)
raw = ctypes.CDLL(os.environ["DCGP_LIB"])
buf = np.zeros((4096, 6), np.uint64)
tasks = ["gram K", "gram Kp", "transpose", "q_sqrt", "q_mu", "ZS"]
for rep in range(3):
    raw.dcgp_debug_prep_trace(None, 1)
    model.compute_log_likelihood(dX, dY, seed=9 + rep)
    raw.dcgp_debug_prep_trace(buf.ctypes.data_as(ctypes.c_void_p), 0)
    live = buf[:, 1] > 0
    b = buf[live].astype(np.int64)
    t0 = b[:, 0].min()
    print("rep %d: %d blocks, launch span %.2f us (from the first block's start)" % (rep, len(b), (b[:, 1].max() - t0) / 100.0))
    for z in range(8):
        for y in range(6):
            m = b[:, 2] == (z | (y << 8))
            if m.any():
                d = (b[m, 1] - b[m, 0]) / 100.0
                print("  layer %d %-10s %4d blocks  first start %6.2f  last start %6.2f  last end %6.2f   block duration mean %5.2f max %5.2f  shader clock %.2f GHz  segment known +%.2f  layer args +%.2f" % (
                    z, tasks[y], m.sum(), (b[m, 0].min() - t0) / 100.0, (b[m, 0].max() - t0) / 100.0, (b[m, 1].max() - t0) / 100.0, d.mean(), d.max(), (b[m, 3] / np.maximum(d, 0.01)).mean() / 1e3,
                    ((b[m, 4] - b[m, 0]) / 100.0).mean(), ((b[m, 5] - b[m, 4]) / 100.0).mean()))
