#!/usr/bin/env python
"""usage (GPU box): python tools/fused_power.py  -- is the one-launch conv layer bound by the chip's power budget?  (a) the same 64-column strips on
90 / 180 / 248 / 256 x 2.8 CUs (batch 4 / 8 / 11 / 32 of the headline config, shape 0, no sharing of the last round): strip duration in cycles and the
shader clock held; (b) board power and clocks sampled by rocm-smi while the full-batch step loops for a few seconds."""
import os
import subprocess
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepcgp_amd import device as dev, synthetic as syn          # noqa: E402
from deepcgp_amd.models import build_from_spec                   # noqa: E402

spec, X, Y = syn.make_config("cfg2_mnist_CH_M256")
ctx = dev.get_context()
for b in (4, 8, 11, 32):
    model = build_from_spec(spec, X[:b], Y[:b])
    dX, dY = ctx.to_device(X[:b]), ctx.to_device(Y[:b], np.int32)
    with ctx.options(fused_shape=0, fused_split=0, fused_persist=0):
        for i in range(300):
            model.compute_log_likelihood(dX, dY, seed=i)
        ctx.timing_enable(3); ctx.timing_reset()
        for i in range(60):
            model.compute_log_likelihood(dX, dY, seed=i)
        ctx.sync()
        tim = ctx.timing().get("conv_fused", (0, 0.0)); ctx.timing_enable(0)
        buf = ctx.to_device(np.zeros((32, 16, 16), np.int64), np.int64)
        dev.lib().dcgp_debug_set_fused_trace(ctx.handle, buf.ptr)
        model.compute_log_likelihood(dX, dY, seed=9)
        dev.lib().dcgp_debug_set_fused_trace(ctx.handle, None)
    t = buf.numpy().astype(np.float64)
    rows = []
    for s in range(0, 32, 4):
        if not t[s, 0, 0]:
            continue
        live = t[s, :, 0] > 0
        cyc = t[s, live, 9].max() - t[s, live, 0].min()
        wall = (t[s, 0, 11] - t[s, 0, 10]) / 100.0
        rows.append((cyc, wall, (t[s, 0, 9] - t[s, 0, 0]) / wall / 1e3, t[s, live, 6].max() - t[s, live, 5].min()))
    r = np.array(rows)
    print("batch %2d (%4d strips): kernel %.1f us; sampled strips: %.0f cycles (second product %.0f), wall %.1f us, shader clock %.3f GHz" % (
        b, (b * 1440 + 63) // 64, 1e3 * tim[1] / max(tim[0], 1), r[:, 0].mean(), r[:, 3].mean(), r[:, 1].mean(), r[:, 2].mean()), flush=True)
    model.close()

model = build_from_spec(spec, X, Y)
dX, dY = ctx.to_device(X), ctx.to_device(Y, np.int32)
stop = False
samples = []


def sampler():
    while not stop:
        try:
            out = subprocess.run(["/opt/rocm/bin/rocm-smi", "--showpower", "--showclocks", "--csv"], capture_output=True, text=True, timeout=10).stdout
            samples.append(out)
        except Exception as e:   # noqa: BLE001
            samples.append("error %r" % e)
        time.sleep(0.3)


th = threading.Thread(target=sampler)
th.start()
t0 = time.time()
n = 0
while time.time() - t0 < 6.0:
    for i in range(200):
        model.compute_log_likelihood(dX, dY, seed=i)
    n += 200
ctx.sync()
stop = True
th.join()
print("%d steps in %.2f s" % (n, time.time() - t0))
for s in samples[:3] + samples[-4:]:
    print(s.strip())
try:
    print(subprocess.run(["/opt/rocm/bin/rocm-smi", "--showmaxpower", "--showpowercap" if False else "--showmaxpower"], capture_output=True, text=True, timeout=10).stdout)
    print(subprocess.run(["/opt/rocm/bin/amd-smi", "metric", "-p", "-c"], capture_output=True, text=True, timeout=20).stdout[:3000])
except Exception as e:   # noqa: BLE001
    print("smi error", e)
