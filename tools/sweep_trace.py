#!/usr/bin/env python
"""usage (GPU box): python tools/sweep_trace.py <config> <family: kuf | kuf_long | head_sweep> [option=value ...]

Where a patch-sweep launch (csrc/head_units.hip) spends its time: every workgroup x wave stamps entry / image staged / set-up done /
first unit / last unit / exit (dcgp_debug_set_sweep_trace).  Prints the launch's span, the distribution of workgroup start times,
set-up and unit durations, and how many waves are busy over time (so that a thin tail or a slow ramp shows)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepcgp_amd import device as dev, synthetic as syn          # noqa: E402
from deepcgp_amd.models import build_from_spec                   # noqa: E402

name, family = sys.argv[1], sys.argv[2]
opts = dict(kv.split("=") for kv in sys.argv[3:])
cfg = syn.CONFIGS[name]
seed = 1234 + list(syn.CONFIGS).index(name)
spec = syn.make_spec(cfg["hwc"], cfg["convs"], cfg["head"], cfg["M"], S=10, num_data=cfg["num_data"], seed=seed)
X, Y = syn.make_batch(cfg["hwc"], cfg["batch"], seed=seed)
ctx = dev.get_context()
model = build_from_spec(spec, X, Y)
dX, dY = ctx.to_device(X), ctx.to_device(Y, np.int32)
WGS, WAVES = 16384, 4
with ctx.options(no_fused_layer=1, no_early_sweep=1, head_no_overlap=1, **{k: int(v) for k, v in opts.items()}):
    for i in range(20):
        model.compute_log_likelihood(dX, dY, seed=i)
    buf = ctx.to_device(np.zeros((WGS * WAVES, 8), np.int64), np.int64)
    dev.lib().dcgp_debug_set_sweep_trace(ctx.handle, buf.ptr, WGS, family.encode())
    model.compute_log_likelihood(dX, dY, seed=99)
    ctx.sync()
    dev.lib().dcgp_debug_set_sweep_trace(ctx.handle, None, 0, None)
raw = buf.numpy()
where = (raw[:, 7] >> 32) & 0xfffff      # XCC_ID << 16 | HW_ID[15:0] of the wave
raw = raw.copy()
raw[:, 7] &= 0xffffffff
t = raw.astype(np.float64)
live = t[:, 0] > 0
t = t[live]
where = where[live]
if not len(t):
    raise SystemExit("no stamps: family %r was not launched" % family)
w0, w1 = t[:, 0].min(), t[:, 6].max()
span_us = (w1 - w0) / 100.0
print("%s %s: %d waves stamped, launch span %.1f us (wall clock, first entry -> last exit)" % (name, family, len(t), span_us))
# shader clock per wave from its own wall / cycle pair (long-lived waves only)
dur_w = (t[:, 6] - t[:, 0]) / 100.0
ran = t[:, 7] > 0
last = np.where(t[:, 5] > 0, t[:, 5], t[:, 4])
cyc = last - t[:, 1]
ok = ran & (dur_w > 2.0)
ghz = np.median(cyc[ok] / dur_w[ok]) / 1e3 if ok.any() else 2.4
print("shader clock %.2f GHz (median over waves)" % ghz)
start = (t[:, 0] - w0) / 100.0
end = (t[:, 6] - w0) / 100.0
q = lambda a: "min %.1f  p10 %.1f  median %.1f  p90 %.1f  max %.1f" % (a.min(), np.percentile(a, 10), np.median(a), np.percentile(a, 90), a.max())   # noqa: E731
print("wave start (us after the first):  " + q(start))
print("wave end:                         " + q(end))
print("image staged (us after entry):    " + q((t[:, 2] - t[:, 1]) / ghz / 1e3))
print("set-up done (us after entry):     " + q((t[:, 3] - t[:, 1]) / ghz / 1e3))
if ran.any():
    first = (t[ran, 4] - t[ran, 3]) / ghz / 1e3
    print("first unit (us):                  " + q(first))
    more = ran & (t[:, 7] > 1)
    if more.any():
        print("further units, each (us):         " + q((t[more, 5] - t[more, 4]) / ghz / 1e3 / (t[more, 7] - 1)))
    print("units per wave: %s" % dict(zip(*np.unique(t[:, 7].astype(int), return_counts=True))))
# occupancy over time
edges = np.linspace(0, span_us, 21)
print("busy waves over time (20 slices of %.1f us): %s" % (span_us / 20, " ".join(
    "%d" % int(((start < hi) & (end > lo)).sum()) for lo, hi in zip(edges[:-1], edges[1:]))))
# placement: waves per SIMD among those that start late (the launch's second round), and how long their unit takes against the number of
# late waves that share the SIMD
simd = (where >> 16) * 4096 + ((where >> 8) & 0xff) * 16 + ((where >> 4) & 3)     # (XCC, SE/SH/CU, SIMD)
late = start > 0.4 * span_us
if late.any() and ran.any():
    ids, cnt = np.unique(simd[late], return_counts=True)
    n_simd = len(np.unique(simd))
    hist = dict(zip(*np.unique(cnt, return_counts=True)))
    hist[0] = n_simd - len(ids)
    print("late waves (start > 40 %% of the span): %d on %d SIMDs seen; SIMDs by number of late waves: %s" % (late.sum(), n_simd, dict(sorted(hist.items()))))
    per = dict(zip(ids, cnt))
    k = np.array([per[s] for s in simd[late]])
    unit = (t[late, 4] - t[late, 3]) / ghz / 1e3
    for c in sorted(set(k)):
        sel = (k == c) & (t[late, 7] > 0)
        if sel.any():
            print("  late waves on a SIMD with %d of them: %d, first unit median %.1f us (min %.1f max %.1f)" % (c, sel.sum(), np.median(unit[sel]), unit[sel].min(), unit[sel].max()))
if os.environ.get("SWEEP_PLACE"):
    # where the launch's first-round workgroups sit: (workgroup, wave) -> SIMD, for the first CUs seen
    idx = np.nonzero(live)[0]
    wg, wv = idx // WAVES, idx % WAVES
    cu = (where >> 16) * 4096 + ((where >> 8) & 0xff)
    sid = (where >> 4) & 3
    early = start < 0.2 * span_us
    for c in np.unique(cu)[:3]:
        sel = early & (cu == c)
        order = np.argsort(wg[sel] * WAVES + wv[sel])
        print("CU %05x first-round (workgroup.wave:SIMD): %s" % (c, " ".join("%d.%d:%d" % (a, b, d) for a, b, d in zip(wg[sel][order], wv[sel][order], sid[sel][order]))))
    # how often do the two waves of a workgroup sit on SIMDs (0,1) / (2,3) / other pairs
    pairs = {}
    for w in np.unique(wg[early]):
        m = early & (wg == w)
        key = tuple(sid[m][np.argsort(wv[m])])
        pairs[key] = pairs.get(key, 0) + 1
    print("SIMDs of a workgroup's waves (first round): %s" % dict(sorted(pairs.items(), key=lambda kv: -kv[1])[:12]))
model.close()
