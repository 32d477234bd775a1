"""Time / profile dcgp_elbo_grad on a bench configuration (needs a GPU):  python tools/grad_time.py [config] [steps]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deepcgp_amd import synthetic as syn, device as dev
from deepcgp_amd.models import build_from_spec

name = sys.argv[1] if len(sys.argv) > 1 else "cfg2_mnist_CH_M256"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
cfg = syn.CONFIGS[name]
S = cfg.get("S", 10)
spec = syn.make_spec(cfg["hwc"], cfg["convs"], cfg["head"], cfg["M"], S=S, num_data=cfg["num_data"], seed=1)
X, Y = syn.make_batch(cfg["hwc"], cfg["batch"], seed=1)
model = build_from_spec(spec, X, Y)
model.dedup_layer0 = bool(int(os.environ.get("DCGP_DEDUP", "0")))   # exact layer-0 de-duplication (training default)
ctx = dev.get_context()
dX, dY = ctx.to_device(X), ctx.to_device(Y, np.int32)
for i in range(2):
    model.compute_gradients(dX, dY, seed=i, fetch=False)
ctx.sync()
t0 = time.perf_counter()
for i in range(steps):
    model.compute_gradients(dX, dY, seed=i, fetch=False)
ctx.sync()
print("%s: value+grad %.3f ms/step" % (name, 1e3 * (time.perf_counter() - t0) / steps))
if os.environ.get("DCGP_TRAIN", "1") != "0":   # the one-call training step: value, gradient and the Adam update (dcgp_model_train_step_adam)
    for i in range(2):
        model.train_step(dX, dY, 1e-9, seed=i)
    t0 = time.perf_counter()
    for i in range(steps):
        model.train_step(dX, dY, 1e-9, seed=i)
    print("%s: training step (value + gradient + Adam, one call) %.3f ms/step" % (name, 1e3 * (time.perf_counter() - t0) / steps))
if os.environ.get("DCGP_GRAD_HOST"):   # host time to enqueue a step (timers on: a few more events per step)
    ctx.timing_enable(1); ctx.timing_reset()
    for i in range(steps):
        model.compute_gradients(dX, dY, seed=i, fetch=False)
    ctx.sync()
    t = ctx.timing()
    for k in ("grad_host_enqueue", "host_enqueue"):
        if k in t:
            print("  %s: %.1f us/step" % (k, 1e3 * t[k][1] / max(1, t[k][0])))
