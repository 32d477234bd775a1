import sys, time, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepcgp_amd import synthetic as syn, device as dev
from deepcgp_amd.models import build_from_spec
cfg = syn.CONFIGS["cfg2_mnist_CH_M256"]
spec = syn.make_spec(cfg["hwc"], cfg["convs"], cfg["head"], cfg["M"], S=10, num_data=cfg["num_data"], seed=1, conv_q_sqrt_scale=0.3)
X, Y = syn.make_batch(cfg["hwc"], cfg["batch"], seed=1)
model = build_from_spec(spec, X, Y); model.dedup_layer0 = True
ctx = dev.get_context(); dX, dY = ctx.to_device(X), ctx.to_device(Y, np.int32)
model.compute_gradients(dX, dY, seed=0, fetch=False)
for i in range(2): model.natgrad_step(1e-4)
ctx.sync(); t0 = time.perf_counter()
for i in range(10): model.natgrad_step(1e-4)
ctx.sync(); print("natgrad step %.3f ms" % (1e3 * (time.perf_counter() - t0) / 10))
