#!/usr/bin/env python
"""usage (GPU box): python tools/grad_repeat.py [config ...]  -- the training step's gradients must come out with the same bits step after step
(the reverse pass runs on three streams; an unordered pair of accesses shows up as a difference in steady state).  Full-size configurations,
tiled and de-duplicated; tests/test_gpu_model.py::test_gradients_repeat_in_steady_state is the small-size version in the suite."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepcgp_amd import synthetic as syn, device as dev          # noqa: E402
from deepcgp_amd.models import build_from_spec                   # noqa: E402

names = sys.argv[1:] or ["cfg2_mnist_CH_M256", "cfg2_mnist_H_M256", "cfg3_mnist_3layer_M256"]
ctx = dev.get_context()
bad = 0
for name in names:
    cfg = syn.CONFIGS[name]
    spec = syn.make_spec(cfg["hwc"], cfg["convs"], cfg["head"], cfg["M"], S=10, num_data=cfg["num_data"], seed=1)
    X, Y = syn.make_batch(cfg["hwc"], cfg["batch"], seed=1)
    model = build_from_spec(spec, X, Y)
    dX, dY = ctx.to_device(X), ctx.to_device(Y, np.int32)
    for dedup in (True, False):
        model.dedup_layer0 = dedup
        first = None
        for it in range(8):
            e, g = model.compute_gradients(dX, dY, seed=3)
            if first is None:
                first = (e, g)
                continue
            if e != first[0]:
                bad += 1
                print(name, "dedup" if dedup else "tiled", "step", it, "ELBO differs", e - first[0])
            for li, (a, b) in enumerate(zip(g, first[1])):
                for k in a:
                    if not np.array_equal(a[k], b[k]):
                        bad += 1
                        print(name, "dedup" if dedup else "tiled", "step", it, "layer", li, k, "max diff", np.abs(a[k] - b[k]).max())
        print(name, "dedup" if dedup else "tiled", "ELBO", first[0], "checked 7 repeats")
    model.close()
print("differences:", bad)
