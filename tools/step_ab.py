#!/usr/bin/env python
"""usage (GPU box): STEP_AB_SETS="name=value[,name=value] ..." python tools/step_ab.py [config ...]  -- the synchronous forward ELBO step of BASELINE configs
under ctx option sets: ms per step (wall clock of 300 steps behind 100 untimed), steps/s and the ELBO (identical for every set of a config)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepcgp_amd import device as dev, synthetic as syn          # noqa: E402
from deepcgp_amd.models import build_from_spec                   # noqa: E402

names = [a for a in sys.argv[1:] if not a.startswith("-")] or ["cfg2_mnist_H_M256", "cfg2_mnist_CH_M256", "cfg1_mnist_H_M32"]
sets = os.environ.get("STEP_AB_SETS", "no_tail_ride=1,prep_on_chain=1 no_tail_ride=0,prep_on_chain=1 no_tail_ride=1,prep_on_chain=0 no_tail_ride=0,prep_on_chain=0").split()
ctx = dev.get_context()
for name in names:
    spec, X, Y = syn.make_config(name)
    scale = float(spec["num_data"]) / X.shape[0]
    model = build_from_spec(spec, X, Y)
    dX, dY = ctx.to_device(X), ctx.to_device(Y, np.int32)
    ref = None
    for rep in range(2):
        for st in sets:
            kw = {k: int(v) for k, v in (kv.split("=") for kv in st.split(","))}
            with ctx.options(**kw):
                for i in range(100):
                    e = model.compute_log_likelihood(dX, dY, seed=i, scale=scale)
                ctx.sync()
                t0 = time.perf_counter()
                for i in range(300):
                    e = model.compute_log_likelihood(dX, dY, seed=7, scale=scale)
                ctx.sync()
                dt = (time.perf_counter() - t0) / 300
            if ref is None:
                ref = e
            print("%-22s %-40s %.4f ms/step  %7.1f steps/s  elbo %.12g  rel diff %.1e" % (name, st, 1e3 * dt, 1.0 / dt, e, abs(e - ref) / abs(ref)), flush=True)
    model.close()
