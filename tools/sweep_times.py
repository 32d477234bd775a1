#!/usr/bin/env python
"""usage (GPU box, repo root): python tools/sweep_times.py [config ...]

The patch sweeps (csrc/head_units.hip) of every BASELINE configuration, each launch ALONE on the chip, against their rooflines:
  * "kuf": the storing form (MultiOutputConvKernel.Kuf, conv_gp/layers.py:23-32) of every conv layer on the sweep + GEMM route --
    achieved HBM GB/s on the algorithmic bytes 8 (N' H W C + M L + P M N');
  * "head_sweep": the reducing form (ConvKernel.Kzx + Kdiag, conv_gp/kernels.py:106-133) -- TFLOP/s by N' P (P + M)(2 L + 4).
HIP events around every launch of the family (ctx timing mode 1), ctx options no_fused_layer / no_early_sweep / head_no_overlap so
that nothing runs beside the sweep.  Prints one line per (configuration, layer)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deepcgp_amd import device as dev          # noqa: E402
from deepcgp_amd import synthetic as syn       # noqa: E402
from deepcgp_amd.models import build_from_spec  # noqa: E402


def geometry(c):
    Ho, Wo = (c["H"] - c["f"]) // c["s"] + 1, (c["W"] - c["f"]) // c["s"] + 1
    return Ho * Wo, c["f"] * c["f"] * c["C"]


def main():
    names = sys.argv[1:] or list(syn.CONFIGS)
    ctx = dev.get_context()
    S = 10
    for name in names:
        cfg = syn.CONFIGS[name]
        seed = 1234 + list(syn.CONFIGS).index(name)
        spec = syn.make_spec(cfg["hwc"], cfg["convs"], cfg["head"], cfg["M"], S=S, num_data=cfg["num_data"], seed=seed)
        X, Y = syn.make_batch(cfg["hwc"], cfg["batch"], seed=seed)
        model = build_from_spec(spec, X, Y)
        dX, dY = ctx.to_device(X), ctx.to_device(Y, np.int32)
        rows = cfg["batch"] * S
        with ctx.options(no_fused_layer=1, no_early_sweep=1, head_no_overlap=1):
            for i in range(12):
                model.compute_log_likelihood(dX, dY, seed=i)
            ctx.timing_enable(1)
            ctx.timing_reset()
            n = 20
            for i in range(n):
                model.compute_log_likelihood(dX, dY, seed=i)
            ctx.sync()
            tim = ctx.timing()
            ctx.timing_enable(0)
        for fam in ("kuf", "kuf_long"):      # short patches (first layers, HBM-bound) / long patches (L = 250, MFMA-bound)
            if fam not in tim or not tim[fam][0]:
                continue
            us = 1e3 * tim[fam][1] / n
            nbytes = flops = 0.0
            desc = []
            for c in spec["convs"]:
                P, L = geometry(c)
                if (L > 64) != (fam == "kuf_long"):
                    continue
                nbytes += 8.0 * (rows * c["H"] * c["W"] * c["C"] + c["M"] * L + float(P) * c["M"] * rows)
                flops += float(P) * c["M"] * rows * (2 * L + 4)
                desc.append("P=%d L=%d M=%d" % (P, L, c["M"]))
            print("%-24s %-10s %8.1f us/step  %7.1f MB  %7.0f GB/s  frac %.3f  %6.1f TF/s  (%s; %d launches/step)" %
                  (name, fam, us, nbytes / 1e6, nbytes / us / 1e3, nbytes / us / 1e3 / 8000.0, flops / us / 1e6, ", ".join(desc), tim[fam][0] // n))
        if "head_sweep" in tim and tim["head_sweep"][0]:
            h = spec["head"]
            P, L = geometry(h)
            us = 1e3 * tim["head_sweep"][1] / tim["head_sweep"][0]
            flops = float(rows) * P * (P + h["M"]) * (2 * L + 4)
            print("%-24s head_sweep %8.1f us       %7.2f GF  %7.1f TF/s  frac %.3f  (P=%d L=%d M=%d)" %
                  (name, us, flops / 1e9, flops / us / 1e6, flops / us / 1e6 / 78.6, P, L, h["M"]))
        sys.stdout.flush()
        model.close()


if __name__ == "__main__":
    main()
