"""Generates tests/golden/*.npz: seeded inputs + expected outputs of the hot path.

The reference (TensorFlow 1.x + GPflow 1.2.0 + doubly_stochastic_dgp) cannot be imported in the build
container, so these vectors are produced by the float64 oracle (oracle/, PARITY UNPINNED -- see
oracle/__init__.py) after it has been cross-checked against the independent restatement oracle/alt.py
(agreement is asserted here before anything is written).  Run from the repo root:

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from deepcgp_amd import synthetic as syn          # noqa: E402
from oracle import alt                            # noqa: E402
from oracle_build import oracle_model             # noqa: E402

# (name, HWC, convs [(f, s, R)], head (f, s), M, N, S, white)  -- SURVEY.md section 8(c) shapes
CASES = [
    ("g1_8x8x1_f3s1_M4_R2", (8, 8, 1), [(3, 1, 2)], (3, 1), 4, 3, 2, False),
    ("g2_9x7x3_f4s2_M5_R3", (9, 7, 3), [(4, 2, 3)], (2, 1), 5, 2, 2, False),
    ("g3_12x12x10_f5s1_M8_R10", (12, 12, 10), [(5, 1, 10)], (5, 1), 8, 2, 3, False),
    ("g4_28x28x1_f5s2_M16_R10", (28, 28, 1), [(5, 2, 10)], (5, 1), 16, 2, 2, False),
    ("g5_white_10x10x2_M6", (10, 10, 2), [(3, 2, 3)], (3, 1), 6, 3, 2, True),
    ("g6_3layer_14x14x1_M7", (14, 14, 1), [(4, 2, 3), (3, 1, 2)], (3, 1), 7, 2, 2, False),
    ("g7_headonly_12x12x1_M9", (12, 12, 1), [], (5, 1), 9, 4, 2, False),
]


def flatten_spec(spec):
    out = {"S": spec["S"], "num_data": spec["num_data"], "n_convs": len(spec["convs"])}
    for i, c in enumerate(spec["convs"]):
        for k, v in c.items():
            out["conv%d_%s" % (i, k)] = v
    for k, v in spec["head"].items():
        out["head_%s" % k] = v
    return out


def unflatten_spec(d):
    spec = {"S": int(d["S"]), "num_data": int(d["num_data"]), "convs": []}
    scalars = ("H", "W", "C", "f", "s", "M", "R")
    for i in range(int(d["n_convs"])):
        c = {}
        for k in ("H", "W", "C", "f", "s", "M", "R", "Z", "Z0", "variance", "ls", "q_mu", "q_sqrt", "white"):
            v = d["conv%d_%s" % (i, k)]
            c[k] = int(v) if k in scalars else (bool(v) if k == "white" else (float(v) if k in ("variance", "ls") else np.array(v)))
        spec["convs"].append(c)
    h = {}
    for k in ("H", "W", "C", "f", "s", "M", "R", "Z", "variance", "ls", "w", "q_mu", "q_sqrt", "white"):
        v = d["head_%s" % k]
        h[k] = int(v) if k in scalars else (bool(v) if k == "white" else (float(v) if k in ("variance", "ls") else np.array(v)))
    spec["head"] = h
    return spec


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    for idx, (name, hwc, convs, head, M, N, S, white) in enumerate(CASES):
        spec = syn.make_spec(hwc, convs, head, M, S=S, num_data=1000 + idx, seed=100 + idx, white=white,
                             conv_q_sqrt_scale=0.3, head_q_sqrt_scale=0.7)
        # move Z away from the frozen prior Z0 so the KL quirk (conv_gp/layers.py:147,150) is exercised
        rng = np.random.default_rng(500 + idx)
        for c in spec["convs"]:
            c["Z"] = c["Z0"] + 0.05 * rng.standard_normal(c["Z0"].shape)
        spec["head"]["w"] = 0.5 + rng.random(spec["head"]["w"].shape)
        X, Y = syn.make_batch(hwc, N, seed=100 + idx)
        zs = syn.make_noise(spec, N, seed=100 + idx)
        model = oracle_model(spec, X, Y)
        Fs, Fm, Fv = model.propagate(X, S=S, zs=zs)
        data, kl = model.data_term(X, Y, zs=zs), model.KL()
        elbo = model.compute_log_likelihood(X, Y, zs=zs)
        e2, d2, k2 = alt.elbo(spec, X, Y, zs, spec["num_data"])
        assert abs(elbo - e2) <= 1e-10 * abs(elbo) and abs(data - d2) <= 1e-10 * abs(data) and abs(kl - k2) <= 1e-10 * abs(kl), name
        out = flatten_spec(spec)
        out.update(X=X, Y=Y, elbo=elbo, data_term=data, kl=kl)
        for i, z in enumerate(zs):
            out["z%d" % i] = z
        for i in range(len(Fs)):
            out["Fs%d" % i], out["Fmean%d" % i], out["Fvar%d" % i] = Fs[i], Fm[i], Fv[i]
        # the training step: gradient of that ELBO with respect to every parameter group (oracle/grad.py, itself pinned by
        # finite differences in tests/test_oracle_cpu.py)
        from oracle.grad import elbo_and_grad
        eg, grads = elbo_and_grad(model, X, Y, zs)
        assert abs(eg - elbo) <= 1e-12 * abs(elbo), name
        for li, g in enumerate(grads):
            for gname, val in g.items():
                out["grad%d_%s" % (li, gname)] = np.asarray(val)
        np.savez_compressed(os.path.join(here, name + ".npz"), **out)
        print(name, "elbo", elbo, "bytes", os.path.getsize(os.path.join(here, name + ".npz")))


if __name__ == "__main__":
    main()
