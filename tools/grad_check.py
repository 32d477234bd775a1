"""Debug helper: per-parameter gradient error of dcgp_elbo_grad against oracle/grad.py (needs a GPU)."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from deepcgp_amd import synthetic as syn
from deepcgp_amd.models import build_from_spec
from oracle_build import oracle_model
from oracle.grad import elbo_and_grad

white = len(sys.argv) > 1 and sys.argv[1] == "white"
hwc, N, S = (14, 14, 1), 3, 2
spec = syn.make_spec(hwc, [(3, 1, 3), (4, 2, 2)], (3, 1), 20, S=S, num_data=500, seed=9, white=white,
                     conv_q_sqrt_scale=0.3, variance=2.0, ls=1.5)
rng = np.random.default_rng(9)
spec["head"]["w"] = 0.5 + rng.random(spec["head"]["w"].shape)
X, Y = syn.make_batch(hwc, N, seed=9)
zs = syn.make_noise(spec, N, seed=9)
ref = oracle_model(spec, X, Y)
model = build_from_spec(spec, X, Y)
e, grads = model.compute_gradients(X, Y, zs=zs)
eo, go = elbo_and_grad(ref, X, Y, zs)
print("elbo", e, eo)
for li, (g, o) in enumerate(zip(grads, go)):
    for name, val in o.items():
        scale = max(np.abs(val).max(), 1e-6)
        d = np.abs(g[name] - val)
        print(li, name, "max|ref|=%.3e" % np.abs(val).max(), "err=%.3e" % (d.max() / scale), "argmax", np.unravel_index(d.argmax(), d.shape) if d.ndim else ())
        if name == "q_sqrt" and d.max() / scale > 1e-8:
            r = np.unravel_index(d.argmax(), d.shape)[0]
            print("  diag dev", np.diag(g[name][r])[:5], "ref", np.diag(val[r])[:5])
            off = d[r] - np.diag(np.diag(d[r]))
            print("  offdiag max err", off.max(), " diag max err", np.diag(d[r]).max())
