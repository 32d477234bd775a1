"""Generates tests/golden/checkpoint/ref_checkpoint_3layer.npy + ref_checkpoint_3layer_expected.npz.

The .npy is a checkpoint in the reference's own format -- ``np.save(path, {param.pathname: value, 'global_step': int})``
(conv_gp/experiment.py:56-64) -- whose KEY SET is exactly the path-name list a reference-trained 3-layer model prints in
notebooks/Inspect.ipynb cell 6 (M='m,m,m', strides '2,1,1', feature_maps 'a,b', filter_sizes '4,3,3', last_kernel conv), the
doubled ``DGP/likelihood/likelihood/invlink/epsilon`` of the BroadcastingLikelihood wrapper included, at toy sizes with seeded
values.  The .npz holds inputs and the oracle's outputs (ELBO parts, layer moments) for the model those parameters define, so
that the loader (deepcgp_amd.models.read_checkpoint / ModelBuilder) is tested on a file it did not write.
Run from the repo root:  python tests/golden/checkpoint/make_checkpoint_fixture.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from deepcgp_amd import synthetic as syn          # noqa: E402
from oracle_build import oracle_model             # noqa: E402

# the path names of notebooks/Inspect.ipynb cell 6 (its output lists the conv kernels' two parameters twice: once through
# layer.conv_kernel.base_kernel, once through layer.base_kernel -- the same gpflow Param)
INSPECT_CELL6 = [
    "DGP/likelihood/likelihood/invlink/epsilon",
    "DGP/layers/0/conv_kernel/base_kernel/variance", "DGP/layers/0/conv_kernel/base_kernel/lengthscales",
    "DGP/layers/0/feature/Z", "DGP/layers/0/q_mu", "DGP/layers/0/q_sqrt",
    "DGP/layers/1/conv_kernel/base_kernel/variance", "DGP/layers/1/conv_kernel/base_kernel/lengthscales",
    "DGP/layers/1/feature/Z", "DGP/layers/1/q_mu", "DGP/layers/1/q_sqrt",
    "DGP/layers/2/feature/Z", "DGP/layers/2/kern/base_kernel/variance", "DGP/layers/2/kern/base_kernel/lengthscales",
    "DGP/layers/2/kern/patch_weights", "DGP/layers/2/q_mu", "DGP/layers/2/q_sqrt",
]
HWC, CONVS, HEAD, M, N, S = (14, 14, 1), [(4, 2, 3), (3, 1, 2)], (3, 1), 6, 3, 2
FLAGS = ["--name", "fixture", "-M", "6,6,6", "--feature-maps", "3,2", "--filter-sizes", "4,3,3", "--strides", "2,1,1",
         "--num-samples", str(S), "--batch-size", str(N)]


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    rng = np.random.default_rng(2024)
    spec = syn.make_spec(HWC, CONVS, HEAD, M, S=S, num_data=40, seed=77, conv_q_sqrt_scale=0.4, head_q_sqrt_scale=0.8)
    for c in spec["convs"]:                      # a TRAINED state: everything moved away from its initial value
        c["Z"] = c["Z0"] + 0.05 * rng.standard_normal(c["Z0"].shape)
        c["variance"], c["ls"] = float(3.0 + rng.random()), float(4.0 + rng.random())
    h = spec["head"]
    h["variance"], h["ls"] = 4.25, 5.5
    h["w"] = 0.5 + rng.random(h["w"].shape)
    ckpt = {"DGP/likelihood/likelihood/invlink/epsilon": np.array(0.001), "global_step": 25000}
    for i, c in enumerate(spec["convs"]):
        base = "DGP/layers/%d/" % i
        ckpt[base + "conv_kernel/base_kernel/variance"] = np.array(c["variance"])
        ckpt[base + "conv_kernel/base_kernel/lengthscales"] = np.array(c["ls"])
        ckpt[base + "feature/Z"], ckpt[base + "q_mu"], ckpt[base + "q_sqrt"] = c["Z"], c["q_mu"], c["q_sqrt"]
    base = "DGP/layers/2/"
    ckpt[base + "feature/Z"], ckpt[base + "q_mu"], ckpt[base + "q_sqrt"] = h["Z"], h["q_mu"], h["q_sqrt"]
    ckpt[base + "kern/base_kernel/variance"], ckpt[base + "kern/base_kernel/lengthscales"] = np.array(h["variance"]), np.array(h["ls"])
    ckpt[base + "kern/patch_weights"] = h["w"]
    assert set(ckpt) == set(INSPECT_CELL6) | {"global_step"}
    np.save(os.path.join(here, "ref_checkpoint_3layer.npy"), ckpt)

    # what the reference's ModelBuilder makes of such a file: the KL prior of a conv layer is built on the Z it was CONSTRUCTED
    # with -- the loaded one (conv_gp/models.py:108-112 -> layers.py:149-152), so Z0 = Z here
    for c in spec["convs"]:
        c["Z0"] = c["Z"].copy()
    X, Y = syn.make_batch(HWC, N, seed=77)
    zs = syn.make_noise(spec, N, seed=77)
    model = oracle_model(spec, X, Y)
    Fs, Fm, Fv = model.propagate(X, S=S, zs=zs)
    out = dict(X=X, Y=Y, elbo=model.compute_log_likelihood(X, Y, zs=zs), data_term=model.data_term(X, Y, zs=zs), kl=model.KL(),
               flags=np.array(FLAGS), num_data=spec["num_data"])
    for i, z in enumerate(zs):
        out["z%d" % i] = z
    for i in range(len(Fs)):
        out["Fmean%d" % i], out["Fvar%d" % i] = Fm[i], Fv[i]
    np.savez_compressed(os.path.join(here, "ref_checkpoint_3layer_expected.npz"), **out)
    print("wrote", sorted(ckpt)[:3], "... elbo", out["elbo"])


if __name__ == "__main__":
    main()
