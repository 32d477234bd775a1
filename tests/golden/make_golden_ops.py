"""Generates tests/golden/ops_*.npz: the INTERMEDIATES of SURVEY.md section 8(c)'s golden-vector list for the four geometries it names,
in both whitenings -- seeded inputs X, Z, variance, lengthscale, jitter, q_mu, q_sqrt, w, Y and, from the float64 oracle (oracle/, PARITY
UNPINNED: see oracle/__init__.py): patches in both layouts, Kuu, Kuf, chol(Kuu), A = inv(L) Kuf (inv(L)^T inv(L) Kuf when not white),
the conditional's mean / var, ConvLayer.conditional_ND and KL, ConvKernel Kzx / Kdiag / Kzz, the RobustMax expectations.
tests/test_golden_ops.py asserts them against the oracle (CPU) and at the device's operator entry points (GPU).

    python tests/golden/make_golden_ops.py
"""
import os
import sys

import numpy as np
import scipy.linalg as sla

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle.gpflow_ref import RBF, gauss_kl, MultiClass, JITTER     # noqa: E402
from oracle.views import FullView                                   # noqa: E402
from oracle.layers import MultiOutputConvKernel, ConvLayer          # noqa: E402
from oracle.kernels import ConvKernel                               # noqa: E402
from oracle.conditionals import conditional                         # noqa: E402

# (H, W, C, f, s, M, R, N, S) -- SURVEY.md section 8(c)
CASES = [(8, 8, 1, 3, 1, 4, 2, 3, 2), (9, 7, 3, 4, 2, 5, 3, 2, 2), (12, 12, 10, 5, 1, 8, 10, 2, 3), (28, 28, 1, 5, 2, 16, 10, 2, 2)]
VARIANCE, LENGTHSCALE = 5.0, 5.0      # conv_gp/models.py:115-116


def case_name(c, white):
    return "ops_%dx%dx%d_f%ds%d_M%d_R%d%s" % (c[0], c[1], c[2], c[3], c[4], c[5], c[6], "_white" if white else "")


def build(c, white, seed):
    H, W, C, f, s, M, R, N, S = c
    rng = np.random.default_rng(seed)
    view = FullView((H, W), f, C, s)
    P, L = view.patch_count, view.patch_length
    X = rng.standard_normal((N, H, W, C))
    Z = rng.standard_normal((M, L))
    q_mu = rng.standard_normal((M, R))
    q_sqrt = np.tril(rng.standard_normal((R, M, M))) * 0.3 + np.eye(M)[None]
    w = 0.5 + rng.random(P)
    Y = rng.integers(0, 10, N)
    base = RBF(L, VARIANCE, LENGTHSCALE)
    out = dict(H=H, W=W, C=C, f=f, s=s, M=M, R=R, N=N, S=S, white=white, variance=VARIANCE, lengthscale=LENGTHSCALE, jitter=JITTER,
               X=X, Z=Z, q_mu=q_mu, q_sqrt=q_sqrt, w=w, Y=Y)
    # conv_gp/views.py:40-54
    out["patches_NPL"] = view.extract_patches(X)
    PNL = out["patches_PNL"] = view.extract_patches_PNL(X)
    # conv_gp/layers.py:18-32,43-50
    mok = MultiOutputConvKernel(base, L, P)
    Kuu, Kuf, Kdiag = mok.Kuu(Z), mok.Kuf(Z, PNL), mok.Kdiag(PNL)
    out.update(Kuu=Kuu, Kuf=Kuf, Kdiag=Kdiag)
    # conv_gp/conditionals.py:29-47
    Lm = np.linalg.cholesky(Kuu)
    A = np.stack([sla.solve_triangular(Lm, Kuf[p], lower=True) for p in range(P)])
    if not white:
        A = np.stack([sla.solve_triangular(Lm.T, A[p], lower=False) for p in range(P)])
    out.update(chol=Lm, A=A)
    mean, var = conditional(Kuf, Kuu, Kdiag, q_mu, q_sqrt=q_sqrt, white=white)
    out.update(cond_mean=mean, cond_var=var)
    # conv_gp/layers.py:96-147
    layer = ConvLayer(base, None, Z, view, white=white, gp_count=R, q_mu=q_mu, q_sqrt=q_sqrt)
    lm, lv = layer.conditional_ND(X.reshape(N, -1))
    out.update(layer_mean=lm, layer_var=lv, layer_KL=layer.KL(), gauss_kl=gauss_kl(q_mu, q_sqrt, None if white else Kuu))
    # conv_gp/kernels.py:106-136 on the same images (the head's view takes input_size [H, W, C], models.py:173)
    hview = FullView((H, W, C), f, C, s)
    ck = ConvKernel(base, hview, w)
    Xf = X.reshape(N, -1)
    out.update(convkernel_Kzx=ck.Kzx(Z, Xf), convkernel_Kdiag=ck.Kdiag(Xf), convkernel_Kzz=ck.Kzz(Z))
    # MultiClass(10) + RobustMax on rows built from the layer's first ten output columns (any finite means / variances do)
    cols = lm.shape[1]
    mu10 = np.stack([lm[:, (k * 7) % cols] for k in range(10)], axis=1)
    var10 = np.stack([lv[:, (k * 5) % cols] for k in range(10)], axis=1)
    out.update(lik_mu=mu10, lik_var=var10, varexp=MultiClass(10).variational_expectations(mu10, var10, Y))
    return out


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    for i, c in enumerate(CASES):
        for white in (False, True):
            name = case_name(c, white)
            out = build(c, white, seed=900 + i)     # the white twin shares its inputs
            np.savez_compressed(os.path.join(here, name + ".npz"), **out)
            print(name, "bytes", os.path.getsize(os.path.join(here, name + ".npz")))


if __name__ == "__main__":
    main()
