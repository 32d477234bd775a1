"""SURVEY.md section 8(c)'s golden-vector list, asserted at the operator entry points: tests/golden/ops_*.npz (made by
tests/golden/make_golden_ops.py from the float64 oracle; PARITY UNPINNED, see oracle/__init__.py) hold, for the four geometries the survey
names and both whitenings, the patches in both layouts, Kuu, Kuf, chol(Kuu), A, the conditional's mean / var, ConvLayer.conditional_ND
and KL, ConvKernel Kzx / Kdiag / Kzz and the RobustMax expectations.  CPU: the committed numbers still are what the oracle computes, and
what an independently written closed form gives.  GPU: every Level-1 class / C-ABI entry point reproduces them (1e-9; patches bit-exact)."""
import glob
import os

import numpy as np
import pytest

OPS = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "ops_*.npz")))
IDS = [os.path.basename(p)[4:-4] for p in OPS]


def close(a, b, rtol, name):
    a, b = np.asarray(a, float), np.asarray(b, float)
    assert a.shape == b.shape, (name, a.shape, b.shape)
    err = np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-30)
    assert err <= rtol, "%s: max rel err %.3e > %.1e" % (name, err, rtol)


def test_fixture_list_is_complete():
    assert len(OPS) == 8, OPS          # four geometries x {not white, white}
    need = {"patches_NPL", "patches_PNL", "Kuu", "Kuf", "Kdiag", "chol", "A", "cond_mean", "cond_var", "layer_mean", "layer_var", "layer_KL",
            "gauss_kl", "convkernel_Kzx", "convkernel_Kdiag", "convkernel_Kzz", "varexp", "X", "Z", "q_mu", "q_sqrt", "w", "Y"}
    for p in OPS:
        assert need <= set(np.load(p).files), p


@pytest.mark.parametrize("path", OPS, ids=IDS)
def test_fixtures_pin_the_oracle_and_a_closed_form(path):
    """The generator's own route again (a drift of the oracle shows here), and the same quantities from formulas written differently:
    patches by explicit index arithmetic (conv_gp/views.py:32-54 element order), the RBF Gram from pairwise differences, A through an
    explicit inverse, the conditional's moments in the textbook SVGP form  mean = Kfu Kuu^-1 f (or Kfu L^-T f),
    var = Kff - Kfu Kuu^-1 Kuf + || S^T a ||^2."""
    from tests_golden_gen import build_case
    d = np.load(path)
    again = build_case(d)
    for k in ("patches_NPL", "patches_PNL", "Kuu", "Kuf", "chol", "A", "cond_mean", "cond_var", "layer_mean", "layer_var", "convkernel_Kzx",
              "convkernel_Kdiag", "convkernel_Kzz", "varexp"):
        close(again[k], d[k], 1e-12, k)
    assert abs(again["layer_KL"] - float(d["layer_KL"])) <= 1e-11 * abs(float(d["layer_KL"]))
    H, W, C, f, s, M, R, N = (int(d[k]) for k in ("H", "W", "C", "f", "s", "M", "R", "N"))
    white, var0, ls, jit = bool(d["white"]), float(d["variance"]), float(d["lengthscale"]), float(d["jitter"])
    X, Z = d["X"], d["Z"]
    Ho, Wo = (H - f) // s + 1, (W - f) // s + 1
    pat = np.empty((N, Ho * Wo, f * f * C))
    for oh in range(Ho):
        for ow in range(Wo):
            pat[:, oh * Wo + ow] = X[:, oh * s:oh * s + f, ow * s:ow * s + f, :].reshape(N, -1)    # l = (kh f + kw) C + c
    np.testing.assert_array_equal(pat, d["patches_NPL"])
    np.testing.assert_array_equal(pat.transpose(1, 0, 2), d["patches_PNL"])
    rbf = lambda A_, B_: var0 * np.exp(-0.5 * ((A_[:, None, :] - B_[None, :, :]) ** 2).sum(-1) / ls ** 2)   # noqa: E731
    Kuu = rbf(Z, Z) + jit * np.eye(M)
    close(Kuu, d["Kuu"], 1e-12, "Kuu (pairwise differences)")
    Kuf = np.stack([rbf(Z, pat[:, p]) for p in range(Ho * Wo)])
    close(Kuf, d["Kuf"], 1e-11, "Kuf (pairwise differences)")
    Lc = d["chol"]
    close(Lc @ Lc.T, Kuu, 1e-12, "chol chol^T")
    Ki = np.linalg.inv(Kuu)
    A = np.stack([(np.linalg.inv(Lc) if white else Ki) @ Kuf[p] for p in range(Ho * Wo)])
    close(A, d["A"], 1e-8, "A (explicit inverse)")
    S = np.tril(d["q_sqrt"])
    mean = np.stack([A[p].T @ d["q_mu"] for p in range(Ho * Wo)]).transpose(1, 0, 2)              # N x P x R
    close(mean, d["cond_mean"], 1e-8, "conditional mean (textbook)")
    var = np.empty((R, Ho * Wo, N))
    for p in range(Ho * Wo):
        base = var0 - np.einsum("mn,mk,kn->n", Kuf[p], Ki, Kuf[p])
        for r in range(R):
            var[r, p] = base + ((S[r].T @ A[p]) ** 2).sum(0)
    close(var, d["cond_var"], 1e-7, "conditional var (textbook)")
    close(d["layer_mean"], mean.reshape(N, -1), 1e-8, "ConvLayer mean = conditional mean, N x (P R)")
    close(d["layer_var"], var.transpose(2, 1, 0).reshape(N, -1), 1e-7, "ConvLayer var")
    # gauss_kl, closed form
    Kp = None if white else Kuu
    kl = 0.0
    for r in range(R):
        Sig = S[r] @ S[r].T
        mu = d["q_mu"][:, r]
        if Kp is None:
            kl += 0.5 * (np.trace(Sig) + mu @ mu - M - np.linalg.slogdet(Sig)[1])
        else:
            kl += 0.5 * (np.trace(Ki @ Sig) + mu @ Ki @ mu - M + np.linalg.slogdet(Kp)[1] - np.linalg.slogdet(Sig)[1])
    assert abs(kl - float(d["gauss_kl"])) <= 1e-8 * abs(kl)


@pytest.mark.gpu
@pytest.mark.parametrize("path", OPS, ids=IDS)
def test_operator_entry_points_reproduce_the_fixtures(ctx, path):
    from deepcgp_amd.kernels import RBF, ConvKernel, PatchInducingFeatures
    from deepcgp_amd.layers import MultiOutputConvKernel, ConvLayer, _potrf
    from deepcgp_amd.conditionals import conditional
    from deepcgp_amd.likelihoods import MultiClass
    from deepcgp_amd.views import FullView
    d = np.load(path)
    H, W, C, f, s, M, R, N = (int(d[k]) for k in ("H", "W", "C", "f", "s", "M", "R", "N"))
    white, var0, ls, jit = bool(d["white"]), float(d["variance"]), float(d["lengthscale"]), float(d["jitter"])
    X, Z, q_mu, q_sqrt, w = d["X"], d["Z"], d["q_mu"], d["q_sqrt"], d["w"]
    view = FullView((H, W), f, C, s)
    L = view.patch_length
    np.testing.assert_array_equal(view.extract_patches(X), d["patches_NPL"])             # dcgp_extract_patches: a copy, bit exact
    np.testing.assert_array_equal(view.extract_patches_PNL(X), d["patches_PNL"])
    base = RBF(L, var0, ls)
    mok = MultiOutputConvKernel(base, H * W * C, view.patch_count)
    Kuu = mok.Kuu(Z)                                                                     # dcgp_kuu_rbf
    close(Kuu, d["Kuu"], 1e-12, "Kuu")
    close(mok.Kuf(Z, (X, view)), d["Kuf"], 1e-11, "Kuf (patch gather fused)")            # dcgp_kuf_patches_rbf
    close(mok.Kuf(Z, d["patches_PNL"]), d["Kuf"], 1e-11, "Kuf (from patches)")
    close(mok.Kdiag(d["patches_PNL"]), d["Kdiag"], 1e-13, "Kdiag")
    close(_potrf(d["Kuu"]), d["chol"], 1e-10, "chol")                                    # dcgp_potrf_lower
    m, v = conditional(d["Kuf"], d["Kuu"], d["Kdiag"], q_mu, q_sqrt=q_sqrt, white=white)  # dcgp_conditional
    close(m, d["cond_mean"], 1e-9, "conditional mean")
    close(v, d["cond_var"], 1e-9, "conditional var")
    # A itself is never formed on the device (the conditional re-associated); what it determines is: mean = A^T q_mu
    close(np.einsum("pmn,mr->npr", d["A"], q_mu), m, 1e-9, "A^T q_mu")
    layer = ConvLayer(base, None, PatchInducingFeatures(Z), view, white=white, gp_count=R, q_mu=q_mu, q_sqrt=q_sqrt)
    lm, lv = layer.conditional_ND(X.reshape(N, -1))                                      # dcgp_conv_layer_forward
    close(lm, d["layer_mean"], 1e-9, "layer mean")
    close(lv, d["layer_var"], 1e-9, "layer var")
    assert abs(layer.KL() - float(d["layer_KL"])) <= 1e-10 * abs(float(d["layer_KL"]))   # dcgp_gauss_kl
    hview = FullView((H, W, C), f, C, s)
    ck = ConvKernel(base, hview, w)
    Xf = X.reshape(N, -1)
    close(ck.Kzx(Z, Xf), d["convkernel_Kzx"], 1e-11, "ConvKernel.Kzx")                  # dcgp_convkernel_kzx
    close(ck.Kdiag(Xf), d["convkernel_Kdiag"], 1e-11, "ConvKernel.Kdiag")                # dcgp_convkernel_kdiag
    close(ck.Kzz(Z), d["convkernel_Kzz"], 1e-12, "ConvKernel.Kzz")
    close(MultiClass(10).variational_expectations(d["lik_mu"], d["lik_var"], d["Y"]), d["varexp"], 1e-11, "varexp")   # dcgp_robustmax_varexp
