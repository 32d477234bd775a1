"""GPU parity, operator by operator: every C-ABI entry point of the hot path against the float64 oracle
on the same seeded inputs.  Tolerance: the north star asks for 1e-4 relative; the path computes in
fp64 end to end, so the tests hold it to 1e-9 relative to the tensor's scale."""
import numpy as np
import pytest

import oracle
from oracle.gpflow_ref import RBF as ORBF, ArcCosine as OArcCosine, gauss_kl as o_gauss_kl, MultiClass as OMultiClass, JITTER
from oracle.views import FullView as OFullView
from oracle.layers import MultiOutputConvKernel as OMOK, ConvLayer as OConvLayer
from oracle.kernels import ConvKernel as OConvKernel, AdditivePatchKernel as OAdd, Kuu as o_Kuu
from oracle.conditionals import conditional as o_conditional
from oracle.dgp import SVGP_Layer as OSVGP

pytestmark = pytest.mark.gpu

RTOL = 1e-9


def close(a, b, rtol=RTOL, name=""):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, (name, a.shape, b.shape)
    scale = max(np.max(np.abs(b)), 1e-30)
    err = np.max(np.abs(a - b)) / scale
    assert err <= rtol, "%s: max rel err %.3e > %.1e" % (name, err, rtol)


def rand_spd_inputs(rng, M, R, L, scale=0.5):
    Z = rng.standard_normal((M, L))
    q_mu = rng.standard_normal((M, R))
    q_sqrt = np.tril(rng.standard_normal((R, M, M))) * scale + np.eye(M)[None]
    return Z, q_mu, q_sqrt


GEOMS = [  # H, W, C, f, s
    (8, 8, 1, 3, 1), (9, 7, 3, 4, 2), (12, 12, 10, 5, 1), (28, 28, 1, 5, 2), (28, 28, 1, 5, 1), (15, 15, 10, 5, 1),
]


@pytest.mark.parametrize("H,W,C,f,s", GEOMS)
def test_extract_patches(ctx, H, W, C, f, s):
    from deepcgp_amd.views import FullView
    rng = np.random.default_rng(0)
    X = rng.standard_normal((3, H, W, C))
    v, ov = FullView((H, W), f, C, s), OFullView((H, W), f, C, s)
    assert (v.patch_count, v.patch_length) == (ov.patch_count, ov.patch_length)
    np.testing.assert_array_equal(v.extract_patches(X), ov.extract_patches(X))          # pure copy: bit exact
    np.testing.assert_array_equal(v.extract_patches_PNL(X), ov.extract_patches_PNL(X))


@pytest.mark.parametrize("M", [4, 16, 37, 64, 200, 256])
def test_kuu_potrf_trtri(ctx, M):
    from deepcgp_amd.kernels import RBF
    from deepcgp_amd.layers import _potrf
    from deepcgp_amd import device as dev
    import ctypes as C
    rng = np.random.default_rng(M)
    L = 25
    Z = rng.standard_normal((M, L)) * 2.0
    k, ok = RBF(L, 5.0, 5.0), ORBF(L, 5.0, 5.0)
    Kuu = k._gram(Z, JITTER)
    close(Kuu, ok.K(Z) + JITTER * np.eye(M), 1e-13, "Kuu")
    Lc = _potrf(Kuu)
    Lref = np.linalg.cholesky(ok.K(Z) + JITTER * np.eye(M))
    close(Lc, Lref, 1e-9, "potrf")
    assert np.all(np.triu(Lc, 1) == 0.0)
    dL, dX = ctx.to_device(Lref), ctx.empty((M, M))
    ctx._check(dev.lib().dcgp_trtri_lower(ctx.handle, dL.ptr, M, dX.ptr))
    close(dX.numpy() @ Lref, np.eye(M), 1e-9, "trtri")
    assert np.all(np.triu(dX.numpy(), 1) == 0.0)


@pytest.mark.parametrize("M", [384, 1000, 1024])
def test_kuu_potrf_trtri_realistic_large_M(ctx, M):
    """The factorisation chain at the M of BASELINE configs[3..4] (12- / 32-panel chains; 1000 = ragged last panel) on the
    realistic ill-conditioned inducing patches of synthetic.make_spec: patches cut from blurred images + N(0, 0.01^2), RBF
    variance 5, lengthscale 5, jitter 1e-3 -> cond(Kuu) ~ 1e6 at M = 1024.  Checked against LAPACK and through
    the residuals of the factor and of its inverse (the conditional applies inv(L) explicitly)."""
    from deepcgp_amd.kernels import RBF
    from deepcgp_amd.layers import _potrf
    from deepcgp_amd import device as dev
    from deepcgp_amd import synthetic as syn
    spec = syn.make_spec((28, 28, 1), [(5, 2, 10)], (5, 1), M=M, S=1, seed=77)
    Z = spec["convs"][0]["Z"]
    k, ok = RBF(25, 5.0, 5.0), ORBF(25, 5.0, 5.0)
    Kref = ok.K(Z) + JITTER * np.eye(M)
    assert np.linalg.cond(Kref) > 1e5
    Kuu = k._gram(Z, JITTER)
    close(Kuu, Kref, 1e-13, "Kuu")
    Lc = _potrf(Kuu)
    Lref = np.linalg.cholesky(Kref)
    close(Lc, Lref, 1e-9, "potrf")
    close(Lc @ Lc.T, Kref, 1e-13, "L L^T residual")
    assert np.all(np.triu(Lc, 1) == 0.0)
    dL, dX = ctx.to_device(Lref), ctx.empty((M, M))
    ctx._check(dev.lib().dcgp_trtri_lower(ctx.handle, dL.ptr, M, dX.ptr))
    Li = dX.numpy()
    import scipy.linalg
    close(Li, scipy.linalg.solve_triangular(Lref, np.eye(M), lower=True), 1e-9, "trtri vs LAPACK")
    close(Li @ Lref, np.eye(M), 1e-9, "inv(L) L residual")
    assert np.all(np.triu(Li, 1) == 0.0)


def test_potrf_not_pd(ctx):
    from deepcgp_amd.layers import _potrf
    from deepcgp_amd.device import NotPositiveDefinite
    A = np.eye(40)
    A[17, 17] = -1.0
    with pytest.raises(NotPositiveDefinite) as e:
        _potrf(A)
    assert e.value.column == 18


@pytest.mark.parametrize("H,W,C,f,s", GEOMS)
@pytest.mark.parametrize("M", [5, 64])
def test_kuf(ctx, H, W, C, f, s, M):
    from deepcgp_amd.kernels import RBF
    from deepcgp_amd.layers import MultiOutputConvKernel
    from deepcgp_amd.views import FullView
    rng = np.random.default_rng(1)
    N = 3
    X = rng.standard_normal((N, H, W, C))
    v, ov = FullView((H, W), f, C, s), OFullView((H, W), f, C, s)
    Z = rng.standard_normal((M, v.patch_length))
    mok = MultiOutputConvKernel(RBF(v.patch_length, 5.0, 5.0), H * W * C, v.patch_count)
    omok = OMOK(ORBF(v.patch_length, 5.0, 5.0), H * W * C, v.patch_count)
    ref = omok.Kuf(Z, ov.extract_patches_PNL(X))
    close(mok.Kuf(Z, (X, v)), ref, 1e-12, "Kuf fused")
    if v.patch_count * v.patch_length * 8 < 100 * 1024:
        close(mok.Kuf(Z, ov.extract_patches_PNL(X)), ref, 1e-12, "Kuf from PNL patches")


def test_sweep_exp_is_accurate_to_an_ulp_over_its_whole_range(ctx):
    """The K_uf sweep's exponential element by element: exp(-v^2 / 2) for arguments from 0 down through the subnormal range and past
    underflow.  Since round 3 the sweep folds both norms into the MFMA product and evaluates 2^t (csrc/head_units.hip), so the
    exponent carries three roundings of a base-2 value 1.44 |t| large: relative error <= (4 + 2.5 |t|) ulp -- 1e-14 where kernel
    values matter (|t| < 30; the reference's own |x|^2 + |z|^2 - 2 x.z has the same sensitivity) --, exact zeros where double
    precision underflows, exactly the variance at distance 0.  (Kuu and the parameter-only kernels keep csrc/common.h exp_sweep.)"""
    from deepcgp_amd.kernels import RBF
    from deepcgp_amd.layers import MultiOutputConvKernel
    rng = np.random.default_rng(0)
    arg = -np.concatenate([[0.0, 1e-300, 1e-12, 0.5 * np.log(2) / 64, 700.0, 708.3, 745.0, 746.5, 800.0], rng.random(4000) * 745.0,
                           rng.random(1000) * 1e-3])
    v = np.sqrt(-2.0 * arg)
    Kuf = MultiOutputConvKernel(RBF(1, 1.0, 1.0), 1, 1).Kuf(np.zeros((1, 1)), v.reshape(1, -1, 1))[0, 0]
    want = np.exp(-0.5 * v * v)
    normal = want > 1e-300
    assert np.all(np.abs(Kuf[normal] / want[normal] - 1.0) <= 2.3e-16 * (4.0 + 2.5 * np.abs(arg[normal])))
    assert np.all(np.abs(Kuf[~normal] - want[~normal]) <= 1e-307) and Kuf[arg < -765.0].max() == 0.0
    assert Kuf[0] == 1.0


@pytest.mark.parametrize("white", [False, True])
@pytest.mark.parametrize("P,M,N,R", [(3, 4, 5, 2), (6, 16, 5, 3), (2, 37, 130, 10), (4, 128, 40, 10), (1, 256, 64, 10),
                                     (16, 264, 144, 10)])   # Mp = 272: 128-row tiles overhang the matrix (zero-filled by the buffer bounds check)
def test_conditional(ctx, white, P, M, N, R):
    from deepcgp_amd.conditionals import conditional
    rng = np.random.default_rng(7 * M + P)
    L = 9
    Z, q_mu, q_sqrt = rand_spd_inputs(rng, M, R, L)
    k = ORBF(L, 5.0, 3.0)
    Kmm = k.K(Z) + JITTER * np.eye(M)
    Xp = rng.standard_normal((P, N, L))
    Kmn = np.stack([k.K(Z, Xp[p]) for p in range(P)])
    Knn = np.full((P, N), 5.0) + rng.random((P, N)) * 0.1
    q_full = q_sqrt + np.triu(rng.standard_normal((R, M, M)), 1)     # upper junk must be ignored (band_part)
    m, v = conditional(Kmn, Kmm, Knn, q_mu, q_sqrt=q_full, white=white)
    om, ov = o_conditional(Kmn, Kmm, Knn, q_mu, q_sqrt=q_sqrt, white=white)
    close(m, om, 1e-9, "mean")
    close(v, ov, 1e-9, "var")
    m2, v2 = conditional(Kmn, Kmm, Knn, q_mu, q_sqrt=None, white=white)
    om2, ov2 = o_conditional(Kmn, Kmm, Knn, q_mu, q_sqrt=None, white=white)
    close(m2, om2, 1e-9, "mean(no q_sqrt)")
    close(v2, ov2, 1e-9, "var(no q_sqrt)")


@pytest.mark.parametrize("white", [False, True])
@pytest.mark.parametrize("M", [384, 1024])
def test_conditional_large_M_realistic(ctx, white, M):
    """conditional() at M = 384 / 1024 with the ill-conditioned Kuu of the benchmark configurations and the reference's own
    initial q_sqrt = 1e-5 chol(Kuu) (models.py:136-138): the state in which var = Knn - sum A^2 + sum (Lq^T A)^2 cancels
    hardest.  The product applies explicit inverses of L where the reference solves; this pins that choice at cond ~ 1e6."""
    from deepcgp_amd.conditionals import conditional
    from deepcgp_amd import synthetic as syn
    spec = syn.make_spec((28, 28, 1), [(5, 2, 10)], (5, 1), M=M, S=1, seed=78, white=white)
    c = spec["convs"][0]
    Z, q_mu, q_sqrt = c["Z"], c["q_mu"], c["q_sqrt"]
    X, _ = syn.make_batch((28, 28, 1), 2, seed=78)
    ov = OFullView((28, 28), 5, 1, 2)
    Xp = ov.extract_patches_PNL(X.reshape(2, 28, 28, 1))[::9]          # 16 of the 144 patch positions, 2 images
    k = ORBF(25, 5.0, 5.0)
    Kmm = k.K(Z) + JITTER * np.eye(M)
    Kmn = np.stack([k.K(Z, Xp[p]) for p in range(Xp.shape[0])])
    Knn = np.full(Kmn.shape[::2], 5.0)
    m, v = conditional(Kmn, Kmm, Knn, q_mu, q_sqrt=q_sqrt, white=white)
    om, ovv = o_conditional(Kmn, Kmm, Knn, q_mu, q_sqrt=q_sqrt, white=white)
    close(m, om, 1e-8, "mean")
    close(v, ovv, 1e-8, "var")
    assert np.all(v > -1e-9)


@pytest.mark.parametrize("white", [False, True])
@pytest.mark.parametrize("P,M,N,R", [(3, 7, 4, 2), (5, 40, 9, 10), (2, 130, 33, 3)])
def test_conditional_full_cov(ctx, white, P, M, N, R):
    """full_cov=True (conv_gp/conditionals.py:36-38,62-63; the per-patch shapes its comments declare, see
    deepcgp_amd/conditionals.py): R x P x N x N against the oracle, with and without q_sqrt."""
    from deepcgp_amd.conditionals import conditional
    rng = np.random.default_rng(11 * M + P)
    L = 6
    Z, q_mu, q_sqrt = rand_spd_inputs(rng, M, R, L)
    k = ORBF(L, 5.0, 3.0)
    Kmm = k.K(Z) + JITTER * np.eye(M)
    Xp = rng.standard_normal((P, N, L))
    Kmn = np.stack([k.K(Z, Xp[p]) for p in range(P)])
    Knn = np.stack([k.K(Xp[p]) for p in range(P)])
    m, v = conditional(Kmn, Kmm, Knn, q_mu, full_cov=True, q_sqrt=q_sqrt + np.triu(rng.standard_normal((R, M, M)), 1), white=white)
    om, ov = o_conditional(Kmn, Kmm, Knn, q_mu, full_cov=True, q_sqrt=q_sqrt, white=white)
    close(m, om, 1e-9, "mean")
    close(v, ov, 1e-9, "var")
    m2, v2 = conditional(Kmn, Kmm, Knn, q_mu, full_cov=True, q_sqrt=None, white=white)
    om2, ov2 = o_conditional(Kmn, Kmm, Knn, q_mu, full_cov=True, q_sqrt=None, white=white)
    close(m2, om2, 1e-9, "mean(no q_sqrt)")
    close(v2, ov2, 1e-9, "var(no q_sqrt)")
    with pytest.raises(ValueError):
        conditional(Kmn, Kmm, Knn[:, :, 0], q_mu, full_cov=True)          # full_cov needs the P x N x N auto-covariance


@pytest.mark.parametrize("idmean", [False, True])
def test_conv_layer_full_cov(ctx, idmean):
    """ConvLayer.conditional_ND(full_cov=True) (conv_gp/layers.py:114-126): N x N x num_outputs, its diagonal the marginal variance."""
    from deepcgp_amd.kernels import RBF, PatchInducingFeatures
    from deepcgp_amd.layers import ConvLayer
    from deepcgp_amd.views import FullView
    rng = np.random.default_rng(3)
    H, W, C, f, s, M, R, N = 9, 7, 2, 3, 2, 11, 3, 5
    v, ov = FullView((H, W), f, C, s), OFullView((H, W), f, C, s)
    Z, q_mu, q_sqrt = rand_spd_inputs(rng, M, R, v.patch_length, scale=0.2)
    X = rng.standard_normal((N, H * W * C))
    layer = ConvLayer(RBF(v.patch_length, 5.0, 5.0), "conv2d" if idmean else None, PatchInducingFeatures(Z), v, gp_count=R, q_mu=q_mu, q_sqrt=q_sqrt)
    mean, var = layer.conditional_ND(X, full_cov=True)
    mean_d, var_d = layer.conditional_ND(X)
    assert var.shape == (N, N, layer.num_outputs)
    close(mean, mean_d, 1e-12, "mean")
    close(np.einsum("nnd->nd", var), var_d, 1e-9, "diagonal of the full covariance")
    if not idmean:
        olayer = OConvLayer(ORBF(v.patch_length, 5.0, 5.0), None, Z, ov, gp_count=R, q_mu=q_mu, q_sqrt=q_sqrt)
        om, ovar = olayer.conditional_ND(X, full_cov=True)
        close(mean, om, 1e-9, "mean vs oracle")
        close(var, ovar, 1e-9, "var vs oracle")


def test_sample_from_conditional_full_cov(ctx):
    """Layer.conditional_SND / sample_from_conditional(full_cov=True) (doubly_stochastic_dgp Layer + reparameterize's full_cov branch;
    the call shape of conv_gp/utils/tensorboard.py:73-81) on a ConvLayer: S x N x N x D covariance, samples through chol(var + jitter I)."""
    from deepcgp_amd.kernels import RBF, PatchInducingFeatures
    from deepcgp_amd.layers import ConvLayer
    from deepcgp_amd.views import FullView
    from oracle.dgp import sample_from_conditional as o_sample
    rng = np.random.default_rng(4)
    H, W, C, f, s, M, R, N, S = 7, 6, 2, 3, 2, 9, 2, 4, 3
    v, ov = FullView((H, W), f, C, s), OFullView((H, W), f, C, s)
    Z, q_mu, q_sqrt = rand_spd_inputs(rng, M, R, v.patch_length, scale=0.2)
    X = rng.standard_normal((S, N, H * W * C))
    layer = ConvLayer(RBF(v.patch_length, 5.0, 5.0), None, PatchInducingFeatures(Z), v, gp_count=R, q_mu=q_mu, q_sqrt=q_sqrt)
    olayer = OConvLayer(ORBF(v.patch_length, 5.0, 5.0), None, Z, ov, gp_count=R, q_mu=q_mu, q_sqrt=q_sqrt)
    z = rng.standard_normal((S, N, layer.num_outputs))
    smp, mean, var = layer.sample_from_conditional(X, z=z, full_cov=True)
    osmp, omean, ovar = o_sample(olayer, X, z=z, full_cov=True)
    assert var.shape == (S, N, N, layer.num_outputs) and smp.shape == (S, N, layer.num_outputs)
    close(mean, omean, 1e-9, "mean")
    close(var, ovar, 1e-9, "var")
    close(smp, osmp, 1e-8, "sample")
    # one input per sample: the full covariance is the marginal variance, and the sample the diagonal path's
    s1, m1, v1 = layer.sample_from_conditional(X[:, :1], z=z[:, :1], full_cov=True)
    s0, m0, v0 = layer.sample_from_conditional(X[:, :1], z=z[:, :1])
    close(v1[:, 0], v0, 1e-9, "1 x 1 covariance")
    close(s1, s0, 1e-9, "sample, N = 1")


def test_conditional_errors(ctx):
    from deepcgp_amd.conditionals import conditional
    with pytest.raises(ValueError):
        conditional(np.zeros((2, 3, 4)), np.eye(3), np.zeros((2, 4)), np.zeros((3, 1)), q_sqrt=np.zeros((3, 3)))


@pytest.mark.parametrize("white", [False, True])
@pytest.mark.parametrize("H,W,C,f,s,M,R", [(8, 8, 1, 3, 1, 4, 2), (9, 7, 3, 4, 2, 5, 3), (12, 12, 10, 5, 1, 8, 10),
                                            (28, 28, 1, 5, 2, 16, 10), (15, 15, 10, 5, 1, 48, 10)])
def test_conv_layer(ctx, white, H, W, C, f, s, M, R):
    from deepcgp_amd.kernels import RBF, PatchInducingFeatures
    from deepcgp_amd.layers import ConvLayer
    from deepcgp_amd.views import FullView
    rng = np.random.default_rng(11)
    N = 3
    X = rng.standard_normal((N, H * W * C))
    v, ov = FullView((H, W), f, C, s), OFullView((H, W), f, C, s)
    Z, q_mu, q_sqrt = rand_spd_inputs(rng, M, R, v.patch_length, 0.2)
    layer = ConvLayer(RBF(v.patch_length, 5.0, 5.0), None, PatchInducingFeatures(Z), v, white=white, gp_count=R,
                      q_mu=q_mu, q_sqrt=q_sqrt)
    olayer = OConvLayer(ORBF(v.patch_length, 5.0, 5.0), None, Z, ov, white=white, gp_count=R, q_mu=q_mu, q_sqrt=q_sqrt)
    m, var = layer.conditional_ND(X)
    om, ovar = olayer.conditional_ND(X)
    close(m, om, 1e-9, "mean")
    close(var, ovar, 1e-9, "var")
    close(layer.KL(), olayer.KL(), 1e-10, "KL")
    z = rng.standard_normal((2, N, layer.num_outputs))
    Xs = np.tile(X[None], [2, 1, 1])
    smp, sm, sv = layer.sample_from_conditional(Xs, z=z)
    close(smp, om[None] + z * np.sqrt(ovar[None] + JITTER), 1e-9, "sample")
    # default construction: q_mu = 0, q_sqrt = chol(Kuu) => mean = 0, var = Kdiag (known answer)
    if not white:
        l0 = ConvLayer(RBF(v.patch_length, 5.0, 5.0), None, PatchInducingFeatures(Z), v, gp_count=R)
        m0, v0 = l0.conditional_ND(X)
        assert np.max(np.abs(m0)) == 0.0
        close(v0, np.full_like(v0, 5.0), 1e-9, "var at init")
        assert abs(l0.KL()) < 1e-7


@pytest.mark.parametrize("white", [False, True])
@pytest.mark.parametrize("H,W,C,f,s,M,R,N", [
    (28, 28, 1, 5, 2, 256, 10, 5),     # cfg2 conv0: 64-column strips, 8 waves x 2 row fragments, strips straddling two images
    (13, 13, 10, 5, 1, 256, 10, 3),    # cfg3 conv1: L = 250, big images
    (28, 28, 1, 4, 2, 200, 10, 3),     # Mp = 208: 13 row fragments (odd), padded rows
    (32, 32, 3, 4, 2, 384, 10, 2),     # cfg4 conv0: 32-column strips, 12 waves
    (15, 15, 10, 5, 1, 384, 10, 2),    # cfg4 conv1
    (28, 28, 1, 5, 2, 500, 7, 2),      # Mp = 512: 16 waves x 2 fragments
    (28, 28, 1, 5, 2, 1024, 10, 2),    # cfg5 conv0: 16-column strips, 16 waves x 4 row fragments
    (9, 7, 3, 4, 2, 5, 3, 4),          # P = 6: many images per strip, a single (padded) row fragment
    (12, 12, 2, 3, 1, 37, 16, 3),      # R = 16
])
def test_fused_conv_layer_matches_the_unfused_route_and_the_oracle(ctx, white, H, W, C, f, s, M, R, N):
    """conv_fused.hip (the whole layer of a column strip in one workgroup) against the sweep + GEMM route it replaces
    (option no_fused_layer) and against the oracle, over every tile shape the launcher picks."""
    import os
    from deepcgp_amd.kernels import RBF, PatchInducingFeatures
    from deepcgp_amd.layers import ConvLayer
    from deepcgp_amd.views import FullView
    rng = np.random.default_rng(1000 * M + R)
    X = rng.standard_normal((N, H * W * C))
    v, ov = FullView((H, W), f, C, s), OFullView((H, W), f, C, s)
    Z, q_mu, q_sqrt = rand_spd_inputs(rng, M, R, v.patch_length, scale=0.05)
    Z *= 1.5
    layer = ConvLayer(RBF(v.patch_length, 5.0, 5.0), None, PatchInducingFeatures(Z), v, white=white, gp_count=R, q_mu=q_mu, q_sqrt=q_sqrt)
    z = rng.standard_normal((N, layer.num_outputs))
    assert ctx.get_option("no_fused_layer") == 0
    with ctx.options(fused_large=1):            # M > 256 takes the one-launch route on request only
        smp, mean, var = layer._forward(X, z)
        with ctx.options(no_fused_layer=1):
            smp_u, mean_u, var_u = layer._forward(X, z)
    close(mean, mean_u, 1e-11, "mean vs unfused")
    close(var, var_u, 1e-10, "var vs unfused")
    close(smp, smp_u, 1e-10, "sample vs unfused")
    if M <= 512:
        olayer = OConvLayer(ORBF(v.patch_length, 5.0, 5.0), None, Z, ov, white=white, gp_count=R, q_mu=q_mu, q_sqrt=q_sqrt)
        om, ovar = olayer.conditional_ND(X)
        close(mean, om, 1e-9, "mean vs oracle")
        close(var, ovar, 1e-9, "var vs oracle")


@pytest.mark.parametrize("shape", range(8))
@pytest.mark.parametrize("M,N", [(70, 3), (256, 2)])
def test_every_fused_strip_shape(ctx, shape, M, N):
    """Each instantiated strip shape of the one-launch layer kernel (csrc/conv_fused.hip kShapes: strip width x waves x teams), forced
    through the ctx option fused_shape, against the sweep + GEMM route on the same inputs -- the launcher's own choice only ever
    exercises the shapes its cost model picks (shape 7, the 16-wave 32-column strip, is picked for a rank's shard of few columns)."""
    from deepcgp_amd.kernels import RBF, PatchInducingFeatures
    from deepcgp_amd.layers import ConvLayer
    from deepcgp_amd.views import FullView
    rng = np.random.default_rng(77 + M)
    H, W, C, f, s, R = 28, 28, 1, 5, 2, 10
    X = rng.standard_normal((N, H * W * C))
    v = FullView((H, W), f, C, s)
    Z, q_mu, q_sqrt = rand_spd_inputs(rng, M, R, v.patch_length, scale=0.05)
    Z *= 1.5
    layer = ConvLayer(RBF(v.patch_length, 5.0, 5.0), None, PatchInducingFeatures(Z), v, gp_count=R, q_mu=q_mu, q_sqrt=q_sqrt)
    z = rng.standard_normal((N, layer.num_outputs))
    with ctx.options(no_fused_layer=1):
        smp_u, mean_u, var_u = layer._forward(X, z)
    with ctx.options(fused_shape=shape):
        smp, mean, var = layer._forward(X, z)
    close(mean, mean_u, 1e-11, "mean, shape %d" % shape)
    close(var, var_u, 1e-10, "var, shape %d" % shape)
    close(smp, smp_u, 1e-10, "sample, shape %d" % shape)


@pytest.mark.parametrize("shape", [0, 7])
@pytest.mark.parametrize("split", [2, 3, 4, 7])
@pytest.mark.parametrize("M,N,R", [(70, 3, 10), (256, 2, 10), (96, 2, 3)])
def test_fused_strip_shared_by_workgroups(ctx, shape, split, M, N, R):
    """The strips of the layer kernel's partial last round are shared by `fused_split` workgroups (each runs the sweep and the first product
    and the outputs r = q, q + Q, ... of the R-batched product): every output is computed by the same instructions on the same operands as
    in the whole-strip launch, so the result is bit-identical."""
    from deepcgp_amd.kernels import RBF, PatchInducingFeatures
    from deepcgp_amd.layers import ConvLayer
    from deepcgp_amd.views import FullView
    rng = np.random.default_rng(5 + M + split)
    H, W, C, f, s = 28, 28, 1, 5, 2
    X = rng.standard_normal((N, H * W * C))
    v = FullView((H, W), f, C, s)
    Z, q_mu, q_sqrt = rand_spd_inputs(rng, M, R, v.patch_length, scale=0.05)
    Z *= 1.5
    layer = ConvLayer(RBF(v.patch_length, 5.0, 5.0), None, PatchInducingFeatures(Z), v, gp_count=R, q_mu=q_mu, q_sqrt=q_sqrt)
    z = rng.standard_normal((N, layer.num_outputs))
    with ctx.options(fused_shape=shape, fused_split=0, fused_parts=0):
        smp_w, mean_w, var_w = layer._forward(X, z)
    with ctx.options(fused_shape=shape, fused_split=split, fused_parts=0):
        smp, mean, var = layer._forward(X, z)
    np.testing.assert_array_equal(mean, mean_w)
    np.testing.assert_array_equal(var, var_w)
    np.testing.assert_array_equal(smp, smp_w)


@pytest.mark.parametrize("shape", [0, 7])
@pytest.mark.parametrize("parts", [-2, 2, 3, 5, 10])
@pytest.mark.parametrize("M,N,R", [(70, 3, 10), (256, 4, 10), (96, 2, 3)])
def test_fused_strips_handed_over_and_dealt_as_parts(ctx, shape, parts, M, N, R):
    """A layer of few strips (< 1.5 rounds of the CUs: a rank's shard): every strip's sweep + first product by one item of a persistent launch, A1 handed over
    through memory, its outputs by `fused_parts` items that fetch it (part q: r = q, q + Q, ...; -2: Q chosen by the simulated deal; off by default: measured slower, tools/parts_try.py).  Same instructions on the
    same operands as the whole-strip launch: bit-identical, also launch after launch (the counters and flag epochs of the hand-over)."""
    from deepcgp_amd.kernels import RBF, PatchInducingFeatures
    from deepcgp_amd.layers import ConvLayer
    from deepcgp_amd.views import FullView
    rng = np.random.default_rng(11 + M + parts)
    H, W, C, f, s = 28, 28, 1, 5, 2
    X = rng.standard_normal((N, H * W * C))
    v = FullView((H, W), f, C, s)
    Z, q_mu, q_sqrt = rand_spd_inputs(rng, M, R, v.patch_length, scale=0.05)
    Z *= 1.5
    layer = ConvLayer(RBF(v.patch_length, 5.0, 5.0), None, PatchInducingFeatures(Z), v, gp_count=R, q_mu=q_mu, q_sqrt=q_sqrt)
    z = rng.standard_normal((N, layer.num_outputs))
    with ctx.options(fused_shape=shape, fused_split=0, fused_parts=0):
        smp_w, mean_w, var_w = layer._forward(X, z)
    for _ in range(3):
        with ctx.options(fused_shape=shape, fused_parts=parts):
            smp, mean, var = layer._forward(X, z)
        np.testing.assert_array_equal(mean, mean_w)
        np.testing.assert_array_equal(var, var_w)
        np.testing.assert_array_equal(smp, smp_w)


def test_conv_layer_identity_mean(ctx):
    from deepcgp_amd.kernels import RBF, PatchInducingFeatures
    from deepcgp_amd.layers import ConvLayer
    from deepcgp_amd.views import FullView
    rng = np.random.default_rng(3)
    H, W, C, f, s, M, R, N = 9, 9, 2, 3, 2, 6, 3, 2
    v = FullView((H, W), f, C, s)
    X = rng.standard_normal((N, H * W * C))
    Z, q_mu, q_sqrt = rand_spd_inputs(rng, M, R, v.patch_length, 0.2)
    base = ConvLayer(RBF(v.patch_length, 5.0, 5.0), None, PatchInducingFeatures(Z), v, gp_count=R, q_mu=q_mu, q_sqrt=q_sqrt)
    idm = ConvLayer(RBF(v.patch_length, 5.0, 5.0), 'conv2d', PatchInducingFeatures(Z), v, gp_count=R, q_mu=q_mu, q_sqrt=q_sqrt)
    m0, v0 = base.conditional_ND(X)
    m1, v1 = idm.conditional_ND(X)
    Xi = X.reshape(N, H, W, C)
    centre = Xi[:, 1:1 + (v.out_image_height - 1) * s + 1:s, 1:1 + (v.out_image_width - 1) * s + 1:s, 0]
    add = np.zeros((N, v.patch_count, R))
    add[:, :, 0] = centre.reshape(N, -1)                       # Conv2dMean: centre pixel of channel 0 -> map 0
    close(m1, m0 + add.reshape(N, -1), 1e-12, "identity mean")
    np.testing.assert_array_equal(v0, v1)


def test_ctx_options_and_workspace_query(ctx):
    """dcgp_ctx_set_option / dcgp_ctx_get_option (the A/B switches live in the ctx: nothing on the step path reads the environment),
    the scoped form, unknown names, the creation-only switch; dcgp_workspace_query reports what the library holds for the ctx."""
    from deepcgp_amd import device as dev
    assert ctx.get_option("no_fused_layer") == 0 and ctx.get_option("fused_shape") == -1
    with ctx.options(no_fused_layer=1, kuf_split=3):
        assert ctx.get_option("no_fused_layer") == 1 and ctx.get_option("kuf_split") == 3
        with ctx.options(no_fused_layer=0):
            assert ctx.get_option("no_fused_layer") == 0
        assert ctx.get_option("no_fused_layer") == 1
    assert ctx.get_option("no_fused_layer") == 0 and ctx.get_option("kuf_split") == -1
    with pytest.raises(dev.DcgpError):
        ctx.set_option("no_such_switch", 1)
    with pytest.raises(dev.DcgpError):
        ctx.get_option("no_such_switch")
    with pytest.raises(dev.DcgpError):
        ctx.set_option("cu_partition", 1)          # read at dcgp_ctx_create only
    before, n_before = ctx.workspace_bytes()
    rng = np.random.default_rng(3)
    from deepcgp_amd.kernels import RBF, ConvKernel
    from deepcgp_amd.views import FullView
    v = FullView((10, 10, 2), 3, 2, 1)
    ConvKernel(RBF(v.patch_length, 1.0, 1.0), v).Kzx(rng.standard_normal((7, v.patch_length)), rng.standard_normal((3, 200)))
    after, n_after = ctx.workspace_bytes()
    assert after >= before and n_after >= n_before and after > 0 and n_after > 0
    # the measured ceilings bench.py quotes (csrc/peaks.hip): sane magnitudes on an MI355X
    tf = ctx.measured_mfma_f64_tflops()
    gbs = ctx.measured_store_gbs(64, 144, 256)
    assert 40.0 < tf < 90.0 and 500.0 < gbs < 9000.0, (tf, gbs)


def test_mean_function_objects(ctx):
    """conv_gp/mean_functions.py:6-41 as objects (constructed at conv_gp/models.py:29-33,95-99): Conv2dMean(...) handed to ConvLayer gives
    what the 'conv2d' alias and the oracle's Conv2dMean give (the layer launch adds the centre pixel itself); a Conv2dMean with a changed
    filter, an IdentityConv2dMean and a plain callable go through __call__ (device convolution: dcgp_extract_patches + dcgp_gemm_strided)
    and are added to mean and sample; the model-level path refuses what it cannot fuse."""
    from deepcgp_amd.kernels import RBF, PatchInducingFeatures
    from deepcgp_amd.layers import ConvLayer
    from deepcgp_amd.mean_functions import Conv2dMean, IdentityConv2dMean, Zero
    from deepcgp_amd.views import FullView
    from oracle.mean_functions import Conv2dMean as OConv2dMean
    rng = np.random.default_rng(33)
    H, W, C, f, s, M, R, N = 9, 8, 3, 3, 2, 6, 4, 3
    v = FullView((H, W), f, C, s)
    X = rng.standard_normal((N, H * W * C))
    X4 = X.reshape(N, H, W, C)
    Z, q_mu, q_sqrt = rand_spd_inputs(rng, M, R, v.patch_length, 0.2)
    mk = lambda mf: ConvLayer(RBF(v.patch_length, 5.0, 5.0), mf, PatchInducingFeatures(Z), v, gp_count=R, q_mu=q_mu, q_sqrt=q_sqrt)   # noqa: E731
    m0, v0 = mk(None).conditional_ND(X)
    # the reference's own object == the alias == the oracle
    cm = Conv2dMean(f, C, R, stride=s)
    cm.set_trainable(False)
    m_obj, v_obj = mk(cm).conditional_ND(X)
    m_str, _ = mk('conv2d').conditional_ND(X)
    np.testing.assert_array_equal(m_obj, m_str)
    np.testing.assert_array_equal(v_obj, v0)
    close(m_obj, m0 + OConv2dMean(f, C, R, s)(X4), 1e-12, "Conv2dMean object vs oracle")
    close(cm(X4), OConv2dMean(f, C, R, s)(X4), 1e-13, "Conv2dMean.__call__ on the device")
    np.testing.assert_array_equal(mk(Zero()).conditional_ND(X)[0], m0)
    # the device convolution against a NumPy VALID convolution with an arbitrary filter
    idm = IdentityConv2dMean(f, C, R, stride=s)
    idm.conv_filter = rng.standard_normal(idm.conv_filter.shape)
    Ho, Wo = v.out_image_height, v.out_image_width
    want = np.zeros((N, Ho, Wo, R))
    for oy in range(Ho):
        for ox in range(Wo):
            want[:, oy, ox, :] = np.einsum("nabc,abcr->nr", X4[:, oy * s:oy * s + f, ox * s:ox * s + f, :], idm.conv_filter)
    close(idm(X4), want, 1e-12, "IdentityConv2dMean conv")
    # initial filter: the sum over the input channels of the centre pixel, in every output map (what models.identity_conv propagates)
    c0 = f // 2
    centre = X4[:, c0:c0 + (Ho - 1) * s + 1:s, c0:c0 + (Wo - 1) * s + 1:s, :].sum(-1)
    close(IdentityConv2dMean(f, C, R, stride=s)(X4), np.repeat(centre[..., None], R, -1), 1e-13, "identity filter")
    # anything the launch cannot add itself goes through __call__: mean and sample both move by it
    z = rng.standard_normal((N, v.patch_count * R))
    s0, _, _ = mk(None)._forward(X, z)
    s1, m1, v1 = mk(idm)._forward(X, z)
    close(m1, m0 + want.reshape(N, -1), 1e-12, "generic mean added")
    close(s1, s0 + want.reshape(N, -1), 1e-12, "generic mean added to the sample")
    np.testing.assert_array_equal(v1, v0)
    moved = Conv2dMean(f, C, R, stride=s)
    moved.conv_filter = moved.conv_filter * 2.0
    close(mk(moved).conditional_ND(X)[0], m0 + 2.0 * OConv2dMean(f, C, R, s)(X4), 1e-12, "Conv2dMean with a changed filter")
    close(mk(lambda img: np.full((img.shape[0], v.patch_count * R), 0.25)).conditional_ND(X)[0], m0 + 0.25, 1e-13, "plain callable")
    # the one-call model path takes the fused form only
    from deepcgp_amd.dgp import DGP_Base
    from deepcgp_amd.kernels import ConvKernel
    from deepcgp_amd.layers import SVGP_Layer
    from deepcgp_amd.likelihoods import MultiClass
    hv = FullView((Ho, Wo, R), 2, R, 1)
    head = SVGP_Layer(kern=ConvKernel(RBF(hv.patch_length, 1.0, 1.0), hv), num_outputs=10, feature=PatchInducingFeatures(rng.standard_normal((5, hv.patch_length))))
    Y = rng.integers(0, 10, N)
    good = DGP_Base(X, Y, likelihood=MultiClass(10), layers=[mk(cm), head], num_samples=2, minibatch_size=None, num_data=N, name="DGP")
    assert np.isfinite(good.compute_log_likelihood(X, Y, seed=1))
    good.close()
    bad = DGP_Base(X, Y, likelihood=MultiClass(10), layers=[mk(idm), head], num_samples=2, minibatch_size=None, num_data=N, name="DGP")
    with pytest.raises(ValueError):
        bad.compute_log_likelihood(X, Y, seed=1)


@pytest.mark.parametrize("H,W,C,f,s,M", [(8, 8, 1, 3, 1, 4), (12, 12, 10, 5, 1, 8), (28, 28, 1, 5, 1, 32), (9, 9, 10, 5, 1, 20),
                                          (11, 11, 10, 5, 1, 70),
                                          (40, 40, 4, 5, 3, 8)])   # an image past the unit sweep's LDS budget: the one-image-per-workgroup sweeps of rbf.hip take over
def test_head_kernels(ctx, H, W, C, f, s, M):
    from deepcgp_amd.kernels import RBF, ConvKernel, AdditivePatchKernel
    from deepcgp_amd.views import FullView
    rng = np.random.default_rng(5)
    N = 4
    X = rng.standard_normal((N, H * W * C))
    v, ov = FullView((H, W, C), f, C, s), OFullView((H, W, C), f, C, s)
    w = rng.random(v.patch_count) + 0.5
    Z = rng.standard_normal((M, v.patch_length))
    k, okk = ConvKernel(RBF(v.patch_length, 5.0, 5.0), v, w), OConvKernel(ORBF(v.patch_length, 5.0, 5.0), ov, w)
    close(k.Kzx(Z, X), okk.Kzx(Z, X), 1e-12, "Kzx")
    close(k.Kdiag(X), okk.Kdiag(X), 1e-12, "Kdiag")
    close(k.Kzz(Z), okk.Kzz(Z), 1e-13, "Kzz")
    a, oa = AdditivePatchKernel(RBF(v.patch_length, 5.0, 5.0), v, w), OAdd(ORBF(v.patch_length, 5.0, 5.0), ov, w)
    close(a.Kzx(Z, X), oa.Kzx(Z, X), 1e-12, "add Kzx")
    close(a.Kdiag(X), oa.Kdiag(X), 1e-12, "add Kdiag")


@pytest.mark.parametrize("H,W,C,f,s,M,N", [(7, 7, 1, 2, 1, 5, 3),        # L = 4: the norm slots need a sub-step of their own
                                            (6, 6, 3, 1, 1, 3, 2),        # L = 3: slots split over two sub-steps
                                            (8, 8, 2, 3, 2, 17, 5),       # L = 18, stride 2, P = 9 < 16
                                            (10, 10, 1, 5, 1, 16, 6),     # 5 x 5 x 1 patches: the register-resident form, P = 36
                                            (28, 28, 1, 5, 1, 256, 3),    # the MNIST head at full size (36 column fragments)
                                            (28, 28, 1, 5, 1, 40, 1),
                                            (13, 11, 10, 5, 1, 33, 4)])   # L = 250, P = 63, M not a multiple of 16
def test_head_unit_sweep(ctx, H, W, C, f, s, M, N):
    """ConvKernel.Kzx / Kdiag (conv_gp/kernels.py:106-133) through the unit sweep (csrc/head_units.hip): every slot layout of the
    folded norms, ragged patch and inducing counts, signed patch weights, and image rows far from everything (underflow)."""
    from deepcgp_amd.kernels import RBF, ConvKernel
    from deepcgp_amd.views import FullView
    rng = np.random.default_rng(11 * M + H)
    X = rng.standard_normal((N, H * W * C))
    X[0, : H * W * C // 2] += 6.0           # half an image far away: kernel values down to exp(-hundreds)
    v, ov = FullView((H, W, C), f, C, s), OFullView((H, W, C), f, C, s)
    w = rng.standard_normal(v.patch_count)
    Z = rng.standard_normal((M, v.patch_length))
    for var, ls in ((5.0, 5.0), (0.7, 1.3)):
        k, okk = ConvKernel(RBF(v.patch_length, var, ls), v, w), OConvKernel(ORBF(v.patch_length, var, ls), ov, w)
        # (1e-11: with the shifted half image |x|^2 c reaches several thousand, and |x|^2 + |z|^2 - 2 x.z -- the reference's own
        # square_dist form -- cancels to ~1e-12 whatever the order of the additions)
        close(k.Kzx(Z, X), okk.Kzx(Z, X), 1e-11, "Kzx")
        close(k.Kdiag(X), okk.Kdiag(X), 1e-11, "Kdiag")


@pytest.mark.parametrize("H,W,M,N", [(12, 12, 256, 5), (13, 11, 33, 4), (9, 9, 48, 3), (5, 5, 20, 6)])
def test_patch_row_form_of_the_long_patch_sweeps_is_bit_identical(ctx, H, W, M, N):
    """5 x 5 x 10 patches (every long patch of the BASELINE configurations) take the streamed sweep in its patch-row form (csrc/head_units.hip, RW: the gathers
    of a patch row are one per-lane address + immediates, the k loop straight-line code per two rows); the ctx option sweep_no_rows keeps the generic streamed
    loop.  Same operands into the same MFMAs in the same order: Kzx, Kdiag and the stored K_uf are bit-identical, and equal to the oracle's."""
    from deepcgp_amd.kernels import RBF, ConvKernel
    from deepcgp_amd.layers import MultiOutputConvKernel
    from deepcgp_amd.views import FullView
    rng = np.random.default_rng(3 * M + H)
    C, f, s = 10, 5, 1
    X = rng.standard_normal((N, H * W * C))
    v, ov = FullView((H, W, C), f, C, s), OFullView((H, W, C), f, C, s)
    w = rng.standard_normal(v.patch_count)
    Z = rng.standard_normal((M, v.patch_length)) * 0.5
    k, okk = ConvKernel(RBF(v.patch_length, 2.0, 7.0), v, w), OConvKernel(ORBF(v.patch_length, 2.0, 7.0), ov, w)
    mok = MultiOutputConvKernel(RBF(v.patch_length, 2.0, 7.0), H * W * C, v.patch_count)
    Xi = X.reshape(N, H, W, C)
    got = (k.Kzx(Z, X), k.Kdiag(X), mok.Kuf(Z, (Xi, v)))
    with ctx.options(sweep_no_rows=1):
        old = (k.Kzx(Z, X), k.Kdiag(X), mok.Kuf(Z, (Xi, v)))
    for g, o in zip(got, old):
        np.testing.assert_array_equal(g, o)
    close(got[0], okk.Kzx(Z, X), 1e-11, "Kzx")
    close(got[1], okk.Kdiag(X), 1e-11, "Kdiag")
    close(got[2], OMOK(ORBF(v.patch_length, 2.0, 7.0), H * W * C, v.patch_count).Kuf(Z, ov.extract_patches_PNL(Xi)), 1e-12, "Kuf")


@pytest.mark.parametrize("M,N,R", [(256, 3, 10), (70, 2, 3)])
def test_patch_row_form_of_the_layer_kernels_sweep_is_bit_identical(ctx, M, N, R):
    """A conv layer on 5 x 5 x 10 patches (cfg3's second layer) in the one-launch layer kernel: its in-kernel sweep walks patch rows (conv_fused.hip, the BTP = 2
    instance of the 64-column strip); ctx option sweep_no_rows keeps the generic sweep.  Bit-identical, and equal to the unfused route."""
    from deepcgp_amd.kernels import RBF, PatchInducingFeatures
    from deepcgp_amd.layers import ConvLayer
    from deepcgp_amd.views import FullView
    rng = np.random.default_rng(17 + M)
    H, W, C, f, s = 12, 12, 10, 5, 1
    X = rng.standard_normal((N, H * W * C))
    v = FullView((H, W), f, C, s)
    Z, q_mu, q_sqrt = rand_spd_inputs(rng, M, R, v.patch_length, scale=0.05)
    layer = ConvLayer(RBF(v.patch_length, 5.0, 9.0), None, PatchInducingFeatures(Z), v, gp_count=R, q_mu=q_mu, q_sqrt=q_sqrt)
    z = rng.standard_normal((N, layer.num_outputs))
    with ctx.options(fused_shape=0):
        smp, mean, var = layer._forward(X, z)
    with ctx.options(fused_shape=0, sweep_no_rows=1):
        smp_g, mean_g, var_g = layer._forward(X, z)
    with ctx.options(no_fused_layer=1):
        smp_u, mean_u, var_u = layer._forward(X, z)
    np.testing.assert_array_equal(mean, mean_g)
    np.testing.assert_array_equal(var, var_g)
    np.testing.assert_array_equal(smp, smp_g)
    close(mean, mean_u, 1e-11, "mean")
    close(var, var_u, 1e-10, "var")


def test_head_unit_sweep_exp_accuracy(ctx):
    """The unit sweep's 2^t (magic-number split + degree-11 minimax polynomial + ldexp) value by value: one patch per image
    (P = 1, weight 1), one inducing patch at the origin, so Kzx[0, n] = variance * exp(-|x_n|^2 / (2 l^2)).  The exponent reaches the
    kernel through the MFMA accumulator (both norms folded into the product: scaled pixel, square, sum -- three roundings of a
    base-2 exponent 1.44 |t| large), so its rounding error scales with |t|: relative error <= (4 + 2.5 |t|) ulp -- 1e-14 where kernel
    values matter (|t| < 30) --, exact zeros below the subnormal range, 1 ulp at t = 0."""
    from deepcgp_amd.kernels import RBF, ConvKernel
    from deepcgp_amd.views import FullView
    rng = np.random.default_rng(0)
    arg = -np.concatenate([[0.0, 1e-300, 1e-12, 0.5 * np.log(2) / 64, 700.0, 708.3, 745.0, 760.0, 800.0, 5000.0], rng.random(4000) * 745.0,
                           rng.random(1000) * 1e-3])
    X = np.zeros((arg.size, 25))
    X[:, 7] = np.sqrt(-2.0 * arg)
    v = FullView((5, 5, 1), 5, 1, 1)
    got = ConvKernel(RBF(25, 1.0, 1.0), v, np.ones(1)).Kzx(np.zeros((1, 25)), X)[0]
    want = np.exp(arg)
    normal = want > 1e-300
    rel = np.abs(got[normal] / want[normal] - 1.0)
    assert np.all(rel <= 2.3e-16 * (4.0 + 2.5 * np.abs(arg[normal]))), rel.max()
    assert np.all(np.abs(got[~normal] - want[~normal]) <= 1e-300) and got[arg < -765.0].max() == 0.0
    assert abs(got[0] - 1.0) <= 2.3e-16


@pytest.mark.parametrize("white,M", [(False, 24), (True, 24), (False, 300), (False, 500), (True, 264)])
def test_svgp_head(ctx, white, M):
    """M = 300 / 500 / 264: the one-launch head conditional with TWO row fragments per wave (M > 256: nf = 19 -- an odd count, the middle fragment
    alone on its wave -- 32 with padded rows, 17), on 23 columns (a ragged second strip)."""
    from deepcgp_amd.kernels import RBF, ConvKernel, PatchInducingFeatures
    from deepcgp_amd.layers import SVGP_Layer
    from deepcgp_amd.views import FullView
    rng = np.random.default_rng(9)
    H, W, C, f, s, R, N = 12, 12, 10, 5, 1, 10, 7 if M == 24 else 23
    X = rng.standard_normal((N, H * W * C))
    v, ov = FullView((H, W, C), f, C, s), OFullView((H, W, C), f, C, s)
    w = rng.random(v.patch_count) + 0.5
    Z, q_mu, q_sqrt = rand_spd_inputs(rng, M, R, v.patch_length, 0.3)
    layer = SVGP_Layer(ConvKernel(RBF(v.patch_length, 5.0, 5.0), v, w), R, PatchInducingFeatures(Z), None, white, q_mu, q_sqrt)
    olayer = OSVGP(OConvKernel(ORBF(v.patch_length, 5.0, 5.0), ov, w), R, Z, None, white, q_mu, q_sqrt)
    m, var = layer.conditional_ND(X)
    om, ovar = olayer.conditional_ND(X)
    close(m, om, 1e-9, "mean")
    close(var, ovar, 1e-9, "var")
    close(layer.KL(), olayer.KL(), 1e-10, "KL")
    close(layer.KL(), o_gauss_kl(q_mu, q_sqrt, None if white else o_Kuu(Z, olayer.kern, JITTER)), 1e-10, "KL == gauss_kl")


@pytest.mark.parametrize("n", [1, 7, 320])
def test_robustmax(ctx, n):
    from deepcgp_amd.likelihoods import MultiClass
    rng = np.random.default_rng(n)
    mu = rng.standard_normal((n, 10)) * 2.0
    var = rng.random((n, 10)) * 3.0
    var[0, :3] = 0.0                       # exercises the 1e-10 clip
    y = rng.integers(0, 10, n)
    lik, olik = MultiClass(10), OMultiClass(10)
    close(lik.variational_expectations(mu, var, y), olik.variational_expectations(mu, var, y), 1e-11, "varexp")
    p, pv = lik.predict_mean_and_var(mu, var)
    op, opv = olik.predict_mean_and_var(mu, var)
    close(p, op, 1e-11, "predict mean")
    close(pv, opv, 1e-11, "predict var")


# ---- ArcCosine(order = 0) base kernel of the conv layers (--base-kernel acos, conv_gp/models.py:118-119) ----
# tolerance 1e-8: theta = acos(cos) loses digits where patches are nearly parallel (cos -> 1), in the oracle and on
# the device alike -- a property of the reference's formula, not of either implementation
@pytest.mark.parametrize("H,W,C,f,s", [(8, 8, 1, 3, 1), (28, 28, 1, 5, 2), (12, 12, 10, 5, 1)])
@pytest.mark.parametrize("M", [5, 64])
def test_acos_kuu_kuf(ctx, H, W, C, f, s, M):
    from deepcgp_amd.kernels import ArcCosine
    from deepcgp_amd.layers import MultiOutputConvKernel
    from deepcgp_amd.views import FullView
    rng = np.random.default_rng(5)
    N = 3
    X = rng.standard_normal((N, H, W, C))
    v, ov = FullView((H, W), f, C, s), OFullView((H, W), f, C, s)
    Z = rng.standard_normal((M, v.patch_length))
    k, ok = ArcCosine(v.patch_length, order=0, variance=1.7, weight_variances=0.8, bias_variance=0.3), \
        OArcCosine(v.patch_length, order=0, variance=1.7, weight_variances=0.8, bias_variance=0.3)
    mok, omok = MultiOutputConvKernel(k, H * W * C, v.patch_count), OMOK(ok, H * W * C, v.patch_count)
    # Kuu: on the diagonal cos can round above the formula's 1e-15 guard (long patches) and the reference formula
    # -- hence the oracle -- returns NaN there; the device clamps the acos argument.  Compare wherever the oracle is
    # finite, require finite device values everywhere and the clamped value on the rest.
    with np.errstate(invalid="ignore"):
        ref = omok.Kuu(Z)
    got = mok.Kuu(Z)
    ok_ = np.isfinite(ref)
    assert np.all(np.isfinite(got)) and np.all(ok_ | np.eye(M, dtype=bool))
    close(np.where(ok_, got, 0.0), np.where(ok_, ref, 0.0), 1e-8, "Kuu acos")
    assert np.allclose(got[~ok_], 1.7 + JITTER, rtol=0, atol=1e-7)
    close(mok.Kuf(Z, (X, v)), omok.Kuf(Z, ov.extract_patches_PNL(X)), 1e-8, "Kuf acos")
    close(mok.Kdiag(ov.extract_patches_PNL(X)), omok.Kdiag(ov.extract_patches_PNL(X)), 1e-15, "Kdiag acos")


@pytest.mark.parametrize("white", [False, True])
def test_acos_conv_layer(ctx, white):
    from deepcgp_amd.kernels import ArcCosine, PatchInducingFeatures
    from deepcgp_amd.layers import ConvLayer
    from deepcgp_amd.views import FullView
    rng = np.random.default_rng(12)
    H, W, C, f, s, M, R, N = 12, 12, 3, 5, 1, 24, 4, 3
    X = rng.standard_normal((N, H * W * C))
    v, ov = FullView((H, W), f, C, s), OFullView((H, W), f, C, s)
    Z, q_mu, q_sqrt = rand_spd_inputs(rng, M, R, v.patch_length, 0.2)
    layer = ConvLayer(ArcCosine(v.patch_length, order=0), None, PatchInducingFeatures(Z), v, white=white, gp_count=R,
                      q_mu=q_mu, q_sqrt=q_sqrt)
    olayer = OConvLayer(OArcCosine(v.patch_length, order=0), None, Z, ov, white=white, gp_count=R, q_mu=q_mu, q_sqrt=q_sqrt)
    m, var = layer.conditional_ND(X)
    om, ovar = olayer.conditional_ND(X)
    close(m, om, 1e-8, "mean")
    close(var, ovar, 1e-8, "var")
    close(layer.KL(), olayer.KL(), 1e-8, "KL")


# ---- dense RBF-ARD head (--last-kernel rbf, conv_gp/models.py:160-168): gpflow RBF(D, ARD=True) + InducingPoints ----
@pytest.mark.parametrize("white", [False, True])
@pytest.mark.parametrize("D,M,N", [(20, 7, 5), (360, 24, 9), (1440, 48, 6)])
def test_rbf_ard_dense_head(ctx, white, D, M, N):
    from deepcgp_amd.kernels import RBF, InducingPoints
    from deepcgp_amd.layers import SVGP_Layer
    rng = np.random.default_rng(D + M)
    R = 10
    X = rng.standard_normal((N, D))
    Z = rng.standard_normal((M, D))
    ls = 3.0 * np.sqrt(D / 20.0) * (0.8 + 0.4 * rng.random(D))
    _, q_mu, q_sqrt = rand_spd_inputs(rng, M, R, D, 0.2)
    k, ok = RBF(D, 2.5, ls, ARD=True), ORBF(D, 2.5, ls, ARD=True)
    close(k.K(Z), ok.K(Z), 1e-10, "K(Z)")
    close(k.K(Z, X), ok.K(Z, X), 1e-10, "K(Z, X)")
    layer = SVGP_Layer(k, R, InducingPoints(Z), None, white=white, q_mu=q_mu, q_sqrt=q_sqrt)
    olayer = OSVGP(ok, R, Z, None, white=white, q_mu=q_mu, q_sqrt=q_sqrt)
    m, v = layer.conditional_ND(X)
    om, ov = olayer.conditional_ND(X)
    close(m, om, 1e-9, "mean")
    close(v, ov, 1e-9, "var")
    close(layer.KL(), olayer.KL(), 1e-9, "KL")


def test_kmeans_matches_numpy_lloyd(ctx):
    """dcgp_kmeans (the inducing-patch initialisation, conv_gp/kernels.py:147-164) against Lloyd's algorithm in NumPy
    from the same initial rows with the same stopping rule: same centres; and it is deterministic."""
    from deepcgp_amd import kernels as K
    rng = np.random.default_rng(0)
    n, d, k = 3000, 9, 17
    P = np.concatenate([rng.standard_normal((n // 3, d)) + c for c in (0.0, 3.0, -2.5)])
    n = P.shape[0]

    class FixedRng:
        def __init__(self):
            self.rows = np.random.default_rng(5).choice(n, size=k, replace=False)

        def choice(self, n_, size, replace):
            return self.rows
    fr = FixedRng()
    C1 = K.kmeans(P, k, rng=fr)
    C2 = K.kmeans(P, k, rng=fr)
    assert np.array_equal(C1, C2)
    C = P[fr.rows].copy()
    tol = 1e-4 * np.mean(np.var(P, axis=0))
    for _ in range(300):
        dist = (C * C).sum(1)[None, :] - 2.0 * P @ C.T
        a = dist.argmin(1)
        newC = C.copy()
        for j in range(k):
            if np.any(a == j):
                newC[j] = P[a == j].mean(0)
        shift = ((newC - C) ** 2).sum()
        C = newC
        if shift <= tol:
            break
    assert np.abs(C1 - C).max() < 1e-9


@pytest.mark.parametrize("mode", ["nn", "tn", "nt", "tt"])
def test_strided_gemm_layouts_edges_and_epilogues(ctx, mode):
    """dcgp_gemm_strided (csrc/gemm_gen.hip, every product of the reverse pass): all four operand layouts, sizes off the
    tile grid, batches, accumulation, column / k scaling, lower-triangular output, long contractions through the
    deterministic split-k path (64- and 128-tile kernels) -- against NumPy."""
    from deepcgp_amd import device as dev
    L = dev.lib()
    rng = np.random.default_rng(1)

    def run(M, N, K, batch=1, alpha=1.0, accumulate=False, colscale=False, kscale=False, lower=False):
        A = rng.standard_normal((batch, K, M) if mode[0] == "t" else (batch, M, K))
        B = rng.standard_normal((batch, N, K) if mode[1] == "t" else (batch, K, N))
        C0 = rng.standard_normal((batch, M, N + 3))                # c_rs > N: a view into a wider buffer
        cs = rng.standard_normal((batch, N)) if colscale else None
        ks = rng.standard_normal((batch, K)) if kscale else None
        dA, dB, dC = ctx.to_device(A), ctx.to_device(B), ctx.to_device(C0)
        dcs = ctx.to_device(cs) if colscale else None
        dks = ctx.to_device(ks) if kscale else None
        a_rs, a_cs = (1, M) if mode[0] == "t" else (K, 1)
        b_rs, b_cs = (1, K) if mode[1] == "t" else (N, 1)
        ctx._check(L.dcgp_gemm_strided(ctx.handle, dA.ptr, a_rs, a_cs, M * K, dB.ptr, b_rs, b_cs, K * N, dC.ptr, N + 3, M * (N + 3),
                                       M, N, K, batch, alpha, int(accumulate), dcs.ptr if colscale else None, 1, N,
                                       dks.ptr if kscale else None, 1, K, int(lower)))
        out = dC.numpy()
        for b in range(batch):
            a = A[b].T if mode[0] == "t" else A[b]
            bb = B[b].T if mode[1] == "t" else B[b]
            if kscale:
                bb = bb * ks[b][:, None]
            want = alpha * (a @ bb)
            if colscale:
                want = want * cs[b][None, :]
            if lower:
                want = np.tril(want)
            if accumulate:
                want = want + C0[b][:, :N]
            err = np.abs(out[b][:, :N] - want).max()
            assert err <= 1e-11 * max(np.abs(want).max(), 1.0) * max(K, 1) ** 0.5, (M, N, K, batch, err)
            assert np.array_equal(out[b][:, N:], C0[b][:, N:])      # nothing written beside the result
    run(1, 1, 1)
    run(37, 53, 19, batch=3, alpha=-0.5)
    run(64, 64, 16, accumulate=True)
    run(65, 129, 33, colscale=True, kscale=True)
    run(96, 96, 40, batch=2, lower=True)                        # rectangular grid, zeros above the diagonal
    run(96, 96, 40, batch=2, lower=True, accumulate=True)       # compact lower-tile grid
    run(20, 7, 5000, kscale=True)                               # split-k, 64-tile kernel
    run(130, 130, 4100, batch=2, lower=True, alpha=2.0)         # split-k, 128-tile kernel, compact grid
    run(256, 25, 9000, accumulate=True)                         # tall contraction into a narrow result


def test_strided_gemm_adjoint_epilogues(ctx):
    """dcgp_gemm_strided_ex: the row-scaled correction (E X - rowsum(E) o Z), Murray's Phi and the mirrored store of a symmetric
    product, direct store and split-k, batches, accumulation -- against NumPy."""
    from deepcgp_amd import device as dev
    L = dev.lib()
    rng = np.random.default_rng(2)

    def run(M, N, K, batch=1, alpha=1.0, accumulate=False, sub=False, flags=0, sym=False):
        A = rng.standard_normal((batch, M, K))
        B = np.transpose(A, (0, 2, 1)).copy() if sym else rng.standard_normal((batch, K, N))
        C0 = rng.standard_normal((batch, M, N + 2))
        v = rng.standard_normal((batch, M))
        Xs = rng.standard_normal((batch, M, N + 1))                # sx_rs > N
        dA, dB, dC, dv, dX = (ctx.to_device(a) for a in (A, B, C0, v, Xs))
        ctx._check(L.dcgp_gemm_strided_ex(ctx.handle, dA.ptr, K, 1, M * K, dB.ptr, N, 1, K * N, dC.ptr, N + 2, M * (N + 2), M, N, K, batch,
                                          alpha, int(accumulate), dv.ptr if sub else None, M, dX.ptr if sub else None, N + 1, M * (N + 1), flags))
        out = dC.numpy()
        for b in range(batch):
            want = A[b] @ B[b]
            if sub:
                want = want - v[b][:, None] * Xs[b][:, :N]
            want = alpha * want
            if flags & 4:
                want = np.tril(want, -1) + 0.5 * np.diag(np.diag(want))
            elif flags & 2:
                want = np.tril(want) + np.tril(want, -1).T          # the lower triangle, mirrored (== want when the product is symmetric)
            elif flags & 1:
                want = np.tril(want)
            if accumulate:
                want = want + C0[b][:, :N]
            err = np.abs(out[b][:, :N] - want).max()
            assert err <= 1e-11 * max(np.abs(want).max(), 1.0) * max(K, 1) ** 0.5, (M, N, K, batch, flags, err)
            assert np.array_equal(out[b][:, N:], C0[b][:, N:])
    run(37, 53, 19, batch=2, alpha=0.7, sub=True)
    run(256, 25, 9000, accumulate=True, sub=True)               # split-k: the correction in the reduction's epilogue
    run(70, 70, 33, batch=3, flags=4)                           # Phi, direct store
    run(40, 40, 3000, flags=4, alpha=-1.0)                      # Phi behind split-k (32-tile kernel)
    run(96, 96, 40, batch=2, flags=3, sym=True)                 # mirrored store, rectangular grid
    run(130, 130, 4100, batch=2, flags=3, sym=True, alpha=2.0)  # mirrored store, split-k, 128-tile kernel, compact grid
    with pytest.raises(dev.DcgpError):
        run(8, 9, 4, flags=1)                                    # lower_only needs a square result
