"""CPU: pins the oracle (test infrastructure) -- known-answer tests, agreement with the independent
restatement oracle/alt.py, the shape facts the reference's own tests hold, and the committed golden vectors."""
import glob
import os

import numpy as np
import pytest

from deepcgp_amd import synthetic as syn
from oracle import alt
from oracle.gpflow_ref import RBF, gauss_kl, MultiClass, JITTER
from oracle.views import FullView
from oracle.layers import ConvLayer, MultiOutputConvKernel
from oracle.kernels import ConvKernel, AdditivePatchKernel
from oracle.conditionals import conditional
from oracle.dgp import SVGP_Layer
from oracle_build import oracle_model
from golden.make_golden import unflatten_spec

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "g[0-9]*.npz")))   # (ops_*.npz: tests/test_golden_ops.py)


def test_reference_test_shape_facts():
    # /root/reference/tests/test_mean_functions.py:25,35 -- FullView((28,28),5,1) has 576 patches
    v = FullView((28, 28), 5, 1)
    assert v.patch_count == 576 and v.patch_length == 25
    assert v.extract_patches_PNL(np.zeros((10, 28, 28, 1))).shape == (576, 10, 25)
    # /root/reference/tests/test_mean_functions.py:41-46 -- filter 3 on 28x28 gives 676 positions
    assert FullView((28, 28), 3, 1).patch_count == 676
    # /root/reference/tests/test_conv_kernel.py:58-68 -- Kuf is patch_count x M x N
    rng = np.random.default_rng(0)
    X = rng.standard_normal((2, 28, 28, 1))
    v3 = FullView((28, 28), 3, 1)
    mok = MultiOutputConvKernel(RBF(9), 784, v3.patch_count)
    Z = rng.standard_normal((16, 9))
    assert mok.Kuf(Z, v3.extract_patches_PNL(X)).shape == (676, 16, 2)
    # /root/reference/tests/test_conv_kernel.py:46-56 -- Kuu is M x M with rbf(Z0,Z0) + jitter on the diagonal
    Kuu = mok.Kuu(Z)
    assert Kuu.shape == (16, 16)
    np.testing.assert_allclose(np.diag(Kuu), 1.0 + JITTER, rtol=1e-14)


def test_patch_index_maps():
    # p = oh*W' + ow ; l = (kh*f + kw)*C + c  (/root/reference/tests/test_views.py:33-34 checks one patch this way)
    rng = np.random.default_rng(1)
    H, W, C, f, s = 9, 7, 3, 4, 2
    X = rng.standard_normal((2, H, W, C))
    v = FullView((H, W), f, C, s)
    pat = v.extract_patches(X)
    for (n, oh, ow, kh, kw, c) in [(0, 0, 0, 0, 0, 0), (1, 2, 1, 3, 2, 1), (0, 1, 0, 2, 3, 2)]:
        p, l = oh * v.out_image_width + ow, (kh * f + kw) * C + c
        assert pat[n, p, l] == X[n, oh * s + kh, ow * s + kw, c]
    np.testing.assert_array_equal(pat, alt.patches_NPL(X, f, s))
    np.testing.assert_array_equal(pat[0, 0], X[0, 0:f, 0:f, :].ravel())


@pytest.mark.parametrize("white", [False, True])
def test_conditional_matches_closed_form(white):
    rng = np.random.default_rng(2)
    H, W, C, f, s, M, R, N = 8, 8, 2, 3, 1, 6, 3, 4
    X = rng.standard_normal((N, H, W, C))
    v = FullView((H, W), f, C, s)
    Z = rng.standard_normal((M, v.patch_length))
    q_mu = rng.standard_normal((M, R))
    q_sqrt = np.tril(rng.standard_normal((R, M, M))) * 0.3 + np.eye(M)
    layer = ConvLayer(RBF(v.patch_length, 5.0, 5.0), None, Z, v, white=white, gp_count=R, q_mu=q_mu, q_sqrt=q_sqrt)
    m, var = layer.conditional_ND(X.reshape(N, -1))
    m2, v2 = alt.conv_layer_moments(X, f, s, Z, 5.0, 5.0, q_mu, q_sqrt, white)
    np.testing.assert_allclose(m, m2, rtol=0, atol=1e-10)
    np.testing.assert_allclose(var, v2, rtol=0, atol=1e-10)
    kp = None if white else alt.rbf(Z, Z, 5.0, 5.0) + JITTER * np.eye(M)
    np.testing.assert_allclose(layer.KL(), alt.gauss_kl(q_mu, q_sqrt, kp), rtol=1e-11)


def test_known_answers():
    rng = np.random.default_rng(3)
    H, W, C, f, s, M, R, N = 10, 10, 1, 5, 2, 12, 4, 3
    X = rng.standard_normal((N, H * W * C))
    v = FullView((H, W), f, C, s)
    Z = rng.standard_normal((M, v.patch_length))
    k = RBF(v.patch_length, 5.0, 5.0)
    # (1) not white, q_mu = 0, q_sqrt = chol(Kuu): mean = 0, var = Kdiag; KL = 0  (state of conv_gp/layers.py:154-161)
    l0 = ConvLayer(k, None, Z, v, gp_count=R)
    m, var = l0.conditional_ND(X)
    assert np.max(np.abs(m)) == 0.0
    np.testing.assert_allclose(var, 5.0, rtol=0, atol=1e-11)
    assert abs(l0.KL()) < 1e-9
    # (2) white, q_mu = 0, q_sqrt = I: var = Kdiag  (conv_gp/layers.py:89); KL = 0
    l1 = ConvLayer(k, None, Z, v, white=True, gp_count=R)
    m, var = l1.conditional_ND(X)
    np.testing.assert_allclose(var, 5.0, rtol=0, atol=1e-11)
    assert abs(l1.KL()) < 1e-12
    # (3) a patch equal to an inducing patch and q_sqrt -> 0: var -> O(jitter), mean -> row of K^-1 K . q_mu
    Xi = rng.standard_normal((1, H, W, C))
    Mz = 6
    Zp = v.extract_patches(Xi)[0, :Mz].copy()
    q_mu6 = rng.standard_normal((Mz, R))
    l2 = ConvLayer(k, None, Zp, v, gp_count=R, q_mu=q_mu6, q_sqrt=np.tile(np.eye(Mz)[None], [R, 1, 1]) * 1e-9)
    m, var = l2.conditional_ND(Xi.reshape(1, -1))
    var = var.reshape(v.patch_count, R)
    assert np.all(var[:Mz] < 5 * JITTER) and np.all(var[:Mz] > 0)
    Kuu = k.K(Zp) + JITTER * np.eye(Mz)
    np.testing.assert_allclose(m.reshape(v.patch_count, R)[:Mz], k.K(Zp) @ np.linalg.solve(Kuu, q_mu6), atol=1e-9)
    q_mu = rng.standard_normal((M, R))
    # (4) RBF Kdiag == variance, Kuu diagonal == variance + jitter
    np.testing.assert_array_equal(k.Kdiag(Z), np.full(M, 5.0))
    # (5) S-replication invariance of layer 0 (propagate tiles X S times)
    m1, v1 = ConvLayer(k, None, Z, v, gp_count=R, q_mu=q_mu).conditional_ND(np.tile(X, [3, 1]))
    np.testing.assert_array_equal(m1[:N], m1[N:2 * N])
    np.testing.assert_array_equal(v1[:N], v1[2 * N:])


def test_head_and_likelihood_against_alt():
    rng = np.random.default_rng(4)
    H, W, C, f, s, M, R, N = 7, 7, 3, 3, 1, 8, 10, 5
    X = rng.standard_normal((N, H * W * C))
    v = FullView((H, W, C), f, C, s)
    w = rng.random(v.patch_count) + 0.5
    Z = rng.standard_normal((M, v.patch_length))
    q_mu = rng.standard_normal((M, R))
    q_sqrt = np.tril(rng.standard_normal((R, M, M))) * 0.3 + np.eye(M)
    kern = ConvKernel(RBF(v.patch_length, 5.0, 5.0), v, w)
    Xi = X.reshape(N, H, W, C)
    np.testing.assert_allclose(kern.Kzx(Z, X), alt.conv_kernel_Kzx(Xi, f, s, Z, 5.0, 5.0, w), atol=1e-12)
    np.testing.assert_allclose(kern.Kdiag(X), alt.conv_kernel_Kdiag(Xi, f, s, 5.0, 5.0, w), atol=1e-12)
    add = AdditivePatchKernel(RBF(v.patch_length, 5.0, 5.0), v, w)
    np.testing.assert_allclose(add.Kzx(Z, X), kern.Kzx(Z, X), atol=1e-14)       # identical arithmetic
    np.testing.assert_allclose(add.Kdiag(X), np.full(N, 5.0 * w.mean()), atol=1e-14)
    for white in (False, True):
        head = SVGP_Layer(kern, R, Z, None, white, q_mu, q_sqrt)
        m, var = head.conditional_ND(X)
        m2, v2 = alt.svgp_head_moments(Xi, f, s, Z, 5.0, 5.0, w, q_mu, q_sqrt, white)
        np.testing.assert_allclose(m, m2, atol=1e-10)
        np.testing.assert_allclose(var, v2, atol=1e-10)
        kp = None if white else alt.rbf(Z, Z, 5.0, 5.0) + JITTER * np.eye(M)
        np.testing.assert_allclose(head.KL(), alt.gauss_kl(q_mu, q_sqrt, kp), rtol=1e-11)
        np.testing.assert_allclose(head.KL(), gauss_kl(q_mu, q_sqrt, kp), rtol=1e-12)
    mu, var, y = rng.standard_normal((6, 10)), rng.random((6, 10)) + 0.1, rng.integers(0, 10, 6)
    lik = MultiClass(10)
    np.testing.assert_allclose(lik.variational_expectations(mu, var, y), alt.robustmax_varexp(mu, var, y), rtol=1e-12)
    p, _ = lik.predict_mean_and_var(mu, var)
    assert np.all(p > 0) and np.all(p < 1)
    # one dominant class with tiny variance: every clamped cdf factor saturates at 1 - 1e-4
    mu2 = np.zeros((1, 10)); mu2[0, 3] = 50.0
    p_sat = (1 - 1e-4) ** 9
    np.testing.assert_allclose(lik.variational_expectations(mu2, np.full((1, 10), 1e-3), [3]),
                               p_sat * np.log(1 - 1e-3) + (1 - p_sat) * np.log(1e-3 / 9), rtol=1e-10)


def test_conditional_q_sqrt_rank_error():
    with pytest.raises(ValueError):
        conditional(np.zeros((1, 2, 3)), np.eye(2), np.zeros((1, 3)), np.zeros((2, 1)), q_sqrt=np.zeros((2, 2)))


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_golden_vectors_pin_the_oracle(path):
    d = np.load(path)
    spec = unflatten_spec(d)
    nl = len(spec["convs"]) + 1
    zs = [d["z%d" % i] for i in range(nl)]
    model = oracle_model(spec, d["X"], d["Y"])
    Fs, Fm, Fv = model.propagate(d["X"], S=spec["S"], zs=zs)
    for i in range(nl):
        np.testing.assert_allclose(Fs[i], d["Fs%d" % i], rtol=1e-11, atol=1e-11)
        np.testing.assert_allclose(Fm[i], d["Fmean%d" % i], rtol=1e-11, atol=1e-11)
        np.testing.assert_allclose(Fv[i], d["Fvar%d" % i], rtol=1e-11, atol=1e-11)
    np.testing.assert_allclose(model.compute_log_likelihood(d["X"], d["Y"], zs=zs), float(d["elbo"]), rtol=1e-12)
    e2, d2, k2 = alt.elbo(spec, d["X"], d["Y"], zs, spec["num_data"])
    np.testing.assert_allclose([e2, d2, k2], [float(d["elbo"]), float(d["data_term"]), float(d["kl"])], rtol=1e-10)
    # the training step's fixtures: gradient of that ELBO with respect to every parameter group
    from oracle.grad import elbo_and_grad
    _, grads = elbo_and_grad(model, d["X"], d["Y"], zs)
    for li, g in enumerate(grads):
        for name, val in g.items():
            want = d["grad%d_%s" % (li, name)]
            np.testing.assert_allclose(val, want, rtol=1e-9, atol=1e-9 * max(np.abs(want).max(), 1.0))


def test_shard_sum_equals_full_batch():
    # the multi-GPU decomposition: sum of per-shard data terms == full-batch data term (N not divisible too)
    from deepcgp_amd.dist import shard_batch, assemble_elbo
    hwc = (10, 10, 1)
    spec = syn.make_spec(hwc, [(3, 2, 3)], (3, 1), M=6, S=2, num_data=777, seed=8, conv_q_sqrt_scale=0.3)
    X, Y = syn.make_batch(hwc, 7, seed=8)
    zs = syn.make_noise(spec, 7, seed=8)
    model = oracle_model(spec, X, Y)
    full = model.compute_log_likelihood(X, Y, zs=zs)
    for world in (1, 2, 4, 8):
        total = 0.0
        for rank in range(world):
            Xs, Ys, zl = shard_batch(X, Y, zs, rank, world)
            if len(Xs):
                total += model.data_term(Xs, Ys, zs=zl)
        np.testing.assert_allclose(assemble_elbo(total, model.KL(), spec["num_data"], 7), full, rtol=1e-12)


def test_arccosine_order0_known_answers():
    """gpflow.kernels.ArcCosine(order=0) as the oracle restates it: k(x, x) ~ variance (theta = acos(1 - 1e-15)),
    orthogonal augmented inputs give variance / 2, antiparallel ones ~0; weight / bias variances enter through
    <x, z> = w x.z + b."""
    from oracle.gpflow_ref import ArcCosine
    k = ArcCosine(2, order=0, variance=3.0, weight_variances=1.0, bias_variance=0.0)
    X = np.array([[1.0, 0.0], [0.0, 2.0], [-3.0, 0.0]])
    K = k.K(X)
    assert np.allclose(np.diag(K), 3.0, rtol=0, atol=1e-7) and np.all(np.diag(K) <= 3.0)
    assert abs(K[0, 1] - 1.5) < 1e-12 and abs(K[0, 2]) < 1e-7
    assert np.allclose(K, K.T) and np.allclose(k.Kdiag(X), 3.0)
    kb = ArcCosine(2, order=0, variance=1.0, weight_variances=0.5, bias_variance=2.0)
    x, z = np.array([[1.0, 2.0]]), np.array([[-1.0, 0.5]])
    c = (0.5 * 0.0 + 2.0) / np.sqrt((0.5 * 5.0 + 2.0) * (0.5 * 1.25 + 2.0))
    assert abs(kb.K(x, z)[0, 0] - (1.0 - np.arccos(1e-15 + (1 - 2e-15) * c) / np.pi)) < 1e-15
    assert np.allclose(kb.K(x, z), kb.K(z, x).T)


@pytest.mark.parametrize("white,additive,idmean", [(False, False, False), (False, True, False), (True, False, False),
                                                   (True, True, False), (False, False, True), (False, "dense", False),
                                                   (True, "dense", False)])
def test_hand_written_gradient_against_finite_differences(white, additive, idmean):
    """oracle/grad.py (the checker of the device backward pass) against central differences of the oracle ELBO
    along random directions of every parameter group; three layers so that the sample path between layers,
    overlapping-patch scatter and both head kernels are exercised."""
    from oracle.grad import elbo_and_grad
    from oracle_build import oracle_param_handles
    hwc, seed = (10, 10, 1), 3
    dense = additive == "dense"                  # RBF(ARD=True) head on the flattened features (--last-kernel rbf)
    additive = additive is True
    spec = syn.make_spec(hwc, [(3, 1, 2), (3, 2, 2)], (2, 1), 6, S=2, num_data=100, seed=seed, white=white,
                         conv_q_sqrt_scale=0.3, variance=2.0, ls=1.5, head_kernel="rbf" if dense else "conv")
    rng = np.random.default_rng(seed)
    spec["head"]["w"] = 0.5 + rng.random(spec["head"]["w"].shape)
    if idmean:                                   # Conv2dMean on every conv layer (--identity-mean)
        for c in spec["convs"]:
            c["mean_function"] = "conv2d"
    X, Y = syn.make_batch(hwc, 3, seed=seed)
    zs = syn.make_noise(spec, 3, seed=seed)
    m = oracle_model(spec, X, Y)
    if additive:
        k = m.layers[-1].kern
        m.layers[-1].kern = AdditivePatchKernel(k.base_kernel, k.view, k.patch_weights)
    e, grads = elbo_and_grad(m, X, Y, zs)
    assert abs(e - m.compute_log_likelihood(X, Y, zs=zs)) <= 1e-12 * abs(e)
    for li, name, get, set_ in oracle_param_handles(m):
        v0 = np.array(get(), np.float64)
        g = np.asarray(grads[li][name])
        assert g.shape == v0.shape
        for _ in range(2):
            d = rng.standard_normal(v0.shape)
            if name == "q_sqrt":
                d = np.tril(d)
            h = 1e-5
            set_(v0 + h * d); ep = m.compute_log_likelihood(X, Y, zs=zs)
            set_(v0 - h * d); em = m.compute_log_likelihood(X, Y, zs=zs)
            set_(v0)
            fd, an = (ep - em) / (2 * h), float(np.sum(g * d))
            assert abs(fd - an) <= 1e-4 * max(abs(fd), abs(an), 1e-3), (li, name, fd, an)


def test_shard_gradients_sum_to_full_batch_gradient():
    """The decomposition the multi-GPU training step relies on, on the oracle: sum over batch shards of
    grad[scale * data(shard) - KL / shards] == grad of the full-batch ELBO (scale = num_data / global batch)."""
    from oracle.grad import elbo_and_grad
    hwc, N = (10, 10, 1), 4
    spec = syn.make_spec(hwc, [(3, 1, 2)], (3, 1), 6, S=2, num_data=100, seed=5, conv_q_sqrt_scale=0.3, variance=2.0, ls=1.5)
    X, Y = syn.make_batch(hwc, N, seed=5)
    zs = syn.make_noise(spec, N, seed=5)
    m = oracle_model(spec, X, Y)
    e, g = elbo_and_grad(m, X, Y, zs)
    scale = 100.0 / N
    tot_e, tot = 0.0, None
    for lo, hi in ((0, 1), (1, 4)):
        es, gs = elbo_and_grad(m, X[lo:hi], Y[lo:hi], [z[:, lo:hi] for z in zs], scale=scale, kl_weight=0.5)
        tot_e += es
        tot = gs if tot is None else [{k: a[k] + b[k] for k in a} for a, b in zip(tot, gs)]
    assert abs(tot_e - e) <= 1e-12 * abs(e)
    for a, b in zip(tot, g):
        for k in b:
            assert np.abs(a[k] - b[k]).max() <= 1e-10 * max(np.abs(b[k]).max(), 1.0), k


def test_arccosine_adjoint_against_finite_differences():
    """oracle/grad.py _acos_backward (checker of the device ArcCosine reverse pass) at kernel level: sum(dK o K) with respect
    to Z, X, variance, weight_variances, bias_variance -- cross-covariance, and K_uu with its diagonal excluded (there
    c == 1 identically and acos(1 - 1e-15 +- rounding) makes any finite difference meaningless)."""
    from oracle.gpflow_ref import ArcCosine
    from oracle.grad import _acos_backward, _acos_kuu_backward
    rng = np.random.default_rng(0)
    k = ArcCosine(9, order=0, variance=1.7, weight_variances=0.8, bias_variance=1.3)
    Z, X = rng.standard_normal((5, 9)), rng.standard_normal((7, 9))

    def check(f, grads, arrays):
        for name, g in grads.items():
            if name in arrays:
                a = arrays[name]
                d = rng.standard_normal(a.shape)
                a0 = a.copy()
                a[...] = a0 + 1e-6 * d; fp = f()
                a[...] = a0 - 1e-6 * d; fm = f()
                a[...] = a0
                fd, an = (fp - fm) / 2e-6, float(np.sum(g * d))
            else:
                v0 = getattr(k, name)
                setattr(k, name, v0 + 1e-6); fp = f()
                setattr(k, name, v0 - 1e-6); fm = f()
                setattr(k, name, v0)
                fd, an = (fp - fm) / 2e-6, float(g)
            assert abs(fd - an) <= 1e-6 * max(abs(fd), abs(an), 1.0), (name, fd, an)

    dK = rng.standard_normal((5, 7))
    dZ, dX, dv, dw, db = _acos_backward(k, Z, X, dK)
    check(lambda: float(np.sum(dK * k.K(Z, X))), {"Z": dZ, "X": dX, "variance": dv, "weight_variances": dw, "bias_variance": db},
          {"Z": Z, "X": X})
    dK2 = rng.standard_normal((5, 5))
    np.fill_diagonal(dK2, 0.0)
    dZ2, dv2, dw2, db2 = _acos_kuu_backward(k, Z, dK2)
    check(lambda: float(np.sum(dK2 * k.K(Z))), {"Z": dZ2, "variance": dv2, "weight_variances": dw2, "bias_variance": db2}, {"Z": Z})


def test_oracle_full_cov_branch_agrees_with_the_marginal_path_and_the_closed_form():
    """full_cov=True (conv_gp/conditionals.py:36-38,62-63 in the per-patch shapes its comments declare): the diagonal of every
    N x N block is the full_cov=False variance, every block is symmetric PSD-ish and equals the textbook
    Knn - Knm Kmm^-1 Kmn + Knm Kmm^-1 S Kmm^-1 Kmn; ConvLayer.conditional_ND lays it out N x N x (P*R)."""
    from oracle.conditionals import conditional
    from oracle.gpflow_ref import RBF, JITTER
    from oracle.layers import ConvLayer
    from oracle.views import FullView
    rng = np.random.default_rng(5)
    P, M, N, R, L = 3, 7, 4, 2, 5
    Z = rng.standard_normal((M, L))
    Xp = rng.standard_normal((P, N, L))
    k = RBF(L, 2.0, 1.5)
    Kmm = k.K(Z) + JITTER * np.eye(M)
    Kmn = np.stack([k.K(Z, Xp[p]) for p in range(P)])
    Knn = np.stack([k.K(Xp[p]) for p in range(P)])
    f = rng.standard_normal((M, R))
    q_sqrt = np.tril(rng.standard_normal((R, M, M))) * 0.3 + np.eye(M)[None]
    for white in (False, True):
        m_d, v_d = conditional(Kmn, Kmm, np.stack([np.diag(Knn[p]) for p in range(P)]), f, q_sqrt=q_sqrt, white=white)
        m_f, v_f = conditional(Kmn, Kmm, Knn, f, full_cov=True, q_sqrt=q_sqrt, white=white)
        assert v_f.shape == (R, P, N, N) and np.allclose(m_f, m_d, rtol=0, atol=1e-13)
        assert np.allclose(np.einsum("rpnn->rpn", v_f), v_d, rtol=0, atol=1e-12)
        assert np.allclose(v_f, np.transpose(v_f, (0, 1, 3, 2)), rtol=0, atol=1e-12)
        if not white:
            Ki = np.linalg.inv(Kmm)
            for r in range(R):
                S = q_sqrt[r] @ q_sqrt[r].T
                for p in range(P):
                    want = Knn[p] - Kmn[p].T @ Ki @ Kmn[p] + Kmn[p].T @ Ki @ S @ Ki @ Kmn[p]
                    assert np.allclose(v_f[r, p], want, rtol=0, atol=1e-10)
    view = FullView((6, 6), 3, 2, 1)
    layer = ConvLayer(RBF(view.patch_length, 2.0, 3.0), None, rng.standard_normal((5, view.patch_length)), view, gp_count=3,
                      q_mu=rng.standard_normal((5, 3)), q_sqrt=np.tril(rng.standard_normal((3, 5, 5))) * 0.2 + np.eye(5)[None])
    X = rng.standard_normal((4, 72))
    m1, v1 = layer.conditional_ND(X)
    m2, v2 = layer.conditional_ND(X, full_cov=True)
    assert v2.shape == (4, 4, layer.num_outputs) and np.allclose(m1, m2)
    assert np.allclose(np.einsum("nnd->nd", v2), v1, rtol=0, atol=1e-12)


@pytest.mark.parametrize("hwc,convs,head,M,white", [((12, 12, 1), [(3, 2, 4)], (3, 1), 24, False), ((14, 14, 2), [(4, 2, 3), (3, 1, 2)], (3, 1), 17, False),
                                                     ((10, 10, 1), [(3, 1, 3)], (3, 1), 9, True), ((12, 12, 1), [], (5, 1), 20, False)])
def test_batched_best_cpu_form_equals_the_reference_order_oracle(hwc, convs, head, M, white):
    """oracle/fast_cpu.py (bench.py's `cpu_baseline.best_cpu` row: one triangular solve over all K columns, one (R M) x M x K GEMM) is the
    same ELBO as the oracle in the reference's operation order (conv_gp/conditionals.py:29-65 under map_fn)."""
    from deepcgp_amd import synthetic as syn
    from oracle import fast_cpu
    spec = syn.make_spec(hwc, convs, head, M, S=3, num_data=500, seed=5, white=white, conv_q_sqrt_scale=0.3)
    X, Y = syn.make_batch(hwc, 5, seed=5)
    zs = syn.make_noise(spec, 5, seed=5)
    m = oracle_model(spec, X, Y)
    want = (m.compute_log_likelihood(X, Y, zs=zs), m.data_term(X, Y, zs=zs), m.KL())
    got = fast_cpu.elbo(spec, X, Y, zs=zs)
    np.testing.assert_allclose(got, want, rtol=1e-10)
