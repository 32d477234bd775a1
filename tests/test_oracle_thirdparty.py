"""CPU: the oracle's restatements of the un-vendored dependencies (GPflow 1.2.0, doubly_stochastic_dgp -- SURVEY.md App. A, "parity
unpinned") against INDEPENDENT third-party code that ships in this image: torch.distributions / torch.linalg (PyTorch, CPU),
sklearn.gaussian_process.kernels, and plain Monte-Carlo estimates of the quantities' definitions.  None of these share code or
authorship with oracle/ (or with oracle/alt.py, the builder's own second restatement), so agreement here is evidence about the
FORMULAS: a recalled-wrong sign, factor or axis in the oracle fails these tests even though the HIP path agrees with the oracle.

What is checked, and against what:
  gauss_kl (conv_gp/layers.py:145,147)                 torch.distributions.kl_divergence(MultivariateNormal, MultivariateNormal)
  RBF.K / Kdiag (layers.py:20,29; kernels.py:114-136)  sklearn.gaussian_process.kernels.RBF (scalar and per-dimension length scales)
  ArcCosine(order 0) (models.py:118-119)               Monte-Carlo 2 E_w[step(w.x~) step(w.z~)], w ~ N(0, I)  (Cho & Saul 2009, eq. 1)
  RobustMax prob_is_largest / variational_expectations Monte-Carlo P(f_y = max f), E_q[log p(y | f)] over 10^6 draws
  MultiClass.predict_mean_and_var                      Monte-Carlo class probabilities
  SVGP_Layer marginals (models.py:192-198)             torch.linalg closed form  k - k^T K^-1 k + k^T K^-1 S K^-1 k
  conditional() (conv_gp/conditionals.py:6-67)         the same closed form per patch (dense solves, no Cholesky), both whitenings
  reparameterize / DGP ELBO assembly                   torch.distributions.Normal rsample algebra; sum_n E/S * num_data/N - sum KL
"""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

from oracle.gpflow_ref import RBF, ArcCosine, MultiClass, gauss_kl, JITTER          # noqa: E402
from oracle.conditionals import conditional                                         # noqa: E402
from oracle.dgp import SVGP_Layer, reparameterize, sample_from_conditional                                   # noqa: E402
from oracle.kernels import ConvKernel                                               # noqa: E402
from oracle.views import FullView                                                   # noqa: E402

T = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float64)                   # noqa: E731


def _spd(rng, M, jitter=0.1):
    A = rng.standard_normal((M, M))
    return A @ A.T / M + jitter * np.eye(M)


@pytest.mark.parametrize("white", [False, True])
def test_gauss_kl_against_torch_distributions(white):
    """KL[N(q_mu_r, Lq_r Lq_r^T) || N(0, K)] summed over the R outputs; K = I when whitened (K=None)."""
    from torch.distributions import MultivariateNormal, kl_divergence
    rng = np.random.default_rng(10 + white)
    M, R = 7, 3
    K = _spd(rng, M)
    q_mu = rng.standard_normal((M, R))
    q_sqrt = np.stack([np.linalg.cholesky(_spd(rng, M)) for _ in range(R)])
    q_sqrt_full = q_sqrt + np.triu(rng.standard_normal((R, M, M)), 1)      # gpflow band-parts q_sqrt: the upper triangle must not matter
    prior = MultivariateNormal(torch.zeros(M, dtype=torch.float64), covariance_matrix=T(np.eye(M) if white else K))
    want = sum(float(kl_divergence(MultivariateNormal(T(q_mu[:, r]), scale_tril=T(q_sqrt[r])), prior)) for r in range(R))
    got = gauss_kl(q_mu, q_sqrt_full, None if white else K)
    assert abs(got - want) <= 1e-10 * abs(want), (got, want)
    # q == p: zero
    L = np.linalg.cholesky(K)
    assert abs(gauss_kl(np.zeros((M, R)), np.tile((np.eye(M) if white else L)[None], (R, 1, 1)), None if white else K)) < 1e-10


def test_rbf_against_sklearn():
    from sklearn.gaussian_process.kernels import RBF as SkRBF
    rng = np.random.default_rng(1)
    X, Z = rng.standard_normal((9, 5)) * 2.0, rng.standard_normal((4, 5))
    k = RBF(5, variance=2.5, lengthscales=1.7)
    np.testing.assert_allclose(k.K(Z, X), 2.5 * SkRBF(length_scale=1.7)(Z, X), rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(k.K(X), 2.5 * SkRBF(length_scale=1.7)(X), rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(k.Kdiag(X), 2.5 * SkRBF(length_scale=1.7).diag(X), rtol=1e-15)
    ls = np.array([0.5, 1.0, 2.0, 3.0, 5.0])
    ka = RBF(5, variance=0.8, lengthscales=ls, ARD=True)                    # the dense head's kernel, conv_gp/models.py:163-164
    np.testing.assert_allclose(ka.K(Z, X), 0.8 * SkRBF(length_scale=ls)(Z, X), rtol=1e-12, atol=1e-14)


def test_convkernel_against_sklearn_and_loops():
    """ConvKernel.Kzx / Kdiag / Kzz (conv_gp/kernels.py:106-136) written out as loops over patches cut by hand, base kernel from sklearn."""
    from sklearn.gaussian_process.kernels import RBF as SkRBF
    rng = np.random.default_rng(2)
    H, W, C, f, s, M, N = 7, 6, 2, 3, 2, 4, 3
    X = rng.standard_normal((N, H, W, C))
    Z = rng.standard_normal((M, f * f * C))
    w = rng.standard_normal(((H - f) // s + 1) * ((W - f) // s + 1))
    kern = ConvKernel(RBF(f * f * C, 1.3, 2.2), FullView((H, W, C), f, C, s), patch_weights=w)
    sk = lambda A, B: 1.3 * SkRBF(length_scale=2.2)(A, B)                   # noqa: E731
    patches = np.array([[X[n, oy:oy + f, ox:ox + f, :].ravel() for oy in range(0, H - f + 1, s) for ox in range(0, W - f + 1, s)]
                        for n in range(N)])                                  # N x P x L, p = oh W' + ow, l = (kh f + kw) C + c
    P = patches.shape[1]
    Kzx = np.stack([sum(w[p] * sk(Z, patches[n, p:p + 1])[:, 0] for p in range(P)) / P for n in range(N)], axis=1)
    Kd = np.array([sum(w[p] * w[q] * sk(patches[n, p:p + 1], patches[n, q:q + 1])[0, 0] for p in range(P) for q in range(P)) / P ** 2
                   for n in range(N)])
    Xf = X.reshape(N, -1)
    np.testing.assert_allclose(kern.Kzx(Z, Xf), Kzx, rtol=1e-11, atol=1e-13)
    np.testing.assert_allclose(kern.Kdiag(Xf), Kd, rtol=1e-11, atol=1e-13)
    np.testing.assert_allclose(kern.Kzz(Z), sk(Z, Z), rtol=1e-12)


def test_arccosine_order0_against_monte_carlo():
    """k_0(x, z) = 1 - theta / pi = 2 E_w[Theta(w . x~) Theta(w . z~)], w ~ N(0, I) (Cho & Saul 2009), with gpflow's augmented inputs
    x~ = (sqrt(weight_variances) x, sqrt(bias_variance)); times the kernel variance."""
    rng = np.random.default_rng(3)
    D, wv, bv, var = 4, 0.7, 0.4, 1.9
    X, Z = rng.standard_normal((3, D)), rng.standard_normal((2, D))
    k = ArcCosine(D, order=0, variance=var, weight_variances=wv, bias_variance=bv)
    aug = lambda A: np.concatenate([np.sqrt(wv) * A, np.full((A.shape[0], 1), np.sqrt(bv))], axis=1)   # noqa: E731
    n = 2_000_000
    Wd = rng.standard_normal((n, D + 1))
    hx, hz = (Wd @ aug(X).T) > 0, (Wd @ aug(Z).T) > 0
    mc = var * 2.0 * (hz.T.astype(np.float64) @ hx.astype(np.float64)) / n            # [Z, X]
    assert np.abs(k.K(Z, X) - mc).max() < 5 * var * 0.5 / np.sqrt(n) + 1e-3, (k.K(Z, X), mc)
    np.testing.assert_allclose(k.Kdiag(X), var)                                     # theta = 0 on the diagonal


def _mc_class_probs(rng, mu, var, n):
    f = mu[None] + np.sqrt(var)[None] * rng.standard_normal((n, mu.shape[0]))
    return np.bincount(f.argmax(1), minlength=mu.shape[0]) / n


def test_robustmax_against_monte_carlo():
    """prob_is_largest = P(f_y = max_k f_k), f ~ N(mu, diag(var)); variational_expectations = E_q[log p(y | f)] with
    p(y | f) = 1 - eps if y = argmax f else eps / (K - 1); predict_mean_and_var mixes the same probabilities with eps."""
    rng = np.random.default_rng(4)
    lik = MultiClass(10)
    n = 1_000_000
    for case in range(3):
        mu = rng.standard_normal(10) * (0.5 + case)
        var = rng.uniform(0.05, 2.0, 10)
        pmc = _mc_class_probs(rng, mu, var, n)
        tol = 5 * np.sqrt(0.25 / n) + 1.5e-3          # 5 sigma of the estimate + gpflow's own cdf clamp (x (1 - 2e-4) + 1e-4, nine factors)
        for y in (int(np.argmax(mu)), int(np.argmin(mu)), 3):
            p = lik.prob_is_largest(np.array([y]), mu[None], var[None])[0]
            assert abs(p - pmc[y]) < tol, (case, y, p, pmc[y])
            ve = lik.variational_expectations(mu[None], var[None], np.array([y]))[0]
            ve_mc = pmc[y] * np.log(1 - 1e-3) + (1 - pmc[y]) * np.log(1e-3 / 9)
            assert abs(ve - ve_mc) < tol * abs(np.log(1e-3 / 9)), (ve, ve_mc)
        pm, pv = lik.predict_mean_and_var(mu[None], var[None])
        want = pmc * (1 - 1e-3) + (1 - pmc) * 1e-3 / 9
        assert np.abs(pm[0] - want).max() < tol and abs(pm.sum() - 1.0) < 5e-3
        np.testing.assert_allclose(pv, pm - pm ** 2)


def _dense_marginals(Kmm, Kmn, knn, q_mu, q_sqrt, white):
    """q(f_n) of an SVGP with q(u) = N(q_mu, S): torch.linalg dense solves, no Cholesky of K anywhere on the un-whitened branch.
    white: u = L v with q(v) = N(q_mu, S)."""
    Kmm, Kmn, knn, q_mu, q_sqrt = T(Kmm), T(Kmn), T(knn), T(q_mu), T(q_sqrt)
    S = q_sqrt.tril() @ q_sqrt.tril().transpose(-1, -2)                       # R x M x M
    if white:
        L = torch.linalg.cholesky(Kmm)
        mean_u = L @ q_mu
        S = L @ S @ L.T
    else:
        mean_u = q_mu
    B = torch.linalg.solve(Kmm, Kmn)                                           # K^-1 k, M x N
    mean = B.T @ mean_u                                                        # N x R
    var = knn[None, :] - (Kmn * B).sum(0)[None, :] + torch.einsum("mn,rmk,kn->rn", B, S, B)
    return mean.numpy(), var.numpy()                                           # N x R, R x N


@pytest.mark.parametrize("white", [False, True])
def test_conv_conditional_against_torch_closed_form(white):
    """conditional() (conv_gp/conditionals.py:6-67) patch by patch: mean N x P x R, var R x P x N."""
    rng = np.random.default_rng(5 + white)
    P, M, N, R = 3, 6, 4, 2
    Z = rng.standard_normal((M, 5))
    Xp = rng.standard_normal((P, N, 5))
    k = RBF(5, 1.4, 1.8)
    Kmm = k.K(Z) + JITTER * np.eye(M)
    Kmn = np.stack([k.K(Z, Xp[p]) for p in range(P)])
    Knn = np.stack([k.Kdiag(Xp[p]) for p in range(P)])
    q_mu = rng.standard_normal((M, R))
    q_sqrt = np.stack([np.linalg.cholesky(_spd(rng, M)) for _ in range(R)]) * 0.3
    mean, var = conditional(Kmn, Kmm, Knn, q_mu, q_sqrt=q_sqrt, white=white)
    assert mean.shape == (N, P, R) and var.shape == (R, P, N)
    for p in range(P):
        m, v = _dense_marginals(Kmm, Kmn[p], Knn[p], q_mu, q_sqrt, white)
        np.testing.assert_allclose(mean[:, p, :], m, rtol=1e-9, atol=1e-11)
        np.testing.assert_allclose(var[:, p, :], v, rtol=1e-8, atol=1e-10)


@pytest.mark.parametrize("white", [False, True])
def test_conv_conditional_full_cov_against_torch_closed_form(white):
    """conditional(full_cov=True) (conv_gp/conditionals.py:36-38,62-63 in the per-patch form its comments declare): for every patch and output the
    N x N covariance  K_ff - K_fu K^-1 K_uf + K_fu K^-1 S K^-1 K_uf  by dense torch.linalg solves; its diagonal is the full_cov=False variance."""
    rng = np.random.default_rng(15 + white)
    P, M, N, R = 2, 6, 5, 2
    Z = rng.standard_normal((M, 4))
    Xp = rng.standard_normal((P, N, 4))
    k = RBF(4, 1.3, 1.6)
    Kmm = k.K(Z) + JITTER * np.eye(M)
    Kmn = np.stack([k.K(Z, Xp[p]) for p in range(P)])
    Kff = np.stack([k.K(Xp[p]) for p in range(P)])
    q_mu = rng.standard_normal((M, R))
    q_sqrt = np.stack([np.linalg.cholesky(_spd(rng, M)) for _ in range(R)]) * 0.4
    mean, var = conditional(Kmn, Kmm, Kff, q_mu, full_cov=True, q_sqrt=q_sqrt, white=white)
    assert mean.shape == (N, P, R) and var.shape == (R, P, N, N)
    _, var_diag = conditional(Kmn, Kmm, np.stack([k.Kdiag(Xp[p]) for p in range(P)]), q_mu, q_sqrt=q_sqrt, white=white)
    Kt, St = T(Kmm), T(q_sqrt).tril() @ T(q_sqrt).tril().transpose(-1, -2)
    if white:
        Lt = torch.linalg.cholesky(Kt)
        St = Lt @ St @ Lt.T
    for p in range(P):
        B = torch.linalg.solve(Kt, T(Kmn[p]))                                   # K^-1 K_uf
        for r in range(R):
            want = T(Kff[p]) - T(Kmn[p]).T @ B + B.T @ St[r] @ B
            np.testing.assert_allclose(var[r, p], want.numpy(), rtol=1e-8, atol=1e-10)
            np.testing.assert_allclose(np.diag(var[r, p]), var_diag[r, p], rtol=1e-9, atol=1e-11)


@pytest.mark.parametrize("white", [False, True])
def test_svgp_layer_against_torch_closed_form(white):
    """SVGP_Layer.conditional_ND (models.py:192-198) with the ConvKernel: Kuu = Kzz + jitter I, Kuf = Kzx, Kdiag; KL against K_uu."""
    from torch.distributions import MultivariateNormal, kl_divergence
    rng = np.random.default_rng(7 + white)
    H, W, C, f, s, M, N, R = 6, 6, 1, 3, 1, 5, 4, 3
    X = rng.standard_normal((N, H * W * C))
    Z = rng.standard_normal((M, f * f * C))
    kern = ConvKernel(RBF(f * f * C, 2.0, 1.5), FullView((H, W, C), f, C, s), patch_weights=rng.uniform(0.5, 1.5, 16))
    q_mu = rng.standard_normal((M, R))
    q_sqrt = np.stack([np.linalg.cholesky(_spd(rng, M)) for _ in range(R)]) * 0.5
    layer = SVGP_Layer(kern, R, Z, white=white, q_mu=q_mu, q_sqrt=q_sqrt)
    mean, var = layer.conditional_ND(X)
    Kmm = kern.Kzz(Z) + JITTER * np.eye(M)
    m, v = _dense_marginals(Kmm, kern.Kzx(Z, X), kern.Kdiag(X), q_mu, q_sqrt, white)
    np.testing.assert_allclose(mean, m, rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(var, v.T, rtol=1e-8, atol=1e-10)
    prior = MultivariateNormal(torch.zeros(M, dtype=torch.float64), covariance_matrix=T(np.eye(M) if white else Kmm))
    want = sum(float(kl_divergence(MultivariateNormal(T(q_mu[:, r]), scale_tril=T(q_sqrt[r])), prior)) for r in range(R))
    assert abs(layer.KL() - want) <= 1e-9 * abs(want)


def test_reparameterize_and_elbo_assembly_against_torch():
    """reparameterize = Normal(mean, sqrt(var + jitter)).rsample with the noise given; the ELBO of a model = sum_n mean_s E_q[log p(y|f)] *
    num_data / N - sum_l KL_l (Salimbeni & Deisenroth 2017, eq. 11) -- assembled here from the third-party pieces above."""
    import sys
    import os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from oracle_build import oracle_model
    from deepcgp_amd import synthetic as syn
    rng = np.random.default_rng(9)
    mean, var, z = rng.standard_normal((2, 3, 4)), rng.uniform(0.1, 1.0, (2, 3, 4)), rng.standard_normal((2, 3, 4))
    want = (T(mean) + T(z) * torch.sqrt(T(var) + JITTER)).numpy()
    np.testing.assert_allclose(reparameterize(mean, var, z), want, rtol=1e-15)
    hwc = (8, 8, 1)
    spec = syn.make_spec(hwc, [(3, 1, 2)], (3, 1), M=5, S=3, num_data=123, seed=3, conv_q_sqrt_scale=0.5)
    X, Y = syn.make_batch(hwc, 4, seed=3)
    zs = syn.make_noise(spec, 4, seed=3)
    model = oracle_model(spec, X, Y)
    elbo = model.compute_log_likelihood(X, Y, zs=zs)
    # propagate by hand: layer 0 on the tiled batch, head on its samples; every piece from the layers' own (checked above) methods
    S, N = 3, 4
    sX = np.tile(X[None], (S, 1, 1))
    f0, _, _ = sample_from_conditional(model.layers[0], sX, z=np.reshape(zs[0], (S, N, -1)))
    m1, v1 = model.layers[1].conditional_ND(f0.reshape(S * N, -1))
    ve = MultiClass(10).variational_expectations(m1, v1, np.tile(Y, S)).reshape(S, N)
    want_elbo = ve.mean(0).sum() * 123.0 / N - sum(l.KL() for l in model.layers)
    assert abs(elbo - want_elbo) <= 1e-12 * abs(want_elbo), (elbo, want_elbo)


def test_natgrad_reference_against_torch_autograd():
    """The natural-gradient step the device NatGrad is checked against (tests/natgrad_ref.py, Salimbeni et al. 2018: theta <- theta + gamma dL/d eta
    in natural / expectation parameters) with dL/d eta taken by torch autograd through eta -> (mu, S) -> chol(S), for an arbitrary smooth objective
    of (mu, tril(L)) -- the reference's step uses the gradients with respect to (mu, L) and the Cholesky adjoint written out by hand."""
    import sys
    import os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from natgrad_ref import natgrad_reference
    rng = np.random.default_rng(31)
    M, R, gamma = 5, 2, 0.07
    mu = rng.standard_normal((M, R))
    Lq = np.stack([np.linalg.cholesky(_spd(rng, M)) for _ in range(R)])
    a, Bm, Cm = rng.standard_normal((M, R)), _spd(rng, M), rng.standard_normal((R, M, M))

    def objective(mu_t, L_t):   # any smooth function of (mu, lower triangle of L)
        Ll = L_t.tril()
        return (T(a) * mu_t).sum() - 0.5 * torch.einsum("mr,mk,kr->", mu_t, T(Bm), mu_t) + (T(Cm) * Ll).sum() \
            + torch.log(torch.diagonal(Ll, dim1=1, dim2=2) ** 2).sum() - 0.3 * (Ll @ Ll.transpose(1, 2)).diagonal(dim1=1, dim2=2).sum()

    mu_t, L_t = T(mu).requires_grad_(True), T(Lq).requires_grad_(True)
    g_mu, g_L = torch.autograd.grad(objective(mu_t, L_t), [mu_t, L_t])
    new_mu, new_L = natgrad_reference(mu, Lq, g_mu.numpy(), np.tril(g_L.numpy()), gamma)
    for r in range(R):
        S = Lq[r] @ Lq[r].T
        eta1 = T(mu[:, r]).requires_grad_(True)
        eta2 = T(S + np.outer(mu[:, r], mu[:, r])).requires_grad_(True)
        m_of = eta1
        S_of = 0.5 * (eta2 + eta2.T) - torch.outer(eta1, eta1)
        L_of = torch.linalg.cholesky(S_of)
        mu_all = torch.stack([m_of if q == r else T(mu[:, q]) for q in range(R)], 1)
        L_all = torch.stack([L_of if q == r else T(Lq[q]) for q in range(R)], 0)
        d1, d2 = torch.autograd.grad(objective(mu_all, L_all), [eta1, eta2])
        Sinv = np.linalg.inv(S)
        theta1 = Sinv @ mu[:, r] + gamma * d1.numpy()
        theta2 = -0.5 * Sinv + gamma * d2.numpy()
        S_new = np.linalg.inv(-2.0 * theta2)
        np.testing.assert_allclose(new_mu[:, r], S_new @ theta1, rtol=1e-9, atol=1e-11)
        np.testing.assert_allclose(new_L[r] @ new_L[r].T, S_new, rtol=1e-9, atol=1e-11)
