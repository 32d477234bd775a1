"""The oracle's hand-written reverse pass (oracle/grad.py -- what the device gradients of csrc/grad.hip are checked against) against
PyTorch autograd (CPU, float64) of an independently written forward pass.

The reference gets its gradient from TensorFlow autodiff (conv_gp/experiment.py:84-108); TensorFlow is not in the image, torch is.
The forward below is the textbook form of the model -- patches by ``torch.nn.functional.unfold``, K^-1 through ``cholesky_solve``,
the conditional as  mean = K_fu K^-1 m,  var = k_ff - diag(K_fu K^-1 K_uf) + diag(K_fu K^-1 S K^-1 K_uf)  (not the oracle's
inv(L)-whitened re-association), the KL as the Gaussian closed form -- so agreement pins both the oracle's forward value and every entry of
its gradient by third-party differentiation machinery, to 1e-9 where the finite-difference pin (tests/test_oracle_cpu.py) reaches 1e-4.
Test infrastructure only: nothing under deepcgp_amd/ imports torch or this file."""
import math

import numpy as np
import pytest

torch = pytest.importorskip("torch")

from deepcgp_amd import synthetic as syn          # noqa: E402
from oracle_build import oracle_model             # noqa: E402

JITTER = 1e-3
T = torch.float64


def _patches(x_nhwc, f, s):
    """[N, H, W, C] -> [N, P, L] with l = (kh * f + kw) * C + c and p = oh * Wo + ow (tf.extract_image_patches order, views.py:32-54)."""
    N, H, W, C = x_nhwc.shape
    u = torch.nn.functional.unfold(x_nhwc.permute(0, 3, 1, 2), kernel_size=f, stride=s)      # [N, C * f * f, P], channel-major rows
    P = u.shape[-1]
    u = u.reshape(N, C, f * f, P).permute(0, 3, 2, 1)                                          # [N, P, f * f, C]
    return u.reshape(N, P, f * f * C)


def _rbf(A, B, variance, ls):
    return variance * torch.exp(-0.5 * torch.cdist(A / ls, B / ls, compute_mode="donot_use_mm_for_euclid_dist") ** 2)


def _acos0(A, B, variance, wv, bv):
    """gpflow ArcCosine(order 0): variance / pi * (pi - theta), theta = acos(1e-15 + (1 - 2e-15) cos), <x, z> = wv x.z + bv."""
    num = wv * (A @ B.T) + bv
    da = torch.sqrt(wv * (A * A).sum(1) + bv)
    db = torch.sqrt(wv * (B * B).sum(1) + bv)
    theta = torch.acos(1e-15 + (1.0 - 2e-15) * num / da[:, None] / db[None, :])
    return variance * (math.pi - theta) / math.pi


def _conditional(Kuu, Kuf, kff, q_mu, q_sqrt, white):
    """columns c of Kuf [M, C]; returns mean [C, R], var [C, R] of q(f_c) for the R outputs (full_cov = False)."""
    M = Kuu.shape[0]
    L = torch.linalg.cholesky(Kuu)
    S = torch.tril(q_sqrt) @ torch.tril(q_sqrt).transpose(1, 2)                                # [R, M, M]
    if white:   # u = L v, v ~ N(q_mu, S): mean = K_fu L^-T q_mu, cov = k_ff - a^T a + a^T S a with a = L^-1 K_uf
        a = torch.linalg.solve_triangular(L, Kuf, upper=False)
        mean = a.T @ q_mu
        var = kff[:, None] - (a * a).sum(0)[:, None] + torch.einsum("mc,rmk,kc->cr", a, S, a)
    else:
        KiKuf = torch.cholesky_solve(Kuf, L)                                                   # K^-1 K_uf
        mean = KiKuf.T @ q_mu
        var = kff[:, None] - (Kuf * KiKuf).sum(0)[:, None] + torch.einsum("mc,rmk,kc->cr", KiKuf, S, KiKuf)
    return mean, var


def _gauss_kl(q_mu, q_sqrt, K):
    """sum over the R outputs of KL[N(q_mu_r, Lq_r Lq_r^T) || N(0, K)]; K None: the whitened prior N(0, I)."""
    M, R = q_mu.shape
    Lq = torch.tril(q_sqrt)
    S = Lq @ Lq.transpose(1, 2)
    logdet_q = torch.log(torch.diagonal(Lq, dim1=1, dim2=2) ** 2).sum()
    if K is None:
        return 0.5 * ((q_mu ** 2).sum() + torch.diagonal(S, dim1=1, dim2=2).sum() - M * R - logdet_q)
    Ki = torch.linalg.inv(K)
    return 0.5 * (torch.einsum("mr,mk,kr->", q_mu, Ki, q_mu) + torch.einsum("mk,rkm->", Ki, S) - M * R - logdet_q + R * torch.logdet(K))


def _robustmax_ve(mu, var, y, eps=1e-3, n_gh=20):
    """E_q[log p(y | f)] of gpflow's MultiClass + RobustMax: 20-point Gauss-Hermite over the true class's latent."""
    gx, gw = np.polynomial.hermite.hermgauss(n_gh)
    gx, gw = torch.tensor(gx, dtype=T), torch.tensor(gw / math.sqrt(math.pi), dtype=T)
    n, K = mu.shape
    on = torch.nn.functional.one_hot(y, K).to(T)
    mu_y, var_y = (on * mu).sum(1), (on * var).sum(1)
    X = mu_y[:, None] + gx[None, :] * torch.sqrt(torch.clamp(2.0 * var_y, min=1e-10))[:, None]
    dist = (X[:, None, :] - mu[:, :, None]) / torch.sqrt(torch.clamp(var, min=1e-10))[:, :, None]
    cdf = 0.5 * (1.0 + torch.erf(dist / math.sqrt(2.0))) * (1 - 2e-4) + 1e-4
    cdf = cdf * (1.0 - on)[:, :, None] + on[:, :, None]
    p = cdf.prod(1) @ gw
    return p * math.log(1.0 - eps) + (1.0 - p) * math.log(eps / (K - 1.0))


def _robustmax_predict(mu, var, eps=1e-3):
    """class probabilities of gpflow's MultiClass.predict_mean_and_var: p_k = P(f_k largest) (1 - eps) + (1 - P) eps / (K - 1)."""
    n, K = mu.shape
    cols = []
    for k in range(K):
        y = torch.full((n,), k, dtype=torch.long)
        ve = _robustmax_ve(mu, var, y, eps)                      # p log(1 - eps) + (1 - p) log(eps / (K - 1)) -> p
        a, b = math.log(1.0 - eps), math.log(eps / (K - 1.0))
        p = (ve - b) / (a - b)
        cols.append(p * (1.0 - eps) + (1.0 - p) * eps / (K - 1.0))
    return torch.stack(cols, 1)


def _torch_elbo(spec, X, Y, zs, want_head=False):
    """ELBO of the conv layers + head (ConvKernel, AdditivePatchKernel or dense RBF(ARD)) of `spec` and the leaf tensors it depends on,
    [{name: tensor}] per layer."""
    leaves = []
    S, N = spec["S"], X.shape[0]
    F = torch.tensor(np.tile(X[None], [S, 1, 1]).reshape(S * N, -1), dtype=T)
    kl = torch.zeros((), dtype=T)

    def leaf(a):
        return torch.tensor(np.array(a, np.float64), dtype=T, requires_grad=True)
    for li, c in enumerate(spec["convs"]):
        p = dict(Z=leaf(c["Z"]), q_mu=leaf(c["q_mu"]), q_sqrt=leaf(c["q_sqrt"]), variance=leaf(c["variance"]), lengthscales=leaf(c["ls"]))
        leaves.append(p)
        M, R = c["M"], c["R"]
        pt = _patches(F.reshape(S * N, c["H"], c["W"], c["C"]), c["f"], c["s"])               # [SN, P, L]
        P = pt.shape[1]
        cols = pt.reshape(S * N * P, -1)                                                     # column (n, p)
        if c.get("base", "rbf") == "acos":   # ArcCosine(order 0) base kernel with gpflow's default hyper-parameters (value only: see the test below)
            one = torch.ones((), dtype=T)
            kern = lambda A, B: _acos0(A, B, one, one, one)                                  # noqa: E731
            p["variance"] = one
        else:
            kern = lambda A, B: _rbf(A, B, p["variance"], p["lengthscales"])                   # noqa: E731
        Kuu = kern(p["Z"], p["Z"]) + JITTER * torch.eye(M, dtype=T)
        Kuf = kern(p["Z"], cols)
        kff = p["variance"] * torch.ones(cols.shape[0], dtype=T)
        mean, var = _conditional(Kuu, Kuf, kff, p["q_mu"], p["q_sqrt"], c["white"])          # [SNP, R]
        mean, var = mean.reshape(S * N, P * R), var.reshape(S * N, P * R)                    # output index p * R + r (layers.py:128-131)
        if c.get("mean_function") == "conv2d":   # Conv2dMean (mean_functions.py:28-41): output map 0 of a patch = the centre pixel of input channel 0
            centre = pt.reshape(S * N, P, c["f"], c["f"], c["C"])[:, :, c["f"] // 2, c["f"] // 2, 0]
            mean = mean + torch.cat([centre[:, :, None], torch.zeros(S * N, P, R - 1, dtype=T)], 2).reshape(S * N, P * R)
        z = torch.tensor(np.asarray(zs[li]).reshape(S * N, P * R), dtype=T)
        F = mean + z * torch.sqrt(var + JITTER)
        Z0 = torch.tensor(np.array(c["Z0"], np.float64), dtype=T)                             # the prior's inducing patches are frozen (layers.py:149-152)
        Kp = None if c["white"] else kern(Z0, Z0) + JITTER * torch.eye(M, dtype=T)
        kl = kl + _gauss_kl(p["q_mu"], p["q_sqrt"], Kp)
    h = spec["head"]
    M = h["M"]
    if h.get("kernel", "conv") == "rbf":   # dense head: gpflow RBF(ARD=True) on the flattened features (models.py:160-168)
        p = dict(Z=leaf(h["Z"]), q_mu=leaf(h["q_mu"]), q_sqrt=leaf(h["q_sqrt"]), variance=leaf(h["variance"]), lengthscales=leaf(h["ls_ard"]))
        leaves.append(p)
        Kuu = _rbf(p["Z"], p["Z"], p["variance"], p["lengthscales"][None, :]) + JITTER * torch.eye(M, dtype=T)
        Kzx = _rbf(p["Z"], F, p["variance"], p["lengthscales"][None, :])
        kdiag = p["variance"] * torch.ones(F.shape[0], dtype=T)
    else:
        p = dict(Z=leaf(h["Z"]), q_mu=leaf(h["q_mu"]), q_sqrt=leaf(h["q_sqrt"]), variance=leaf(h["variance"]), lengthscales=leaf(h["ls"]),
                 patch_weights=leaf(h["w"]))
        leaves.append(p)
        pt = _patches(F.reshape(S * N, h["H"], h["W"], h["C"]), h["f"], h["s"])                  # [SN, P, L]
        P = pt.shape[1]
        w = p["patch_weights"]
        Kall = _rbf(p["Z"], pt.reshape(S * N * P, -1), p["variance"], p["lengthscales"]).reshape(M, S * N, P)
        Kzx = (Kall * w[None, None, :]).sum(2) / P                                               # kernels.py:63-74 / :117-133
        if h.get("kernel", "conv") == "add":   # AdditivePatchKernel.Kdiag (kernels.py:53-61): mean_p w_p k(x_p, x_p)
            kdiag = p["variance"] * w.mean() * torch.ones(S * N, dtype=T)
        else:                                  # ConvKernel.Kdiag (kernels.py:106-115): all patch pairs of an image
            Kpp = p["variance"] * torch.exp(-0.5 * torch.cdist(pt / p["lengthscales"], pt / p["lengthscales"], compute_mode="donot_use_mm_for_euclid_dist") ** 2)
            kdiag = torch.einsum("npq,p,q->n", Kpp, w, w) / P ** 2
        Kuu = _rbf(p["Z"], p["Z"], p["variance"], p["lengthscales"]) + JITTER * torch.eye(M, dtype=T)
    mean, var = _conditional(Kuu, Kzx, kdiag, p["q_mu"], p["q_sqrt"], h["white"])
    kl = kl + _gauss_kl(p["q_mu"], p["q_sqrt"], None if h["white"] else Kuu)                  # the head's prior shares the live Z
    y = torch.tensor(np.tile(np.asarray(Y).reshape(1, N), [S, 1]).reshape(S * N), dtype=torch.long)
    ve = _robustmax_ve(mean, var, y).reshape(S, N).mean(0).sum()
    if want_head:   # the last layer's marginals [S, N, R] as well (DGP_Base.propagate / predict_y)
        return ve * (spec["num_data"] / N) - kl, leaves, mean.reshape(S, N, -1), var.reshape(S, N, -1)
    return ve * (spec["num_data"] / N) - kl, leaves


@pytest.mark.parametrize("white,variant", [(False, "conv"), (True, "conv"), (False, "three_layers_stride2"), (False, "additive"), (False, "dense_ard"),
                                           (True, "dense_ard"), (False, "conv2d_mean")])
def test_hand_written_gradient_matches_torch_autograd(white, variant):
    from oracle.grad import elbo_and_grad
    hwc, N, S = ((14, 14, 1) if variant == "three_layers_stride2" else (10, 10, 1)), 3, 2
    convs = [(4, 2, 2), (3, 1, 2)] if variant == "three_layers_stride2" else [(3, 1, 2)]
    spec = syn.make_spec(hwc, convs, (3, 1), 7, S=S, num_data=200, seed=11, white=white, conv_q_sqrt_scale=0.3, variance=2.0, ls=1.5,
                         head_kernel="rbf" if variant == "dense_ard" else "conv")
    rng = np.random.default_rng(11)
    if variant != "dense_ard":
        spec["head"]["w"] = 0.5 + rng.random(spec["head"]["w"].shape)
    if variant == "additive":
        spec["head"]["kernel"] = "add"
    if variant == "conv2d_mean":
        spec["convs"][0]["mean_function"] = "conv2d"
    spec["convs"][0]["Z0"] = spec["convs"][0]["Z"] + 0.05 * rng.standard_normal(spec["convs"][0]["Z"].shape)   # prior patches != live patches
    X, Y = syn.make_batch(hwc, N, seed=11)
    zs = syn.make_noise(spec, N, seed=11)
    ref = oracle_model(spec, X, Y)
    if variant == "additive":
        from oracle.kernels import AdditivePatchKernel
        k = ref.layers[-1].kern
        ref.layers[-1].kern = AdditivePatchKernel(k.base_kernel, k.view, k.patch_weights)
    e_oracle, g_oracle = elbo_and_grad(ref, X, Y, zs)
    e_torch, leaves = _torch_elbo(spec, X, Y, zs)
    assert abs(e_torch.item() - e_oracle) <= 1e-10 * abs(e_oracle)
    flat = [(li, k, t) for li, p in enumerate(leaves) for k, t in p.items()]
    grads = torch.autograd.grad(e_torch, [t for _, _, t in flat])
    for (li, name, _), g in zip(flat, grads):
        want = np.asarray(g_oracle[li][name], np.float64)
        got = g.numpy()
        if name == "q_sqrt":
            got = np.tril(got)                      # only the lower triangle is a parameter
            want = np.tril(want)
        err = np.abs(got - want).max()
        assert err <= 1e-9 * max(1.0, np.abs(want).max()), (variant, li, name, err, np.abs(want).max())


@pytest.mark.parametrize("white", [False, True])
def test_arccosine_forward_matches_torch(white):
    """ArcCosine(order 0) conv layers (--base-kernel acos): the oracle's ELBO against the torch forward.  Value only: the oracle's gradient skips the
    coincident points of K_uu on purpose (acos' slope at 1 - 1e-15 is ~2e7 and would only amplify rounding; oracle/grad.py), autograd does not."""
    hwc, N, S = (10, 10, 1), 3, 2
    spec = syn.make_spec(hwc, [(3, 1, 2)], (3, 1), 7, S=S, num_data=200, seed=13, white=white, conv_q_sqrt_scale=0.3, variance=2.0, ls=1.5, base_kernel="acos")
    X, Y = syn.make_batch(hwc, N, seed=13)
    zs = syn.make_noise(spec, N, seed=13)
    ref = oracle_model(spec, X, Y)
    with torch.no_grad():
        e_t, _ = _torch_elbo(spec, X, Y, zs)
    e_o = ref.compute_log_likelihood(X, Y, zs=zs)
    assert abs(e_t.item() - e_o) <= 1e-9 * abs(e_o), (e_t.item(), e_o)
