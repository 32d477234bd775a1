"""Test / bench infrastructure (never on the product path): the "best CPU" form of the forward ELBO -- the same arithmetic as the
reference's conv_gp/{views,layers,conditionals,kernels}.py (float64, jitter 1e-3), but batched the way a CPU likes it instead of in the
reference's operation order (which oracle/{layers,conditionals}.py keep: a Python loop of 2 P triangular solves and the dense tensordot of
conditionals.py:58):
  * every patch of the batch in ONE [M x K] K_uf (K = S N P columns) from one GEMM + exp (layers.py:23-32 without the map_fn),
  * ONE triangular solve over all K columns for  A = inv(L) K_uf  (conditionals.py:31-33), one more for inv(L)^T A (:44-47),
  * ONE (R M) x M x K GEMM for  L_q^T A  (:58) whose [R, M, K] result is reduced at once (:65),
  * ConvKernel.Kzx / Kdiag (kernels.py:106-133) by GEMMs over image chunks.
bench.py times it beside the reference-order oracle as `cpu_baseline.best_cpu`; tests/test_oracle_cpu.py holds it to the oracle at 1e-10."""
import numpy as np
import scipy.linalg as sla
from numpy.lib.stride_tricks import sliding_window_view

from .gpflow_ref import gauss_kl, MultiClass, JITTER


def _patches(X4, f, s):
    """[n, H, W, C] -> [n, P, L], p = oh W' + ow, l = (kh f + kw) C + c  (conv_gp/views.py:32-54)."""
    win = sliding_window_view(X4, (f, f), axis=(1, 2))[:, ::s, ::s]      # n, Ho, Wo, C, f, f
    win = np.transpose(win, (0, 1, 2, 4, 5, 3))
    return np.ascontiguousarray(win).reshape(X4.shape[0], win.shape[1] * win.shape[2], f * f * X4.shape[3])


def _rbf(A, B, variance, ls):
    """variance exp(-|a - b|^2 / (2 ls^2)) through gpflow's square_dist expansion: one GEMM."""
    A, B = A / ls, B / ls
    d2 = (A * A).sum(1)[:, None] + (B * B).sum(1)[None, :] - 2.0 * (A @ B.T)
    np.maximum(d2, 0.0, out=d2)
    d2 *= -0.5
    np.exp(d2, out=d2)
    d2 *= variance
    return d2


def _moments(Lc, Kuf, knn, q_mu, q_sqrt, white):
    """conditionals.py:29-65 for all K columns at once: mean [K, R], var [K, R]."""
    A = sla.solve_triangular(Lc, Kuf, lower=True, check_finite=False)
    var = knn - np.einsum("mk,mk->k", A, A)
    if not white:
        A = sla.solve_triangular(Lc.T, A, lower=False, check_finite=False)
    mean = A.T @ q_mu
    R, M = q_sqrt.shape[0], q_sqrt.shape[1]
    LqT = np.transpose(np.tril(q_sqrt), (0, 2, 1)).reshape(R * M, M)       # rows (r, j): L_q[r][:, j]
    T = (LqT @ A).reshape(R, M, -1)
    return mean, var[:, None] + np.einsum("rmk,rmk->kr", T, T)


def elbo(spec, X, Y, zs=None, rng=None):
    """Forward ELBO of a model spec (deepcgp_amd.synthetic.make_spec) on the batch X [N, H*W*C], Y [N]; zs: per-layer noise [S, N, D_l]
    or None (drawn from rng).  Returns (elbo, data term, KL).  RBF base kernels, zero mean functions, ConvKernel head."""
    S, N = int(spec["S"]), X.shape[0]
    rng = rng or np.random.default_rng()
    F = np.tile(np.asarray(X, np.float64).reshape(N, -1)[None], [S, 1, 1]).reshape(S * N, -1)
    kl = 0.0
    for li, c in enumerate(spec["convs"]):
        pat = _patches(F.reshape(S * N, c["H"], c["W"], c["C"]), c["f"], c["s"])
        n, P, L = pat.shape
        M, R = c["q_mu"].shape
        Kuu = _rbf(c["Z"], c["Z"], c["variance"], c["ls"]) + JITTER * np.eye(M)
        Lc = np.linalg.cholesky(Kuu)
        Kuf = _rbf(c["Z"], pat.reshape(n * P, L), c["variance"], c["ls"])
        mean, var = _moments(Lc, Kuf, c["variance"], c["q_mu"], c["q_sqrt"], c["white"])
        mean, var = mean.reshape(n, P * R), var.reshape(n, P * R)
        z = zs[li].reshape(mean.shape) if zs is not None else rng.standard_normal(mean.shape)
        F = mean + z * np.sqrt(var + JITTER)
        Kp = None if c["white"] else _rbf(c["Z0"], c["Z0"], c["variance"], c["ls"]) + JITTER * np.eye(M)
        kl += gauss_kl(c["q_mu"], c["q_sqrt"], Kp)
    h = spec["head"]
    pat = _patches(F.reshape(S * N, h["H"], h["W"], h["C"]), h["f"], h["s"])
    n, P, L = pat.shape
    M, R = h["q_mu"].shape
    w = np.asarray(h["w"], np.float64)
    Kuu = _rbf(h["Z"], h["Z"], h["variance"], h["ls"]) + JITTER * np.eye(M)
    Lc = np.linalg.cholesky(Kuu)
    Kzx, kdiag = np.empty((M, n)), np.empty(n)
    step = max(1, int(4e7 // max(P * P, P * M)))                          # image chunks of ~320 MB of kernel values
    for i0 in range(0, n, step):
        pc = pat[i0:i0 + step]
        nc = pc.shape[0]
        Kzx[:, i0:i0 + nc] = (_rbf(h["Z"], pc.reshape(nc * P, L), h["variance"], h["ls"]).reshape(M, nc, P) @ w) / P
        pl = pc / h["ls"]
        sq = np.einsum("npl,npl->np", pl, pl)
        G = sq[:, :, None] + sq[:, None, :] - 2.0 * (pl @ np.transpose(pl, (0, 2, 1)))
        np.maximum(G, 0.0, out=G)
        G *= -0.5
        np.exp(G, out=G)
        kdiag[i0:i0 + nc] = h["variance"] * np.einsum("p,npq,q->n", w, G, w) / P ** 2
    mean, var = _moments(Lc, Kzx, kdiag, h["q_mu"], h["q_sqrt"], h["white"])
    kl += gauss_kl(h["q_mu"], h["q_sqrt"], None if h["white"] else Kuu)
    ve = MultiClass(10).variational_expectations(mean, var, np.tile(np.asarray(Y).reshape(-1), S))
    data = float(ve.reshape(S, N).mean(0).sum())
    scale = float(spec["num_data"]) / N
    return data * scale - kl, data, kl
