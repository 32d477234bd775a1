"""Oracle (test infrastructure): reverse-mode gradient of the oracle ELBO, written out by hand in numpy.

The reference obtains gradients from TensorFlow autodiff of the graph that `oracle/dgp.py`
restates (`/root/reference/conv_gp/experiment.py:84-108` builds the Adam action on
`model.likelihood_tensor`); there is no gradient code in the reference to follow line by line.  This
file differentiates the oracle's forward pass (`oracle/layers.py`, `oracle/conditionals.py`,
`oracle/kernels.py`, `oracle/dgp.py`, `oracle/gpflow_ref.py`) and is itself pinned by central finite
differences of `DGP_Base.compute_log_likelihood` (tests/test_oracle_cpu.py).  PARITY UNPINNED in the same
sense as the rest of the oracle.

Scope: RBF base kernels with one lengthscale -- or ArcCosine(order 0) on the conv layers, pinned at kernel level only
(finite differences of the ELBO cannot see past the rounding noise of acos(1 - 1e-15) on the K_uu diagonal) --, `ConvLayer`s (mean function None or the fixed `Conv2dMean`) followed by an
`SVGP_Layer` whose kernel is `ConvKernel`, `AdditivePatchKernel` or the dense `RBF(ARD=True)` of `--last-kernel rbf`;
whitened or not.  Gradients are taken
with respect to the constrained values (variance, lengthscales, Z, q_mu, q_sqrt (lower triangle),
patch_weights); the frozen prior inducing patches `Z0` of a ConvLayer receive none.
"""
import numpy as np
from scipy.linalg import solve_triangular
from scipy.special import erf

from .gpflow_ref import JITTER
from . import kernels as _k


# ---------------------------------------------------------------------------------------------------
# small pieces
# ---------------------------------------------------------------------------------------------------
def _rbf(kern, A, B):
    d2 = np.sum(A * A, 1)[:, None] + np.sum(B * B, 1)[None, :] - 2.0 * A @ B.T
    return kern.variance * np.exp(-0.5 * d2 / kern.lengthscales ** 2), d2


def _rbf_cross_backward(kern, Zm, Xc, Kzx, d2, E_scale):
    """K = rbf(Zm, Xc); E_scale = dLoss/dK.  Returns (dZ, dX, dvariance, dlengthscale)."""
    E = E_scale * Kzx
    l2 = kern.lengthscales ** 2
    dZ = (E @ Xc - E.sum(1)[:, None] * Zm) / l2
    dX = (E.T @ Zm - E.sum(0)[:, None] * Xc) / l2
    return dZ, dX, E.sum() / kern.variance, np.sum(E * d2) / kern.lengthscales ** 3


def _chol_backward(L, dL):
    """dLoss/dK (symmetric) from dLoss/dL for K = L L^T (Murray 2016, eq. 9 in blocked form)."""
    P = np.tril(L.T @ dL)
    P[np.diag_indices_from(P)] *= 0.5
    S = solve_triangular(L, solve_triangular(L, P.T, lower=True, trans='T').T, lower=True, trans='T')
    return 0.5 * (S + S.T)


def _lsolve_T(L, B):
    return solve_triangular(L, B, lower=True, trans='T')


def _cond_backward(L, A1, alpha, G, gm, gv, white, q_mu, Lq):
    """Shared by ConvLayer and SVGP_Layer.  Forward (either whitening):
         mean[c, r] = sum_m alpha[m, r] A1[m, c];  var[c, r] = Knn[c] - sum_m A1^2 + sum_m (G_r^T A1)^2
       with A1 = inv(L) Kuf, alpha / G_r = q_mu / tril(q_sqrt_r) (white) or inv(L) of those.
       gm, gv: [C, R].  Returns dKuf, dL (lower), dq_mu, dq_sqrt, dKnn[c]."""
    R = gm.shape[1]
    dA1 = alpha @ gm.T - 2.0 * A1 * gv.sum(1)[None, :]
    dalpha = A1 @ gm
    dG = np.zeros_like(G)
    for r in range(R):
        T = G[r].T @ A1
        dT = 2.0 * T * gv[:, r][None, :]
        dA1 += G[r] @ dT
        dG[r] = np.tril(A1 @ dT.T)
    dL = np.zeros_like(L)
    if white:
        dq_mu, dq_sqrt = dalpha, dG
    else:
        dq_mu = _lsolve_T(L, dalpha)
        dL -= np.tril(dq_mu @ alpha.T)
        dq_sqrt = np.zeros_like(G)
        for r in range(R):
            B = _lsolve_T(L, dG[r])
            dq_sqrt[r] = np.tril(B)
            dL -= np.tril(B @ G[r].T)
    dKuf = _lsolve_T(L, dA1)
    dL -= np.tril(dKuf @ A1.T)
    return dKuf, dL, dq_mu, dq_sqrt, gv.sum(1)


def _kl_backward(q_mu, Lq, Lp):
    """d KL / d(q_mu, q_sqrt, K_prior).  Lp None = whitened prior."""
    R = q_mu.shape[1]
    dLq = np.zeros_like(Lq)
    if Lp is None:
        for r in range(R):
            dLq[r] = np.tril(Lq[r]) - np.diag(1.0 / np.diag(Lq[r]))
        return q_mu.copy(), dLq, None
    Kinv_mu = _lsolve_T(Lp, solve_triangular(Lp, q_mu, lower=True))
    M = Lp.shape[0]
    Kinv = _lsolve_T(Lp, solve_triangular(Lp, np.eye(M), lower=True))
    dK = 0.5 * (R * Kinv - Kinv_mu @ Kinv_mu.T)
    for r in range(R):
        KiL = _lsolve_T(Lp, solve_triangular(Lp, Lq[r], lower=True))
        dLq[r] = np.tril(KiL) - np.diag(1.0 / np.diag(Lq[r]))
        dK -= 0.5 * KiL @ KiL.T
    return Kinv_mu, dLq, dK


def _kuu_backward(kern, Z, dK):
    """K = rbf(Z, Z) + jitter I with dLoss/dK (any, not necessarily symmetric)."""
    K, d2 = _rbf(kern, Z, Z)
    E = dK * K
    Es = E + E.T
    dZ = (Es @ Z - Es.sum(1)[:, None] * Z) / kern.lengthscales ** 2
    return dZ, E.sum() / kern.variance, np.sum(E * d2) / kern.lengthscales ** 3


def _acos_backward(kern, Zm, Xc, dK, skip_diag=False):
    """ArcCosine(order 0) adjoint (oracle/gpflow_ref.py ArcCosine.K): K = variance (pi - theta) / pi, theta = acos(c'),
    c' = 1e-15 + (1 - 2e-15) c, c = (w x.z + b) / sqrt((w |x|^2 + b)(w |z|^2 + b)).  dK = dLoss/dK [M x N].
    Returns dZ, dX, dvariance, dweight_variances, dbias_variance.  ``skip_diag`` (K_uu: both arguments are Z): on the
    diagonal c == 1 identically in z, w and b, so those entries carry no gradient except through ``variance`` -- while
    dK/dc = variance / (pi sin(theta)) is ~2e7 there and would multiply rounding noise."""
    w, b, var = kern.weight_variances, kern.bias_variance, kern.variance
    A = w * np.sum(Xc * Xc, 1) + b                      # [N]
    Q = w * np.sum(Zm * Zm, 1) + b                      # [M]
    s = Zm @ Xc.T
    rt = np.sqrt(Q[:, None] * A[None, :])
    c = (w * s + b) / rt
    cp = np.minimum(1e-15 + (1.0 - 2e-15) * c, 1.0)
    theta = np.arccos(cp)
    K = var * (np.pi - theta) / np.pi
    with np.errstate(divide="ignore", invalid="ignore"):
        F = dK * var / np.pi * (1.0 - 2e-15) / np.sin(theta)           # dLoss/dc
    if skip_diag:
        F[np.diag_indices(min(F.shape))] = 0.0
    F1 = F / rt
    F2 = F * c
    dZ = w * (F1 @ Xc - (F2.sum(1) / Q)[:, None] * Zm)
    dX = w * (F1.T @ Zm - (F2.sum(0) / A)[:, None] * Xc)
    a2, q2 = (A - b) / w, (Q - b) / w                  # |x|^2, |z|^2
    dw = np.sum(F1 * s) - 0.5 * np.sum(F2 * (a2[None, :] / A[None, :] + q2[:, None] / Q[:, None]))
    db = np.sum(F1) - 0.5 * np.sum(F2 * (1.0 / A[None, :] + 1.0 / Q[:, None]))
    return dZ, dX, np.sum(dK * K) / var, dw, db


def _acos_kuu_backward(kern, Z, dK):
    """K = acos(Z, Z) (+ jitter I) with dLoss/dK any matrix: both arguments are Z."""
    dZa, dZb, dvar, dw, db = _acos_backward(kern, Z, Z, dK, skip_diag=True)
    return dZa + dZb, dvar, dw, db


def robustmax_backward(lik, Fmu, Fvar, Y):
    """d sum_n ve_n / d(Fmu, Fvar) for MultiClass.variational_expectations (oracle/gpflow_ref.py)."""
    gh_x, gh_w = np.polynomial.hermite.hermgauss(lik.num_gauss_hermite_points)
    Y = np.asarray(Y).reshape(-1).astype(np.int64)
    n, K = Fmu.shape
    on = np.zeros((n, K))
    on[np.arange(n), Y] = 1.0
    mu_y, var_y = Fmu[np.arange(n), Y], Fvar[np.arange(n), Y]
    clip_y = 2.0 * var_y > 1e-10
    s_y = np.sqrt(np.clip(2.0 * var_y, 1e-10, np.inf))
    Xg = mu_y[:, None] + gh_x[None, :] * s_y[:, None]                      # n x G
    clip_k = Fvar > 1e-10
    sig = np.sqrt(np.clip(Fvar, 1e-10, np.inf))                            # n x K
    dist = (Xg[:, None, :] - Fmu[:, :, None]) / sig[:, :, None]            # n x K x G
    cdf = (0.5 * (1.0 + erf(dist / np.sqrt(2.0)))) * (1 - 2e-4) + 1e-4
    cdf = cdf * (1.0 - on)[:, :, None] + on[:, :, None]
    prod = np.prod(cdf, axis=1)                                            # n x G
    wg = gh_w / np.sqrt(np.pi)
    pdf = np.exp(-0.5 * dist ** 2) / np.sqrt(2.0 * np.pi) * (1 - 2e-4)
    q = wg[None, None, :] * prod[:, None, :] / cdf * pdf * (1.0 - on)[:, :, None]   # dp / d dist
    c = np.log(1.0 - lik.epsilon) - np.log(lik.eps_k1)                     # d ve / d p
    dmu = -q.sum(2) / sig
    dvar = np.where(clip_k, -(q * dist).sum(2) / (2.0 * np.clip(Fvar, 1e-10, np.inf)), 0.0)
    qs = (q / sig[:, :, None]).sum(1)                                      # n x G : dp / d Xg
    dmu[np.arange(n), Y] = qs.sum(1)
    dvar[np.arange(n), Y] = np.where(clip_y, (qs * gh_x[None, :]).sum(1) / s_y, 0.0)
    return c * dmu, c * dvar


# ---------------------------------------------------------------------------------------------------
# layers
# ---------------------------------------------------------------------------------------------------
def _patch_scatter(view, dPatches_NPL, N):
    """adjoint of FullView.extract_patches: N x P x L -> N x H x W x C."""
    H, W = view.input_size[0], view.input_size[1]
    C, f, s = view.feature_maps, view.filter_size, view.stride
    Ho, Wo = view.out_image_height, view.out_image_width
    d = dPatches_NPL.reshape(N, Ho, Wo, f, f, C)
    out = np.zeros((N, H, W, C))
    for oh in range(Ho):
        for ow in range(Wo):
            out[:, oh * s:oh * s + f, ow * s:ow * s + f, :] += d[:, oh, ow]
    return out


def conv_layer_backward(layer, X, gmean, gvar):
    """ConvLayer.conditional_ND backward.  X [Nt, D]; gmean / gvar [Nt, P*R].  Returns dX and a dict."""
    v, kern = layer.view, layer.base_kernel
    Nt, P, R, M = X.shape[0], layer.patch_count, layer.gp_count, layer.num_inducing
    NHWC = X.reshape(Nt, v.input_size[0], v.input_size[1], layer.feature_maps_in)
    Xc = v.extract_patches_PNL(NHWC).reshape(P * Nt, v.patch_length)        # column c = p * Nt + n
    acos = not hasattr(kern, "lengthscales")             # ArcCosine(order 0) base kernel (--base-kernel acos)
    Kuf, d2 = (kern.K(layer.Z, Xc), None) if acos else _rbf(kern, layer.Z, Xc)
    Kuu = layer.conv_kernel.Kuu(layer.Z)
    L = np.linalg.cholesky(Kuu)
    A1 = solve_triangular(L, Kuf, lower=True)
    Lq = np.tril(layer.q_sqrt)
    if layer.white:
        alpha, G = layer.q_mu, Lq
    else:
        alpha = solve_triangular(L, layer.q_mu, lower=True)
        G = np.stack([solve_triangular(L, Lq[r], lower=True) for r in range(R)])
    gm = np.transpose(gmean.reshape(Nt, P, R), (1, 0, 2)).reshape(P * Nt, R)
    gv = np.transpose(gvar.reshape(Nt, P, R), (1, 0, 2)).reshape(P * Nt, R)
    dKuf, dL, dq_mu, dq_sqrt, dKnn = _cond_backward(L, A1, alpha, G, gm, gv, layer.white, layer.q_mu, Lq)
    if acos:
        dZ, dXc, dvar, dw, db = _acos_backward(kern, layer.Z, Xc, dKuf)
        dZ2, dvar2, dw2, db2 = _acos_kuu_backward(kern, layer.Z, _chol_backward(L, dL))
        hyper = {"weight_variances": dw + dw2, "bias_variance": db + db2}
    else:
        dZ, dXc, dvar, dls = _rbf_cross_backward(kern, layer.Z, Xc, Kuf, d2, dKuf)
        dZ2, dvar2, dls2 = _kuu_backward(kern, layer.Z, _chol_backward(L, dL))
        hyper = {"lengthscales": dls + dls2}
    dvar += dKnn.sum()                                                      # Knn = variance for every column
    dX = _patch_scatter(v, np.transpose(dXc.reshape(P, Nt, -1), (1, 0, 2)), Nt).reshape(Nt, -1)
    if layer.mean_function is not None:          # Conv2dMean (fixed filter): mean += centre pixel of channel 0 on map 0
        dX = dX + layer.mean_function.backward(NHWC, gmean).reshape(Nt, -1)
    g = {"Z": dZ + dZ2, "variance": dvar + dvar2, "q_mu": dq_mu, "q_sqrt": dq_sqrt}
    g.update(hyper)
    return dX, g


def conv_layer_kl_backward(layer):
    Lq = np.tril(layer.q_sqrt)
    if layer.white:
        dq_mu, dLq, _ = _kl_backward(layer.q_mu, Lq, None)
        return {"q_mu": dq_mu, "q_sqrt": dLq}
    Lp = np.linalg.cholesky(layer.conv_kernel.Kuu(layer.Z0))
    dq_mu, dLq, dK = _kl_backward(layer.q_mu, Lq, Lp)
    if not hasattr(layer.base_kernel, "lengthscales"):
        _, dvar, dw, db = _acos_kuu_backward(layer.base_kernel, layer.Z0, dK)     # Z0 is frozen
        return {"q_mu": dq_mu, "q_sqrt": dLq, "variance": dvar, "weight_variances": dw, "bias_variance": db}
    _, dvar, dls = _kuu_backward(layer.base_kernel, layer.Z0, dK)          # Z0 is frozen
    return {"q_mu": dq_mu, "q_sqrt": dLq, "variance": dvar, "lengthscales": dls}


def head_backward(layer, X, gmean, gvar):
    """SVGP_Layer.conditional_ND backward with a ConvKernel / AdditivePatchKernel.  X [Nt, D]; gmean/gvar [Nt, R]."""
    if layer.mean_function is not None:
        raise NotImplementedError("gradients: mean_function must be None")
    kern = layer.kern
    base, v, w = kern.base_kernel, kern.view, kern.patch_weights
    Nt, P, R, M = X.shape[0], kern.patch_count, layer.num_outputs, layer.num_inducing
    NHWC = kern._reshape_X(X)
    patches = v.extract_patches(NHWC)                                       # Nt x P x L
    Xc = patches.reshape(Nt * P, -1)                                        # column c = n * P + p
    Kfull, d2 = _rbf(base, layer.Z, Xc)
    Kzx = (Kfull.reshape(M, Nt, P) * w[None, None, :]).sum(2) / P
    Ku = base.K(layer.Z) + np.eye(M) * JITTER
    L = np.linalg.cholesky(Ku)
    A1 = solve_triangular(L, Kzx, lower=True)
    Lq = layer.q_sqrt                         # SVGP_Layer uses q_sqrt as stored (oracle/dgp.py SK = q_sqrt q_sqrt^T - ...)
    if layer.white:
        alpha, G = layer.q_mu, Lq
    else:
        alpha = solve_triangular(L, layer.q_mu, lower=True)
        G = np.stack([solve_triangular(L, Lq[r], lower=True) for r in range(R)])
    dKzx, dL, dq_mu, dq_sqrt, dkd = _cond_backward(L, A1, alpha, G, gmean, gvar, layer.white, layer.q_mu, Lq)
    # Kzx[m, n] = 1/P sum_p w_p k(Z_m, x_np)
    Q = np.repeat(dKzx, P, axis=1) * np.tile(w, Nt)[None, :] / P            # dLoss / dKfull
    dZ, dXc, dvar, dls = _rbf_cross_backward(base, layer.Z, Xc, Kfull, d2, Q)
    dw = (np.repeat(dKzx, P, axis=1) * Kfull).reshape(M, Nt, P).sum((0, 1)) / P
    dPatches = dXc.reshape(Nt, P, -1)
    # Kdiag
    if isinstance(kern, _k.ConvKernel):
        W2 = w[None, :] * w[:, None] / P ** 2
        for n in range(Nt):
            Kn, d2n = _rbf(base, patches[n], patches[n])
            E = dkd[n] * W2 * Kn
            dvar += E.sum() / base.variance
            dls += np.sum(E * d2n) / base.lengthscales ** 3
            dw += 2.0 * dkd[n] * (Kn @ w) / P ** 2
            Es = E + E.T
            dPatches[n] += (Es @ patches[n] - Es.sum(1)[:, None] * patches[n]) / base.lengthscales ** 2
    else:                                      # AdditivePatchKernel.Kdiag = mean_p w_p variance
        dvar += dkd.sum() * w.mean()
        dw += dkd.sum() * base.variance / P
    dZ2, dvar2, dls2 = _kuu_backward(base, layer.Z, _chol_backward(L, dL))
    dX = _patch_scatter(v, dPatches, Nt).reshape(Nt, -1)
    return dX, {"Z": dZ + dZ2, "variance": dvar + dvar2, "lengthscales": dls + dls2, "q_mu": dq_mu,
                "q_sqrt": dq_sqrt, "patch_weights": dw}


class _UnitRBF:
    """view of an ARD RBF kernel on inputs already divided by the lengthscales"""
    def __init__(self, variance):
        self.variance, self.lengthscales = variance, 1.0


def dense_head_backward(layer, X, gmean, gvar):
    """SVGP_Layer.conditional_ND backward with gpflow RBF(D, ARD=True) on the flattened features (the dense head of
    --last-kernel rbf, conv_gp/models.py:160-168).  X [Nt, D]; gmean / gvar [Nt, R].  ``lengthscales`` gets a [D] gradient."""
    if layer.mean_function is not None:
        raise NotImplementedError("gradients: mean_function must be None")
    kern = layer.kern
    ls = np.asarray(kern.lengthscales, np.float64)
    unit = _UnitRBF(kern.variance)
    M, R = layer.num_inducing, layer.num_outputs
    Zs, Xs = layer.Z / ls, np.asarray(X, np.float64) / ls
    Kzx, d2 = _rbf(unit, Zs, Xs)
    Ku = _rbf(unit, Zs, Zs)[0] + np.eye(M) * JITTER
    L = np.linalg.cholesky(Ku)
    A1 = solve_triangular(L, Kzx, lower=True)
    Lq = layer.q_sqrt
    if layer.white:
        alpha, G = layer.q_mu, Lq
    else:
        alpha = solve_triangular(L, layer.q_mu, lower=True)
        G = np.stack([solve_triangular(L, Lq[r], lower=True) for r in range(R)])
    dKzx, dL, dq_mu, dq_sqrt, dkd = _cond_backward(L, A1, alpha, G, gmean, gvar, layer.white, layer.q_mu, Lq)
    dZs, dXs, dvar, _ = _rbf_cross_backward(unit, Zs, Xs, Kzx, d2, dKzx)
    dvar += dkd.sum()                                            # Kdiag = variance
    S = _chol_backward(L, dL)
    kl = {}
    if layer.white:
        d = np.tril(Lq)
        for r in range(R):
            d[r] -= np.diag(1.0 / np.diag(Lq[r]))
        kl = {"q_mu": layer.q_mu.copy(), "q_sqrt": d}
    else:
        Kinv_mu = _lsolve_T(L, solve_triangular(L, layer.q_mu, lower=True))
        Kinv = _lsolve_T(L, solve_triangular(L, np.eye(M), lower=True))
        dK = 0.5 * (R * Kinv - Kinv_mu @ Kinv_mu.T)
        dLq = np.zeros_like(Lq)
        for r in range(R):
            KiL = _lsolve_T(L, solve_triangular(L, Lq[r], lower=True))
            dLq[r] = np.tril(KiL) - np.diag(1.0 / np.diag(Lq[r]))
            dK -= 0.5 * KiL @ KiL.T
        kl = {"q_mu": Kinv_mu, "q_sqrt": dLq, "dK": dK}
    return dKzx, (dZs, dXs, dvar, S, dq_mu, dq_sqrt, kl, Zs, Xs, ls, unit)


def dense_head_finish(parts, kl_weight):
    dZs, dXs, dvar, S, dq_mu, dq_sqrt, kl, Zs, Xs, ls, unit = parts
    if "dK" in kl:
        S = S - kl_weight * kl["dK"]
    dZs2, dvar2, _ = _kuu_backward(unit, Zs, S)
    dZs = dZs + dZs2
    g = {"Z": dZs / ls, "variance": dvar + dvar2,
         "lengthscales": -(np.sum(dZs * Zs, 0) + np.sum(dXs * Xs, 0)) / ls,
         "q_mu": dq_mu - kl_weight * kl["q_mu"], "q_sqrt": dq_sqrt - kl_weight * kl["q_sqrt"]}
    return dXs / ls, g


def head_kl_backward(layer):
    """SVGP_Layer.KL backward (live Z in the prior)."""
    Lq = layer.q_sqrt
    if layer.white:
        # KL uses sum(q_sqrt^2) and the diagonal only: full matrix as stored
        R = layer.num_outputs
        d = np.tril(Lq)
        for r in range(R):
            d[r] -= np.diag(1.0 / np.diag(Lq[r]))
        return {"q_mu": layer.q_mu.copy(), "q_sqrt": d}
    base = layer.kern.base_kernel
    Lp = np.linalg.cholesky(base.K(layer.Z) + np.eye(layer.num_inducing) * JITTER)
    R = layer.num_outputs
    Kinv_mu = _lsolve_T(Lp, solve_triangular(Lp, layer.q_mu, lower=True))
    Kinv = _lsolve_T(Lp, solve_triangular(Lp, np.eye(layer.num_inducing), lower=True))
    dK = 0.5 * (R * Kinv - Kinv_mu @ Kinv_mu.T)
    dLq = np.zeros_like(Lq)
    for r in range(R):
        KiL = _lsolve_T(Lp, solve_triangular(Lp, Lq[r], lower=True))
        dLq[r] = np.tril(KiL) - np.diag(1.0 / np.diag(Lq[r]))
        dK -= 0.5 * KiL @ KiL.T
    dZ, dvar, dls = _kuu_backward(base, layer.Z, dK)
    return {"q_mu": Kinv_mu, "q_sqrt": dLq, "Z": dZ, "variance": dvar, "lengthscales": dls}


# ---------------------------------------------------------------------------------------------------
# model
# ---------------------------------------------------------------------------------------------------
def elbo_and_grad(model, X, Y, zs, scale=None, kl_weight=1.0):
    """(ELBO, [dict per layer]) of oracle DGP_Base.compute_log_likelihood(X, Y, zs) with explicit noise.
    ``scale`` / ``kl_weight`` restate one shard of a data-parallel step: scale * data_term(shard) - kl_weight * KL
    (scale = num_data / GLOBAL batch, kl_weight = 1 / shards), whose sum over the shards is the full-batch value."""
    X = np.asarray(X, np.float64)
    N, S = X.shape[0], model.num_samples
    Fs, Fm, Fv = model.propagate(X, S=S, zs=zs)
    nl = len(model.layers)
    if scale is None:
        scale = float(model.num_data) / float(N)
    D = Fm[-1].shape[2]
    Yt = np.tile(np.asarray(Y).reshape(1, N), [S, 1]).reshape(S * N)
    mu, var = Fm[-1].reshape(S * N, D), Fv[-1].reshape(S * N, D)
    ve = model.likelihood.variational_expectations(mu, var, Yt)
    elbo = scale * ve.sum() / S - kl_weight * model.KL()
    gm, gv = robustmax_backward(model.likelihood, mu, var, Yt)
    gm *= scale / S
    gv *= scale / S
    grads = [None] * nl
    inputs = [np.tile(X[None], [S, 1, 1])] + Fs[:-1]
    for li in range(nl - 1, -1, -1):
        layer = model.layers[li]
        Xin = inputs[li].reshape(S * N, -1)
        if li == nl - 1 and not hasattr(layer.kern, "base_kernel"):      # dense RBF(ARD) head
            _, parts = dense_head_backward(layer, Xin, gm, gv)
            dX, g = dense_head_finish(parts, kl_weight)
            kl = {}
        elif li == nl - 1:
            dX, g = head_backward(layer, Xin, gm, gv)
            kl = head_kl_backward(layer)
        else:
            dX, g = conv_layer_backward(layer, Xin, gm, gv)
            kl = conv_layer_kl_backward(layer)
        for k_, val in kl.items():
            g[k_] = g[k_] - kl_weight * val
        grads[li] = g
        if li > 0:
            # sample = mean + z sqrt(var + jitter) of the layer below
            z = np.asarray(zs[li - 1], np.float64).reshape(S * N, -1)
            sd = np.sqrt(Fv[li - 1].reshape(S * N, -1) + JITTER)
            gm, gv = dX, dX * z / (2.0 * sd)
    return elbo, grads
