"""Oracle cross-check (test infrastructure): a SECOND, independently written restatement of the
hot path using textbook closed forms and different primitives (explicit differences instead of
the |x|^2+|z|^2-2xz expansion, dense inverses / slogdet instead of triangular solves,
sliding_window_view instead of slicing loops, scipy.stats.norm instead of erf).

Because the reference itself cannot be imported here (see ``oracle/__init__.py``), agreement of
``oracle/*`` with this file to ~1e-10 is what substitutes for running the reference.
"""
import numpy as np
from numpy.lib.stride_tricks import sliding_window_view
from scipy.stats import norm

JITTER = 1e-3


def patches_NPL(X, f, s):
    """N x P x L with p = oh*W'+ow, l = (kh*f+kw)*C + c."""
    X = np.asarray(X, np.float64)
    N, H, W, C = X.shape
    win = sliding_window_view(X, (f, f), axis=(1, 2))[:, ::s, ::s]      # N,Ho,Wo,C,f,f
    win = np.transpose(win, (0, 1, 2, 4, 5, 3))                         # N,Ho,Wo,kh,kw,C
    return win.reshape(N, win.shape[1] * win.shape[2], f * f * C)


def rbf(A, B, variance, ls):
    d = (A[:, None, :] - B[None, :, :]) / ls
    return variance * np.exp(-0.5 * np.sum(d * d, -1))


def conv_layer_moments(X, f, s, Z, variance, ls, q_mu, q_sqrt, white):
    """Marginal mean/var of every patch response: returns mean, var as N x (P*R) (HWC order)."""
    pat = patches_NPL(X, f, s)
    N, P, L = pat.shape
    M, R = q_mu.shape
    Kuu = rbf(Z, Z, variance, ls) + JITTER * np.eye(M)
    Kinv = np.linalg.inv(Kuu)
    Kfu = rbf(pat.reshape(N * P, L), Z, variance, ls)                   # (N*P) x M
    Lq = np.tril(q_sqrt)
    S = np.einsum('rij,rkj->rik', Lq, Lq)
    base = variance - np.einsum('km,mn,kn->k', Kfu, Kinv, Kfu)
    if not white:
        Ainv = Kfu @ Kinv
        mean = Ainv @ q_mu
        extra = np.einsum('km,rmn,kn->kr', Ainv, S, Ainv)
    else:
        Lc = np.linalg.cholesky(Kuu)
        Aw = Kfu @ np.linalg.inv(Lc).T                                  # rows = (L^-1 kuf)^T
        mean = Aw @ q_mu
        extra = np.einsum('km,rmn,kn->kr', Aw, S, Aw)
    var = base[:, None] + extra
    return mean.reshape(N, P * R), var.reshape(N, P * R)


def gauss_kl(q_mu, q_sqrt, K):
    """sum_r KL[N(mu_r, S_r) || N(0, K)], K=None -> identity."""
    M, R = q_mu.shape
    if K is None:
        K = np.eye(M)
    Kinv = np.linalg.inv(K)
    _, logdetK = np.linalg.slogdet(K)
    total = 0.0
    for r in range(R):
        Lq = np.tril(q_sqrt[r])
        S = Lq @ Lq.T
        _, logdetS = np.linalg.slogdet(S)
        total += 0.5 * (np.trace(Kinv @ S) + q_mu[:, r] @ Kinv @ q_mu[:, r] - M + logdetK - logdetS)
    return total


def conv_kernel_Kzx(X, f, s, Z, variance, ls, w):
    pat = patches_NPL(X, f, s)
    N, P, L = pat.shape
    out = np.zeros((Z.shape[0], N))
    for n in range(N):
        out[:, n] = rbf(Z, pat[n], variance, ls) @ w / P
    return out


def conv_kernel_Kdiag(X, f, s, variance, ls, w):
    pat = patches_NPL(X, f, s)
    N, P, L = pat.shape
    return np.array([w @ rbf(pat[n], pat[n], variance, ls) @ w for n in range(N)]) / P ** 2


def svgp_head_moments(X, f, s, Z, variance, ls, w, q_mu, q_sqrt, white):
    """Head SVGP marginals N x R with the conv kernel."""
    M, R = q_mu.shape
    Ku = rbf(Z, Z, variance, ls) + JITTER * np.eye(M)
    Kuf = conv_kernel_Kzx(X, f, s, Z, variance, ls, w)
    kdiag = conv_kernel_Kdiag(X, f, s, variance, ls, w)
    Kinv = np.linalg.inv(Ku)
    Lq = np.tril(q_sqrt)
    S = np.einsum('rij,rkj->rik', Lq, Lq)
    base = kdiag - np.einsum('mn,mk,kn->n', Kuf, Kinv, Kuf)
    if not white:
        A = Kinv @ Kuf
    else:
        A = np.linalg.inv(np.linalg.cholesky(Ku)) @ Kuf
    mean = A.T @ q_mu
    var = base[:, None] + np.einsum('mn,rmk,kn->nr', A, S, A)
    return mean, var


def robustmax_varexp(mu, var, y, eps=1e-3, K=10, npts=20):
    gx, gw = np.polynomial.hermite.hermgauss(npts)
    out = np.zeros(mu.shape[0])
    for i in range(mu.shape[0]):
        yi = int(y[i])
        p = 0.0
        for x, w in zip(gx, gw):
            t = mu[i, yi] + x * np.sqrt(max(2.0 * var[i, yi], 1e-10))
            prod = 1.0
            for k in range(K):
                if k == yi:
                    continue
                c = norm.cdf((t - mu[i, k]) / np.sqrt(max(var[i, k], 1e-10)))
                prod *= c * (1 - 2e-4) + 1e-4
            p += prod * w / np.sqrt(np.pi)
        out[i] = p * np.log(1 - eps) + (1 - p) * np.log(eps / (K - 1))
    return out


def elbo(model, X, Y, zs, num_data):
    """model: dict(S=, convs=[dict(H,W,C,f,s,Z,Z0,variance,ls,q_mu,q_sqrt,white)...], head=dict(...))."""
    S = model['S']
    N = X.shape[0]
    F = np.tile(X.reshape(N, -1)[None], [S, 1, 1]).reshape(S * N, -1)
    kl = 0.0
    for li, c in enumerate(model['convs']):
        Xi = F.reshape(S * N, c['H'], c['W'], c['C'])
        mean, var = conv_layer_moments(Xi, c['f'], c['s'], c['Z'], c['variance'], c['ls'],
                                       c['q_mu'], c['q_sqrt'], c['white'])
        F = mean + zs[li].reshape(mean.shape) * np.sqrt(var + JITTER)
        Kp = None if c['white'] else rbf(c['Z0'], c['Z0'], c['variance'], c['ls']) + JITTER * np.eye(c['Z0'].shape[0])
        kl += gauss_kl(c['q_mu'], c['q_sqrt'], Kp)
    h = model['head']
    Xi = F.reshape(S * N, h['H'], h['W'], h['C'])
    mean, var = svgp_head_moments(Xi, h['f'], h['s'], h['Z'], h['variance'], h['ls'], h['w'],
                                  h['q_mu'], h['q_sqrt'], h['white'])
    Kp = None if h['white'] else rbf(h['Z'], h['Z'], h['variance'], h['ls']) + JITTER * np.eye(h['Z'].shape[0])
    kl += gauss_kl(h['q_mu'], h['q_sqrt'], Kp)
    ve = robustmax_varexp(mean, var, np.tile(np.asarray(Y).reshape(-1), S))
    data = np.sum(ve.reshape(S, N).mean(0))
    return data * num_data / N - kl, data, kl
