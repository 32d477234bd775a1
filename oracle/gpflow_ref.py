"""Oracle (test infrastructure): restatement of the GPflow 1.2.0 pieces the hot path calls.

GPflow 1.2.0 is a pip dependency (``/root/reference/requirements.txt:2``) whose
source is NOT under ``/root/reference``; these functions restate its published
algorithms and are anchored on the reference's call
sites, cited per function.  PARITY UNPINNED (see ``oracle/__init__.py``).
"""
import numpy as np
from scipy.linalg import solve_triangular
from scipy.special import erf

# /root/reference/gpflowrc:6-11 -- the reference runs float64 with jitter 1e-3.
JITTER = 1e-3
FLOAT = np.float64


class RBF:
    """gpflow.kernels.RBF(input_dim, variance, lengthscales, ARD=False).

    Call sites: conv_gp/layers.py:20,29,40,49; conv_gp/kernels.py:114,123,136; construction
    conv_gp/models.py:114-117 (variance=5, lengthscales=5, scalar) and :163-164 (ARD=True on the flattened features
    of the dense head: one lengthscale per input dimension, initialised to the scalar).
    """

    def __init__(self, input_dim, variance=1.0, lengthscales=1.0, ARD=False):
        self.input_dim = int(input_dim)
        self.variance = float(variance)
        self.ARD = bool(ARD)
        if self.ARD:
            ls = np.asarray(lengthscales, FLOAT)
            self.lengthscales = np.full(self.input_dim, float(ls)) if ls.ndim == 0 else ls.reshape(self.input_dim).copy()
        else:
            self.lengthscales = float(lengthscales)

    def square_dist(self, X, X2=None):
        # GPflow 1.2 Stationary.square_dist: scale, then |x|^2 + |x'|^2 - 2 x.x' (no clamp).
        X = np.asarray(X, FLOAT) / self.lengthscales
        Xs = np.sum(np.square(X), axis=1)
        if X2 is None:
            return -2.0 * X @ X.T + Xs[:, None] + Xs[None, :]
        X2 = np.asarray(X2, FLOAT) / self.lengthscales
        X2s = np.sum(np.square(X2), axis=1)
        return -2.0 * X @ X2.T + Xs[:, None] + X2s[None, :]

    def K(self, X, X2=None):
        return self.variance * np.exp(-self.square_dist(X, X2) / 2.0)

    def Kdiag(self, X):
        return np.full(np.shape(X)[0], self.variance, FLOAT)

    # gpflow.features.InducingPoints dispatch (Kuu = K(Z) + jitter I, Kuf = K(Z, X)) in the Kzz / Kzx vocabulary of
    # conv_gp/kernels.py:172-178, so that the dense head goes through the same SVGP_Layer code
    def Kzz(self, Z):
        return self.K(Z)

    def Kzx(self, Z, X):
        return self.K(Z, X)


class ArcCosine:
    """gpflow.kernels.ArcCosine(input_dim, order=0, variance=1, weight_variances=1, bias_variance=1), the conv
    layers' base kernel under ``--base-kernel acos`` (construction conv_gp/models.py:118-119).

    UNVERIFIED restatement of GPflow 1.2 (source not in /root/reference): with <x, z> = sum(weight_variances * x * z)
    + bias_variance, cos = <x, z> / sqrt(<x, x> <z, z>), theta = acos(jitter + (1 - 2 jitter) cos), jitter = 1e-15,
    K = variance * (1 / pi) * J_0(theta) with J_0(theta) = pi - theta; Kdiag = variance * (1 / pi) * J_0(0) = variance.
    """

    def __init__(self, input_dim, order=0, variance=1.0, weight_variances=1.0, bias_variance=1.0):
        if order != 0:
            raise NotImplementedError("order 0 only (what the reference constructs)")
        self.input_dim = int(input_dim)
        self.variance = float(variance)
        self.weight_variances = float(weight_variances)
        self.bias_variance = float(bias_variance)

    def _weighted_product(self, X, X2=None):
        if X2 is None:
            return np.sum(self.weight_variances * np.square(X), axis=1) + self.bias_variance
        return (self.weight_variances * X) @ X2.T + self.bias_variance

    def K(self, X, X2=None):
        X = np.asarray(X, FLOAT)
        den = np.sqrt(self._weighted_product(X))
        if X2 is None:
            X2, den2 = X, den
        else:
            X2 = np.asarray(X2, FLOAT)
            den2 = np.sqrt(self._weighted_product(X2))
        cos_theta = self._weighted_product(X, X2) / den[:, None] / den2[None, :]
        jitter = 1e-15
        theta = np.arccos(jitter + (1.0 - 2.0 * jitter) * cos_theta)
        return self.variance * (1.0 / np.pi) * (np.pi - theta)

    def Kdiag(self, X):
        return np.full(np.shape(X)[0], self.variance, FLOAT)


def gauss_kl(q_mu, q_sqrt, K=None):
    """gpflow.kullback_leiblers.gauss_kl (call sites conv_gp/layers.py:145,147).

    KL[ N(q_mu, q_sqrt q_sqrt^T) || N(0, K) ] summed over the R columns of q_mu;
    ``K is None`` is the whitened prior N(0, I).  q_mu M x R, q_sqrt R x M x M.
    """
    q_mu = np.asarray(q_mu, FLOAT)
    q_sqrt = np.asarray(q_sqrt, FLOAT)
    M, B = q_mu.shape
    white = K is None
    if white:
        alpha = q_mu
    else:
        Lp = np.linalg.cholesky(np.asarray(K, FLOAT))
        alpha = solve_triangular(Lp, q_mu, lower=True)
    Lq = np.tril(q_sqrt)                       # matrix_band_part(q_sqrt, -1, 0)
    Lq_diag = np.diagonal(Lq, axis1=1, axis2=2)
    mahalanobis = np.sum(np.square(alpha))
    constant = -float(B * M)
    logdet_qcov = np.sum(np.log(np.square(Lq_diag)))
    if white:
        trace = np.sum(np.square(Lq))
    else:
        trace = 0.0
        for r in range(B):
            trace += np.sum(np.square(solve_triangular(Lp, Lq[r], lower=True)))
    twoKL = mahalanobis + constant - logdet_qcov + trace
    if not white:
        twoKL += B * np.sum(np.log(np.square(np.diag(Lp))))
    return 0.5 * twoKL


class MultiClass:
    """gpflow.likelihoods.MultiClass(num_classes) with the default RobustMax(eps=1e-3)
    inverse link and 20 Gauss-Hermite points (call site conv_gp/models.py:67)."""

    def __init__(self, num_classes=10, epsilon=1e-3, num_gauss_hermite_points=20):
        self.num_classes = int(num_classes)
        self.epsilon = float(epsilon)
        self.eps_k1 = self.epsilon / (self.num_classes - 1.0)
        self.num_gauss_hermite_points = int(num_gauss_hermite_points)

    def prob_is_largest(self, Y, mu, var):
        gh_x, gh_w = np.polynomial.hermite.hermgauss(self.num_gauss_hermite_points)
        Y = np.asarray(Y).reshape(-1).astype(np.int64)
        mu = np.asarray(mu, FLOAT)
        var = np.asarray(var, FLOAT)
        n = mu.shape[0]
        oh_on = np.zeros((n, self.num_classes), FLOAT)
        oh_on[np.arange(n), Y] = 1.0
        mu_sel = np.sum(oh_on * mu, 1)
        var_sel = np.sum(oh_on * var, 1)
        X = mu_sel[:, None] + gh_x[None, :] * np.sqrt(np.clip(2.0 * var_sel, 1e-10, np.inf))[:, None]
        dist = (X[:, None, :] - mu[:, :, None]) / np.sqrt(np.clip(var, 1e-10, np.inf))[:, :, None]
        cdfs = 0.5 * (1.0 + erf(dist / np.sqrt(2.0)))
        cdfs = cdfs * (1 - 2e-4) + 1e-4
        oh_off = 1.0 - oh_on
        cdfs = cdfs * oh_off[:, :, None] + oh_on[:, :, None]
        return np.prod(cdfs, axis=1) @ (gh_w / np.sqrt(np.pi))

    def variational_expectations(self, Fmu, Fvar, Y):
        p = self.prob_is_largest(Y, Fmu, Fvar)
        return p * np.log(1.0 - self.epsilon) + (1.0 - p) * np.log(self.eps_k1)

    def predict_mean_and_var(self, Fmu, Fvar):
        n = np.shape(Fmu)[0]
        ps = []
        for k in range(self.num_classes):
            p = self.prob_is_largest(np.full(n, k), Fmu, Fvar)
            ps.append(p * (1.0 - self.epsilon) + (1.0 - p) * self.eps_k1)
        ps = np.stack(ps, axis=1)
        return ps, ps - np.square(ps)
