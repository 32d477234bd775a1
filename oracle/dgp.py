"""Oracle (test infrastructure): restatement of the ``doubly_stochastic_dgp`` pieces the hot path's
callers use.

``doubly_stochastic_dgp`` is the un-vendored git submodule ``submodules/Doubly-Stochastic-DGP``
(URL https://github.com/kekeblom/Doubly-Stochastic-DGP, ``/root/reference/.gitmodules:1-3``; the
directory is empty in the mount, pinned SHA unknown).  What follows restates the published
Doubly-Stochastic-DGP algorithm (Salimbeni & Deisenroth 2017) in the fork's call signature, which
the reference fixes at ``conv_gp/models.py:65-70`` (DGP_Base), ``conv_gp/models.py:192-198``
(SVGP_Layer), ``conv_gp/layers.py:52`` (Layer) and ``conv_gp/utils/tensorboard.py:73-74``
(sample_from_conditional returns (samples, mean, var), each S x N x D).  PARITY UNPINNED.
"""
import numpy as np
from scipy.linalg import solve_triangular, cho_solve

from .gpflow_ref import JITTER
from . import kernels as _kernels


def reparameterize(mean, var, z, full_cov=False):
    """doubly_stochastic_dgp.utils.reparameterize: mean + z * sqrt(var + jitter); full_cov=True (UNVERIFIED, recalled -- SURVEY
    App. A): mean S x N x D, var S x N x N x D -> SDN / SDNN, chol(var + jitter I) applied to z per (sample, output), back to SND."""
    if var is None:
        return mean
    if full_cov:
        S, N, D = mean.shape
        m = np.transpose(mean, (0, 2, 1))                                  # S D N
        v = np.transpose(var, (0, 3, 1, 2)) + JITTER * np.eye(N)[None, None]   # S D N N
        chol = np.linalg.cholesky(v)
        f = m + np.matmul(chol, np.transpose(z, (0, 2, 1))[..., None])[..., 0]
        return np.transpose(f, (0, 2, 1))
    return mean + z * (var + JITTER) ** 0.5


def conditional_SND(layer, X, full_cov=False):
    """Layer.conditional_SND: flatten S x N x D -> (S*N) x D, conditional_ND, reshape back; full_cov=True: sample by sample
    (the tf.map_fn branch), var S x N x N x D."""
    S, N, D = X.shape
    if full_cov:
        mv = [layer.conditional_ND(X[s], full_cov=True) for s in range(S)]
        return np.stack([m for m, _ in mv]), np.stack([v for _, v in mv])
    mean, var = layer.conditional_ND(X.reshape(S * N, D))
    return mean.reshape(S, N, layer.num_outputs), var.reshape(S, N, layer.num_outputs)


def sample_from_conditional(layer, X, z=None, rng=None, full_cov=False):
    """Layer.sample_from_conditional(X[S,N,D], z=None) -> (samples, mean, var)."""
    mean, var = conditional_SND(layer, X, full_cov=full_cov)
    if z is None:
        z = (rng or np.random.default_rng()).standard_normal(mean.shape)
    return reparameterize(mean, var, z, full_cov=full_cov), mean, var


class SVGP_Layer:
    """doubly_stochastic_dgp.layers.SVGP_Layer in the fork's signature (conv_gp/models.py:192-198):
    kern has Kzx/Kzz/Kdiag, ``feature_Z`` is M x L; Kuu/Kuf go through the dispatch at
    conv_gp/kernels.py:172-178 with jitter = settings.jitter."""

    def __init__(self, kern, num_outputs, feature_Z, mean_function=None, white=False,
                 q_mu=None, q_sqrt=None):
        self.kern = kern
        self.num_outputs = int(num_outputs)
        self.Z = np.array(feature_Z, np.float64)
        self.num_inducing = self.Z.shape[0]
        self.white = white
        self.mean_function = mean_function
        if q_mu is None:
            q_mu = np.zeros((self.num_inducing, self.num_outputs))
        self.q_mu = np.array(q_mu, np.float64)
        if q_sqrt is None:
            if white:
                q_sqrt = np.tile(np.eye(self.num_inducing)[None], [self.num_outputs, 1, 1])
            else:
                Lu = np.linalg.cholesky(_kernels.Kuu(self.Z, kern, jitter=JITTER))
                q_sqrt = np.tile(Lu[None], [self.num_outputs, 1, 1])
        self.q_sqrt = np.array(q_sqrt, np.float64)

    def conditional_ND(self, X, full_cov=False):
        if full_cov:
            raise NotImplementedError
        Ku = _kernels.Kuu(self.Z, self.kern, jitter=JITTER)
        Lu = np.linalg.cholesky(Ku)
        Kuf = _kernels.Kuf(self.Z, self.kern, X)
        A = solve_triangular(Lu, Kuf, lower=True)
        if not self.white:
            A = solve_triangular(Lu.T, A, lower=False)
        mean = A.T @ self.q_mu
        I = np.eye(self.num_inducing)
        SK = -(I if self.white else Ku)[None] + self.q_sqrt @ np.transpose(self.q_sqrt, (0, 2, 1))
        B = SK @ A[None]                                   # R x M x N
        delta_cov = np.sum(A[None] * B, 1)                 # R x N
        var = (self.kern.Kdiag(X)[None, :] + delta_cov).T  # N x R
        if self.mean_function is not None:
            mean = mean + self.mean_function(X)
        return mean, var

    def KL(self):
        M, R = self.num_inducing, self.num_outputs
        KL = -0.5 * R * M
        KL -= 0.5 * np.sum(np.log(np.diagonal(self.q_sqrt, axis1=1, axis2=2) ** 2))
        if not self.white:
            Ku = _kernels.Kuu(self.Z, self.kern, jitter=JITTER)
            Lu = np.linalg.cholesky(Ku)
            KL += np.sum(np.log(np.diag(Lu))) * R
            for r in range(R):
                KL += 0.5 * np.sum(np.square(solve_triangular(Lu, self.q_sqrt[r], lower=True)))
            KL += 0.5 * np.sum(self.q_mu * cho_solve((Lu, True), self.q_mu))
        else:
            KL += 0.5 * np.sum(np.square(self.q_sqrt))
            KL += 0.5 * np.sum(self.q_mu ** 2)
        return KL


class DGP_Base:
    """doubly_stochastic_dgp.dgp.DGP_Base, explicit-minibatch semantics (the reference evaluates
    forward-only ELBOs on explicit batches at conv_gp/utils/tensorboard.py:22-35)."""

    def __init__(self, X, Y, likelihood, layers, num_samples=1, num_data=None):
        self.X = np.asarray(X, np.float64)
        self.Y = np.asarray(Y)
        self.likelihood = likelihood
        self.layers = list(layers)
        self.num_samples = int(num_samples)
        self.num_data = int(num_data if num_data is not None else self.X.shape[0])

    def propagate(self, X, S=1, zs=None, rng=None):
        sX = np.tile(np.asarray(X, np.float64)[None], [S, 1, 1])
        Fs, Fmeans, Fvars = [], [], []
        F = sX
        zs = zs or [None] * len(self.layers)
        for layer, z in zip(self.layers, zs):
            F, Fmean, Fvar = sample_from_conditional(layer, F, z=z, rng=rng)
            Fs.append(F), Fmeans.append(Fmean), Fvars.append(Fvar)
        return Fs, Fmeans, Fvars

    def E_log_p_Y(self, X, Y, zs=None, rng=None):
        _, Fmeans, Fvars = self.propagate(X, S=self.num_samples, zs=zs, rng=rng)
        Fmean, Fvar = Fmeans[-1], Fvars[-1]
        S, N, D = Fmean.shape
        Yt = np.tile(np.asarray(Y).reshape(1, N), [S, 1]).reshape(S * N)
        ve = self.likelihood.variational_expectations(
            Fmean.reshape(S * N, D), Fvar.reshape(S * N, D), Yt).reshape(S, N)
        return np.mean(ve, 0)

    def data_term(self, X, Y, zs=None, rng=None):
        return float(np.sum(self.E_log_p_Y(X, Y, zs=zs, rng=rng)))

    def KL(self):
        return float(sum(layer.KL() for layer in self.layers))

    def compute_log_likelihood(self, X=None, Y=None, zs=None, rng=None):
        """_build_likelihood: sum_n E_q[log p(y_n|f_n)] * num_data / batch - sum_l KL_l."""
        X = self.X if X is None else X
        Y = self.Y if Y is None else Y
        L = self.data_term(X, Y, zs=zs, rng=rng)
        scale = float(self.num_data) / float(np.shape(X)[0])
        return L * scale - self.KL()

    def predict_y(self, X, S, zs=None, rng=None):
        _, Fmeans, Fvars = self.propagate(X, S=S, zs=zs, rng=rng)
        Fmean, Fvar = Fmeans[-1], Fvars[-1]
        S_, N, D = Fmean.shape
        m, v = self.likelihood.predict_mean_and_var(Fmean.reshape(S_ * N, D), Fvar.reshape(S_ * N, D))
        return m.reshape(S_, N, -1), v.reshape(S_, N, -1)
