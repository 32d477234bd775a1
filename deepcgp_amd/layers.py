"""Layers of the conv-GP path -- same surface as /root/reference/conv_gp/layers.py plus the two layer
classes the reference takes from doubly_stochastic_dgp (``Layer``, ``SVGP_Layer``), backed by HIP."""
import ctypes as C

import numpy as np

from . import device as dev
from .conditionals import conditional
from .kernels import JITTER, Kuu as _Kuu, Kuf as _Kuf
from .views import FullView


def _potrf(A):
    ctx = dev.get_context()
    A = np.ascontiguousarray(A, np.float64)
    M = A.shape[0]
    dA, info = ctx.to_device(A), C.c_int(0)
    ctx._check(dev.lib().dcgp_potrf_lower(ctx.handle, dA.ptr, M, C.byref(info)), info)
    return dA.numpy()


def gauss_kl(q_mu, q_sqrt, K=None):
    """gpflow.kullback_leiblers.gauss_kl (call sites conv_gp/layers.py:145,147)."""
    ctx = dev.get_context()
    q_mu = np.ascontiguousarray(q_mu, np.float64)
    M, R = q_mu.shape
    dmu, dsq = ctx.to_device(q_mu), ctx.to_device(q_sqrt)
    dK = ctx.to_device(K) if K is not None else None
    out, info = C.c_double(0.0), C.c_int(0)
    rc = dev.lib().dcgp_gauss_kl(ctx.handle, dmu.ptr, dsq.ptr, dK.ptr if dK else None, M, R, C.byref(out), C.byref(info))
    ctx._check(rc, info)
    return out.value


def reparameterize(mean, var, z, full_cov=False):
    """doubly_stochastic_dgp.utils.reparameterize: mean + z * sqrt(var + jitter); full_cov=True: mean S x N x D, var S x N x N x D,
    per (sample, output) mean + chol(var + jitter I) z (off the training path: one device factorisation per matrix)."""
    if var is None:
        return mean
    if full_cov:
        mean, var, z = (np.asarray(a, np.float64) for a in (mean, var, z))
        S, N, D = mean.shape
        out = np.empty_like(mean)
        for s_ in range(S):
            for d_ in range(D):
                out[s_, :, d_] = mean[s_, :, d_] + _potrf(var[s_, :, :, d_] + JITTER * np.eye(N)) @ z[s_, :, d_]
        return out
    ctx = dev.get_context()
    mean = np.ascontiguousarray(mean, np.float64)
    dm, dv, dz = ctx.to_device(mean), ctx.to_device(var), ctx.to_device(z)
    out = ctx.empty(mean.shape)
    ctx._check(dev.lib().dcgp_reparam(ctx.handle, dm.ptr, dv.ptr, dz.ptr, mean.size, JITTER, out.ptr))
    return out.numpy()


class MultiOutputConvKernel:
    """conv_gp/layers.py:12-50."""

    def __init__(self, base_kernel, input_dim, patch_count):
        self.base_kernel = base_kernel
        self.input_dim = input_dim
        self.patch_count = patch_count

    def Kuu(self, ML_Z):
        return self.base_kernel._gram(ML_Z, JITTER)

    def Kuf(self, ML_Z, PNL_patches):
        """patch_count x M x N.  PNL_patches may be the P x N x L array of the reference, or an
        ``(NHWC_X, view)`` pair -- the latter is the fused form that never materialises patches."""
        ctx = dev.get_context()
        Z = np.ascontiguousarray(ML_Z, np.float64)
        M = Z.shape[0]
        if isinstance(PNL_patches, tuple):
            X, view = PNL_patches
            X = np.ascontiguousarray(X, np.float64)
            N, H, W, Cc = X.shape
            f, s = view.filter_size, view.stride
        else:
            # every patch is a 1x1 "image" with L channels, filter 1: same sweep, no gather
            PNL = np.ascontiguousarray(PNL_patches, np.float64)
            P, N, L = PNL.shape
            X = np.ascontiguousarray(np.transpose(PNL, (1, 0, 2))).reshape(N, P, 1, L)
            H, W, Cc, f, s = P, 1, L, 1, 1
        P = ((H - f) // s + 1) * ((W - f) // s + 1)
        if N == 0:
            return np.zeros((P, M, 0))
        dX, dZ, out = ctx.to_device(X), ctx.to_device(Z), ctx.empty((P, M, N))
        self.base_kernel._kuf(ctx, dX, N, H, W, Cc, f, s, dZ, M, out, 0)
        return out.numpy()

    def Kff(self, PNL_patches):
        """P x N x N auto-covariance of the inputs, patch by patch (conv_gp/layers.py:34-41; full_cov=True only)."""
        PNL = np.ascontiguousarray(PNL_patches, np.float64)
        return np.stack([self.base_kernel._gram(PNL[p], 0.0) for p in range(PNL.shape[0])]) if PNL.shape[0] else np.zeros((0,) + PNL.shape[1:2] * 2)

    def Kdiag(self, PNL_patches):
        P, N = np.shape(PNL_patches)[:2]
        return np.full((P, N), self.base_kernel.variance)


class Layer:
    """doubly_stochastic_dgp.layers.Layer: conditional_SND / sample_from_conditional on top of a
    subclass's conditional_ND (subclassed at conv_gp/layers.py:52)."""

    num_outputs = None

    def conditional_ND(self, X, full_cov=False):
        raise NotImplementedError

    def KL(self):
        return 0.0

    def conditional_SND(self, X, full_cov=False):
        """mean S x N x D and var S x N x D; full_cov=True: the samples one by one through conditional_ND(full_cov=True)
        (the reference's tf.map_fn branch), var S x N x N x D."""
        X = np.asarray(X, np.float64)
        S, N, D = X.shape
        if full_cov:
            mv = [self.conditional_ND(X[s_], full_cov=True) for s_ in range(S)]
            return np.stack([m for m, _ in mv]), np.stack([v for _, v in mv])
        mean, var = self.conditional_ND(X.reshape(S * N, D))
        return mean.reshape(S, N, self.num_outputs), var.reshape(S, N, self.num_outputs)

    def sample_from_conditional(self, X, z=None, full_cov=False):
        mean, var = self.conditional_SND(X, full_cov=full_cov)
        if z is None:
            z = np.random.standard_normal(mean.shape)
        samples = reparameterize(mean, var, np.reshape(z, mean.shape), full_cov=full_cov)
        return samples, mean, var


class ConvLayer(Layer):
    """conv_gp/layers.py:52-161.  ``mean_function``: a ``deepcgp_amd.mean_functions`` object as the reference builds them
    (``Conv2dMean(filter_size, feature_maps_in, feature_maps_out, stride)`` or ``Zero()``, conv_gp/models.py:95-99), any callable
    on the NHWC image returning N x num_outputs (or N x H' x W' x gp_count) values, or the aliases None / 'zero' / 'conv2d'.
    A ``Conv2dMean`` with the filter its constructor built is added inside the layer's own launch; any other callable through
    its ``__call__`` (the model-level one-call ELBO path takes the former only)."""

    def __init__(self, base_kernel, mean_function, feature=None, view=None, white=False, gp_count=1,
                 q_mu=None, q_sqrt=None, **kwargs):
        self.base_kernel = base_kernel
        self.view = view
        self.feature_maps_in = self.view.feature_maps
        self.gp_count = int(gp_count)
        self.patch_count = self.view.patch_count
        self.patch_length = self.view.patch_length
        self.num_outputs = self.patch_count * self.gp_count
        self.conv_kernel = MultiOutputConvKernel(base_kernel, int(np.prod(view.input_size)) * view.feature_maps,
                                                 patch_count=self.patch_count)
        self.white = bool(white)
        self.feature = feature
        self.num_inducing = len(feature)
        # the KL prior is built on the *initial* Z (conv_gp/layers.py:149-152)
        self.Z_prior = np.array(feature.Z, np.float64)
        if q_mu is None:
            q_mu = self._initial_q_mu()
        self.q_mu = np.array(q_mu, np.float64)
        if q_sqrt is None:
            if not self.white:
                q_sqrt = self._init_q_S()
            else:
                q_sqrt = np.tile(np.eye(self.num_inducing)[None, :, :], [self.gp_count, 1, 1])
        self.q_sqrt = np.array(q_sqrt, np.float64)
        self.mean_function = mean_function
        self._build_prior_cholesky()

    @property
    def identity_mean(self):
        """The mean the layer kernels add themselves: Conv2dMean's centre pixel of input channel 0 into output map 0
        (conv_gp/mean_functions.py:28-41), geometry equal to the view's."""
        mf = self.mean_function
        if isinstance(mf, str):
            return mf == 'conv2d'
        if mf is None or not getattr(mf, 'is_conv2d_mean', False):
            return False
        if hasattr(mf, 'has_initial_filter') and not mf.has_initial_filter():
            return False
        v = self.view
        return (getattr(mf, 'filter_size', v.filter_size) == v.filter_size and getattr(mf, 'stride', v.stride) == v.stride and
                getattr(mf, 'feature_maps_in', v.feature_maps) == v.feature_maps and
                getattr(mf, 'feature_maps_out', self.gp_count) == self.gp_count)

    @property
    def generic_mean(self):
        """A callable mean function the kernels do not add themselves (``mean + self.mean_function(mean_view)``, layers.py:133-134)."""
        mf = self.mean_function
        from .mean_functions import Zero
        if mf is None or isinstance(mf, (str, Zero)) or self.identity_mean:
            if isinstance(mf, str) and mf not in ('zero', 'conv2d'):
                raise ValueError("mean_function %r: expected a callable, None, 'zero' or 'conv2d'" % (mf,))
            return None
        if not callable(mf):
            raise ValueError("mean_function must be callable, None, 'zero' or 'conv2d'")
        return mf

    def _generic_mean_value(self, X4):
        mf = self.generic_mean
        if mf is None:
            return None
        val = np.asarray(mf(self.view.mean_view(X4, None)), np.float64)
        N = X4.shape[0]
        if val.size != N * self.num_outputs:
            raise ValueError("mean_function returned %s values for %d x %d outputs" % (val.shape, N, self.num_outputs))
        return val.reshape(N, self.num_outputs)

    def conditional_ND(self, ND_X, full_cov=False):
        """mean, var of q(f | m, S), each N x (patch_count * gp_count), HWC column order."""
        if full_cov:
            return self._conditional_full_cov(ND_X)
        return self._forward(ND_X, None)[1:]

    def _conditional_full_cov(self, ND_X):
        """conv_gp/layers.py:114-126 with full_cov=True: mean N x num_outputs, var N x N x num_outputs (per output = (patch, gp)
        an N x N covariance over the inputs; conditionals._conditional_full_cov documents the per-patch reading).  Off the training
        path (predict_f_full_cov-style calls): composed from the operator-level device calls."""
        ND_X = np.ascontiguousarray(ND_X, np.float64)
        N, v = ND_X.shape[0], self.view
        X4 = ND_X.reshape(N, v.input_size[0], v.input_size[1], self.feature_maps_in)
        PNL = v.extract_patches_PNL(X4)
        mean, var = conditional(self.conv_kernel.Kuf(self.feature.Z, (X4, v)), self.conv_kernel.Kuu(self.feature.Z), self.conv_kernel.Kff(PNL),
                                self.q_mu, full_cov=True, q_sqrt=self.q_sqrt, white=self.white)      # N x P x R, R x P x N x N
        var = np.transpose(var, (2, 3, 1, 0)).reshape(N, N, self.num_outputs)
        mean = mean.reshape(N, self.num_outputs)
        if self.identity_mean:
            f, st = v.filter_size, v.stride
            Ho, Wo = (v.input_size[0] - f) // st + 1, (v.input_size[1] - f) // st + 1
            c0 = f // 2
            centre = X4[:, c0:c0 + (Ho - 1) * st + 1:st, c0:c0 + (Wo - 1) * st + 1:st, 0]
            mean = mean.copy()
            mean.reshape(N, v.patch_count, self.gp_count)[:, :, 0] += centre.reshape(N, v.patch_count)
        extra = self._generic_mean_value(X4)
        if extra is not None:
            mean = mean + extra
        return mean, var

    def _forward(self, ND_X, z):
        ctx = dev.get_context()
        ND_X = np.ascontiguousarray(ND_X, np.float64)
        N = ND_X.shape[0]
        v = self.view
        H, W = v.input_size[0], v.input_size[1]
        if ND_X.shape[1] != H * W * self.feature_maps_in:
            raise ValueError("expected N x %d inputs, got %s" % (H * W * self.feature_maps_in, ND_X.shape))
        D = self.num_outputs
        if N == 0:
            return np.zeros((0, D)), np.zeros((0, D)), np.zeros((0, D))
        if not hasattr(self.base_kernel, "lengthscales"):
            return self._forward_composed(ND_X, z)
        dX, dZ = ctx.to_device(ND_X), ctx.to_device(self.feature.Z)
        dmu, dsq = ctx.to_device(self.q_mu), ctx.to_device(self.q_sqrt)
        dz = ctx.to_device(np.reshape(z, (N, D))) if z is not None else None
        ds = ctx.empty((N, D)) if z is not None else None
        dm, dv = ctx.empty((N, D)), ctx.empty((N, D))
        info = C.c_int(0)
        rc = dev.lib().dcgp_conv_layer_forward(
            ctx.handle, dX.ptr, N, H, W, self.feature_maps_in, v.filter_size, v.stride, dZ.ptr,
            self.num_inducing, self.gp_count, self.base_kernel.variance, self.base_kernel.lengthscales,
            dmu.ptr, dsq.ptr, int(self.white), int(self.identity_mean), dz.ptr if dz else None, JITTER,
            ds.ptr if ds else None, dm.ptr, dv.ptr, C.byref(info))
        ctx._check(rc, info)
        smp, mean, var = (ds.numpy() if ds else None), dm.numpy(), dv.numpy()
        extra = self._generic_mean_value(ND_X.reshape(N, H, W, self.feature_maps_in))
        if extra is not None:
            mean = mean + extra
            smp = None if smp is None else smp + extra
        return smp, mean, var

    def _forward_composed(self, ND_X, z):
        """conditional_ND as the reference composes it (conv_gp/layers.py:108-135) from the operator-level calls --
        Kuu, fused patches + Kuf, Kdiag, conditional -- for base kernels the one-call layer operator has no
        signature for (ArcCosine); the model path (dcgp_elbo_forward) handles them natively."""
        N = ND_X.shape[0]
        v = self.view
        X4 = ND_X.reshape(N, v.input_size[0], v.input_size[1], self.feature_maps_in)
        MM_Kuu = self.conv_kernel.Kuu(self.feature.Z)
        PMN_Kuf = self.conv_kernel.Kuf(self.feature.Z, (X4, v))
        Knn = np.full((v.patch_count, N), self.base_kernel.variance)
        mean, var = conditional(PMN_Kuf, MM_Kuu, Knn, self.q_mu, q_sqrt=self.q_sqrt, white=self.white)   # N x P x R, R x P x N
        mean = mean.reshape(N, self.num_outputs)
        var = np.transpose(var, (2, 1, 0)).reshape(N, self.num_outputs)
        if self.identity_mean:
            f, st = v.filter_size, v.stride
            Ho, Wo = (v.input_size[0] - f) // st + 1, (v.input_size[1] - f) // st + 1
            c0 = f // 2
            centre = X4[:, c0:c0 + (Ho - 1) * st + 1:st, c0:c0 + (Wo - 1) * st + 1:st, 0]
            mean = mean.copy()
            mean.reshape(N, v.patch_count, self.gp_count)[:, :, 0] += centre.reshape(N, v.patch_count)
        extra = self._generic_mean_value(X4)
        if extra is not None:
            mean = mean + extra
        sample = None if z is None else reparameterize(mean, var, np.reshape(z, mean.shape))
        return sample, mean, var

    def sample_from_conditional(self, X, z=None, full_cov=False):
        if full_cov:   # (the call shape of conv_gp/utils/tensorboard.py:73-81) off the one-launch path: Layer's generic route
            return Layer.sample_from_conditional(self, X, z=z, full_cov=True)
        X = np.asarray(X, np.float64)
        S, N, D = X.shape
        if z is None:
            z = np.random.standard_normal((S, N, self.num_outputs))
        s, m, v = self._forward(X.reshape(S * N, D), np.reshape(z, (S * N, self.num_outputs)))
        shape = (S, N, self.num_outputs)
        return s.reshape(shape), m.reshape(shape), v.reshape(shape)

    def KL(self):
        """KL[q(u) || p(u)] summed over the gp_count GPs (conv_gp/layers.py:137-147)."""
        if self.white:
            return gauss_kl(self.q_mu, self.q_sqrt, K=None)
        return gauss_kl(self.q_mu, self.q_sqrt, self.conv_kernel.Kuu(self.Z_prior))

    def _build_prior_cholesky(self):
        self.MM_Ku_prior = self.conv_kernel.Kuu(self.Z_prior)

    def _init_q_S(self):
        MM_Lu = _potrf(self.conv_kernel.Kuu(self.feature.Z))
        return np.tile(MM_Lu[None, :, :], [self.gp_count, 1, 1])

    def _initial_q_mu(self):
        return np.zeros((self.num_inducing, self.gp_count))


class SVGP_Layer(Layer):
    """doubly_stochastic_dgp.layers.SVGP_Layer in the fork's signature (conv_gp/models.py:192-198)."""

    def __init__(self, kern, num_outputs, feature=None, mean_function=None, white=False, q_mu=None, q_sqrt=None):
        self.kern = kern
        self.num_outputs = int(num_outputs)
        self.feature = feature
        self.num_inducing = len(feature)
        self.white = bool(white)
        self.mean_function = mean_function
        if q_mu is None:
            q_mu = np.zeros((self.num_inducing, self.num_outputs))
        self.q_mu = np.array(q_mu, np.float64)
        if q_sqrt is None:
            if self.white:
                q_sqrt = np.tile(np.eye(self.num_inducing)[None], [self.num_outputs, 1, 1])
            else:
                Lu = _potrf(_Kuu(self.feature, self.kern, jitter=JITTER))
                q_sqrt = np.tile(Lu[None], [self.num_outputs, 1, 1])
        self.q_sqrt = np.array(q_sqrt, np.float64)

    def conditional_ND(self, X, full_cov=False):
        if full_cov:
            raise NotImplementedError("full_cov=True is outside the accelerated hot path")
        ctx = dev.get_context()
        X = np.ascontiguousarray(X, np.float64)
        N, M, R = X.shape[0], self.num_inducing, self.num_outputs
        if N == 0:
            return np.zeros((0, R)), np.zeros((0, R))
        Ku = _Kuu(self.feature, self.kern, jitter=JITTER)
        Kuf = _Kuf(self.feature, self.kern, X)
        Kdiag = self.kern.Kdiag(X)
        d = [ctx.to_device(a) for a in (Kuf, Ku, Kdiag, self.q_mu, self.q_sqrt)]
        mean, var, info = ctx.empty((N, R)), ctx.empty((N, R)), C.c_int(0)
        rc = dev.lib().dcgp_svgp_conditional(ctx.handle, d[0].ptr, d[1].ptr, d[2].ptr, d[3].ptr, d[4].ptr,
                                             int(self.white), M, N, R, mean.ptr, var.ptr, C.byref(info))
        ctx._check(rc, info)
        return mean.numpy(), var.numpy()

    def KL(self):
        if self.white:
            return gauss_kl(self.q_mu, self.q_sqrt, K=None)
        return gauss_kl(self.q_mu, self.q_sqrt, _Kuu(self.feature, self.kern, jitter=JITTER))
