"""Kernels of the conv-GP path -- same surface as /root/reference/conv_gp/kernels.py, backed by HIP.

``RBF`` stands in for gpflow.kernels.RBF with a scalar lengthscale (what conv_gp/models.py:114-117
builds); the patch kernels keep the reference's names, constructor arguments and method names.
"""
import numpy as np

from . import device as dev

JITTER = 1e-3    # settings.jitter from /root/reference/gpflowrc:11


class RBF:
    """gpflow.kernels.RBF(input_dim, variance, lengthscales, ARD=False) -- parameters + evaluations on device.
    Scalar lengthscale: the base kernel of the conv layers / ConvKernel heads (conv_gp/models.py:114-117,178-185).
    ``ARD=True`` (one lengthscale per input dimension): the dense head of ``--last-kernel rbf`` on the flattened features
    (conv_gp/models.py:160-168), used with ``InducingPoints``; inputs are divided by the lengthscales and the unit-
    lengthscale kernel evaluated, which is gpflow's own formulation."""

    def __init__(self, input_dim, variance=1.0, lengthscales=1.0, ARD=False):
        self.input_dim = int(input_dim)
        self.variance = float(variance)
        self.ARD = bool(ARD)
        if self.ARD:
            ls = np.asarray(lengthscales, np.float64)
            self.lengthscales = np.full(self.input_dim, float(ls)) if ls.ndim == 0 else ls.reshape(self.input_dim).copy()
        else:
            self.lengthscales = float(lengthscales)
        if not (self.variance > 0 and np.all(np.asarray(self.lengthscales) > 0)):
            raise ValueError("variance and lengthscales must be positive")

    def K(self, X, X2=None):
        return self._gram(X, 0.0) if X2 is None else self.Kzx(X, X2)

    def _scaled(self, A):
        A = np.ascontiguousarray(A, np.float64)
        if A.shape[1] != self.input_dim:
            raise ValueError("expected inputs of length %d, got %d" % (self.input_dim, A.shape[1]))
        return np.ascontiguousarray(A / self.lengthscales) if self.ARD else A

    def _gram(self, Z, jitter):
        ctx = dev.get_context()
        Z = self._scaled(Z)
        M, L = Z.shape
        dZ, out = ctx.to_device(Z), ctx.empty((M, M))
        ctx._check(dev.lib().dcgp_kuu_rbf(ctx.handle, dZ.ptr, M, L, self.variance, 1.0 if self.ARD else self.lengthscales,
                                          float(jitter), out.ptr))
        return out.numpy()

    # gpflow's InducingPoints dispatch: Kuu = K(Z) + jitter I, Kuf = K(Z, X)
    def Kzz(self, Z):
        return self._gram(Z, 0.0)

    def Kzx(self, Z, X):
        """K(Z, X), M x N, for dense inputs: every row is a 1 x 1 image with D channels, one patch, weight 1."""
        ctx = dev.get_context()
        Z, X = self._scaled(Z), self._scaled(X)
        (M, D), N = Z.shape, X.shape[0]
        if N == 0:
            return np.zeros((M, 0))
        dX, dZ, dw, out = ctx.to_device(X), ctx.to_device(Z), ctx.to_device(np.ones(1)), ctx.empty((M, N))
        ctx._check(dev.lib().dcgp_convkernel_kzx(ctx.handle, dX.ptr, N, 1, 1, D, 1, 1, dZ.ptr, M, self.variance,
                                                 1.0 if self.ARD else self.lengthscales, dw.ptr, out.ptr))
        return out.numpy()

    def Kdiag(self, X):
        return np.full(np.shape(X)[0], self.variance)

    def _describe(self):
        """{type, variance, p1, p2} of dcgp_model_set_param(..., "base_kernel", ...)."""
        return [0.0, self.variance, 1.0 if self.ARD else self.lengthscales, 0.0]

    def _kuf(self, ctx, dX, N, H, W, C, f, s, dZ, M, out, layout):
        if self.ARD:
            raise NotImplementedError("ARD lengthscales belong to the dense head (conv_gp/models.py:160-168)")
        ctx._check(dev.lib().dcgp_kuf_patches_rbf(ctx.handle, dX.ptr, N, H, W, C, f, s, dZ.ptr, M, self.variance,
                                                  self.lengthscales, out.ptr, layout))


class ArcCosine:
    """gpflow.kernels.ArcCosine(input_dim, order=0) -- the conv layers' base kernel under ``--base-kernel acos``
    (conv_gp/models.py:118-119).  k = variance * (pi - theta) / pi with theta the angle between the augmented inputs,
    <x, z> = weight_variances * x.z + bias_variance; Kdiag = variance.  Order 0 only."""

    def __init__(self, input_dim, order=0, variance=1.0, weight_variances=1.0, bias_variance=1.0):
        if order != 0:
            raise NotImplementedError("only ArcCosine(order=0) is on the accelerated path (the reference uses order 0)")
        self.input_dim = int(input_dim)
        self.order = 0
        self.variance = float(variance)
        self.weight_variances = float(weight_variances)
        self.bias_variance = float(bias_variance)
        if not (self.variance > 0 and self.weight_variances > 0 and self.bias_variance >= 0):
            raise ValueError("variance and weight_variances must be positive, bias_variance non-negative")

    def K(self, X, X2=None):
        if X2 is not None:
            raise NotImplementedError("cross-covariances are evaluated by the fused patch kernels")
        return self._gram(X, 0.0)

    def _gram(self, Z, jitter):
        ctx = dev.get_context()
        Z = np.ascontiguousarray(Z, np.float64)
        M, L = Z.shape
        if L != self.input_dim:
            raise ValueError("expected inputs of length %d, got %d" % (self.input_dim, L))
        dZ, out = ctx.to_device(Z), ctx.empty((M, M))
        ctx._check(dev.lib().dcgp_kuu_acos(ctx.handle, dZ.ptr, M, L, self.variance, self.weight_variances,
                                           self.bias_variance, float(jitter), out.ptr))
        return out.numpy()

    def Kdiag(self, X):
        return np.full(np.shape(X)[0], self.variance)

    def _describe(self):
        return [1.0, self.variance, self.weight_variances, self.bias_variance]

    def _kuf(self, ctx, dX, N, H, W, C, f, s, dZ, M, out, layout):
        ctx._check(dev.lib().dcgp_kuf_patches_acos(ctx.handle, dX.ptr, N, H, W, C, f, s, dZ.ptr, M, self.variance,
                                                   self.weight_variances, self.bias_variance, out.ptr, layout))


class AdditivePatchKernel:
    """K(x, x') = mean_i w_i k(x[i], x'[i]) (conv_gp/kernels.py:15-77); Kzx / Kdiag / Kzz only --
    the full K() of the reference is off the training path."""

    kernel_type = 1

    def __init__(self, base_kernel, view, patch_weights=None):
        self.base_kernel = base_kernel
        self.view = view
        self.patch_length = view.patch_length
        self.patch_count = view.patch_count
        self.image_size = self.view.input_size
        if patch_weights is None or np.size(patch_weights) != self.patch_count:      # kernels.py:26-27
            patch_weights = np.ones(self.patch_count)
        self.patch_weights = np.array(patch_weights, np.float64)

    def _reshape_X(self, ND_X):
        ND_X = np.ascontiguousarray(ND_X, np.float64)
        size = list(self.view.input_size)
        if len(size) == 2:
            size = size + [self.view.feature_maps]
        return ND_X.reshape([ND_X.shape[0]] + size)

    def _geom(self, X):
        N, H, W, Cc = X.shape
        return N, H, W, Cc, self.view.filter_size, self.view.stride

    def Kzx(self, ML_Z, ND_X):
        ctx = dev.get_context()
        X = self._reshape_X(ND_X)
        Z = np.ascontiguousarray(ML_Z, np.float64)
        N, H, W, Cc, f, s = self._geom(X)
        M = Z.shape[0]
        if N == 0:
            return np.zeros((M, 0))
        dX, dZ, dw = ctx.to_device(X), ctx.to_device(Z), ctx.to_device(self.patch_weights)
        out = ctx.empty((M, N))
        ctx._check(dev.lib().dcgp_convkernel_kzx(ctx.handle, dX.ptr, N, H, W, Cc, f, s, dZ.ptr, M,
                                                 self.base_kernel.variance, self.base_kernel.lengthscales, dw.ptr, out.ptr))
        return out.numpy()

    def Kdiag(self, ND_X):
        ctx = dev.get_context()
        N = np.shape(ND_X)[0]
        if N == 0:
            return np.zeros((0,))
        dw, out = ctx.to_device(self.patch_weights), ctx.empty((N,))
        ctx._check(dev.lib().dcgp_additive_kdiag(ctx.handle, N, self.patch_count, self.base_kernel.variance, dw.ptr, out.ptr))
        return out.numpy()

    def Kzz(self, Z):
        return self.base_kernel.K(Z)


class ConvKernel(AdditivePatchKernel):
    """Weighted convolutional kernel of the classification head (conv_gp/kernels.py:79-136)."""

    kernel_type = 0

    def Kdiag(self, ND_X):
        ctx = dev.get_context()
        X = self._reshape_X(ND_X)
        N, H, W, Cc, f, s = self._geom(X)
        if N == 0:
            return np.zeros((0,))
        dX, dw, out = ctx.to_device(X), ctx.to_device(self.patch_weights), ctx.empty((N,))
        ctx._check(dev.lib().dcgp_convkernel_kdiag(ctx.handle, dX.ptr, N, H, W, Cc, f, s, self.base_kernel.variance,
                                                   self.base_kernel.lengthscales, dw.ptr, out.ptr))
        return out.numpy()


def _sample_patches(HW_image, N, patch_size, patch_length):
    """N random patch_size x patch_size patches of ONE image, flattened -- the helper the reference's notebooks import from
    conv_gp/kernels.py (:139-145); same draw order (row, then column, per patch)."""
    rows = np.empty((N, patch_length))
    hi_y, hi_x = HW_image.shape[0] - patch_size, HW_image.shape[1] - patch_size
    for i in range(N):
        top, left = np.random.randint(0, hi_y), np.random.randint(0, hi_x)
        rows[i] = np.reshape(HW_image[top:top + patch_size, left:left + patch_size], patch_length)
    return rows


def kmeans(points, k, max_iter=300, tol=1e-4, rng=None, distinct_start=False):
    """Lloyd's k-means on the device (dcgp_kmeans): sklearn.cluster.KMeans(n_clusters=k, init='random', n_init=1) in
    its own terms -- k random observations as the initial centres, ``tol`` relative to the mean feature variance,
    stop when the summed squared centre shift falls below it.  Returns the [k, d] centres."""
    import ctypes as C
    from . import device as dev
    ctx = dev.get_context()
    P = np.ascontiguousarray(points, np.float64)
    n, d = P.shape
    rng = rng or np.random
    if distinct_start:
        # start from k rows that differ in VALUE where the sample has that many (sklearn relocates empty clusters instead)
        _, first = np.unique(P.round(12), axis=0, return_index=True)
        pool = first if first.size >= k else np.arange(n)
        rows = np.ascontiguousarray(rng.choice(pool, size=k, replace=False), np.int32)
    else:
        rows = np.ascontiguousarray(rng.choice(n, size=k, replace=False), np.int32)
    dP, dC = ctx.to_device(P), ctx.empty((k, d))
    iters = C.c_int(0)
    ctx._check(dev.lib().dcgp_kmeans(ctx.handle, dP.ptr, n, d, k, rows.ctypes.data, int(max_iter), float(tol * np.mean(np.var(P, axis=0))),
                                     dC.ptr, C.byref(iters)))
    return dC.numpy()


def _cluster_patches(NHWC_X, M, patch_size):
    """Inducing patches of PatchInducingFeatures.from_images (conv_gp/kernels.py:147-164): k-means centres of 100 * M patches cut at
    random from random images.  The sample is one vectorised gather (deepcgp_amd.models.draw_patches); Lloyd's iterations run
    on the device (``kmeans``), started from M DISTINCT patches -- on images with large constant regions (MNIST borders) a
    plain random start picks many identical rows, whose clusters then stay empty and leave duplicate inducing patches."""
    from .models import draw_patches
    sample = draw_patches(np.asarray(NHWC_X, np.float64), 100 * M, patch_size)
    return kmeans(sample, M, distinct_start=True)


class InducingPoints:
    """gpflow.features.InducingPoints(Z) -- the dense head's feature (conv_gp/models.py:168)."""

    def __init__(self, Z):
        self.Z = np.array(Z, np.float64)

    def __len__(self):
        return self.Z.shape[0]


class PatchInducingFeatures:
    """conv_gp/kernels.py:166-170 (InducingPointsBase: ``Z`` and ``len``)."""

    def __init__(self, Z):
        self.Z = np.array(Z, np.float64)

    def __len__(self):
        return self.Z.shape[0]

    @classmethod
    def from_images(cls, NHWC_X, M, patch_size):
        return cls(_cluster_patches(NHWC_X, M, patch_size))


def Kuu(feature, kern, jitter=0.0):
    """dispatch at conv_gp/kernels.py:172-174 (patch features); gpflow's default for InducingPoints + a plain kernel."""
    return (kern.base_kernel if hasattr(kern, "base_kernel") else kern)._gram(feature.Z, jitter)


def Kuf(feature, kern, Xnew):
    """dispatch at conv_gp/kernels.py:176-178."""
    return kern.Kzx(feature.Z, Xnew)
