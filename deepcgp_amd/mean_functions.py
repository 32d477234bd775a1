"""Mean functions of the conv layers -- same surface as /root/reference/conv_gp/mean_functions.py:6-41
(``IdentityConv2dMean``, ``Conv2dMean``; constructed at conv_gp/models.py:29-33,95-99 and imported flat at models.py:11), plus the
``Zero`` the reference takes from gpflow (models.py:98).

A mean function is a callable on the NHWC image (``View.mean_view``, conv_gp/views.py:12-16, hands the image through).  The
convolution runs on the device: patches through ``dcgp_extract_patches``, the filter as one MFMA GEMM (``dcgp_gemm_strided``).
``ConvLayer`` recognises a ``Conv2dMean`` whose filter is still the one its constructor built (the reference freezes it:
``conv_mean.set_trainable(False)``, models.py:100) and adds the centre pixel inside its own launch instead
(``csrc/cond.hip`` finalize / ``csrc/conv_fused.hip`` epilogue); any other callable is evaluated through ``__call__`` and added.
"""
import numpy as np

from . import device as dev


class MeanFunction:
    """Stand-in for gpflow.mean_functions.MeanFunction (base class of conv_gp/mean_functions.py:6)."""

    def __call__(self, X):
        raise NotImplementedError

    def set_trainable(self, flag):
        """The reference calls ``conv_mean.set_trainable(False)`` (models.py:100); the filter is never trained on this path."""
        self.trainable = bool(flag)


class Zero(MeanFunction):
    """gpflow.mean_functions.Zero (conv_gp/models.py:98)."""

    def __call__(self, X):
        X = np.asarray(X)
        return np.zeros((X.shape[0], 1))


class IdentityConv2dMean(MeanFunction):
    """conv_gp/mean_functions.py:6-26: VALID convolution, NHWC in, NHWC out; the initial filter copies the centre pixel of every
    input channel into every output map (the sum over input channels)."""

    def __init__(self, filter_size, feature_maps_in, feature_maps_out=1, stride=1):
        self.filter_size = int(filter_size)
        self.feature_maps_in = int(feature_maps_in)
        self.feature_maps_out = int(feature_maps_out)
        self.stride = int(stride)
        self.trainable = True
        self.conv_filter = self._init_filter()

    def _init_filter(self):
        f = np.zeros((self.filter_size, self.filter_size, self.feature_maps_in, self.feature_maps_out))
        f[self.filter_size // 2, self.filter_size // 2, :, :] = 1.0
        return f

    def _conv(self, NHWC_X):
        """tf.nn.conv2d(X, conv_filter, strides=[1, s, s, 1], 'VALID', 'NHWC') (mean_functions.py:16-20) -> N x Ho x Wo x Cout."""
        X = np.ascontiguousarray(NHWC_X, np.float64)
        if X.ndim != 4 or X.shape[3] != self.feature_maps_in:
            raise ValueError("expected N x H x W x %d images, got %s" % (self.feature_maps_in, X.shape))
        N, H, W, Cin = X.shape
        f, s, Cout = self.filter_size, self.stride, self.feature_maps_out
        if f > H or f > W:
            raise ValueError("filter_size %d does not fit %d x %d images" % (f, H, W))
        Ho, Wo = (H - f) // s + 1, (W - f) // s + 1
        if N == 0:
            return np.zeros((0, Ho, Wo, Cout))
        filt = np.ascontiguousarray(self.conv_filter, np.float64)
        if filt.shape != (f, f, Cin, Cout):
            raise ValueError("conv_filter must be %s, got %s" % ((f, f, Cin, Cout), filt.shape))
        ctx = dev.get_context()
        L, P = f * f * Cin, Ho * Wo
        dX = ctx.to_device(X)
        patches = ctx.empty((N * P, L))          # l = (kh f + kw) C + c: the filter's own C-order flattening
        ctx._check(dev.lib().dcgp_extract_patches(ctx.handle, dX.ptr, N, H, W, Cin, f, s, patches.ptr, 0))
        dF = ctx.to_device(filt.reshape(L, Cout))
        out = ctx.empty((N * P, Cout))
        ctx.gemm(patches, (L, 1, 0), dF, (Cout, 1, 0), out, Cout, 0, N * P, Cout, L)
        return out.numpy().reshape(N, Ho, Wo, Cout)

    def __call__(self, NHWC_X):
        return self._conv(NHWC_X)


class Conv2dMean(IdentityConv2dMean):
    """conv_gp/mean_functions.py:28-41: the first output map copies the centre pixel of input channel 0, the others are zero
    mean; the result is flattened to N x (P * feature_maps_out), patch-major like ``ConvLayer``'s mean (layers.py:131)."""

    is_conv2d_mean = True     # what ConvLayer looks for (kept for objects that only quack like one)

    def _init_filter(self):
        f = np.zeros((self.filter_size, self.filter_size, self.feature_maps_in, self.feature_maps_out))
        f[self.filter_size // 2, self.filter_size // 2, 0, 0] = 1.0
        return f

    def has_initial_filter(self):
        """True while ``conv_filter`` is what ``_init_filter`` built: the case the layer kernels add themselves."""
        return np.array_equal(np.asarray(self.conv_filter), self._init_filter())

    def __call__(self, NHWC_X):
        value = self._conv(NHWC_X)
        return value.reshape(value.shape[0], -1)
