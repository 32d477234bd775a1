"""Patch views -- same surface as /root/reference/conv_gp/views.py (FullView), backed by HIP.

The hot path never materialises patches (the patch gather is fused into the K_uf sweep,
``csrc/rbf.hip``); ``extract_patches`` / ``extract_patches_PNL`` exist for API parity and tests.
"""
import ctypes as C

import numpy as np

from . import device as dev


class View:
    """conv_gp/views.py:6-16."""

    def mean_view(self, NHWC_X, PNL_patches):
        return NHWC_X


class FullView(View):
    """The full view uses all patches of the image (conv_gp/views.py:18-68)."""

    def __init__(self, input_size, filter_size, feature_maps, stride=1):
        self.input_size = list(input_size)
        self.stride = int(stride)
        self.dilation = 1
        self.filter_size = int(filter_size)
        self.feature_maps = int(feature_maps)
        self.patch_shape = [self.filter_size, self.filter_size]
        self.out_image_height, self.out_image_width = self._out_image_size()
        self.patch_count = self._patch_count()
        self.patch_length = self._patch_length()
        if self.out_image_height <= 0 or self.out_image_width <= 0:
            raise ValueError("filter_size %d does not fit input_size %s" % (filter_size, self.input_size))

    def _patch_length(self):
        return self.feature_maps * int(np.prod(self.patch_shape))

    def _patch_count(self):
        return self.out_image_height * self.out_image_width

    def _out_image_size(self):
        h = (self.input_size[0] - self.patch_shape[0]) // self.stride + 1
        w = (self.input_size[1] - self.patch_shape[1]) // self.stride + 1
        return h, w

    def _extract(self, NHWC_X, pnl):
        ctx = dev.get_context()
        X = np.ascontiguousarray(NHWC_X, np.float64)
        N, H, W, Cc = X.shape
        if [H, W] != self.input_size[:2] or Cc != self.feature_maps:
            raise ValueError("expected N x %s x %d images, got %s" % (self.input_size[:2], self.feature_maps, X.shape))
        P, L = self.patch_count, self.patch_length
        if N == 0:
            return np.zeros((P, 0, L) if pnl else (0, P, L))
        dX = ctx.to_device(X)
        out = ctx.empty((P, N, L) if pnl else (N, P, L))
        ctx._check(dev.lib().dcgp_extract_patches(ctx.handle, dX.ptr, N, H, W, Cc, self.filter_size, self.stride,
                                                  out.ptr, int(pnl)))
        return out.numpy()

    def extract_patches(self, NHWC_X):
        """N x patch_count x patch_length (conv_gp/views.py:46-54)."""
        return self._extract(NHWC_X, False)

    def extract_patches_PNL(self, NHWC_X):
        """patch_count x N x patch_length (conv_gp/views.py:40-44)."""
        return self._extract(NHWC_X, True)


class RandomPartialView(View):
    """Name kept so that ``from views import FullView, RandomPartialView`` (conv_gp/models.py:10) resolves under the flat shim.
    The view itself (a random subset of patches, conv_gp/views.py:70-124) is outside this build's scope: it is
    never constructed by ``ModelBuilder``'s flags; constructing it says so."""

    def __init__(self, input_size, filter_size, feature_maps, patch_count):
        raise NotImplementedError("RandomPartialView is out of scope of the MI355X path (no reference flag selects it); use FullView")
