"""Oracle (test infrastructure): restatement of /root/reference/conv_gp/views.py FullView."""
import numpy as np


class FullView:
    """conv_gp/views.py:18-68.  VALID sliding window, dilation 1.

    Patch index p = oh * W' + ow; element index l = (kh * f + kw) * C + c -- the depth
    order of tf.extract_image_patches (rows, cols, channels), views.py:32-38.
    """

    def __init__(self, input_size, filter_size, feature_maps, stride=1):
        self.input_size = list(input_size)                      # views.py:22
        self.stride = int(stride)
        self.dilation = 1
        self.filter_size = int(filter_size)
        self.feature_maps = int(feature_maps)
        self.patch_shape = [self.filter_size, self.filter_size]
        self.out_image_height, self.out_image_width = self._out_image_size()
        self.patch_count = self.out_image_height * self.out_image_width   # views.py:60-63
        self.patch_length = self.feature_maps * self.filter_size ** 2     # views.py:56-58

    def _out_image_size(self):                                  # views.py:65-68
        h = (self.input_size[0] - self.patch_shape[0]) // self.stride + 1
        w = (self.input_size[1] - self.patch_shape[1]) // self.stride + 1
        return h, w

    def extract_patches(self, NHWC_X):
        """N x P x L (views.py:46-54)."""
        X = np.asarray(NHWC_X, np.float64)
        N, H, W, C = X.shape
        f, s = self.filter_size, self.stride
        Ho, Wo = self.out_image_height, self.out_image_width
        out = np.empty((N, Ho, Wo, f, f, C), np.float64)
        for oh in range(Ho):
            for ow in range(Wo):
                out[:, oh, ow] = X[:, oh * s:oh * s + f, ow * s:ow * s + f, :]
        return out.reshape(N, self.patch_count, self.patch_length)

    def extract_patches_PNL(self, NHWC_X):
        """P x N x L (views.py:40-44)."""
        return np.transpose(self.extract_patches(NHWC_X), (1, 0, 2))

    def mean_view(self, NHWC_X, PNL_patches):                   # views.py:14-16
        return NHWC_X
