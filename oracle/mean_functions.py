"""Oracle (test infrastructure): restatement of /root/reference/conv_gp/mean_functions.py."""
import numpy as np


class Conv2dMean:
    """conv_gp/mean_functions.py:6-41.  VALID convolution with a fixed (non-trainable, conv_gp/models.py:100)
    filter that copies the centre pixel of input channel 0 into output map 0 and leaves the other maps at zero;
    output flattened to N x (P * feature_maps_out), patch-major like ConvLayer's mean (layers.py:131)."""

    def __init__(self, filter_size, feature_maps_in, feature_maps_out=1, stride=1):
        self.filter_size = int(filter_size)
        self.feature_maps_in = int(feature_maps_in)
        self.feature_maps_out = int(feature_maps_out)
        self.stride = int(stride)

    def centre_pixels(self, NHWC_X):
        f, s = self.filter_size, self.stride
        H, W = NHWC_X.shape[1], NHWC_X.shape[2]
        hs = np.arange(f // 2, H - f + f // 2 + 1, s)
        ws = np.arange(f // 2, W - f + f // 2 + 1, s)
        return hs, ws

    def __call__(self, NHWC_X):
        X = np.asarray(NHWC_X, np.float64)
        hs, ws = self.centre_pixels(X)
        out = np.zeros((X.shape[0], len(hs), len(ws), self.feature_maps_out))
        out[:, :, :, 0] = X[:, hs][:, :, ws][:, :, :, 0]
        return out.reshape(X.shape[0], -1)

    def backward(self, NHWC_X, g):
        """adjoint of __call__: g [N, P * feature_maps_out] -> dX [N, H, W, C]."""
        X = np.asarray(NHWC_X, np.float64)
        hs, ws = self.centre_pixels(X)
        g = g.reshape(X.shape[0], len(hs), len(ws), self.feature_maps_out)
        dX = np.zeros_like(X)
        dX[np.ix_(np.arange(X.shape[0]), hs, ws, [0])] = g[:, :, :, :1]
        return dX
