"""Oracle (test infrastructure): restatement of /root/reference/conv_gp/kernels.py (head kernels)."""
import numpy as np


class AdditivePatchKernel:
    """conv_gp/kernels.py:15-77 (Kzx/Kdiag/Kzz; the full K() is off the training path)."""

    def __init__(self, base_kernel, view, patch_weights=None):
        self.base_kernel = base_kernel
        self.view = view
        self.patch_length = view.patch_length
        self.patch_count = view.patch_count
        if patch_weights is None or np.size(patch_weights) != self.patch_count:   # :26-27
            patch_weights = np.ones(self.patch_count)
        self.patch_weights = np.array(patch_weights, np.float64)

    def _reshape_X(self, ND_X):                                   # :30-32
        ND_X = np.asarray(ND_X, np.float64)
        return ND_X.reshape([ND_X.shape[0]] + list(self.view.input_size))

    def Kdiag(self, ND_X):                                        # :53-61
        PNL = self.view.extract_patches_PNL(self._reshape_X(ND_X))
        PN = np.stack([w * self.base_kernel.Kdiag(NL) for NL, w in zip(PNL, self.patch_weights)])
        return np.mean(PN, 0)

    def Kzx(self, ML_Z, ND_X):                                    # :63-74
        PNL = self.view.extract_patches_PNL(self._reshape_X(ND_X))
        KMN = np.stack([w * self.base_kernel.K(ML_Z, NL) for NL, w in zip(PNL, self.patch_weights)])
        return np.mean(KMN, 0)

    def Kzz(self, Z):                                             # :76-77
        return self.base_kernel.K(Z)


class ConvKernel(AdditivePatchKernel):
    """conv_gp/kernels.py:79-136 (weighted convolutional kernel of the classification head)."""

    def Kdiag(self, ND_X):                                        # :106-115
        patches = self.view.extract_patches(self._reshape_X(ND_X))     # N x P x L
        w = self.patch_weights
        W = w[None, :] * w[:, None]
        out = np.array([np.sum(self.base_kernel.K(p) * W) for p in patches])
        return out / (self.patch_count ** 2)

    def Kzx(self, Z, ND_X):                                       # :117-133
        NHWC_X = self._reshape_X(ND_X)
        patches = self.view.extract_patches(NHWC_X).reshape(-1, self.patch_length)
        Kzx = self.base_kernel.K(Z, patches)                      # M x (N*P)
        M, N = np.shape(Z)[0], NHWC_X.shape[0]
        Kzx = Kzx.reshape(M, N, self.patch_count) * self.patch_weights
        return np.sum(Kzx, 2) / self.patch_count


def Kuu(feature_Z, kern, jitter=0.0):                             # dispatch :172-174
    return kern.Kzz(feature_Z) + np.eye(np.shape(feature_Z)[0]) * jitter


def Kuf(feature_Z, kern, Xnew):                                   # dispatch :176-178
    return kern.Kzx(feature_Z, Xnew)
