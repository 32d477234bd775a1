"""Oracle (test infrastructure): restatement of /root/reference/conv_gp/layers.py."""
import numpy as np

from .gpflow_ref import JITTER, gauss_kl
from .conditionals import conditional


class MultiOutputConvKernel:
    """conv_gp/layers.py:12-50."""

    def __init__(self, base_kernel, input_dim, patch_count):
        self.base_kernel = base_kernel
        self.input_dim = input_dim
        self.patch_count = patch_count

    def Kuu(self, ML_Z):                                         # :18-21
        M = np.shape(ML_Z)[0]
        return self.base_kernel.K(ML_Z) + np.eye(M) * JITTER

    def Kuf(self, ML_Z, PNL_patches):                            # :23-32  -> P x M x N
        return np.stack([self.base_kernel.K(ML_Z, NL) for NL in PNL_patches])

    def Kff(self, PNL_patches):                                  # :34-41  -> P x N x N
        return np.stack([self.base_kernel.K(NL) for NL in PNL_patches])

    def Kdiag(self, PNL_patches):                                # :43-50  -> P x N
        return np.stack([self.base_kernel.Kdiag(NL) for NL in PNL_patches])


class ConvLayer:
    """conv_gp/layers.py:52-161.  ``feature_Z`` is the M x L inducing patches
    (PatchInducingFeatures.Z, conv_gp/kernels.py:166-170)."""

    def __init__(self, base_kernel, mean_function, feature_Z, view, white=False, gp_count=1,
                 q_mu=None, q_sqrt=None):
        self.base_kernel = base_kernel
        self.view = view
        self.feature_maps_in = view.feature_maps                 # :63
        self.gp_count = gp_count
        self.patch_count = view.patch_count
        self.patch_length = view.patch_length
        self.num_outputs = self.patch_count * gp_count           # :69
        self.conv_kernel = MultiOutputConvKernel(
            base_kernel, np.prod(view.input_size) * view.feature_maps, self.patch_count)
        self.white = white
        self.Z = np.array(feature_Z, np.float64)
        self.num_inducing = self.Z.shape[0]
        # _build_prior_cholesky :149-152 -- the prior Kuu uses the *initial* Z (read_value at
        # construction) but live hyper-parameters; Z0 is therefore frozen here.
        self.Z0 = self.Z.copy()
        if q_mu is None:
            q_mu = np.zeros((self.num_inducing, gp_count))       # :160-161
        self.q_mu = np.array(q_mu, np.float64)
        if q_sqrt is None:
            if not white:                                        # _init_q_S :154-158
                Lu = np.linalg.cholesky(self.conv_kernel.Kuu(self.Z0))
                q_sqrt = np.tile(Lu[None], [gp_count, 1, 1])
            else:                                                # :89
                q_sqrt = np.tile(np.eye(self.num_inducing)[None], [gp_count, 1, 1])
        self.q_sqrt = np.array(q_sqrt, np.float64)
        self.mean_function = mean_function                       # None == gpflow Zero()

    def conditional_ND(self, ND_X, full_cov=False):              # :96-135
        ND_X = np.asarray(ND_X, np.float64)
        N = ND_X.shape[0]
        NHWC_X = ND_X.reshape(N, self.view.input_size[0], self.view.input_size[1], self.feature_maps_in)
        PNL = self.view.extract_patches_PNL(NHWC_X)
        Kuu = self.conv_kernel.Kuu(self.Z)
        Kuf = self.conv_kernel.Kuf(self.Z, PNL)
        Knn = self.conv_kernel.Kff(PNL) if full_cov else self.conv_kernel.Kdiag(PNL)      # :114-117
        mean, var = conditional(Kuf, Kuu, Knn, self.q_mu, full_cov=full_cov,
                                q_sqrt=self.q_sqrt, white=self.white)
        if full_cov:
            var = np.transpose(var, [2, 3, 1, 0]).reshape(N, N, self.num_outputs)    # :122-125
        else:
            var = np.transpose(var, [2, 1, 0]).reshape(N, self.num_outputs)      # :128-129
        mean = mean.reshape(N, self.num_outputs)                             # :131
        if self.mean_function is not None:                                   # :133-134
            mean = mean + self.mean_function(self.view.mean_view(NHWC_X, PNL))
        return mean, var

    def KL(self):                                                # :137-147
        if self.white:
            return gauss_kl(self.q_mu, self.q_sqrt, K=None)
        return gauss_kl(self.q_mu, self.q_sqrt, self.conv_kernel.Kuu(self.Z0))
