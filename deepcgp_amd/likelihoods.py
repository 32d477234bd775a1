"""MultiClass likelihood with the RobustMax inverse link -- the gpflow.likelihoods.MultiClass(10) the
reference builds at /root/reference/conv_gp/models.py:67 (20 Gauss-Hermite points, epsilon 1e-3)."""
import numpy as np

from . import device as dev


class MultiClass:
    def __init__(self, num_classes=10, epsilon=1e-3):
        self.num_classes = int(num_classes)
        self.epsilon = float(epsilon)

    def variational_expectations(self, Fmu, Fvar, Y):
        ctx = dev.get_context()
        Fmu = np.ascontiguousarray(Fmu, np.float64)
        n, K = Fmu.shape
        if K != self.num_classes:
            raise ValueError("expected %d latent functions, got %d" % (self.num_classes, K))
        Y = np.ascontiguousarray(np.reshape(Y, -1), np.int32)
        if Y.shape[0] != n or Y.min(initial=0) < 0 or Y.max(initial=0) >= K:
            raise ValueError("labels must be %d integers in [0, %d)" % (n, K))
        if n == 0:
            return np.zeros((0,))
        dmu, dvar, dy = ctx.to_device(Fmu), ctx.to_device(Fvar), ctx.to_device(Y, np.int32)
        out = ctx.empty((n,))
        ctx._check(dev.lib().dcgp_robustmax_varexp(ctx.handle, dmu.ptr, dvar.ptr, dy.ptr, n, K, self.epsilon, out.ptr))
        return out.numpy()

    def predict_mean_and_var(self, Fmu, Fvar):
        ctx = dev.get_context()
        Fmu = np.ascontiguousarray(Fmu, np.float64)
        n, K = Fmu.shape
        if n == 0:
            return np.zeros((0, K)), np.zeros((0, K))
        dmu, dvar = ctx.to_device(Fmu), ctx.to_device(Fvar)
        out = ctx.empty((n, K))
        ctx._check(dev.lib().dcgp_robustmax_predict(ctx.handle, dmu.ptr, dvar.ptr, n, K, self.epsilon, out.ptr))
        ps = out.numpy()
        return ps, ps - np.square(ps)
