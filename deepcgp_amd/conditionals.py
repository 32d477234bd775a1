"""conditional() -- same signature and layouts as /root/reference/conv_gp/conditionals.py:6-67."""
import ctypes as C

import numpy as np

from . import device as dev


def conditional(Kmn, Kmm, Knn, f, *, full_cov=False, q_sqrt=None, white=False):
    """q(g1) = int q(g2) p(g1|g2): Kmn P x M x N, Kmm M x M, Knn P x N, f M x R,
    q_sqrt R x M x M (lower triangular).  Returns fmean N x P x R, fvar R x P x N."""
    if full_cov:
        raise NotImplementedError("full_cov=True is outside the accelerated hot path (SURVEY.md section 8 f-2)")
    Kmn = np.ascontiguousarray(Kmn, np.float64)
    if Kmn.ndim != 3:
        raise ValueError("Kmn must be P x M x N")
    P, M, N = Kmn.shape
    f = np.ascontiguousarray(f, np.float64)
    R = f.shape[1]
    Knn = np.ascontiguousarray(Knn, np.float64)
    if np.shape(Kmm) != (M, M) or Knn.shape != (P, N) or f.shape[0] != M:
        raise ValueError("inconsistent shapes: Kmn %s Kmm %s Knn %s f %s" % (Kmn.shape, np.shape(Kmm), Knn.shape, f.shape))
    if q_sqrt is not None:
        q_sqrt = np.ascontiguousarray(q_sqrt, np.float64)
        if q_sqrt.ndim != 3:                                    # conditionals.py:59-61
            raise ValueError("Bad dimension for q_sqrt: %s" % str(q_sqrt.ndim))
        if q_sqrt.shape != (R, M, M):
            raise ValueError("q_sqrt must be R x M x M")
    if N == 0:
        return np.zeros((0, P, R)), np.zeros((R, P, 0))
    ctx = dev.get_context()
    d = [ctx.to_device(a) for a in (Kmn, Kmm, Knn, f)]
    dq = ctx.to_device(q_sqrt) if q_sqrt is not None else None
    mean, var = ctx.empty((N, P, R)), ctx.empty((R, P, N))
    info = C.c_int(0)
    rc = dev.lib().dcgp_conditional(ctx.handle, d[0].ptr, d[1].ptr, d[2].ptr, d[3].ptr, dq.ptr if dq else None,
                                    int(bool(white)), P, M, N, R, mean.ptr, var.ptr, C.byref(info))
    ctx._check(rc, info)
    return mean.numpy(), var.numpy()
