"""conditional() -- same signature and layouts as /root/reference/conv_gp/conditionals.py:6-67."""
import ctypes as C

import numpy as np

from . import device as dev


def conditional(Kmn, Kmm, Knn, f, *, full_cov=False, q_sqrt=None, white=False):
    """q(g1) = int q(g2) p(g1|g2): Kmn P x M x N, Kmm M x M, Knn P x N, f M x R,
    q_sqrt R x M x M (lower triangular).  Returns fmean N x P x R, fvar R x P x N
    (full_cov=True: Knn P x N x N, fvar R x P x N x N -- see _conditional_full_cov)."""
    if full_cov:
        return _conditional_full_cov(Kmn, Kmm, Knn, f, q_sqrt, white)
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


def _conditional_full_cov(Kmn, Kmm, Knn, f, q_sqrt, white):
    """full_cov=True (conv_gp/conditionals.py:36-38,62-63), in the shapes its docstring and comments DECLARE: per patch p an
    N x N covariance,
        fvar[r, p] = Knn[p] - A1[p]^T A1[p] + (Lq_r^T A[p])^T (Lq_r^T A[p]),     A1[p] = inv(Lm) Kmn[p],  A[p] = A1[p] or inv(Lm)^T A1[p],
    returned R x P x N x N.  (As written, the reference contracts ``tensordot(A, A, [[1], [1]])`` over M only, which pairs every
    patch with every patch -- P x N x P x N, not broadcastable against Knn unless N == P -- so its own full_cov branch cannot run;
    the per-patch form is what ConvLayer.conditional_ND's reshape to N x N x num_outputs (layers.py:122-125) expects.)
    Every product is a batched device GEMM (dcgp_gemm_strided); factor and inverse from the operator entry points."""
    from .layers import _potrf
    Kmn = np.ascontiguousarray(Kmn, np.float64)
    if Kmn.ndim != 3:
        raise ValueError("Kmn must be P x M x N")
    P, M, N = Kmn.shape
    f = np.ascontiguousarray(f, np.float64)
    R = f.shape[1]
    Knn = np.ascontiguousarray(Knn, np.float64)
    if np.shape(Kmm) != (M, M) or Knn.shape != (P, N, N) or f.shape[0] != M:
        raise ValueError("inconsistent shapes: Kmn %s Kmm %s Knn %s f %s" % (Kmn.shape, np.shape(Kmm), Knn.shape, f.shape))
    if q_sqrt is not None:
        q_sqrt = np.tril(np.ascontiguousarray(q_sqrt, np.float64))          # matrix_band_part(q_sqrt, -1, 0)
        if q_sqrt.ndim != 3:
            raise ValueError("Bad dimension for q_sqrt: %s" % str(q_sqrt.ndim))
        if q_sqrt.shape != (R, M, M):
            raise ValueError("q_sqrt must be R x M x M")
    if N == 0:
        return np.zeros((0, P, R)), np.zeros((R, P, 0, 0))
    ctx = dev.get_context()
    Lm = _potrf(Kmm)
    dL, dLinv = ctx.to_device(Lm), ctx.empty((M, M))
    ctx._check(dev.lib().dcgp_trtri_lower(ctx.handle, dL.ptr, M, dLinv.ptr))
    dK, dA1 = ctx.to_device(Kmn), ctx.empty((P, M, N))
    ctx.gemm(dLinv, (M, 1, 0), dK, (N, 1, M * N), dA1, N, M * N, M, N, M, batch=P)                     # A1[p] = inv(Lm) Kmn[p]
    dV0 = ctx.to_device(Knn)
    ctx.gemm(dA1, (1, N, M * N), dA1, (N, 1, M * N), dV0, N, N * N, N, N, M, batch=P, alpha=-1.0, accumulate=True)   # Knn[p] - A1^T A1
    if white:
        dA = dA1
    else:
        dA = ctx.empty((P, M, N))
        ctx.gemm(dLinv, (1, M, 0), dA1, (N, 1, M * N), dA, N, M * N, M, N, M, batch=P)                 # A[p] = inv(Lm)^T A1[p]
    dmean, df = ctx.empty((P, N, R)), ctx.to_device(f)
    ctx.gemm(dA, (1, N, M * N), df, (R, 1, 0), dmean, R, N * R, N, R, M, batch=P)                      # A[p]^T f
    fmean = np.ascontiguousarray(np.transpose(dmean.numpy(), (1, 0, 2)))
    V0 = dV0.numpy()
    fvar = np.tile(V0[None], [R, 1, 1, 1])
    if q_sqrt is not None:
        dq, dT = ctx.to_device(q_sqrt), ctx.empty((P, M, N))
        for r in range(R):
            dLq = dev.DeviceArray.__new__(dev.DeviceArray)       # view of q_sqrt[r] inside dq (not owned)
            dLq.ctx, dLq.shape, dLq.dtype, dLq.nbytes, dLq.ptr = ctx, (M, M), np.dtype(np.float64), 0, dq.ptr + r * M * M * 8
            ctx.gemm(dLq, (1, M, 0), dA, (N, 1, M * N), dT, N, M * N, M, N, M, batch=P)                # LTA[r, :, p, :] = Lq_r^T A[p]
            dVr = ctx.to_device(V0)
            ctx.gemm(dT, (1, N, M * N), dT, (N, 1, M * N), dVr, N, N * N, N, N, M, batch=P, accumulate=True)
            fvar[r] = dVr.numpy()
            dLq.ptr = None                                       # the view must not free the parent's memory
    return fmean, fvar
