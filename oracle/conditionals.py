"""Oracle (test infrastructure): restatement of /root/reference/conv_gp/conditionals.py."""
import numpy as np
from scipy.linalg import solve_triangular


def conditional(Kmn, Kmm, Knn, f, *, full_cov=False, q_sqrt=None, white=False):
    """conv_gp/conditionals.py:6-67, same operation order.

    Kmn P x M x N, Kmm M x M, Knn P x N (full_cov: P x N x N), f M x R, q_sqrt R x M x M.
    Returns fmean N x P x R and fvar R x P x N (full_cov: R x P x N x N).
    """
    Kmn = np.asarray(Kmn, np.float64)
    f = np.asarray(f, np.float64)
    num_func = f.shape[1]
    Lm = np.linalg.cholesky(np.asarray(Kmm, np.float64))                      # :29
    A = np.stack([solve_triangular(Lm, Kmn[p], lower=True)                   # :31-33
                  for p in range(Kmn.shape[0])])
    if full_cov:
        # :36-38 as its comments declare it ("P x N x N", "R x N x N"): per patch, Knn[p] - A[p]^T A[p].  As written the
        # reference's tensordot(A, A, [[1], [1]]) contracts over M only and yields P x N x P x N (every patch against every
        # patch), which does not broadcast against Knn (P x N x N) unless N == P: the branch cannot run as is, so the declared
        # per-patch form -- the one ConvLayer.conditional_ND's reshape to N x N x num_outputs needs (layers.py:122-125) -- is
        # what is restated here and in the product.
        fvar = np.asarray(Knn, np.float64) - np.einsum("pmn,pmk->pnk", A, A)
        fvar = np.tile(fvar[None], [num_func, 1, 1, 1])                       # R x P x N x N
    else:
        fvar = Knn - np.sum(np.square(A), 1)                                  # :40
        fvar = np.tile(fvar[None], [num_func, 1, 1])                          # :41
    if not white:                                                             # :44-47
        A = np.stack([solve_triangular(Lm.T, A[p], lower=False) for p in range(A.shape[0])])
    fmean = np.tensordot(A, f, [[1], [0]])                                    # :50  P x N x R
    fmean = np.transpose(fmean, [1, 0, 2])                                    # :51  N x P x R
    if q_sqrt is not None:
        q_sqrt = np.asarray(q_sqrt, np.float64)
        if q_sqrt.ndim != 3:                                                  # :59-61
            raise ValueError("Bad dimension for q_sqrt: %s" % str(q_sqrt.ndim))
        L = np.tril(q_sqrt)                                                   # :55
        LTA = np.tensordot(L, A, [[1], [1]])                                  # :58  R x M x P x N
        if full_cov:
            fvar = fvar + np.einsum("rmpn,rmpk->rpnk", LTA, LTA)              # :62-63, per patch (see above)
        else:
            fvar = fvar + np.sum(np.square(LTA), 1)                           # :65
    return fmean, fvar
