"""Oracle (test infrastructure): restatement of /root/reference/conv_gp/conditionals.py."""
import numpy as np
from scipy.linalg import solve_triangular


def conditional(Kmn, Kmm, Knn, f, *, full_cov=False, q_sqrt=None, white=False):
    """conv_gp/conditionals.py:6-67, same operation order.

    Kmn P x M x N, Kmm M x M, Knn P x N, f M x R, q_sqrt R x M x M.
    Returns fmean N x P x R and fvar R x P x N.
    """
    Kmn = np.asarray(Kmn, np.float64)
    f = np.asarray(f, np.float64)
    num_func = f.shape[1]
    Lm = np.linalg.cholesky(np.asarray(Kmm, np.float64))                      # :29
    A = np.stack([solve_triangular(Lm, Kmn[p], lower=True)                   # :31-33
                  for p in range(Kmn.shape[0])])
    if full_cov:
        # :36-38,62-63 -- off the training path ("next"); the reference's
        # tensordot there yields P x N x P x N, so there is no well-defined behaviour to restate.
        raise NotImplementedError("full_cov=True is outside the hot-path scope")
    fvar = Knn - np.sum(np.square(A), 1)                                      # :40
    fvar = np.tile(fvar[None], [num_func, 1, 1])                              # :41
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
        fvar = fvar + np.sum(np.square(LTA), 1)                               # :65
    return fmean, fvar
