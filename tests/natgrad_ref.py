"""Test helper: the natural-gradient step of gpflow's NatGradOptimizer written out in NumPy (Salimbeni, Eleftheriadis &
Hensman 2018; natural-parameter form), the checker of dcgp_model_natgrad_step."""
import numpy as np
from scipy.linalg import solve_triangular


def natgrad_reference(mu, Lq, g_mu, g_L, gamma):
    """mu [M, R], Lq [R, M, M] lower, gradients of the ELBO with respect to both -> (new mu, new Lq)."""
    M, R = mu.shape
    new_mu, new_L = np.empty_like(mu), np.zeros_like(Lq)
    I = np.eye(M)
    for r in range(R):
        Lr = np.tril(Lq[r])
        P = np.tril(Lr.T @ np.tril(g_L[r]))                  # Cholesky adjoint of S = L L^T, symmetric form
        P[np.diag_indices(M)] *= 0.5
        Sbar = solve_triangular(Lr, solve_triangular(Lr, P.T, lower=True, trans='T').T, lower=True, trans='T')
        Sbar = 0.5 * (Sbar + Sbar.T)
        Linv = solve_triangular(Lr, I, lower=True)
        Sinv = Linv.T @ Linv
        m = mu[:, r]
        theta1 = Sinv @ m + gamma * (g_mu[:, r] - 2.0 * Sbar @ m)
        prec = Sinv - 2.0 * gamma * Sbar
        Lp = np.linalg.cholesky(0.5 * (prec + prec.T))
        Lpinv = solve_triangular(Lp, I, lower=True)
        S_new = Lpinv.T @ Lpinv
        new_mu[:, r] = S_new @ theta1
        new_L[r] = np.linalg.cholesky(0.5 * (S_new + S_new.T))
    return new_mu, new_L
