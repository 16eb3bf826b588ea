"""NumPy restatement of the GP training objective for tests.

TEST INFRASTRUCTURE ONLY.  NLML of an exact SE-ARD GP and its analytic gradient w.r.t.
(lengthscales, kernel variance, noise variance); the objective GPflow's GPR.training_loss
minimises in MGPR.optimize (pilco/models/mgpr.py:47-75) is this minus the Gamma log-priors."""
import numpy as np

from .tf_path import se_ard_K


def nlml_and_grad(X, y, ls, var, noise):
    X = np.asarray(X, np.float64)
    N, D = X.shape
    K = se_ard_K(X, None, ls[None, :], np.array([var]))[0]
    Ky = K + noise * np.eye(N)
    L = np.linalg.cholesky(Ky)
    alpha = np.linalg.solve(L.T, np.linalg.solve(L, y))
    nlml = 0.5 * y @ alpha + np.sum(np.log(np.diag(L))) + 0.5 * N * np.log(2 * np.pi)
    iK = np.linalg.inv(Ky)
    W = iK - np.outer(alpha, alpha)
    g = np.empty(D + 2)
    for d in range(D):
        diff2 = (X[:, d][:, None] - X[:, d][None, :]) ** 2
        g[d] = 0.5 * np.sum(W * K * diff2) / ls[d] ** 3
    g[D] = 0.5 * np.sum(W * K) / var
    g[D + 1] = 0.5 * np.trace(W)
    return nlml, g
