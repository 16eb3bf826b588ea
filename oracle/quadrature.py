"""Independent check of the moment-matching integrals by Gauss-Hermite quadrature.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  This does not follow any
reference code: it integrates the GP posterior directly,

    M_a    = E_x[ mu_a(x) ]
    S_ab   = E_x[ mu_a(x) mu_b(x) ] - M_a M_b + delta_ab E_x[ sigma_a^2(x) ]
    V      = s^{-1} cov_x( x, mu(x) )

for x ~ N(m, s), with mu_a(x) = k_a(x)^T beta_a and
sigma_a^2(x) = sigma_f,a^2 - k_a(x)^T iK_a k_a(x) (function uncertainty only, no
observation noise -- the convention of gp0.m:5 / mgpr.py:145-146).  It pins the
*meaning* of the formulas both restatements implement; accuracy is limited only
by the quadrature order, so it is used for small D.
"""
from __future__ import annotations

import itertools

import numpy as np

from .tf_path import se_ard_K


def gp_moments_quadrature(X, lengthscales, variance, m, s, iK, beta, order=40):
    X = np.asarray(X, np.float64)
    ls = np.asarray(lengthscales, np.float64)
    var = np.asarray(variance, np.float64)
    m = np.asarray(m, np.float64).reshape(-1)
    s = np.asarray(s, np.float64)
    E, D = ls.shape
    nodes, weights = np.polynomial.hermite_e.hermegauss(order)   # weight exp(-x^2/2)
    weights = weights / np.sqrt(2.0 * np.pi)
    Lc = np.linalg.cholesky(s)
    grid = np.array(list(itertools.product(range(order), repeat=D)))
    z = nodes[grid]                                              # (G,D)
    w = np.prod(weights[grid], axis=1)                           # (G,)
    x = m[None, :] + z @ Lc.T
    Kx = se_ard_K(x, X, ls, var)                                 # (E,G,N)
    mu = np.einsum('egn,en->eg', Kx, beta)                       # (E,G)
    sig2 = var[:, None] - np.einsum('egn,enm,egm->eg', Kx, iK, Kx)
    M = mu @ w
    S = np.einsum('ag,bg,g->ab', mu, mu, w) - np.outer(M, M) + np.diag(sig2 @ w)
    cov_xf = np.einsum('gd,eg,g->de', x - m[None, :], mu, w)
    V = np.linalg.solve(s, cov_xf)
    return M[None, :], S, V
