"""NumPy float64 transliteration of the MATLAB PILCO v0.9 routines that the
reference's own tests call through Octave (tests/Matlab Code/*.m).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  These are the ground truth
of tests/test_predictions.py, test_sparse_predictions.py, test_cascade.py,
test_controllers.py and test_rewards.py.  Octave is not installed in the build
image, so the .m files are restated here line by line (same loop structure,
same operation order, MATLAB's column-vector conventions).  ``hyp`` is the
(D+2, E) matrix of log-hyper-parameters [log l; log sigma_f; log sigma_n]
exactly as the tests build it (tests/test_predictions.py:44-48).
"""
from __future__ import annotations

import numpy as np


def hyp_from(lengthscales, variance, noise):
    """tests/test_predictions.py:40-48: hyp = log([l, sqrt(var), sqrt(noise)]).T"""
    ls = np.asarray(lengthscales, np.float64)
    var = np.asarray(variance, np.float64)
    nz = np.asarray(noise, np.float64)
    return np.log(np.hstack((ls, np.sqrt(var[:, None]), np.sqrt(nz[:, None])))).T


def maha(a, b, Q=None):
    """maha.m:22-29: pair-wise (a-b) Q (a-b)^T."""
    if Q is None:
        return (a * a).sum(1)[:, None] + (b * b).sum(1)[None, :] - 2.0 * a @ b.T
    aQ = a @ Q
    return (aQ * a).sum(1)[:, None] + ((b @ Q) * b).sum(1)[None, :] - 2.0 * aQ @ b.T


def _chol_lower(A):
    return np.linalg.cholesky(A)


def _gp_cache(inputs, targets, hyp):
    """gp0.m:46-62 / gp2.m:52-68: K, iK = L'\\(L\\I), beta = L'\\(L\\y)."""
    n, D = inputs.shape
    E = targets.shape[1]
    iK = np.zeros((n, n, E))
    beta = np.zeros((n, E))
    for i in range(E):
        inp = inputs / np.exp(hyp[:D, i])[None, :]
        K = np.exp(2.0 * hyp[D, i] - maha(inp, inp) / 2.0)
        L = _chol_lower(K + np.exp(2.0 * hyp[D + 1, i]) * np.eye(n))
        iK[:, :, i] = np.linalg.solve(L.T, np.linalg.solve(L, np.eye(n)))
        beta[:, i] = np.linalg.solve(L.T, np.linalg.solve(L, targets[:, i]))
    return iK, beta


def _mean_part(inp, hyp, s, beta, D, E):
    """gp0.m:68-82 (identical in gp1.m:93-108, gp2.m:75-89)."""
    n = inp.shape[0] if inp.ndim == 2 else inp.shape[0]
    k = np.zeros((n, E))
    M = np.zeros((E, 1))
    V = np.zeros((D, E))
    for i in range(E):
        x = inp if inp.ndim == 2 else inp[:, :, i]
        iL = np.diag(np.exp(-hyp[:D, i]))
        in_ = x @ iL
        B = iL @ s @ iL + np.eye(D)
        t = np.linalg.solve(B.T, in_.T).T          # in/B
        l = np.exp(-np.sum(in_ * t, 1) / 2.0)
        lb = l * beta[:, i]
        tiL = t @ iL
        c = np.exp(2.0 * hyp[D, i]) / np.sqrt(np.linalg.det(B))
        M[i, 0] = np.sum(lb) * c
        V[:, i] = tiL.T @ lb * c
        k[:, i] = 2.0 * hyp[D, i] - np.sum(in_ * in_, 1) / 2.0
    return M, V, k


def gp0(inputs, targets, hyp, m, s):
    """gp0.m:36-104.  m is (D,1).  Returns M (E,1), S (E,E), V (D,E)."""
    inputs = np.asarray(inputs, np.float64)
    targets = np.asarray(targets, np.float64)
    m = np.asarray(m, np.float64).reshape(-1, 1)
    n, D = inputs.shape
    E = targets.shape[1]
    iK, beta = _gp_cache(inputs, targets, hyp)
    inp = inputs - m.T
    M, V, k = _mean_part(inp, hyp, s, beta, D, E)
    S = np.zeros((E, E))
    for i in range(E):
        ii = inp / np.exp(2.0 * hyp[:D, i])[None, :]
        for j in range(i + 1):
            R = s @ np.diag(np.exp(-2.0 * hyp[:D, i]) + np.exp(-2.0 * hyp[:D, j])) + np.eye(D)
            t = 1.0 / np.sqrt(np.linalg.det(R))
            ij = inp / np.exp(2.0 * hyp[:D, j])[None, :]
            L = np.exp(k[:, i][:, None] + k[:, j][None, :]
                       + maha(ii, -ij, np.linalg.solve(R, s) / 2.0))
            if i == j:
                S[i, i] = t * (beta[:, i] @ L @ beta[:, i] - np.sum(iK[:, :, i] * L))
            else:
                S[i, j] = beta[:, i] @ L @ beta[:, j] * t
                S[j, i] = S[i, j]
        S[i, i] = S[i, i] + np.exp(2.0 * hyp[D, i])
    S = S - M @ M.T
    return M, S, V


def gp2(inputs, targets, hyp, m, s):
    """gp2.m:40-106: mean-function-only GP (RBF network), jitter 1e-6 on diag(S)."""
    inputs = np.asarray(inputs, np.float64)
    targets = np.asarray(targets, np.float64)
    m = np.asarray(m, np.float64).reshape(-1, 1)
    n, D = inputs.shape
    E = targets.shape[1]
    _, beta = _gp_cache(inputs, targets, hyp)
    inp = inputs - m.T
    M, V, k = _mean_part(inp, hyp, s, beta, D, E)
    S = np.zeros((E, E))
    for i in range(E):
        ii = inp / np.exp(2.0 * hyp[:D, i])[None, :]
        for j in range(i + 1):
            R = s @ np.diag(np.exp(-2.0 * hyp[:D, i]) + np.exp(-2.0 * hyp[:D, j])) + np.eye(D)
            t = 1.0 / np.sqrt(np.linalg.det(R))
            ij = inp / np.exp(2.0 * hyp[:D, j])[None, :]
            L = np.exp(k[:, i][:, None] + k[:, j][None, :]
                       + maha(ii, -ij, np.linalg.solve(R, s) / 2.0))
            S[i, j] = t * (beta[:, i] @ L @ beta[:, j])
            S[j, i] = S[i, j]
        S[i, i] = S[i, i] + 1e-6
    S = S - M @ M.T
    return M, S, V


def gp1(inputs, targets, hyp, induce, m, s):
    """gp1.m:37-124: FITC sparse GP.  ``induce`` is (np, D) or (np, D, pE)."""
    inputs = np.asarray(inputs, np.float64)
    targets = np.asarray(targets, np.float64)
    induce = np.asarray(induce, np.float64)
    if induce.ndim == 2:
        induce = induce[:, :, None]
    m = np.asarray(m, np.float64).reshape(-1, 1)
    ridge = 1e-6
    n, D = inputs.shape
    E = targets.shape[1]
    npi, _, pE = induce.shape
    iK2 = np.zeros((npi, npi, E))
    beta = np.zeros((npi, E))
    for i in range(E):
        ell = np.exp(hyp[:D, i])[None, :]
        pinp = induce[:, :, min(i, pE - 1)] / ell
        inp = inputs / ell
        Kmm = np.exp(2.0 * hyp[D, i] - maha(pinp, pinp) / 2.0) + ridge * np.eye(npi)
        Kmn = np.exp(2.0 * hyp[D, i] - maha(pinp, inp) / 2.0)
        L = _chol_lower(Kmm)
        Vm = np.linalg.solve(L, Kmn)
        G = np.exp(2.0 * hyp[D, i]) - np.sum(Vm ** 2, 0)
        G = np.sqrt(1.0 + G / np.exp(2.0 * hyp[D + 1, i]))
        Vm = Vm / G[None, :]
        Am = _chol_lower(np.exp(2.0 * hyp[D + 1, i]) * np.eye(npi) + Vm @ Vm.T)
        At = L @ Am
        iAt = np.linalg.solve(At, np.eye(npi))
        iKi = (np.linalg.solve(Am, Vm / G[None, :]).T @ iAt).T      # (np, n)
        beta[:, i] = iKi @ targets[:, i]
        iB = iAt.T @ iAt * np.exp(2.0 * hyp[D + 1, i])
        iK2[:, :, i] = np.linalg.solve(Kmm, np.eye(npi)) - iB
    inp = np.zeros((npi, D, E))
    for i in range(E):
        inp[:, :, i] = induce[:, :, min(i, pE - 1)] - m.T
    M, V, k = _mean_part(inp, hyp, s, beta, D, E)
    S = np.zeros((E, E))
    for i in range(E):
        ii = inp[:, :, i] / np.exp(2.0 * hyp[:D, i])[None, :]
        for j in range(i + 1):
            R = s @ np.diag(np.exp(-2.0 * hyp[:D, i]) + np.exp(-2.0 * hyp[:D, j])) + np.eye(D)
            t = 1.0 / np.sqrt(np.linalg.det(R))
            ij = inp[:, :, j] / np.exp(2.0 * hyp[:D, j])[None, :]
            L = np.exp(k[:, i][:, None] + k[:, j][None, :]
                       + maha(ii, -ij, np.linalg.solve(R, s) / 2.0))
            if i == j:
                S[i, i] = t * (beta[:, i] @ L @ beta[:, i] - np.sum(iK2[:, :, i] * L))
            else:
                S[i, j] = beta[:, i] @ L @ beta[:, j] * t
                S[j, i] = S[i, j]
        S[i, i] = S[i, i] + np.exp(2.0 * hyp[D, i])
    S = S - M @ M.T
    return M, S, V


def conlin(w, b, m, s):
    """conlin.m:50-62 (moments only).  w (E,D), b (E,1), m (D,1)."""
    w = np.asarray(w, np.float64)
    b = np.asarray(b, np.float64).reshape(-1, 1)
    m = np.asarray(m, np.float64).reshape(-1, 1)
    M = w @ m + b
    S = w @ s @ w.T
    S = (S + S.T) / 2.0
    V = w.T.copy()
    return M, S, V


def gSin(m, v, e=None):
    """gSin.m:33-48 as modified by the reference (all dims, scalar e)."""
    m = np.asarray(m, np.float64).reshape(-1, 1)
    v = np.asarray(v, np.float64)
    d = m.shape[0]
    ev = np.ones((d, 1)) if e is None else np.asarray(e, np.float64).reshape(-1, 1) * np.ones((d, 1))
    vii = np.diag(v).reshape(-1, 1)
    M = ev * np.exp(-vii / 2.0) * np.sin(m)
    lq = -(vii + vii.T) / 2.0
    q = np.exp(lq)
    V = (np.exp(lq + v) - q) * np.cos(m - m.T) - (np.exp(lq - v) - q) * np.cos(m + m.T)
    V = ev @ ev.T * V / 2.0
    C = np.diag((ev * np.exp(-vii / 2.0) * np.cos(m))[:, 0])
    return M, V, C


def reward(m, S, z, W):
    """reward.m:35-57.  Returns (muR, sR) with the 1e-12 clamp of line 57."""
    m = np.asarray(m, np.float64).reshape(-1, 1)
    z = np.asarray(z, np.float64).reshape(-1, 1)
    D = m.shape[0]
    SW = S @ W
    iSpW = np.linalg.solve((np.eye(D) + SW).T, W.T).T           # W/(I+SW)
    muR = float((np.exp(-(m - z).T @ iSpW @ (m - z) / 2.0) / np.sqrt(np.linalg.det(np.eye(D) + SW))).item())
    i2SpW = np.linalg.solve((np.eye(D) + 2.0 * SW).T, W.T).T
    r2 = float((np.exp(-(m - z).T @ i2SpW @ (m - z)) / np.sqrt(np.linalg.det(np.eye(D) + 2.0 * SW))).item())
    sR = r2 - muR ** 2
    if sR < 1e-12:
        sR = 0.0
    return muR, sR


def propagate(m, s, inputs, targets, hyp, w, b, maxU):
    """propagate.m:33-85 as modified by the reference: no trig augmentation, no
    measurement noise (line 56), linear policy + gSin, single gp0 dynamics model,
    all state dims are difference-trained (plant.difi = 1:d, test_cascade.py:68)."""
    m = np.asarray(m, np.float64).reshape(-1, 1)
    D0 = m.shape[0]
    nU = np.asarray(maxU).size
    D1 = D0
    D2 = D1 + nU
    D3 = D2 + D0
    M = np.zeros((D3, 1))
    M[:D0] = m
    S = np.zeros((D3, D3))
    S[:D0, :D0] = s
    mm = M[:D0].copy()
    ss = S[:D0, :D0].copy()          # + diag(sn2) with sn2 = 0 (line 56)
    # 2) control signal
    i = np.arange(D0)
    j = np.arange(D1)
    k = np.arange(D1, D2)
    Mk, Sk, C = conlin(w, b, mm[i], ss[np.ix_(i, i)])
    Mk, Sk, C2 = gSin(Mk, Sk, maxU)
    M[k] = Mk
    S[np.ix_(k, k)] = Sk
    C = C @ C2
    q = S[np.ix_(j, i)] @ C
    S[np.ix_(j, k)] = q
    S[np.ix_(k, j)] = q.T
    # 3) dynamics GP
    ii = np.concatenate([np.arange(D0), np.arange(D1, D2)])
    j = np.arange(D2)
    k = np.arange(D2, D3)
    Mk, Sk, C = gp0(inputs, targets, hyp, M[ii], S[np.ix_(ii, ii)])
    M[k] = Mk
    S[np.ix_(k, k)] = Sk
    q = S[np.ix_(j, ii)] @ C
    S[np.ix_(j, k)] = q
    S[np.ix_(k, j)] = q.T
    # 4) next state
    P = np.hstack([np.zeros((D0, D2)), np.eye(D0)])
    P[:, :D0] = P[:, :D0] + np.eye(D0)      # P(difi,difi) = eye
    Mnext = P @ M
    Snext = P @ S @ P.T
    Snext = (Snext + Snext.T) / 2.0
    return Mnext, Snext


def pred(m, s, H, inputs, targets, hyp, w, b, maxU):
    """pred.m:29-39: returns the (D,H+1) means and (D,D,H+1) covariances."""
    m = np.asarray(m, np.float64).reshape(-1, 1)
    D = m.shape[0]
    Ms = np.zeros((D, H + 1))
    Ss = np.zeros((D, D, H + 1))
    Ms[:, 0] = m[:, 0]
    Ss[:, :, 0] = s
    for i in range(H):
        m, s = propagate(m, s, inputs, targets, hyp, w, b, maxU)
        Ms[:, i + 1] = m[-D:, 0]
        Ss[:, :, i + 1] = s[-D:, -D:]
    return Ms, Ss
