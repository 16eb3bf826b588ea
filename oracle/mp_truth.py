"""Extended-precision (mpmath, 40 digits) evaluation of one moment-matching step.

TEST INFRASTRUCTURE ONLY.  At GPflow's noise floor (sigma_n^2 = 1e-6) the Gram
matrix of the reference's own test (tests/test_predictions.py) has a condition
number of ~1e9 and beta ~1e5: float64 evaluations of the same formula differ
from each other by ~1e-5 relative in S (the executed reference vs the MATLAB
routine: 1.4e-5).  To decide which float64 evaluation is *right*, this module
evaluates the formulas of pilco/models/mgpr.py:81-149 with every operation
(Gram, inverse, exponentials, sums) in 40-digit arithmetic from the float64
inputs.  Used by oracle/gen_golden.py (``*_mp`` keys) for N <= ~150 only.
"""
from __future__ import annotations

import mpmath as mp
import numpy as np


def factorize(X, Y, lengthscales, variance, noise, dps=40):
    """iK_a and beta_a of every output (mgpr.py:81-89) in `dps`-digit arithmetic from the float64 inputs."""
    mp.mp.dps = dps
    X = np.asarray(X, np.float64)
    Y = np.asarray(Y, np.float64)
    ls = np.asarray(lengthscales, np.float64)
    N, D = X.shape
    E = Y.shape[1]
    f = mp.mpf
    iKs, betas = [], []
    for a in range(E):
        K = mp.matrix(N, N)
        for i in range(N):
            for j in range(i, N):
                r2 = sum(((f(X[i, d]) - f(X[j, d])) / f(ls[a, d])) ** 2 for d in range(D))
                K[i, j] = K[j, i] = f(variance[a]) * mp.exp(-r2 / 2)
            K[i, i] += f(noise[a])
        iK = mp.inverse(K)
        iKs.append(iK)
        betas.append(iK * mp.matrix([f(Y[i, a]) for i in range(N)]))
    return iKs, betas


def moments_mp(X, lengthscales, variance, fact, m, sm, dps=40):
    """mgpr.py:91-149 from a cached factorisation; m: list of D mpf, sm: mp.matrix (D, D).  Returns mpf lists / matrices
    (M [E], S (E,E), V [E][D])."""
    mp.mp.dps = dps
    X = np.asarray(X, np.float64)
    ls = np.asarray(lengthscales, np.float64)
    N, D = X.shape
    E = ls.shape[0]
    f = mp.mpf
    iKs, betas = fact
    zeta = [[f(X[i, d]) - m[d] for d in range(D)] for i in range(N)]
    M = [f(0)] * E
    V = [[f(0)] * D for _ in range(E)]
    kk = [[f(0)] * N for _ in range(E)]
    for a in range(E):
        iL = mp.diag([1 / f(ls[a, d]) for d in range(D)])
        B = iL * sm * iL + mp.eye(D)
        iB = mp.inverse(B)
        c = f(variance[a]) / mp.sqrt(mp.det(B))
        for i in range(N):
            iN = mp.matrix([zeta[i][d] / f(ls[a, d]) for d in range(D)])
            t = iB.T * iN                       # row vector iN B^-1 (B symmetric)
            lb = mp.exp(-(iN.T * t)[0] / 2) * betas[a][i]
            M[a] += lb * c
            for d in range(D):
                V[a][d] += t[d] / f(ls[a, d]) * lb * c
            kk[a][i] = mp.log(f(variance[a])) - sum(iN[d] ** 2 for d in range(D)) / 2
    S = mp.matrix(E, E)
    for a in range(E):
        for b in range(E):
            Lam = mp.diag([1 / f(ls[a, d]) ** 2 + 1 / f(ls[b, d]) ** 2 for d in range(D)])
            Rm = sm * Lam + mp.eye(D)
            Q = mp.inverse(Rm) * sm / 2
            za = [mp.matrix([zeta[i][d] / f(ls[a, d]) ** 2 for d in range(D)]) for i in range(N)]
            wb = [mp.matrix([zeta[i][d] / f(ls[b, d]) ** 2 for d in range(D)]) for i in range(N)]
            Qz = [Q * z for z in za]
            Qw = [Q * w for w in wb]
            u = [kk[a][i] + (za[i].T * Qz[i])[0] for i in range(N)]
            v = [kk[b][j] + (wb[j].T * Qw[j])[0] for j in range(N)]
            acc = f(0)
            for i in range(N):
                # (z+w)^T Q (z+w) = z^T Q z + w^T Q w + z^T (Q + Q^T) w
                p = Qz[i] + Q.T * za[i]
                for j in range(N):
                    Lij = mp.exp(u[i] + v[j] + (p.T * wb[j])[0])
                    wgt = betas[a][i] * betas[b][j]
                    if a == b:
                        wgt -= iKs[a][i, j]
                    acc += wgt * Lij
            S[a, b] = acc / mp.sqrt(mp.det(Rm)) - M[a] * M[b]
            if a == b:
                S[a, b] += f(variance[a])
    return M, S, V


def moments(X, Y, lengthscales, variance, noise, m, s, dps=40):
    """One step from float64 inputs, rounded to float64 at the very end: M (1,E), S (E,E), V (D,E)."""
    mp.mp.dps = dps
    f = mp.mpf
    D = np.asarray(X).shape[1]
    E = np.asarray(Y).shape[1]
    sm = mp.matrix(D, D)
    for i in range(D):
        for j in range(D):
            sm[i, j] = f(s[i, j])
    fact = factorize(X, Y, lengthscales, variance, noise, dps)
    M, S, V = moments_mp(X, lengthscales, variance, fact, [f(m[0, d]) for d in range(D)], sm, dps)
    Mo = np.array([[float(x) for x in M]])
    So = np.array([[float(S[a, b]) for b in range(E)] for a in range(E)])
    Vo = np.array([[float(V[a][d]) for a in range(E)] for d in range(D)])
    return Mo, So, Vo


def cascade(X, Y, lengthscales, variance, noise, W, b, max_action, m0, s0, H, dps=40):
    """The H-step rollout of tests/test_cascade.py:17-78 with EVERY operation in `dps`-digit arithmetic: LinearController
    + squash_sin (controllers.py:13-36,52-58), joint Gaussian and propagate (pilco.py:138-153), the moment-matching step
    above, ExponentialReward with W = I, t = 0 (rewards.py:32-39; reward of the pre-propagation state, pilco.py:133).
    Only the inputs are float64.  Returns float64 trajectories M (E, H+1), S (E, E, H+1), running reward (H+1)."""
    mp.mp.dps = dps
    f = mp.mpf
    X = np.asarray(X, np.float64)
    E = np.asarray(Y).shape[1]
    D = X.shape[1]
    U = D - E
    Wm = mp.matrix(U, E)
    for u in range(U):
        for e in range(E):
            Wm[u, e] = f(np.asarray(W)[u, e])
    bm = [f(x) for x in np.ravel(b)]
    ea = [f(x) for x in np.ravel(np.asarray(max_action, np.float64) * np.ones(U))]
    fact = factorize(X, Y, lengthscales, variance, noise, dps)
    mx = [f(x) for x in np.ravel(m0)]
    sx = mp.matrix(E, E)
    for i in range(E):
        for j in range(E):
            sx[i, j] = f(np.asarray(s0)[i, j])
    Ms, Ss, Rs = [[float(x) for x in mx]], [np.array(sx.tolist(), dtype=float)], [0.0]
    total = f(0)
    for _ in range(H):
        # reward of the current state: exp(-m (I + s)^-1 m^T / 2) / sqrt(det(I + s))
        A = mp.eye(E) + sx
        mv = mp.matrix(mx)
        total += mp.exp(-(mv.T * mp.inverse(A) * mv)[0] / 2) / mp.sqrt(mp.det(A))
        # controller: M = m W^T + b, S = W s W^T, V = W^T, then squash_sin
        mu = [sum(Wm[u, e] * mx[e] for e in range(E)) + bm[u] for u in range(U)]
        su = Wm * sx * Wm.T
        Msq = [ea[u] * mp.exp(-su[u, u] / 2) * mp.sin(mu[u]) for u in range(U)]
        Ssq = mp.matrix(U, U)
        for u in range(U):
            for v in range(U):
                lq = -(su[u, u] + su[v, v]) / 2
                q = mp.exp(lq)
                Ssq[u, v] = ea[u] * ea[v] * ((mp.exp(lq + su[u, v]) - q) * mp.cos(mu[u] - mu[v])
                                             - (mp.exp(lq - su[u, v]) - q) * mp.cos(mu[u] + mu[v])) / 2
        C = mp.diag([ea[u] * mp.exp(-su[u, u] / 2) * mp.cos(mu[u]) for u in range(U)])
        cxu = Wm.T * C                                   # (E, U)
        # joint Gaussian of (x, u)
        sc = sx * cxu
        jm = mx + Msq
        js = mp.matrix(D, D)
        for i in range(E):
            for j in range(E):
                js[i, j] = sx[i, j]
            for u in range(U):
                js[i, E + u] = sc[i, u]
                js[E + u, i] = sc[i, u]
        for u in range(U):
            for v in range(U):
                js[E + u, E + v] = Ssq[u, v]
        Mg, Sg, Vg = moments_mp(X, lengthscales, variance, fact, jm, js, dps)
        Cdx = mp.matrix(D, E)
        for a in range(E):
            for d in range(D):
                Cdx[d, a] = Vg[a][d]
        s1 = mp.matrix(E, D)
        for i in range(E):
            for d in range(D):
                s1[i, d] = js[i, d]
        t1 = s1 * Cdx
        sx = Sg + sx + t1 + t1.T
        mx = [Mg[a] + mx[a] for a in range(E)]
        Ms.append([float(x) for x in mx])
        Ss.append(np.array(sx.tolist(), dtype=float))
        Rs.append(float(total))
    return np.array(Ms).T, np.stack(Ss, 2), np.array(Rs)
