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


def moments(X, Y, lengthscales, variance, noise, m, s, dps=40):
    mp.mp.dps = dps
    X = np.asarray(X, np.float64)
    Y = np.asarray(Y, np.float64)
    ls = np.asarray(lengthscales, np.float64)
    N, D = X.shape
    E = Y.shape[1]
    f = mp.mpf
    zeta = [[f(X[i, d]) - f(m[0, d]) for d in range(D)] for i in range(N)]
    sm = mp.matrix(D, D)
    for i in range(D):
        for j in range(D):
            sm[i, j] = f(s[i, j])
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
    Mo = np.array([[float(x) for x in M]])
    So = np.array([[float(S[a, b]) for b in range(E)] for a in range(E)])
    Vo = np.array([[float(V[a][d]) for a in range(E)] for d in range(D)])
    return Mo, So, Vo
