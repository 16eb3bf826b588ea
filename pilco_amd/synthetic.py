"""Seeded synthetic workloads for the parity tests and bench.py.

The recipes are the ones SURVEY.md section 8(d) fixes for BASELINE.json's
configurations; config 1 is the literal recipe of the reference's
tests/test_predictions.py:14-35 (hyper-parameters are fixed positive values
instead of GPflow's L-BFGS result, which both sides of that test share anyway).
Pure NumPy, no device code.
"""
from __future__ import annotations

import numpy as np


def config_c1(noise=(1e-4, 3e-4), seed=0):
    """BASELINE config 1: N=100, D=3, E=2, one moment-matching step."""
    rs = np.random.RandomState(seed)
    d, k = 3, 2
    X0 = rs.rand(100, d)
    A = rs.rand(d, k)
    Y = np.sin(X0).dot(A) + 1e-3 * (rs.rand(100, k) - 0.5)
    m = rs.rand(1, d)
    s = rs.rand(d, d)
    s = s.dot(s.T)
    X = 5 * rs.rand(100, d)          # data replaced after the first predict (test_predictions.py:33-35)
    ls = np.array([[1.3, 0.9, 2.1], [0.7, 1.8, 1.1]])
    var = np.array([1.2, 0.6])
    return dict(X=X, Y=Y, X_first=X0, lengthscales=ls, variance=var,
                noise=np.asarray(noise, np.float64), m=m, s=s)


def config_c2(N=1000, D=10, E=10, noise=1e-2, seed=1234, control_dim=None):
    """BASELINE config 2 (and C2u when D = E + U): SURVEY.md 8(d) 'Synthetic inputs'."""
    rs = np.random.RandomState(seed)
    X = rs.randn(N, D)
    A = rs.randn(D, E) / np.sqrt(D)
    Y = np.sin(X) @ A + 1e-2 * rs.randn(N, E)
    ls = 1.5 + rs.rand(E, D)
    var = 0.5 + rs.rand(E)
    nz = noise * np.ones(E)
    m0 = 0.1 * rs.randn(1, E)
    S0 = 0.1 * np.eye(E)
    U = D - E if control_dim is None else control_dim
    W = 0.1 * rs.randn(U, E)
    b = np.zeros((1, U))
    return dict(X=X, Y=Y, lengthscales=ls, variance=var, noise=nz, m0=m0, S0=S0,
                W=W, b=b, control_dim=U, state_dim=E)


def config_c4(N=5000, M=200, D=10, E=10, noise=1e-2, seed=1234):
    """BASELINE config 4: sparse FITC model, Z = rand(M, D) as in smgpr.py:20."""
    cfg = config_c2(N=N, D=D, E=E, noise=noise, seed=seed)
    rs = np.random.RandomState(seed + 1)
    cfg["Z"] = rs.rand(M, D)
    return cfg


def config_cascade(seed=0, horizon=10):
    """tests/test_cascade.py:18-37 shape: state 2, control 1, N=100, H=10, e=10."""
    rs = np.random.RandomState(seed)
    d, k = 2, 1
    X = rs.rand(100, d + k)
    A = rs.rand(d + k, d)
    Y = np.sin(X).dot(A) + 1e-3 * (rs.rand(100, d) - 0.5)
    m = rs.rand(1, d)
    s = rs.rand(d, d)
    s = s.dot(s.T)
    ls = np.array([[1.1, 2.3, 0.8], [1.9, 0.7, 1.4]])
    var = np.array([0.9, 1.4])
    nz = np.array([2e-4, 1e-4])
    W = rs.randn(k, d)
    b = rs.randn(1, k)
    return dict(X=X, Y=Y, lengthscales=ls, variance=var, noise=nz, m=m, s=s,
                W=W, b=b, max_action=np.array([[10.0]]), horizon=horizon)
