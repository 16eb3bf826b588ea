"""NumPy float64 restatement of the reference's TensorFlow/GPflow hot path.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Every function cites the
reference file:line it restates (paths relative to /root/reference).  The
restatement keeps the reference's operation order (including the materialised
(E,E,N,N) tensors in ``predict_given_factorizations``) so that it is an
op-for-op stand-in; ``predict_given_factorizations_pairs`` is the memory-lean
variant of the same arithmetic used for the large benchmark configuration and
as the timed CPU baseline.

Conventions (reference's): row-major float64, ``m`` is a (1,D) row vector,
``s`` is (D,D); a GP layer returns ``(M (1,E), S (E,E), V (D,E))`` where
``V = s^{-1} cov(x, f)``.
"""
from __future__ import annotations

import numpy as np
import scipy.linalg as sla


# --------------------------------------------------------------------------- kernels
def se_ard_K(X1, X2, lengthscales, variance):
    """Per-output squared-exponential ARD Gram matrices, (E,N1,N2).

    pilco/models/mgpr.py:154-157 -> gpflow.kernels.SquaredExponential.K, which
    scales the inputs by 1/lengthscale and uses |a|^2+|b|^2-2ab^T (the same
    expansion as tests/Matlab Code/maha.m:25-26).
    """
    X1 = np.asarray(X1, np.float64)
    X2 = X1 if X2 is None else np.asarray(X2, np.float64)
    ls = np.atleast_2d(np.asarray(lengthscales, np.float64))
    var = np.atleast_1d(np.asarray(variance, np.float64))
    out = np.empty((ls.shape[0], X1.shape[0], X2.shape[0]))
    for a in range(ls.shape[0]):
        A = X1 / ls[a]
        B = X2 / ls[a]
        r2 = (A * A).sum(1)[:, None] + (B * B).sum(1)[None, :] - 2.0 * A @ B.T
        r2 = np.maximum(r2, 0.0)  # gpflow clamps the squared distance at 0
        out[a] = var[a] * np.exp(-0.5 * r2)
    return out


# --------------------------------------------------------------------------- MGPR
def calculate_factorizations(X, Y, lengthscales, variance, noise):
    """iK = (K + sigma_n^2 I)^{-1} (explicit), beta = iK y.

    pilco/models/mgpr.py:81-89: batched Cholesky, cholesky_solve against the
    identity for iK, cholesky_solve against Y^T for beta.
    """
    X = np.asarray(X, np.float64)
    Y = np.asarray(Y, np.float64)
    N = X.shape[0]
    K = se_ard_K(X, None, lengthscales, variance)
    E = K.shape[0]
    iK = np.empty((E, N, N))
    beta = np.empty((E, N))
    eye = np.eye(N)
    for a in range(E):
        c = sla.cho_factor(K[a] + noise[a] * eye, lower=True)
        iK[a] = sla.cho_solve(c, eye)
        beta[a] = sla.cho_solve(c, Y[:, a])
    return iK, beta


def predict_given_factorizations(Xc, lengthscales, variance, m, s, iK, beta):
    """One moment-matching step, op-for-op after pilco/models/mgpr.py:91-149.

    ``Xc`` are the (un-centred) GP inputs: the training inputs X for MGPR, the
    inducing inputs Z for SMGPR (smgpr.py:47-48), the centres for the RBF
    controller.  Materialises the (E,E,N,N) tensors exactly like the reference.
    """
    Xc = np.asarray(Xc, np.float64)
    ls = np.asarray(lengthscales, np.float64)          # (E,D)
    var = np.asarray(variance, np.float64)             # (E,)
    m = np.asarray(m, np.float64).reshape(1, -1)
    s = np.asarray(s, np.float64)
    E, D = ls.shape
    inp = np.broadcast_to((Xc - m)[None], (E,) + Xc.shape)      # mgpr.py:100,151-152

    # mean and inv(s) * input-output covariance          (mgpr.py:102-118)
    iL = np.stack([np.diag(1.0 / ls[a]) for a in range(E)])     # (E,D,D)
    iN = inp @ iL                                               # (E,N,D)
    B = iL @ s[None] @ iL + np.eye(D)                           # (E,D,D)
    # t = iN B^{-1}: solve(B, iN^T, adjoint=True) then transpose (mgpr.py:109-111)
    t = np.stack([np.linalg.solve(B[a].T, iN[a].T).T for a in range(E)])
    lb = np.exp(-0.5 * np.sum(iN * t, -1)) * beta               # (E,N)
    tiL = t @ iL
    c = var / np.sqrt(np.linalg.det(B))                         # (E,)
    M = (lb.sum(-1) * c)[:, None]                               # (E,1)
    V = np.einsum('end,en->ed', tiL, lb) * c[:, None]           # (E,D)

    # predictive covariance                                  (mgpr.py:120-147)
    il2 = 1.0 / np.square(ls)                                   # (E,D)
    R = s[None, None] @ _batched_diag(il2[None, :, :] + il2[:, None, :]) + np.eye(D)
    X1 = inp[None, :, :, :] / np.square(ls[:, None, None, :])   # (E,1->E,N,D)
    X2 = -inp[:, None, :, :] / np.square(ls[None, :, None, :])  # (E,E,N,D)
    X1 = np.broadcast_to(X1, X2.shape) if X1.shape != X2.shape else X1
    Q = np.linalg.solve(R, np.broadcast_to(s, R.shape)) / 2.0   # (E,E,D,D)
    Xs = np.sum(X1 @ Q * X1, -1)                                # (E,E,N)
    X2s = np.sum(X2 @ Q * X2, -1)
    maha = -2.0 * (X1 @ Q) @ np.swapaxes(X2, -1, -2) + Xs[..., :, None] + X2s[..., None, :]
    k = np.log(var)[:, None] - 0.5 * np.sum(np.square(iN), -1)  # (E,N)
    L = np.exp(k[:, None, :, None] + k[None, :, None, :] + maha)  # (E,E,N,N)
    S = np.einsum('ai,abij,bj->ab', beta, L, beta)
    diagL = np.stack([L[a, a] for a in range(E)])               # mgpr.py:143
    S = S - np.diag(np.sum(iK * diagL, axis=(1, 2)))
    S = S / np.sqrt(np.linalg.det(R))
    S = S + np.diag(var)
    S = S - M @ M.T
    return M.T.copy(), S, V.T.copy()


def _batched_diag(v):
    out = np.zeros(v.shape + (v.shape[-1],))
    idx = np.arange(v.shape[-1])
    out[..., idx, idx] = v
    return out


def predict_given_factorizations_pairs(Xc, lengthscales, variance, m, s, iK, beta,
                                       symmetric=True, workers=0):
    """Same arithmetic as ``predict_given_factorizations`` (mgpr.py:91-149) but
    looping over output pairs so that only one (N,N) tile is alive at a time.

    ``symmetric=True`` evaluates b<=a only and mirrors (what gp0.m:87,96 does);
    ``symmetric=False`` evaluates all E^2 pairs like the TF code.  Used for the
    large configuration and as the timed CPU baseline.
    """
    Xc = np.asarray(Xc, np.float64)
    ls = np.asarray(lengthscales, np.float64)
    var = np.asarray(variance, np.float64)
    m = np.asarray(m, np.float64).reshape(1, -1)
    s = np.asarray(s, np.float64)
    E, D = ls.shape
    zeta = Xc - m
    M = np.empty(E)
    V = np.empty((E, D))
    k = np.empty((E, Xc.shape[0]))
    for a in range(E):
        iL = np.diag(1.0 / ls[a])
        iN = zeta @ iL
        B = iL @ s @ iL + np.eye(D)
        t = np.linalg.solve(B.T, iN.T).T
        lb = np.exp(-0.5 * np.sum(iN * t, 1)) * beta[a]
        c = var[a] / np.sqrt(np.linalg.det(B))
        M[a] = lb.sum() * c
        V[a] = (t @ iL).T @ lb * c
        k[a] = np.log(var[a]) - 0.5 * np.sum(iN * iN, 1)
    S = np.zeros((E, E))

    def pair_value(a, b):
        za = zeta / np.square(ls[a])
        wb = -zeta / np.square(ls[b])
        R = s @ np.diag(1.0 / np.square(ls[a]) + 1.0 / np.square(ls[b])) + np.eye(D)
        Q = np.linalg.solve(R, s) / 2.0
        zQ = za @ Q
        maha = (-2.0 * zQ @ wb.T + np.sum(zQ * za, 1)[:, None]
                + np.sum(wb @ Q * wb, 1)[None, :])
        L = np.exp(k[a][:, None] + k[b][None, :] + maha)
        val = beta[a] @ L @ beta[b]
        if a == b:
            val -= np.sum(iK[a] * L)
        return val / np.sqrt(np.linalg.det(R))

    todo = [(a, b) for a in range(E) for b in range(a + 1 if symmetric else E)]
    if workers and workers > 1:
        # the pairs are independent: a thread pool over them (NumPy releases the GIL inside its ufuncs and BLAS calls) is
        # what a multi-threaded CPU runtime does with this graph; every pair's arithmetic -- and so every bit -- is unchanged
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=workers) as ex:
            vals = list(ex.map(lambda ab: pair_value(*ab), todo))
    else:
        vals = [pair_value(a, b) for a, b in todo]
    for (a, b), val in zip(todo, vals):
        S[a, b] = val
        if symmetric:
            S[b, a] = val
    S = S + np.diag(var) - np.outer(M, M)
    return M[None, :].copy(), S, V.T.copy()


# --------------------------------------------------------------------------- SMGPR
def fitc_factorizations(X, Y, Z, lengthscales, variance, noise, jitter=1e-6):
    """FITC factorisation of pilco/models/smgpr.py:24-45 (iK (E,M,M), beta (E,M))."""
    X = np.asarray(X, np.float64)
    Y = np.asarray(Y, np.float64)
    Z = np.asarray(Z, np.float64)
    Mi = Z.shape[0]
    E = np.asarray(lengthscales).shape[0]
    eye = np.eye(Mi)
    Kmm = se_ard_K(Z, None, lengthscales, variance) + jitter * eye[None]
    Kmn = se_ard_K(Z, X, lengthscales, variance)
    iK = np.empty((E, Mi, Mi))
    beta = np.empty((E, Mi))
    for a in range(E):
        L = np.linalg.cholesky(Kmm[a])
        Vm = sla.solve_triangular(L, Kmn[a], lower=True)
        G = variance[a] - np.sum(np.square(Vm), axis=0)
        G = np.sqrt(1.0 + G / noise[a])
        Vm = Vm / G[None, :]
        Am = np.linalg.cholesky(Vm @ Vm.T + noise[a] * eye)
        At = L @ Am
        iAt = sla.solve_triangular(At, eye, lower=True)
        rhs = (Vm / G[None, :]) @ Y[:, a]
        tmp = sla.cho_solve((Am, True), rhs)
        beta[a] = sla.solve_triangular(L, tmp, lower=True, trans='T')
        iB = iAt.T @ iAt * noise[a]
        iK[a] = sla.cho_solve((L, True), eye) - iB
    return iK, beta


# --------------------------------------------------------------------------- controllers
def squash_sin(m, s, max_action=None):
    """Moments of e*sin(x), x~N(m,s): pilco/controllers.py:13-36."""
    m = np.asarray(m, np.float64).reshape(1, -1)
    s = np.asarray(s, np.float64)
    k = m.shape[1]
    e = np.ones((1, k)) if max_action is None else max_action * np.ones((1, k))
    ds = np.diag(s)
    M = e * np.exp(-ds / 2.0) * np.sin(m)
    lq = -(ds[:, None] + ds[None, :]) / 2.0
    q = np.exp(lq)
    S = (np.exp(lq + s) - q) * np.cos(m.T - m) - (np.exp(lq - s) - q) * np.cos(m.T + m)
    S = e * e.T * S / 2.0
    C = e * np.diag(np.exp(-ds / 2.0) * np.cos(m[0]))
    return M, S, C.reshape(k, k)


def linear_controller(m, s, W, b, max_action=1.0, squash=True):
    """pilco/controllers.py:46-58.  W is (k,d), b is (1,k)."""
    m = np.asarray(m, np.float64).reshape(1, -1)
    W = np.asarray(W, np.float64)
    M = m @ W.T + np.asarray(b, np.float64).reshape(1, -1)
    S = W @ s @ W.T
    V = W.T.copy()
    if squash:
        M, S, V2 = squash_sin(M, S, max_action)
        V = V @ V2
    return M, S, V


def rbf_controller(m, s, centres, targets, lengthscales, max_action=1.0, squash=True,
                   variance=None, noise=None):
    """pilco/controllers.py:108-121: deterministic GP (iK zeroed), then squash.

    beta comes from the full factorisation with kernel variance 1 (controllers.py:92)
    and likelihood variance 1e-4 (controllers.py:67,76).
    """
    targets = np.asarray(targets, np.float64)
    E = targets.shape[1]
    var = np.ones(E) if variance is None else np.asarray(variance, np.float64)
    nz = 1e-4 * np.ones(E) if noise is None else np.asarray(noise, np.float64)
    iK, beta = calculate_factorizations(centres, targets, lengthscales, var, nz)
    M, S, V = predict_given_factorizations(centres, lengthscales, var, m, s, 0.0 * iK, beta)
    S = S - np.diag(var - 1e-6)
    if squash:
        M, S, V2 = squash_sin(M, S, max_action)
        V = V @ V2
    return M, S, V


# --------------------------------------------------------------------------- rewards
def exponential_reward(m, s, W=None, t=None):
    """pilco/rewards.py:19-51 (no 1e-12 clamp, unlike reward.m:57)."""
    m = np.asarray(m, np.float64).reshape(1, -1)
    s = np.asarray(s, np.float64)
    d = m.shape[1]
    W = np.eye(d) if W is None else np.asarray(W, np.float64).reshape(d, d)
    t = np.zeros((1, d)) if t is None else np.asarray(t, np.float64).reshape(1, d)
    SW = s @ W
    eye = np.eye(d)
    iSpW = np.linalg.solve((eye + SW).T, W.T).T
    muR = np.exp(-(m - t) @ iSpW @ (m - t).T / 2.0) / np.sqrt(np.linalg.det(eye + SW))
    i2SpW = np.linalg.solve((eye + 2.0 * SW).T, W.T).T
    r2 = np.exp(-(m - t) @ i2SpW @ (m - t).T) / np.sqrt(np.linalg.det(eye + 2.0 * SW))
    sR = r2 - muR @ muR
    return muR.reshape(1, 1), sR.reshape(1, 1)


def linear_reward(m, s, W):
    """pilco/rewards.py:58-61."""
    m = np.asarray(m, np.float64).reshape(1, -1)
    W = np.asarray(W, np.float64).reshape(-1, 1)
    return m @ W, W.T @ np.asarray(s, np.float64) @ W


def combined_rewards(m, s, rewards, coefs=None):
    """pilco/rewards.py:73-81.  ``rewards`` is a list of callables (m,s)->(mu,var)."""
    coefs = np.ones(len(rewards)) if coefs is None else np.asarray(coefs, np.float64)
    mu = 0.0
    var = 0.0
    for r, c in zip(rewards, coefs):
        a, b = r(m, s)
        mu = mu + c * a
        var = var + c ** 2 * b
    return mu, var


# --------------------------------------------------------------------------- rollout
class Model:
    """Bundle of GP state for the rollout restatement (not a reference class)."""

    def __init__(self, X, Y, lengthscales, variance, noise, Z=None, pairs=False):
        self.X = np.asarray(X, np.float64)
        self.Y = np.asarray(Y, np.float64)
        self.ls = np.asarray(lengthscales, np.float64)
        self.var = np.asarray(variance, np.float64)
        self.noise = np.asarray(noise, np.float64)
        self.Z = None if Z is None else np.asarray(Z, np.float64)
        self.pairs = pairs
        self.workers = 0      # > 1: the pair loop of the `pairs` form runs on that many threads (bench.py's CPU baseline)
        self._cache = None

    def factorize(self):
        if self.Z is None:
            return calculate_factorizations(self.X, self.Y, self.ls, self.var, self.noise)
        return fitc_factorizations(self.X, self.Y, self.Z, self.ls, self.var, self.noise)

    def predict_on_noisy_inputs(self, m, s, cache=False):
        """mgpr.py:77-79: re-factorises on every call unless ``cache``."""
        if cache:
            if self._cache is None:
                self._cache = self.factorize()
            iK, beta = self._cache
        else:
            iK, beta = self.factorize()
        pts = self.X if self.Z is None else self.Z
        if self.pairs:
            return predict_given_factorizations_pairs(pts, self.ls, self.var, m, s, iK, beta, workers=self.workers)
        return predict_given_factorizations(pts, self.ls, self.var, m, s, iK, beta)


def propagate(model, controller, m_x, s_x, cache=False):
    """pilco/models/pilco.py:138-153.  ``controller`` is a callable (m,s)->(M,S,V)."""
    m_x = np.asarray(m_x, np.float64).reshape(1, -1)
    s_x = np.asarray(s_x, np.float64)
    m_u, s_u, c_xu = controller(m_x, s_x)
    m = np.concatenate([m_x, m_u], axis=1)
    s1 = np.concatenate([s_x, s_x @ c_xu], axis=1)
    s2 = np.concatenate([(s_x @ c_xu).T, s_u], axis=1)
    s = np.concatenate([s1, s2], axis=0)
    M_dx, S_dx, C_dx = model.predict_on_noisy_inputs(m, s, cache=cache)
    M_x = M_dx + m_x
    S_x = S_dx + s_x + s1 @ C_dx + C_dx.T @ s1.T
    return M_x, S_x


def predict(model, controller, reward, m_x, s_x, n, cache=False):
    """pilco/models/pilco.py:118-136: reward uses the PRE-propagation state."""
    m_x = np.asarray(m_x, np.float64).reshape(1, -1)
    s_x = np.asarray(s_x, np.float64)
    total = np.zeros((1, 1))
    for _ in range(n):
        r = reward(m_x, s_x)[0]
        m_x, s_x = propagate(model, controller, m_x, s_x, cache=cache)
        total = total + r
    return m_x, s_x, total


def no_controller(m_x, s_x):
    """control_dim == 0 (BASELINE config 2 read literally: D == E)."""
    d = np.asarray(m_x).reshape(1, -1).shape[1]
    return np.zeros((1, 0)), np.zeros((0, 0)), np.zeros((d, 0))
