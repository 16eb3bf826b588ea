"""Gradient oracle: the reference's hot path restated in torch float64 on the CPU so
that autograd plays the role TensorFlow's reverse mode plays in the reference
(pilco/models/pilco.py:85-90 differentiates training_loss through the while_loop).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Mirrors oracle/tf_path.py
(mgpr.py:91-149, controllers.py:13-58, rewards.py:19-51, pilco.py:118-153); the
forward values are checked against tf_path in tests/test_oracle.py.
"""
from __future__ import annotations

import torch

DT = torch.float64


def t(x):
    return torch.as_tensor(x, dtype=DT)


def predict_given_factorizations(X, ls, var, m, s, iK, beta):
    """mgpr.py:91-149 with one (N,N) tile per output pair (b <= a)."""
    X, ls, var, iK, beta = t(X), t(ls), t(var), t(iK), t(beta)
    m = m.reshape(1, -1)
    E, D = ls.shape
    zeta = X - m
    eye = torch.eye(D, dtype=DT)
    M, V, k = [], [], []
    for a in range(E):
        iL = torch.diag(1.0 / ls[a])
        iN = zeta @ iL
        B = iL @ s @ iL + eye
        tt = torch.linalg.solve(B.T, iN.T).T
        lb = torch.exp(-0.5 * torch.sum(iN * tt, 1)) * beta[a]
        c = var[a] / torch.sqrt(torch.linalg.det(B))
        M.append(lb.sum() * c)
        V.append((tt @ iL).T @ lb * c)
        k.append(torch.log(var[a]) - 0.5 * torch.sum(iN * iN, 1))
    M = torch.stack(M)
    V = torch.stack(V, dim=1)
    S = torch.zeros((E, E), dtype=DT)
    rows = []
    for a in range(E):
        za = zeta / ls[a] ** 2
        row = []
        for b in range(a + 1):
            wb = -zeta / ls[b] ** 2
            R = s @ torch.diag(1.0 / ls[a] ** 2 + 1.0 / ls[b] ** 2) + eye
            Q = torch.linalg.solve(R, s) / 2.0
            zQ = za @ Q
            maha = -2.0 * zQ @ wb.T + torch.sum(zQ * za, 1)[:, None] + torch.sum(wb @ Q * wb, 1)[None, :]
            L = torch.exp(k[a][:, None] + k[b][None, :] + maha)
            val = beta[a] @ L @ beta[b]
            if a == b:
                val = val - torch.sum(iK[a] * L)
            row.append(val / torch.sqrt(torch.linalg.det(R)))
        rows.append(row)
    S = torch.stack([torch.stack([rows[max(a, b)][min(a, b)] for b in range(E)]) for a in range(E)])
    S = S + torch.diag(var) - torch.outer(M, M)
    return M[None, :], S, V


def squash_sin(m, s, max_action):
    k = m.shape[1]
    e = t(max_action) * torch.ones((1, k), dtype=DT)
    ds = torch.diagonal(s)
    M = e * torch.exp(-ds / 2.0) * torch.sin(m)
    lq = -(ds[:, None] + ds[None, :]) / 2.0
    q = torch.exp(lq)
    S = (torch.exp(lq + s) - q) * torch.cos(m.T - m) - (torch.exp(lq - s) - q) * torch.cos(m.T + m)
    S = e * e.T * S / 2.0
    C = e * torch.diag(torch.exp(-ds / 2.0) * torch.cos(m[0]))
    return M, S, C.reshape(k, k)


def linear_controller(m, s, W, b, max_action=1.0, squash=True):
    M = m @ W.T + b.reshape(1, -1)
    S = W @ s @ W.T
    V = W.T
    if squash:
        M, S, V2 = squash_sin(M, S, max_action)
        V = V @ V2
    return M, S, V


def exponential_reward(m, s, W=None, tg=None):
    d = m.shape[1]
    W = torch.eye(d, dtype=DT) if W is None else t(W).reshape(d, d)
    tg = torch.zeros((1, d), dtype=DT) if tg is None else t(tg).reshape(1, d)
    eye = torch.eye(d, dtype=DT)
    SW = s @ W
    iSpW = torch.linalg.solve((eye + SW).T, W.T).T
    return torch.exp(-(m - tg) @ iSpW @ (m - tg).T / 2.0) / torch.sqrt(torch.linalg.det(eye + SW))


def propagate(gp, controller, m_x, s_x):
    m_u, s_u, c_xu = controller(m_x, s_x)
    m = torch.cat([m_x, m_u], dim=1)
    s1 = torch.cat([s_x, s_x @ c_xu], dim=1)
    s2 = torch.cat([(s_x @ c_xu).T, s_u], dim=1)
    s = torch.cat([s1, s2], dim=0)
    M_dx, S_dx, C_dx = gp(m, s)
    return M_dx + m_x, S_dx + s_x + s1 @ C_dx + C_dx.T @ s1.T


def predict(gp, controller, reward, m_x, s_x, n):
    total = torch.zeros((1, 1), dtype=DT)
    for _ in range(n):
        total = total + reward(m_x, s_x)
        m_x, s_x = propagate(gp, controller, m_x, s_x)
    return m_x, s_x, total


def rbf_controller(m, s, X, Y, ls, noise, max_action=1.0, squash=True):
    """controllers.py:108-121: deterministic GP with unit signal variance, iK := 0, S -= diag(var - 1e-6)."""
    n, d = X.shape
    U = Y.shape[1]
    var = torch.ones(U, dtype=DT)
    beta = []
    for a in range(U):
        K = torch.exp(-0.5 * torch.sum(((X[:, None, :] - X[None, :, :]) / ls[a]) ** 2, -1))
        beta.append(torch.linalg.solve(K + noise[a] * torch.eye(n, dtype=DT), Y[:, a]))
    beta = torch.stack(beta)
    M, S, V = predict_given_factorizations(X, ls, var, m, s, torch.zeros((U, n, n), dtype=DT), beta)
    S = S - torch.diag(var - 1e-6)
    if squash:
        M, S, V2 = squash_sin(M, S, max_action)
        V = V @ V2
    return M, S, V
