"""NumPy prototype of the reverse sweep of the policy gradient (TEST INFRASTRUCTURE ONLY, see oracle/__init__.py).

The product's gradient is the native sweep in csrc/grad.hip (pilco_rollout_grad / pilco_rollout_grad_rbf).  This module
holds the derivation it was written from: the O(D^3) links of the chain (propagate pilco.py:147-149, joint Gaussian
pilco.py:141-144, linear / RBF controller + squash controllers.py:13-58,108-121, rewards rewards.py:19-81) differentiated
by hand in NumPy, driving the device VJP of the moment-matching step (pilco_gp_predict_vjp) step by step.  Tests compare
the native sweeps with it (tests/test_gpu_parity.py) and its host-only pieces with torch autograd (tests/test_oracle.py).
"""
from __future__ import annotations

import numpy as np

from pilco_amd import _lib


def squash_fwd(mu0, su0, e):
    ds = np.diag(su0)
    ex = np.exp(-ds / 2.0)
    M = e * ex * np.sin(mu0)
    Cd = e * ex * np.cos(mu0)
    lq = -(ds[:, None] + ds[None, :]) / 2.0
    q = np.exp(lq)
    Ep, Em = np.exp(lq + su0), np.exp(lq - su0)
    dm, sm = mu0[:, None] - mu0[None, :], mu0[:, None] + mu0[None, :]
    ee = np.outer(e, e)
    S = ee / 2.0 * ((Ep - q) * np.cos(dm) - (Em - q) * np.cos(sm))
    return M, S, Cd, (q, Ep, Em, dm, sm, ee)


def squash_vjp(mu0, su0, e, Mbar, Sbar, Cdbar):
    """VJP of squash_sin (controllers.py:13-36; derivatives as in gSin.m:50-74)."""
    M, S, Cd, (q, Ep, Em, dm, sm, ee) = squash_fwd(mu0, su0, e)
    D1 = ee / 2.0 * (-(Ep - q) * np.sin(dm) + (Em - q) * np.sin(sm))   # dS_uv / dmu_u
    D2 = ee / 2.0 * ((Ep - q) * np.sin(dm) + (Em - q) * np.sin(sm))    # dS_uv / dmu_v
    mubar = (Sbar * D1).sum(1) + (Sbar * D2).sum(0) + Mbar * Cd - Cdbar * M
    G = ee / 2.0 * (Ep * np.cos(dm) + Em * np.cos(sm))                  # direct dS_uv / dsu0_uv
    subar = Sbar * G
    dd = -0.5 * ((Sbar * S).sum(1) + (Sbar * S).sum(0)) - 0.5 * Mbar * M - 0.5 * Cdbar * Cd
    subar[np.diag_indices_from(subar)] += dd
    return mubar, subar


def exp_reward_grad(m, S, W, t):
    """d muR / d m, d muR / d S of rewards.py:32-39 (formulas of reward.m:47-50), symmetric W."""
    d = (m - t).reshape(-1, 1)
    k = d.shape[0]
    iSpW = np.linalg.solve((np.eye(k) + S @ W).T, W.T).T
    muR = float(np.exp(-0.5 * (d.T @ iSpW @ d)[0, 0]) / np.sqrt(np.linalg.det(np.eye(k) + S @ W)))
    dm = -muR * (d.T @ iSpW)
    dS = muR * (iSpW @ d @ d.T - np.eye(k)) @ iSpW / 2.0
    return muR, dm.reshape(1, -1), 0.5 * (dS + dS.T)


def rbf_policy_fwd(m, s, X, Y, ls, noise):
    """Moment matching through the RBF policy = deterministic GP (controllers.py:108-117 calling
    mgpr.py:91-149 with iK = 0, unit signal variance): returns (M (U,), S (U,U), V (d,U), cache)."""
    m = np.asarray(m, np.float64).reshape(-1)
    n, d = X.shape
    U = ls.shape[0]
    zeta = X - m[None, :]
    beta = np.empty((U, n))
    Ainv = []
    diff2 = (X[:, None, :] - X[None, :, :]) ** 2
    Ks = []
    for a in range(U):
        K = np.exp(-0.5 * (diff2 / ls[a] ** 2).sum(-1))
        Ai = np.linalg.inv(K + noise[a] * np.eye(n))        # FakeGPR likelihood variance, controllers.py:67-77
        Ainv.append(Ai)
        Ks.append(K)
        beta[a] = Ai @ Y[:, a]
    M = np.empty(U)
    V = np.empty((d, U))
    mean = []
    for a in range(U):
        T = np.linalg.inv(s + np.diag(ls[a] ** 2))
        G = zeta @ T
        h = -0.5 * (zeta * G).sum(1)
        logc = -0.5 * (np.linalg.slogdet(s + np.diag(ls[a] ** 2))[1] - 2.0 * np.log(ls[a]).sum())
        ex = np.exp(h + logc)
        q = beta[a] * ex
        M[a] = q.sum()
        V[:, a] = G.T @ q
        mean.append((T, G, ex, q))
    S = np.empty((U, U))
    pair = {}
    for a in range(U):
        ia = 1.0 / ls[a] ** 2
        z = zeta * ia
        ka = -0.5 * (zeta * z).sum(1)
        for b in range(U):
            ib = 1.0 / ls[b] ** 2
            w = zeta * ib
            kb = -0.5 * (zeta * w).sum(1)
            R = s * (ia + ib)[None, :] + np.eye(d)
            Ri = np.linalg.inv(R)
            Q = 0.5 * Ri @ s
            zQ, wQ = z @ Q, w @ Q
            L = np.exp((ka + (zQ * z).sum(1))[:, None] + (kb + (wQ * w).sum(1))[None, :] + 2.0 * zQ @ w.T)
            r = 1.0 / np.sqrt(np.linalg.det(R))
            val = beta[a] @ L @ beta[b]
            S[a, b] = val * r - M[a] * M[b] + (1e-6 if a == b else 0.0)   # + var - (var - 1e-6), controllers.py:117
            pair[a, b] = (z, w, Ri, Q, zQ, wQ, L, r, val)
    cache = dict(m=m, s=s, X=X, ls=ls, zeta=zeta, beta=beta, Ainv=Ainv, Ks=Ks, mean=mean, pair=pair, M=M, diff2=diff2)
    return M, S, V, cache


def rbf_policy_vjp(cache, Mbar, Sbar, Vbar):
    """Cotangents of (m, s, X, Y, ls) given those of (M, S, V) of rbf_policy_fwd; line-by-line
    reverse of the forward pass, including beta = (K + noise I)^-1 Y."""
    s, X, ls, zeta, beta, M = cache["s"], cache["X"], cache["ls"], cache["zeta"], cache["beta"], cache["M"]
    n, d = X.shape
    U = ls.shape[0]
    Mbar = np.array(Mbar, np.float64).reshape(U).copy()
    Sbar = np.asarray(Sbar, np.float64).reshape(U, U)
    Vbar = np.asarray(Vbar, np.float64).reshape(d, U)
    zb = np.zeros((n, d))          # cotangent of zeta
    sb = np.zeros((d, d))
    lsb = np.zeros((U, d))
    ib_ = np.zeros((U, d))         # cotangent of 1 / ls^2
    bb = np.zeros((U, n))          # cotangent of beta
    for a in range(U):
        for b in range(U):
            g = Sbar[a, b]
            if g == 0.0:
                continue
            z, w, Ri, Q, zQ, wQ, L, r, val = cache["pair"][a, b]
            Mbar[a] -= g * M[b]
            Mbar[b] -= g * M[a]
            valb = g * r
            ldb = -0.5 * g * val * r                       # cotangent of log det R
            Lb_ = L @ beta[b]
            bb[a] += valb * Lb_
            bb[b] += valb * (beta[a] @ L)
            Eb = valb * (beta[a][:, None] * beta[b][None, :]) * L
            ub, vb = Eb.sum(1), Eb.sum(0)
            zQb = 2.0 * Eb @ w + ub[:, None] * z
            wb = 2.0 * Eb.T @ zQ + vb[:, None] * wQ
            zb_ = ub[:, None] * zQ
            wQb = vb[:, None] * w
            zb_ += zQb @ Q.T
            wb += wQb @ Q.T
            Qb = z.T @ zQb + w.T @ wQb
            zb -= 0.5 * ub[:, None] * z + 0.5 * vb[:, None] * w      # k_a, k_b
            zb_ -= 0.5 * ub[:, None] * zeta
            wb -= 0.5 * vb[:, None] * zeta
            ia, ib = 1.0 / ls[a] ** 2, 1.0 / ls[b] ** 2
            zb += zb_ * ia + wb * ib
            ib_[a] += (zb_ * zeta).sum(0)
            ib_[b] += (wb * zeta).sum(0)
            Rb = -Ri.T @ Qb @ Q.T + ldb * Ri.T
            sb += 0.5 * Ri.T @ Qb + Rb * (ia + ib)[None, :]
            lam = (s * Rb).sum(0)
            ib_[a] += lam
            ib_[b] += lam
    for a in range(U):
        T, G, ex, q = cache["mean"][a]
        qb = Mbar[a] + G @ Vbar[:, a]
        Gb = np.outer(q, Vbar[:, a])
        hb = qb * q
        logcb = hb.sum()
        bb[a] += qb * ex
        zb -= 0.5 * hb[:, None] * G
        Gb -= 0.5 * hb[:, None] * zeta
        zb += Gb @ T.T
        Tb = zeta.T @ Gb
        Ab = -0.5 * logcb * T - T.T @ Tb @ T.T
        lsb[a] += logcb / ls[a] + 2.0 * ls[a] * np.diag(Ab)
        sb += Ab
    lsb += ib_ * (-2.0 / ls ** 3)
    Xb = zb.copy()
    mb = -zb.sum(0)
    Yb = np.zeros((n, U))
    for a in range(U):
        Ai, K = cache["Ainv"][a], cache["Ks"][a]
        g = Ai @ bb[a]                                     # Ai symmetric
        Yb[:, a] = g
        Wk = -np.outer(g, beta[a]) * K
        Wk = Wk + Wk.T
        Xb -= (Wk.sum(1)[:, None] * X - Wk @ X) / ls[a] ** 2
        lsb[a] += 0.5 * np.einsum('ij,ijd->d', Wk, cache["diff2"]) / ls[a] ** 3
    return mb.reshape(1, d), sb, Xb, Yb, lsb


def reward_grad(terms, m, S):
    """d/dm, d/dS of the mean reward for a list of reward terms (rewards.py:19-81)."""
    k = m.shape[1]
    dm, dS = np.zeros((1, k)), np.zeros((k, k))
    for tm in terms:
        c = float(tm.get("coef", 1.0))
        if tm["kind"] == _lib.REWARD_EXPONENTIAL:
            _, rm, rS = exp_reward_grad(m, S, np.asarray(tm["W"]).reshape(k, k), np.asarray(tm["t"]).reshape(1, k))
            dm += c * rm
            dS += c * rS
        elif tm["kind"] == _lib.REWARD_LINEAR:               # muR = m W, rewards.py:58-60
            dm += c * np.asarray(tm["W"]).reshape(1, k)
        else:
            raise TypeError("analytic policy gradient: unknown reward term")
    return dm, dS


def rollout_value_and_grad_py(pilco):
    """(reward, grads) like pilco_amd.adjoint.rollout_value_and_grad, but the reverse sweep is driven from here in
    NumPy: device VJP of the moment-matching step (pilco_gp_predict_vjp) + the hand-derived links below.  The derivation
    prototype of csrc/grad.hip; the native sweeps are checked against it on the GPU and it against autograd."""
    from pilco_amd.controllers import LinearController, RbfController
    ctl, rew = pilco.controller, pilco.reward
    linear = isinstance(ctl, LinearController)
    if not linear and not isinstance(ctl, RbfController):
        raise TypeError("analytic policy gradient: LinearController or RbfController")
    E, U, H = pilco.state_dim, pilco.control_dim, pilco.horizon
    D = E + U
    e = np.broadcast_to(np.asarray(ctl.max_action, np.float64).reshape(-1), (U,)).copy()
    if linear:
        W, b = ctl.W.numpy(), ctl.b.numpy().reshape(-1)
        Wbar, bbar = np.zeros_like(W), np.zeros_like(b)
    else:
        Xp, Yp, lsp, nzp = ctl.X.copy(), ctl.Y.copy(), ctl.lengthscales.copy(), ctl.noise.copy()
        Xbar, Ybar, lsbar = np.zeros_like(Xp), np.zeros_like(Yp), np.zeros_like(lsp)
    terms = rew.terms()
    pilco.mgpr._user_factors = None
    pilco.mgpr._ensure_factorized()
    ctx = pilco.ctx
    mH, SH, R, traj, tape = ctx.rollout_tape(pilco._policy_spec(), terms, pilco.m_init, pilco.S_init, H)
    o = [0, D, D + D * D, D + D * D + E * D, D + D * D + E * D + E, D + D * D + E * D + E + E * E]
    mbar = np.zeros((1, E))
    sbar = np.zeros((E, E))
    for t in range(H - 1, -1, -1):
        m_x = traj[t, :E].reshape(1, E)
        s_x = traj[t, E:].reshape(E, E)
        rec = tape[t]
        m_j, s_j = rec[o[0]:o[1]].reshape(1, D), rec[o[1]:o[2]].reshape(D, D)
        s1, V = rec[o[2]:o[3]].reshape(E, D), rec[o[5]:].reshape(D, E)
        # propagate (pilco.py:147-149): M_x = M + m_x, S_x = S + s_x + s1 V + (s1 V)^T
        G = sbar + sbar.T
        Mb, Sb, Vb = mbar, sbar, s1.T @ G
        s1bar = G @ V.T
        mxb, sxb = mbar.copy(), sbar.copy()
        mjb, sjb = ctx.gp_predict_vjp(_lib.SLOT_DYNAMICS, m_j, s_j, Mb, Sb, Vb, D, E)
        # joint Gaussian (pilco.py:141-144)
        mxb += mjb[:, :E]
        mub = mjb[0, E:]
        sxb += sjb[:E, :E] + s1bar[:, :E]
        Bb = sjb[:E, E:] + sjb[E:, :E].T + s1bar[:, E:]
        sub = sjb[E:, E:]
        # controller: (mu0, su0, V0) -> squash_sin -> (m_u, s_u, c = V0 diag(Cd)), controllers.py:46-58,108-121
        if linear:
            mu0 = (m_x @ W.T).reshape(-1) + b
            su0 = W @ s_x @ W.T
            V0 = W.T
        else:
            mu0, su0, V0, cache = rbf_policy_fwd(m_x, s_x, Xp, Yp, lsp, nzp)
        _, _, Cd, _ = squash_fwd(mu0, su0, e)
        c = V0 * Cd[None, :]
        sxb += Bb @ c.T
        cb = s_x.T @ Bb
        V0b = cb * Cd[None, :]
        Cdbar = np.einsum('eu,eu->u', V0, cb)
        mu0b, su0b = squash_vjp(mu0, su0, e, mub, sub, Cdbar)
        if linear:
            Wbar += V0b.T + np.outer(mu0b, m_x[0]) + su0b @ W @ s_x.T + su0b.T @ W @ s_x
            bbar += mu0b
            mxb += (W.T @ mu0b)[None, :]
            sxb += W.T @ su0b @ W
        else:
            pm, ps, pX, pY, pl = rbf_policy_vjp(cache, mu0b, su0b, V0b)
            mxb += pm
            sxb += ps
            Xbar += pX
            Ybar += pY
            lsbar += pl
        # reward of the pre-propagation state (pilco.py:133)
        rm, rS = reward_grad(terms, m_x, s_x)
        mxb += rm
        sxb += rS
        mbar, sbar = mxb, 0.5 * (sxb + sxb.T)
    if linear:
        return float(R[0, 0]), (Wbar, bbar.reshape(ctl.b.shape))
    return float(R[0, 0]), (Xbar, Ybar, lsbar)
