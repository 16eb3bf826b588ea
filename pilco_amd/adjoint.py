"""Reverse-mode gradient of the rollout reward w.r.t. the controller parameters.

The reference obtains it from TensorFlow's autodiff through the tf.while_loop
(pilco/models/pilco.py:85-90, 126-135).  Here the O(N^2) part -- the adjoint of the GP
moment-matching step -- runs on the device (pilco_gp_predict_vjp); the
O(D^3) links of the chain (propagate pilco.py:147-149, joint Gaussian pilco.py:141-144, linear
controller + squash controllers.py:13-58, exponential reward rewards.py:32-39) are differentiated
here by hand in NumPy.  Only (m_t, s_t) per step is checkpointed (the rollout tape); everything
else is recomputed.  Deterministic: the same inputs give bitwise the same gradient.
"""
from __future__ import annotations

import numpy as np

from . import _lib


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


def rollout_value_and_grad(pilco):
    """(reward, d reward / d W, d reward / d b) for a LinearController policy."""
    from .controllers import LinearController
    from .rewards import ExponentialReward
    ctl, rew = pilco.controller, pilco.reward
    if not isinstance(ctl, LinearController):
        raise TypeError("analytic policy gradient: LinearController only (the RBF policy uses finite differences)")
    if not isinstance(rew, ExponentialReward):
        raise TypeError("analytic policy gradient: ExponentialReward only")
    E, U, H = pilco.state_dim, pilco.control_dim, pilco.horizon
    D = E + U
    W, b = ctl.W.numpy(), ctl.b.numpy().reshape(-1)
    e = np.broadcast_to(np.asarray(ctl.max_action, np.float64).reshape(-1), (U,)).copy()
    Wr, tr = rew.W.numpy(), rew.t.numpy().reshape(1, -1)
    pilco.mgpr._user_factors = None
    pilco.mgpr._ensure_factorized()
    ctx = pilco.ctx
    mH, SH, R, traj, tape = ctx.rollout_tape(pilco._policy_spec(), rew.terms(), pilco.m_init, pilco.S_init, H)
    o = [0, D, D + D * D, D + D * D + E * D, D + D * D + E * D + E, D + D * D + E * D + E + E * E]
    mbar = np.zeros((1, E))
    sbar = np.zeros((E, E))
    Wbar, bbar = np.zeros_like(W), np.zeros_like(b)
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
        # controller (controllers.py:46-58): mu0 = W m + b, su0 = W s W^T, c = W^T diag(Cd)
        mu0 = (m_x @ W.T).reshape(-1) + b
        su0 = W @ s_x @ W.T
        _, _, Cd, _ = squash_fwd(mu0, su0, e)
        c = W.T * Cd[None, :]
        sxb += Bb @ c.T
        cb = s_x.T @ Bb
        Wbar += Cd[:, None] * cb.T
        Cdbar = np.einsum('ue,eu->u', W, cb)
        mu0b, su0b = squash_vjp(mu0, su0, e, mub, sub, Cdbar)
        Wbar += np.outer(mu0b, m_x[0]) + su0b @ W @ s_x.T + su0b.T @ W @ s_x
        bbar += mu0b
        mxb += (W.T @ mu0b)[None, :]
        sxb += W.T @ su0b @ W
        # reward of the pre-propagation state (pilco.py:133)
        _, rm, rS = exp_reward_grad(m_x, s_x, Wr, tr)
        mxb += rm
        sxb += rS
        mbar, sbar = mxb, 0.5 * (sxb + sxb.T)
    return float(R[0, 0]), Wbar, bbar.reshape(ctl.b.shape)
