"""Safe-PILCO extension with the reference's interface (/root/reference/safe_pilco_extension/).

``SafePILCO.predict`` (safe_pilco.py:29-50) runs the same moment-matching rollout and, besides the
additive reward, accumulates the product of (1 - risk(m_t, s_t)) over the pre-propagation states;
``reward_total = reward_add + mu * (1 - prod)``.  The rollout runs on the device
(PILCO.predict_trajectory); the risk terms are O(H) scalar Normal CDFs evaluated on the host, exactly
as written in rewards_safe.py:20-61 (including its use of the variance entry as the Normal scale)."""
from __future__ import annotations

import numpy as np
from scipy.stats import norm

from .models.pilco import PILCO
from .params import Parameter


def _cdf_interval(mu, scale, high, low):
    """P = cdf(high) - cdf(low) of Normal(mu, scale) (a missing bound: -inf / +inf) with dP/dmu and dP/dscale."""
    P, dmu, dsc = 0.0, 0.0, 0.0
    if high is not None:
        z = (high - mu) / scale
        P += norm.cdf(z); dmu -= norm.pdf(z) / scale; dsc -= z * norm.pdf(z) / scale
    else:
        P += 1.0
    if low is not None:
        z = (low - mu) / scale
        P -= norm.cdf(z); dmu += norm.pdf(z) / scale; dsc += z * norm.pdf(z) / scale
    return P, dmu, dsc


class RiskOfCollision:
    """rewards_safe.py:13-25."""

    def __init__(self, state_dim, low, high):
        self.state_dim = state_dim
        self.low = np.asarray(low, np.float64)
        self.high = np.asarray(high, np.float64)

    def compute_reward(self, m, s):
        infl = 2 * np.diag(s)
        d1 = norm(loc=m[0, 0], scale=infl[0])
        d2 = norm(loc=m[0, 2], scale=infl[2])
        risk = (d1.cdf(self.high[0]) - d1.cdf(self.low[0])) * (d2.cdf(self.high[1]) - d2.cdf(self.low[1]))
        return risk, 0.0001 * np.ones(1)

    def compute_reward_grad(self, m, s):
        """(risk, d risk / d m (E), d risk / d s (E, E)) of compute_reward (the reference differentiates it by TF autodiff)."""
        E = m.shape[1]
        fac, dmu, dsc = [], [], []
        for k, (dim, hi, lo) in enumerate(((0, self.high[0], self.low[0]), (2, self.high[1], self.low[1]))):
            f, a, b = _cdf_interval(m[0, dim], 2.0 * s[dim, dim], hi, lo)
            fac.append(f); dmu.append(a); dsc.append(2.0 * b)          # scale = 2 s_dd
        dm, ds = np.zeros(E), np.zeros((E, E))
        dm[0], dm[2] = dmu[0] * fac[1], dmu[1] * fac[0]
        ds[0, 0], ds[2, 2] = dsc[0] * fac[1], dsc[1] * fac[0]
        return fac[0] * fac[1], dm, ds


class SingleConstraint:
    """rewards_safe.py:27-61."""

    def __init__(self, dim, high=None, low=None, inside=True):
        if high is None and low is None:
            raise Exception("At least one of bounds (high,low) has to be defined")
        self.high, self.low, self.dim, self.inside = high, low, dim, bool(inside)

    def compute_reward(self, m, s):
        dist = norm(loc=m[0, self.dim], scale=s[self.dim, self.dim])
        if self.high is None:
            risk = 1 - dist.cdf(self.low)
        elif self.low is None:
            risk = dist.cdf(self.high)
        else:
            risk = dist.cdf(self.high) - dist.cdf(self.low)
        if not self.inside:
            risk = 1 - risk
        return risk, 0.0001 * np.ones(1)

    def compute_reward_grad(self, m, s):
        """(risk, d risk / d m (E), d risk / d s (E, E)) of compute_reward."""
        E = m.shape[1]
        f, a, b = _cdf_interval(m[0, self.dim], s[self.dim, self.dim], self.high, self.low)   # low None: cdf(high); high None: 1 - cdf(low)
        sign = 1.0 if self.inside else -1.0
        dm, ds = np.zeros(E), np.zeros((E, E))
        dm[self.dim], ds[self.dim, self.dim] = sign * a, sign * b
        return (f if self.inside else 1.0 - f), dm, ds


class ObjectiveFunction:
    """rewards_safe.py:63-73 (the reference version references an un-imported Parameter)."""

    def __init__(self, reward_f, risk_f, mu=1.0):
        self.reward_f, self.risk_f, self.mu = reward_f, risk_f, float(mu)

    def compute_reward(self, m, s):
        reward, var = self.reward_f.compute_reward(m, s)
        risk, _ = self.risk_f.compute_reward(m, s)
        return reward - self.mu * risk, var

    # as the reward of a PILCO: the reward part on the device where it can be, the risk part (and whatever else the device does
    # not evaluate) on the host along the trajectory, differentiated through cotangent seeds (models/pilco.py)
    def terms(self):
        return self.reward_f.terms() if hasattr(self.reward_f, "terms") else []

    def host_terms(self):
        base = self.reward_f.host_terms() if hasattr(self.reward_f, "terms") else [(1.0, self.reward_f)]
        return list(base) + [(-self.mu, self.risk_f)]


class SafePILCO(PILCO):
    def __init__(self, data, num_induced_points=None, horizon=30, controller=None, reward_add=None,
                 reward_mult=None, m_init=None, S_init=None, name=None, mu=5.0, ctx=None):
        super().__init__(data, num_induced_points=num_induced_points, horizon=horizon, controller=controller,
                         reward=reward_add, m_init=m_init, S_init=S_init, name=name, ctx=ctx)
        if reward_mult is None:
            raise Exception("have to define multiplicative reward")
        self.mu = Parameter(float(mu), trainable=False, name="mu")   # safe_pilco.py:25: gpflow.Parameter(mu, trainable=False): callers use .numpy() / .assign()
        self.reward_mult = reward_mult

    def predict(self, m_x, s_x, n):
        E = self.state_dim
        M, S, reward_add, traj = self.predict_trajectory(m_x, s_x, n)
        mult = 1.0
        for t in range(n):  # pre-propagation states, like the additive reward
            mult *= 1.0 - float(self.reward_mult.compute_reward(traj[t, :E].reshape(1, E), traj[t, E:].reshape(E, E))[0])
        return M, S, reward_add + self._mu() * (1.0 - mult)

    def _mu(self):
        return float(self.mu.numpy()) if isinstance(self.mu, Parameter) else float(self.mu)

    def trajectory_objective(self, traj):
        """The part of predict()'s total reward that is not the additive reward, mu (1 - prod_t (1 - risk_t)), and its
        cotangent seeds d / d (m_t, s_t), t = 0..H (traj: (H+1, E + E*E), state t < H pre-propagation).  optimize_policy
        hands the seeds to the native reverse sweep (pilco_rollout_grad_seeded): the reference gets the same gradient from
        TensorFlow's reverse mode through its predict() (safe_pilco.py:29-50, pilco.py:85-90)."""
        if not hasattr(self.reward_mult, "compute_reward_grad"):
            return None
        E, H = self.state_dim, traj.shape[0] - 1
        risks, grads = np.empty(H), []
        for t in range(H):
            r, dm, ds = self.reward_mult.compute_reward_grad(traj[t, :E].reshape(1, E), traj[t, E:].reshape(E, E))
            risks[t] = r
            grads.append((dm, ds))
        one, mu = 1.0 - risks, self._mu()
        seeds = np.zeros_like(traj)
        for t in range(H):
            w = mu * np.prod(np.delete(one, t))               # d [mu (1 - prod)] / d risk_t
            seeds[t, :E] = w * grads[t][0]
            seeds[t, E:] = (w * grads[t][1]).ravel()
        base = PILCO.trajectory_objective(self, traj)         # host-evaluated terms of the additive reward, if any
        if base is None:
            return None
        return base[0] + mu * (1.0 - np.prod(one)), base[1] + seeds
