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


class ObjectiveFunction:
    """rewards_safe.py:63-73 (the reference version references an un-imported Parameter)."""

    def __init__(self, reward_f, risk_f, mu=1.0):
        self.reward_f, self.risk_f, self.mu = reward_f, risk_f, float(mu)

    def compute_reward(self, m, s):
        reward, var = self.reward_f.compute_reward(m, s)
        risk, _ = self.risk_f.compute_reward(m, s)
        return reward - self.mu * risk, var


class SafePILCO(PILCO):
    def __init__(self, data, num_induced_points=None, horizon=30, controller=None, reward_add=None,
                 reward_mult=None, m_init=None, S_init=None, name=None, mu=5.0, ctx=None):
        super().__init__(data, num_induced_points=num_induced_points, horizon=horizon, controller=controller,
                         reward=reward_add, m_init=m_init, S_init=S_init, name=name, ctx=ctx)
        if reward_mult is None:
            raise Exception("have to define multiplicative reward")
        self.mu = float(mu)
        self.reward_mult = reward_mult

    def predict(self, m_x, s_x, n):
        E = self.state_dim
        M, S, reward_add, traj = self.predict_trajectory(m_x, s_x, n)
        mult = 1.0
        for t in range(n):  # pre-propagation states, like the additive reward
            mult *= 1.0 - float(self.reward_mult.compute_reward(traj[t, :E].reshape(1, E), traj[t, E:].reshape(E, E))[0])
        return M, S, reward_add + self.mu * (1.0 - mult)
