"""Rewards with the reference's interface (/root/reference/pilco/rewards.py):
``compute_reward(m (1,k), s (k,k)) -> (muR (1,1), sR (1,1))``."""
from __future__ import annotations

import numpy as np

from . import _lib
from .params import Parameter


class _Reward:
    _ctx = None

    @property
    def ctx(self):
        if self._ctx is None:
            self._ctx = _lib.resolve_ctx(self)
        return self._ctx

    def terms(self):
        raise NotImplementedError

    def host_terms(self):
        """(coefficient, reward object) pairs that are evaluated on the HOST along the trajectory (models/pilco.py): base
        rewards that are not device reward terms.  None for the built-in rewards."""
        return []

    def compute_reward(self, m, s):
        return self.ctx.reward_eval(self.terms(), self.state_dim, m, s)


class ExponentialReward(_Reward):
    """rewards.py:7-51."""

    def __init__(self, state_dim, W=None, t=None):
        self.state_dim = state_dim
        W = np.eye(state_dim) if W is None else np.reshape(W, (state_dim, state_dim))
        t = np.zeros((1, state_dim)) if t is None else np.reshape(t, (1, state_dim))
        self.W = Parameter(W, trainable=False)
        self.t = Parameter(t, trainable=False)

    def terms(self):
        return [dict(kind=_lib.REWARD_EXPONENTIAL, coef=1.0, W=self.W.numpy(), t=self.t.numpy().reshape(-1))]


class LinearReward(_Reward):
    """rewards.py:53-61."""

    def __init__(self, state_dim, W):
        self.state_dim = state_dim
        self.W = Parameter(np.reshape(W, (state_dim, 1)), trainable=False)

    def terms(self):
        return [dict(kind=_lib.REWARD_LINEAR, coef=1.0, W=self.W.numpy().reshape(-1))]


class CombinedRewards(_Reward):
    """rewards.py:64-81: weighted sum of base rewards (variance weights coef^2)."""

    def __init__(self, state_dim, rewards=[], coefs=None):
        self.state_dim = state_dim
        self.base_rewards = rewards
        self.coefs = Parameter(np.ones(len(rewards)) if coefs is None else coefs, trainable=False)

    def terms(self):
        """The device reward terms: base rewards of this package (exponential / linear, nested combinations)."""
        out = []
        for r, c in zip(self.base_rewards, np.asarray(self.coefs.numpy()).reshape(-1)):
            if not hasattr(r, "terms"):
                continue
            for t in r.terms():
                t = dict(t)
                t["coef"] = float(c) * t.get("coef", 1.0)
                out.append(t)
        return out

    def host_terms(self):
        """Base rewards that only offer compute_reward(m, s) -- e.g. the Safe-PILCO constraints the reference's
        examples/safe_swimmer_run.py:59-64 puts into a CombinedRewards: evaluated on the host on the rollout's states."""
        out = []
        for r, c in zip(self.base_rewards, np.asarray(self.coefs.numpy()).reshape(-1)):
            if hasattr(r, "terms"):
                out += [(float(c) * ck, rk) for ck, rk in r.host_terms()]
            else:
                out.append((float(c), r))
        return out

    def compute_reward(self, m, s):
        terms, host = self.terms(), self.host_terms()
        mu, var = self.ctx.reward_eval(terms, self.state_dim, m, s) if terms else (np.zeros((1, 1)), np.zeros((1, 1)))
        for c, r in host:                      # rewards.py:73-81: mean sum c_k mu_k, variance sum c_k^2 var_k
            mk, vk = r.compute_reward(np.asarray(m, np.float64).reshape(1, -1), np.asarray(s, np.float64))
            mu = mu + c * float(np.ravel(mk)[0])
            var = var + c * c * float(np.ravel(vk)[0])
        return mu, var
