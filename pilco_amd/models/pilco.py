"""PILCO rollout driver with the reference's interface
(/root/reference/pilco/models/pilco.py:15-160).  ``predict`` runs the whole
H-step moment-matching rollout on the device in one call (pilco_rollout)."""
from __future__ import annotations

import time

import numpy as np

from .. import _lib, controllers, rewards
from ..params import parameters_of, set_trainable
from .mgpr import MGPR
from .smgpr import SMGPR


class PILCO:
    def __init__(self, data, num_induced_points=None, horizon=30, controller=None,
                 reward=None, m_init=None, S_init=None, name=None, ctx=None):
        self.name = name
        self._ctx = ctx
        if num_induced_points is None:
            self.mgpr = MGPR(data, ctx=ctx)
        else:
            self.mgpr = SMGPR(data, num_induced_points, ctx=ctx)
        self.state_dim = data[1].shape[1]
        self.control_dim = data[0].shape[1] - data[1].shape[1]
        self.horizon = horizon
        if controller is None:
            self.controller = (controllers.LinearController(self.state_dim, self.control_dim, ctx=ctx)
                               if self.control_dim > 0 else None)
        else:
            self.controller = controller
        self.reward = rewards.ExponentialReward(self.state_dim) if reward is None else reward
        if m_init is None or S_init is None:
            # pilco.py:37-41: first state of the data set, 0.1 * I
            self.m_init = np.asarray(data[0])[0:1, 0:self.state_dim]
            self.S_init = np.diag(np.ones(self.state_dim) * 0.1)
        else:
            self.m_init = m_init
            self.S_init = S_init
        self.optimizer = None

    @property
    def ctx(self):
        return self.mgpr.ctx

    def _policy_spec(self):
        if self.control_dim == 0 or self.controller is None:
            return dict(kind=_lib.POLICY_NONE, state_dim=self.state_dim, control_dim=0)
        return self.controller.policy_spec(True)

    # pilco.py:47-50
    def training_loss(self):
        return -self.predict(self.m_init, self.S_init, self.horizon)[2]

    # pilco.py:52-73 (the pandas pretty-printing is cosmetic and omitted)
    def optimize_models(self, maxiter=200, restarts=1, verbose=True):
        self.mgpr.optimize(restarts=restarts)
        if verbose:
            print('-----Learned models------')
            for i, model in enumerate(self.mgpr.models):
                print('GP%d lengthscales %s variance %.3g noise %.3g' % (
                    i, np.array2string(np.asarray(model.kernel.lengthscales.numpy()), precision=3),
                    float(model.kernel.variance.numpy()), float(model.likelihood.variance.numpy())))

    # pilco.py:75-113
    def optimize_policy(self, maxiter=50, restarts=1, verbose=True):
        from ..training import optimize_policy
        return optimize_policy(self, maxiter=maxiter, restarts=restarts, verbose=verbose)

    # pilco.py:115-116
    def compute_action(self, x_m):
        return self.controller.compute_action(x_m, np.zeros([self.state_dim, self.state_dim]))[0]

    # pilco.py:118-136
    def predict(self, m_x, s_x, n):
        self.mgpr._user_factors = None
        self.mgpr._ensure_factorized()
        return self.ctx.rollout(self._policy_spec(), self.reward.terms(), m_x, s_x, int(n))

    def predict_trajectory(self, m_x, s_x, n):
        """Extension: also returns the (n+1, E + E*E) per-step states."""
        self.mgpr._user_factors = None
        self.mgpr._ensure_factorized()
        return self.ctx.rollout(self._policy_spec(), self.reward.terms(), m_x, s_x, int(n), want_traj=True)

    # pilco.py:138-153
    def propagate(self, m_x, s_x):
        self.mgpr._user_factors = None
        self.mgpr._ensure_factorized()
        return self.ctx.propagate(self._policy_spec(), m_x, s_x)

    # pilco.py:155-156
    def compute_reward(self):
        return -self.training_loss()

    @property
    def maximum_log_likelihood_objective(self):
        return -self.training_loss()

    @property
    def trainable_parameters(self):
        ps = list(self.mgpr.trainable_parameters)
        if self.controller is not None:
            own = getattr(type(self.controller), "trainable_parameters", None)    # an RbfController lists centres, targets, lengthscales
            ps += list(self.controller.trainable_parameters) if own is not None else [p for p in parameters_of(self.controller) if p.trainable]
        return ps
