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
        self._ctx = ctx   # None: decided on first use (the ctx property) -- constructing a model touches no device
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
        import weakref
        for comp in (self.mgpr, self.controller, getattr(self.controller, "_gp", None), self.reward):
            if comp is not None and getattr(comp, "_ctx", None) is None:   # components without a context will ask this object for its
                ref = getattr(comp, "_ctx_owner", None)
                if ref is not None and ref() is not None:
                    continue   # a component shared with another live PILCO object stays with it (and this object joins: ctx below)
                try:
                    comp._ctx_owner = weakref.ref(self)
                except AttributeError:
                    pass
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
        if self._ctx is None:
            # a component that already lives on a context decides; otherwise a context of this object's own: the default one
            # for the first live PILCO object, a pooled one for every further (_lib.context_for)
            # (only the components that hold device slots decide -- the models, the controller and its GP; a stateless
            # reward object reused by several PILCO objects must not pull them all onto one context)
            comps = [c for c in (self.mgpr, self.controller, getattr(self.controller, "_gp", None)) if c is not None]
            for comp in comps:
                if getattr(comp, "_ctx", None) is not None:
                    self._ctx = comp._ctx
                    _lib.context_adopted(self._ctx, self)   # ... and this object is now that context's holder in the pool
                    break
            if self._ctx is None:
                for comp in comps:   # a component that belongs to another live PILCO object (a shared controller): one context for both
                    ref = getattr(comp, "_ctx_owner", None)
                    other = ref() if ref is not None else None
                    if other is not None and other is not self:
                        self._ctx = other.ctx
                        break
            if self._ctx is None:
                self._ctx = _lib.context_for(self)
        return self._ctx

    def _policy_spec(self):
        if self.control_dim == 0 or self.controller is None:
            return dict(kind=_lib.POLICY_NONE, state_dim=self.state_dim, control_dim=0)
        return self.controller.policy_spec(True)

    # -- reward terms: on the device (exponential / linear) and, for anything else, on the host along the trajectory
    def _reward_terms(self):
        terms = self.reward.terms() if hasattr(self.reward, "terms") else []
        return terms or [dict(kind=_lib.REWARD_LINEAR, coef=0.0, W=np.zeros(self.state_dim))]    # the C ABI wants one term

    def _host_reward_terms(self):
        r = getattr(self, "reward", None)
        if r is None:
            return []
        if hasattr(r, "terms"):
            return r.host_terms() if hasattr(r, "host_terms") else []
        return [(1.0, r)]                       # a reward object of the caller's own: compute_reward(m, s) only

    def _host_reward_value(self, traj, n):
        """sum over the pre-propagation states t < n of the host reward terms (pilco.py:133 accumulates the same way)."""
        E, tot = self.state_dim, 0.0
        for t in range(int(n)):
            m, s = traj[t, :E].reshape(1, E), traj[t, E:].reshape(E, E)
            for c, r in self._host_reward_terms():
                tot += c * float(np.ravel(r.compute_reward(m, s)[0])[0])
        return tot

    def trajectory_objective(self, traj):
        """What predict() adds to the device reward as a function of the state trajectory -- the host reward terms -- and
        its cotangent seeds d / d (m_t, s_t) for the native reverse sweep (pilco_rollout_grad_seeded).  None if a host term
        has no compute_reward_grad (optimize_policy then differentiates training_loss by finite differences)."""
        host = self._host_reward_terms()
        E, H = self.state_dim, traj.shape[0] - 1
        seeds, value = np.zeros_like(traj), 0.0
        for c, r in host:
            if not hasattr(r, "compute_reward_grad"):
                return None
            for t in range(H):
                v, dm, ds = r.compute_reward_grad(traj[t, :E].reshape(1, E), traj[t, E:].reshape(E, E))
                value += c * float(v)
                seeds[t, :E] += c * np.ravel(dm)
                seeds[t, E:] += c * np.ravel(ds)
        return value, seeds

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
    def value_and_gradient(self, seed_fn=None):
        """(reward, grads) of the rollout reward w.r.t. the controller parameters: grads = (dW, db) for a LinearController,
        (dX, dY, dlengthscales) for an RbfController.  The reference obtains it from TensorFlow's reverse mode through the
        tf.while_loop (pilco/models/pilco.py:85-90,126-135); here the whole reverse sweep is native (pilco_rollout_grad /
        pilco_rollout_grad_rbf) and deterministic: the same inputs give bitwise the same gradient.
        seed_fn(traj (H+1, E+E*E)) -> cotangent seeds d objective / d (m_t, s_t) for an objective beyond the additive
        reward (the returned reward is the additive part; the gradients are those of additive reward + seeded objective)."""
        ctl = self.controller
        linear = isinstance(ctl, controllers.LinearController)
        if not linear and not isinstance(ctl, controllers.RbfController):
            raise TypeError("analytic policy gradient: LinearController or RbfController")
        self.mgpr._user_factors = None
        self.mgpr._ensure_factorized()
        if linear:
            r, dW, db = self.ctx.rollout_grad(self._policy_spec(), self._reward_terms(), self.m_init, self.S_init, self.horizon, seed_fn=seed_fn)
            return r, (dW.reshape(ctl.W.shape), db.reshape(ctl.b.shape))
        r, dX, dY, dl = self.ctx.rollout_grad_rbf(self._policy_spec(), self._reward_terms(), self.m_init, self.S_init, self.horizon,
                                                  ctl.X, ctl.Y, ctl.lengthscales, ctl.noise, seed_fn=seed_fn)
        return r, (dX, dY, dl)

    def compute_action(self, x_m):
        return self.controller.compute_action(x_m, np.zeros([self.state_dim, self.state_dim]))[0]

    # pilco.py:118-136
    def predict(self, m_x, s_x, n):
        self.mgpr._user_factors = None
        self.mgpr._ensure_factorized()
        if not self._host_reward_terms():
            return self.ctx.rollout(self._policy_spec(), self._reward_terms(), m_x, s_x, int(n))
        M, S, R, _ = self.predict_trajectory(m_x, s_x, n)
        return M, S, R

    def predict_trajectory(self, m_x, s_x, n):
        """Extension: also returns the (n+1, E + E*E) per-step states."""
        self.mgpr._user_factors = None
        self.mgpr._ensure_factorized()
        M, S, R, traj = self.ctx.rollout(self._policy_spec(), self._reward_terms(), m_x, s_x, int(n), want_traj=True)
        if self._host_reward_terms():
            R = R + self._host_reward_value(traj, n)
        return M, S, R, traj

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
