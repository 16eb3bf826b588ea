"""Reverse-mode gradient of the rollout reward w.r.t. the controller parameters.

The reference obtains it from TensorFlow's autodiff through the tf.while_loop (pilco/models/pilco.py:85-90, 126-135).
Here the whole reverse sweep is native (csrc/grad.hip: pilco_rollout_grad / pilco_rollout_grad_rbf): the adjoint of the
GP moment-matching step runs on the device, the O(D^3) links of the chain in the library's host
code.  Only (m_t, s_t) and the joint Gaussian per step are checkpointed (the rollout tape).  Deterministic: the same
inputs give bitwise the same gradient.  (The NumPy prototype of the sweep lives with the test oracles:
oracle/adjoint_sweep.py.)
"""
from __future__ import annotations


def rollout_value_and_grad(pilco, seed_fn=None):
    """(reward, grads): grads = (dW, db) for a LinearController, (dX, dY, dlengthscales) for an RbfController.
    seed_fn(traj (H+1, E+E*E)) -> cotangent seeds d objective / d (m_t, s_t) for an objective beyond the additive reward
    (the returned reward is the additive part; the gradients are those of additive reward + seeded objective)."""
    from .controllers import LinearController, RbfController
    ctl, rew = pilco.controller, pilco.reward
    linear = isinstance(ctl, LinearController)
    if not linear and not isinstance(ctl, RbfController):
        raise TypeError("analytic policy gradient: LinearController or RbfController")
    pilco.mgpr._user_factors = None
    pilco.mgpr._ensure_factorized()
    if linear:
        r, dW, db = pilco.ctx.rollout_grad(pilco._policy_spec(), pilco._reward_terms(), pilco.m_init, pilco.S_init, pilco.horizon, seed_fn=seed_fn)
        return r, (dW.reshape(ctl.W.shape), db.reshape(ctl.b.shape))
    r, dX, dY, dl = pilco.ctx.rollout_grad_rbf(pilco._policy_spec(), pilco._reward_terms(), pilco.m_init, pilco.S_init, pilco.horizon,
                                               ctl.X, ctl.Y, ctl.lengthscales, ctl.noise, seed_fn=seed_fn)
    return r, (dX, dY, dl)
