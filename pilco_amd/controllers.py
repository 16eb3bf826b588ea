"""Controllers with the reference's interface (/root/reference/pilco/controllers.py):
``compute_action(m, s, squash=True) -> (M (1,k), S (k,k), V (d,k))`` and
``randomize()``.  Arithmetic runs on the device (k_glue in csrc/moment.hip)."""
from __future__ import annotations

import numpy as np

from . import _lib
from .params import Parameter


def squash_sin(m, s, max_action=None, ctx=None):
    """Moments of e*sin(x), x ~ N(m, s)  (controllers.py:13-36).

    Evaluated on the device as an identity linear map followed by the squashing
    stage of the controller kernel."""
    m = np.asarray(m, np.float64).reshape(1, -1)
    k = m.shape[1]
    ctx = ctx or _lib.get_context()
    spec = dict(kind=_lib.POLICY_LINEAR, state_dim=k, control_dim=k, W=np.eye(k), b=np.zeros(k),
                max_action=np.ones(k) if max_action is None else max_action, squash=True)
    return ctx.policy_action(spec, m, s)


class LinearController:
    """controllers.py:39-63."""

    def __init__(self, state_dim, control_dim, max_action=1.0, ctx=None):
        self.state_dim = state_dim
        self.control_dim = control_dim
        self.W = Parameter(np.random.rand(control_dim, state_dim), name="W")
        self.b = Parameter(np.random.rand(1, control_dim), name="b")
        self.max_action = max_action
        self._ctx = ctx

    @property
    def ctx(self):
        if self._ctx is None:
            self._ctx = _lib.get_context()
        return self._ctx

    def policy_spec(self, squash=True):
        return dict(kind=_lib.POLICY_LINEAR, state_dim=self.state_dim, control_dim=self.control_dim,
                    W=self.W.numpy(), b=self.b.numpy().reshape(-1), max_action=self.max_action, squash=squash)

    def compute_action(self, m, s, squash=True):
        return self.ctx.policy_action(self.policy_spec(squash), m, s)

    def randomize(self):
        mean, sigma = 0, 1
        self.W.assign(mean + sigma * np.random.normal(size=self.W.shape))
        self.b.assign(mean + sigma * np.random.normal(size=self.b.shape))

    @property
    def trainable_parameters(self):
        return [p for p in (self.W, self.b) if p.trainable]
