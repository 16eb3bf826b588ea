"""Controllers with the reference's interface (/root/reference/pilco/controllers.py):
``compute_action(m, s, squash=True) -> (M (1,k), S (k,k), V (d,k))`` and
``randomize()``.  Arithmetic runs on the device (k_glue in csrc/glue.hip)."""
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
            self._ctx = _lib.resolve_ctx(self)
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


class _PolicyData(Parameter):
    """The centres X / targets Y of an RbfController as gpflow-like Parameters (controllers.py:70-73: both are
    trainable ``Parameter``s of the reference's FakeGPR); values live in the policy GP, ``assign`` goes through set_data."""

    def __init__(self, gp, which, name):
        self._gp = None                      # (the base constructor's `self._v = ...` must not re-seat the GP's data)
        self._which = which
        Parameter.__init__(self, gp._X if which == 0 else gp._Y, name=name)
        self._gp = gp

    @property
    def _v(self):
        return self._gp._X if self._which == 0 else self._gp._Y

    @_v.setter
    def _v(self, value):
        if getattr(self, "_gp", None) is None:
            return
        value = np.array(value, dtype=np.float64)
        self._gp.set_data((value, self._gp._Y) if self._which == 0 else (self._gp._X, value))


class RbfController:
    """RBF network controller = deterministic GP (controllers.py:80-129; Deisenroth et al. 2015, sec. 5.3.2).

    In the reference this is an MGPR subclass whose ``compute_action`` calls
    ``predict_given_factorizations(m, s, 0.0 * iK, beta)`` and subtracts ``diag(variance - 1e-6)``.
    Here the policy GP lives in device slot 1 (PILCO_SLOT_POLICY) and is evaluated by the same
    moment-matching kernels with the iK stream switched off."""

    def __init__(self, state_dim, control_dim, num_basis_functions, max_action=1.0, ctx=None):
        from .models.mgpr import MGPR

        class _PolicyGP(MGPR):
            _slot = _lib.SLOT_POLICY

        self.state_dim = state_dim
        self.control_dim = control_dim
        self.num_basis_functions = num_basis_functions
        self.max_action = max_action
        data = (np.random.randn(num_basis_functions, state_dim), 0.1 * np.random.randn(num_basis_functions, control_dim))
        self._gp = _PolicyGP(data, ctx=ctx)
        self.create_models(data)
        self.centres = _PolicyData(self._gp, 0, "DataX")     # shared by every output (controllers.py:104-106)
        self.targets = _PolicyData(self._gp, 1, "DataY")

    # -- controllers.py:96-106
    def create_models(self, data):
        """One model per control dimension over shared centres: unit kernel variance (fixed), likelihood variance 1e-4
        (fixed), lengthscales one with the lower bound 1e-3 of the reference's transform."""
        if data is not None and (np.shape(data[0]) != np.shape(self._gp._X) or np.shape(data[1]) != np.shape(self._gp._Y)
                                 or not (np.array_equal(data[0], self._gp._X) and np.array_equal(data[1], self._gp._Y))):
            self._gp.set_data(data)                          # the reference builds the models FROM this data (controllers.py:96-106)
            self.num_basis_functions = self._gp.num_datapoints
        self._gp.create_models(data)
        for model in self._gp.models:
            model.kernel.variance.assign(1.0)               # controllers.py:92-93
            model.kernel.variance.trainable = False
            model.likelihood.variance.assign(1e-4)          # FakeGPR, controllers.py:67,76-77
            model.likelihood.variance.trainable = False
            model.kernel.lengthscales.lower = 1e-3          # positive(lower=1e-3), controllers.py:100
        self._gp._invalidate()

    def __getattr__(self, name):
        """The reference's RbfController IS an MGPR (controllers.py:80): whatever of that surface is not spelled out below
        (num_outputs, predict_given_factorizations, centralized_input, K, data, ...) is the policy GP's."""
        if name.startswith("_"):
            raise AttributeError(name)
        return getattr(self._gp, name)

    # -- the MGPR surface the reference's callers use on an RbfController
    @property
    def models(self):
        return self._gp.models

    @property
    def ctx(self):
        return self._gp.ctx

    def set_data(self, data):
        self._gp.set_data(data)
        self.num_basis_functions = self._gp.num_datapoints

    @property
    def X(self):
        return self._gp.X

    @property
    def Y(self):
        return self._gp.Y

    @property
    def lengthscales(self):
        return self._gp.lengthscales

    @property
    def variance(self):
        return self._gp.variance

    @property
    def noise(self):
        return self._gp.noise

    def calculate_factorizations(self):
        return self._gp.calculate_factorizations()

    def sync(self):
        """Push centres / targets / lengthscales to the device and refresh beta."""
        self._gp._user_factors = None
        self._gp._ensure_factorized()

    def policy_spec(self, squash=True):
        self.sync()
        return dict(kind=_lib.POLICY_RBF, state_dim=self.state_dim, control_dim=self.control_dim,
                    max_action=self.max_action, squash=squash)

    def compute_action(self, m, s, squash=True):
        return self.ctx.policy_action(self.policy_spec(squash), m, s)

    def randomize(self):
        """controllers.py:123-129, with its draw order from NumPy's global generator: model by model the centres (the
        shared X Parameter is re-assigned by every model, so the last model's draw stays), that model's targets, that
        model's lengthscales -- a seeded restart starts where the reference's starts for any control_dim."""
        bf, U = self._gp.X.shape[0], self._gp.Y.shape[1]
        Y = np.empty((bf, U))
        scale = np.broadcast_to(np.asarray(self.max_action, np.float64).reshape(-1), (U,)) if np.size(self.max_action) in (1, U) else None
        for k, m in enumerate(self.models):
            X = np.random.normal(size=self._gp.X.shape)
            draw = np.random.normal(size=(bf, 1))
            Y[:, k:k + 1] = (self.max_action if scale is None else scale[k]) / 10 * draw
            m.kernel.lengthscales.assign(1 + 0.1 * np.random.normal(size=m.kernel.lengthscales.shape))
        self._gp.set_data((X, Y))

    @property
    def trainable_parameters(self):
        """Centres, targets and per-output lengthscales -- the reference's trainable set (controllers.py:70-73,100)."""
        return [p for p in [self.centres, self.targets] + [m.kernel.lengthscales for m in self.models] if p.trainable]
