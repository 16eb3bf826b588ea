"""MGPR: E independent squared-exponential GPs sharing their inputs, with the
analytic moment-matching prediction at Gaussian inputs.

Same public surface as the reference's ``pilco.models.MGPR``
(/root/reference/pilco/models/mgpr.py:17-190); the arithmetic runs in
libpilco_hip.so on the MI355X.  Differences from the reference, all API-legal:
the factorisation is cached on the device and recomputed only when data or
hyper-parameters change (the reference recomputes it on every
``predict_on_noisy_inputs`` call, mgpr.py:77-79).
"""
from __future__ import annotations

import numpy as np

from .. import _lib
from ..params import GPModelView, parameters_of, tensor_value


def randomize(model, mean=1, sigma=0.01):
    """mgpr.py:8-15: a fresh start for one output's model -- Normal(mean, sigma) draws from NumPy's global generator for the
    lengthscales, the kernel variance and (only if trainable) the likelihood variance, in that order."""
    k, lik = model.kernel, model.likelihood.variance
    k.lengthscales.assign(mean + sigma * np.random.normal(size=np.shape(k.lengthscales.numpy())))
    k.variance.assign(mean + sigma * np.random.normal(size=np.shape(k.variance.numpy())))
    if lik.trainable:
        lik.assign(mean + sigma * np.random.normal())


class MGPR:
    _slot = _lib.SLOT_DYNAMICS

    def __init__(self, data, name=None, ctx=None):
        self.name = name
        self._ctx = ctx
        X, Y = np.asarray(data[0], np.float64), np.asarray(data[1], np.float64)
        self.num_outputs = Y.shape[1]
        self.num_dims = X.shape[1]
        self.num_datapoints = X.shape[0]
        self._X, self._Y = X.copy(), Y.copy()
        self._data_dirty = True
        self._hyp_dirty = True
        self._user_factors = None
        self.create_models(data)
        self.optimizers = []

    # -- reference: mgpr.py:28-36
    def create_models(self, data):
        self.models = [GPModelView(self, i, self.num_dims) for i in range(self.num_outputs)]

    @property
    def ctx(self):
        if self._ctx is None:
            self._ctx = _lib.resolve_ctx(self)
        return self._ctx

    def _invalidate(self):
        self._hyp_dirty = True

    # -- reference: mgpr.py:38-45
    def set_data(self, data):
        X, Y = np.asarray(data[0], np.float64), np.asarray(data[1], np.float64)
        if X.shape[1] != self.num_dims or Y.shape[1] != self.num_outputs:
            raise ValueError("set_data: D and E are fixed at construction")
        self._X, self._Y = X.copy(), Y.copy()
        self.num_datapoints = X.shape[0]
        self._data_dirty = True

    # -- device synchronisation
    def _points(self):
        return self._X

    def _claim_slot(self):
        """The device slot is shared by every model of this kind on the context.  If another instance used it since
        this one last synchronised, everything this instance believes to be on the device is stale: push it all again
        (data, hyper-parameters, inducing inputs, factorisation)."""
        owner = self.ctx._slot_owner.get(self._slot)
        if owner is not self:
            self.ctx._slot_owner[self._slot] = self
            self._data_dirty = True
            self._hyp_dirty = True
            self._user_factors = None
            self._on_slot_taken()

    def _on_slot_taken(self):
        self._reset_inducing = True        # an SMGPR may have left inducing inputs in the slot

    def _sync(self):
        self._claim_slot()
        if self._data_dirty:
            self.ctx.gp_set_data(self._slot, self._X, self._Y, owner=self)
            self._data_dirty = False
            self._hyp_dirty = True
            if getattr(self, "_reset_inducing", False):
                self.ctx.gp_set_inducing(self._slot, None, owner=self)      # back to the exact GP (pilco_gp_set_data keeps M otherwise)
                self._reset_inducing = False
            self._after_set_data()
        if self._hyp_dirty:
            self.ctx.gp_set_hyp(self._slot, self.lengthscales, self.variance, self.noise, owner=self)
            self._hyp_dirty = False
            self._user_factors = None

    def _after_set_data(self):
        pass

    def _ensure_factorized(self):
        self._sync()
        if self._user_factors is None:
            self.ctx.gp_factorize(self._slot)                   # cached on the device: a no-op while nothing changed

    # -- reference: mgpr.py:47-75
    def optimize(self, restarts=1, keep="last"):
        """keep='last': the reference's end state (the last restart's fit stays, mgpr.py:59-75); keep='best': per output the
        better fit (extension).  See training.optimize_mgpr."""
        from ..training import optimize_mgpr
        return optimize_mgpr(self, restarts=restarts, keep=keep)

    # -- reference: mgpr.py:77-79
    def predict_on_noisy_inputs(self, m, s):
        self._user_factors = None
        self._ensure_factorized()
        return self.ctx.gp_predict(self._slot, m, s, self.num_dims, self.num_outputs)

    # -- reference: mgpr.py:81-89
    def calculate_factorizations(self):
        self._user_factors = None
        self._ensure_factorized()
        return self.ctx.gp_get_factors(self._slot, self.num_outputs)

    # -- reference: mgpr.py:91-149
    def predict_given_factorizations(self, m, s, iK, beta):
        self._sync()
        iK = None if iK is None else np.asarray(iK, np.float64)
        if iK is not None and not np.any(iK):
            iK = None  # 0.0 * iK of the RBF controller (controllers.py:116): skip the stream
        self.ctx.gp_set_factors(self._slot, iK, beta, owner=self)
        self._user_factors = True
        return self.ctx.gp_predict(self._slot, m, s, self.num_dims, self.num_outputs)

    # -- reference: mgpr.py:151-157
    def centralized_input(self, m):
        return self._points() - np.asarray(m, np.float64).reshape(1, -1)

    def K(self, X1, X2=None):
        self._sync()
        return self.ctx.gp_gram(self._slot, X1, X2, self.num_outputs)

    # -- reference: mgpr.py:159-190
    @property
    def Y(self):
        return tensor_value(self._Y)

    @property
    def X(self):
        return tensor_value(self._X)

    @property
    def lengthscales(self):
        return tensor_value(np.stack([np.asarray(m.kernel.lengthscales.numpy(), np.float64).reshape(-1) for m in self.models]))

    @property
    def variance(self):
        return tensor_value(np.array([float(m.kernel.variance.numpy()) for m in self.models]))

    @property
    def noise(self):
        return tensor_value(np.array([float(m.likelihood.variance.numpy()) for m in self.models]))

    @property
    def data(self):
        return (self.X, self.Y)

    @property
    def trainable_parameters(self):
        return [p for p in parameters_of(self.models) if p.trainable]
