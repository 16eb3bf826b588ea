"""SMGPR: sparse FITC variant of MGPR with M inducing inputs.

Same surface as /root/reference/pilco/models/smgpr.py:11-52.  The FITC
factorisation (smgpr.py:24-45) runs on the device; the moment matching then runs
over Z instead of X (smgpr.py:47-52).  As in the reference, every output GP owns
an inducing set but prediction uses the one of model 0 for all outputs."""
from __future__ import annotations

import numpy as np

from ..params import Parameter, tensor_value
from .mgpr import MGPR


class _Inducing:
    def __init__(self, Z, on_change):
        self.Z = Parameter(Z, name="Z", on_change=on_change)


class SMGPR(MGPR):
    def __init__(self, data, num_induced_points, name=None, ctx=None):
        self.num_induced_points = num_induced_points
        MGPR.__init__(self, data, name, ctx=ctx)

    def create_models(self, data):
        MGPR.create_models(self, data)
        for model in self.models:
            Z = np.random.rand(self.num_induced_points, self.num_dims)      # smgpr.py:20
            model.inducing_variable = _Inducing(Z, self._z_changed)
        self._z_dirty = True

    def _z_changed(self):
        self._z_dirty = True

    def _after_set_data(self):
        self._z_dirty = True

    def _on_slot_taken(self):
        self._z_dirty = True
        self._reset_inducing = False

    def _sync(self):
        MGPR._sync(self)
        if self._z_dirty:
            self.ctx.gp_set_inducing(self._slot, self.Z, owner=self)
            self._z_dirty = False
            self._user_factors = None

    def _points(self):
        return self.Z

    # -- reference: MGPR.optimize applied to GPRFITC models (mgpr.py:47-75 with smgpr.py:16-22)
    def optimize(self, restarts=1, keep="last"):
        """Every output's GPRFITC model is fitted as the reference does it: kernel hyper-parameters, noise variance AND
        the output's own M x D inducing inputs by L-BFGS-B on the FITC marginal likelihood, evaluated with its analytic
        gradient on the device (pilco_gp_fitc_nlml, csrc/fitc_train.hip).  Prediction then uses model 0's inducing
        inputs for every output, as the reference does (smgpr.py:47-52)."""
        from ..training import optimize_smgpr
        return optimize_smgpr(self, restarts=restarts, keep=keep)

    @property
    def Z(self):
        return tensor_value(self.models[0].inducing_variable.Z.numpy())
