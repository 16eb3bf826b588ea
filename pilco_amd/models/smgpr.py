"""SMGPR: sparse FITC variant of MGPR with M inducing inputs.

Same surface as /root/reference/pilco/models/smgpr.py:11-52.  The FITC
factorisation (smgpr.py:24-45) runs on the device; the moment matching then runs
over Z instead of X (smgpr.py:47-52).  As in the reference, every output GP owns
an inducing set but prediction uses the one of model 0 for all outputs."""
from __future__ import annotations

import numpy as np

from ..params import Parameter
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
    def optimize(self, restarts=1, max_subset=1024):
        """Hyper-parameter fit for the sparse model.

        The reference hands every GPRFITC model (hyper-parameters AND its own M x D inducing inputs) to
        SciPy through GPflow's autodiff.  The FITC objective and its Z-gradient are not built here; instead
        (documented deviation, SURVEY.md 8c marks training as unpinned): the kernel hyper-parameters are fitted
        with the exact-GP objective on the device (pilco_gp_nlml) over at most ``max_subset`` randomly chosen
        data points, and the inducing inputs of every output are set to a random subset of the training inputs
        (the usual FITC initialisation) instead of being moved by gradient steps."""
        from ..training import optimize_mgpr
        n = self.num_datapoints
        idx = np.arange(n) if n <= max_subset else np.sort(np.random.choice(n, max_subset, replace=False))
        dense = MGPR((self._X[idx], self._Y[idx]), ctx=self.ctx)
        for src, dst in zip(self.models, dense.models):
            dst.kernel.lengthscales.assign(src.kernel.lengthscales.numpy())
            dst.kernel.variance.assign(src.kernel.variance.numpy())
            dst.likelihood.variance.assign(src.likelihood.variance.numpy())
        per = optimize_mgpr(dense, restarts=restarts)
        zi = np.random.choice(n, self.num_induced_points, replace=n < self.num_induced_points)
        for src, dst in zip(dense.models, self.models):
            dst.kernel.lengthscales.assign(src.kernel.lengthscales.numpy())
            dst.kernel.variance.assign(src.kernel.variance.numpy())
            dst.likelihood.variance.assign(src.likelihood.variance.numpy())
            dst.inducing_variable.Z.assign(self._X[zi])
        # the temporary dense model used this model's device slot: push everything again
        self._data_dirty = True
        self._hyp_dirty = True
        self._z_dirty = True
        self._user_factors = None
        return per

    @property
    def Z(self):
        return np.asarray(self.models[0].inducing_variable.Z.numpy(), np.float64)
