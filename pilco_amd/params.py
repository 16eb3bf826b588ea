"""Minimal stand-ins for the gpflow objects the reference's callers touch:
``Parameter`` (``.numpy()``, ``.assign()``, ``.trainable``, ``.prior``),
``set_trainable`` and the ``model.kernel.lengthscales`` / ``model.kernel.variance``
/ ``model.likelihood.variance`` attribute chain used by the reference's tests
and examples (tests/test_cascade.py:49-51, examples/mountain_car.py:51-53)."""
from __future__ import annotations

import numpy as np


class TensorValue(np.ndarray):
    """What the reference's read-only properties return is a TensorFlow tensor (mgpr.py:159-190, smgpr.py:50-52): callers
    use it as an array or call .numpy() on it (tests/test_sparse_predictions.py:47).  An ndarray that also answers .numpy()."""

    def numpy(self):
        return np.asarray(self)


def tensor_value(a):
    return np.asarray(a, np.float64).view(TensorValue)


class Parameter:
    def __init__(self, value, trainable=True, name=None, lower=None, on_change=None):
        self._v = np.array(value, dtype=np.float64)
        self.trainable = trainable
        self.name = name
        self.prior = None
        self.lower = lower          # positive(lower=...) transform of the reference's RBF lengthscales
        self._on_change = on_change

    def numpy(self):
        return self._v.copy() if self._v.ndim else float(self._v)

    def value(self):
        """gpflow.Parameter.value() / read_value(): the constrained value as a tensor (examples/safe_swimmer_run.py:115
        multiplies it: `R.coefs.assign(R.coefs.value() * [...])`)."""
        return tensor_value(self._v.copy())

    read_value = value

    def assign(self, value):
        value = value.numpy() if isinstance(value, Parameter) else value
        v = np.asarray(value, dtype=np.float64)
        if v.shape != self._v.shape:
            v = np.broadcast_to(v, self._v.shape) if v.size == 1 else v.reshape(self._v.shape)
        self._v = np.array(v, dtype=np.float64)
        if self._on_change is not None:
            self._on_change()

    @property
    def shape(self):
        return self._v.shape

    def __array__(self, dtype=None, copy=None):
        return self._v if dtype is None else self._v.astype(dtype)

    def __repr__(self):
        return f"Parameter({self._v!r}, trainable={self.trainable})"


def set_trainable(obj, flag):
    """gpflow.set_trainable for a Parameter or any object holding Parameters."""
    if isinstance(obj, Parameter):
        obj.trainable = bool(flag)
        return
    for p in parameters_of(obj):
        p.trainable = bool(flag)


def parameters_of(obj, _seen=None):
    _seen = set() if _seen is None else _seen
    out = []
    if id(obj) in _seen:
        return out
    _seen.add(id(obj))
    if isinstance(obj, Parameter):
        return [obj]
    if isinstance(obj, (list, tuple)):
        for o in obj:
            out += parameters_of(o, _seen)
        return out
    if hasattr(obj, "__dict__"):
        for k, v in vars(obj).items():
            if k.startswith("_"):
                continue
            if isinstance(v, (Parameter, list, tuple)) or hasattr(v, "__dict__"):
                if isinstance(v, (np.ndarray, str, bytes)) or callable(v) and not hasattr(v, "__dict__"):
                    continue
                out += parameters_of(v, _seen)
    return out


class _Kernel:
    def __init__(self, D, on_change):
        self.lengthscales = Parameter(np.ones(D), name="lengthscales", on_change=on_change)
        self.variance = Parameter(1.0, name="variance", on_change=on_change)


class _Likelihood:
    def __init__(self, on_change, variance=1.0):
        self.variance = Parameter(variance, name="likelihood_variance", on_change=on_change)


class GPModelView:
    """What ``mgpr.models[i]`` exposes (gpflow.models.GPR in the reference)."""

    def __init__(self, owner, index, D, noise=1.0):
        self._owner = owner
        self.index = index
        cb = owner._invalidate
        self.kernel = _Kernel(D, cb)
        self.likelihood = _Likelihood(cb, noise)

    @property
    def data(self):
        X, Y = self._owner._X, self._owner._Y
        return (X, Y[:, self.index:self.index + 1])
