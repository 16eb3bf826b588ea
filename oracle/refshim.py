"""Stand-in modules for ``tensorflow`` / ``tensorflow_probability`` / ``gpflow``.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Nothing in ``pilco_amd``
may import this file.

Purpose: TensorFlow, TF-Probability and GPflow are not installed in the build
image, so the reference (/root/reference/pilco/*.py) cannot be executed as
is.  Its source, however, only uses ~40 ``tf.*`` ops and a thin slice of
GPflow (Parameter / Module / SquaredExponential / GPR / GPRFITC / Scipy).  This
file provides exactly that slice on top of torch-CPU-float64, so that
``oracle/ref_exec.py`` can import the reference's modules UNMODIFIED from where
they lie and run them: every line of ``pilco/models/mgpr.py``, ``smgpr.py``,
``pilco.py``, ``pilco/controllers.py`` and ``pilco/rewards.py`` then executes
as written (moment matching, FITC factorisation, propagate, the while-loop
rollout, controllers, rewards), with torch autograd standing in for TF
autodiff (so the reference's ``training_loss`` gradients are available too).

What is *reference source executed* and what is *recollection*:

* every ``tf.*`` entry below is a one-line mapping onto the torch op of the
  same documented meaning (``tf.transpose`` without ``perm`` reverses all axes,
  ``tf.linalg.solve(adjoint=True)`` solves A^H x = b, ``tf.linalg.cholesky_solve
  (L, rhs)``, ``tf.linalg.triangular_solve(lower=True, adjoint=False)``, ...);
* the GPflow arithmetic lives in third-party code that is not under
  /root/reference (requirements.txt:2 ``gpflow>=2.1.0,<2.2.0``).  Restated here
  from GPflow 2.1's published source (recollection, cannot be diffed here):
  ``SquaredExponential.K`` (inputs scaled by 1/lengthscale, squared distance by
  the |a|^2+|b|^2-2ab expansion, ``variance*exp(-r2/2)``), ``GPR`` /
  ``GPRFITC`` log marginal likelihoods, the softplus ``positive`` bijector with
  its lower bound, the Gaussian likelihood's 1e-6 variance floor, log-priors
  evaluated on the constrained value without a Jacobian term (PriorOn.
  CONSTRAINED, GPflow >= 2.0.1), and ``optimizers.Scipy`` = SciPy L-BFGS-B on
  the unconstrained variables.  Only ``SquaredExponential.K`` sits on the
  prediction path that the reference's tests pin.
"""
from __future__ import annotations

import contextlib
import types

import numpy as np
import torch

F64 = torch.float64


# =========================================================================== tensors
class T(torch.Tensor):
    """torch.Tensor with the few extra methods TF tensors have, and operators that
    accept ndarrays / lists / gpflow Parameters on either side."""

    def set_shape(self, shape):            # static-shape hint in TF; checked here
        assert tuple(self.shape) == tuple(shape), (tuple(self.shape), tuple(shape))

    def numpy(self):
        return torch.Tensor.numpy(self.detach().as_subclass(torch.Tensor))

    def __array__(self, dtype=None, copy=None):
        a = self.numpy()
        return a if dtype is None else a.astype(dtype)


def _t(x, dtype=None):
    """Anything -> T (float64 unless it already is an integer / bool tensor)."""
    if isinstance(x, Parameter):
        return x.value()
    if isinstance(x, torch.Tensor):
        out = x if isinstance(x, T) else x.as_subclass(T)
    elif isinstance(x, (list, tuple)) and any(isinstance(e, (torch.Tensor, Parameter)) for e in x):
        out = torch.stack([_t(e) for e in x]).as_subclass(T)
    else:
        a = np.asarray(x)
        if a.dtype.kind == "f" or dtype is not None:
            a = a.astype(np.float64)
        out = torch.from_numpy(np.ascontiguousarray(a)).as_subclass(T)
    if dtype is not None and out.dtype != dtype:
        out = out.to(dtype)
    return out


def _binop(name):
    base = getattr(torch.Tensor, name)

    def op(self, other):
        o = _t(other)
        if o.dtype != self.dtype and self.dtype == F64:
            o = o.to(F64)
        return base(self, o)
    op.__name__ = name
    return op


for _n in ("add", "sub", "mul", "truediv", "matmul", "pow"):
    setattr(T, f"__{_n}__", _binop(f"__{_n}__"))
    setattr(T, f"__r{_n}__", _binop(f"__r{_n}__"))
for _n in ("lt", "le", "gt", "ge"):
    setattr(T, f"__{_n}__", _binop(f"__{_n}__"))


# =========================================================================== gpflow.Parameter
class _Softplus:
    """gpflow.utilities.positive(lower): y = softplus(x) + lower."""

    def __init__(self, lower=0.0):
        self.lower = float(lower)

    def forward(self, x):
        return torch.nn.functional.softplus(x, threshold=1e9) + self.lower

    def inverse(self, y):
        z = np.asarray(y, np.float64) - self.lower
        if np.any(z <= 0):
            raise ValueError("value below the bijector's lower bound")
        return z + np.log(-np.expm1(-z))


def positive(lower=None, base=None):
    return _Softplus(0.0 if lower is None else lower)


class Parameter:
    """gpflow.Parameter: an unconstrained variable + bijector + prior + trainable flag."""

    def __init__(self, value, *, transform=None, prior=None, prior_on=None, trainable=True, dtype=None, name=None):
        if isinstance(value, Parameter):
            if transform is None:
                transform = value.transform
            value = value.numpy()
        elif isinstance(value, torch.Tensor):
            value = value.detach().numpy()
        value = np.array(value, dtype=np.float64)
        self.transform = transform
        self.prior = prior
        self.name = name
        u = value if transform is None else transform.inverse(value)
        self._u = torch.tensor(np.array(u, dtype=np.float64), dtype=F64, requires_grad=bool(trainable))

    # ---- value access
    def value(self):
        v = self._u if self.transform is None else self.transform.forward(self._u)
        return v.as_subclass(T)

    read_value = value

    @property
    def unconstrained_variable(self):
        return self._u

    def numpy(self):
        # a COPY, as tf.Variable.numpy() gives: the reference saves "best" parameter values with it and goes on optimising
        # (pilco.py:96,105); an alias of the variable's storage would silently follow the optimiser
        return np.array(self.value().numpy(), copy=True)

    def __array__(self, dtype=None, copy=None):
        a = self.numpy()
        return a if dtype is None else a.astype(dtype)

    def assign(self, value):
        if isinstance(value, Parameter):
            value = value.numpy()
        elif isinstance(value, torch.Tensor):
            value = value.detach().numpy()
        value = np.broadcast_to(np.asarray(value, np.float64), tuple(self._u.shape))
        u = value if self.transform is None else self.transform.inverse(value)
        with torch.no_grad():
            self._u.copy_(torch.from_numpy(np.array(u, dtype=np.float64)))
        return self

    @property
    def trainable(self):
        return self._u.requires_grad

    @trainable.setter
    def trainable(self, flag):
        self._u.requires_grad_(bool(flag))

    @property
    def shape(self):
        return self._u.shape

    @property
    def dtype(self):
        return F64

    def log_prior_density(self):
        if self.prior is None:
            return torch.zeros((), dtype=F64)
        return self.prior.log_prob(self.value()).sum()

    # ---- tensor protocol
    def __getitem__(self, idx):
        return self.value()[idx]

    def __iter__(self):
        return iter(self.value())

    def __len__(self):
        return self._u.shape[0]

    def __neg__(self):
        return -self.value()

    def __float__(self):
        return float(self.value())


def _pbin(name):
    def op(self, other):
        return getattr(self.value(), name)(other)
    op.__name__ = name
    return op


for _n in ("add", "sub", "mul", "truediv", "matmul", "pow"):
    setattr(Parameter, f"__{_n}__", _pbin(f"__{_n}__"))
    setattr(Parameter, f"__r{_n}__", _pbin(f"__r{_n}__"))
for _n in ("lt", "le", "gt", "ge"):
    setattr(Parameter, f"__{_n}__", _pbin(f"__{_n}__"))


def set_trainable(obj, flag):
    if isinstance(obj, Parameter):
        obj.trainable = flag
    else:
        for p in obj.parameters:
            p.trainable = flag


# =========================================================================== gpflow.Module
class Module:
    """tf.Module / gpflow.Module: attribute walk collecting Parameters (de-duplicated)."""

    def __init__(self, name=None):
        self._name = name

    @property
    def parameters(self):
        out, seen = [], set()

        def walk(o):
            if isinstance(o, Parameter):
                if id(o) not in seen:
                    seen.add(id(o))
                    out.append(o)
            elif isinstance(o, Module):
                if id(o) in seen:
                    return
                seen.add(id(o))
                for k in sorted(vars(o)):
                    walk(vars(o)[k])
            elif isinstance(o, (list, tuple)):
                for e in o:
                    walk(e)
            elif isinstance(o, dict):
                for k in sorted(o):
                    walk(o[k])
        walk(self)
        return tuple(out)

    @property
    def trainable_parameters(self):
        return tuple(p for p in self.parameters if p.trainable)

    @property
    def trainable_variables(self):
        return tuple(p.unconstrained_variable for p in self.trainable_parameters)


class BayesianModel(Module):
    """gpflow.models.BayesianModel: training_loss = -(objective + log prior)."""

    def log_prior_density(self):
        ps = [p for p in self.trainable_parameters if p.prior is not None]
        if not ps:
            return torch.zeros((), dtype=F64)
        return sum(p.log_prior_density() for p in ps)

    def training_loss(self):
        return -(self.maximum_log_likelihood_objective() + self.log_prior_density())


# =========================================================================== kernels / likelihoods / models
def _square_distance(X, X2):
    # gpflow.utilities.ops.square_distance (recollection): |a|^2 + |b|^2 - 2 a.b
    Xs = (X * X).sum(-1)
    if X2 is None:
        return -2.0 * X @ X.transpose(-1, -2) + Xs[:, None] + Xs[None, :]
    X2s = (X2 * X2).sum(-1)
    return -2.0 * X @ X2.transpose(-1, -2) + Xs[:, None] + X2s[None, :]


class SquaredExponential(Module):
    def __init__(self, variance=1.0, lengthscales=1.0, **kw):
        Module.__init__(self)
        self.variance = Parameter(variance, transform=positive())
        self.lengthscales = Parameter(lengthscales, transform=positive())

    def K(self, X, X2=None):
        ls = _t(self.lengthscales)
        Xs = _t(X) / ls
        X2s = None if X2 is None else _t(X2) / ls
        r2 = _square_distance(Xs, X2s)
        return (_t(self.variance) * torch.exp(-0.5 * r2)).as_subclass(T)

    def K_diag(self, X):
        return (_t(self.variance) * torch.ones(_t(X).shape[0], dtype=F64)).as_subclass(T)

    def __call__(self, X, X2=None, full_cov=True):
        return self.K(X, X2) if full_cov else self.K_diag(X)


class Gaussian(Module):
    """gpflow.likelihoods.Gaussian: variance with the 1e-6 lower bound (GPflow >= 2.0.2)."""

    def __init__(self, variance=1.0, variance_lower_bound=1e-6):
        Module.__init__(self)
        self.variance = Parameter(variance, transform=positive(lower=variance_lower_bound))


class GPR(BayesianModel):
    def __init__(self, data, kernel, mean_function=None, noise_variance=1.0):
        Module.__init__(self)
        self.data = (_t(data[0]), _t(data[1]))
        self.kernel = kernel
        self.likelihood = Gaussian(noise_variance)

    def maximum_log_likelihood_objective(self):
        X, Y = (_t(self.data[0]), _t(self.data[1]))
        N = X.shape[0]
        K = self.kernel.K(X)
        ks = K + _t(self.likelihood.variance) * torch.eye(N, dtype=F64)
        L = torch.linalg.cholesky(ks)
        alpha = torch.linalg.solve_triangular(L, Y, upper=False)
        # gpflow.logdensities.multivariate_normal, summed over the (single) output column
        return (-0.5 * (alpha * alpha).sum() - Y.shape[1] * torch.log(torch.diagonal(L)).sum()
                - 0.5 * N * Y.shape[1] * np.log(2 * np.pi))


class _InducingPoints(Module):
    def __init__(self, Z):
        Module.__init__(self)
        self.Z = Parameter(Z)

    def __len__(self):
        return self.Z.shape[0]


class GPRFITC(BayesianModel):
    """gpflow.models.GPRFITC (sgpr.py, GPflow 2.1): common_terms + fitc_log_marginal_likelihood."""

    def __init__(self, data, kernel, inducing_variable, mean_function=None, noise_variance=1.0):
        Module.__init__(self)
        self.data = (_t(data[0]), _t(data[1]))
        self.kernel = kernel
        self.likelihood = Gaussian(noise_variance)
        self.inducing_variable = _InducingPoints(inducing_variable)

    def maximum_log_likelihood_objective(self, jitter=1e-6):
        X, Y = self.data
        N, M = X.shape[0], len(self.inducing_variable)
        Z = _t(self.inducing_variable.Z)
        Kdiag = self.kernel.K_diag(X)
        kuf = self.kernel.K(Z, X)
        kuu = self.kernel.K(Z) + jitter * torch.eye(M, dtype=F64)
        Luu = torch.linalg.cholesky(kuu)
        V = torch.linalg.solve_triangular(Luu, kuf, upper=False)
        nu = Kdiag - (V * V).sum(0) + _t(self.likelihood.variance)
        B = torch.eye(M, dtype=F64) + (V / nu) @ V.transpose(0, 1)
        L = torch.linalg.cholesky(B)
        beta = Y / nu[:, None]
        alpha = V @ beta
        gamma = torch.linalg.solve_triangular(L, alpha, upper=False)
        maha = -0.5 * (Y * Y / nu[:, None]).sum() + 0.5 * (gamma * gamma).sum()
        const = -0.5 * N * np.log(2 * np.pi)
        logdet = -0.5 * torch.log(nu).sum() - torch.log(torch.diagonal(L)).sum()
        return maha + (const + logdet) * Y.shape[1]


# =========================================================================== optimizers.Scipy
class Scipy:
    """gpflow.optimizers.Scipy: scipy.optimize.minimize(method='L-BFGS-B', jac=True) over the
    flattened unconstrained variables; the closure is re-evaluated eagerly each time."""

    def minimize(self, closure, variables, method="L-BFGS-B", step_callback=None, compile=True, **scipy_kwargs):
        import scipy.optimize

        variables = [v for v in variables]
        if not variables:
            return None
        sizes = [v.numel() for v in variables]

        def unpack(x):
            off = 0
            with torch.no_grad():
                for v, n in zip(variables, sizes):
                    v.copy_(torch.from_numpy(x[off:off + n].reshape(tuple(v.shape)).copy()))
                    off += n

        def fun(x):
            unpack(x)
            for v in variables:
                v.grad = None
            try:
                loss = closure()
                loss = loss.sum() if isinstance(loss, torch.Tensor) else loss
                loss.backward()
            except torch.linalg.LinAlgError:
                return 1e30, np.zeros_like(x)
            g = np.concatenate([(v.grad if v.grad is not None else torch.zeros_like(v)).reshape(-1).numpy()
                                for v in variables])
            return float(loss.detach()), g.astype(np.float64)

        x0 = np.concatenate([v.detach().reshape(-1).numpy() for v in variables]).astype(np.float64)
        res = scipy.optimize.minimize(fun, x0, jac=True, method=method, **scipy_kwargs)
        unpack(res.x)
        return res


# =========================================================================== tfp.distributions
class Gamma:
    def __init__(self, concentration, rate):
        self.a, self.b = _t(concentration), _t(rate)

    def log_prob(self, x):
        x = _t(x)
        return self.a * torch.log(self.b) + (self.a - 1) * torch.log(x) - self.b * x - torch.lgamma(self.a)


class Normal:
    def __init__(self, loc, scale):
        self.loc, self.scale = _t(loc), _t(scale)

    def cdf(self, x):
        return (0.5 * (1 + torch.erf((_t(x) - self.loc) / (self.scale * np.sqrt(2.0))))).as_subclass(T)


# =========================================================================== tensorflow
def _tf_module():
    tf = types.ModuleType("tensorflow")
    tf.float64, tf.int32 = F64, torch.int32
    w = lambda r: r.as_subclass(T)

    def _axes(axis):
        if axis is None:
            return None
        return tuple(axis) if isinstance(axis, (list, tuple)) else int(axis)

    def reduce_sum(x, axis=None, keepdims=False):
        x = _t(x)
        return w(x.sum() if axis is None else x.sum(dim=_axes(axis), keepdim=keepdims))

    def transpose(x, perm=None):
        x = _t(x)
        return w(x.permute(*(perm if perm is not None else reversed(range(x.dim())))))

    def matmul(a, b, transpose_a=False, transpose_b=False, adjoint_a=False, adjoint_b=False, name=None):
        a, b = _t(a), _t(b)
        if transpose_a or adjoint_a:
            a = a.transpose(-1, -2)
        if transpose_b or adjoint_b:
            b = b.transpose(-1, -2)
        return w(a @ b)

    def eye(n, num_columns=None, batch_shape=None, dtype=F64, name=None):
        e = torch.eye(int(n), int(num_columns) if num_columns is not None else int(n), dtype=dtype)
        if batch_shape:
            e = e.expand(*[int(b) for b in batch_shape], *e.shape).clone()
        return w(e)

    def solve(A, B, adjoint=False, name=None):
        A = _t(A)
        return w(torch.linalg.solve(A.transpose(-1, -2) if adjoint else A, _t(B)))

    def triangular_solve(matrix, rhs, lower=True, adjoint=False, name=None):
        A = _t(matrix)
        if adjoint:
            return w(torch.linalg.solve_triangular(A.transpose(-1, -2), _t(rhs), upper=lower))
        return w(torch.linalg.solve_triangular(A, _t(rhs), upper=not lower))

    def while_loop(cond, body, loop_vars, **kw):
        vs = tuple(loop_vars)
        while bool(cond(*vs)):
            vs = tuple(body(*vs))
        return vs

    def constant(value, dtype=None, shape=None, name=None):
        if dtype in (torch.int32, torch.int64):
            return w(torch.tensor(value, dtype=dtype))
        return _t(value, F64 if dtype is None and np.asarray(value).dtype.kind == "f" else dtype)

    def shape(x):
        return tuple(_t(x).shape)

    def tile(x, multiples):
        return w(_t(x).repeat(*[int(k) for k in multiples]))

    def ones(shape_, dtype=F64, name=None):
        return w(torch.ones(*[int(k) for k in (shape_ if isinstance(shape_, (list, tuple)) else [shape_])], dtype=dtype))

    def zeros(shape_, dtype=F64, name=None):
        return w(torch.zeros(*[int(k) for k in (shape_ if isinstance(shape_, (list, tuple)) else [shape_])], dtype=dtype))

    def reshape(x, shape=None, name=None):  # noqa: A002 - tf's keyword is `shape`
        return w(_t(x).reshape(*[int(k) for k in shape]))

    tf.reduce_sum, tf.transpose, tf.matmul, tf.eye, tf.while_loop = reduce_sum, transpose, matmul, eye, while_loop
    tf.constant, tf.shape, tf.tile, tf.ones, tf.zeros, tf.reshape = constant, shape, tile, ones, zeros, reshape
    tf.convert_to_tensor = lambda x, dtype=None: _t(x, dtype)
    tf.cast = lambda x, dtype: _t(x, dtype)
    tf.exp = lambda x: w(torch.exp(_t(x)))
    tf.sin = lambda x: w(torch.sin(_t(x)))
    tf.cos = lambda x: w(torch.cos(_t(x)))
    tf.sqrt = lambda x: w(torch.sqrt(_t(x)))
    tf.square = lambda x: w(_t(x) * _t(x))
    tf.add = lambda a, b: w(_t(a) + _t(b))
    tf.multiply = lambda a, b: w(_t(a) * _t(b))
    tf.stack = lambda xs, axis=0: w(torch.stack([_t(x) for x in xs], dim=axis))
    tf.concat = lambda xs, axis: w(torch.cat([_t(x) for x in xs], dim=axis))
    tf.name_scope = lambda name: contextlib.nullcontext(name)

    tf.math = types.ModuleType("tensorflow.math")
    tf.math.log = lambda x: w(torch.log(_t(x)))

    la = types.ModuleType("tensorflow.linalg")
    la.diag = lambda x: w(torch.diag_embed(_t(x)))
    la.diag_part = lambda x: w(torch.diagonal(_t(x), dim1=-2, dim2=-1))
    la.matrix_transpose = lambda x: w(_t(x).transpose(-1, -2))
    la.det = lambda x: w(torch.linalg.det(_t(x)))
    la.cholesky = lambda x: w(torch.linalg.cholesky(_t(x)))
    la.cholesky_solve = lambda chol, rhs, name=None: w(torch.cholesky_solve(_t(rhs), _t(chol), upper=False))
    la.solve, la.triangular_solve, la.matmul = solve, triangular_solve, matmul
    tf.linalg = la
    return tf


def make_modules():
    """name -> module object for sys.modules injection (tensorflow, tensorflow_probability, gpflow)."""
    tf = _tf_module()

    tfp = types.ModuleType("tensorflow_probability")
    tfd = types.ModuleType("tensorflow_probability.distributions")
    tfd.Gamma, tfd.Normal = Gamma, Normal
    tfp.distributions = tfd

    g = types.ModuleType("gpflow")
    g.__path__ = []
    g.Parameter, g.Module, g.set_trainable = Parameter, Module, set_trainable
    g.default_float = lambda: F64
    cfg = types.ModuleType("gpflow.config")
    cfg.default_float = lambda: F64
    cfg.default_jitter = lambda: 1e-6
    util = types.ModuleType("gpflow.utilities")
    util.to_default_float = lambda x: _t(x, F64)
    util.positive, util.set_trainable = positive, set_trainable
    kern = types.ModuleType("gpflow.kernels")
    kern.SquaredExponential = kern.RBF = SquaredExponential
    lik = types.ModuleType("gpflow.likelihoods")
    lik.Gaussian = Gaussian
    models = types.ModuleType("gpflow.models")
    models.GPR, models.GPRFITC, models.BayesianModel = GPR, GPRFITC, BayesianModel
    opt = types.ModuleType("gpflow.optimizers")
    opt.Scipy = Scipy
    g.config, g.utilities, g.kernels, g.likelihoods, g.models, g.optimizers = cfg, util, kern, lik, models, opt

    return {
        "tensorflow": tf, "tensorflow.math": tf.math, "tensorflow.linalg": tf.linalg,
        "tensorflow_probability": tfp, "tensorflow_probability.distributions": tfd,
        "gpflow": g, "gpflow.config": cfg, "gpflow.utilities": util, "gpflow.kernels": kern,
        "gpflow.likelihoods": lik, "gpflow.models": models, "gpflow.optimizers": opt,
    }
