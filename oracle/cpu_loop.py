"""CPU stand-in for BASELINE config 5: the same optimise-models / optimise-policy steps the HIP path runs, on the
host cores, so that examples/inverted_pendulum.py and bench.py can time them side by side.

TEST / MEASUREMENT INFRASTRUCTURE ONLY (see oracle/__init__.py): nothing in pilco_amd imports this.

* optimize_models: MGPR.optimize (pilco/models/mgpr.py:47-75) -- per-output MAP objective (NLML from
  oracle/gp_train.py + the Gamma priors of mgpr.py:33-34, softplus parameters, noise floor 1e-6), SciPy L-BFGS-B, one
  randomised restart per output as the reference does with restarts=1; the same objective as pilco_amd/training.py.
* optimize_policy: PILCO.optimize_policy (pilco/models/pilco.py:75-113) for an RbfController -- torch-CPU-float64
  reverse mode through the restated rollout (oracle/torch_path.py) standing in for TensorFlow's, L-BFGS-B, maxiter
  as given.  It is a port ("kind": "port"), never the reference itself.
"""
from __future__ import annotations

import os
import time

import numpy as np
from scipy.optimize import minimize
from scipy.special import gammaln

from . import gp_train
from . import tf_path as tp

NOISE_LOWER = 1e-6
CPU_THREADS = 8     # torch intra-op threads used by optimize_policy (stated with the timings)
_sp = lambda u: np.logaddexp(0.0, u)
_spi = lambda x: np.where(x > 30.0, x, np.log(np.expm1(np.minimum(np.maximum(x, 1e-300), 30.0))))
_dsp = lambda u: 1.0 / (1.0 + np.exp(-u))


def _gamma(x, shape, rate):
    return shape * np.log(rate) - gammaln(shape) + (shape - 1.0) * np.log(x) - rate * x, (shape - 1.0) / x - rate


def optimize_models(X, Y, ls, var, noise, restarts=1, rs=None):
    """-> (ls (E,D), var (E), noise (E), seconds)."""
    rs = rs or np.random
    t0 = time.perf_counter()
    E, D = Y.shape[1], X.shape[1]
    ls, var, noise = np.array(ls, float), np.array(var, float), np.array(noise, float)
    for a in range(E):
        y = Y[:, a]

        def fun(u):
            l, v, n = _sp(u[:D]), _sp(u[D]), NOISE_LOWER + _sp(u[D + 1])
            try:
                f, g = gp_train.nlml_and_grad(X, y, l, v, n)
            except np.linalg.LinAlgError:
                return 1e25, np.zeros_like(u)
            pl, dpl = _gamma(l, 1.1, 0.1)
            pv, dpv = _gamma(v, 1.5, 0.5)
            gg = np.concatenate([(g[:D] - dpl) * _dsp(u[:D]), [(g[D] - dpv) * _dsp(u[D])], [g[D + 1] * _dsp(u[D + 1])]])
            return f - pl.sum() - pv, gg

        best_u, best_f = None, np.inf
        starts = [np.concatenate([_spi(ls[a]), [_spi(var[a])], [_spi(max(noise[a] - NOISE_LOWER, 1e-12))]])]
        for _ in range(restarts):      # randomize(model), mgpr.py:8-15
            starts.append(np.concatenate([_spi(1 + 0.01 * rs.normal(size=D)), [_spi(1 + 0.01 * rs.normal())],
                                          [_spi(1 + 0.01 * rs.normal() - NOISE_LOWER)]]))
        for u0 in starts:
            res = minimize(fun, u0, jac=True, method="L-BFGS-B", options=dict(maxiter=1000))
            if res.fun < best_f:
                best_u, best_f = res.x, res.fun
        ls[a], var[a], noise[a] = _sp(best_u[:D]), _sp(best_u[D]), NOISE_LOWER + _sp(best_u[D + 1])
    return ls, var, noise, time.perf_counter() - t0


def optimize_policy(X, Y, ls, var, noise, Xp, Yp, lsp, m_init, S_init, H, max_action, maxiter=50, reward_W=None, reward_t=None):
    """-> (Xp, Yp, lsp, reward, seconds)."""
    import torch
    from . import torch_path as tq
    torch.set_num_threads(min(CPU_THREADS, os.cpu_count() or 1))   # the ops are small: more threads only add dispatch overhead
    t0 = time.perf_counter()
    iK, beta = tp.calculate_factorizations(X, Y, ls, var, noise)
    bf, d = Xp.shape
    U = Yp.shape[1]
    lower = 1e-3
    gp = lambda m, s: tq.predict_given_factorizations(X, ls, var, m, s, iK, beta)
    nzp = torch.full((U,), 1e-4, dtype=torch.float64)
    shapes = [(bf, d), (bf, U), (U, d)]
    sizes = [bf * d, bf * U, U * d]

    def unpack(u):
        parts, off = [], 0
        for shp, n in zip(shapes, sizes):
            parts.append(u[off:off + n].reshape(shp))
            off += n
        return parts

    def fun(u):
        ut = torch.tensor(u, dtype=torch.float64, requires_grad=True)
        tX, tY, tu = unpack(ut)
        tl = lower + torch.nn.functional.softplus(tu)
        pol = lambda m, s: tq.rbf_controller(m, s, tX, tY, tl, nzp, max_action)
        rw = lambda m, s: tq.exponential_reward(m, s, reward_W, reward_t)
        _, _, R = tq.predict(gp, pol, rw, tq.t(m_init), tq.t(S_init), H)
        loss = -R.sum()
        loss.backward()
        return float(loss.detach()), ut.grad.numpy().copy()

    u0 = np.concatenate([Xp.ravel(), Yp.ravel(), _spi(lsp - lower).ravel()])
    res = minimize(fun, u0, jac=True, method="L-BFGS-B", options=dict(maxiter=maxiter))
    Xn, Yn, un = unpack(res.x)
    return Xn.copy(), Yn.copy(), lower + _sp(un), -float(res.fun), time.perf_counter() - t0


def compute_action(x, Xp, Yp, lsp, max_action):
    """pilco.py:115-116: controller mean at zero input covariance."""
    return tp.rbf_controller(x.reshape(1, -1), np.zeros((x.size, x.size)), Xp, Yp, lsp, max_action=max_action)[0]
