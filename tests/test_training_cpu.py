"""The host side of MGPR.optimize / SMGPR.optimize (pilco_amd/training.py: transforms, priors, per-output L-BFGS-B runs in
lockstep, restart draws, bookkeeping) against the end points of the EXECUTED reference (tests/golden/models_optimisation.npz,
sparse_models_optimisation.npz), with the objective values supplied by a CPU stand-in for the device calls
(tests/helpers/cpu_objective_context.py).  The same fixtures are met on the GPU with the device objective
(tests/test_gpu_parity.py::test_*_optimize_ends_where_the_executed_reference_ends)."""
import os

import numpy as np
import pytest

from helpers.cpu_objective_context import CpuObjectiveContext
from pilco_amd import training
from pilco_amd.models import MGPR, SMGPR

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _start(m, g):
    for i, mdl in enumerate(m.models):
        mdl.kernel.lengthscales.assign(g["ls_start"][i])
        mdl.kernel.variance.assign(g["var_start"][i])
        mdl.likelihood.variance.assign(g["noise_start"][i])


def test_exact_models_end_where_the_executed_reference_ends_with_and_without_restarts():
    g = np.load(os.path.join(GOLDEN, "models_optimisation.npz"))
    m = MGPR((g["X"], g["Y"]), ctx=CpuObjectiveContext())
    _start(m, g)
    per = m.optimize(restarts=0)
    np.testing.assert_allclose(per, g["loss_end"], rtol=1e-6)
    np.testing.assert_allclose(m.lengthscales, g["ls_end"], rtol=2e-2)
    np.testing.assert_allclose(m.noise, g["noise_end"], rtol=2e-2)
    for keep in ("last", "best"):
        np.random.seed(int(g["restart_seed"]))
        m2 = MGPR((g["X"], g["Y"]), ctx=CpuObjectiveContext())
        _start(m2, g)
        per2 = m2.optimize(restarts=int(g["restarts"]), keep=keep)
        if keep == "last":     # the reference's bookkeeping leaves the last restart's fit assigned (mgpr.py:59-75)
            np.testing.assert_allclose(per2, g["r_loss_end"], rtol=1e-6)
            np.testing.assert_allclose(m2.lengthscales, g["r_ls_end"], rtol=2e-2)
        else:
            assert np.all(per2 <= np.minimum(g["loss_end"], g["r_loss_end"]) * (1 + 1e-6))
    with pytest.raises(ValueError):
        m.optimize(restarts=0, keep="first")


def test_sparse_models_end_where_the_executed_reference_ends():
    g = np.load(os.path.join(GOLDEN, "sparse_models_optimisation.npz"))
    np.random.seed(2)
    ctx = CpuObjectiveContext()
    m = SMGPR((g["X"], g["Y"]), num_induced_points=g["Z_start"].shape[1], ctx=ctx)
    _start(m, g)
    for i, mdl in enumerate(m.models):
        mdl.inducing_variable.Z.assign(g["Z_start"][i])
    before, _ = training.smgpr_objective(m, np.concatenate([training._mgpr_pack(m), g["Z_start"].ravel()]))
    np.testing.assert_allclose(before, g["loss_start"], rtol=1e-10)
    per = m.optimize(restarts=0)
    np.testing.assert_allclose(per, g["loss_end"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(m.lengthscales, g["ls_end"], rtol=5e-3)
    np.testing.assert_allclose(m.variance, g["var_end"], rtol=5e-3)


def test_sparse_restarts_keep_their_inducing_inputs_and_the_iteration_cap_is_the_module_default(monkeypatch):
    """randomize() leaves the inducing inputs alone (mgpr.py:8-15): a restart starts from the inducing inputs the previous fit
    ended with.  MODEL_FIT_MAXITER (SciPy's default, as the reference passes no options) bounds every fit."""
    g = np.load(os.path.join(GOLDEN, "sparse_models_optimisation.npz"))
    monkeypatch.setattr(training, "MODEL_FIT_MAXITER", 5)
    starts = []
    real = training.lockstep_minimize

    def spy(eval_all, u0, parts, maxiter, wall):
        starts.append((np.array(u0), maxiter))
        out = real(eval_all, u0, parts, maxiter, wall)
        starts.append((np.array(out[0]), None))
        return out
    monkeypatch.setattr(training, "lockstep_minimize", spy)
    np.random.seed(3)
    ctx = CpuObjectiveContext()
    m = SMGPR((g["X"], g["Y"]), num_induced_points=g["Z_start"].shape[1], ctx=ctx)
    _start(m, g)
    for i, mdl in enumerate(m.models):
        mdl.inducing_variable.Z.assign(g["Z_start"][i])
    per = m.optimize(restarts=1, keep="last")
    (u0a, it_a), (enda, _), (u0b, it_b), (endb, _) = starts
    nk = 2 * 3 + 2 * 2
    assert it_a == it_b == 5
    assert np.array_equal(u0b[nk:], enda[nk:]) and not np.array_equal(u0b[:nk], enda[:nk])
    Zend = np.stack([mdl.inducing_variable.Z.numpy() for mdl in m.models])
    assert np.array_equal(Zend.ravel(), endb[nk:]) and np.all(np.isfinite(per))
    assert ctx.calls < 80            # 2 fits x (at most ~5 iterations of a few evaluations): every round is ONE batched call


@pytest.mark.parametrize("cls", [MGPR, SMGPR])
def test_a_fixed_likelihood_variance_stays_where_it_is(cls, monkeypatch):
    """set_trainable(model.likelihood.variance, False) (examples/safe_cars_run.py:88-90): the fits leave the noise alone
    and randomize() draws nothing for it (mgpr.py:13-15), restarts included."""
    from pilco_amd.params import set_trainable
    g = np.load(os.path.join(GOLDEN, "sparse_models_optimisation.npz"))
    monkeypatch.setattr(training, "MODEL_FIT_MAXITER", 4)
    np.random.seed(1)
    kw = dict(num_induced_points=g["Z_start"].shape[1]) if cls is SMGPR else {}
    m = cls((g["X"], g["Y"]), ctx=CpuObjectiveContext(), **kw)
    _start(m, g)
    for mdl in m.models:
        mdl.likelihood.variance.assign(0.001)
        set_trainable(mdl.likelihood.variance, False)
    ls0 = m.lengthscales.copy()
    np.random.seed(7)
    m.optimize(restarts=1, keep="last")
    drawn = np.random.normal()
    np.random.seed(7)
    np.random.normal(size=2 * (3 + 1))            # 2 models x (3 lengthscales + 1 kernel variance), nothing for the noise
    assert np.random.normal() == drawn
    assert np.array_equal(m.noise, [0.001, 0.001]) and not np.allclose(m.lengthscales, ls0)
