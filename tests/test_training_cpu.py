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

    def spy(eval_all, u0, parts, maxiter, wall, **kw):
        starts.append((np.array(u0), maxiter))
        out = real(eval_all, u0, parts, maxiter, wall, **kw)
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


def test_mixed_trainability_is_per_output_and_draws_follow_the_reference_order():
    """mgpr.py:47-66 hands model.trainable_variables to the optimiser MODEL BY MODEL and randomize() (mgpr.py:8-15) draws a
    likelihood variance only for a model whose variance is trainable: with output 1's noise fixed, output 0's noise is
    still fitted, output 1's stays put, a fixed lengthscale vector stays put, and the global generator ends where the
    reference's draw sequence ends (D + 1 + [1 if trainable] normals per model and restart)."""
    g = np.load(os.path.join(GOLDEN, "models_optimisation.npz"))
    X, Y = g["X"], g["Y"]
    E, D = Y.shape[1], X.shape[1]
    m = MGPR((X, Y), ctx=CpuObjectiveContext())
    _start(m, g)
    m.models[1].likelihood.variance.trainable = False
    m.models[0].kernel.lengthscales.trainable = False
    ls0, nz1 = m.lengthscales[0].copy(), float(m.noise[1])
    per = m.optimize(restarts=0)
    assert np.array_equal(m.lengthscales[0], ls0) or np.allclose(m.lengthscales[0], ls0, rtol=1e-14)
    assert float(m.noise[1]) == nz1
    assert abs(float(m.noise[0]) - float(g["noise_start"][0])) > 1e-9          # output 0's noise WAS fitted
    assert np.all(np.isfinite(per))
    # draw order with restarts
    np.random.seed(5)
    m.optimize(restarts=2)
    after = np.random.normal()
    np.random.seed(5)
    for a in range(E):
        for r in range(2):
            np.random.normal(size=(D,))
            np.random.normal(size=())
            if a != 1:
                np.random.normal()
    assert after == np.random.normal()
    assert float(m.noise[1]) == nz1


def test_not_positive_definite_output_is_isolated_in_the_model_fit():
    """One output whose Gram matrix fails during the fit (here: forced by the stand-in for lengthscales below a threshold)
    is a wall for that output alone: the other output ends where it ends when fitted alone."""
    from pilco_amd import _lib
    g = np.load(os.path.join(GOLDEN, "models_optimisation.npz"))

    class Flaky(CpuObjectiveContext):
        """output 1 'fails its Cholesky' whenever its first lengthscale exceeds a threshold the fit crosses"""
        thr = None

        def gp_nlml(self, slot, D, E, want_grad=True):
            if self.thr is not None and self.ls[1][0] > self.thr:
                exc = _lib.NotPositiveDefiniteError(2, "output 1 (stand-in)")
                exc.output = 1
                self.hits = getattr(self, "hits", 0) + 1
                raise exc
            return super().gp_nlml(slot, D, E, want_grad)

    ref = MGPR((g["X"], g["Y"]), ctx=CpuObjectiveContext())
    _start(ref, g)
    per_ref = ref.optimize(restarts=0)
    ctx = Flaky()
    m = MGPR((g["X"], g["Y"]), ctx=ctx)
    _start(m, g)
    lo, hi = float(g["ls_start"][1][0]), float(ref.lengthscales[1][0])
    ctx.thr = lo + 0.5 * (hi - lo) if hi > lo else 1e9
    per = m.optimize(restarts=0)
    if hi > lo:
        assert ctx.hits > 0                                   # the wall was hit ...
        assert m.lengthscales[1][0] <= ctx.thr                 # ... and respected by output 1
    np.testing.assert_allclose(per[0], per_ref[0], rtol=1e-12)   # output 0 never noticed
    np.testing.assert_allclose(m.lengthscales[0], ref.lengthscales[0], rtol=1e-12)
