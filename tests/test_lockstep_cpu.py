"""pilco_amd.training.lockstep_minimize on the CPU: E independent L-BFGS-B problems evaluated through ONE batched callback
per round must end exactly where E separate scipy.optimize.minimize runs end (the reference fits one optimiser per output,
mgpr.py:47-56) -- and a joint run on the sum of the losses need not."""
import numpy as np
from scipy.optimize import minimize

from pilco_amd.training import lockstep_minimize


def _problems(rs, E, n):
    return [(rs.randn(n, n), rs.randn(n), 0.5 + rs.rand()) for _ in range(E)]


def _f(prob, x):
    """A smooth non-convex test function with several local minima."""
    A, b, c = prob
    y = A @ x - b
    val = 0.5 * y @ y + c * np.sum(np.cos(3.0 * x))
    return val, A.T @ y - 3.0 * c * np.sin(3.0 * x)


def test_lockstep_runs_equal_separate_runs_bitwise():
    rs = np.random.RandomState(0)
    E, n = 5, 4
    probs = _problems(rs, E, n)
    parts = [np.arange(a * n, (a + 1) * n) for a in range(E)]
    u0 = rs.randn(E * n)
    calls = []

    def eval_all(u):
        calls.append(1)
        vals, grad = np.empty(E), np.empty(E * n)
        for a in range(E):
            vals[a], grad[parts[a]] = _f(probs[a], u[parts[a]])
        return vals, grad

    u, vals = lockstep_minimize(eval_all, u0, parts, maxiter=200)
    n_evals = []
    for a in range(E):
        cnt = []
        res = minimize(lambda x: (cnt.append(1), _f(probs[a], x))[1], u0[parts[a]], jac=True, method="L-BFGS-B", options=dict(maxiter=200))
        n_evals.append(len(cnt))
        assert np.array_equal(res.x, u[parts[a]]) and res.fun == vals[a]
    # one batched evaluation per round: as many rounds as the longest single run needs (+ the closing evaluation)
    assert len(calls) == max(n_evals) + 1


def test_a_failing_round_is_a_wall_not_a_crash():
    parts = [np.arange(2), np.arange(2, 4)]

    def eval_all(u):
        if u[0] > 2.0:
            raise RuntimeError("not positive definite (stand-in)")
        return np.array([np.sum((u[:2] - 1.0) ** 2), np.sum((u[2:] + 0.5) ** 2)]), 2.0 * np.concatenate([u[:2] - 1.0, u[2:] + 0.5])

    u, vals = lockstep_minimize(eval_all, np.array([0.0, 0.0, 3.0, 3.0]), parts, maxiter=100)
    np.testing.assert_allclose(u, [1.0, 1.0, -0.5, -0.5], atol=1e-5)


def test_an_error_in_the_batched_evaluation_or_in_one_run_releases_every_thread():
    import threading
    parts = [np.arange(2), np.arange(2, 4), np.arange(4, 6)]
    n = [0]

    def eval_all(u):
        n[0] += 1
        if n[0] == 3:
            raise ValueError("device error (stand-in)")      # not a wall: must surface, with no run left waiting
        return np.array([np.sum(u[p] ** 4) for p in parts]), 4.0 * u ** 3

    before = threading.active_count()
    try:
        lockstep_minimize(eval_all, np.arange(1.0, 7.0), parts, maxiter=50)
        raise AssertionError("the error was swallowed")
    except ValueError as exc:
        assert "stand-in" in str(exc)
    assert threading.active_count() == before

    # a run that fails inside its optimiser while the others wait for the round: surfaced, nobody left waiting
    from pilco_amd import training
    real = training.minimize

    def flaky(fun, x0, **kw):
        if x0[0] == 3.0:
            fun(x0)
            raise FloatingPointError("optimiser error (stand-in)")
        return real(fun, x0, **kw)
    training.minimize = flaky
    try:
        n[0] = 10
        lockstep_minimize(eval_all, np.arange(1.0, 7.0), parts, maxiter=50)
        raise AssertionError("the error was swallowed")
    except FloatingPointError:
        pass
    finally:
        training.minimize = real
    assert threading.active_count() == before


def test_a_wall_is_confined_to_the_output_that_caused_it():
    """The reference runs one optimiser per output (mgpr.py:47-56): a Gram matrix that is not positive definite at output
    0's trial point must not feed a spurious 1e25 / zero-gradient evaluation to output 1.  The batched evaluation names
    the failing output (attribute `output`, pilco_last_not_pd_output); with it, every output ends exactly -- to the last
    bit, after the same number of evaluations -- where its own separate scipy run with a private wall ends."""
    rs = np.random.RandomState(3)
    E, n = 3, 3
    probs = _problems(rs, E, n)
    parts = [np.arange(a * n, (a + 1) * n) for a in range(E)]
    u0 = rs.randn(E * n)
    limit = [0.9, 1e9, 0.7]          # outputs 0 and 2 have a forbidden region (x[0] > limit), output 1 has none

    class Wall(RuntimeError):
        pass

    seen = {a: [] for a in range(E)}

    def eval_all(u):
        for a in range(E):           # like the batched factorisation: reports the FIRST failing output
            if u[parts[a]][0] > limit[a]:
                exc = Wall("not positive definite (stand-in)")
                exc.output = a
                raise exc
        vals, grad = np.empty(E), np.empty(E * n)
        for a in range(E):
            vals[a], grad[parts[a]] = _f(probs[a], u[parts[a]])
        return vals, grad

    u, vals = lockstep_minimize(eval_all, u0, parts, maxiter=200, wall=(Wall,))
    for a in range(E):
        def single(x, a=a):
            seen[a].append(1)
            if x[0] > limit[a]:
                return 1e25, np.zeros(n)
            return _f(probs[a], x)
        res = minimize(single, u0[parts[a]], jac=True, method="L-BFGS-B", options=dict(maxiter=200))
        assert np.array_equal(res.x, u[parts[a]]), a
        assert res.fun == vals[a], a
    # the walls were actually hit (otherwise this test shows nothing)
    hit = [0]

    def counting(u):
        try:
            return eval_all(u)
        except Wall:
            hit[0] += 1
            raise
    lockstep_minimize(counting, u0, parts, maxiter=200, wall=(Wall,))
    assert hit[0] > 0


def test_a_start_that_cannot_be_evaluated_walls_only_its_own_output():
    parts = [np.arange(2), np.arange(2, 4)]

    class Wall(RuntimeError):
        pass

    def eval_all(u):
        if u[0] > 2.0:
            exc = Wall("not positive definite (stand-in)")
            exc.output = 0
            raise exc
        return np.array([np.sum((u[:2] - 1.0) ** 2), np.sum((u[2:] + 0.5) ** 2)]), 2.0 * np.concatenate([u[:2] - 1.0, u[2:] + 0.5])

    # output 0 starts inside its forbidden region: without a safe point nothing can be pinned on it -> the round is a wall
    # for both (old behaviour, still not a crash) ...
    u, vals = lockstep_minimize(eval_all, np.array([3.0, 0.0, 3.0, 3.0]), parts, maxiter=100, wall=(Wall,))
    assert vals[0] == 1e25
    # ... with one, output 1 is fitted as if output 0 did not exist, and output 0 reports the wall value at its end point
    u, vals = lockstep_minimize(eval_all, np.array([3.0, 0.0, 3.0, 3.0]), parts, maxiter=100, wall=(Wall,), safe=np.zeros(4))
    np.testing.assert_allclose(u[2:], [-0.5, -0.5], atol=1e-5)
    assert vals[0] == 1e25 and vals[1] < 1e-9


def test_two_live_pilco_objects_get_a_context_each_and_a_dead_ones_context_is_handed_out_again():
    """_lib.context_for: a context holds one dynamics model on the device, so PILCO objects that are alive together must not
    share the default context (round 2: every switch between them re-uploaded and re-factorised the model)."""
    import gc

    from helpers.cpu_rollout_context import CpuRolloutContext
    from pilco_amd import _lib
    from pilco_amd.controllers import LinearController
    from pilco_amd.models import PILCO

    saved = _lib._default_ctx
    try:
        _lib.set_context(CpuRolloutContext())
        rng = np.random.default_rng(0)
        data = (rng.standard_normal((12, 3)), rng.standard_normal((12, 2)))
        p1 = PILCO(data, horizon=2)
        p2 = PILCO(data, horizon=2)
        assert p1.ctx is _lib.get_context() and p2.ctx is not p1.ctx
        assert p1.controller.ctx is p1.ctx and p2.controller.ctx is p2.ctx     # their components follow
        c2 = p2.ctx
        del p2
        gc.collect()
        p3 = PILCO(data, horizon=2)
        assert p3.ctx is c2                                                    # handed out again
        ctl = LinearController(2, 1, ctx=p1.ctx)
        assert PILCO(data, horizon=2, controller=ctl).ctx is p1.ctx            # a component that lives somewhere decides
        shared = LinearController(2, 1)                                        # ... and so does one shared by two live objects,
        q1 = PILCO(data, horizon=2, controller=shared)                         # whichever of them touches the device first
        q2 = PILCO(data, horizon=2, controller=shared)
        assert q2.ctx is q1.ctx and shared.ctx is q1.ctx
    finally:
        _lib.set_context(saved)


def test_a_shared_reward_does_not_merge_contexts_and_a_sharded_default_context_is_never_pooled():
    """Round-3 advice: (i) a stateless reward object reused by two live PILCO objects must not pull the second one onto the
    first one's context; (ii) an object that adopts a pooled context through a component becomes that context's holder;
    (iii) pooled siblings take over the runtime knobs set on the default context; (iv) a default context that is sharded
    (several ranks / a communicator) is the one context of its process."""
    from helpers.cpu_rollout_context import CpuRolloutContext
    from pilco_amd import _lib
    from pilco_amd.controllers import LinearController
    from pilco_amd.models import PILCO
    from pilco_amd.rewards import ExponentialReward

    class Knobs(CpuRolloutContext):
        def __init__(self, device=None):
            super().__init__()
            self.device, self._settings, self.got = device, {}, {}

        def set_pair_kernel(self, v):
            self._settings["set_pair_kernel"] = (v,)
            self.got["pair"] = v

    saved = _lib._default_ctx
    try:
        d = Knobs()
        d.set_pair_kernel(2)
        _lib.set_context(d)
        rng = np.random.default_rng(1)
        data = (rng.standard_normal((12, 3)), rng.standard_normal((12, 2)))
        rew = ExponentialReward(2)
        p1 = PILCO(data, horizon=2, reward=rew)
        p2 = PILCO(data, horizon=2, reward=rew)
        assert p1.ctx is d and p2.ctx is not d                       # (i)
        assert p2.ctx.got.get("pair") == 2                           # (iii)
        c2 = p2.ctx
        ctl = LinearController(2, 1, ctx=c2)
        del p2
        import gc
        gc.collect()
        p3 = PILCO(data, horizon=2, controller=ctl)                  # adopts c2 through its controller ...
        assert p3.ctx is c2
        p4 = PILCO(data, horizon=2)
        assert p4.ctx is not c2 and p4.ctx is not d                  # (ii) ... so c2 is not handed out a second time
        d.nranks = 2                                                 # (iv)
        assert PILCO(data, horizon=2).ctx is d
    finally:
        _lib.set_context(saved)


def test_an_incomplete_training_objective_is_refused_and_frozen_parameters_carry_no_prior():
    """Round-3 advice: a context sharded by output without a communicator returns NaN for the outputs it does not own; the
    optimiser entry point must raise instead of feeding NaN to L-BFGS-B.  And GPflow's loss holds the log-priors of the
    TRAINABLE parameters only."""
    import pytest
    from helpers.cpu_objective_context import CpuObjectiveContext
    from pilco_amd import training
    from pilco_amd.models import MGPR

    class Sharded(CpuObjectiveContext):
        nranks, has_comm = 2, False   # sharded by output, nobody to exchange with

        def gp_nlml(self, slot, D, E, want_grad=True):
            nlml, grad = super().gp_nlml(slot, D, E, want_grad)
            nlml[1::2] = np.nan
            grad[1::2] = np.nan
            return nlml, grad

    rng = np.random.default_rng(3)
    X, Y = rng.standard_normal((20, 2)), rng.standard_normal((20, 2))
    with pytest.raises(RuntimeError, match="communicator"):
        MGPR((X, Y), ctx=Sharded()).optimize(restarts=0)
    # (round-4 advice) ... and ONLY there: a non-finite objective on an unsharded context is not reported as a sharding problem
    training._require_complete(np.array([np.nan, 1.0]), "MGPR", CpuObjectiveContext())
    with pytest.raises(RuntimeError, match="communicator"):
        training._require_complete(np.array([np.nan, 1.0]), "MGPR", Sharded())
    m = MGPR((X, Y), ctx=CpuObjectiveContext())
    u = training._mgpr_pack(m)
    full, _ = training.mgpr_objective(m, u)
    no_ls, g = training.mgpr_objective(m, u, ls_trainable=[False, True])
    lp_l, _ = training._gamma_logpdf_and_grad(np.asarray(m.lengthscales), 1.1, 0.1)
    np.testing.assert_allclose(no_ls - full, [lp_l[0].sum(), 0.0], atol=1e-12)
    assert np.all(g[:2] == 0.0) and np.all(g[2:4] != 0.0)
