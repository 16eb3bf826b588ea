"""The example loops (examples/inverted_pendulum.py = BASELINE config 5 as the reference writes it, examples/safe_cars.py =
the reference's Safe-PILCO loop) driven through the product's Python layer on the CPU at reduced size, every device call
answered by the oracle stand-in (tests/helpers/cpu_standin_context.py).  What this holds without a GPU: the host logic of a
whole learning loop -- model fits, policy optimisation with an RBF policy, compute_action on the plant, set_data with a
growing data set, SafePILCO's risk bookkeeping and mu adaptation -- runs and behaves (the reward goes up, the risk is a
probability, data accumulate).  On the GPU the same loops are timed by bench.py and asserted by tests/test_gpu_parity.py."""
import importlib.util
import os

import numpy as np
import pytest

from helpers.cpu_standin_context import CpuStandInContext
from pilco_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _example(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "examples", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture
def standin():
    saved = _lib._default_ctx
    ctx = CpuStandInContext()
    _lib.set_context(ctx)
    try:
        yield ctx
    finally:
        _lib.set_context(saved)


def test_inverted_pendulum_loop_on_the_standin(standin):
    ip = _example("inverted_pendulum")
    out = ip.run_hip(J=2, T=12, iters=2, maxiter=4, rollout_steps=8, seed=0, verbose=False)
    it = out["iterations"]
    assert len(it) == 2 and 2 <= it[0]["N"] <= 24 and it[1]["N"] == it[0]["N"] + it[0]["steps_balanced"]      # random episodes end when the pole falls
    assert all(np.isfinite(s["predicted_reward"]) and 0.0 < s["predicted_reward"] <= 40.0 for s in it)   # 40 steps of a reward in (0, 1]
    assert standin.grad_calls >= 2


def test_safe_cars_loop_on_the_standin(standin):
    sc = _example("safe_cars")
    out = sc.run(iters=1, seed=0, verbose=False)
    it = out["iterations"][0]
    assert 0.0 <= it["predicted_risk"] <= 1.0 and it["mu"] == -300.0 and np.isfinite(it["predicted_return"])
    assert it["N"] in (125, 150)           # the new rollout joins the data only if the predicted risk is below the threshold


def test_the_standin_itself_answers_like_the_executed_reference(standin):
    """The stand-in is only as good as its answers: rewards (mean and variance, all three kinds), a GP prediction, a policy
    action and a 10-step rollout through it equal the executed reference's fixtures (reward.npz, predictions.npz, cascade.npz)."""
    from pilco_amd.controllers import LinearController
    from pilco_amd.models import MGPR, PILCO
    from pilco_amd.rewards import CombinedRewards, ExponentialReward, LinearReward
    G = os.path.join(ROOT, "tests", "golden")
    g = np.load(os.path.join(G, "reward.npz"))
    E = g["m"].shape[1]
    for rew, km, ks in ((ExponentialReward(E), "muR", "sR"), (ExponentialReward(E, W=g["W2"], t=g["t2"]), "muR2", "sR2"),
                        (LinearReward(E, g["W_lin"]), "muR_lin", "sR_lin"),
                        (CombinedRewards(E, [LinearReward(E, g["W_lin"]), ExponentialReward(E)], coefs=list(g["coefs"])), "muR_comb", "sR_comb")):
        mu, var = rew.compute_reward(g["m"], g["s"])
        np.testing.assert_allclose(np.ravel(mu), np.ravel(g[km]), rtol=1e-9)
        np.testing.assert_allclose(np.ravel(var), np.ravel(g[ks]), rtol=1e-7)
    g = np.load(os.path.join(G, "predictions.npz"))
    m = MGPR((g["X"], g["Y"]))
    for i, mdl in enumerate(m.models):
        mdl.kernel.lengthscales.assign(g["lengthscales"][i]); mdl.kernel.variance.assign(g["variance"][i]); mdl.likelihood.variance.assign(g["noise"][i])
    M, S, V = m.predict_on_noisy_inputs(g["m"], g["s"])
    for got, key in ((M, "M"), (S, "S"), (V, "V")):
        np.testing.assert_allclose(got, g[key], rtol=1e-8)
    g = np.load(os.path.join(G, "cascade.npz"))
    p = PILCO((g["X"], g["Y"]), horizon=int(g["horizon"]), controller=LinearController(2, 1, max_action=g["max_action"]))
    for i, mdl in enumerate(p.mgpr.models):
        mdl.kernel.lengthscales.assign(g["lengthscales"][i]); mdl.kernel.variance.assign(g["variance"][i]); mdl.likelihood.variance.assign(g["noise"][i])
    p.controller.W.assign(g["W"]); p.controller.b.assign(g["b"])
    H = int(g["horizon"])
    MH, SH, R = p.predict(g["m"], g["s"], H)
    np.testing.assert_allclose(np.ravel(MH), g["M_traj"][:, H], rtol=1e-8)
    np.testing.assert_allclose(SH, g["S_traj"][:, :, H], rtol=1e-7)
    np.testing.assert_allclose(float(np.ravel(R)[0]), g["R_traj"][H], rtol=1e-8)
    M1, S1 = p.propagate(g["m"], g["s"])
    np.testing.assert_allclose(np.ravel(M1), g["M_traj"][:, 1], rtol=1e-8)


@pytest.mark.parametrize("policy", ["linear", "rbf"])
def test_one_learning_iteration_ends_where_the_executed_reference_ends(standin, policy):
    """Drop-in equivalence of the optimisation loop's host side, end to end: the same seeded script -- PILCO((X, Y)),
    optimize_models(restarts=1), optimize_policy(maxiter=8, restarts=2), new data, optimize_models again -- run once on the
    reference's own source (executed on the shim) and once on the product's Python layer (device calls answered by the
    stand-in) must end with the same hyper-parameters, the same controller and the same predicted reward: constructor
    draws, restart draws and their order, which fit / controller is kept, transforms, priors, optimiser options all match."""
    from oracle import ref_exec
    if not ref_exec.available():
        pytest.skip("/root/reference is not present on this box")
    from pilco_amd.models import PILCO
    R = ref_exec.load()
    n_ = ref_exec.to_np
    rs = np.random.RandomState(3)
    X = rs.randn(40, 3)
    f = lambda Z: np.stack([0.3 * np.sin(Z[:, 0]) + 0.2 * Z[:, 2], 0.25 * np.cos(Z[:, 1]) * Z[:, 0]], 1)
    Y = f(X) + 0.02 * rs.randn(40, 2)
    X2 = rs.randn(10, 3)
    Y2 = f(X2) + 0.02 * rs.randn(10, 2)

    def script(P, Rbf, to_np):
        np.random.seed(7)
        if policy == "linear":
            p = P((X, Y), horizon=5)                      # default LinearController with random weights (pilco.py:30-33)
        else:
            p = P((X, Y), horizon=5, controller=Rbf(2, 1, 5, max_action=1.5))
        p.optimize_models(restarts=1)
        p.optimize_policy(maxiter=8, restarts=2)
        if policy == "linear":
            W, b = to_np(p.controller.W), to_np(p.controller.b)
        else:                                             # centres / targets / lengthscales of the policy GP
            m0 = p.controller.models[0]
            W = to_np(m0.X) if hasattr(m0, "X") else np.asarray(p.controller.X)
            b = np.concatenate([np.ravel(to_np(m0.Y) if hasattr(m0, "Y") else p.controller.Y), np.ravel(to_np(m0.kernel.lengthscales))])
        out = dict(ls1=np.stack([to_np(m.kernel.lengthscales) for m in p.mgpr.models]),
                   nz1=np.array([float(to_np(m.likelihood.variance)) for m in p.mgpr.models]),
                   W=W, b=b, r=float(np.ravel(to_np(p.compute_reward()))[0]))
        p.mgpr.set_data((np.vstack([X, X2]), np.vstack([Y, Y2])))
        p.optimize_models(restarts=1)
        out.update(ls2=np.stack([to_np(m.kernel.lengthscales) for m in p.mgpr.models]),
                   var2=np.array([float(to_np(m.kernel.variance)) for m in p.mgpr.models]),
                   r2=float(np.ravel(to_np(p.compute_reward()))[0]), tail=np.random.normal())
        return out

    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        ref = script(R.PILCO, R.controllers.RbfController, n_)
        from pilco_amd.controllers import RbfController
        ours = script(PILCO, RbfController, lambda v: np.asarray(v.numpy() if hasattr(v, "numpy") else v))
    assert ours["tail"] == ref["tail"]                      # the same number of draws consumed from NumPy's global generator
    for k, tol in (("ls1", 1e-3), ("nz1", 1e-3), ("W", 1e-3), ("b", 1e-3), ("ls2", 1e-3), ("var2", 1e-3)):
        np.testing.assert_allclose(ours[k], ref[k], rtol=tol, atol=1e-6, err_msg=k)
    np.testing.assert_allclose([ours["r"], ours["r2"]], [ref["r"], ref["r2"]], rtol=1e-6)


def test_one_safe_pilco_iteration_ends_where_the_executed_extension_ends(standin):
    """The same for SafePILCO (safe_pilco_extension/safe_pilco.py): optimize_models, then optimize_policy on the TOTAL
    objective (additive reward + mu (1 - prod (1 - risk_t))) -- TF reverse mode through predict() in the reference, cotangent
    seeds into the reverse sweep in the product -- must end at the same controller and the same total reward."""
    import contextlib
    import io
    from oracle import ref_exec
    if not ref_exec.available():
        pytest.skip("/root/reference is not present on this box")
    from pilco_amd.rewards import ExponentialReward
    from pilco_amd.safe import SafePILCO, SingleConstraint
    Rs = ref_exec.load(safe=True)
    n_ = ref_exec.to_np
    rs = np.random.RandomState(4)
    X = rs.randn(36, 3)
    Y = np.stack([0.3 * np.sin(X[:, 0]) + 0.2 * X[:, 2], 0.25 * np.cos(X[:, 1]) * X[:, 0]], 1) + 0.02 * rs.randn(36, 2)
    m0, S0 = np.array([[0.3, -0.1]]), 0.05 * np.eye(2)

    def script(make, to_np):
        np.random.seed(11)
        p = make()
        p.optimize_models(restarts=1)
        r0 = float(np.ravel(to_np(p.compute_reward()))[0])
        p.optimize_policy(maxiter=6, restarts=1)
        return dict(W=to_np(p.controller.W), b=to_np(p.controller.b), r0=r0, r=float(np.ravel(to_np(p.compute_reward()))[0]),
                    tail=np.random.normal())

    with contextlib.redirect_stdout(io.StringIO()):
        ref = script(lambda: Rs.safe_pilco.SafePILCO((X, Y), horizon=4, reward_add=Rs.rewards.ExponentialReward(2),
                                                     reward_mult=Rs.rewards_safe.SingleConstraint(0, high=0.8, inside=False),
                                                     mu=-2.0, m_init=m0, S_init=S0), n_)
        ours = script(lambda: SafePILCO((X, Y), horizon=4, reward_add=ExponentialReward(2),
                                        reward_mult=SingleConstraint(0, high=0.8, inside=False), mu=-2.0, m_init=m0, S_init=S0),
                      lambda v: np.asarray(v.numpy() if hasattr(v, "numpy") else v))
    assert ours["tail"] == ref["tail"]
    np.testing.assert_allclose([ours["r0"], ours["r"]], [ref["r0"], ref["r"]], rtol=1e-6)
    assert ref["r"] > ref["r0"]
    np.testing.assert_allclose(ours["W"], ref["W"], rtol=1e-3, atol=1e-6)
    np.testing.assert_allclose(ours["b"], ref["b"], rtol=1e-3, atol=1e-6)
