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
