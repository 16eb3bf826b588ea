"""The oracle pinned to the reference's OWN SOURCE, executed.

One test per reference test file (tests/test_predictions.py, test_sparse_predictions.py, test_cascade.py,
test_controllers.py, test_rewards.py under /root/reference).  Each one
  (1) imports the reference's unmodified modules through oracle/ref_exec.py (tensorflow / gpflow stand-ins of
      oracle/refshim.py) and runs the procedure of the reference's test,
  (2) repeats the reference's own assertion -- against the MATLAB routine (transliteration, Octave being absent) at
      the reference's own tolerance,
  (3) checks that the committed fixture tests/golden/*.npz (which travels to the GPU box) holds exactly these
      executed outputs, and
  (4) checks the NumPy restatement oracle/tf_path.py (the thing the GPU parity tests and bench.py's CPU baseline
      call at sizes without fixtures) against the executed reference.
/root/reference exists only in the build container: (1)-(3) skip elsewhere; the fixture-vs-restatement checks of
tests/test_oracle.py always run."""
import os

import numpy as np
import pytest

from oracle import matlab_path as mp
from oracle import ref_exec
from oracle import tf_path as tp

pytestmark = pytest.mark.skipif(not ref_exec.available(), reason="/root/reference is not present on this box")
n_ = ref_exec.to_np
TIGHT = 1e-9          # executed reference vs committed fixture / vs the NumPy restatement (well-conditioned cases)


@pytest.fixture(scope="module")
def R():
    return ref_exec.load()


def _g(golden_dir, name):
    g = np.load(os.path.join(golden_dir, name))
    assert str(g["provenance"]).startswith("reference source executed")
    return g


def _set_hyp(models, g):
    for i, m in enumerate(models):
        m.kernel.lengthscales.assign(g["lengthscales"][i])
        m.kernel.variance.assign(g["variance"][i])
        m.likelihood.variance.assign(g["noise"][i])


def test_reference_modules_come_from_the_reference_tree(R):
    import inspect
    for mod in (R.pilco.models.mgpr, R.pilco.models.smgpr, R.pilco.models.pilco, R.controllers, R.rewards):
        assert inspect.getsourcefile(mod).startswith(ref_exec.REFERENCE_ROOT)
    # the stand-ins do not stay importable by unrelated code
    import sys
    assert "tensorflow" not in sys.modules and "gpflow" not in sys.modules


def test_predictions(R, golden_dir):
    """tests/test_predictions.py:13-63 (fixed hyper-parameters), incl. the set_data stale-cache step."""
    g = _g(golden_dir, "predictions.npz")
    mgpr = R.MGPR((g["X_first"], g["Y"]))
    _set_hyp(mgpr.models, g)
    M1, S1, V1 = [n_(x) for x in mgpr.predict_on_noisy_inputs(g["m"], g["s"])]
    mgpr.set_data((g["X"], g["Y"]))
    M, S, V = [n_(x) for x in mgpr.predict_on_noisy_inputs(g["m"], g["s"])]
    assert not np.allclose(M1, M)
    # (2) the reference's assertion, at its tolerance (test_predictions.py:58-63)
    M_mat, S_mat, V_mat = mp.gp0(g["X"], g["Y"], g["hyp"], g["m"].T, g["s"])
    assert M.shape == M_mat.T.shape and S.shape == S_mat.shape and V.shape == V_mat.shape
    for a, b in ((M, M_mat.T), (S, S_mat), (V, V_mat)):
        np.testing.assert_allclose(a, b, rtol=1e-4)
    # (3) fixture == executed reference
    for a, k in ((M, "M"), (S, "S"), (V, "V"), (M1, "M_first"), (S1, "S_first"), (V1, "V_first")):
        np.testing.assert_allclose(a, g[k], rtol=TIGHT)
    # (4) restatement == executed reference
    iK, beta = tp.calculate_factorizations(g["X"], g["Y"], g["lengthscales"], g["variance"], g["noise"])
    iKr, betar = [n_(x) for x in mgpr.calculate_factorizations()]
    np.testing.assert_allclose(beta, betar, rtol=1e-7)
    np.testing.assert_allclose(iK, iKr, rtol=1e-6, atol=1e-6 * np.abs(iKr).max())
    for fn in (tp.predict_given_factorizations, tp.predict_given_factorizations_pairs):
        Mt, St, Vt = fn(g["X"], g["lengthscales"], g["variance"], g["m"], g["s"], iK, beta)
        for a, b in ((Mt, M), (St, S), (Vt, V)):
            np.testing.assert_allclose(a, b, rtol=TIGHT)
    # the K the reference builds (mgpr.py:154-157 -> gpflow SquaredExponential, restated from recollection) vs tf_path
    np.testing.assert_allclose(n_(mgpr.K(g["X"])), tp.se_ard_K(g["X"], None, g["lengthscales"], g["variance"]), rtol=1e-12)


def test_predictions_at_the_noise_floor_conditioning(golden_dir):
    """The literal reference procedure (MGPR.optimize() drives the noise to GPflow's 1e-6 floor): every float64
    evaluation -- the executed reference included -- is several 1e-6 away from the 40-digit truth in S, and the
    reference is 1.4e-5 from its own MATLAB oracle (its test allows 1e-4)."""
    g = _g(golden_dir, "predictions_lownoise.npz")
    rel = lambda a, b: np.max(np.abs(a - b) / np.abs(b))
    assert np.all(g["noise"] < 1.1e-6)
    assert rel(g["S"], g["S_mp"]) < 1e-5 and rel(g["S_matlab"], g["S_mp"]) < 2e-5
    assert rel(g["S"], g["S_matlab"]) < 1e-4                       # the reference's own assertion
    assert rel(g["M"], g["M_mp"]) < 1e-8 and rel(g["V"], g["V_mp"]) < 1e-8
    iK, beta = tp.calculate_factorizations(g["X"], g["Y"], g["lengthscales"], g["variance"], g["noise"])
    M, S, V = tp.predict_given_factorizations(g["X"], g["lengthscales"], g["variance"], g["m"], g["s"], iK, beta)
    assert rel(S, g["S_mp"]) < 1e-5 and rel(M, g["M_mp"]) < 1e-8 and rel(V, g["V_mp"]) < 1e-8


def test_sparse_predictions(R, golden_dir):
    """tests/test_sparse_predictions.py:12-57."""
    g = _g(golden_dir, "sparse_predictions.npz")
    np.random.seed(11)
    sm = R.SMGPR((g["X"], g["Y"]), num_induced_points=30)
    _set_hyp(sm.models, g)
    sm.models[0].inducing_variable.Z.assign(g["Z"])
    M, S, V = [n_(x) for x in sm.predict_on_noisy_inputs(g["m"], g["s"])]
    M_mat, S_mat, V_mat = mp.gp1(g["X"], g["Y"], g["hyp"], n_(sm.Z), g["m"].T, g["s"])
    for a, b in ((M, M_mat.T), (S, S_mat), (V, V_mat)):
        np.testing.assert_allclose(a, b, rtol=1e-4)
    for a, k in ((M, "M"), (S, "S"), (V, "V")):
        np.testing.assert_allclose(a, g[k], rtol=1e-8)
    iK, beta = tp.fitc_factorizations(g["X"], g["Y"], g["Z"], g["lengthscales"], g["variance"], g["noise"])
    np.testing.assert_allclose(beta, g["beta"], rtol=1e-6)
    Mt, St, Vt = tp.predict_given_factorizations(g["Z"], g["lengthscales"], g["variance"], g["m"], g["s"], iK, beta)
    for a, b in ((Mt, M), (St, S), (Vt, V)):
        np.testing.assert_allclose(a, b, rtol=1e-7)


def test_cascade(R, golden_dir):
    """tests/test_cascade.py:17-78: PILCO.predict, H = 10, LinearController with max_action [[10]]; every
    intermediate state and the running reward are pinned too (the reference compares the final state only)."""
    g = _g(golden_dir, "cascade.npz")
    np.random.seed(5)
    pilco = R.PILCO((g["X"], g["Y"]))
    pilco.controller.max_action = g["max_action"]
    _set_hyp(pilco.mgpr.models, g)
    pilco.controller.W.assign(g["W"])
    pilco.controller.b.assign(g["b"])
    H = int(g["horizon"])
    M, S, reward = [n_(x) for x in pilco.predict(g["m"], g["s"], H)]
    M_mat, S_mat = mp.pred(g["m"].T, g["s"], H, g["X"], g["Y"], g["hyp"], g["W"], g["b"].T, g["max_action"])
    np.testing.assert_allclose(M[0], M_mat[:, -1].T, rtol=2e-4)       # test_cascade.py:77-78
    np.testing.assert_allclose(S, S_mat[:, :, -1], rtol=2e-4)
    np.testing.assert_allclose(M[0], g["M_traj"][:, -1], rtol=TIGHT)
    np.testing.assert_allclose(S, g["S_traj"][:, :, -1], rtol=TIGHT)
    np.testing.assert_allclose(reward.ravel()[0], g["R_traj"][-1], rtol=TIGHT)
    # restatement: every step, reward included; cached and re-factorised
    model = tp.Model(g["X"], g["Y"], g["lengthscales"], g["variance"], g["noise"])
    ctrl = lambda m, s: tp.linear_controller(m, s, g["W"], g["b"], g["max_action"])
    for n in range(H + 1):
        Mt, St, Rt = tp.predict(model, ctrl, tp.exponential_reward, g["m"], g["s"], n, cache=True)
        np.testing.assert_allclose(Mt[0], g["M_traj"][:, n], rtol=1e-8)
        np.testing.assert_allclose(St, g["S_traj"][:, :, n], rtol=1e-7)
        np.testing.assert_allclose(Rt[0, 0], g["R_traj"][n], rtol=1e-8, atol=1e-300)
    # n = 0 returns the inputs and zero reward (examples/safe_cars_run.py:110)
    M0, S0, R0 = pilco.predict(g["m"], g["s"], 0)
    assert np.array_equal(n_(M0), g["m"]) and np.array_equal(n_(S0), g["s"]) and float(n_(R0).ravel()[0]) == 0.0
    # compute_action evaluates at zero covariance (pilco.py:115-116)
    u = n_(pilco.compute_action(g["m"]))
    ut = tp.linear_controller(g["m"], np.zeros((2, 2)), g["W"], g["b"], g["max_action"])[0]
    np.testing.assert_allclose(u, ut, rtol=1e-12)


def test_cascade_trained_fixture(golden_dir):
    """The literal procedure (optimize_models(restarts=5), optimize_policy(restarts=5), seed 0) was executed once by
    oracle/gen_golden.py (2 min); here its stored outputs are checked against the MATLAB routine at the reference's
    tolerance and against the restatement."""
    g = _g(golden_dir, "cascade_trained.npz")
    H = int(g["horizon"])
    np.testing.assert_allclose(g["M_traj"][:, -1], g["M_traj_matlab"][:, -1], rtol=2e-4)
    np.testing.assert_allclose(g["S_traj"][:, :, -1], g["S_traj_matlab"][:, :, -1], rtol=2e-4)
    model = tp.Model(g["X"], g["Y"], g["lengthscales"], g["variance"], g["noise"])
    ctrl = lambda m, s: tp.linear_controller(m, s, g["W"], g["b"], g["max_action"])
    Mt, St, Rt = tp.predict(model, ctrl, tp.exponential_reward, g["m"], g["s"], H, cache=True)
    np.testing.assert_allclose(Mt[0], g["M_traj"][:, -1], rtol=1e-5)
    np.testing.assert_allclose(St, g["S_traj"][:, :, -1], rtol=1e-5)
    np.testing.assert_allclose(Rt[0, 0], g["R_traj"][-1], rtol=1e-6)


def test_controllers(R, golden_dir):
    """tests/test_controllers.py:13-113: RbfController vs gp2.m, LinearController vs conlin.m, squash_sin vs gSin.m."""
    g = _g(golden_dir, "rbf_controller.npz")
    rbf = R.controllers.RbfController(3, 2, 100)
    rbf.set_data((g["X"], g["Y"]))
    for i, mdl in enumerate(rbf.models):
        mdl.kernel.lengthscales.assign(g["lengthscales"][i])
    M, S, V = [n_(x) for x in rbf.compute_action(g["m"], g["s"], squash=False)]
    lengthscales = np.stack([n_(m.kernel.lengthscales) for m in rbf.models])
    variance = np.stack([n_(m.kernel.variance) for m in rbf.models]).reshape(-1)
    noise = np.stack([n_(m.likelihood.variance) for m in rbf.models]).reshape(-1)
    np.testing.assert_allclose(variance, 1.0, rtol=1e-12)
    np.testing.assert_allclose(noise, 1e-4, rtol=1e-9)
    M_mat, S_mat, V_mat = mp.gp2(g["X"], g["Y"], mp.hyp_from(lengthscales, variance, noise), g["m"].T, g["s"])
    for a, b in ((M, M_mat.T), (S, S_mat), (V, V_mat)):
        np.testing.assert_allclose(a, b, rtol=1e-4)
    for a, k in ((M, "M"), (S, "S"), (V, "V")):
        np.testing.assert_allclose(a, g[k], rtol=1e-8)
    Mt, St, Vt = tp.rbf_controller(g["m"], g["s"], g["X"], g["Y"], g["lengthscales"], squash=False)
    for a, b in ((Mt, M), (St, S), (Vt, V)):
        np.testing.assert_allclose(a, b, rtol=1e-7)
    Mq, Sq, Vq = [n_(x) for x in rbf.compute_action(g["m"], g["s"], squash=True)]
    Mt, St, Vt = tp.rbf_controller(g["m"], g["s"], g["X"], g["Y"], g["lengthscales"], squash=True)
    for a, b in ((Mt, Mq), (St, Sq), (Vt, Vq)):
        np.testing.assert_allclose(a, b, rtol=1e-7)

    g = _g(golden_dir, "linear_controller.npz")
    lin = R.controllers.LinearController(3, 2)
    lin.W.assign(g["W"])
    lin.b.assign(g["b"])
    M, S, V = [n_(x) for x in lin.compute_action(g["m"], g["s"], squash=False)]
    M_mat, S_mat, V_mat = mp.conlin(g["W"], g["b"].T, g["m"].T, g["s"])
    np.testing.assert_allclose(S, S_mat, rtol=1e-4)
    np.testing.assert_allclose(V, V_mat, rtol=1e-4)
    Mt, St, Vt = tp.linear_controller(g["m"], g["s"], g["W"], g["b"], squash=False)
    for a, b, k in ((Mt, M, "M"), (St, S, "S"), (Vt, V, "V")):
        np.testing.assert_allclose(a, b, rtol=1e-13)
        np.testing.assert_allclose(b, g[k], rtol=1e-13)

    g = _g(golden_dir, "squash.npz")
    M, S, V = [n_(x) for x in R.controllers.squash_sin(g["m"], g["s"], float(g["e"]))]
    M_mat, S_mat, V_mat = mp.gSin(g["m"].T, g["s"], float(g["e"]))
    for a, b in ((M, M_mat.T), (S, S_mat), (V, V_mat)):
        np.testing.assert_allclose(a, b, rtol=1e-4)
    Mt, St, Vt = tp.squash_sin(g["m"], g["s"], float(g["e"]))
    for a, b, k in ((Mt, M, "M"), (St, S, "S"), (Vt, V, "V")):
        np.testing.assert_allclose(a, b, rtol=1e-12)
        np.testing.assert_allclose(b, g[k], rtol=1e-12)


def test_rewards(R, golden_dir):
    """tests/test_rewards.py:13-31 (default rtol 1e-7 there), plus the reward classes the reference leaves untested."""
    g = _g(golden_dir, "reward.npz")
    k = 2
    r = R.rewards.ExponentialReward(k)
    mu, sr = [float(n_(x).ravel()[0]) for x in r.compute_reward(g["m"], g["s"])]
    mu_mat, sr_mat = mp.reward(g["m"].T, g["s"], np.zeros((k, 1)), np.eye(k))
    np.testing.assert_allclose(mu, np.ravel(mu_mat)[0], rtol=1e-7)
    np.testing.assert_allclose(sr, np.ravel(sr_mat)[0], rtol=1e-7)
    np.testing.assert_allclose([mu, sr], [g["muR"], g["sR"]], rtol=1e-12)
    mut, srt = tp.exponential_reward(g["m"], g["s"])
    np.testing.assert_allclose([mut[0, 0], srt[0, 0]], [mu, sr], rtol=1e-10)
    r2 = R.rewards.ExponentialReward(k, W=g["W2"], t=g["t2"])
    mu2, sr2 = [float(n_(x).ravel()[0]) for x in r2.compute_reward(g["m"], g["s"])]
    mut, srt = tp.exponential_reward(g["m"], g["s"], g["W2"], g["t2"])
    np.testing.assert_allclose([mut[0, 0], srt[0, 0]], [mu2, sr2], rtol=1e-10)
    np.testing.assert_allclose([mu2, sr2], [g["muR2"], g["sR2"]], rtol=1e-12)
    rl = R.rewards.LinearReward(k, g["W_lin"])
    rc = R.rewards.CombinedRewards(k, [rl, r], coefs=g["coefs"])
    muc, src = [float(n_(x).ravel()[0]) for x in rc.compute_reward(g["m"], g["s"])]
    mut, srt = tp.combined_rewards(g["m"], g["s"], [lambda m, s: tp.linear_reward(m, s, g["W_lin"]), tp.exponential_reward],
                                   g["coefs"])
    np.testing.assert_allclose([np.ravel(mut)[0], np.ravel(srt)[0]], [muc, src], rtol=1e-10)
    np.testing.assert_allclose([muc, src], [g["muR_comb"], g["sR_comb"]], rtol=1e-12)


def test_policy_gradient_through_the_executed_reference(R, golden_dir):
    """Reverse mode through the reference's own training_loss (pilco.py:47-50; what its optimiser differentiates,
    pilco.py:85-90) agrees with autograd of the torch restatement the GPU gradient tests use (oracle/torch_path.py)
    and with the committed fixture."""
    import torch
    from oracle import torch_path as tq
    g = _g(golden_dir, "policy_gradient.npz")
    H = int(g["H"])
    np.random.seed(3)
    pilco = R.PILCO((g["X"], g["Y"]), horizon=H, reward=R.rewards.ExponentialReward(2, W=g["W_reward"], t=g["t_reward"]),
                    m_init=g["m"], S_init=g["s"])
    pilco.controller.max_action = float(g["max_action"])
    _set_hyp(pilco.mgpr.models, g)
    pilco.controller.W.assign(g["W"])
    pilco.controller.b.assign(g["b"])
    loss = pilco.training_loss()
    gW, gb = torch.autograd.grad(loss.sum(), [pilco.controller.W.unconstrained_variable,
                                               pilco.controller.b.unconstrained_variable])
    np.testing.assert_allclose(-gW.numpy(), g["dreward_dW"], rtol=1e-9)
    np.testing.assert_allclose(-gb.numpy(), g["dreward_db"], rtol=1e-9)
    iK, beta = tp.calculate_factorizations(g["X"], g["Y"], g["lengthscales"], g["variance"], g["noise"])
    Wt = torch.tensor(g["W"], dtype=torch.float64, requires_grad=True)
    bt = torch.tensor(g["b"], dtype=torch.float64, requires_grad=True)
    gp = lambda m, s: tq.predict_given_factorizations(g["X"], g["lengthscales"], g["variance"], m, s, iK, beta)
    ctl = lambda m, s: tq.linear_controller(m, s, Wt, bt, float(g["max_action"]))
    rw = lambda m, s: tq.exponential_reward(m, s, g["W_reward"], g["t_reward"])
    _, _, Rr = tq.predict(gp, ctl, rw, tq.t(g["m"]), tq.t(g["s"]), H)
    Rr.sum().backward()
    np.testing.assert_allclose(Rr.item(), g["reward"], rtol=1e-9)
    np.testing.assert_allclose(Wt.grad.numpy(), g["dreward_dW"], rtol=1e-7)
    np.testing.assert_allclose(bt.grad.numpy(), g["dreward_db"], rtol=1e-7)


def test_sparse_rollout_and_wide_gradient_fixtures_equal_the_executed_reference(R, golden_dir):
    """sparse_rollout.npz (PILCO(num_induced_points) rollout + reverse mode) and policy_gradient_wide.npz (reverse mode at
    D = 18): re-executed here, the committed numbers must come out again."""
    import torch
    g = _g(golden_dir, "sparse_rollout.npz")
    H, Zs = int(g["H"]), g["Z_all"]
    np.random.seed(8)
    p = R.PILCO((g["X"], g["Y"]), num_induced_points=Zs.shape[1], horizon=H, m_init=g["m"], S_init=g["s"])
    _set_hyp(p.mgpr.models, g)
    for i, mdl in enumerate(p.mgpr.models):
        mdl.inducing_variable.Z.assign(Zs[i])
    p.controller.W.assign(g["W"])
    p.controller.b.assign(g["b"])
    p.controller.max_action = g["max_action"]
    Mh, Sh, Rh = p.predict(g["m"], g["s"], H)
    np.testing.assert_allclose(n_(Mh)[0], g["M_traj"][:, -1], rtol=1e-12)
    np.testing.assert_allclose(n_(Sh), g["S_traj"][:, :, -1], rtol=1e-12)
    loss = p.training_loss()
    gW, gb = torch.autograd.grad(loss.sum(), [p.controller.W.unconstrained_variable, p.controller.b.unconstrained_variable])
    np.testing.assert_allclose(-float(loss.detach().sum()), float(g["reward"]), rtol=1e-12)
    np.testing.assert_allclose(-gW.numpy(), g["dreward_dW"], rtol=1e-9)
    np.testing.assert_allclose(-gb.numpy(), g["dreward_db"], rtol=1e-9)
    w = _g(golden_dir, "policy_gradient_wide.npz")
    np.random.seed(6)
    q = R.PILCO((w["X"], w["Y"]), horizon=int(w["H"]), m_init=w["m0"], S_init=w["S0"])
    q.controller.max_action = float(w["max_action"])
    _set_hyp(q.mgpr.models, w)
    q.controller.W.assign(w["W"])
    q.controller.b.assign(w["b"])
    lw = q.training_loss()
    hW, hb = torch.autograd.grad(lw.sum(), [q.controller.W.unconstrained_variable, q.controller.b.unconstrained_variable])
    np.testing.assert_allclose(-float(lw.detach().sum()), float(w["reward"]), rtol=1e-12)
    np.testing.assert_allclose(-hW.numpy(), w["dreward_dW"], rtol=1e-9)
    np.testing.assert_allclose(-hb.numpy(), w["dreward_db"], rtol=1e-9)


def test_config4_fixture_equals_the_executed_reference_and_pins_the_restatement(R, golden_dir):
    """c4_sparse.npz (BASELINE config 4 at its size: M=200, N=5000, D=10, E=10): the reference's SMGPR factorisation
    (smgpr.py:24-45) and three steps of its rollout are executed again here -- the committed numbers must come out -- and
    the NumPy restatement the GPU tests use at sizes without a fixture (oracle/tf_path.fitc_factorizations) is held to the
    executed factors at 1e-9."""
    import torch
    from oracle import tf_path as tp
    from pilco_amd import synthetic
    g = _g(golden_dir, "c4_sparse.npz")
    c = synthetic.config_c4()
    E, M = int(g["E"]), int(g["M"])
    np.random.seed(1)
    ctl = R.controllers.LinearController(E, 0, max_action=1.0)
    p = R.PILCO((c["X"], c["Y"]), num_induced_points=M, horizon=3, controller=ctl, m_init=c["m0"], S_init=c["S0"])
    for i, mdl in enumerate(p.mgpr.models):
        mdl.kernel.lengthscales.assign(c["lengthscales"][i])
        mdl.kernel.variance.assign(c["variance"][i])
        mdl.likelihood.variance.assign(c["noise"][i])
        mdl.inducing_variable.Z.assign(c["Z"])
    with torch.no_grad():
        iK, beta = p.mgpr.calculate_factorizations()
        Mh, Sh, Rh = p.predict(c["m0"], c["S0"], 3)
    iK, beta = n_(iK), n_(beta)
    np.testing.assert_allclose(beta, g["beta"], rtol=1e-9, atol=1e-9 * np.abs(g["beta"]).max())
    P = np.random.RandomState(int(g["probe_seed"])).randn(E, M, 4)
    np.testing.assert_allclose(np.einsum("aij,ajk->aik", iK, P), g["iK_probe"], rtol=1e-8, atol=1e-9 * np.abs(g["iK_probe"]).max())
    np.testing.assert_allclose(n_(Mh)[0], g["M_traj"][:, 3], rtol=1e-10)
    np.testing.assert_allclose(n_(Sh), g["S_traj"][:, :, 3], rtol=1e-9, atol=1e-14)
    np.testing.assert_allclose(float(n_(Rh).ravel()[0]), g["R_traj"][3], rtol=1e-12)
    iKo, betao = tp.fitc_factorizations(c["X"], c["Y"], c["Z"], c["lengthscales"], c["variance"], c["noise"])
    for a in range(E):
        assert np.linalg.norm(iKo[a] - iK[a]) / np.linalg.norm(iK[a]) < 1e-9
        assert np.linalg.norm(betao[a] - beta[a]) / np.linalg.norm(beta[a]) < 1e-9


def test_safe_pilco_extension_executed_and_host_side_risk_terms(golden_dir):
    """safe_pilco_extension/ executed (RbfController + RiskOfCollision, the pairing of examples/safe_cars_run.py:72-86):
    the committed fixture equals the executed output, and the product's host-side risk terms (pilco_amd/safe.py: value
    of mu (1 - prod (1 - risk_t)) and the analytic derivatives that seed the native policy gradient) agree with the
    reference's own RiskOfCollision evaluated -- and differentiated by autograd -- on the same states."""
    import torch
    from oracle import ref_exec
    from pilco_amd.safe import RiskOfCollision
    Rs = ref_exec.load(safe=True)
    g = _g(golden_dir, "safe_pilco_rbf.npz")
    H = int(g["H"])
    np.random.seed(4)
    ctl = Rs.controllers.RbfController(4, 1, g["rbf_X"].shape[0], max_action=float(g["max_action"]))
    ctl.set_data((g["rbf_X"], g["rbf_Y"]))
    ctl.models[0].kernel.lengthscales.assign(g["rbf_lengthscales"][0])
    risk_ref = Rs.rewards_safe.RiskOfCollision(2, g["low"], g["high"])
    p = Rs.safe_pilco.SafePILCO((g["X"], g["Y"]), horizon=H, controller=ctl, reward_add=Rs.rewards.LinearReward(4, g["W_lin"]),
                                reward_mult=risk_ref, mu=float(g["mu"]), m_init=g["m0"], S_init=g["S0"])
    _set_hyp(p.mgpr.models, g)
    M, S, Rt = p.predict(g["m0"], g["S0"], H)
    np.testing.assert_allclose(n_(M), g["M"], rtol=1e-12)
    np.testing.assert_allclose(float(n_(Rt).ravel()[0]), float(g["reward_total"]), rtol=1e-12)
    # the reference's risk term and its autograd derivatives at a few states against the product's closed forms
    risk_mine = RiskOfCollision(2, g["low"], g["high"])
    rs = np.random.RandomState(5)
    for _ in range(4):
        A = 0.3 * rs.randn(4, 4)
        m = 0.3 * rs.randn(1, 4)
        s = A @ A.T + 0.2 * np.eye(4)
        mt = torch.tensor(m, dtype=torch.float64, requires_grad=True)
        st = torch.tensor(s, dtype=torch.float64, requires_grad=True)
        r_t = torch.as_tensor(risk_ref.compute_reward(mt, st)[0])   # (the shim's tensor type is a torch.Tensor subclass)
        gm, gs = torch.autograd.grad(r_t.sum(), [mt, st])
        r, dm, ds = risk_mine.compute_reward_grad(m, s)
        np.testing.assert_allclose(r, float(r_t.detach().sum()), rtol=1e-12)
        np.testing.assert_allclose(dm, gm.numpy().ravel(), rtol=1e-10, atol=1e-14)
        np.testing.assert_allclose(ds, gs.numpy(), rtol=1e-10, atol=1e-14)


def test_reference_side_binding_patches_the_real_reference_classes(R):
    """examples/reference_binding.py against the REAL reference package: patch() must find
    the classes and methods it replaces; the arithmetic itself needs a GPU (tests/test_gpu_parity.py runs it)."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.exists(os.path.join(root, "pilco_amd", "libpilco_hip.so")):
        pytest.skip("libpilco_hip.so not built")
    spec = importlib.util.spec_from_file_location("reference_binding", os.path.join(root, "examples", "reference_binding.py"))
    rb = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(rb)
    MGPR, PILCO = R.pilco.models.MGPR, R.pilco.models.PILCO
    keep = (MGPR.predict_on_noisy_inputs, PILCO.predict)
    try:
        rb.patch(R.pilco)
        assert MGPR.predict_on_noisy_inputs is not keep[0] and PILCO.predict is not keep[1]
        # every attribute the stub reads exists on the reference's objects
        np.random.seed(0)
        p = PILCO((np.random.rand(20, 3), np.random.rand(20, 2)))
        for obj, names in ((p.mgpr, ("data", "lengthscales", "variance", "noise", "num_dims", "num_outputs")),
                           (p, ("state_dim", "control_dim", "controller", "reward")), (p.controller, ("W", "b", "max_action")),
                           (p.reward, ("W", "t"))):
            for nme in names:
                assert hasattr(obj, nme), (type(obj).__name__, nme)
    finally:
        MGPR.predict_on_noisy_inputs, PILCO.predict = keep


def test_product_never_imports_the_oracle_or_the_shim():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    bad = []
    for dp, _, fs in os.walk(os.path.join(root, "pilco_amd")):
        for f in fs:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dp, f)).read()
                for needle in ("import oracle", "from oracle", "refshim", "ref_exec", "import torch", "tensorflow"):
                    if needle in src:
                        bad.append((f, needle))
    assert not bad, bad


def test_randomize_draws_what_the_executed_reference_draws(R):
    """Seeded restarts start where the reference's start: randomize() of both controllers (controllers.py:60-63,123-129) and
    of a GP model (mgpr.py:8-15) take the same values from NumPy's global generator in the same order as the executed
    reference code -- for a two-output RBF policy too, where every model re-draws the shared centres."""
    from pilco_amd.controllers import LinearController, RbfController
    from pilco_amd.models import MGPR
    from pilco_amd.models.mgpr import randomize
    from pilco_amd.params import set_trainable
    n_ = ref_exec.to_np
    for U in (1, 2):
        np.random.seed(3)
        ref = R.controllers.RbfController(state_dim=3, control_dim=U, num_basis_functions=5, max_action=0.7)
        ours = RbfController(state_dim=3, control_dim=U, num_basis_functions=5, max_action=0.7)
        np.random.seed(11); ref.randomize()
        np.random.seed(11); ours.randomize()
        assert np.array_equal(n_(ref.models[0].X), ours.X)
        assert np.array_equal(np.hstack([n_(m.Y) for m in ref.models]), ours.Y)
        np.testing.assert_allclose(np.stack([n_(m.kernel.lengthscales) for m in ref.models]), ours.lengthscales, rtol=1e-15)
    ref, ours = R.controllers.LinearController(3, 2, max_action=1.0), LinearController(3, 2, max_action=1.0)
    np.random.seed(5); ref.randomize()
    np.random.seed(5); ours.randomize()
    assert np.array_equal(n_(ref.W), ours.W.numpy()) and np.array_equal(n_(ref.b), ours.b.numpy())
    rs = np.random.RandomState(0)
    X, Y = rs.randn(6, 3), rs.randn(6, 2)
    mr, mo = R.MGPR((X, Y)), MGPR((X, Y))
    for fixed in (False, True):
        if fixed:
            R.gpflow.set_trainable(mr.models[1].likelihood.variance, False)
            set_trainable(mo.models[1].likelihood.variance, False)
        np.random.seed(9)
        for m in mr.models:
            R.mgpr_module.randomize(m)
        tail_ref = np.random.normal()
        np.random.seed(9)
        for m in mo.models:
            randomize(m)
        assert np.random.normal() == tail_ref
        np.testing.assert_allclose(np.stack([n_(m.kernel.lengthscales) for m in mr.models]), mo.lengthscales, rtol=1e-15)
        np.testing.assert_allclose([float(n_(m.likelihood.variance)) for m in mr.models], mo.noise, rtol=1e-12)


def test_constructors_consume_the_random_draws_the_executed_reference_consumes(R):
    """A script that seeds NumPy and builds its model starts from the reference's initial state: default controller weights
    (controllers.py:40-41), default m_init / S_init (pilco.py:37-41), every sparse output's inducing inputs (smgpr.py:20),
    the RBF policy's centres / targets (controllers.py:84-87) -- and the generator is left where the reference leaves it."""
    from pilco_amd.controllers import RbfController
    from pilco_amd.models import PILCO
    n_ = ref_exec.to_np
    rs = np.random.RandomState(0)
    X, Y = rs.randn(12, 4), rs.randn(12, 3)
    np.random.seed(1); pr = R.PILCO((X, Y)); tail = np.random.normal()
    np.random.seed(1); po = PILCO((X, Y))
    assert np.random.normal() == tail
    assert np.array_equal(n_(pr.controller.W), po.controller.W.numpy()) and np.array_equal(n_(pr.controller.b), po.controller.b.numpy())
    assert np.array_equal(n_(pr.m_init), po.m_init) and np.array_equal(n_(pr.S_init), po.S_init)
    assert (pr.state_dim, pr.control_dim, pr.horizon) == (po.state_dim, po.control_dim, po.horizon)
    np.random.seed(2); pr = R.PILCO((X, Y), num_induced_points=5); tail = np.random.normal()
    np.random.seed(2); po = PILCO((X, Y), num_induced_points=5)
    assert np.random.normal() == tail
    for a, b in zip(pr.mgpr.models, po.mgpr.models):
        assert np.array_equal(n_(a.inducing_variable.Z), b.inducing_variable.Z.numpy())
    np.random.seed(3); cr = R.controllers.RbfController(3, 2, 6, max_action=0.5); tail = np.random.normal()
    np.random.seed(3); co = RbfController(3, 2, 6, max_action=0.5)
    assert np.random.normal() == tail
    assert np.array_equal(n_(cr.models[0].X), co.X) and np.array_equal(np.hstack([n_(m.Y) for m in cr.models]), co.Y)
    np.testing.assert_allclose([float(n_(m.likelihood.variance)) for m in cr.models], co.noise, rtol=1e-12)
    np.testing.assert_allclose([float(n_(m.kernel.variance)) for m in cr.models], co.variance, rtol=1e-12)


def test_every_public_class_and_function_of_the_reference_packages_has_its_counterpart_at_the_same_module_path():
    """pilco/{controllers,rewards,models/*}.py and safe_pilco_extension/*.py: every top-level class / function name is
    importable from the same relative module path under pilco_amd, with every public method the reference class defines
    (inherited ones count)."""
    import ast
    import importlib
    pairs = {"pilco/controllers.py": "pilco_amd.controllers", "pilco/rewards.py": "pilco_amd.rewards",
             "pilco/models/mgpr.py": "pilco_amd.models.mgpr", "pilco/models/smgpr.py": "pilco_amd.models.smgpr",
             "pilco/models/pilco.py": "pilco_amd.models.pilco",
             "safe_pilco_extension/safe_pilco.py": "pilco_amd.safe_pilco_extension.safe_pilco",
             "safe_pilco_extension/rewards_safe.py": "pilco_amd.safe_pilco_extension.rewards_safe"}
    internal = {"FakeGPR"}      # controllers.py:66-78: a container for the policy GP's data, replaced by the device slot
    for rel, modname in pairs.items():
        tree = ast.parse(open(os.path.join(ref_exec.REFERENCE_ROOT, rel)).read())
        mod = importlib.import_module(modname)
        for node in tree.body:
            if isinstance(node, ast.FunctionDef):
                assert callable(getattr(mod, node.name, None)), (modname, node.name)
            elif isinstance(node, ast.ClassDef) and node.name not in internal:
                cls = getattr(mod, node.name, None)
                assert isinstance(cls, type), (modname, node.name)
                for item in node.body:
                    if isinstance(item, ast.FunctionDef) and not item.name.startswith("_"):
                        assert hasattr(cls, item.name), (modname, node.name, item.name)
    import pilco_amd.models as pm
    for name in ("PILCO", "MGPR", "SMGPR"):                      # pilco/models/__init__.py:1-3
        assert isinstance(getattr(pm, name), type)


def test_trainable_parameter_sets_match_the_executed_reference(R):
    """PILCO.trainable_parameters (what the reference's optimize_policy saves and restores, pilco.py:96-110): same shapes
    for a linear policy; for an RBF policy the same set with the per-model target columns (bf, 1) x U held as one (bf, U)."""
    from pilco_amd.controllers import RbfController
    from pilco_amd.models import PILCO
    rs = np.random.RandomState(0)
    X, Y = rs.randn(12, 4), rs.randn(12, 3)
    shapes = lambda ps, f: sorted(tuple(np.shape(f(q))) for q in ps)
    ref, ours = R.PILCO((X, Y)), PILCO((X, Y))
    assert shapes(ref.trainable_parameters, n_) == shapes(ours.trainable_parameters, lambda q: q.numpy())
    ref = R.PILCO((X, Y), controller=R.controllers.RbfController(3, 2, 5))
    ours = PILCO((X, Y), controller=RbfController(3, 2, 5))
    a, b = shapes(ref.trainable_parameters, n_), shapes(ours.trainable_parameters, lambda q: q.numpy())
    assert [s for s in a if s != (5, 1)] == [s for s in b if s != (5, 2)] and a.count((5, 1)) == 2 and b.count((5, 2)) == 1
