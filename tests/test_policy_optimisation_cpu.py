"""The host side of PILCO.optimize_policy (pilco_amd/training.py, adjoint.py, controllers.py: packing of the trainable set,
softplus transform of the RBF lengthscales with its 1e-3 lower bound, signs, L-BFGS-B options, restart bookkeeping) against
the end points of the EXECUTED reference's optimize_policy (tests/golden/policy_optimisation*.npz), with rollout values and
gradients supplied by a CPU stand-in for the device calls (tests/helpers/cpu_rollout_context.py).  The same fixtures are met on
the GPU with the native reverse sweep (tests/test_gpu_parity.py::test_optimize_policy*_ends_where_the_executed_reference_ends)."""
import os

import numpy as np
import pytest

from helpers.cpu_rollout_context import CpuRolloutContext
from pilco_amd.controllers import LinearController, RbfController
from pilco_amd.models import PILCO
from pilco_amd.rewards import CombinedRewards, ExponentialReward, LinearReward

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _hyp(p, g):
    for i, mdl in enumerate(p.mgpr.models):
        mdl.kernel.lengthscales.assign(g["lengthscales"][i])
        mdl.kernel.variance.assign(g["variance"][i])
        mdl.likelihood.variance.assign(g["noise"][i])


def test_linear_policy_ends_where_the_executed_reference_ends():
    g = np.load(os.path.join(GOLDEN, "policy_optimisation.npz"))
    ctx = CpuRolloutContext()
    E = g["Y"].shape[1]
    ctl = LinearController(E, g["X"].shape[1] - E, max_action=g["max_action"], ctx=ctx)
    p = PILCO((g["X"], g["Y"]), horizon=int(g["H"]), controller=ctl, reward=ExponentialReward(E), m_init=g["m"], S_init=g["s"], ctx=ctx)
    _hyp(p, g)
    ctl.W.assign(g["W"]); ctl.b.assign(g["b"])
    np.testing.assert_allclose(float(p.compute_reward()[0, 0]), float(g["reward_start"]), rtol=1e-9)
    r = p.optimize_policy(maxiter=int(g["maxiter"]), restarts=1, verbose=False)
    np.testing.assert_allclose(r, float(g["reward_end"]), rtol=1e-6)
    np.testing.assert_allclose(ctl.W.numpy(), g["W_end"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(ctl.b.numpy(), g["b_end"], rtol=1e-4, atol=1e-6)
    assert ctx.grad_calls > 0


def test_rbf_policy_ends_where_the_executed_reference_ends():
    g = np.load(os.path.join(GOLDEN, "policy_optimisation_rbf.npz"))
    ctx = CpuRolloutContext()
    ctl = RbfController(state_dim=2, control_dim=1, num_basis_functions=g["rbf_X"].shape[0], max_action=float(g["max_action"]), ctx=ctx)
    ctl.set_data((g["rbf_X"], g["rbf_Y"]))
    ctl.models[0].kernel.lengthscales.assign(g["rbf_lengthscales"][0])
    rew = CombinedRewards(2, [ExponentialReward(2), LinearReward(2, g["W_lin"])], coefs=list(g["coefs"]))
    p = PILCO((g["X"], g["Y"]), horizon=int(g["H"]), controller=ctl, reward=rew, m_init=g["m"], S_init=g["s"], ctx=ctx)
    _hyp(p, g)
    np.testing.assert_allclose(float(p.compute_reward()[0, 0]), float(g["reward_start"]), rtol=1e-9)
    r = p.optimize_policy(maxiter=int(g["maxiter"]), restarts=1, verbose=False)
    np.testing.assert_allclose(r, float(g["reward_end"]), rtol=1e-6)
    np.testing.assert_allclose(ctl.X, g["X_end"], rtol=1e-3, atol=1e-5)
    np.testing.assert_allclose(ctl.Y, g["Y_end"], rtol=1e-3, atol=1e-5)
    np.testing.assert_allclose(np.ravel(ctl.lengthscales), np.ravel(g["ls_end"]), rtol=1e-3)


@pytest.mark.parametrize("lanes", ["1", "0"])
def test_random_restarts_keep_the_controller_the_executed_reference_keeps(lanes, monkeypatch):
    """optimize_policy(restarts=3) (pilco.py:93-110): two seeded controller.randomize() restarts after the first run, the
    best controller by reward restored -- the first run's for the linear policy, a restart's for the RBF policy.  Both ways
    the product runs them: the three L-BFGS-B walks side by side, one batched value-and-gradient call per round
    (training._optimize_policy_lanes, the default), and one after the other as the reference does (PILCO_RESTART_LANES=0)."""
    monkeypatch.setenv("PILCO_RESTART_LANES", lanes)
    r_ = np.load(os.path.join(GOLDEN, "policy_optimisation_restarts.npz"))
    g = np.load(os.path.join(GOLDEN, "policy_optimisation.npz"))
    ctx = CpuRolloutContext()
    E = g["Y"].shape[1]
    ctl = LinearController(E, g["X"].shape[1] - E, max_action=g["max_action"], ctx=ctx)
    p = PILCO((g["X"], g["Y"]), horizon=int(r_["lin_H"]), controller=ctl, reward=ExponentialReward(E), m_init=g["m"], S_init=g["s"], ctx=ctx)
    _hyp(p, g)
    ctl.W.assign(g["W"]); ctl.b.assign(g["b"])
    np.random.seed(int(r_["lin_seed"]))
    r = p.optimize_policy(maxiter=int(r_["lin_maxiter"]), restarts=int(r_["restarts"]), verbose=False)
    np.testing.assert_allclose(r, float(r_["lin_reward_end"]), rtol=1e-6)
    np.testing.assert_allclose(ctl.W.numpy(), r_["lin_W_end"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(float(p.compute_reward()[0, 0]), r, rtol=1e-12)      # the kept controller is the assigned one

    g = np.load(os.path.join(GOLDEN, "policy_optimisation_rbf.npz"))
    ctx = CpuRolloutContext()
    ctl = RbfController(state_dim=2, control_dim=1, num_basis_functions=g["rbf_X"].shape[0], max_action=float(g["max_action"]), ctx=ctx)
    ctl.set_data((g["rbf_X"], g["rbf_Y"]))
    ctl.models[0].kernel.lengthscales.assign(g["rbf_lengthscales"][0])
    rew = CombinedRewards(2, [ExponentialReward(2), LinearReward(2, g["W_lin"])], coefs=list(g["coefs"]))
    p = PILCO((g["X"], g["Y"]), horizon=int(r_["rbf_H"]), controller=ctl, reward=rew, m_init=g["m"], S_init=g["s"], ctx=ctx)
    _hyp(p, g)
    np.random.seed(int(r_["rbf_seed"]))
    r = p.optimize_policy(maxiter=int(r_["rbf_maxiter"]), restarts=int(r_["restarts"]), verbose=False)
    np.testing.assert_allclose(r, float(r_["rbf_reward_end"]), rtol=1e-6)
    assert r > float(g["reward_end"])                                                  # a restart won
    np.testing.assert_allclose(ctl.X, r_["rbf_X_end"], rtol=1e-3, atol=1e-5)
    np.testing.assert_allclose(ctl.Y, r_["rbf_Y_end"], rtol=1e-3, atol=1e-5)
    np.testing.assert_allclose(np.ravel(ctl.lengthscales), np.ravel(r_["rbf_ls_end"]), rtol=1e-3)


def test_safe_pilco_objective_and_its_gradient_reach_the_optimiser_as_in_the_executed_extension():
    """SafePILCO (safe_pilco_extension/safe_pilco.py:29-50): what optimize_policy differentiates must be the TOTAL reward,
    mu (1 - prod (1 - risk_t)) included.  The host side -- predict()'s accumulator over the trajectory, trajectory_objective's
    cotangent seeds, their hand-over to the reverse sweep, signs and packing in policy_loss_and_grad -- against reverse mode
    through the EXECUTED extension (fixtures safe_pilco.npz: linear policy + SingleConstraint; safe_pilco_rbf.npz: RBF policy +
    RiskOfCollision, the pairing of examples/safe_cars_run.py), the rollout itself supplied by the CPU stand-in."""
    from pilco_amd.safe import RiskOfCollision, SafePILCO, SingleConstraint
    from pilco_amd.training import _policy_params, policy_loss_and_grad
    g = np.load(os.path.join(GOLDEN, "safe_pilco.npz"))
    ctx, H = CpuRolloutContext(), int(g["H"])
    ctl = LinearController(2, 1, max_action=g["max_action"], ctx=ctx)
    p = SafePILCO((g["X"], g["Y"]), horizon=H, controller=ctl, reward_add=ExponentialReward(2),
                  reward_mult=SingleConstraint(0, high=float(g["high"]), inside=False), mu=float(g["mu"]), m_init=g["m"], S_init=g["s"], ctx=ctx)
    _hyp(p, g)
    ctl.W.assign(g["W"]); ctl.b.assign(g["b"])
    M, S, R = p.predict(g["m"], g["s"], H)
    np.testing.assert_allclose(M, g["M"], rtol=1e-9)
    np.testing.assert_allclose(float(np.ravel(R)[0]), float(g["reward_total"]), rtol=1e-9)
    get, put = _policy_params(ctl)
    f, grad = policy_loss_and_grad(p, get(), put)
    np.testing.assert_allclose(-f, float(g["reward_total"]), rtol=1e-9)
    np.testing.assert_allclose(-grad[:2].reshape(1, 2), g["dtotal_dW"], rtol=1e-8)
    np.testing.assert_allclose(-grad[2:].reshape(1, 1), g["dtotal_db"], rtol=1e-8)

    g = np.load(os.path.join(GOLDEN, "safe_pilco_rbf.npz"))
    ctx, H = CpuRolloutContext(), int(g["H"])
    ctl = RbfController(state_dim=4, control_dim=1, num_basis_functions=g["rbf_X"].shape[0], max_action=float(g["max_action"]), ctx=ctx)
    ctl.set_data((g["rbf_X"], g["rbf_Y"]))
    ctl.models[0].kernel.lengthscales.assign(g["rbf_lengthscales"][0])
    p = SafePILCO((g["X"], g["Y"]), horizon=H, controller=ctl, reward_add=LinearReward(4, g["W_lin"]),
                  reward_mult=RiskOfCollision(2, g["low"], g["high"]), mu=float(g["mu"]), m_init=g["m0"], S_init=g["S0"], ctx=ctx)
    _hyp(p, g)
    np.testing.assert_allclose(float(np.ravel(p.predict(g["m0"], g["S0"], H)[2])[0]), float(g["reward_total"]), rtol=1e-9)
    get, put = _policy_params(ctl)
    u = get()
    f, grad = policy_loss_and_grad(p, u, put)
    np.testing.assert_allclose(-f, float(g["reward_total"]), rtol=1e-9)
    nX, nY = g["rbf_X"].size, g["rbf_Y"].size
    np.testing.assert_allclose(-grad[:nX].reshape(g["rbf_X"].shape), g["dtotal_dX"], rtol=1e-7, atol=1e-10 * float(np.abs(g["dtotal_dX"]).max()))
    np.testing.assert_allclose(-grad[nX:nX + nY].reshape(g["rbf_Y"].shape), g["dtotal_dY"], rtol=1e-7, atol=1e-10 * float(np.abs(g["dtotal_dY"]).max()))
    # the lengthscale entries of the packed gradient are w.r.t. the unconstrained variable: d ls / du = sigmoid(u)
    dls = -grad[nX + nY:] / (1.0 / (1.0 + np.exp(-u[nX + nY:])))
    np.testing.assert_allclose(dls.reshape(g["dtotal_dls"].shape), g["dtotal_dls"], rtol=1e-7)


def test_constraints_inside_a_combined_reward_as_the_reference_safe_swimmer_script_uses_them():
    """examples/safe_swimmer_run.py:59-78 gives a PLAIN PILCO the reward CombinedRewards([LinearReward, SingleConstraint, ...],
    coefs=[1, -10, ...]): constraint terms the device does not evaluate.  The product evaluates them on the host along the
    rollout's states and hands their derivatives to the reverse sweep as cotangent seeds; value and policy gradient against
    the EXECUTED reference (TF reverse mode through the same reward), then one optimize_policy run ending where its ends."""
    import contextlib
    import io
    import torch
    from oracle import ref_exec
    import pytest
    if not ref_exec.available():
        pytest.skip("/root/reference is not present on this box")
    from pilco_amd.safe import SingleConstraint
    from pilco_amd.training import _policy_params, policy_loss_and_grad
    Rs = ref_exec.load(safe=True)
    n_ = ref_exec.to_np
    g = np.load(os.path.join(GOLDEN, "policy_optimisation.npz"))
    E, H = 2, 5
    Wl = np.array([[0.4], [-0.3]])
    ref = Rs.PILCO((g["X"], g["Y"]), horizon=H, m_init=g["m"], S_init=g["s"],
                   reward=Rs.rewards.CombinedRewards(E, [Rs.rewards.LinearReward(E, Wl), Rs.rewards.ExponentialReward(E),
                                                          Rs.rewards_safe.SingleConstraint(0, low=-0.5, high=0.9, inside=False),
                                                          Rs.rewards_safe.SingleConstraint(1, high=0.4)], coefs=[1.0, 0.5, -3.0, 0.7]))
    for i, mdl in enumerate(ref.mgpr.models):
        mdl.kernel.lengthscales.assign(g["lengthscales"][i]); mdl.kernel.variance.assign(g["variance"][i]); mdl.likelihood.variance.assign(g["noise"][i])
    ref.controller.W.assign(g["W"]); ref.controller.b.assign(g["b"]); ref.controller.max_action = g["max_action"]
    loss = ref.training_loss()
    gW, gb = torch.autograd.grad(loss.sum(), [ref.controller.W.unconstrained_variable, ref.controller.b.unconstrained_variable])

    from helpers.cpu_standin_context import CpuStandInContext
    ctx = CpuStandInContext()
    ctl = LinearController(E, 1, max_action=g["max_action"], ctx=ctx)
    rew = CombinedRewards(E, [LinearReward(E, Wl), ExponentialReward(E), SingleConstraint(0, low=-0.5, high=0.9, inside=False),
                              SingleConstraint(1, high=0.4)], coefs=[1.0, 0.5, -3.0, 0.7])
    rew._ctx = ctx
    p = PILCO((g["X"], g["Y"]), horizon=H, controller=ctl, reward=rew, m_init=g["m"], S_init=g["s"], ctx=ctx)
    _hyp(p, g)
    ctl.W.assign(g["W"]); ctl.b.assign(g["b"])
    assert len(rew.terms()) == 2 and len(rew.host_terms()) == 2
    np.testing.assert_allclose(float(p.compute_reward()[0, 0]), -float(n_(loss).ravel()[0]), rtol=1e-9)
    get, put = _policy_params(ctl)
    f, grad = policy_loss_and_grad(p, get(), put)
    np.testing.assert_allclose(f, float(n_(loss).ravel()[0]), rtol=1e-9)
    np.testing.assert_allclose(grad[:2].reshape(1, 2), gW.numpy(), rtol=1e-7)
    np.testing.assert_allclose(grad[2:].reshape(1, 1), gb.numpy(), rtol=1e-7)
    mu_r, var_r = ref.reward.compute_reward(g["m"], g["s"])            # CombinedRewards.compute_reward itself (rewards.py:73-81)
    mu_o, var_o = rew.compute_reward(g["m"], g["s"])
    np.testing.assert_allclose(np.ravel(mu_o), np.ravel(n_(mu_r)), rtol=1e-9)
    np.testing.assert_allclose(np.ravel(var_o), np.ravel(n_(var_r)), rtol=1e-7)
    with contextlib.redirect_stdout(io.StringIO()):
        ref.optimize_policy(maxiter=6, restarts=1)
    r = p.optimize_policy(maxiter=6, restarts=1, verbose=False)
    np.testing.assert_allclose(r, float(n_(ref.compute_reward()).ravel()[0]), rtol=1e-6)
    np.testing.assert_allclose(ctl.W.numpy(), n_(ref.controller.W), rtol=1e-3, atol=1e-6)


def test_objective_function_reward_takes_the_analytic_path():
    """rewards_safe.ObjectiveFunction(reward_f, risk_f, mu) (rewards_safe.py:63-73; the reference's version cannot be
    constructed: it uses a Parameter it never imports) as the reward of a plain PILCO: reward part on the device, risk part as a
    host reward term.  Its policy gradient must equal a central difference of training_loss."""
    from helpers.cpu_standin_context import CpuStandInContext
    from pilco_amd.safe import ObjectiveFunction, SingleConstraint
    from pilco_amd.training import _policy_params, policy_loss_and_grad
    g = np.load(os.path.join(GOLDEN, "policy_optimisation.npz"))
    ctx = CpuStandInContext()
    ctl = LinearController(2, 1, max_action=g["max_action"], ctx=ctx)
    rew = ObjectiveFunction(ExponentialReward(2), SingleConstraint(0, high=0.6), mu=1.7)
    p = PILCO((g["X"], g["Y"]), horizon=4, controller=ctl, reward=rew, m_init=g["m"], S_init=g["s"], ctx=ctx)
    _hyp(p, g)
    ctl.W.assign(g["W"]); ctl.b.assign(g["b"])
    get, put = _policy_params(ctl)
    u = get()
    f, grad = policy_loss_and_grad(p, u, put)
    assert ctx.grad_calls == 1                              # the analytic (seeded) path, not 2n+1 rollouts
    for i in range(u.size):
        h = 1e-4
        up, um = u.copy(), u.copy()
        up[i] += h; um[i] -= h
        put(up); fp = float(p.training_loss()[0, 0])
        put(um); fm = float(p.training_loss()[0, 0])
        np.testing.assert_allclose(grad[i], (fp - fm) / (2 * h), rtol=3e-6, atol=1e-9)
    put(u)
    np.testing.assert_allclose(f, float(p.training_loss()[0, 0]), rtol=1e-12)


def test_safe_pilco_restarts_as_lanes_walk_the_walks_of_the_sequential_loop(monkeypatch):
    """SafePILCO.optimize_policy(restarts=2) (examples/safe_cars_run.py:102): with the restarts as lanes every lane's objective
    is the TOTAL reward of ITS trajectory (per-lane cotangent seeds into the batched reverse sweep, training._optimize_policy_lanes)
    -- the two L-BFGS-B walks, the kept controller and its reward are those of the sequential loop (PILCO_RESTART_LANES=0).
    Host logic only: the stand-in answers the batched call lane by lane."""
    from pilco_amd.safe import SafePILCO, SingleConstraint
    from pilco_amd.training import _policy_params, _restart_lanes_apply
    g = np.load(os.path.join(GOLDEN, "safe_pilco.npz"))
    H, ends = int(g["H"]), {}
    for lanes in ("0", "1"):
        monkeypatch.setenv("PILCO_RESTART_LANES", lanes)
        ctx = CpuRolloutContext()
        ctl = LinearController(2, 1, max_action=g["max_action"], ctx=ctx)
        p = SafePILCO((g["X"], g["Y"]), horizon=H, controller=ctl, reward_add=ExponentialReward(2),
                      reward_mult=SingleConstraint(0, high=float(g["high"]), inside=False), mu=float(g["mu"]), m_init=g["m"], S_init=g["s"], ctx=ctx)
        _hyp(p, g)
        ctl.W.assign(g["W"]); ctl.b.assign(g["b"])
        assert _restart_lanes_apply(p) == ("seeded" if lanes == "1" else False)
        np.random.seed(4)
        r = p.optimize_policy(maxiter=6, restarts=2, verbose=False)
        ends[lanes] = (r, _policy_params(ctl)[0]())
    np.testing.assert_allclose(ends["1"][0], ends["0"][0], rtol=1e-12)
    np.testing.assert_allclose(ends["1"][1], ends["0"][1], rtol=1e-10, atol=1e-12)
