"""Host-side pieces of the Safe-PILCO extension mirror (pilco_amd/safe.py): the risk terms' analytic derivatives, which
seed the native policy gradient (pilco_rollout_grad_seeded), against central differences of the reference's formulas
(rewards_safe.py:13-61), and the trajectory objective mu (1 - prod (1 - risk_t)) against its own finite differences."""
import numpy as np

from pilco_amd.safe import RiskOfCollision, SafePILCO, SingleConstraint


def _state(rs, E=4):
    A = rs.randn(E, E) * 0.3
    return rs.randn(1, E) * 0.3, A @ A.T + 0.2 * np.eye(E)


def test_risk_term_derivatives_match_central_differences():
    rs = np.random.RandomState(0)
    for obj in (RiskOfCollision(4, [-0.5, -0.3], [0.7, 0.9]), SingleConstraint(1, high=0.4, inside=False),
                SingleConstraint(0, low=-0.2), SingleConstraint(2, high=0.5, low=-0.5, inside=True)):
        m, s = _state(rs)
        r, dm, ds = obj.compute_reward_grad(m, s)
        assert abs(r - float(obj.compute_reward(m, s)[0])) < 1e-14
        assert np.count_nonzero(ds - np.diag(np.diag(ds))) == 0      # the formulas see only diagonal entries of s
        h = 1e-6
        for k in range(4):
            mp, mm = m.copy(), m.copy()
            mp[0, k] += h
            mm[0, k] -= h
            fd = (float(obj.compute_reward(mp, s)[0]) - float(obj.compute_reward(mm, s)[0])) / (2 * h)
            assert abs(fd - dm[k]) < 1e-7
            sp, sm = s.copy(), s.copy()
            sp[k, k] += h
            sm[k, k] -= h
            fd = (float(obj.compute_reward(m, sp)[0]) - float(obj.compute_reward(m, sm)[0])) / (2 * h)
            assert abs(fd - ds[k, k]) < 1e-7


def test_trajectory_objective_value_and_seeds():
    rs = np.random.RandomState(1)
    E, H = 4, 5
    obj = SafePILCO.__new__(SafePILCO)          # no device: only the host-side objective is exercised
    obj.mu, obj.reward_mult, obj.state_dim = 3.0, RiskOfCollision(E, [-0.5, -0.3], [0.7, 0.9]), E
    traj = np.zeros((H + 1, E + E * E))
    for t in range(H + 1):
        m, s = _state(rs, E)
        traj[t, :E], traj[t, E:] = m.ravel(), s.ravel()

    def value(tr):
        mult = 1.0
        for t in range(H):
            mult *= 1.0 - float(obj.reward_mult.compute_reward(tr[t, :E].reshape(1, E), tr[t, E:].reshape(E, E))[0])
        return obj.mu * (1.0 - mult)

    v, seeds = obj.trajectory_objective(traj)
    assert abs(v - value(traj)) < 1e-14 and np.all(seeds[H] == 0.0)
    h = 1e-6
    for t, k in ((0, 0), (2, 2), (4, E + 0), (3, E + 2 * E + 2)):       # m_0, m_2 and the (0,0), (2,2) entries of s
        tp, tm = traj.copy(), traj.copy()
        tp[t, k] += h
        tm[t, k] -= h
        assert abs((value(tp) - value(tm)) / (2 * h) - seeds[t, k]) < 1e-6


def test_parameter_value_and_coefficient_updates_as_the_reference_examples_write_them():
    """examples/safe_swimmer_run.py:115-127: `R.coefs.assign(R.coefs.value() * [...])` on a CombinedRewards; the new
    coefficients must reach the reward terms handed to the device."""
    import numpy as np
    from pilco_amd.rewards import CombinedRewards, ExponentialReward, LinearReward
    R = CombinedRewards(2, [ExponentialReward(2), LinearReward(2, np.array([1.0, 0.0]))], coefs=[1.0, -10.0])
    R.coefs.assign(R.coefs.value() * [1.0, 0.75])
    assert np.array_equal(R.coefs.numpy(), [1.0, -7.5]) and np.array_equal(R.coefs.read_value().numpy(), [1.0, -7.5])
    assert [t["coef"] for t in R.terms()] == [1.0, -7.5]
