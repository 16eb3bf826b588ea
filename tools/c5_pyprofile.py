"""cProfile of optimize_policy(maxiter=50) at config-5 size (N=225, D=5, E=4, RbfController bf=10, H=40): where the host time of
one policy optimisation goes beside the device rollouts.  Developer tool."""
import os, sys, time, cProfile, pstats
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pilco_amd.controllers import RbfController
from pilco_amd.models import PILCO
rs = np.random.RandomState(0)
X = rs.randn(225, 5) * np.array([0.3, 0.1, 0.5, 0.8, 2.0])
Y = 0.05 * np.stack([np.sin(X @ rs.randn(5)) for _ in range(4)], 1) + 1e-3 * rs.randn(225, 4)
np.random.seed(0)
ctl = RbfController(state_dim=4, control_dim=1, num_basis_functions=10, max_action=3.0)
from pilco_amd.rewards import ExponentialReward
p = PILCO((X, Y), controller=ctl, horizon=40, reward=ExponentialReward(4, t=np.array([0.0, 0.0, 0.0, 0.0])))
for m in p.mgpr.models:
    m.kernel.lengthscales.assign(np.array([0.5, 0.3, 1.0, 1.5, 3.0])); m.kernel.variance.assign(0.01); m.likelihood.variance.assign(1e-5)
p.optimize_policy(maxiter=5)
t0 = time.perf_counter(); p.optimize_policy(maxiter=50); dt = time.perf_counter() - t0
print("optimize_policy(maxiter=50): %.3f s" % dt)
pr = cProfile.Profile(); pr.enable(); p.optimize_policy(maxiter=50); pr.disable()
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(22)
