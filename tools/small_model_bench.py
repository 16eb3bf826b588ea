"""Where one policy-objective evaluation goes at config-5 size (N=225, D=5, E=4, RbfController bf=10, H=40)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pilco_amd.controllers import RbfController
from pilco_amd.models import PILCO
N = int(os.environ.get("N", 225))
rs = np.random.RandomState(0)
X = rs.randn(N, 5) * np.array([0.3, 0.1, 0.5, 0.8, 2.0])
Y = 0.05 * np.stack([np.sin(X @ rs.randn(5)) for _ in range(4)], 1) + 1e-3 * rs.randn(N, 4)
np.random.seed(0)
ctl = RbfController(state_dim=4, control_dim=1, num_basis_functions=10, max_action=3.0)
p = PILCO((X, Y), controller=ctl, horizon=40)
for m in p.mgpr.models:
    m.kernel.lengthscales.assign(np.array([0.5, 0.3, 1.0, 1.5, 3.0])); m.kernel.variance.assign(0.01); m.likelihood.variance.assign(1e-5)
def med(fn, n=15):
    fn(); fn(); ts = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); ts.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(ts))
print("N=%d: forward rollout %.3f ms, value+gradient %.3f ms" % (N, med(lambda: p.compute_reward()), med(lambda: p.value_and_gradient())))
