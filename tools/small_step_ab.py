"""One-launch step of small models (pilco_set_small_step) against the two-launch step: config 4 (SMGPR M=200, N=5000, H=40),
config-5 size (N=225, D=5, E=4, RbfController bf=10) and a cascade-size model.  Developer tool."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("PILCO_AB_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # (PILCO_AB_ROOT: a directory holding another build's pilco_amd/)
from pilco_amd import _lib, synthetic
def med(fn, n=25):
    fn(); fn(); ts = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); ts.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(ts))
ctx = _lib.get_context()
pol = dict(kind=_lib.POLICY_NONE, state_dim=10, control_dim=0)
rw = [dict(kind=_lib.REWARD_EXPONENTIAL, coef=1.0, W=np.eye(10), t=np.zeros(10))]
c4 = synthetic.config_c4()
ctx.gp_set_data(0, c4["X"], c4["Y"]); ctx.gp_set_hyp(0, c4["lengthscales"], c4["variance"], c4["noise"]); ctx.gp_set_inducing(0, c4["Z"])
ctx.gp_factorize(0)
for on in (1, 0, 1, 0):
    ctx.set_small_step(on)
    ms = med(lambda: ctx.rollout(pol, rw, c4["m0"], c4["S0"], 40))
    print("config 4 (M=200): small_step=%d  %.3f ms per rollout  %.0f rollouts/s  (%.1f us per step)" % (on, ms, 1e3 / ms, ms * 1e3 / 40))
ctx.gp_set_inducing(0, None)
from pilco_amd.controllers import RbfController, LinearController
from pilco_amd.models import PILCO
rs = np.random.RandomState(0)
X = rs.randn(225, 5) * np.array([0.3, 0.1, 0.5, 0.8, 2.0])
Y = 0.05 * np.stack([np.sin(X @ rs.randn(5)) for _ in range(4)], 1) + 1e-3 * rs.randn(225, 4)
np.random.seed(0)
for name, ctl in (("rbf", RbfController(state_dim=4, control_dim=1, num_basis_functions=10, max_action=3.0)), ("linear", LinearController(4, 1, max_action=3.0))):
    p = PILCO((X, Y), controller=ctl, horizon=40)
    for m in p.mgpr.models:
        m.kernel.lengthscales.assign(np.array([0.5, 0.3, 1.0, 1.5, 3.0])); m.kernel.variance.assign(0.01); m.likelihood.variance.assign(1e-5)
    for on in (1, 0):
        p.ctx.set_small_step(on)
        print("config-5 size, %s controller: small_step=%d forward %.3f ms, value+gradient %.3f ms" % (name, on, med(lambda: p.compute_reward()), med(lambda: p.value_and_gradient())))
    p.ctx.set_small_step(1)
