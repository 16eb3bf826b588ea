"""Secondary measurements for DESIGN.md: factorisation time, config 4 (SMGPR) rollouts/s, R-grad."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pilco_amd import _lib, synthetic
ctx = _lib.Context()
c = synthetic.config_c2()
ctx.gp_set_data(0, c["X"], c["Y"]); ctx.gp_set_hyp(0, c["lengthscales"], c["variance"], c["noise"])
print("exact GP factorisation (N=1000, D=10, E=10): %.2f ms" % ctx.factorize_timed(0, 5))
pol = dict(kind=_lib.POLICY_NONE, state_dim=10, control_dim=0)
rw = [dict(kind=_lib.REWARD_EXPONENTIAL, coef=1.0, W=np.eye(10), t=np.zeros(10))]
c4 = synthetic.config_c4()
ctx.gp_set_data(0, c4["X"], c4["Y"]); ctx.gp_set_hyp(0, c4["lengthscales"], c4["variance"], c4["noise"]); ctx.gp_set_inducing(0, c4["Z"])
print("FITC factorisation (M=200, N=5000, D=10, E=10): %.2f ms" % ctx.factorize_timed(0, 5))
ctx.rollout_timed(pol, rw, c4["m0"], c4["S0"], 40, 2, time_pair=False)
r = ctx.rollout_timed(pol, rw, c4["m0"], c4["S0"], 40, 20, time_pair=False)
print("config 4 SMGPR rollout H=40: %.3f ms -> %.0f rollouts/s" % (r["ms_total"] / 20, 20e3 / r["ms_total"]))
