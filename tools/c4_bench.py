"""config 4 (SMGPR M=200, N=5000) rollout timing, for stream-K wave-count experiments (PILCO_SK_WAVES)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pilco_amd import _lib, synthetic
ctx = _lib.Context()
pol = dict(kind=_lib.POLICY_NONE, state_dim=10, control_dim=0)
rw = [dict(kind=_lib.REWARD_EXPONENTIAL, coef=1.0, W=np.eye(10), t=np.zeros(10))]
c4 = synthetic.config_c4()
ctx.gp_set_data(0, c4["X"], c4["Y"]); ctx.gp_set_hyp(0, c4["lengthscales"], c4["variance"], c4["noise"]); ctx.gp_set_inducing(0, c4["Z"])
ctx.gp_factorize(0)
ctx.rollout_timed(pol, rw, c4["m0"], c4["S0"], 40, 3, time_pair=False)
r = ctx.rollout_timed(pol, rw, c4["m0"], c4["S0"], 40, 30, time_pair=False)
p = ctx.rollout_timed(pol, rw, c4["m0"], c4["S0"], 40, 1, time_pair=True)
print("SK_WAVES=%s config 4 rollout H=40: %.3f ms -> %.0f rollouts/s; pair kernel %.1f us" % (os.environ.get("PILCO_SK_WAVES", "auto"), r["ms_total"] / 30, 30e3 / r["ms_total"], 1e3 * p["ms_pair"] / max(p["n_pair_launches"], 1)))
