"""World-size-1 check that the RCCL branch of the rollout is captured into a hipGraph and replays correctly."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pilco_amd import _lib, synthetic
c = synthetic.config_c2()
res = {}
for mode in ("plain", "rccl"):
    ctx = _lib.Context()
    if mode == "rccl":
        ctx.comm_init(ctx.comm_unique_id(), 0, 1)
    ctx.gp_set_data(0, c["X"], c["Y"]); ctx.gp_set_hyp(0, c["lengthscales"], c["variance"], c["noise"]); ctx.gp_factorize(0)
    pol = dict(kind=_lib.POLICY_NONE, state_dim=10, control_dim=0)
    rw = [dict(kind=_lib.REWARD_EXPONENTIAL, coef=1.0, W=np.eye(10), t=np.zeros(10))]
    ctx.rollout_timed(pol, rw, c["m0"], c["S0"], 40, 2, time_pair=False)
    r = ctx.rollout_timed(pol, rw, c["m0"], c["S0"], 40, 10, time_pair=False)
    res[mode] = r
    print(mode, "ms/rollout %.3f" % (r["ms_total"] / 10), "reward", r["reward"][0, 0])
    ctx.close()
print("identical:", np.array_equal(res["plain"]["mH"], res["rccl"]["mH"]), np.array_equal(res["plain"]["SH"], res["rccl"]["SH"]))
