"""Print in-kernel phase durations (us) of the last prep / glue launch of a short rollout."""
import numpy as np, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pilco_amd import _lib, synthetic
cfg = synthetic.config_c2()
ctx = _lib.Context()
ctx.debug_timestamps(read=False)
ctx.gp_set_data(0, cfg["X"], cfg["Y"]); ctx.gp_set_hyp(0, cfg["lengthscales"], cfg["variance"], cfg["noise"]); ctx.gp_factorize(0)
pol = dict(kind=_lib.POLICY_NONE, state_dim=10, control_dim=0)
rw = [dict(kind=_lib.REWARD_EXPONENTIAL, coef=1.0, W=np.eye(10), t=np.zeros(10))]
for H in (3, 4):
    ctx.rollout(pol, rw, cfg["m0"], cfg["S0"], H)
    ts = ctx.debug_timestamps()
    us = lambda a, b: (ts[b] - ts[a]) / 100.0
    print("H=%d prep: init %.2f gj %.2f rows %.2f reduce %.2f | total %.2f us" % (H, us(0, 1), us(1, 2), us(2, 3), us(3, 4), us(0, 4)))
    print("     glue: loads %.2f pack %.2f assemble %.2f propagate %.2f policy/joint %.2f | total %.2f ; reward block %.2f us" % (
        us(8, 9), us(9, 10), us(10, 11), us(11, 12), us(12, 13), us(8, 13), us(20, 21)))
