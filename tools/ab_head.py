"""A/B of two library builds in ONE gpurun call: per-rollout time and the non-pair time per step (head + boundaries) at C2 / C2u.
usage: python tools/ab_head.py <dir holding a pilco_amd package> [D]"""
import os, sys
import numpy as np
root = os.path.abspath(sys.argv[1])
sys.path.insert(0, root)
from pilco_amd import _lib, synthetic
D = int(sys.argv[2]) if len(sys.argv) > 2 else 10
E, H = 10, 40
cfg = synthetic.config_c2(N=1000, D=D, E=E)
ctx = _lib.Context()
ctx.gp_set_data(0, cfg["X"], cfg["Y"]); ctx.gp_set_hyp(0, cfg["lengthscales"], cfg["variance"], cfg["noise"]); ctx.gp_factorize(0)
if D == E:
    pol = dict(kind=_lib.POLICY_NONE, state_dim=E, control_dim=0)
else:
    pol = dict(kind=_lib.POLICY_LINEAR, state_dim=E, control_dim=D - E, W=cfg["W"], b=cfg["b"].ravel(), max_action=1.0, squash=True)
rw = [dict(kind=_lib.REWARD_EXPONENTIAL, coef=1.0, W=np.eye(E), t=np.zeros(E))]
ctx.rollout_timed(pol, rw, cfg["m0"], cfg["S0"], H, 3, time_pair=False)
res = []
for rep in range(5):
    r = ctx.rollout_timed(pol, rw, cfg["m0"], cfg["S0"], H, 20, time_pair=False)
    res.append(r["ms_total"] / 20)
p = ctx.rollout_timed(pol, rw, cfg["m0"], cfg["S0"], H, 1, time_pair=True)
pair_us = 1e3 * p["ms_pair"] / max(p["n_pair_launches"], 1)
ms = float(np.median(res))
print("%-28s D=%d: %.3f ms per rollout (%.1f /s); pair kernel %.2f us (events); everything else %.2f us per step" % (
    os.path.basename(root) or root, D, ms, 1e3 / ms, pair_us, ms * 1e3 / H - pair_us))
