"""Headline rollout (C2: N=1000, D=10, E=10, H=40) wall-clock median for the library named by PILCO_LIB (developer A/B tool)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pilco_amd import _lib, synthetic
cfg = synthetic.config_c2()
ctx = _lib.Context()
ctx.gp_set_data(0, cfg["X"], cfg["Y"]); ctx.gp_set_hyp(0, cfg["lengthscales"], cfg["variance"], cfg["noise"]); ctx.gp_factorize(0)
pol = dict(kind=_lib.POLICY_NONE, state_dim=10, control_dim=0)
rw = [dict(kind=_lib.REWARD_EXPONENTIAL, coef=1.0, W=np.eye(10), t=np.zeros(10))]
for _ in range(5): ctx.rollout(pol, rw, cfg["m0"], cfg["S0"], 40)
ts = []
for _ in range(40):
    t0 = time.perf_counter(); ctx.rollout(pol, rw, cfg["m0"], cfg["S0"], 40); ts.append((time.perf_counter() - t0) * 1e3)
print("%s: %.4f ms median (min %.4f) -> %.1f rollouts/s" % (os.path.basename(_lib.LIB_PATH), np.median(ts), min(ts), 1e3 / np.median(ts)))
