"""Aggregate throughput of pilco_rollout_batch at C2 for several batch sizes (developer tool)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pilco_amd import _lib, synthetic
N, D, E, H = 1000, 10, 10, 40
cfg = synthetic.config_c2(N=N, D=D, E=E)
ctx = _lib.Context()
ctx.gp_set_data(0, cfg["X"], cfg["Y"]); ctx.gp_set_hyp(0, cfg["lengthscales"], cfg["variance"], cfg["noise"]); ctx.gp_factorize(0)
pol = dict(kind=_lib.POLICY_NONE, state_dim=E, control_dim=0)
rw = [dict(kind=_lib.REWARD_EXPONENTIAL, coef=1.0, W=np.eye(E), t=np.zeros(E))]
rs = np.random.RandomState(0)
print("GPU_MAX_HW_QUEUES =", os.environ.get("GPU_MAX_HW_QUEUES", "(default)"))
for _ in range(3):
    ctx.rollout(pol, rw, cfg["m0"], cfg["S0"], H)
t0 = time.perf_counter()
for _ in range(20):
    ctx.rollout(pol, rw, cfg["m0"], cfg["S0"], H)
print("solo: %.1f rollouts/s" % (20 / (time.perf_counter() - t0)))
for B in (2, 3, 4, 6, 8, 12, 16):
    m0 = cfg["m0"] + 0.05 * rs.randn(B, E)
    S0 = np.stack([cfg["S0"]] * B)
    ctx.rollout_batch([pol] * B, rw, m0, S0, H)
    ctx.rollout_batch([pol] * B, rw, m0, S0, H)
    reps = 10
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.rollout_batch([pol] * B, rw, m0, S0, H)
    dt = time.perf_counter() - t0
    print("B=%2d: %.1f rollouts/s aggregate (%.3f ms per batch)" % (B, B * reps / dt, dt / reps * 1e3), flush=True)
