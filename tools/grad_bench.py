"""Value-and-gradient rollout against the forward rollout at C2u (N=1000, D=11, E=10, H=40): wall-clock medians, the sweep
kernel's mean launch duration (HIP events) and the host-side split (PILCO_GRAD_TIMING).  Developer tool: A/B two builds in
ONE gpurun call with PILCO_LIB=<other .so>."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pilco_amd import _lib, synthetic
N, D, E, H = 1000, 11, 10, 40
cfg = synthetic.config_c2(N=N, D=D, E=E)
ctx = _lib.Context()
ctx.gp_set_data(0, cfg["X"], cfg["Y"]); ctx.gp_set_hyp(0, cfg["lengthscales"], cfg["variance"], cfg["noise"]); ctx.gp_factorize(0)
pol = dict(kind=_lib.POLICY_LINEAR, state_dim=E, control_dim=1, W=cfg["W"], b=cfg["b"].ravel(), max_action=1.0, squash=True)
rw = [dict(kind=_lib.REWARD_EXPONENTIAL, coef=1.0, W=np.eye(E), t=np.zeros(E))]
def med(fn, reps):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(ts)), float(np.min(ts))
fwd = lambda: ctx.rollout(pol, rw, cfg["m0"], cfg["S0"], H)
grd = lambda: ctx.rollout_grad(pol, rw, cfg["m0"], cfg["S0"], H)
for _ in range(3): fwd(); grd()
f_ms, f_min = med(fwd, 15)
g_ms, g_min = med(grd, 15)
g1 = grd(); g2 = grd()
same = all(np.array_equal(np.asarray(a), np.asarray(b)) for a, b in zip(g1, g2))
ctx.set_pair_timing(True); grd(); sw_ms, sw_n = ctx.get_pair_timing(); ctx.set_pair_timing(False)
ctx.set_pair_timing(True); fwd(); pw_ms, pw_n = ctx.get_pair_timing(); ctx.set_pair_timing(False)
print("lib %s" % os.path.basename(_lib.LIB_PATH))
print("R_fwd_C2u %.3f ms (min %.3f)  R_grad_C2u %.3f ms (min %.3f)  ratio %.3f  bitwise repeat %s" % (f_ms, f_min, g_ms, g_min, g_ms / f_ms, same))
print("sweep %.2f us x %d   forward pair %.2f us x %d" % (sw_ms * 1e3 / max(sw_n, 1), sw_n, pw_ms * 1e3 / max(pw_n, 1), pw_n))
print("chain estimate: grad %.3f ms = %d x (head + %.1f us) -> head+gaps %.1f us/step incl. tail; fwd head+gaps %.1f us/step" % (
    g_ms, H, sw_ms * 1e3 / max(sw_n, 1), g_ms * 1e3 / H - sw_ms * 1e3 / max(sw_n, 1), f_ms * 1e3 / H - pw_ms * 1e3 / max(pw_n, 1)))
