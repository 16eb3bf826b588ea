"""Value-and-gradient rollout at C2u under the current environment: timing + the gradient saved / compared.
usage: python tools/grad_ab.py save|cmp <file.npz>   (developer tool: A/B of environment switches or libraries in ONE gpurun call)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pilco_amd import _lib, synthetic
mode, path = sys.argv[1], sys.argv[2]
N, D, E, H = 1000, 11, 10, 40
cfg = synthetic.config_c2(N=N, D=D, E=E)
ctx = _lib.Context()
ctx.gp_set_data(0, cfg["X"], cfg["Y"]); ctx.gp_set_hyp(0, cfg["lengthscales"], cfg["variance"], cfg["noise"]); ctx.gp_factorize(0)
pol = dict(kind=_lib.POLICY_LINEAR, state_dim=E, control_dim=1, W=cfg["W"], b=cfg["b"].ravel(), max_action=1.0, squash=True)
rw = [dict(kind=_lib.REWARD_EXPONENTIAL, coef=1.0, W=np.eye(E), t=np.zeros(E))]
grd = lambda: ctx.rollout_grad(pol, rw, cfg["m0"], cfg["S0"], H)
fwd = lambda: ctx.rollout(pol, rw, cfg["m0"], cfg["S0"], H)
for _ in range(3): grd(); fwd()
def med(fn, n=15):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); ts.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(ts)), float(np.min(ts))
g_ms, g_min = med(grd); f_ms, f_min = med(fwd)
r, dW, db = grd(); r2, dW2, db2 = grd()
print("%s: R_grad %.3f ms (min %.3f)  R_fwd %.3f ms  ratio %.3f  repeat-bitwise %s" % (os.environ.get("AB_TAG", "run"), g_ms, g_min, f_ms, g_ms / f_ms,
      bool(r == r2 and np.array_equal(dW, dW2) and np.array_equal(db, db2))))
if mode == "save":
    np.savez(path, r=r, dW=dW, db=db)
else:
    g = np.load(path)
    rel = lambda a, b: float(np.max(np.abs(np.asarray(a) - np.asarray(b)) / np.maximum(np.abs(np.asarray(b)), 1e-300)))
    print("   vs %s: reward rel %.2e  dW rel %.2e  db rel %.2e" % (os.path.basename(path), rel(r, g["r"]), rel(dW, g["dW"]), rel(db, g["db"])))
