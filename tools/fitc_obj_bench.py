"""FITC training objective (value + all gradients) at config 4 (M = 200, N = 5000, D = 10, E = 10): median wall clock per evaluation.
Developer tool: A/B environment switches or libraries in ONE gpurun call."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pilco_amd import _lib, synthetic
ctx = _lib.Context()
c4 = synthetic.config_c4()
ctx.gp_set_data(0, c4["X"], c4["Y"]); ctx.gp_set_hyp(0, c4["lengthscales"], c4["variance"], c4["noise"])
Z = np.stack([c4["Z"]] * 10)
for _ in range(4): out = ctx.gp_fitc_nlml(0, Z, 10, 10)
ts = []
for _ in range(12):
    t0 = time.perf_counter(); out = ctx.gp_fitc_nlml(0, Z, 10, 10); ts.append((time.perf_counter() - t0) * 1e3)
print("%s: FITC objective evaluation median %.3f ms (min %.3f)" % (os.environ.get("AB_TAG", "run"), np.median(ts), np.min(ts)))
if len(sys.argv) > 2:
    arrs = [np.asarray(a) for a in out]
    if sys.argv[1] == "save":
        np.savez(sys.argv[2], *arrs)
    else:
        g = np.load(sys.argv[2])
        for i, a in enumerate(arrs):
            r = g["arr_%d" % i]
            print("   output %d: max rel diff %.2e (scale %.2e)" % (i, float(np.max(np.abs(a - r)) / max(np.max(np.abs(r)), 1e-300)), float(np.max(np.abs(r)))))
