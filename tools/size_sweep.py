"""Forward rollout (R-fwd) and value + gradient (R-grad) against the model size: N training points at D = 11 (state 10 +
1 control, linear controller), E = 10, H = 40 -- the C2u recipe of SURVEY.md 8(d) at other N."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pilco_amd import synthetic
from pilco_amd.models import PILCO

def med(fn, n):
    fn(); ts = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); ts.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(ts))

print("%6s %12s %12s %12s %10s %14s" % ("N", "R-fwd ms", "rollouts/s", "R-grad ms", "ratio", "Gexp/s (fwd)"))
for N in [int(a) for a in sys.argv[1:]] or [250, 500, 1000, 2000, 3000]:
    c = synthetic.config_c2(N=N, D=11, E=10)
    p = PILCO((c["X"], c["Y"]), horizon=40)
    for i, mdl in enumerate(p.mgpr.models):
        mdl.kernel.lengthscales.assign(c["lengthscales"][i]); mdl.kernel.variance.assign(c["variance"][i]); mdl.likelihood.variance.assign(c["noise"][i])
    p.controller.W.assign(c["W"]); p.controller.b.assign(c["b"]); p.controller.max_action = 1.0
    p.m_init, p.S_init = c["m0"], c["S0"]
    f = med(p.compute_reward, 7)
    g = med(lambda: p.value_and_gradient(), 5)
    exps = 40 * (55 * N * N + 10 * N)
    print("%6d %12.3f %12.1f %12.3f %10.2f %14.1f" % (N, f, 1e3 / f, g, g / f, exps / (f * 1e-3) / 1e9), flush=True)
