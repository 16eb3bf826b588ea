"""C2u (N=1000, D=11, E=10, H=40, linear controller): forward rollout and value+gradient, median ms."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pilco_amd import synthetic
from pilco_amd.models import PILCO
c = synthetic.config_c2(N=1000, D=11, E=10)
p = PILCO((c["X"], c["Y"]), horizon=40)
for i, mdl in enumerate(p.mgpr.models):
    mdl.kernel.lengthscales.assign(c["lengthscales"][i]); mdl.kernel.variance.assign(c["variance"][i]); mdl.likelihood.variance.assign(c["noise"][i])
p.controller.W.assign(c["W"]); p.controller.b.assign(c["b"]); p.controller.max_action = 1.0
p.m_init, p.S_init = c["m0"], c["S0"]
def med(fn, n=15):
    fn(); ts = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); ts.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(ts))
print("C2u forward %.3f ms   value+gradient %.3f ms" % (med(p.compute_reward), med(lambda: p.value_and_gradient(), 7)))
