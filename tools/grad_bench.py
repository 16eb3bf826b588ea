"""R-grad timing: value + gradient of one H=40 rollout w.r.t. a linear controller,
config C2u (N=1000, state 10 + 1 control -> D=11, E=10)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pilco_amd import synthetic
from pilco_amd.models import PILCO
from pilco_amd.adjoint import rollout_value_and_grad
c = synthetic.config_c2(N=1000, D=11, E=10)
p = PILCO((c["X"], c["Y"]), horizon=40)
for i, mdl in enumerate(p.mgpr.models):
    mdl.kernel.lengthscales.assign(c["lengthscales"][i]); mdl.kernel.variance.assign(c["variance"][i]); mdl.likelihood.variance.assign(c["noise"][i])
p.controller.W.assign(c["W"]); p.controller.b.assign(c["b"]); p.controller.max_action = 1.0
p.m_init, p.S_init = c["m0"], c["S0"]
t0 = time.time(); r = p.compute_reward(); t1 = time.time()
for _ in range(2): r = p.compute_reward()
t2 = time.time()
v, (Wb, bb) = rollout_value_and_grad(p)
t3 = time.time()
v, (Wb, bb) = rollout_value_and_grad(p)
t4 = time.time()
print("C2u forward rollout %.2f ms; value+gradient %.1f ms (first %.1f ms); reward %.6f |dW| %.3e" % ((t2 - t1) / 2 * 1e3, (t4 - t3) * 1e3, (t3 - t2) * 1e3, v, np.abs(Wb).max()))
