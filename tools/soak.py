"""Soak test: many back-to-back rollouts / gradients must be bitwise identical (races would show up as flips)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pilco_amd import _lib, synthetic
from pilco_amd.models import PILCO
ctx = _lib.Context()
c = synthetic.config_c2()
ctx.gp_set_data(0, c["X"], c["Y"]); ctx.gp_set_hyp(0, c["lengthscales"], c["variance"], c["noise"]); ctx.gp_factorize(0)
pol = dict(kind=_lib.POLICY_NONE, state_dim=10, control_dim=0)
rw = [dict(kind=_lib.REWARD_EXPONENTIAL, coef=1.0, W=np.eye(10), t=np.zeros(10))]
ref = ctx.rollout(pol, rw, c["m0"], c["S0"], 40)
bad = 0
t0 = time.time()
for i in range(400):
    out = ctx.rollout(pol, rw, c["m0"], c["S0"], 40)
    if not all(np.array_equal(a, b) for a, b in zip(out, ref)):
        bad += 1
print("C2 rollouts: 400 repeats, %d differ from the first (%.2f ms each); reward %.12f" % (bad, (time.time() - t0) / 400 * 1e3, ref[2][0, 0]))
cu = synthetic.config_c2(N=1000, D=11, E=10)
p = PILCO((cu["X"], cu["Y"]), horizon=40, ctx=ctx)
for i, mdl in enumerate(p.mgpr.models):
    mdl.kernel.lengthscales.assign(cu["lengthscales"][i]); mdl.kernel.variance.assign(cu["variance"][i]); mdl.likelihood.variance.assign(cu["noise"][i])
p.controller.W.assign(cu["W"]); p.controller.b.assign(cu["b"]); p.controller.max_action = 1.0
p.m_init, p.S_init = cu["m0"], cu["S0"]
r0, (W0, b0) = p.value_and_gradient()
badg = 0
for i in range(60):
    r, (W, b) = p.value_and_gradient()
    if r != r0 or not np.array_equal(W, W0) or not np.array_equal(b, b0):
        badg += 1
print("C2u value+gradient: 60 repeats, %d differ from the first" % badg)
sys.exit(1 if (bad or badg) else 0)
