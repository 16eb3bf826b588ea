"""cProfile of the reverse sweep (host side) at C2u."""
import cProfile, pstats, os, sys
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
rollout_value_and_grad(p)
pr = cProfile.Profile(); pr.enable()
for _ in range(3): rollout_value_and_grad(p)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
