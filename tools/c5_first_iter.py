"""Where the FIRST optimize_models of the config-5 loop (N = 25) spends its time: one-time costs (library load, first launches,
graph captures) against the fits themselves.  Developer tool."""
import os, sys, time, cProfile, pstats
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
t0 = time.perf_counter()
from pilco_amd import _lib
from pilco_amd.models import PILCO
from pilco_amd.controllers import RbfController
ctx = _lib.get_context()
print("import + context: %.3f s" % (time.perf_counter() - t0))
rs = np.random.RandomState(0)
X = rs.randn(25, 5) * np.array([0.3, 0.1, 0.5, 0.8, 2.0])
Y = 0.05 * np.stack([np.sin(X @ rs.randn(5)) for _ in range(4)], 1) + 1e-3 * rs.randn(25, 4)
np.random.seed(0)
p = PILCO((X, Y), controller=RbfController(state_dim=4, control_dim=1, num_basis_functions=10, max_action=3.0), horizon=40)
for rep in range(3):
    np.random.seed(1)
    for m in p.mgpr.models:
        m.kernel.lengthscales.assign(np.ones(5)); m.kernel.variance.assign(1.0); m.likelihood.variance.assign(1.0)
    pr = cProfile.Profile(); t0 = time.perf_counter(); pr.enable(); p.optimize_models(); pr.disable()
    print("optimize_models #%d (N=25, same start): %.3f s" % (rep, time.perf_counter() - t0))
    if rep in (0, 2):
        pstats.Stats(pr).sort_stats("cumulative").print_stats(12)
