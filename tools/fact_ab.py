"""Exact / FITC factorisation and one evaluation of the training objectives, graph replay against eager launches (developer tool)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pilco_amd import _lib, synthetic
def med(fn, reps=15):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append((time.perf_counter() - t0) * 1e3)
    return "%.3f ms (min %.3f max %.3f)" % (np.median(ts), min(ts), max(ts))
for graph in (1, 0):
    ctx = _lib.Context()
    ctx.use_graph(bool(graph))
    c = synthetic.config_c2()
    E, D = c["Y"].shape[1], c["X"].shape[1]
    ctx.gp_set_data(0, c["X"], c["Y"]); ctx.gp_set_hyp(0, c["lengthscales"], c["variance"], c["noise"])
    ctx.factorize_timed(0, 2)
    f = sorted(ctx.factorize_timed(0, 1) for _ in range(20))
    def nlml():
        ctx.gp_set_hyp(0, c["lengthscales"], c["variance"], c["noise"]); ctx.gp_nlml(0, D, E)
    nlml()
    print("graph=%d exact factorisation %.3f / %.3f / %.3f ms (min / median / max); nlml evaluation %s" % (graph, f[0], np.median(f), f[-1], med(nlml)))
    c4 = synthetic.config_c4()
    ctx.gp_set_data(0, c4["X"], c4["Y"]); ctx.gp_set_hyp(0, c4["lengthscales"], c4["variance"], c4["noise"]); ctx.gp_set_inducing(0, c4["Z"])
    ctx.factorize_timed(0, 2)
    f = sorted(ctx.factorize_timed(0, 1) for _ in range(20))
    Z4 = np.stack([c4["Z"]] * E)
    ctx.gp_fitc_nlml(0, Z4, D, E)
    print("graph=%d FITC factorisation %.3f / %.3f / %.3f ms; FITC objective evaluation %s" % (graph, f[0], np.median(f), f[-1], med(lambda: ctx.gp_fitc_nlml(0, Z4, D, E), 9)))
    ctx.close()
