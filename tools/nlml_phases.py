"""Where one evaluation of the exact-GP training objective (pilco_gp_nlml after pilco_gp_set_hyp, C2 size) spends its wall
time: hyper-parameter upload, factorisation, the objective's own kernels and copies (medians of 40, ms)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pilco_amd import _lib, synthetic
ctx = _lib.Context()
c = synthetic.config_c2()
ctx.gp_set_data(0, c["X"], c["Y"])
hyp = lambda: ctx.gp_set_hyp(0, c["lengthscales"], c["variance"], c["noise"])
def med(fn, n=40):
    for _ in range(3): fn()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); ts.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(ts))
hyp(); ctx.gp_factorize(0)
t_h = med(hyp)
t_hf = med(lambda: (hyp(), ctx.gp_factorize(0)))
t_hn = med(lambda: (hyp(), ctx.gp_nlml(0, 10, 10)))
hyp(); ctx.gp_nlml(0, 10, 10)
t_tail = med(lambda: ctx.gp_nlml(0, 10, 10))
t_tail_v = med(lambda: ctx.gp_nlml(0, 10, 10, want_grad=False))
print("set_hyp %.3f | set_hyp + factorize %.3f | set_hyp + nlml %.3f | nlml with a valid factorisation %.3f (value only %.3f) | device factorisation %.3f"
      % (t_h, t_hf, t_hn, t_tail, t_tail_v, ctx.factorize_timed(0, 20)))
