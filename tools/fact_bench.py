"""Exact-GP factorisation timing at C2 (for a kernel trace)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pilco_amd import _lib, synthetic
ctx = _lib.Context()
c = synthetic.config_c2()
ctx.gp_set_data(0, c["X"], c["Y"]); ctx.gp_set_hyp(0, c["lengthscales"], c["variance"], c["noise"])
print("exact GP factorisation (N=1000, D=10, E=10): %.2f ms" % ctx.factorize_timed(0, int(os.environ.get("FACT_REPS", "5"))))
