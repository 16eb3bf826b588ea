import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
from pilco_amd import _lib, synthetic
ctx = _lib.Context()
c4 = synthetic.config_c4()
ctx.gp_set_data(0, c4["X"], c4["Y"]); ctx.gp_set_hyp(0, c4["lengthscales"], c4["variance"], c4["noise"])
Z = np.stack([c4["Z"]] * 10)
for _ in range(6): ctx.gp_fitc_nlml(0, Z, 10, 10)
ctx.gp_set_inducing(0, c4["Z"]); ctx.gp_factorize(0)
print("FITC factorisation %.3f ms" % ctx.factorize_timed(0, 5))
