import numpy as np, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pilco_amd import _lib, synthetic
from oracle import tf_path as tp
ctx = _lib.Context()
for name, c, D, E in (("c1", synthetic.config_c1(), 3, 2), ("mid", synthetic.config_c2(N=300, D=5, E=4, noise=1e-2, seed=11, control_dim=1), 5, 4), ("c2", synthetic.config_c2(), 10, 10)):
    ctx.gp_set_data(0, c["X"], c["Y"]); ctx.gp_set_hyp(0, c["lengthscales"], c["variance"], c["noise"]); ctx.gp_factorize(0)
    m = c["m"] if "m" in c else np.zeros((1, D)); s = c["s"] if "s" in c else 0.1 * np.eye(D)
    M, S, V = ctx.gp_predict(0, m, s, D, E)
    iK, beta = tp.calculate_factorizations(c["X"], c["Y"], c["lengthscales"], c["variance"], c["noise"])
    Mo, So, Vo = tp.predict_given_factorizations_pairs(c["X"], c["lengthscales"], c["variance"], m, s, iK, beta)
    print(name, "M err", np.max(np.abs(M - Mo)), "S err", np.max(np.abs(S - So)), "V err", np.max(np.abs(V - Vo)), "nan:", np.isnan(M).any(), np.isnan(S).any(), np.isnan(V).any())
