"""Phase stamps of ONE pair workgroup of the fused Jacobian-record kernel (k_mm_jac_rec, pair 0 of step 0) at C2u, from a library
built with EXTRA=-DJAC_STAMPS (PILCO_LIB): load bursts (G blocks + column coefficients) | barrier | MFMA loop over the points |
reduction + N|A|I | record.  The workgroup runs beside 3 x 256 others: these are latencies under load."""
import numpy as np, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pilco_amd import _lib, synthetic
cfg = synthetic.config_c2(N=1000, D=11, E=10)
ctx = _lib.Context()
ctx.debug_timestamps(read=False)
ctx.gp_set_data(0, cfg["X"], cfg["Y"]); ctx.gp_set_hyp(0, cfg["lengthscales"], cfg["variance"], cfg["noise"]); ctx.gp_factorize(0)
pol = dict(kind=_lib.POLICY_LINEAR, state_dim=10, control_dim=1, W=cfg["W"], b=cfg["b"].ravel(), max_action=1.0, squash=True)
rw = [dict(kind=_lib.REWARD_EXPONENTIAL, coef=1.0, W=np.eye(10), t=np.zeros(10))]
H = int(sys.argv[1]) if len(sys.argv) > 1 else 40
for rep in range(4):
    ctx.rollout_grad(pol, rw, cfg["m0"], cfg["S0"], H)
    ts = ctx.debug_timestamps()
    us = lambda a, b: (ts[b] - ts[a]) / 100.0
    if not ts[53]:
        print("library was not built with -DJAC_STAMPS"); break
    print("pair workgroup (0, step 0) of k_mm_jac_rec, H = %d: loads + column coefficients %.2f | barrier %.2f | MFMA loop %.2f | reduce + N|A|I %.2f | record %.2f = %.2f us" % (
        H, us(48, 49), us(49, 50), us(50, 51), us(51, 52), us(52, 53), us(48, 53)))
