"""Phase stamps (us) of the serial link and the operand part in workgroup 0: launch sequence (fused head) vs persistent kernel."""
import numpy as np, sys, os
os.environ["PILCO_PERSIST_DBG"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pilco_amd import _lib, synthetic
D = int(sys.argv[1]) if len(sys.argv) > 1 else 10
cfg = synthetic.config_c2(D=D)
ctx = _lib.Context()
ctx.debug_timestamps(read=False)
ctx.gp_set_data(0, cfg["X"], cfg["Y"]); ctx.gp_set_hyp(0, cfg["lengthscales"], cfg["variance"], cfg["noise"]); ctx.gp_factorize(0)
if D == 10:
    pol = dict(kind=_lib.POLICY_NONE, state_dim=10, control_dim=0)
else:
    pol = dict(kind=_lib.POLICY_LINEAR, state_dim=10, control_dim=D - 10, W=cfg["W"], b=cfg["b"].ravel(), max_action=1.0, squash=True)
rw = [dict(kind=_lib.REWARD_EXPONENTIAL, coef=1.0, W=np.eye(10), t=np.zeros(10))]
for mode in (0, 1):
    ctx.set_rollout_mode(mode)
    for rep in range(3):
        ctx.rollout(pol, rw, cfg["m0"], cfg["S0"], 6)
        ts = ctx.debug_timestamps()
        us = lambda a, b: (ts[b] - ts[a]) / 100.0
        print("mode %d (used %d): link [loads %.2f pack %.2f asm+prop %.2f (%.2f) joint %.2f = %.2f] -> operands [init %.2f gj %.2f rows %.2f = %.2f]  total %.2f" % (
            mode, ctx.last_rollout_mode(), us(56, 57), us(57, 58), us(58, 59), us(59, 60), us(60, 61), us(56, 61), us(61, 1), us(1, 2), us(2, 3), us(61, 4), us(56, 4)))
