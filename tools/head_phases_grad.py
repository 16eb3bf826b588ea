"""Fused-head phase durations (us) at C2u (D=11, linear controller), forward rollout against the value-and-gradient
rollout (Jacobian tape: the head packs the sweep's tile partials and writes the tape), eager H=4; stamps as in head_phases.py."""
import numpy as np, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pilco_amd import _lib, synthetic
D = 11
cfg = synthetic.config_c2(N=1000, D=D, E=10)
ctx = _lib.Context()
ctx.debug_timestamps(read=False)
ctx.gp_set_data(0, cfg["X"], cfg["Y"]); ctx.gp_set_hyp(0, cfg["lengthscales"], cfg["variance"], cfg["noise"]); ctx.gp_factorize(0)
pol = dict(kind=_lib.POLICY_LINEAR, state_dim=10, control_dim=1, W=cfg["W"], b=cfg["b"].ravel(), max_action=1.0, squash=True)
rw = [dict(kind=_lib.REWARD_EXPONENTIAL, coef=1.0, W=np.eye(10), t=np.zeros(10))]
def show(tag):
    ts = ctx.debug_timestamps()
    us = lambda a, b: (ts[b] - ts[a]) / 100.0
    print("%s head: link [loads %.2f pack %.2f asm+prop %.2f (%.2f) joint %.2f = %.2f] -> prep part [init %.2f gj %.2f rows %.2f = %.2f]  head total %.2f" % (
        tag, us(56, 57), us(57, 58), us(58, 59), us(59, 60), us(60, 61), us(56, 61), us(61, 1), us(1, 2), us(2, 3), us(61, 4), us(56, 4)))
for rep in range(3):
    ctx.rollout(pol, rw, cfg["m0"], cfg["S0"], 4)
    show("forward")
for rep in range(3):
    ctx.rollout_grad(pol, rw, cfg["m0"], cfg["S0"], 4)
    show("tape   ")
