"""usage: python tools/head_phases.py [D [grad]].  In-kernel phase durations (us) of the FUSED head (serial link + operands of the next step) in an eager H=3 rollout:
stamps of workgroup (0,0) of the last head (slots 56..61 link, 0..4 prep part) and the pair kernel around it."""
import numpy as np, sys, os
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
GRAD = len(sys.argv) > 2 and sys.argv[2] == "grad"   # the head of a value-and-gradient rollout (Jacobian tape: the link packs the sweep's partials)
for rep in range(4):
    if GRAD: ctx.rollout_grad(pol, rw, cfg["m0"], cfg["S0"], 3)
    else: ctx.rollout(pol, rw, cfg["m0"], cfg["S0"], 3)
    ts = ctx.debug_timestamps()
    us = lambda a, b: (ts[b] - ts[a]) / 100.0
    print("head(2): link [loads %.2f pack %.2f asm+prop %.2f (%.2f) joint %.2f = %.2f] -> prep part [init %.2f gj %.2f rows %.2f = %.2f]  head total %.2f | -> pair w0 start +%.2f, w0 %.2f | final glue [loads %.2f pack %.2f asm+prop %.2f = %.2f]" % (
        us(56, 57), us(57, 58), us(58, 59), us(59, 60), us(60, 61), us(56, 61), us(61, 1), us(1, 2), us(2, 3), us(61, 4), us(56, 4),
        us(4, 16), us(16, 17), us(8, 9), us(9, 10), us(10, 12), us(8, 12)))
print("inside the Gauss-Jordan phase of pair block (0,0): build %.2f | gj_wave %.2f | Q store + barrier %.2f ; first staging wave done %.2f after the phase began" % (
    us(1, 5), us(5, 6), us(6, 2), us(1, 7)))
if ts[36] and ts[37]:   # (a -DLINK_STAMPS build only)
    print("inside the joint phase (linear controller): moments M, S, V %.2f | squash %.2f | joint Gaussian + stores %.2f" % ((ts[36] - ts[60]) / 100.0, us(36, 37), (ts[61] - ts[37]) / 100.0))
if ts[34] and ts[35]:   # (a -DHEAD_ROW_STAMPS build only)
    print("inside the rows phase of pair block (0,0), thread 0: point and x = zeta / l^2 %.2f | y = Q x and the quadratic form %.2f | stores issued %.2f" % (us(2, 34), us(34, 35), us(35, 3)))
if ts[17] > ts[16]:
    print("engine clock during the pair kernel (wave 0): %.0f MHz  (shader-clock ticks %d over %.2f us of wall clock)" % (
    (ts[33] - ts[32]) / ((ts[17] - ts[16]) / 100.0), ts[33] - ts[32], (ts[17] - ts[16]) / 100.0))
