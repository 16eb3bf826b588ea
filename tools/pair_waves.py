"""End stamps of every 8th wave of the stream-K pair kernel (eager H = 3 rollout at C2): how evenly the cost line is cut.
Diagonal pairs come first on the line (waves 0 ..), the off-diagonal pairs behind them."""
import numpy as np, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pilco_amd import _lib, synthetic
D = int(sys.argv[1]) if len(sys.argv) > 1 else 10
cfg = synthetic.config_c2(N=1000, D=D, E=10)
ctx = _lib.Context()
ctx.debug_timestamps(read=False)
ctx.gp_set_data(0, cfg["X"], cfg["Y"]); ctx.gp_set_hyp(0, cfg["lengthscales"], cfg["variance"], cfg["noise"]); ctx.gp_factorize(0)
if D == 10:
    pol = dict(kind=_lib.POLICY_NONE, state_dim=10, control_dim=0)
else:
    pol = dict(kind=_lib.POLICY_LINEAR, state_dim=10, control_dim=D - 10, W=cfg["W"], b=cfg["b"].ravel(), max_action=1.0, squash=True)
rw = [dict(kind=_lib.REWARD_EXPONENTIAL, coef=1.0, W=np.eye(10), t=np.zeros(10))]
for rep in range(3):
    ctx.rollout(pol, rw, cfg["m0"], cfg["S0"], 3)
    ts = ctx.debug_timestamps()
    ends = np.array(ctx.debug_blocks(960 + 400)[960:960 + 384], dtype=np.float64)
    ok = ends > 0
    e = (ends[ok] - ts[16]) / 100.0
    idx = np.nonzero(ok)[0] * 8
    print("%s D=%d: %d stamped waves; wave 0 runs %.1f us; ends (us after wave 0's start): min %.1f  p10 %.1f  median %.1f  p90 %.1f  max %.1f" % (
        os.path.basename(_lib.LIB_PATH), D, ok.sum(), (ts[17] - ts[16]) / 100.0, e.min(), np.percentile(e, 10), np.median(e), np.percentile(e, 90), e.max()))
    for lo in range(0, 3072, 384):
        sel = (idx >= lo) & (idx < lo + 384)
        if sel.any():
            print("   waves %4d..%4d: ends %.1f .. %.1f (mean %.1f)" % (lo, lo + 383, e[sel].min(), e[sel].max(), e[sel].mean()))
