"""Print in-kernel phase durations (us) and inter-kernel gaps of a one-step rollout."""
import numpy as np, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pilco_amd import _lib, synthetic
cfg = synthetic.config_c2()
ctx = _lib.Context()
ctx.debug_timestamps(read=False)
ctx.gp_set_data(0, cfg["X"], cfg["Y"]); ctx.gp_set_hyp(0, cfg["lengthscales"], cfg["variance"], cfg["noise"]); ctx.gp_factorize(0)
pol = dict(kind=_lib.POLICY_NONE, state_dim=10, control_dim=0)
rw = [dict(kind=_lib.REWARD_EXPONENTIAL, coef=1.0, W=np.eye(10), t=np.zeros(10))]
for rep in range(3):
    ctx.rollout(pol, rw, cfg["m0"], cfg["S0"], 1)
    ts = ctx.debug_timestamps()
    us = lambda a, b: (ts[b] - ts[a]) / 100.0
    if ts[30]: print("stamp kernel between prep and pair: prep block0 end -> stamp %.2f us; stamp -> pair w0 start %.2f us" % (us(4, 30), us(30, 16)))
    print("prep row-phase end of waves 0..3 rel. to block0 start: block(0,0) %s | block(30,1): rows start %s end %s" % (
        " ".join("%.1f" % us(0, 40 + w) for w in range(4)), " ".join("%.1f" % us(0, 48 + w) for w in range(4)), " ".join("%.1f" % us(0, 44 + w) for w in range(4))))
    print("glue0 [%.2f] -> gap %.2f -> prep [init %.2f gj %.2f rows %.2f red %.2f = %.2f] -> gap %.2f -> pair [w0 %.2f, last wave ends +%.2f] -> gap(after w0) %.2f -> glue [loads %.2f pack %.2f asm %.2f prop %.2f joint %.2f = %.2f; reward wg %.2f]  total %.2f us" % (
        us(24, 29), us(29, 0), us(0, 1), us(1, 2), us(2, 3), us(3, 4), us(0, 4), us(4, 16), us(16, 17), us(17, 18), us(17, 8),
        us(8, 9), us(9, 10), us(10, 11), us(11, 12), us(12, 13), us(8, 13), us(20, 21), us(24, 13)))

b = ctx.debug_blocks(960 + 256 + 8)
we = b[960:960+256]
print("pair wave end stamps relative to wave 0 end (every 8th wave): min %+.1f max %+.1f us; by position: %s" % (min((x-ts[17])/100.0 for x in we if x), max((x-ts[17])/100.0 for x in we if x), " ".join("%+.0f" % ((x-ts[17])/100.0) for x in we[::16])))
t0 = min(b[0::2])
import collections
print("prep per-block (start, dur) us by chunk row:")
for ch in range(4):
    row = [( (b[2*(ch*55+pl)]-t0)/100.0, (b[2*(ch*55+pl)+1]-b[2*(ch*55+pl)])/100.0) for pl in range(55)]
    print(" ch%d starts: %s" % (ch, " ".join("%.1f" % r[0] for r in row[:55:6])))
    print("     durs:   %s" % (" ".join("%.1f" % r[1] for r in row[:55:6])))
durs = [((b[2*k+1]-b[2*k])/100.0, (b[2*k]-t0)/100.0, k % 55, k // 55) for k in range(220)]
durs.sort(reverse=True)
print("slowest prep blocks (dur, start, pl, ch):", [(round(d,1), round(s,1), pl, ch) for d, s, pl, ch in durs[:12]])
print("fastest:", [(round(d,1), round(s,1), pl, ch) for d, s, pl, ch in durs[-5:]])
