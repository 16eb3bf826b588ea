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
    print("glue0 [%.2f] -> gap %.2f -> prep [init %.2f gj %.2f rows %.2f red %.2f = %.2f] -> gap %.2f -> pair [w0 %.2f, last wave ends +%.2f] -> gap(after w0) %.2f -> glue [loads %.2f pack %.2f asm %.2f prop %.2f joint %.2f = %.2f; reward wg %.2f]  total %.2f us" % (
        us(24, 29), us(29, 0), us(0, 1), us(1, 2), us(2, 3), us(3, 4), us(0, 4), us(4, 16), us(16, 17), us(17, 18), us(17, 8),
        us(8, 9), us(9, 10), us(10, 11), us(11, 12), us(12, 13), us(8, 13), us(20, 21), us(24, 13)))

b = ctx.debug_blocks(960 + 256 + 8)
we = b[960:960+256]
print("pair wave end stamps relative to wave 0 end (every 8th wave): min %+.1f max %+.1f us; by position: %s" % (min((x-ts[17])/100.0 for x in we if x), max((x-ts[17])/100.0 for x in we if x), " ".join("%+.0f" % ((x-ts[17])/100.0) for x in we[::16])))
NX, NY = 61, 4   # prep grid at C2: 55 pair columns + 6 spare columns, 4 row chunks
st = {}
for y in range(NY):
    for x in range(NX):
        k = y * NX + x
        if 2 * k + 1 < 894 and b[2 * k]:
            st[(x, y)] = (b[2 * k], b[2 * k + 1])
t0 = min(v[0] for v in st.values())
def rng(keys):
    v = [((st[k][1] - st[k][0]) / 100.0, (st[k][1] - t0) / 100.0) for k in keys if k in st and st[k][1]]
    return "n=%d dur %.1f..%.1f us, end %.1f..%.1f us" % (len(v), min(x[0] for x in v), max(x[0] for x in v), min(x[1] for x in v), max(x[1] for x in v)) if v else "none"
print("prep pair blocks :", rng([(x, y) for x in range(55) for y in range(NY)]))
spare = [(55 + i // NY, i % NY) for i in range(24)]
print("prep mean blocks :", rng(spare[:20]))
print("prep reward block:", rng(spare[20:21]))
if ts[40]:
    print("mean block (output 0, chunk 0) rel. to prep block0 start: init-sync %.2f | gj+staging %.2f | rows %.2f | wave sums %.2f | final %.2f  (end %.2f us)" % (
        us(0, 40), us(40, 41), us(41, 42), us(42, 43), us(43, 44), us(0, 44)))
