"""Per-workgroup start / end stamps of the LAST fused head of an eager H=4 rollout at C2 (or C2u: argv[1] = 11): pair, mean-part
and reward workgroups against the link stamps of workgroup (0,0) -- who ends the launch."""
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
NX, NY = 61, 4
for rep in range(4):
    ctx.rollout(pol, rw, cfg["m0"], cfg["S0"], 4)
    ts = ctx.debug_timestamps()
    b = ctx.debug_blocks(960)
    st = {}
    for y in range(NY):
        for x in range(NX):
            k = y * NX + x
            if 2 * k + 1 < 894 and b[2 * k]:
                st[(x, y)] = (b[2 * k], b[2 * k + 1])
    t0 = min(v[0] for v in st.values())
    def rng(keys):
        v = [((st[k][0] - t0) / 100.0, (st[k][1] - t0) / 100.0) for k in keys if k in st and st[k][1]]
        if not v: return "none"
        return "n=%d start %.1f..%.1f end %.1f..%.1f (mean %.1f)" % (len(v), min(x[0] for x in v), max(x[0] for x in v), min(x[1] for x in v), max(x[1] for x in v), np.mean([x[1] for x in v]))
    spare = [(55 + i // NY, i % NY) for i in range(24)]
    print("%s D=%d link(0,0) %.1f..%.1f | pair blocks %s | mean blocks %s | reward %s" % (
        os.path.basename(_lib.LIB_PATH), D, (ts[56] - t0) / 100.0, (ts[4] - t0) / 100.0, rng([(x, y) for x in range(55) for y in range(NY)]), rng(spare[:20]), rng(spare[20:21])))
late = sorted(((v[1] - t0) / 100.0, (v[0] - t0) / 100.0, k) for k, v in st.items() if v[1])[-6:]
print("the last workgroups to end (x, y): " + "  ".join("(%d,%d) %.1f..%.1f" % (k[0], k[1], s0, e) for e, s0, k in late))
ts = ctx.debug_timestamps()
us = lambda a, b: (ts[b] - ts[a]) / 100.0
print("mean block (0,0): [gj+stage %.2f rows %.2f wave sums %.2f block sums + store %.2f] = %.2f   (stamps 40..44)" % (us(40, 41), us(41, 42), us(42, 43), us(43, 44), us(40, 44)))
print("pair block (0,0): [init %.2f gj %.2f rows %.2f] " % (us(0, 1), us(1, 2), us(2, 3)))
