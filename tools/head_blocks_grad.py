"""Per-workgroup start / end stamps of the LAST fused head of an eager H=4 rollout at C2u, forward against value-and-gradient
(Jacobian tape): where the head's kernel time goes beyond workgroup (0,0)'s own stamps."""
import numpy as np, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pilco_amd import _lib, synthetic
cfg = synthetic.config_c2(N=1000, D=11, E=10)
ctx = _lib.Context()
ctx.debug_timestamps(read=False)
ctx.gp_set_data(0, cfg["X"], cfg["Y"]); ctx.gp_set_hyp(0, cfg["lengthscales"], cfg["variance"], cfg["noise"]); ctx.gp_factorize(0)
pol = dict(kind=_lib.POLICY_LINEAR, state_dim=10, control_dim=1, W=cfg["W"], b=cfg["b"].ravel(), max_action=1.0, squash=True)
rw = [dict(kind=_lib.REWARD_EXPONENTIAL, coef=1.0, W=np.eye(10), t=np.zeros(10))]
NX, NY = 61, 4
def show(tag):
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
        return "n=%d start %.1f..%.1f end %.1f..%.1f" % (len(v), min(x[0] for x in v), max(x[0] for x in v), min(x[1] for x in v), max(x[1] for x in v)) if v else "none"
    spare = [(55 + i // NY, i % NY) for i in range(24)]
    print("%s  link(0,0) %.1f..%.1f | pair blocks %s | mean blocks %s | reward %s" % (
        tag, (ts[56] - t0) / 100.0, (ts[4] - t0) / 100.0, rng([(x, y) for x in range(55) for y in range(NY)]), rng(spare[:20]), rng(spare[20:21])))
for rep in range(2):
    ctx.rollout(pol, rw, cfg["m0"], cfg["S0"], 4); show("forward")
for rep in range(2):
    ctx.rollout_grad(pol, rw, cfg["m0"], cfg["S0"], 4); show("tape   ")
