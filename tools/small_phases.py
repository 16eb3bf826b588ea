"""In-kernel phase durations (us) of the one-launch step at config 4 (SMGPR M=200): stamps of workgroup (0,0) of the last head of
an eager H=3 rollout -- link (56..61), operand part (61, 1, 2, 3) and the pair phase that follows it (3 -> 4) -- and the
start / end of every workgroup of that launch."""
import numpy as np, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pilco_amd import _lib, synthetic
c4 = synthetic.config_c4()
ctx = _lib.Context()
ctx.debug_timestamps(read=False)
ctx.gp_set_data(0, c4["X"], c4["Y"]); ctx.gp_set_hyp(0, c4["lengthscales"], c4["variance"], c4["noise"]); ctx.gp_set_inducing(0, c4["Z"]); ctx.gp_factorize(0)
pol = dict(kind=_lib.POLICY_NONE, state_dim=10, control_dim=0)
rw = [dict(kind=_lib.REWARD_EXPONENTIAL, coef=1.0, W=np.eye(10), t=np.zeros(10))]
for on in (1, 0):
    ctx.set_small_step(on)
    for rep in range(3):
        ctx.rollout(pol, rw, c4["m0"], c4["S0"], 3)
        ts = ctx.debug_timestamps()
        b = np.array(ctx.debug_blocks(958), dtype=np.int64).reshape(-1, 2)
        ok = (b[:, 0] > 0) & (b[:, 1] > b[:, 0])
        us = lambda a, c: (ts[c] - ts[a]) / 100.0
        t0 = b[ok, 0].min()
        print("small_step=%d link %.2f | init %.2f gj %.2f rows %.2f | pair phase %.2f | wg(0,0) total %.2f || workgroups %d: start %.1f..%.1f end %.1f..%.1f (p50 %.1f)" % (
            on, us(56, 61), us(61, 1), us(1, 2), us(2, 3), us(3, 4), us(56, 4), int(ok.sum()), 0.0, (b[ok, 0].max() - t0) / 100.0,
            (b[ok, 1].min() - t0) / 100.0, (b[ok, 1].max() - t0) / 100.0, (np.median(b[ok, 1]) - t0) / 100.0))
