"""Is the pair kernel held back by the power limit?  Wave 0's duration and the engine clock it saw (shader-clock ticks over the
100 MHz wall clock) for pair launches issued back to back against launches with an idle gap in front of them: a chip at its
power limit lowers the engine clock under sustained fp64 load, and the same launch after an idle gap runs at a higher clock."""
import numpy as np, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pilco_amd import _lib, synthetic
cfg = synthetic.config_c2(N=1000, D=10, E=10)
ctx = _lib.Context()
ctx.debug_timestamps(read=False)
ctx.gp_set_data(0, cfg["X"], cfg["Y"]); ctx.gp_set_hyp(0, cfg["lengthscales"], cfg["variance"], cfg["noise"]); ctx.gp_factorize(0)
pol = dict(kind=_lib.POLICY_NONE, state_dim=10, control_dim=0)
rw = [dict(kind=_lib.REWARD_EXPONENTIAL, coef=1.0, W=np.eye(10), t=np.zeros(10))]
for H, gap in [(40, 0.0), (40, 0.0), (1, 0.0), (1, 0.002), (1, 0.02), (1, 0.2), (40, 0.0), (40, 0.2)]:
    mhz, dur = [], []
    for rep in range(12):
        if gap: time.sleep(gap)
        ctx.rollout(pol, rw, cfg["m0"], cfg["S0"], H)     # (eager with stamps: the LAST step's pair launch is what the stamps hold)
        ts = ctx.debug_timestamps()
        if ts[17] > ts[16]:
            dur.append((ts[17] - ts[16]) / 100.0)
            mhz.append((ts[33] - ts[32]) / ((ts[17] - ts[16]) / 100.0))
    print("H = %2d, idle gap %5.1f ms in front of each rollout: wave 0 of the last pair launch %.1f us (min %.1f), engine clock %.0f MHz (max %.0f)" % (
        H, gap * 1e3, np.median(dur), min(dur), np.median(mhz), max(mhz)))
