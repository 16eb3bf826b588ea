"""In-kernel phase durations (us) of the one-launch step at config-5 size (N=225, D=5, E=4; linear and RBF controller), forward and
value-and-gradient form: stamps of workgroup (0,0) of the last head of an eager H=3 rollout (see tools/small_phases.py)."""
import numpy as np, sys, os
sys.path.insert(0, os.environ.get("PILCO_AB_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # (PILCO_AB_ROOT: a directory holding another build's pilco_amd/)
from pilco_amd import _lib
from pilco_amd.controllers import RbfController, LinearController
from pilco_amd.models import PILCO
rs = np.random.RandomState(0)
X = rs.randn(225, 5) * np.array([0.3, 0.1, 0.5, 0.8, 2.0])
Y = 0.05 * np.stack([np.sin(X @ rs.randn(5)) for _ in range(4)], 1) + 1e-3 * rs.randn(225, 4)
np.random.seed(0)
for name, ctl in (("linear", LinearController(4, 1, max_action=3.0)), ("rbf", RbfController(state_dim=4, control_dim=1, num_basis_functions=10, max_action=3.0))):
    p = PILCO((X, Y), controller=ctl, horizon=3)
    for m in p.mgpr.models:
        m.kernel.lengthscales.assign(np.array([0.5, 0.3, 1.0, 1.5, 3.0])); m.kernel.variance.assign(0.01); m.likelihood.variance.assign(1e-5)
    ctx = p.ctx
    ctx.use_graph(False)
    ctx.debug_timestamps(read=False)
    for what, fn in (("forward", lambda: p.compute_reward()), ("value+gradient", lambda: p.value_and_gradient())):
        for rep in range(3):
            fn()
            ts = ctx.debug_timestamps()
            b = np.array(ctx.debug_blocks(958), dtype=np.int64).reshape(-1, 2)
            ok = (b[:, 0] > 0) & (b[:, 1] > b[:, 0])
            us = lambda a, c: (ts[c] - ts[a]) / 100.0
            t0 = b[ok, 0].min()
        print("%s %s: link %.2f (loads %.2f pack %.2f assemble+propagate %.2f policy+joint %.2f) | init %.2f gj %.2f rows %.2f | pair phase %.2f | wg(0,0) total %.2f || workgroups %d: start ..%.1f end %.1f..%.1f (p50 %.1f)" % (
            name, what, us(56, 61), us(56, 57), us(57, 58), us(58, 60), us(60, 61), us(61, 1), us(1, 2), us(2, 3), us(3, 4), us(56, 4), int(ok.sum()), (b[ok, 0].max() - t0) / 100.0,
            (b[ok, 1].min() - t0) / 100.0, (b[ok, 1].max() - t0) / 100.0, (np.median(b[ok, 1]) - t0) / 100.0))
