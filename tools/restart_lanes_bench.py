"""optimize_policy(maxiter=50, restarts=3) at config-5 size (N=225, D=5, E=4, horizon 40): the restarts as lanes of one batched
value-and-gradient call per round (default) against the reference's loop, one restart after the other (PILCO_RESTART_LANES=0);
and B value-and-gradient rollouts as ONE pilco_rollout_grad[_rbf]_batch call against B solo calls.  Developer tool."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("PILCO_AB_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # (PILCO_AB_ROOT: a directory holding another build's pilco_amd/)
from pilco_amd import _lib
from pilco_amd.controllers import RbfController, LinearController
from pilco_amd.models import PILCO
rs = np.random.RandomState(0)
X = rs.randn(225, 5) * np.array([0.3, 0.1, 0.5, 0.8, 2.0])
Y = 0.05 * np.stack([np.sin(X @ rs.randn(5)) for _ in range(4)], 1) + 1e-3 * rs.randn(225, 4)
def med(fn, n=15):
    fn(); fn(); ts = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); ts.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(ts))
for name in ("rbf", "linear"):
    res = {}
    for lanes in ("0", "1"):
        os.environ["PILCO_RESTART_LANES"] = lanes
        np.random.seed(0)
        ctl = RbfController(state_dim=4, control_dim=1, num_basis_functions=10, max_action=3.0) if name == "rbf" else LinearController(4, 1, max_action=3.0)
        p = PILCO((X, Y), controller=ctl, horizon=40)
        for m in p.mgpr.models:
            m.kernel.lengthscales.assign(np.array([0.5, 0.3, 1.0, 1.5, 3.0])); m.kernel.variance.assign(0.01); m.likelihood.variance.assign(1e-5)
        p.compute_reward()
        ts = []
        for rep in range(3):
            np.random.seed(1)
            p.controller.randomize()
            t0 = time.perf_counter()
            r = p.optimize_policy(maxiter=50, restarts=3, verbose=False)
            ts.append(time.perf_counter() - t0)
        res[lanes] = (min(ts), r)
    print("config-5 size, %s controller: optimize_policy(maxiter=50, restarts=3): sequential %.3f s (reward %.4f), as lanes %.3f s (reward %.4f): %.2fx" % (
        name, res["0"][0], res["0"][1], res["1"][0], res["1"][1], res["0"][0] / res["1"][0]))
    # raw evaluations
    ctx = p.ctx
    rw, H, E, U = p._reward_terms(), 40, 4, 1
    m0, S0 = np.asarray(p.m_init).reshape(-1), np.asarray(p.S_init)
    for B in (1, 2, 3, 4, 8):
        mm, SS = np.tile(m0, (B, 1)), np.tile(S0, (B, 1, 1))
        if name == "linear":
            pols = [dict(p.controller.policy_spec(True), W=rs.randn(1, 4), b=rs.randn(1)) for _ in range(B)]
            tb = med(lambda: ctx.rollout_grad_batch(pols, rw, mm, SS, H))
            tsolo = med(lambda: [ctx.rollout_grad(pols[i], rw, m0, S0, H) for i in range(B)])
        else:
            bf = 10
            Xp, Yp, lsp, nz = rs.randn(B, bf, E), 0.3 * rs.randn(B, bf, U), 1 + 0.1 * rs.rand(B, U, E), np.full((B, U), 1e-4)
            spec = dict(kind=_lib.POLICY_RBF, state_dim=E, control_dim=U, max_action=3.0, squash=True)
            tb = med(lambda: ctx.rollout_grad_rbf_batch([spec] * B, rw, mm, SS, H, Xp, Yp, lsp, nz))
            ctx.rollout_grad_rbf_batch([spec], rw, mm[:1], SS[:1], H, Xp[:1], Yp[:1], lsp[:1], nz[:1])
            tsolo = B * med(lambda: ctx.rollout_grad_rbf(spec, rw, m0, S0, H, Xp[0], Yp[0], lsp[0], nz[0]))
        print("  B=%d value+gradient: one batch call %.3f ms (%.3f ms per lane), solo calls %.3f ms: %.2fx" % (B, tb, tb / B, tsolo, tsolo / tb))
# benchmark size (C2u: N=1000, state 10 + 1 control): the sweep launches fill the chip by themselves, lanes hide heads and finish only
from pilco_amd import synthetic
c = synthetic.config_c2(N=1000, D=11, E=10)
cx = _lib.Context()
cx.gp_set_data(0, c["X"], c["Y"]); cx.gp_set_hyp(0, c["lengthscales"], c["variance"], c["noise"]); cx.gp_factorize(0)
rw = [dict(kind=_lib.REWARD_EXPONENTIAL, coef=1.0, W=np.eye(10), t=np.zeros(10))]
for B in (1, 2, 3):
    pols = [dict(kind=_lib.POLICY_LINEAR, state_dim=10, control_dim=1, W=c["W"] + 0.05 * rs.randn(1, 10), b=c["b"].ravel(), max_action=1.0, squash=True) for _ in range(B)]
    mm, SS = np.tile(c["m0"].ravel(), (B, 1)), np.tile(c["S0"], (B, 1, 1))
    tb = med(lambda: cx.rollout_grad_batch(pols, rw, mm, SS, 40), 8)
    tsolo = med(lambda: [cx.rollout_grad(pols[i], rw, c["m0"], c["S0"], 40) for i in range(B)], 8)
    print("C2u (N=1000, D=11): B=%d value+gradient: one batch call %.3f ms (%.3f ms per lane), solo calls %.3f ms: %.2fx" % (B, tb, tb / B, tsolo, tsolo / tb))
