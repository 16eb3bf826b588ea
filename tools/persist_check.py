"""Persistent whole-rollout launch vs the launch sequence: bitwise comparison of every state + timing (developer tool)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pilco_amd import _lib, synthetic

def run(N, D, E, H, reps=20, noise=1e-2):
    ctx = _lib.Context()
    c = synthetic.config_c2(N=N, D=D, E=E, noise=noise)
    U = D - E
    if U > 0:
        pol = dict(kind=_lib.POLICY_LINEAR, state_dim=E, control_dim=U, W=c["W"], b=c["b"], max_action=np.ones(U), squash=1)
    else:
        pol = dict(kind=_lib.POLICY_NONE, state_dim=E, control_dim=0)
    rw = [dict(kind=_lib.REWARD_EXPONENTIAL, coef=1.0, W=np.eye(E), t=np.zeros(E))]
    ctx.gp_set_data(0, c["X"], c["Y"]); ctx.gp_set_hyp(0, c["lengthscales"], c["variance"], c["noise"]); ctx.gp_factorize(0)
    out = {}
    for mode in (0, 1, 0, 1):
        ctx.set_rollout_mode(mode)
        r = ctx.rollout(pol, rw, c["m0"], c["S0"], H, want_traj=True)
        used = ctx.last_rollout_mode()
        t0 = time.perf_counter()
        for _ in range(reps):
            ctx.rollout(pol, rw, c["m0"], c["S0"], H)
        ms = (time.perf_counter() - t0) * 1e3 / reps
        out.setdefault(mode, []).append((r, used, ms))
    (r0, u0, ms0), (r1, u1, ms1) = out[0][-1], out[1][-1]
    same = all(np.array_equal(np.asarray(x), np.asarray(y)) for x, y in zip(r0, r1))
    dev = max(float(np.max(np.abs(np.asarray(x) - np.asarray(y)) / (np.abs(np.asarray(y)) + 1e-300))) for x, y in zip(r0, r1))
    rep = all(np.array_equal(np.asarray(x), np.asarray(y)) for x, y in zip(out[1][0][0], out[1][1][0]))
    try:
        ctx.set_rollout_mode(1)
        ctx.rollout(pol, rw, c["m0"], c["S0"], H)
        raw = ctx.debug_buffer(0, 5, 64).view(np.uint64)
        for name, o in (("wg0", 8), ("last item wg", 16), ("last wg", 24)):
            st = raw[o:o + 7].astype(np.int64)
            if st[0]:
                d = (st[1:] - st[:-1]) / 100.0
                print("   %-13s link %.2f | operands %.2f | publish+ready wait %.2f | pairs %.2f | drain+barrier %.2f | flag -> next link start %.2f  (us; step total %.2f)"
                      % (name, d[0], d[1], d[2], d[3], d[4], d[5], (st[6] - st[0]) / 100.0))
    except Exception as exc:
        print("   (no stamps: %r)" % (exc,))
    print("N=%d D=%d E=%d H=%d: modes used %d/%d; bitwise equal %s (max rel dev %.2e); persistent repeatable %s; sequence %.3f ms, persistent %.3f ms (%.1f / %.1f rollouts/s)"
          % (N, D, E, H, u0, u1, same, dev, rep, ms0, ms1, 1e3 / ms0, 1e3 / ms1), flush=True)
    ctx.close()

if __name__ == "__main__":
    cases = [(130, 4, 3, 5), (1000, 10, 10, 40), (1000, 11, 10, 40), (300, 6, 4, 20), (200, 10, 10, 40)]
    if len(sys.argv) > 1:
        cases = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]]
    for cs in cases:
        run(*cs)
