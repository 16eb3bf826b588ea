"""A/B of library builds in ONE gpurun call (boxes differ by +-5 %): for the build named by PILCO_LIB -- headline C2 rollout,
C2u forward and value + gradient, config 4 and a config-5-sized model -- wall-clock medians and the O(N^2) kernel's mean
launch duration by HIP events.  One line per measurement, prefixed with the library's file name.
usage: for l in exp/lib_a.so exp/lib_b.so; do PILCO_LIB=$l python tools/ab_libs.py [c2 c2u c4 c5]; done"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pilco_amd import _lib, synthetic
what = set(sys.argv[1:]) or {"c2", "c2u", "c4", "c5"}
tag = os.path.basename(_lib.LIB_PATH)
H = 40
rw10 = [dict(kind=_lib.REWARD_EXPONENTIAL, coef=1.0, W=np.eye(10), t=np.zeros(10))]


def med(fn, reps):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(ts)), float(np.min(ts))


def pair_us(ctx, fn):
    ctx.set_pair_timing(True)
    try:
        fn()
        ms, n = ctx.get_pair_timing()
    finally:
        ctx.set_pair_timing(False)
    return ms * 1e3 / max(n, 1), n


ctx = _lib.Context()
if "c2" in what:
    cfg = synthetic.config_c2()
    ctx.gp_set_data(0, cfg["X"], cfg["Y"]); ctx.gp_set_hyp(0, cfg["lengthscales"], cfg["variance"], cfg["noise"]); ctx.gp_factorize(0)
    pol = dict(kind=_lib.POLICY_NONE, state_dim=10, control_dim=0)
    f = lambda: ctx.rollout(pol, rw10, cfg["m0"], cfg["S0"], H)
    for _ in range(5): f()
    ms, mn = med(f, 60)
    pu, n = pair_us(ctx, f)
    a = f(); b = f()
    print("%s C2   rollout %.4f ms (min %.4f) = %.1f /s | pair %.2f us | rest %.2f us/step | repeat-bitwise %s" % (
        tag, ms, mn, 1e3 / ms, pu, ms * 1e3 / H - pu, all(np.array_equal(x, y) for x, y in zip(a, b))))
    ft = sorted(ctx.factorize_timed(0, 1) for _ in range(12))
    print("%s C2   factorisation %.4f ms (min %.4f)" % (tag, float(np.median(ft)), ft[0]))
if "c2u" in what:
    cfg = synthetic.config_c2(N=1000, D=11, E=10)
    ctx.gp_set_data(0, cfg["X"], cfg["Y"]); ctx.gp_set_hyp(0, cfg["lengthscales"], cfg["variance"], cfg["noise"]); ctx.gp_factorize(0)
    pol = dict(kind=_lib.POLICY_LINEAR, state_dim=10, control_dim=1, W=cfg["W"], b=cfg["b"].ravel(), max_action=1.0, squash=True)
    fwd = lambda: ctx.rollout(pol, rw10, cfg["m0"], cfg["S0"], H)
    grd = lambda: ctx.rollout_grad(pol, rw10, cfg["m0"], cfg["S0"], H)
    for _ in range(3): fwd(); grd()
    f_ms, f_mn = med(fwd, 25)
    g_ms, g_mn = med(grd, 25)
    g1, g2 = grd(), grd()
    same = all(np.array_equal(np.asarray(x), np.asarray(y)) for x, y in zip(g1, g2))
    sw, _ = pair_us(ctx, grd)
    pw, _ = pair_us(ctx, fwd)
    print("%s C2u  fwd %.4f ms (min %.4f) | grad %.4f ms (min %.4f) | ratio %.4f | sweep %.2f us | fwd pair %.2f us | grad rest %.2f us/step | fwd rest %.2f us/step | repeat-bitwise %s" % (
        tag, f_ms, f_mn, g_ms, g_mn, g_ms / f_ms, sw, pw, g_ms * 1e3 / H - sw, f_ms * 1e3 / H - pw, same))
if "c4" in what:
    c4 = synthetic.config_c4()
    ctx.gp_set_data(0, c4["X"], c4["Y"]); ctx.gp_set_hyp(0, c4["lengthscales"], c4["variance"], c4["noise"]); ctx.gp_set_inducing(0, c4["Z"])
    ctx.gp_factorize(0)
    pol = dict(kind=_lib.POLICY_NONE, state_dim=10, control_dim=0)
    f = lambda: ctx.rollout(pol, rw10, c4["m0"], c4["S0"], H)
    for _ in range(5): f()
    ms, mn = med(f, 60)
    ft = sorted(ctx.factorize_timed(0, 1) for _ in range(12))
    print("%s C4   rollout %.4f ms (min %.4f) = %.0f /s = %.2f us/step | FITC factorisation %.4f ms" % (tag, ms, mn, 1e3 / ms, ms * 1e3 / H, float(np.median(ft))))
    ctx.gp_set_inducing(0, None)
if "c5" in what:
    # config-5 size: N = 225, state 4 + 1 control, linear controller (the pendulum loop's model after 5 x 40 + 25 steps)
    cfg = synthetic.config_c2(N=225, D=5, E=4)
    ctx.gp_set_data(0, cfg["X"], cfg["Y"]); ctx.gp_set_hyp(0, cfg["lengthscales"], cfg["variance"], cfg["noise"]); ctx.gp_factorize(0)
    pol = dict(kind=_lib.POLICY_LINEAR, state_dim=4, control_dim=1, W=cfg["W"], b=cfg["b"].ravel(), max_action=1.0, squash=True)
    rw4 = [dict(kind=_lib.REWARD_EXPONENTIAL, coef=1.0, W=np.eye(4), t=np.zeros(4))]
    fwd = lambda: ctx.rollout(pol, rw4, cfg["m0"], cfg["S0"], H)
    grd = lambda: ctx.rollout_grad(pol, rw4, cfg["m0"], cfg["S0"], H)
    for _ in range(3): fwd(); grd()
    f_ms, _ = med(fwd, 40)
    g_ms, _ = med(grd, 40)
    print("%s C5sz fwd %.4f ms = %.2f us/step | grad %.4f ms" % (tag, f_ms, f_ms * 1e3 / H, g_ms))
ctx.close()
