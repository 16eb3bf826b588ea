"""Start, end and hardware id of EVERY wave of the stream-K pair kernel (eager H = 3 rollout at C2; needs a library built with
EXTRA=-DPAIR_WAVE_TRACE, named by PILCO_LIB): where the spread of the waves' ends comes from -- dispatch skew, the SIMD a wave
shares with two others, the XCD, or its place on the cost line (diagonal pairs first)."""
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
NW = 3072
for rep in range(3):
    ctx.rollout(pol, rw, cfg["m0"], cfg["S0"], 3)
    raw = np.array(ctx.debug_blocks(4032 + 3 * NW), dtype=np.uint64)[4032:]
    st, en, hw = raw[:NW].astype(np.float64), raw[NW:2 * NW].astype(np.float64), raw[2 * NW:]
    if not (en > 0).all():
        print("library was not built with -DPAIR_WAVE_TRACE"); break
    t0 = st.min()
    s, e = (st - t0) / 100.0, (en - t0) / 100.0
    hwid = (hw & np.uint64(0xffffffff)).astype(np.int64); xcc = (hw >> np.uint64(32)).astype(np.int64) & 0xf
    simd = (hwid >> 4) & 3; cu = (hwid >> 8) & 15; sh = (hwid >> 12) & 1; se = (hwid >> 13) & 7
    key = ((xcc * 8 + se) * 2 + sh) * 64 + cu * 4 + simd
    print("rep %d: starts %.2f .. %.2f us | ends min %.1f p10 %.1f median %.1f p90 %.1f max %.1f | duration median %.1f" % (
        rep, s.min(), s.max(), e.min(), np.percentile(e, 10), np.median(e), np.percentile(e, 90), e.max(), np.median(e - s)))
    if rep < 2: continue
    uk = np.unique(key)
    cnt = np.array([(key == k).sum() for k in uk])
    print("  %d distinct SIMDs; waves per SIMD: %s" % (len(uk), dict(zip(*np.unique(cnt, return_counts=True)))))
    last = np.array([e[key == k].max() for k in uk]); first = np.array([e[key == k].min() for k in uk])
    print("  per SIMD: LAST wave's end min %.1f median %.1f max %.1f | first-to-last wave of a SIMD: median %.1f max %.1f us" % (
        last.min(), np.median(last), last.max(), np.median(last - first), (last - first).max()))
    cuk = key // 4
    ucu = np.unique(cuk)
    lcu = np.array([e[cuk == k].max() for k in ucu])
    print("  per CU (%d): last end min %.1f median %.1f max %.1f" % (len(ucu), lcu.min(), np.median(lcu), lcu.max()))
    for x in range(8):
        m = xcc == x
        if m.any(): print("  XCD %d: %4d waves, wave index %4d..%4d, ends mean %.1f max %.1f, starts max %.2f" % (x, m.sum(), np.nonzero(m)[0].min(), np.nonzero(m)[0].max(), e[m].mean(), e[m].max(), s[m].max()))
    w = np.arange(NW)
    for lo in range(0, NW, 256):
        m = (w >= lo) & (w < lo + 256)
        print("  waves %4d..%4d (line position): ends mean %.1f max %.1f" % (lo, lo + 255, e[m].mean(), e[m].max()))
    # does a SIMD's end follow from WHICH waves it hosts?  (a wave's line position decides diagonal / off-diagonal steps)
    o = np.argsort(last)
    for k in list(uk[o[:4]]) + list(uk[o[-4:]]):
        m = key == k
        print("  SIMD %5d: waves %s ends %s starts %s" % (k, list(w[m]), np.round(e[m], 1), np.round(s[m], 2)))
