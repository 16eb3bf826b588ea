"""developer probe: pilco_gp_predict_vjp against finite differences of the device forward pass over a grid of (D, E)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pilco_amd import synthetic, _lib
from pilco_amd.models import MGPR
ctx = _lib.get_context()
for D in [14, 15, 16, 17, 18, 19, 20, 22, 23, 24, 26, 27, 28, 30, 31, 32]:
    for E in [2, 3]:
        N = 70
        c = synthetic.config_c2(N=N, D=D, E=E, noise=1e-2, seed=N, control_dim=max(D - E, 0))
        mg = MGPR((c["X"], c["Y"]))
        for i, mdl in enumerate(mg.models):
            mdl.kernel.lengthscales.assign(c["lengthscales"][i]); mdl.kernel.variance.assign(c["variance"][i]); mdl.likelihood.variance.assign(c["noise"][i])
        rs = np.random.RandomState(1)
        m = 0.2 * rs.randn(1, D); A = 0.3 * rs.randn(D, D); s = A @ A.T + 0.05 * np.eye(D)
        Mbar, Sbar, Vbar = rs.randn(1, E), rs.randn(E, E), rs.randn(D, E)
        mg._ensure_factorized()
        res = {}
        for name, (Mb, Sb, Vb) in {"all": (Mbar, Sbar, Vbar), "M": (Mbar, 0 * Sbar, 0 * Vbar), "V": (0 * Mbar, 0 * Sbar, Vbar),
                                   "Sdiag": (0 * Mbar, np.diag(np.diag(Sbar)), 0 * Vbar), "Soff": (0 * Mbar, Sbar - np.diag(np.diag(Sbar)), 0 * Vbar)}.items():
            mbar, sbar = ctx.gp_predict_vjp(0, m, s, Mb, Sb, Vb, D, E)
            dm = rs.randn(1, D); dS = rs.randn(D, D); dS = dS + dS.T; h = 1e-6
            def phi(mm, ss):
                M, S, V = ctx.gp_predict(0, mm, ss, D, E)
                return (Mb * M).sum() + (Sb * S).sum() + (Vb * V).sum()
            fd = (phi(m + h * dm, s + h * dS) - phi(m - h * dm, s - h * dS)) / (2 * h)
            an = (mbar * dm).sum() + (sbar * dS).sum()
            res[name] = abs(an - fd) / max(abs(fd), 1e-12)
        print("D=%d E=%d " % (D, E) + " ".join("%s:%.1e" % (k, v) for k, v in res.items()), flush=True)
