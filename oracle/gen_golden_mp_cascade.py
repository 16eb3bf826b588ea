"""Writes tests/golden/cascade_trained_mp.npz: the H = 10 rollout of the reference's literal tests/test_cascade.py
procedure (the trained models and policy stored in cascade_trained.npz by oracle/gen_golden.py) evaluated with EVERY
operation in 40-digit arithmetic (oracle/mp_truth.cascade).  TEST INFRASTRUCTURE.

At GPflow's noise floor (1e-6, cond(K) ~ 1e9) float64 evaluations of this trajectory scatter by ~1e-5 relative in S; the
40-digit trajectory says which one is right, so the HIP path is held to max(1e-5, the executed reference's own distance
from it) instead of a bare tolerance (tests/test_gpu_parity.py::test_cascade_trained_golden).

    python -m oracle.gen_golden_mp_cascade          # ~10 min on one core (N = 100, E = 2, D = 3, H = 10)
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import mp_truth  # noqa: E402


def main():
    gdir = os.path.join(ROOT, "tests", "golden")
    g = np.load(os.path.join(gdir, "cascade_trained.npz"))
    H = int(g["horizon"])
    t0 = time.time()
    M, S, R = mp_truth.cascade(g["X"], g["Y"], g["lengthscales"], g["variance"], g["noise"], g["W"], g["b"], g["max_action"],
                               g["m"], g["s"], H, dps=40)
    rel = lambda a, b: float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300)))
    print("40-digit cascade: %.0f s" % (time.time() - t0))
    print("executed reference vs truth: M %.2e  S %.2e  R %.2e" % (rel(g["M_traj"][:, -1], M[:, -1]), rel(g["S_traj"][:, :, -1], S[:, :, -1]),
                                                                   rel(g["R_traj"][-1:], R[-1:])))
    print("MATLAB route       vs truth: M %.2e  S %.2e" % (rel(g["M_traj_matlab"][:, -1], M[:, -1]), rel(g["S_traj_matlab"][:, :, -1], S[:, :, -1])))
    np.savez(os.path.join(gdir, "cascade_trained_mp.npz"), M_traj_mp=M, S_traj_mp=S, R_traj_mp=R, dps=40, horizon=H)


if __name__ == "__main__":
    main()
