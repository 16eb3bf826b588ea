"""BASELINE config 4 at its full size, by EXECUTING THE REFERENCE'S OWN SOURCE.

TEST INFRASTRUCTURE ONLY.  Run in the build container from the repo root (a few minutes, ~2 GB):

    python -m oracle.gen_golden_c4

The reference's unmodified SMGPR (pilco/models/smgpr.py:24-52: FITC factorisation, Z of model 0 for every output)
inside its unmodified PILCO (pilco/models/pilco.py:27-32 picks SMGPR when num_induced_points is given) at
M = 200 inducing points, N = 5000, D = 10, E = 10 with the synthetic inputs of pilco_amd/synthetic.config_c4
(SURVEY.md 8d): the factors (iK, beta) of calculate_factorizations, one predict_on_noisy_inputs and the H = 40
rollout (state after every step + running reward; the loop is the body of the reference's tf.while_loop,
pilco.py:126-135, written out, with PILCO.predict run for n = 2 beside it).

iK is (10, 200, 200) = 3.2 MB; the fixture keeps beta in full and pins iK through its diagonal, its Frobenius norm and
its product with 4 seeded probe vectors per output (a wrong entry anywhere moves a probe) -- tests/golden/c4_sparse.npz
stays at ~100 kB.
"""
from __future__ import annotations

import os
import time

import numpy as np

from pilco_amd import synthetic
from . import ref_exec
from .gen_golden import OUT, PROV, _set_hyp

n_ = ref_exec.to_np


def probes(M, E, k=4, seed=404):
    return np.random.RandomState(seed).randn(E, M, k)


def main(H=40):
    import torch
    R = ref_exec.load()
    c = synthetic.config_c4()
    N, D = c["X"].shape
    E, M = c["Y"].shape[1], c["Z"].shape[0]
    np.random.seed(1)
    ctl = R.controllers.LinearController(E, D - E, max_action=1.0)
    pilco = R.PILCO((c["X"], c["Y"]), num_induced_points=M, horizon=H, controller=ctl, m_init=c["m0"], S_init=c["S0"])
    assert type(pilco.mgpr).__name__ == "SMGPR"
    _set_hyp(pilco.mgpr.models, c["lengthscales"], c["variance"], c["noise"])
    for mdl in pilco.mgpr.models:
        mdl.inducing_variable.Z.assign(c["Z"])
    t0 = time.time()
    with torch.no_grad():
        iK, beta = pilco.mgpr.calculate_factorizations()
        iK, beta = n_(iK), n_(beta)
        print(f"[c4] factorisation {time.time() - t0:.1f} s", flush=True)
        M1, S1, V1 = pilco.mgpr.predict_on_noisy_inputs(c["m0"], c["S0"])
        m, s = c["m0"], c["S0"]
        Ms, Ss, Rs = [n_(m)[0]], [n_(s)], [0.0]
        reward = 0.0
        for t in range(H):
            reward = reward + float(n_(pilco.reward.compute_reward(m, s)[0]).ravel()[0])   # pilco.py:133
            m, s = pilco.propagate(m, s)                                                     # pilco.py:132
            Ms.append(n_(m)[0]); Ss.append(n_(s)); Rs.append(reward)
            print(f"[c4] step {t + 1}/{H}  reward {reward:.12f}  ({time.time() - t0:.0f} s)", flush=True)
        M2, S2, R2 = pilco.predict(c["m0"], c["S0"], 2)
    np.testing.assert_allclose(n_(M2)[0], Ms[2], rtol=1e-12)
    np.testing.assert_allclose(n_(S2), Ss[2], rtol=1e-12)
    np.testing.assert_allclose(float(n_(R2).ravel()[0]), Rs[2], rtol=1e-12)
    P = probes(M, E)
    np.savez(os.path.join(OUT, "c4_sparse.npz"),
             provenance=np.array(PROV + "; config: synthetic.config_c4 (M=200, N=5000, D=10, E=10)"),
             N=N, D=D, E=E, M=M, H=H, seed=1234,
             beta=beta, iK_diag=np.stack([np.diag(iK[a]) for a in range(E)]), iK_fro=np.array([np.linalg.norm(iK[a]) for a in range(E)]),
             iK_probe=np.einsum("aij,ajk->aik", iK, P), probe_seed=404,
             pred_M=n_(M1), pred_S=n_(S1), pred_V=n_(V1),
             M_traj=np.stack(Ms, 1), S_traj=np.stack(Ss, 2), R_traj=np.array(Rs), m0=c["m0"], S0=c["S0"])
    print(f"[c4] done in {time.time() - t0:.0f} s")


if __name__ == "__main__":
    main()
