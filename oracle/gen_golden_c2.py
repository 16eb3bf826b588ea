"""Headline-size fixtures by EXECUTING THE REFERENCE'S OWN SOURCE at BASELINE config 2.

TEST INFRASTRUCTURE ONLY.  Run in the build container from the repo root (about 15-30 minutes, ~10 GB):

    python -m oracle.gen_golden_c2

The reference's unmodified PILCO.propagate / reward.compute_reward (pilco/models/pilco.py:138-153,
pilco/rewards.py:19-51, through MGPR.predict_on_noisy_inputs, mgpr.py:77-149 -- including its per-step
re-factorisation and the materialised (E,E,N,N) tensors) are stepped H = 40 times at N=1000, E=10 for

  * C2   D=10 (control_dim 0, the metric read literally) and its sigma_n^2 = 1e-4 stress variant, and
  * C2u  D=11 (one control, LinearController, max_action 1),

with the synthetic inputs of pilco_amd/synthetic.py.  The loop below is the body of the
reference's tf.while_loop (pilco.py:126-135) written out so that every intermediate state can be stored;
PILCO.predict itself is run for n = 2 beside it to show they are the same computation.  Stored: the state
after every step and the running reward -- a few kB per configuration (tests/golden/c2_rollout*.npz).
bench.py checks the (m_H, S_H, reward) it timed against c2_rollout.npz before printing its JSON line.
"""
from __future__ import annotations

import os
import sys
import time

import numpy as np

from pilco_amd import synthetic
from . import ref_exec
from .gen_golden import OUT, PROV, _set_hyp

n_ = ref_exec.to_np


def run(R, tag, D, noise, H=40):
    import torch
    c = synthetic.config_c2(N=1000, D=D, E=10, noise=noise)
    E, U = 10, D - 10
    np.random.seed(1)
    ctl = R.controllers.LinearController(E, U, max_action=1.0)
    pilco = R.PILCO((c["X"], c["Y"]), horizon=H, controller=ctl, m_init=c["m0"], S_init=c["S0"])
    _set_hyp(pilco.mgpr.models, c["lengthscales"], c["variance"], c["noise"])
    ctl.W.assign(c["W"])
    ctl.b.assign(c["b"])
    m, s = c["m0"], c["S0"]
    Ms, Ss, Rs = [n_(m)[0]], [n_(s)], [0.0]
    reward = 0.0
    t0 = time.time()
    with torch.no_grad():
        for t in range(H):
            reward = reward + float(n_(pilco.reward.compute_reward(m, s)[0]).ravel()[0])   # pilco.py:133: pre-propagation state
            m, s = pilco.propagate(m, s)                                                     # pilco.py:132
            Ms.append(n_(m)[0]); Ss.append(n_(s)); Rs.append(reward)
            print(f"[{tag}] step {t + 1}/{H}  reward {reward:.12f}  ({time.time() - t0:.0f} s)", flush=True)
        M2, S2, R2 = pilco.predict(c["m0"], c["S0"], 2)
    np.testing.assert_allclose(n_(M2)[0], Ms[2], rtol=1e-12)
    np.testing.assert_allclose(n_(S2), Ss[2], rtol=1e-12)
    np.testing.assert_allclose(float(n_(R2).ravel()[0]), Rs[2], rtol=1e-12)
    np.savez(os.path.join(OUT, f"c2_rollout{tag}.npz"), provenance=np.array(PROV + "; config: synthetic.config_c2"),
             N=1000, D=D, E=E, H=H, noise=noise, seed=1234, M_traj=np.stack(Ms, 1), S_traj=np.stack(Ss, 2), R_traj=np.array(Rs),
             W=c["W"], b=c["b"], m0=c["m0"], S0=c["S0"])


def run_grad(R, H=5):
    """Full-size gradient: d reward / d (W, b) at C2u (N=1000, D=11, E=10) by reverse mode through the executed
    reference's training_loss (pilco.py:47-50,85-90) -- the (E,E,N,N) tape of every step is kept, hence H = 5 (about 25 GB)."""
    import torch
    c = synthetic.config_c2(N=1000, D=11, E=10)
    np.random.seed(1)
    ctl = R.controllers.LinearController(10, 1, max_action=1.0)
    pilco = R.PILCO((c["X"], c["Y"]), horizon=H, controller=ctl, m_init=c["m0"], S_init=c["S0"])
    _set_hyp(pilco.mgpr.models, c["lengthscales"], c["variance"], c["noise"])
    ctl.W.assign(c["W"])
    ctl.b.assign(c["b"])
    for prm in pilco.mgpr.trainable_parameters:       # optimize_policy freezes the GP (pilco.py:80-82)
        prm.trainable = False
    t0 = time.time()
    loss = pilco.training_loss()
    gW, gb = torch.autograd.grad(loss.sum(), [ctl.W.unconstrained_variable, ctl.b.unconstrained_variable])
    print(f"[c2u_grad] H={H} reward {-float(loss.detach().sum()):.12f} ({time.time() - t0:.0f} s)", flush=True)
    np.savez(os.path.join(OUT, "c2u_grad.npz"), provenance=np.array(PROV + "; config: synthetic.config_c2(D=11); torch autograd through the executed reference"),
             N=1000, D=11, E=10, H=H, reward=-float(loss.detach().sum()), dreward_dW=-gW.numpy(), dreward_db=-gb.numpy(),
             W=c["W"], b=c["b"], m0=c["m0"], S0=c["S0"])


def main():
    R = ref_exec.load()
    which = sys.argv[1:] or ["c2", "c2u", "c2_stress", "c2u_grad"]
    if "c2u_grad" in which:
        run_grad(R)
    if "c2" in which:
        run(R, "", 10, 1e-2)
    if "c2u" in which:
        run(R, "_c2u", 11, 1e-2)
    if "c2_stress" in which:
        run(R, "_stress", 10, 1e-4)


if __name__ == "__main__":
    main()
