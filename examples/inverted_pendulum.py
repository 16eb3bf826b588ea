#!/usr/bin/env python3
"""BASELINE config 5: the PILCO loop of the reference's examples/inverted_pendulum.py:13-39, as it is written there:

    env = InvertedPendulum-v2;  5 random rollouts x 40 steps;  RbfController(bf=10);  PILCO(..., horizon=40)
    3 x [ optimize_models();  optimize_policy()  (maxiter=50, pilco.py:75);  rollout(100 steps);  set_data ]

on the MI355X path (pilco_amd) and, with --cpu, on the CPU stand-in (oracle/cpu_loop.py) beside it.  gym / MuJoCo are
not installed, so a cart-pole with InvertedPendulum-v2's interface is simulated here: observation (x, theta, x_dot,
theta_dot) with theta = 0 upright, one action in [-3, 3] (gear 100 N), 0.04 s per step (2 x 0.02 s), reset to
uniform(-0.01, 0.01), an episode ends when |theta| > 0.2 (examples/utils.py:7-29 breaks the rollout on `done`).

    python examples/inverted_pendulum.py [--quick] [--cpu] [--cpu-iters K]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


class InvertedPendulumLike:
    """Cart-pole with the observation / action / termination conventions of gym's InvertedPendulum-v2."""
    mc, mp, l, g, gear, dt, substeps = 10.5, 5.0, 0.3, 9.81, 100.0, 0.02, 2

    def __init__(self, rs):
        self.rs = rs
        self.q = np.zeros(4)
        self.action_low, self.action_high = -3.0, 3.0

    def reset(self):
        self.q = self.rs.uniform(-0.01, 0.01, size=4)
        return self.q.copy()

    def sample_action(self):
        return self.rs.uniform(self.action_low, self.action_high, size=1)

    def _deriv(self, q, f):
        x, th, xd, thd = q
        s, c = np.sin(th), np.cos(th)
        tot = self.mc + self.mp
        tmp = (f + self.mp * self.l * thd * thd * s) / tot
        thdd = (self.g * s - c * tmp) / (self.l * (4.0 / 3.0 - self.mp * c * c / tot))
        xdd = tmp - self.mp * self.l * thdd * c / tot
        return np.array([xd, thd, xdd, thdd])

    def step(self, u):
        f = self.gear * float(np.clip(np.ravel(u)[0], self.action_low, self.action_high))
        for _ in range(self.substeps):   # RK4
            k1 = self._deriv(self.q, f)
            k2 = self._deriv(self.q + 0.5 * self.dt * k1, f)
            k3 = self._deriv(self.q + 0.5 * self.dt * k2, f)
            k4 = self._deriv(self.q + self.dt * k3, f)
            self.q = self.q + self.dt / 6.0 * (k1 + 2 * k2 + 2 * k3 + k4)
        done = (not np.all(np.isfinite(self.q))) or abs(self.q[1]) > 0.2
        return self.q.copy(), 1.0, done


def rollout(env, action_fn, timesteps):
    """examples/utils.py:7-29: (x, u) -> delta-x pairs; stops when the episode is done."""
    X, Y = [], []
    x = env.reset()
    ret = 0.0
    for _ in range(timesteps):
        u = np.asarray(action_fn(x), np.float64).ravel()
        x_new, r, done = env.step(u)
        ret += r
        X.append(np.hstack((x, u)))
        Y.append(x_new - x)
        x = x_new
        if done:
            break
    return np.stack(X), np.stack(Y), ret


def initial_data(seed, J, T):
    rs = np.random.RandomState(seed)
    env = InvertedPendulumLike(rs)
    X, Y, _ = rollout(env, lambda x: env.sample_action(), T)
    for _ in range(1, J):
        X_, Y_, _ = rollout(env, lambda x: env.sample_action(), T)
        X, Y = np.vstack((X, X_)), np.vstack((Y, Y_))
    return env, X, Y


def run_hip(J=5, T=40, iters=3, maxiter=50, rollout_steps=100, seed=0, verbose=True):
    """The loop on the HIP path.  -> dict of stage timings."""
    from pilco_amd.controllers import RbfController
    from pilco_amd.models import PILCO
    np.random.seed(seed)
    env, X, Y = initial_data(seed, J, T)
    state_dim, control_dim = Y.shape[1], X.shape[1] - Y.shape[1]
    controller = RbfController(state_dim=state_dim, control_dim=control_dim, num_basis_functions=10, max_action=3.0)
    pilco = PILCO((X, Y), controller=controller, horizon=40)     # examples/inverted_pendulum.py:24-27: defaults otherwise
    stages = []
    # one-time cost of the process, not of the loop: HIP runtime start-up, code objects of the library, the context's buffers
    # (TensorFlow pays its counterpart when the reference's script builds its first gpflow model); timed and reported on its own
    t_init = time.perf_counter()
    pilco.ctx
    init_s = time.perf_counter() - t_init
    t_all = time.perf_counter()
    for it in range(iters):
        t0 = time.perf_counter()
        pilco.optimize_models(verbose=False)
        t1 = time.perf_counter()
        r = pilco.optimize_policy(maxiter=maxiter, restarts=1, verbose=False)
        t2 = time.perf_counter()
        X_new, Y_new, ret = rollout(env, lambda x: pilco.compute_action(x[None, :]), rollout_steps)
        X, Y = np.vstack((X, X_new)), np.vstack((Y, Y_new))
        pilco.mgpr.set_data((X, Y))
        stages.append(dict(N=int(X.shape[0] - X_new.shape[0]), optimize_models_s=t1 - t0, optimize_policy_s=t2 - t1,
                           predicted_reward=float(r), steps_balanced=int(X_new.shape[0])))
        if verbose:
            print("[hip] iteration %d: N=%d  optimize_models %.2f s  optimize_policy(maxiter=%d) %.2f s  predicted reward %.3f  "
                  "pole kept up for %d/%d steps" % (it, stages[-1]["N"], t1 - t0, maxiter, t2 - t1, r, X_new.shape[0], rollout_steps))
    return dict(total_s=time.perf_counter() - t_all, init_s=init_s, iterations=stages)


def run_cpu(J=5, T=40, iters=3, maxiter=50, rollout_steps=100, seed=0, verbose=True):
    """The same loop on the CPU stand-in (oracle/cpu_loop.py: NumPy / SciPy / torch-CPU autograd)."""
    from oracle import cpu_loop
    np.random.seed(seed)
    env, X, Y = initial_data(seed, J, T)
    E, U = Y.shape[1], X.shape[1] - Y.shape[1]
    D = E + U
    ls, var, nz = np.ones((E, D)), np.ones(E), np.ones(E)
    Xp, Yp, lsp = np.random.randn(10, E), 0.1 * np.random.randn(10, U), np.ones((U, E))   # controllers.py:87-90
    m_init, S_init = X[0:1, :E], 0.1 * np.eye(E)                                          # pilco.py:37-41
    stages = []
    t_all = time.perf_counter()
    for it in range(iters):
        ls, var, nz, tm = cpu_loop.optimize_models(X, Y, ls, var, nz, restarts=1)
        Xp, Yp, lsp, r, tpol = cpu_loop.optimize_policy(X, Y, ls, var, nz, Xp, Yp, lsp, m_init, S_init, 40, 3.0, maxiter=maxiter)
        X_new, Y_new, ret = rollout(env, lambda x: cpu_loop.compute_action(x, Xp, Yp, lsp, 3.0), rollout_steps)
        stages.append(dict(N=int(X.shape[0]), optimize_models_s=tm, optimize_policy_s=tpol, predicted_reward=float(r),
                           steps_balanced=int(X_new.shape[0])))
        X, Y = np.vstack((X, X_new)), np.vstack((Y, Y_new))
        if verbose:
            print("[cpu] iteration %d: N=%d  optimize_models %.2f s  optimize_policy(maxiter=%d) %.2f s  predicted reward %.3f  "
                  "pole kept up for %d/%d steps" % (it, stages[-1]["N"], tm, maxiter, tpol, r, X_new.shape[0], rollout_steps))
    return dict(total_s=time.perf_counter() - t_all, iterations=stages)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true", help="short smoke variant (2 x 20 random steps, 1 iteration, maxiter 3)")
    ap.add_argument("--cpu", action="store_true", help="also run the CPU stand-in (oracle/cpu_loop.py)")
    ap.add_argument("--cpu-iters", type=int, default=None)
    args = ap.parse_args()
    kw = dict(J=2, T=20, iters=1, maxiter=3, rollout_steps=20) if args.quick else {}
    res = run_hip(**kw)
    print("HIP path: total wall-clock %.2f s (+ %.2f s one-time runtime / library start-up before the loop)" % (res["total_s"], res["init_s"]))
    if args.cpu:
        kc = dict(kw)
        if args.cpu_iters is not None:
            kc["iters"] = args.cpu_iters
        rc = run_cpu(**kc)
        print("CPU stand-in (NumPy/SciPy + torch-CPU autograd, %d host threads): total wall-clock %.2f s" % (os.cpu_count(), rc["total_s"]))


if __name__ == "__main__":
    main()
