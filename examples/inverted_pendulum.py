#!/usr/bin/env python3
"""BASELINE config 5: the PILCO loop of the reference's examples/inverted_pendulum.py:15-39
(random rollouts -> PILCO(RbfController(bf=10), horizon) -> 3 x [optimize_models, optimize_policy,
rollout, set_data]) on the MI355X path.  gym is not installed, so the pendulum swing-up plant is
simulated here (same equations as gym's Pendulum-v0: state (cos th, sin th, th_dot), torque in [-2, 2],
dt = 0.05).  Prints the wall-clock of every stage; `--quick` is a short smoke variant.

    python examples/inverted_pendulum.py [--quick]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pilco_amd.controllers import RbfController  # noqa: E402
from pilco_amd.models import PILCO  # noqa: E402
from pilco_amd.rewards import ExponentialReward  # noqa: E402


class Pendulum:
    """Pendulum-v0 dynamics (g = 10, m = l = 1, dt = 0.05, max torque 2, max speed 8)."""
    max_torque, max_speed, dt = 2.0, 8.0, 0.05

    def __init__(self, rs):
        self.rs = rs
        self.th, self.thdot = np.pi, 0.0

    def reset(self):
        self.th = np.pi + 0.1 * self.rs.randn()
        self.thdot = 0.1 * self.rs.randn()
        return self.obs()

    def obs(self):
        return np.array([np.cos(self.th), np.sin(self.th), self.thdot])

    def step(self, u):
        u = float(np.clip(u, -self.max_torque, self.max_torque))
        self.thdot = np.clip(self.thdot + (-3 * 10.0 / 2 * np.sin(self.th + np.pi) + 3.0 * u) * self.dt,
                             -self.max_speed, self.max_speed)
        self.th = self.th + self.thdot * self.dt
        return self.obs()


def rollout(env, pilco, timesteps, random=False, rs=None):
    """examples/utils.py:7-29: collect (x, u) -> delta-x pairs."""
    X, Y = [], []
    x = env.reset()
    for _ in range(timesteps):
        u = rs.uniform(-env.max_torque, env.max_torque, size=1) if random else np.asarray(pilco.compute_action(x[None, :])).ravel()
        x_new = env.step(u[0])
        X.append(np.hstack((x, u)))
        Y.append(x_new - x)
        x = x_new
    return np.stack(X), np.stack(Y)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    args = ap.parse_args()
    rs = np.random.RandomState(0)
    np.random.seed(0)
    env = Pendulum(rs)
    T, J, iters, maxiter = (20, 2, 1, 3) if args.quick else (40, 5, 3, 20)
    t_all = time.time()
    X, Y = rollout(env, None, T, random=True, rs=rs)
    for _ in range(1, J):
        X_, Y_ = rollout(env, None, T, random=True, rs=rs)
        X, Y = np.vstack((X, X_)), np.vstack((Y, Y_))
    state_dim, control_dim = Y.shape[1], X.shape[1] - Y.shape[1]
    controller = RbfController(state_dim=state_dim, control_dim=control_dim, num_basis_functions=10, max_action=2.0)
    reward = ExponentialReward(state_dim, W=np.diag([2.0, 0.0, 0.3]) + 1e-9 * np.eye(3), t=np.array([1.0, 0.0, 0.0]))
    m_init = np.array([[-1.0, 0.0, 0.0]])
    S_init = np.diag([0.01, 0.05, 0.01])
    pilco = PILCO((X, Y), controller=controller, horizon=T, reward=reward, m_init=m_init, S_init=S_init)
    for it in range(iters):
        t0 = time.time()
        pilco.optimize_models(verbose=False)
        t1 = time.time()
        r = pilco.optimize_policy(maxiter=maxiter, restarts=1, verbose=False)
        t2 = time.time()
        X_new, Y_new = rollout(env, pilco, T)
        X, Y = np.vstack((X, X_new)), np.vstack((Y, Y_new))
        pilco.mgpr.set_data((X, Y))
        print("iteration %d: N=%d  optimize_models %.2f s  optimize_policy(maxiter=%d) %.2f s  predicted reward %.3f  "
              "realised cos(theta) at end %.2f" % (it, X.shape[0] - T, t1 - t0, maxiter, t2 - t1, r, X_new[-1, 0]))
    print("total wall-clock %.1f s" % (time.time() - t_all))


if __name__ == "__main__":
    main()
