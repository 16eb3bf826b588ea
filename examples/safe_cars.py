#!/usr/bin/env python3
"""Safe-PILCO on the two linear cars, the loop of the reference's examples/safe_cars_run.py:41-140 as it is written there:

    env = LinearCars (linear_cars_env.py: two unit masses, Dt = 0.5 s, one bounded force on car 1), states normalised by the
    statistics of 5 random rollouts;  5 x 25 random steps of data;  RbfController(bf=40, max_action=0.2);
    SafePILCO(reward_add=LinearReward on car 1's position, reward_mult=RiskOfCollision (both cars inside the junction),
    mu=-300, horizon=25), likelihood noise fixed at 1e-3;
    5 x [ optimize_models(maxiter=100) if new data;  optimize_policy(maxiter=20, restarts=2);  predicted risk over the
          horizon;  a rollout on the plant;  mu <- 0.75 mu (risk < th/4)  or  1.5 mu (risk >= th) ]

on the MI355X path.  gym is not installed: the plant below restates linear_cars_env.py:7-37 (A, B, initial state, reset
noise) without the gym base class.  What this example exercises beyond inverted_pendulum.py: SafePILCO.predict, the
policy gradient of an objective that is NOT only the additive reward (the risk term's cotangent seeds in the native reverse
sweep, pilco_rollout_grad_rbf_seeded), a fixed (non-trainable) likelihood variance, and `pilco.mu` as a Parameter.

    python examples/safe_cars.py [--iters N]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


class LinearCarsLike:
    """linear_cars_env.py:7-37: x <- x + A x + B u, reward -1 / +1 by the sign of car 1's position."""

    def __init__(self, rs):
        self.rs = rs
        M, b, Dt = 1.0, 0.001, 0.5
        self.A = np.array([[0, Dt, 0, 0], [0, -b * Dt / M, 0, 0], [0, 0, 0, Dt], [0, 0, 0, 0]], dtype=np.float64)
        self.B = np.array([0, Dt / M, 0, 0], dtype=np.float64)
        self.initial_state = np.array([-6.0, 1.0, -5.0, 1.0])
        self.action_low, self.action_high = -0.4, 0.4
        self.state = self.initial_state.copy()

    def reset(self):
        self.state = self.initial_state + 0.03 * self.rs.normal(size=4)
        return self.state.copy()

    def sample_action(self):
        return self.rs.uniform(self.action_low, self.action_high, size=1)

    def step(self, u):
        self.state = self.state + self.A @ self.state + self.B * float(np.ravel(u)[0])
        return self.state.copy(), (-1.0 if self.state[0] < 0 else 1.0), False


class Normalised:
    """safe_cars_run.py:20-39."""

    def __init__(self, env, m, std):
        self.env, self.m, self.std = env, m, std

    def reset(self):
        return (self.env.reset() - self.m) / self.std

    def sample_action(self):
        return self.env.sample_action()

    def step(self, u):
        ob, r, done = self.env.step(u)
        return (ob - self.m) / self.std, r, done


def rollout(env, action_fn, timesteps):
    X, Y, ret = [], [], 0.0
    x = env.reset()
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


def run(iters=5, seed=0, verbose=True):
    from pilco_amd.controllers import RbfController
    from pilco_amd.params import set_trainable
    from pilco_amd.rewards import LinearReward
    from pilco_amd.safe import RiskOfCollision, SafePILCO
    T, th, J = 25, 0.10, 5
    rs = np.random.RandomState(seed)
    np.random.seed(seed)
    raw = LinearCarsLike(rs)
    X1 = np.vstack([rollout(raw, lambda x: raw.sample_action(), T)[0] for _ in range(5)])
    env = Normalised(raw, np.mean(X1[:, :4], 0), np.std(X1[:, :4], 0))
    data = [rollout(env, lambda x: env.sample_action(), T)[:2] for _ in range(J)]
    X, Y = np.vstack([d[0] for d in data]), np.vstack([d[1] for d in data])
    state_dim, control_dim = Y.shape[1], X.shape[1] - Y.shape[1]
    m_init, S_init = X[0:1, :-1], 0.1 * np.eye(state_dim)
    controller = RbfController(state_dim=state_dim, control_dim=control_dim, num_basis_functions=40, max_action=0.2)
    R1 = LinearReward(state_dim, np.array([1.0 * env.std[0], 0.0, 0.0, 0.0]))
    b1, b2 = 1.0 / env.std[0], 1.0 / env.std[2]
    B = RiskOfCollision(2, [-b1 - env.m[0] / env.std[0], -b2 - env.m[2] / env.std[2]], [b1 - env.m[0] / env.std[0], b2 - env.m[2] / env.std[2]])
    pilco = SafePILCO((X, Y), controller=controller, mu=-300.0, reward_add=R1, reward_mult=B, horizon=T, m_init=m_init, S_init=S_init)
    for model in pilco.mgpr.models:
        model.likelihood.variance.assign(0.001)
        set_trainable(model.likelihood.variance, False)
    stages, new_data = [], True
    t_all = time.perf_counter()
    for it in range(iters):
        t0 = time.perf_counter()
        if new_data:
            pilco.optimize_models(maxiter=100, verbose=False)
            new_data = False
        t1 = time.perf_counter()
        pilco.optimize_policy(maxiter=20, restarts=2, verbose=False)
        t2 = time.perf_counter()
        risks = np.zeros(T)
        rewards = np.zeros(T)
        for h in range(T):   # safe_cars_run.py:103-108: the predicted state distribution after h steps
            m_h, S_h, _ = pilco.predict(m_init, S_init, h)
            risks[h] = float(np.ravel(B.compute_reward(m_h, S_h)[0])[0])
            rewards[h] = float(np.ravel(R1.compute_reward(m_h, S_h)[0])[0])
        overall = 1.0 - np.prod(1.0 - risks)
        X_new, Y_new, ret = rollout(env, lambda x: pilco.compute_action(x[None, :])[0, :], T)
        crossed = bool(np.any((np.abs(X_new[:, 0] * env.std[0] + env.m[0]) < 1.0) & (np.abs(X_new[:, 2] * env.std[2] + env.m[2]) < 1.0)))
        mu_before = float(pilco.mu.numpy())
        if overall < th:
            new_data = True
            X, Y = np.vstack((X, X_new)), np.vstack((Y, Y_new))
            pilco.mgpr.set_data((X, Y))
            if overall < th / 4:
                pilco.mu.assign(0.75 * pilco.mu.numpy())
        else:
            pilco.mu.assign(1.5 * pilco.mu.numpy())
        t3 = time.perf_counter()
        stages.append(dict(N=int(X.shape[0]), optimize_models_s=t1 - t0, optimize_policy_s=t2 - t1, risk_check_s=t3 - t2,
                           predicted_return=float(rewards.sum()), predicted_risk=float(overall), mu=mu_before,
                           plant_return=float(ret), both_cars_in_junction=crossed))
        if verbose:
            print("[safe cars] iteration %d: optimize_models %.2f s  optimize_policy(maxiter=20, restarts=2) %.2f s  predicted return %.2f  "
                  "predicted risk %.3g (mu %.0f)  plant return %+.0f  both cars in the junction at once: %s"
                  % (it, t1 - t0, t2 - t1, rewards.sum(), overall, mu_before, ret, crossed))
    return dict(total_s=time.perf_counter() - t_all, iterations=stages)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=5)
    a = ap.parse_args()
    out = run(iters=a.iters)
    print("Safe-PILCO, linear cars, HIP path: total wall-clock %.2f s" % out["total_s"])
