"""Runs one of the reference's own example scripts (/root/reference/examples/*.py), UNMODIFIED, against the product:
`pilco` -> pilco_amd, `gpflow.set_trainable` -> pilco_amd.set_trainable, `gym.make(...)` -> the built-in plant of
examples/inverted_pendulum.py behind gym's interface (gym / MuJoCo are not installed), `pdb.set_trace` (which the
reference's script calls inside its loop) -> nothing.

    python tests/helpers/run_reference_example.py inverted_pendulum.py [--standin]

--standin answers the device calls with the oracle (a box without a GPU: ~3.5 min for the three PILCO iterations); without it
the HIP path answers (seconds).  Prints the final data-set size, the predicted reward and the return of the last real rollout.
TEST INFRASTRUCTURE; own process because it aliases module names."""
import contextlib
import io
import os
import runpy
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "examples")):
    if p not in sys.path:
        sys.path.insert(0, p)
REF_EXAMPLES = "/root/reference/examples"

import numpy as np  # noqa: E402


def main():
    name = sys.argv[1]
    from helpers import run_reference_tests as rr
    from pilco_amd import _lib
    if "--standin" in sys.argv:
        from helpers.cpu_standin_context import CpuStandInContext
        _lib.set_context(CpuStandInContext())
    rr.install_aliases()
    import pilco_amd
    sys.modules["gpflow"].set_trainable = pilco_amd.set_trainable
    import inverted_pendulum as builtin

    class _Space:
        def __init__(self, plant):
            self.plant = plant

        def sample(self):
            return self.plant.sample_action()

    class _Env:
        def __init__(self, env_id):
            self.plant = builtin.InvertedPendulumLike(np.random.RandomState(0))
            self.action_space, self.observation_space, self.env = _Space(self.plant), None, self

        def reset(self):
            return self.plant.reset()

        def step(self, u):
            x, r, done = self.plant.step(u)
            return x, r, done, {}

        def render(self):
            pass

        def close(self):
            pass

    class _MountainCar:
        """gym's MountainCarContinuous-v0 (classic control: a car in a valley, force in [-1, 1], goal at x >= 0.45), restated
        from its published equations; `.env` is the unwrapped environment the reference's Normalised_Env asks for."""

        def __init__(self):
            self.rs = np.random.RandomState(0)
            self.action_space = _Box(-1.0, 1.0, shape=(1,))
            self.observation_space = _Box(np.array([-1.2, -0.07]), np.array([0.6, 0.07]), shape=(2,))
            self.env = self
            self.state = np.array([-0.5, 0.0])

        def reset(self):
            self.state = np.array([self.rs.uniform(-0.6, -0.4), 0.0])
            return self.state.copy()

        def step(self, u):
            x, v = self.state
            f = float(np.clip(np.ravel(u)[0], -1.0, 1.0))
            v = float(np.clip(v + f * 0.0015 - 0.0025 * np.cos(3 * x), -0.07, 0.07))
            x = float(np.clip(x + v, -1.2, 0.6))
            if x == -1.2 and v < 0:
                v = 0.0
            done = bool(x >= 0.45)
            self.state = np.array([x, v])
            return self.state.copy(), (100.0 if done else 0.0) - 0.1 * f * f, done, {}

        def render(self):
            pass

        def close(self):
            pass

    class _Pendulum:
        """gym's Pendulum-v0 (classic control: torque-limited pendulum, observation [cos th, sin th, th_dot]) restated from its
        published equations, with the attributes the reference's myPendulum wrapper reaches into (state, last_u, _get_obs)."""
        max_speed, max_torque, dt, g, m, l = 8.0, 2.0, 0.05, 10.0, 1.0, 1.0

        def __init__(self):
            self.action_space = _Box(-self.max_torque, self.max_torque, shape=(1,))
            self.observation_space = _Box(np.array([-1.0, -1.0, -self.max_speed]), np.array([1.0, 1.0, self.max_speed]), shape=(3,))
            self.env, self.state, self.last_u = self, np.zeros(2), None

        def _get_obs(self):
            th, thd = self.state
            return np.array([np.cos(th), np.sin(th), thd])

        def reset(self):
            self.state = np.random.uniform(low=-np.array([np.pi, 1.0]), high=np.array([np.pi, 1.0]))
            self.last_u = None
            return self._get_obs()

        def step(self, u):
            th, thd = self.state
            u = float(np.clip(np.ravel(u)[0], -self.max_torque, self.max_torque))
            self.last_u = u
            cost = (((th + np.pi) % (2 * np.pi)) - np.pi) ** 2 + 0.1 * thd ** 2 + 0.001 * u ** 2
            thd = thd + (-3 * self.g / (2 * self.l) * np.sin(th + np.pi) + 3.0 / (self.m * self.l ** 2) * u) * self.dt
            th = th + thd * self.dt
            self.state = np.array([th, float(np.clip(thd, -self.max_speed, self.max_speed))])
            return self._get_obs(), -cost, False, {}

        def render(self):
            pass

        def close(self):
            pass

    class _Box:      # gym.spaces.Box as linear_cars_env.py:7-9 uses it
        def __init__(self, low, high, shape=None, dtype=None):
            self.low, self.high, self.shape = np.broadcast_to(low, shape).astype(float), np.broadcast_to(high, shape).astype(float), shape

        def sample(self):
            return np.random.uniform(self.low, self.high)

    class _GymEnv:   # gym.core.Env: a plain base class for the reference's own LinearCars
        pass

    gym = types.ModuleType("gym")
    gym.make = lambda env_id: (_MountainCar() if "MountainCar" in env_id else _Pendulum() if env_id.startswith("Pendulum") else _Env(env_id))
    gym.spaces = types.ModuleType("gym.spaces")
    gym.spaces.Box = _Box
    gym.core = types.ModuleType("gym.core")
    gym.core.Env = gym.Env = _GymEnv
    sys.modules.update({"gym": gym, "gym.spaces": gym.spaces, "gym.core": gym.core})
    sys.modules["gpflow"].config.default_int = lambda: np.int32
    import pilco_amd.safe_pilco_extension as spe
    sys.modules.update({"safe_pilco_extension": spe, "safe_pilco_extension.safe_pilco": spe.safe_pilco,
                        "safe_pilco_extension.rewards_safe": spe.rewards_safe})
    import pdb
    pdb.set_trace = lambda *a, **k: None
    sys.path.insert(0, REF_EXAMPLES)       # the script does `from utils import rollout, policy`
    os.chdir(REF_EXAMPLES)
    out = io.StringIO()
    with contextlib.redirect_stdout(out):
        g = runpy.run_path(os.path.join(REF_EXAMPLES, name), run_name="__main__")
    text = out.getvalue()
    returns = [float(l.split(":")[1]) for l in text.splitlines() if l.startswith("Return so far")]
    if "pilco" in g:
        p = g["pilco"]
        print("RESULT script=%s N=%d predicted_reward=%.6f last_rollout_return=%.1f"
              % (name, p.mgpr.num_datapoints, float(np.ravel(p.compute_reward())[0]), returns[-1]), flush=True)
    else:            # safe_cars_run.py keeps everything inside safe_cars(): report what it printed
        risks = [float(l.split()[-1]) for l in text.splitlines() if l.startswith("Overall risk")]
        mus = [float(l.split()[-1]) for l in text.splitlines() if l.startswith("Mu is")]
        print("RESULT script=%s iterations=%d risks=%s mus=%s" % (name, len(risks), ",".join("%.4g" % r for r in risks),
                                                                    ",".join("%.4g" % m for m in mus)), flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
