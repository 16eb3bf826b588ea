"""Every device call the Python layer makes (pilco_amd/models, controllers, rewards, safe, training), answered on
the CPU by the oracle: TEST INFRASTRUCTURE that lets the CPU suite drive the product's HOST logic end to end -- whole
example loops included -- on a box without a GPU.  Installed with _lib.set_context() by tests only; nothing in pilco_amd/
knows it exists, and it is no fallback: the product raises without libpilco_hip.so and a GPU
(tests/test_sharding_cpu.py::test_no_cpu_fallback_without_gpu)."""
import numpy as np
import torch

from oracle import torch_path as tq
from pilco_amd import _lib

from .cpu_objective_context import CpuObjectiveContext
from .cpu_rollout_context import CpuRolloutContext, T


class CpuStandInContext(CpuRolloutContext):
    # -- training objectives of the model in `slot`
    def _objective(self, slot):
        s, o = self.slots[slot], CpuObjectiveContext()
        o.X, o.Y, o.ls, o.var, o.nz = s["X"], s["Y"], s["ls"], s["var"], s["nz"]
        return o

    def gp_nlml(self, slot, D, E, want_grad=True):
        return self._objective(slot).gp_nlml(slot, D, E, want_grad)

    def gp_fitc_nlml(self, slot, Z_all, D, E, want_grad=True):
        return self._objective(slot).gp_fitc_nlml(slot, Z_all, D, E, want_grad)

    # -- single evaluations
    def gp_predict(self, slot, m, s, D, E):
        assert slot == _lib.SLOT_DYNAMICS
        with torch.no_grad():
            M, S, V = self._dynamics()(T(np.reshape(m, (1, D))), T(np.reshape(s, (D, D))))
        return M.numpy(), S.numpy(), V.numpy()

    def _controller(self, policy):
        params = self._params(policy)
        E = policy["state_dim"]
        if policy["kind"] == _lib.POLICY_NONE:
            return lambda m, s: (torch.zeros((1, 0), dtype=tq.DT), torch.zeros((0, 0), dtype=tq.DT), torch.zeros((E, 0), dtype=tq.DT))
        e = T(np.broadcast_to(np.asarray(policy.get("max_action", 1.0), np.float64).reshape(-1), (policy["control_dim"],)).copy())
        if policy["kind"] == _lib.POLICY_LINEAR:
            return lambda m, s: tq.linear_controller(m, s, params[0], params[1], e, policy.get("squash", True))
        return lambda m, s: tq.rbf_controller(m, s, params[0], params[1], params[2], params[3], e, policy.get("squash", True))

    def policy_action(self, policy, m, s):
        E = policy["state_dim"]
        with torch.no_grad():
            M, S, V = self._controller(policy)(T(np.reshape(m, (1, E))), T(np.reshape(s, (E, E))))
        return M.numpy(), S.numpy(), V.numpy()

    def propagate(self, policy, m_x, s_x):
        E = policy["state_dim"]
        with torch.no_grad():
            M, S = tq.propagate(self._dynamics(), self._controller(policy), T(np.reshape(m_x, (1, E))), T(np.reshape(s_x, (E, E))))
        return M.numpy(), S.numpy()

    def reward_eval(self, terms, E, m, s):
        """(mean, variance) of the reward at N(m, s): rewards.py:19-51 (exponential), :53-61 (linear), :64-81 (combined)."""
        m, s = T(np.reshape(m, (1, E))), T(np.reshape(s, (E, E)))
        mu, var = 0.0, 0.0
        with torch.no_grad():
            for t in terms:
                if t["kind"] == _lib.REWARD_EXPONENTIAL:
                    W = T(t["W"]).reshape(E, E)
                    r1 = float(tq.exponential_reward(m, s, W, t["t"]))
                    eye = torch.eye(E, dtype=tq.DT)
                    d = m - T(t["t"]).reshape(1, E)
                    i2 = torch.linalg.solve((eye + 2 * s @ W).T, W.T).T
                    r2 = float(torch.exp(-d @ i2 @ d.T) / torch.sqrt(torch.linalg.det(eye + 2 * s @ W)))
                    mu, var = mu + t["coef"] * r1, var + t["coef"] ** 2 * (r2 - r1 * r1)
                else:
                    w = T(t["W"]).reshape(E, 1)
                    mu, var = mu + t["coef"] * float(m @ w), var + t["coef"] ** 2 * float(w.T @ s @ w)
        return np.array([[mu]]), np.array([[var]])
