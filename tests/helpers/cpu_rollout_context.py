"""A CPU stand-in for the device calls PILCO.optimize_policy makes (rollout value, value + policy gradient), so that the HOST
side of the policy optimisation -- parameter packing, the softplus transform of the RBF lengthscales with its lower bound,
sign conventions, L-BFGS-B options, restart bookkeeping (pilco_amd/training.py, models/pilco.py, controllers.py) -- can be held to
the executed reference's end points in the CPU suite.  TEST INFRASTRUCTURE: values and gradients come from the torch
restatement of the rollout (oracle/torch_path.py, autograd in the role of TF's reverse mode); the product computes them on the
device (pilco_rollout, pilco_rollout_grad*)."""
import numpy as np
import torch

from oracle import torch_path as tq
from pilco_amd import _lib

T = tq.t


class CpuRolloutContext:
    def __init__(self):
        self._slot_owner = {}
        self.slots = {}
        self.grad_calls = 0

    # -- model upload (slot 0: dynamics, slot 1: the RBF policy's GP -- only its data matter here)
    def gp_set_data(self, slot, X, Y, owner=None):
        self._slot_owner[slot] = owner
        self.slots.setdefault(slot, {}).update(X=np.array(X, np.float64), Y=np.array(Y, np.float64))

    def gp_set_hyp(self, slot, lengthscales, variance, noise, owner=None):
        self._slot_owner[slot] = owner
        self.slots.setdefault(slot, {}).update(ls=np.array(lengthscales, np.float64), var=np.ravel(variance).astype(np.float64),
                                               nz=np.ravel(noise).astype(np.float64))

    def gp_set_inducing(self, slot, Z, owner=None):
        self._slot_owner[slot] = owner
        self.slots.setdefault(slot, {})["Z"] = None if Z is None else np.array(Z, np.float64)

    def gp_factorize(self, slot):
        self.slots.setdefault(slot, {}).pop("user", None)      # a factorisation of the model's own replaces caller-supplied factors

    def gp_set_factors(self, slot, iK, beta, owner=None):
        self._slot_owner[slot] = owner
        n = np.asarray(beta).shape[1]
        self.slots.setdefault(slot, {})["user"] = (np.zeros((np.asarray(beta).shape[0], n, n)) if iK is None else np.array(iK, np.float64),
                                                   np.array(beta, np.float64))

    def _factors(self, slot):
        """(inputs, iK, beta) of the model in `slot` as NumPy arrays: caller-supplied, FITC or exact."""
        from oracle import tf_path as tp
        s = self.slots[slot]
        pts = s["Z"] if s.get("Z") is not None else s["X"]
        if "user" in s:
            return pts, s["user"][0], s["user"][1]
        if s.get("Z") is not None:
            return (pts,) + tuple(tp.fitc_factorizations(s["X"], s["Y"], s["Z"], s["ls"], s["var"], s["nz"]))
        return (pts,) + tuple(tp.calculate_factorizations(s["X"], s["Y"], s["ls"], s["var"], s["nz"]))

    def gp_get_factors(self, slot, E, want_iK=True):
        _, iK, beta = self._factors(slot)
        return (iK if want_iK else None), beta

    def gp_num_points(self, slot):
        return self._factors(slot)[0].shape[0]

    def gp_gram(self, slot, X1, X2, E):
        from oracle import tf_path as tp
        s = self.slots[slot]
        return tp.se_ard_K(np.asarray(X1, np.float64), None if X2 is None else np.asarray(X2, np.float64), s["ls"], s["var"])

    # launch-structure switches of the device path: nothing to switch here
    def set_pair_kernel(self, variant):
        pass

    def set_fused_step(self, on):
        pass

    def use_graph(self, on):
        pass

    def selftest(self):
        pass

    def set_grad_mode(self, mode):
        pass

    # -- the rollout in torch
    def _dynamics(self):
        s = self.slots[_lib.SLOT_DYNAMICS]
        if s.get("Z") is not None or "user" in s:      # sparse model (FITC factors over the inducing inputs, smgpr.py:24-52) / caller's factors
            pts, iK, beta = self._factors(_lib.SLOT_DYNAMICS)
            P, ls, var, iK, beta = T(pts), T(s["ls"]), T(s["var"]), T(iK), T(beta)
            return lambda m, sx: tq.predict_given_factorizations(P, ls, var, m, sx, iK, beta)
        X, ls, var = T(s["X"]), T(s["ls"]), T(s["var"])
        N, E = s["Y"].shape
        iK, beta = [], []
        for a in range(E):
            K = var[a] * torch.exp(-0.5 * torch.sum(((X[:, None, :] - X[None, :, :]) / ls[a]) ** 2, -1)) + s["nz"][a] * torch.eye(N, dtype=tq.DT)
            try:
                torch.linalg.cholesky(K)
                Ki = torch.linalg.inv(K)
            except torch.linalg.LinAlgError as exc:      # the device reports PILCO_E_NOT_PD
                raise _lib.NotPositiveDefiniteError(5, str(exc))
            iK.append(Ki)
            beta.append(Ki @ T(s["Y"][:, a]))
        iK, beta = torch.stack(iK), torch.stack(beta)
        return lambda m, sx: tq.predict_given_factorizations(X, ls, var, m, sx, iK, beta)

    @staticmethod
    def _reward(terms, E):
        def f(m, s):
            tot = torch.zeros((1, 1), dtype=tq.DT)
            for t in terms:
                if t["kind"] == _lib.REWARD_EXPONENTIAL:
                    tot = tot + t["coef"] * tq.exponential_reward(m, s, t["W"], t["t"])
                elif t["kind"] == _lib.REWARD_LINEAR:
                    tot = tot + t["coef"] * (m @ T(t["W"]).reshape(E, 1))
                else:
                    raise NotImplementedError(t["kind"])
            return tot
        return f

    def _value(self, policy, rewards, m0, S0, H, params, seeds=None, traj=None):
        """(m_H, s_H, additive reward [+ sum_t <seeds_t, state_t> when seeds are given]); traj (list) collects the states."""
        E = policy["state_dim"]
        e = T(np.broadcast_to(np.asarray(policy.get("max_action", 1.0), np.float64).reshape(-1), (policy["control_dim"],)).copy())
        if policy["kind"] == _lib.POLICY_LINEAR:
            W, b = params
            ctl = lambda m, s: tq.linear_controller(m, s, W, b, e, policy.get("squash", True))
        elif policy["kind"] == _lib.POLICY_RBF:
            Xp, Yp, lsp, nzp = params
            ctl = lambda m, s: tq.rbf_controller(m, s, Xp, Yp, lsp, nzp, e, policy.get("squash", True))
        elif policy["kind"] == _lib.POLICY_NONE:      # autonomous system: no control inputs
            ctl = lambda m, s: (torch.zeros((1, 0), dtype=tq.DT), torch.zeros((0, 0), dtype=tq.DT), torch.zeros((E, 0), dtype=tq.DT))
        else:
            raise NotImplementedError(policy["kind"])
        gp, reward = self._dynamics(), self._reward(rewards, E)
        m, s = T(np.reshape(m0, (1, E))), T(np.reshape(S0, (E, E)))
        total = torch.zeros((1, 1), dtype=tq.DT)
        for t in range(int(H) + 1):
            if traj is not None:
                traj.append(torch.cat([m.reshape(-1), s.reshape(-1)]))
            if seeds is not None:
                total = total + (T(seeds[t, :E]) * m.reshape(-1)).sum() + (T(seeds[t, E:]).reshape(E, E) * s).sum()
            if t == int(H):
                break
            total = total + reward(m, s)
            m, s = tq.propagate(gp, ctl, m, s)
        return m, s, total

    def _params(self, policy):
        if policy["kind"] == _lib.POLICY_NONE:
            return []
        if policy["kind"] == _lib.POLICY_LINEAR:
            return [T(policy["W"]), T(policy["b"])]
        p = self.slots[_lib.SLOT_POLICY]
        return [T(p["X"]), T(p["Y"]), T(p["ls"]), T(p["nz"])]

    def rollout(self, policy, rewards, m0, S0, H, want_traj=False):
        with torch.no_grad():
            traj = [] if want_traj else None
            M, S, R = self._value(policy, rewards, m0, S0, H, self._params(policy), traj=traj)
        out = (M.numpy(), S.numpy(), R.numpy())
        return out + (torch.stack(traj).numpy(),) if want_traj else out

    def _grad(self, policy, rewards, m0, S0, H, params, n_diff, seed_fn):
        """What pilco_rollout_grad*_seeded does: forward pass, seeds = seed_fn(trajectory), gradient of the additive reward
        plus the seeded objective; the value returned is the additive reward alone."""
        self.grad_calls += 1
        seeds = None
        if seed_fn is not None:
            with torch.no_grad():
                traj = []
                self._value(policy, rewards, m0, S0, H, [p.detach() for p in params], traj=traj)
            seeds = np.asarray(seed_fn(torch.stack(traj).numpy()), np.float64)
        with torch.no_grad():
            r_add = float(self._value(policy, rewards, m0, S0, H, [p.detach() for p in params])[2])
        obj = self._value(policy, rewards, m0, S0, H, params, seeds=seeds)[2]
        g = torch.autograd.grad(obj.sum(), params[:n_diff])
        return (r_add,) + tuple(x.numpy() for x in g)

    def rollout_grad(self, policy, rewards, m0, S0, H, seed_fn=None):
        ps = [T(policy["W"]).clone().requires_grad_(True), T(policy["b"]).clone().requires_grad_(True)]
        return self._grad(policy, rewards, m0, S0, H, ps, 2, seed_fn)

    def rollout_grad_rbf(self, policy, rewards, m0, S0, H, Xp, Yp, lsp, noisep, seed_fn=None):
        ps = [T(np.array(v, np.float64)).clone().requires_grad_(True) for v in (Xp, Yp, lsp)] + [T(np.ravel(noisep))]
        return self._grad(policy, rewards, m0, S0, H, ps, 3, seed_fn)

    # the batched calls of the product's Context (restarts of optimize_policy as lanes): lane by lane on the stand-in -- what
    # matters on the CPU is the HOST logic around them (training._optimize_policy_lanes)
    nranks = 1
    has_comm = False
    lane_seeds = True

    def rollout_grad_batch(self, policies, rewards, m0, S0, H, seed_fns=None):
        out = [self.rollout_grad(pol, rewards, np.asarray(m0)[i], np.asarray(S0)[i], H, seed_fn=seed_fns[i] if seed_fns else None)
               for i, pol in enumerate(policies)]
        U, E = np.shape(policies[0]["W"])
        return (np.array([o[0] for o in out]), np.stack([np.reshape(o[1], (U, E)) for o in out]), np.stack([np.reshape(o[2], (U,)) for o in out]))

    def rollout_grad_rbf_batch(self, policies, rewards, m0, S0, H, Xp, Yp, lsp, noisep, seed_fns=None):
        out = [self.rollout_grad_rbf(pol, rewards, np.asarray(m0)[i], np.asarray(S0)[i], H, Xp[i], Yp[i], lsp[i], noisep[i],
                                     seed_fn=seed_fns[i] if seed_fns else None)
               for i, pol in enumerate(policies)]
        return (np.array([o[0] for o in out]),) + tuple(np.stack([o[k] for o in out]) for k in (1, 2, 3))
