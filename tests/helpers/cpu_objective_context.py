"""A CPU stand-in for the handful of _lib.Context calls the model-fitting glue (pilco_amd/training.py) makes, so that the
HOST side of MGPR.optimize / SMGPR.optimize -- softplus transforms with the 1e-6 noise floor, Gamma priors, packing, one
L-BFGS-B run per output in lockstep, restart draws and bookkeeping -- can be held to the executed reference's end points in
the CPU suite.  TEST INFRASTRUCTURE: the objective values come from oracle/gp_train.py (exact GP) and from a torch restatement
of gpflow's GPRFITC bound below; the product computes both on the device (pilco_gp_nlml, pilco_gp_fitc_nlml)."""
import numpy as np
import torch

from oracle import gp_train
from pilco_amd import _lib


class CpuObjectiveContext:
    def __init__(self):
        self._slot_owner = {}
        self.calls = 0

    def gp_set_data(self, slot, X, Y, owner=None):
        self._slot_owner[slot] = owner
        self.X, self.Y = np.array(X, np.float64), np.array(Y, np.float64)

    def gp_set_hyp(self, slot, lengthscales, variance, noise, owner=None):
        self._slot_owner[slot] = owner
        self.ls, self.var, self.nz = np.array(lengthscales, np.float64), np.ravel(variance).astype(np.float64), np.ravel(noise).astype(np.float64)

    def gp_set_inducing(self, slot, Z, owner=None):
        self._slot_owner[slot] = owner

    def gp_nlml(self, slot, D, E, want_grad=True):
        self.calls += 1
        nlml, grad = np.empty(E), np.empty((E, D + 2))
        try:
            for a in range(E):
                nlml[a], grad[a] = gp_train.nlml_and_grad(self.X, self.Y[:, a], self.ls[a], self.var[a], self.nz[a])
        except np.linalg.LinAlgError as exc:
            raise _lib.NotPositiveDefiniteError(3, str(exc))
        return nlml, grad

    def gp_fitc_nlml(self, slot, Z_all, D, E, want_grad=True):
        """gpflow.models.GPRFITC: -log N(y | 0, Qff + diag(Kff - Qff) + noise I) by the inducing-point identities."""
        self.calls += 1
        nthreads = torch.get_num_threads()
        torch.set_num_threads(1)          # bitwise repeatable sums: the fits are hundreds of iterations in a flat valley
        try:
            return self._fitc(Z_all, D, E)
        finally:
            torch.set_num_threads(nthreads)

    def _fitc(self, Z_all, D, E):
        X, N = torch.from_numpy(self.X), self.X.shape[0]
        nlml, gh, gz = np.empty(E), np.empty((E, D + 2)), np.empty((E,) + Z_all.shape[1:])
        for a in range(E):
            ls = torch.tensor(self.ls[a], requires_grad=True)
            var = torch.tensor(self.var[a], requires_grad=True)
            nz = torch.tensor(self.nz[a], requires_grad=True)
            Z = torch.tensor(np.asarray(Z_all[a], np.float64), requires_grad=True)
            y = torch.from_numpy(self.Y[:, a])
            M = Z.shape[0]

            def k(A, B):
                d = (A / ls)[:, None, :] - (B / ls)[None, :, :]
                return var * torch.exp(-0.5 * (d * d).sum(-1))
            try:
                Luu = torch.linalg.cholesky(k(Z, Z) + 1e-6 * torch.eye(M, dtype=torch.float64))
                V = torch.linalg.solve_triangular(Luu, k(Z, X), upper=False)
                nu = var - (V * V).sum(0) + nz
                L = torch.linalg.cholesky(torch.eye(M, dtype=torch.float64) + (V / nu) @ V.T)
            except torch.linalg.LinAlgError as exc:
                raise _lib.NotPositiveDefiniteError(3, str(exc))
            gamma = torch.linalg.solve_triangular(L, (V @ (y / nu))[:, None], upper=False)
            loss = (0.5 * (y * y / nu).sum() - 0.5 * (gamma * gamma).sum() + 0.5 * N * np.log(2 * np.pi)
                    + 0.5 * torch.log(nu).sum() + torch.log(torch.diagonal(L)).sum())
            g = torch.autograd.grad(loss, [ls, var, nz, Z])
            nlml[a] = float(loss.detach())
            gh[a, :D], gh[a, D], gh[a, D + 1], gz[a] = g[0].numpy(), float(g[1]), float(g[2]), g[3].numpy()
        return nlml, gh, gz
