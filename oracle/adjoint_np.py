"""NumPy prototype of the hand-derived adjoint (vector-Jacobian product) of one
moment-matching step, validated against torch autograd (oracle/torch_path.py).

TEST INFRASTRUCTURE ONLY.  It is the specification the device kernels
(pilco_gp_predict_vjp) implement; derivation in DESIGN.md section 9:

  d e_ij / d m = g_ij,   d e_ij / d s = g_ij g_ij^T / 2,   g_ij = P (z_i + w_j),   P = (I + Lambda_ab s)^{-1}
  d log det R_ab / d s = P Lambda_ab
so that per pair only the row sums r_i = sum_j w_ij L_ij, the column sums c_j and the
row moments m_i = sum_j w_ij L_ij w_j are needed from the O(N^2) sweep.
"""
import numpy as np


def mm_step_vjp(X, ls, var, iK, beta, m, s, Mbar, Sbar, Vbar):
    X = np.asarray(X, np.float64)
    ls = np.asarray(ls, np.float64)
    E, D = ls.shape
    m = np.asarray(m, np.float64).reshape(-1)
    zeta = X - m[None, :]
    eye = np.eye(D)
    mbar = np.zeros(D)
    sbar = np.zeros((D, D))
    # forward mean quantities (needed for the -M M^T term)
    M = np.empty(E)
    cache = []
    for a in range(E):
        T = np.linalg.inv(s + np.diag(ls[a] ** 2))
        c = var[a] * np.sqrt(np.prod(ls[a] ** 2) / np.linalg.det(s + np.diag(ls[a] ** 2)))
        l = np.exp(-0.5 * np.sum((zeta @ T) * zeta, 1)) * beta[a]
        g = l.sum()
        h = zeta.T @ l
        M[a] = c * g
        cache.append((T, c, l, g, h))
    Ssym = Sbar + Sbar.T
    for a in range(E):
        T, c, l, g, h = cache[a]
        mu = Mbar[0, a] - Ssym[a] @ M          # from S = ... - M M^T
        v = Vbar[:, a]
        u = T @ v
        q = mu + zeta @ u
        phi = c * (mu * g + v @ T @ h)
        H2q = (zeta * (l * q)[:, None]).T @ zeta
        mbar += c * (T @ (zeta.T @ (l * q)) - g * u)
        Th = T @ h
        sbar += -0.5 * phi * T + 0.5 * c * T @ H2q @ T - 0.5 * c * (np.outer(u, Th) + np.outer(Th, u))
    k = np.stack([np.log(var[a]) - 0.5 * np.sum((zeta / ls[a]) ** 2, 1) for a in range(E)])
    for a in range(E):
        za = zeta / ls[a] ** 2
        for b in range(a + 1):
            wb = zeta / ls[b] ** 2
            Lam = np.diag(1.0 / ls[a] ** 2 + 1.0 / ls[b] ** 2)
            R = s @ Lam + eye
            P = np.linalg.inv(eye + Lam @ s)
            Q = np.linalg.solve(R, s) / 2.0
            y = za[:, None, :] + wb[None, :, :]                      # (N,N,D)
            L = np.exp(k[a][:, None] + k[b][None, :] + np.einsum('ijd,de,ije->ij', y, Q, y))
            w = np.outer(beta[a], beta[b])
            if a == b:
                w = w - iK[a]
            WL = w * L
            Nab = WL.sum()
            kappa = (Sbar[a, b] + Sbar[b, a] if a != b else Sbar[a, a]) / np.sqrt(np.linalg.det(R))
            r = WL.sum(1)
            cs = WL.sum(0)
            mi = WL @ wb                                             # (N,D) row moments
            A1 = za.T @ r
            A2 = wb.T @ cs
            mbar += kappa * P @ (A1 + A2)
            inner = (za * r[:, None]).T @ za + (wb * cs[:, None]).T @ wb + za.T @ mi + mi.T @ za
            PL = P @ Lam
            sbar += kappa * (0.5 * P @ inner @ P.T - 0.25 * Nab * (PL + PL.T))
    return mbar[None, :], 0.5 * (sbar + sbar.T)
