// Trajectory-only quantities of one step of the policy gradient's reverse chain (included by rev.hip, whose k_rev_local runs it
// as a launch of its own, and by bwd.hip, where it rides as an extra workgroup per step in the records' last launch).
#pragma once
#include "mm_device.h"

namespace pilco {

// ------------------------------------------------------------------ d reward / d (m, s) of every pre-propagation state
// rewards.py:19-51 (exponential: muR = exp(-d^T X d / 2) / sqrt(det(I + S W)), X = W (I + S W)^-1; derivatives as in
// reward.m:47-50) and rewards.py:58-61 (linear), combined with their coefficients (rewards.py:73-81).
// loc [H][rev_loc_size]: (d mu / d m | d mu / d S) | T1 = W s_x^T (U,E) | T2 = W s_x (U,E) | squash_sin forward (controllers.py:
// 13-36): M (U) | Cd (U) | S | q | Ep | Em | cos(dm) | cos(sm) | sin(dm) | sin(sm) | e_u e_v (U,U each) -- the order the chain
// keeps them in LDS.  A singular I + S W leaves non-finite entries in the first block (the chain reports it).
__host__ __device__ inline int rev_loc_size(int E, int U) { return E + E * E + 2 * U * E + 2 * U + 9 * U * U; }
__device__ inline void rev_local_step(int n, const RevRewards& rs, int E, int U, const double* __restrict__ traj, const double* __restrict__ Wp,
                                      const double* __restrict__ bp, const double* __restrict__ maxact, double* __restrict__ loc, int z,
                                      double* sm) {
    const int t = threadIdx.x, SE = E + E * E, nc = 2 * E, NLOC = rev_loc_size(E, U);
    double* mx = sm;               // [E] | sx [E][E]
    double* sx = mx + E;
    double* dm = sx + E * E;       // [E] | dS [E][E]
    double* dS = dm + E;
    double* v = dS + E * E;        // [E]  X d
    double* dTi = v + E;           // [E]  d^T X
    double* d = dTi + E;           // [E]
    double* G0 = d + E;            // [E][2E]
    double* G1 = G0 + E * nc;      // [E][2E]
    for (int e = t; e < SE; e += 256) {
        mx[e] = traj[(long)z * SE + e];
        dm[e] = 0.0;
    }
    __syncthreads();
    for (int k = 0; k < n; ++k) {
        const RewardDev& rw = rs.rw[k];
        const double c = rw.coef;
        if (rw.kind != PILCO_REWARD_EXPONENTIAL) {   // linear: d mu / d m = W
            if (t < E) dm[t] += c * rw.W[t];
            __syncthreads();
            continue;
        }
        // [(I + S W)^T | W^T] -> [I | X^T]
        for (int e = t; e < E * nc; e += 256) {
            const int r = e / nc, cc = e - r * nc;
            double val;
            if (cc < E) {
                double sw = 0.0;   // (S W)[cc][r]
                for (int q = 0; q < E; ++q) sw = fma(sx[cc * E + q], rw.W[q * E + r], sw);
                val = sw + (r == cc ? 1.0 : 0.0);
            } else {
                val = rw.W[(cc - E) * E + r];
            }
            G0[e] = val;
        }
        if (t < E) d[t] = mx[t] - rw.t[t];
        double det;
        const double* res = gauss_jordan(G0, G1, E, nc, det);   // X[i][j] = res[j * nc + E + i]
        if (t < E) {
            double acc = 0.0;
            for (int j = 0; j < E; ++j) acc = fma(res[j * nc + E + t], d[j], acc);
            v[t] = acc;
        } else if (t >= 64 && t < 64 + E) {
            const int j = t - 64;
            double acc = 0.0;
            for (int i = 0; i < E; ++i) acc = fma(d[i], res[j * nc + E + i], acc);
            dTi[j] = acc;
        }
        __syncthreads();
        double quad = 0.0;
        for (int i = 0; i < E; ++i) quad = fma(d[i], v[i], quad);
        const double muR = exp(-0.5 * quad) / sqrt(det);
        if (t < E) dm[t] -= c * muR * dTi[t];
        for (int e = t; e < E * E; e += 256) {
            const int i = e / E, j = e - i * E;
            const double Tij = 0.5 * muR * (v[i] * dTi[j] - res[j * nc + E + i]);
            const double Tji = 0.5 * muR * (v[j] * dTi[i] - res[i * nc + E + j]);
            dS[e] += c * 0.5 * (Tij + Tji);
        }
        __syncthreads();
    }
    double* o = loc + (long)z * NLOC;
    for (int e = t; e < SE; e += 256) o[e] = dm[e];
    // ---- LinearController forward at (m_x, s_x) (controllers.py:46-58): mu0 = W m + b, su0 = W s W^T = T2 W^T
    const int UE = U * E, UU = U * U;
    double* T1 = G0;               // (the elimination's buffers are free again)
    double* T2 = T1 + UE;
    double* mu0 = T2 + UE;
    for (int w = t; w < 2 * UE + U; w += 256) {
        if (w < 2 * UE) {
            const bool first = w < UE;
            const int i = first ? w : w - UE, u = i / E, j = i - u * E;
            double acc = 0.0;
            for (int r = 0; r < E; ++r) acc = fma(Wp[u * E + r], first ? sx[j * E + r] : sx[r * E + j], acc);
            T1[w] = acc;           // T1 | T2 contiguous
            o[SE + w] = acc;
        } else {
            const int u = w - 2 * UE;
            double acc = bp[u];
            for (int r = 0; r < E; ++r) acc = fma(Wp[u * E + r], mx[r], acc);
            mu0[u] = acc;
        }
    }
    __syncthreads();
    double* q0 = o + SE + 2 * UE;   // M | Cd | S | q | Ep | Em | cdm | csm | sdm | ssm | ee
    for (int w = t; w < UU; w += 256) {
        const int u = w / U, v2 = w - u * U;
        double suv = 0.0, suu = 0.0, svv = 0.0;
        for (int j = 0; j < E; ++j) {
            suv = fma(T2[u * E + j], Wp[v2 * E + j], suv);
            suu = fma(T2[u * E + j], Wp[u * E + j], suu);
            svv = fma(T2[v2 * E + j], Wp[v2 * E + j], svv);
        }
        const double lq = -(suu + svv) / 2.0;
        const double q = exp(lq), Ep = exp(lq + suv), Em = exp(lq - suv);
        const double dmv = mu0[u] - mu0[v2], smv = mu0[u] + mu0[v2], ee = maxact[u] * maxact[v2];
        const double cd = cos(dmv), cs = cos(smv);
        double* qq = q0 + 2 * U;
        qq[w] = ee / 2.0 * ((Ep - q) * cd - (Em - q) * cs);
        qq[UU + w] = q;
        qq[2 * UU + w] = Ep;
        qq[3 * UU + w] = Em;
        qq[4 * UU + w] = cd;
        qq[5 * UU + w] = cs;
        qq[6 * UU + w] = sin(dmv);
        qq[7 * UU + w] = sin(smv);
        qq[8 * UU + w] = ee;
        if (u == v2) {
            const double ex = exp(-suu / 2.0);
            q0[u] = maxact[u] * ex * sin(mu0[u]);
            q0[U + u] = maxact[u] * ex * cos(mu0[u]);
        }
    }
}

__host__ __device__ inline size_t rev_local_lds_doubles(int E, int U) {
    const size_t gj = (size_t)4 * E * E, pol = (size_t)2 * U * E + U;
    return (size_t)2 * (E + E * E) + 3 * E + (gj > pol ? gj : pol) + 8;
}

}  // namespace pilco
