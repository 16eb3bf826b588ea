// The reverse chain of the policy gradient on the device (round 6).
//
// TensorFlow differentiates the reference's whole tf.while_loop on the accelerator (pilco/models/pilco.py:85-90,126-135).
// Here the O(N^2) part of every step's adjoint is in the step's Jacobian records (bwd.hip); what is left is a chain of H
// dependent small contractions -- the records with the cotangents of the step's outputs, then the adjoints of propagate
// (pilco.py:147-149), the joint Gaussian (:141-144), the LinearController + squash_sin (controllers.py:46-58,13-36) and the
// reward terms (rewards.py:19-81).  Rounds 1-5 ran that chain on the host (grad.hip: jac_vjp + O(D^3) links), which streamed
// 108 kB of records per step through one core: 0.43 ms of a 5.3 ms value-and-gradient call at C2u.  Now:
//   rev_local_step (rev_local.h; an extra workgroup per step in the records' last launch, bwd.hip: k_mm_jac_fin): everything of
//                 a step's adjoint that depends on the TRAJECTORY only -- d reward / d (m_t, s_t), the controller's W s_x
//                 products, squash_sin's forward quantities (all exp / sin / cos of the chain) -- for all H steps at once;
//   k_rev_step    one workgroup per step: the step's whole reverse map as a 76 x 65 matrix (below);
//   k_rev_chain   ONE workgroup walks t = H-1 .. 0: a matrix-vector product per step, gradient accumulators in registers.
// Fixed summation orders throughout: the gradient is bitwise repeatable, and identical on every rank of a sharded model
// (every rank runs the same chain over the same all-gathered records).
#include "rev_local.h"

namespace pilco {

// ------------------------------------------------------------------ every step's reverse map as a matrix, all steps at once
// The reverse of a step is LINEAR in the cotangents: with x = (mbar (E) | sbar packed, P entries in the pairs' dealing order
// (0,0) .. (E-1,E-1), (1,0), (2,0), (2,1), ..) behind the step,
//     x_t = A_t x_{t+1} + r_t,      dtheta += B_t x_{t+1},      theta = (W (U,E) | b (U)).
// The first device chain (round 6, first half) walked the steps with the 77 x 175 records themselves: 108 kB per step through
// ONE CU's memory pipeline (~20 bytes per cycle) plus six barrier phases of small links -- 6 us per step, 240 us per rollout,
// no faster than the host it replaced.  What depends on the cotangents only linearly can be prepared for all steps in parallel:
// k_rev_step (one workgroup per step) folds the records with the step's propagate coefficients (pilco.py:147-149) into the
// 77 x 65 map x -> (mbar_joint | sbar_joint), pushes the 65 basis cotangents through joint Gaussian, squash and controller
// (pilco.py:141-144, controllers.py:13-58) and leaves [A_t; B_t] (76 x 65, stored by columns) and r_t; what remains
// sequential is one 40 kB matrix-vector product per step with two barriers (k_rev_chain).
struct RevDims {
    int E, U, D, P, NX, NP, NR, NOUT, NT2, nI, recp, reco;
};
__host__ __device__ inline RevDims rev_dims(int E, int U, int D) {
    RevDims d;
    d.E = E; d.U = U; d.D = D; d.P = E * (E + 1) / 2;
    d.NX = E + d.P; d.NP = U * E + U; d.NR = d.NX + d.NP;
    d.NT2 = D * (D + 1) / 2; d.NOUT = D + d.NT2; d.nI = D * D;
    d.recp = 1 + d.NOUT; d.reco = d.NOUT + d.nI + D * d.NT2;
    return d;
}
constexpr int REVS_SPLIT = 4;   // workgroups per step of k_rev_step (column ranges of [A; B]); a flag each
size_t rev_mat_doubles(int E, int U, int D) {   // per step: [A; B] by columns | r | flags
    const RevDims d = rev_dims(E, U, D);
    return (size_t)d.NX * d.NR + d.NX + REVS_SPLIT;
}
__device__ __forceinline__ int rev_tri(int r, int c) { return r <= c ? c * (c + 1) / 2 + r : r * (r + 1) / 2 + c; }   // packed symmetric index
// pair index of (r, c) in the dealing order (diagonal first)
__device__ __forceinline__ int rev_pair(int E, int r, int c) {
    if (r == c) return r;
    const int hi = r > c ? r : c, lo = r > c ? c : r;
    return E + hi * (hi - 1) / 2 + lo;
}

constexpr int REVS_NT = 1024;   // threads of a k_rev_step workgroup
constexpr int REV_MAXU = 4;
// LDS of k_rev_step (doubles): M1 [NX][NOUT] | s1 (E,D) | M (E) | V (D,E) | m_x | s_x | loc | W | gcol [NX][U + U*U] | pab (ints)
__host__ __device__ inline size_t rev_step_lds_doubles(int E, int U, int D) {
    const RevDims d = rev_dims(E, U, D);
    return (size_t)d.NX * d.NOUT + 2 * (size_t)E * D + E + (size_t)E + (size_t)E * E + rev_loc_size(E, U) + (size_t)U * E +
           (size_t)d.NX * (U + U * U) + (size_t)(d.P + 1) / 2 + 2;
}

__global__ __launch_bounds__(REVS_NT) void k_rev_step(RevArgs a) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int tid = threadIdx.x, t = blockIdx.x, part = blockIdx.y;
    const RevDims dm = rev_dims(a.E, a.U, a.D);
    const int E = dm.E, U = dm.U, D = dm.D, P = dm.P, NX = dm.NX, NR = dm.NR, NOUT = dm.NOUT, NT2 = dm.NT2, nI = dm.nI;
    const int UU = U * U, UE = U * E, ED = E * D, SE = E + E * E, NLOC = rev_loc_size(E, U), GC = U + UU;
    // this workgroup's columns of [A; B] = rows of M1 (REVS_SPLIT workgroups per step: one workgroup per step was 42 us -- 10 of
    // set-up, 11 fold, 20 matrix entries -- on 40 of 256 CUs; every part stages the step's small data, folds into and reads its own rows only)
    const int c0 = (int)((long)part * NX / REVS_SPLIT), c1 = (int)((long)(part + 1) * NX / REVS_SPLIT);
    double* M1 = sm;                    // [NX][NOUT]: row a < E: d (mbar_j | sbar_j) / d mbar_a, row E + p: d / d sbar_p
    double* s1 = M1 + (size_t)NX * NOUT;
    double* Mg = s1 + ED;
    double* Vg = Mg + E;
    double* mx = Vg + ED;
    double* sx = mx + E;
    double* loc = sx + E * E;           // rg (E + E*E) | T1 | T2 | sqM | sqCd | sqS | q | Ep | Em | cdm | csm | sdm | ssm | ee
    double* Wl = loc + NLOC;
    double* gcol = Wl + UE;             // [NX][U + UU]: mu0bar | su0bar of every basis column
    int* pab = (int*)(gcol + (size_t)NX * GC);
    const double* rg = loc;
    const double* T1 = loc + SE;
    const double* T2 = T1 + UE;
    const double* sqM = T2 + UE;
    const double* sqCd = sqM + U;
    const double* sqS = sqCd + U;
    const double* sqq = sqS + UU;
    const double* sqEp = sqq + UU;
    const double* sqEm = sqEp + UU;
    const double* cdm = sqEm + UU;
    const double* csm = cdm + UU;
    const double* sdm = csm + UU;
    const double* ssm = sdm + UU;
    const double* see = ssm + UU;
    const double* jr = a.jrec + (long)t * a.gstep;
    const double* tp = a.tape + (long)t * a.TS;
    // ---- 0: the step's small data; the pair table; M1 = (Jo | scale_p Jp)
    for (int i = tid; i < ED + E; i += REVS_NT) s1[i] = tp[D + nI + i];                       // s1 | M
    for (int i = tid; i < ED; i += REVS_NT) Vg[i] = tp[D + nI + ED + E + E * E + i];
    for (int i = tid; i < SE; i += REVS_NT) mx[i] = a.traj[(long)t * SE + i];                 // m_x | s_x
    for (int i = tid; i < NLOC; i += REVS_NT) loc[i] = a.loc[(long)t * NLOC + i];
    for (int i = tid; i < UE; i += REVS_NT) Wl[i] = a.Wp[i];
    for (int pp = tid; pp < P; pp += REVS_NT) {
        int pa = pp, pb = pp;
        if (pp >= E) {
            const int q = pp - E;
            int r = 1;
            while ((r * (r + 1)) / 2 <= q) ++r;
            pa = r;
            pb = q - r * (r - 1) / 2;
        }
        pab[pp] = pa | (pb << 16);
    }
    for (int i = c0 * NOUT + tid; i < c1 * NOUT; i += REVS_NT) {
        const int row = i / NOUT, e = i - row * NOUT;
        double v;
        if (row < E) v = jr[a.out_off + (long)row * dm.reco + e];
        else {
            const int pp = row - E;
            const double scale = (pp < E) ? 1.0 : 2.0;   // an off-diagonal entry of the symmetric cotangent stands for two
            v = scale * jr[(long)(pp % a.W) * a.gblk + (long)(pp / a.W) * dm.recp + 1 + e];
        }
        M1[i] = v;
    }
    __syncthreads();
    // ---- 1: fold.  Item (e, ao): term(ao, b) = -2 M_b Jo[ao][e] + 2 sum_k s1[b][k] Jv[ao, k][e] for every partner b is the
    // part of d / d sbar_{ao b} that comes through mu_ao and Vbar[:, ao]; it belongs to row E + pair(ao, b).  Two passes
    // (b <= ao, then b > ao) so that no two items of a pass touch the same entry: the sums are added in a fixed order.
    const int nitem = NOUT * E;
    for (int i0 = 0; i0 < nitem; i0 += REVS_NT) {
        const int i = i0 + tid;
        const bool on = i < nitem;
        const int ao = on ? i / NOUT : 0, e = on ? i - ao * NOUT : 0;
        double jv[16], jo = 0.0;   // (D <= 14 on this path)
#pragma unroll
        for (int k = 0; k < 16; ++k) jv[k] = 0.0;
        if (on) {
            const double* ro = jr + a.out_off + (long)ao * dm.reco;
            jo = ro[e];
#pragma unroll
            for (int k = 0; k < 16; ++k)
                if (k < D) jv[k] = ro[e < D ? NOUT + k * D + e : NOUT + nI + k * NT2 + (e - D)];
        }
        for (int pass = 0; pass < 2; ++pass) {
            if (on) {
                const int b0 = pass ? ao + 1 : 0, b1 = pass ? E : ao + 1;
                for (int b = b0; b < b1; ++b) {
                    const int rr = E + rev_pair(E, ao, b);
                    if (rr < c0 || rr >= c1) continue;   // (another part's row)
                    double acc = -Mg[b] * jo;
#pragma unroll
                    for (int k = 0; k < 16; ++k)
                        if (k < D) acc = fma(s1[b * D + k], jv[k], acc);
                    M1[(size_t)rr * NOUT + e] += 2.0 * acc;
                }
            }
            __syncthreads();
        }
    }
    // ---- 2: squash_sin's vector-Jacobian product for every basis column (derivatives as in gSin.m:50-74)
    auto sjb = [&](const double* am, int r, int c) { return am[D + rev_tri(r, c)]; };
    auto s1bar = [&](int col, int r, int d2) -> double {   // 2 (basis sbar) V^T
        if (col < E) return 0.0;
        const int pa = pab[col - E] & 0xffff, pb = pab[col - E] >> 16;
        double v = 0.0;
        if (r == pa) v += 2.0 * Vg[d2 * E + pb];
        if (r == pb && pa != pb) v += 2.0 * Vg[d2 * E + pa];
        return v;
    };
    bool bad = false;
    if (tid >= c0 && tid < c1) {
        const int col = tid;
        const double* am = M1 + (size_t)col * NOUT;
        for (int e = 0; e < NOUT; ++e)
            if (!(fabs(am[e]) <= 1.7e308)) bad = true;   // singular s + Lambda^2 or I + Lambda s somewhere in the step's records
        double* g = gcol + (size_t)col * GC;
        for (int u = 0; u < U; ++u) {
            double Cdbar = 0.0;   // sum_i V0[i][u] (s_x^T Bb)[i][u] = sum_l T1[u][l] Bb[l][u]
            for (int l = 0; l < E; ++l) Cdbar = fma(T1[u * E + l], 2.0 * sjb(am, l, E + u) + s1bar(col, l, E + u), Cdbar);
            const double Mbar_u = am[E + u], Mu = sqM[u], Cdu = sqCd[u];
            double acc = Mbar_u * Cdu - Cdbar * Mu;
            double dd = -0.5 * Mbar_u * Mu - 0.5 * Cdbar * Cdu, suu = 0.0;
            for (int v2 = 0; v2 < U; ++v2) {
                const int uv = u * U + v2, vu = v2 * U + u;
                const double s_uv = sjb(am, E + u, E + v2), s_vu = s_uv;
                const double D1 = see[uv] / 2.0 * (-(sqEp[uv] - sqq[uv]) * sdm[uv] + (sqEm[uv] - sqq[uv]) * ssm[uv]);
                const double D2 = see[vu] / 2.0 * ((sqEp[vu] - sqq[vu]) * sdm[vu] + (sqEm[vu] - sqq[vu]) * ssm[vu]);
                acc += s_uv * D1 + s_vu * D2;
                dd -= 0.5 * (s_uv * sqS[uv] + s_vu * sqS[vu]);
                const double sb = s_uv * (see[uv] / 2.0 * (sqEp[uv] * cdm[uv] + sqEm[uv] * csm[uv]));
                if (v2 == u) suu = sb;
                else g[U + uv] = sb;
            }
            g[u] = acc;
            g[U + u * U + u] = suu + dd;
        }
    }
    for (int i = tid; i < SE; i += REVS_NT)
        if (!(fabs(rg[i]) <= 1.7e308)) bad = true;       // singular I + S W in a reward term
    const int anybad = __syncthreads_or(bad ? 1 : 0);
    // ---- 3: the entries of [A; B], column by column (stored by columns: the chain's lanes run along the rows)
    double* AT = a.amat + (long)t * ((long)NX * NR + NX + REVS_SPLIT);
    for (int i = c0 * NR + tid; i < c1 * NR; i += REVS_NT) {
        const int col = i / NR, row = i - col * NR;
        const double* am = M1 + (size_t)col * NOUT;
        const double* g = gcol + (size_t)col * GC;   // mu0bar [U] | su0bar [U][U]
        auto Bb = [&](int r, int u) { return 2.0 * sjb(am, r, E + u) + s1bar(col, r, E + u); };
        double v;
        if (row < E) {                                   // mbar behind the step before: joint + W^T mu0bar
            double acc = (col == row ? 1.0 : 0.0) + am[row];
            for (int u = 0; u < U; ++u) acc = fma(Wl[u * E + row], g[u], acc);
            v = acc;
        } else if (row < NX) {                           // sbar: joint + Bb c^T + W^T su0bar W, symmetrised
            const int pr = pab[row - E] & 0xffff, pc = pab[row - E] >> 16;
            double f[2];
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const int r = s2 ? pc : pr, c = s2 ? pr : pc;
                double acc = ((col >= E && (rev_pair(E, r, c) == col - E)) ? 1.0 : 0.0) + sjb(am, r, c) + s1bar(col, r, c);
                for (int u = 0; u < U; ++u) {
                    acc = fma(Bb(r, u), Wl[u * E + c] * sqCd[u], acc);
                    double wsu = 0.0;
                    for (int v2 = 0; v2 < U; ++v2) wsu = fma(g[U + u * U + v2], Wl[v2 * E + c], wsu);
                    acc = fma(Wl[u * E + r], wsu, acc);
                }
                f[s2] = acc;
            }
            v = 0.5 * (f[0] + f[1]);
        } else if (row < NX + UE) {                      // dW[u][j] (controllers.py:46-58): V0bar^T + mu0bar m_x^T + su0bar terms
            const int q = row - NX, u = q / E, j = q - u * E;
            double cbj = 0.0;
            for (int l = 0; l < E; ++l) cbj = fma(sx[l * E + j], Bb(l, u), cbj);
            double acc = fma(cbj, sqCd[u], g[u] * mx[j]);
            for (int v2 = 0; v2 < U; ++v2) acc += g[U + u * U + v2] * T1[v2 * E + j] + g[U + v2 * U + u] * T2[v2 * E + j];
            v = acc;
        } else {
            v = g[row - NX - UE];                        // db[u]
        }
        AT[i] = v;
    }
    // ---- r_t: the reward's own cotangents (rewards.py:19-81), packed like x
    double* rv = AT + (long)NX * NR;
    for (int row = tid; row < (part == 0 ? NX : 0); row += REVS_NT) {
        double v;
        if (row < E) v = rg[row];
        else {
            const int pr = pab[row - E] & 0xffff, pc = pab[row - E] >> 16;
            v = 0.5 * (rg[E + pr * E + pc] + rg[E + pc * E + pr]);
        }
        rv[row] = v;
    }
    if (tid == 0) rv[NX + part] = anybad ? 1.0 : 0.0;
}

// ------------------------------------------------------------------ the chain: x_t = A_t x_{t+1} + r_t (+ seeds), dtheta += B_t x_{t+1}
constexpr int REV_NT = 512;   // threads of the chain's workgroup
constexpr int REV_RT = 16;    // matrix entries a thread keeps in flight for the NEXT step (registers)
// LDS-only barrier: the workgroup's global loads in flight (the next step's matrix) are NOT waited for
__device__ __forceinline__ void rev_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__global__ __launch_bounds__(REV_NT) void k_rev_chain(RevArgs a) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int tid = threadIdx.x;
    const RevDims dm = rev_dims(a.E, a.U, a.D);
    const int E = dm.E, H = a.H, NX = dm.NX, NR = dm.NR, SE = E + E * E;
    const int NCH = REV_NT / NR, nper = (NX + NCH - 1) / NCH;   // thread (chunk c_, row): columns c_ * nper ..
    const long MS = (long)NX * NR + NX + REVS_SPLIT;
    double* x = sm;                 // [NX] (+ one zero behind it: the coefficient of a slot past a thread's share)
    double* part = x + NX + 1;      // [NCH][NR]
    const int c_ = tid / NR, row = tid - c_ * NR;
    const bool mv = c_ < NCH;
    const int k0 = c_ * nper, nok = mv ? max(0, min(nper, NX - k0)) : 0;
    // requests: buffer loads, per-lane byte offset (column k0 + k, this row), the step's base in the scalar offset
    unsigned off[REV_RT];
#pragma unroll
    for (int k = 0; k < REV_RT; ++k) off[k] = (k < nok) ? 8u * (unsigned)((k0 + k) * NR + row) : 0u;
    const __amdgpu_buffer_rsrc_t mres = buf_rsrc_uniform(a.amat);
    const unsigned ms_b = 8u * (unsigned)MS;
    // additive part of row `row` (threads tid < NX): r_t, and the caller's seeds of state t (packed, symmetrised) if any
    int pr = 0, pc = 0;
    if (tid >= E && tid < NX) {
        const int q = tid - E;
        if (q < E) pr = pc = q;
        else {
            int r = 1;
            while ((r * (r + 1)) / 2 <= q - E) ++r;
            pr = r;
            pc = q - E - r * (r - 1) / 2;
        }
    }
    auto add_of = [&](int t) -> double {
        double v = a.amat[(long)t * MS + (long)NX * NR + tid];
        if (a.seeds) {
            const double* sd = a.seeds + (long)t * SE;
            v += (tid < E) ? sd[tid] : 0.5 * (sd[E + pr * E + pc] + sd[E + pc * E + pr]);
        }
        return v;
    };
    if (tid <= NX) {   // x behind the last step: the caller's seeds of state H, or zero
        double v = 0.0;
        if (a.seeds && tid < NX) {
            const double* sd = a.seeds + (long)H * SE;
            v = (tid < E) ? sd[tid] : 0.5 * (sd[E + pr * E + pc] + sd[E + pc * E + pr]);
        }
        x[tid] = v;
    }
    double rv[REV_RT], addn = 0.0, theta = 0.0, flag = 0.0;
#pragma unroll
    for (int k = 0; k < REV_RT; ++k) rv[k] = 0.0;
    if (H > 0) {
        if (tid < NX) addn = add_of(H - 1);
#pragma unroll
        for (int k = 0; k < REV_RT; ++k) rv[k] = buf_ld(mres, off[k], (unsigned)(H - 1) * ms_b);
    }
    for (int t = tid; t < H * REVS_SPLIT; t += REV_NT) flag += a.amat[(long)(t / REVS_SPLIT) * MS + (long)NX * NR + NX + t % REVS_SPLIT];
    __syncthreads();
    for (int t = H - 1; t >= 0; --t) {
        if (mv) {
            double acc = 0.0;
#pragma unroll
            for (int k = 0; k < REV_RT; ++k) acc = fma(x[k < nok ? k0 + k : NX], rv[k], acc);
            const double* base = a.amat + (long)t * MS;
            for (int k = REV_RT; k < nok; ++k) acc = fma(x[k0 + k], base[(long)(k0 + k) * NR + row], acc);   // (wider models than the prefetch holds)
            part[c_ * NR + row] = acc;
        }
        const double addc = addn;
        __builtin_amdgcn_sched_barrier(0);
        if (t > 0) {   // the next step's matrix and additive part: in flight behind the barrier and the sums
            if (tid < NX) addn = add_of(t - 1);
#pragma unroll
            for (int k = 0; k < REV_RT; ++k) rv[k] = buf_ld(mres, off[k], (unsigned)(t - 1) * ms_b);
        }
        rev_barrier();
        if (tid < NR) {
            double s0 = 0.0;
            for (int c = 0; c < NCH; ++c) s0 += part[c * NR + tid];   // fixed order
            if (tid < NX) x[tid] = s0 + addc;
            else theta += s0;
        }
        rev_barrier();
    }
    // flags: any step with a singular matrix in its records / rewards
    __shared__ double fl[REV_NT];
    fl[tid] = flag;
    __syncthreads();
    if (tid >= NX && tid < NR) a.out[tid - NX] = theta;             // dW [U][E] | db [U]
    if (tid == 0) {
        double f = 0.0;
        for (int i = 0; i < REV_NT; ++i) f += fl[i];
        a.out[dm.NP] = f > 0.0 ? 1.0 : 0.0;                           // status
    }
    if (tid < NX) a.out[dm.NP + 1 + tid] = x[tid];                   // d objective / d (m_0, S_0), packed (not part of the C ABI yet)
    if (tid == REV_NT - 1) a.out[dm.NP + 1 + NX] = a.reward_dev ? *a.reward_dev : 0.0;
}

void launch_rev_chain(hipStream_t st, const RevArgs& a) {
    if (a.H > 0) {
        const size_t lds_s = sizeof(double) * rev_step_lds_doubles(a.E, a.U, a.D);
        static size_t lds_set = 0;
        if (lds_s > 65536 && lds_s > lds_set) {
            (void)hipFuncSetAttribute((const void*)k_rev_step, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_s);
            lds_set = lds_s;
        }
        hipLaunchKernelGGL(k_rev_step, dim3(a.H, REVS_SPLIT), dim3(REVS_NT), lds_s, st, a);
    }
    const RevDims d = rev_dims(a.E, a.U, a.D);
    const size_t lds = sizeof(double) * ((size_t)d.NX + 1 + (size_t)(REV_NT / d.NR) * d.NR);
    hipLaunchKernelGGL(k_rev_chain, dim3(1), dim3(REV_NT), lds, st, a);
}

bool rev_chain_supported(int E, int U, int D) {
    const RevDims d = rev_dims(E, U, D);
    return U > 0 && U <= REV_MAXU && D == E + U && D <= 14 && d.NR <= REV_NT && d.P < 65536 &&
           sizeof(double) * rev_step_lds_doubles(E, U, D) <= 160 * 1024;
}

size_t rev_loc_doubles(int E, int U) { return (size_t)rev_loc_size(E, U); }
RevLocalArgs rev_local_args(int n, const RewardDev* rw, int E, int U, const double* traj, const double* Wp, const double* bp, const double* maxact,
                            double* loc) {
    RevLocalArgs rl{};
    rl.n = n; rl.E = E; rl.U = U; rl.traj = traj; rl.Wp = Wp; rl.bp = bp; rl.maxact = maxact; rl.loc = loc;
    for (int k = 0; k < MAX_REWARD_TERMS; ++k) rl.rs.rw[k] = (k < n) ? rw[k] : RewardDev{};
    return rl;
}


}  // namespace pilco
