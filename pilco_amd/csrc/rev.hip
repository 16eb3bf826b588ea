// The reverse chain of the policy gradient on the device (round 6).
//
// TensorFlow differentiates the reference's whole tf.while_loop on the accelerator (pilco/models/pilco.py:85-90,126-135).
// Here the O(N^2) part of every step's adjoint is in the step's Jacobian records (bwd.hip); what is left is a chain of H
// dependent small contractions -- the records with the cotangents of the step's outputs, then the adjoints of propagate
// (pilco.py:147-149), the joint Gaussian (:141-144), the LinearController + squash_sin (controllers.py:46-58,13-36) and the
// reward terms (rewards.py:19-81).  Rounds 1-5 ran that chain on the host (grad.hip: jac_vjp + O(D^3) links), which streamed
// 108 kB of records per step through one core: 0.43 ms of a 5.3 ms value-and-gradient call at C2u.  Now:
//   k_rev_local   one workgroup per step: everything of a step's adjoint that depends on the TRAJECTORY only -- d reward /
//                 d (m_t, s_t), the controller's W s_x products, squash_sin's forward quantities (all exp / sin / cos of
//                 the chain) -- for all H steps at once;
//   k_rev_chain   ONE workgroup walks t = H-1 .. 0 with the records read from HBM (the next step's are requested into
//                 registers while the current step's links run), everything else in LDS; gradient accumulators in registers.
// Fixed summation orders throughout: the gradient is bitwise repeatable, and identical on every rank of a sharded model
// (every rank runs the same chain over the same all-gathered records).
#include "mm_device.h"

namespace pilco {

// ------------------------------------------------------------------ d reward / d (m, s) of every pre-propagation state
// rewards.py:19-51 (exponential: muR = exp(-d^T X d / 2) / sqrt(det(I + S W)), X = W (I + S W)^-1; derivatives as in
// reward.m:47-50) and rewards.py:58-61 (linear), combined with their coefficients (rewards.py:73-81).
// loc [H][rev_loc_size]: (d mu / d m | d mu / d S) | T1 = W s_x^T (U,E) | T2 = W s_x (U,E) | squash_sin forward (controllers.py:
// 13-36): M (U) | Cd (U) | S | q | Ep | Em | cos(dm) | cos(sm) | sin(dm) | sin(sm) | e_u e_v (U,U each) -- the order the chain
// keeps them in LDS.  A singular I + S W leaves non-finite entries in the first block (the chain reports it).
__host__ __device__ inline int rev_loc_size(int E, int U) { return E + E * E + 2 * U * E + 2 * U + 9 * U * U; }
__global__ __launch_bounds__(256) void k_rev_local(int n, RevRewards rs, int E, int U, const double* __restrict__ traj,
                                                   const double* __restrict__ Wp, const double* __restrict__ bp,
                                                   const double* __restrict__ maxact, double* __restrict__ loc) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int t = threadIdx.x, z = blockIdx.x, SE = E + E * E, nc = 2 * E, NLOC = rev_loc_size(E, U);
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

// ------------------------------------------------------------------ the chain
constexpr int REV_NT = 512;   // threads of the chain's workgroup
constexpr int REV_RT = 32;    // record entries a thread keeps in flight for the NEXT step (registers)

// LDS-only barrier: the workgroup's global loads in flight (the next step's records) are NOT waited for
__device__ __forceinline__ void rev_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// developer aid (PILCO_REV_STAMPS=1): thread 0's shader-clock stamp behind every phase of every step, [H][8] + wall clock [2]
#define REV_STAMP(slot_)                                                                          \
    do {                                                                                          \
        if (a.dbg && tid == 0) a.dbg[(long)(H - 1 - t) * 8 + (slot_)] = __builtin_readcyclecounter(); \
    } while (0)

// LDS layout of the chain (offsets in doubles)
struct RevLay {
    int one, zero;                      // the constants 1.0 and 0.0 (operands of the generic dot / sum forms below)
    int Wl;                             // [U][E]
    int mbar, sbar;                     // cotangents of the state behind the step: [E], [E][E] (exactly symmetric)
    int s1, Mg, Vg, mx, sx;             // step data in the order of the flat prefetch index: s1 (E,D) | M (E) | V (D,E) | m_x | s_x
    int rg, T1, T2, sqM, sqCd, sqS, sqq, sqEp, sqEm, cdm, csm, sdm, ssm, see;   // ... | the step's k_rev_local record (same order as in memory)
    int sd2;                            // the step data twice: step t reads copy t & 1 while the data of step t - 1 lands in the other (distance of the copies)
    int s1b, cvec, part, am, Bb, sub, su0b, mu0b, mxb, sxb, status, total;
};
__host__ __device__ inline RevLay rev_layout(int E, int U, int D, int P) {
    const int SE = E + E * E, NT2 = D * (D + 1) / 2, NOUT = D + NT2, NTERM = P + E + E * D, NCHK = REV_NT / NOUT;
    const int UU = U * U, UE = U * E, ED = E * D;
    RevLay L;
    int o = 0;
    auto take = [&o](int n) { const int q = o; o += n; return q; };
    L.one = take(1); L.zero = take(1);
    L.Wl = take(UE);
    L.mbar = take(E); L.sbar = take(E * E);
    L.s1 = take(ED); L.Mg = take(E); L.Vg = take(ED); L.mx = take(E); L.sx = take(E * E);
    L.rg = take(SE); L.T1 = take(UE); L.T2 = take(UE); L.sqM = take(U); L.sqCd = take(U);
    L.sqS = take(UU); L.sqq = take(UU); L.sqEp = take(UU); L.sqEm = take(UU);
    L.cdm = take(UU); L.csm = take(UU); L.sdm = take(UU); L.ssm = take(UU); L.see = take(UU);
    L.sd2 = o - L.s1;
    take(L.sd2);
    L.s1b = take(ED); L.cvec = take(NTERM + 1);
    L.part = take(NCHK * NOUT); L.am = take(NOUT);
    L.Bb = take(E * U); L.sub = take(UU); L.su0b = take(UU); L.mu0b = take(U);
    L.mxb = take(E); L.sxb = take(E * E);
    L.status = take(2);
    L.total = o;
    return L;
}
size_t rev_chain_lds_doubles(int E, int U, int D, int P) { return (size_t)rev_layout(E, U, D, P).total; }

// offset (doubles, from the step's base) of record entry (term kg, output e): see RevArgs
__device__ __forceinline__ int rev_rec_off(const RevArgs& a, int kg, int e, int recp, int reco, int NOUT, int nI, int NT2) {
    const int D = a.D;
    if (kg < a.P) return (kg % a.W) * (int)a.gblk + (kg / a.W) * recp + 1 + e;
    if (kg < a.P + a.E) return (int)a.out_off + (kg - a.P) * reco + e;
    const int q = kg - a.P - a.E, ao = q / D, k = q - ao * D;
    return (int)a.out_off + ao * reco + (e < D ? NOUT + k * D + e : NOUT + nI + k * NT2 + (e - D));
}

// sum_x A[x * as] * B[x * bs] over LDS, four terms requested together (the chain is a latency chain: a plain loop would pay one
// LDS round trip per term); fixed order
__device__ __forceinline__ double rev_dot(const double* sm, int a0, int as, int b0, int bs, int len) {
    double acc = 0.0;
    int x = 0;
    for (; x + 4 <= len; x += 4) {
        const double p0 = sm[a0 + x * as], p1 = sm[a0 + (x + 1) * as], p2 = sm[a0 + (x + 2) * as], p3 = sm[a0 + (x + 3) * as];
        const double q0 = sm[b0 + x * bs], q1 = sm[b0 + (x + 1) * bs], q2 = sm[b0 + (x + 2) * bs], q3 = sm[b0 + (x + 3) * bs];
        acc = fma(p0, q0, acc);
        acc = fma(p1, q1, acc);
        acc = fma(p2, q2, acc);
        acc = fma(p3, q3, acc);
    }
    for (; x < len; ++x) acc = fma(sm[a0 + x * as], sm[b0 + x * bs], acc);
    return acc;
}

// Every thread's part in every phase is fixed for the whole chain -- at most ONE item per phase (rev_chain_supported) -- and is
// worked out once, before the first step, into a handful of LDS offsets: no index division inside the chain (the first
// version recomputed them per step: 8.9 us per step, against 1.6 for the arithmetic).  Per step, with six barriers:
//   A  the step's small data (prefetched into registers during the previous step) -> LDS          [merged with H of the step before]
//   B  coefficients of the record product: Shat_p | mu_a = Mbar_a - 2 sum_b Sbar_ab M_b | Vbar = 2 s1^T Sbar; s1bar = 2 Sbar V^T  (propagate, pilco.py:147-149)
//   C  the record product (registers x coefficients), partial sums to LDS; the NEXT step's records and small data requested
//   D  partial sums added in their fixed order -> (mbar_joint | packed symmetric sbar_joint)
//   E  joint Gaussian (pilco.py:141-144): cotangents of (m_x, s_x) so far, of the cross term (Bb) and of the action's covariance
//   G  squash_sin's vector-Jacobian product (derivatives as in gSin.m:50-74), one thread per action
//   H  LinearController (controllers.py:46-58): parameter gradients into registers; cotangents of (m_x, s_x) completed and
//      symmetrised straight into (mbar, sbar) for the step before
// Helpers: the chain's workgroup alone pulls a step's 108 kB of records (C2u) out of HBM / the Infinity Cache at what ONE CU's
// miss queue sustains -- 18 GB/s measured, 6 us per step, all of it on the chain.  Workgroups 8, 16, .. of the same launch --
// observed to be dispatched to the chain workgroup's XCD (block b -> XCD b % 8; an affinity used for speed only: nothing
// depends on it) -- stream the records in the chain's order and drop them: they land in that XCD's L2 ahead of the chain,
// whose own requests then hit there.  The other workgroups of the launch exit at once.
constexpr int REV_NH = 7;   // helper workgroups
__device__ void rev_helper(const RevArgs& a, int j) {
    const int tid = threadIdx.x;
    const long n2 = a.gstep / 2;   // 16-byte granules per step and rank block (the tail double, if any, is left to the chain)
    double acc = 0.0;
    for (int t = a.H - 1; t >= 0; --t)
        for (int r = 0; r < a.W; ++r) {
            const double2* base = (const double2*)(a.jrec + (long)r * a.gblk + (long)t * a.gstep);
            if ((((uintptr_t)base) & 15) != 0) base = (const double2*)((const double*)base + 1);
            for (long i0 = (long)j * REV_NT + tid; i0 < n2 - 1; i0 += (long)REV_NH * REV_NT * 8) {
                double2 v[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const long i = i0 + (long)q * REV_NH * REV_NT;
                    v[q] = (i < n2 - 1) ? base[i] : double2{0.0, 0.0};
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) acc += v[q].x + v[q].y;
            }
        }
    if (acc == 1.2345e301 && a.dbg) a.dbg[0] = 0;   // (keeps the loads alive; never true in practice, and harmless if it were)
}

__global__ __launch_bounds__(REV_NT) void k_rev_chain(RevArgs a) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    if (blockIdx.x != 0) {
        if ((blockIdx.x & 7) == 0) rev_helper(a, (int)(blockIdx.x >> 3) - 1);
        return;
    }
    const int tid = threadIdx.x;
    // Sizes and the LDS layout: derived once here and AGAIN at the top of every step from laundered copies of the dimensions
    // (and the per-thread offsets below are laundered there too) -- left alone, the compiler hoists every phase's address
    // arithmetic out of the step loop and keeps all of it live across all phases: 256 VGPRs and 100+ spilled in the first builds.
#define REV_DIMS                                                                                                             \
    const int SE = E + E * E, nI = D * D, NT2 = D * (D + 1) / 2, NOUT = D + NT2, recp = 1 + NOUT, reco = NOUT + nI + D * NT2;  \
    const int NTERM = P + E + E * D, NCHK = REV_NT / NOUT, nper = (NTERM + NCHK - 1) / NCHK;                                 \
    const int UU = U * U, UE = U * E, ED = E * D, NLOC = rev_loc_size(E, U), NSD = 2 * ED + E + SE + NLOC;                   \
    const RevLay L = rev_layout(E, U, D, P);                                                                                 \
    (void)recp; (void)reco; (void)nper; (void)UU; (void)UE; (void)NLOC; (void)NSD; (void)NT2; (void)nI; (void)NCHK; (void)NTERM;
    const int H = a.H;
    const int E = a.E, U = a.U, D = a.D, P = a.P;
    REV_DIMS
    auto tri = [](int r, int c) { return r <= c ? c * (c + 1) / 2 + r : r * (r + 1) / 2 + c; };   // packed symmetric index

    // ---- this thread's items
    // B: out = (init ? sm[init] : 0) + scale * dot
    // (*_sd = 1: the operand lives in the step data -- its offset moves with the copy in use)
    int b_out = -1, b_a0 = 0, b_as = 0, b_b0 = 0, b_bs = 0, b_len = 0, b_init = L.zero, b_asd = 0, b_bsd = 0, e_sd2 = 0, e_sd3 = 0;
    double b_scale = 0.0;
    {
        int i = tid;
        if (i < P) {
            int pa = i, pb = i;
            if (i >= E) {   // dealing order: the diagonal first, then (1,0), (2,0), (2,1), ..
                const int q = i - E;
                int r = 1;
                while ((r * (r + 1)) / 2 <= q) ++r;
                pa = r;
                pb = q - r * (r - 1) / 2;
            }
            b_out = L.cvec + i; b_a0 = L.sbar + pa * E + pb; b_b0 = L.one; b_len = 1; b_scale = (pa == pb) ? 1.0 : 2.0;
        } else if ((i -= P) < E) {
            b_out = L.cvec + P + i; b_a0 = L.sbar + i * E; b_as = 1; b_b0 = L.Mg; b_bsd = 1; b_bs = 1; b_len = E; b_init = L.mbar + i; b_scale = -2.0;
        } else if ((i -= E) < ED) {   // cvec[P + E + ao * D + k] = Vbar[k][ao]
            const int ao = i / D, k = i - ao * D;
            b_out = L.cvec + P + E + i; b_a0 = L.s1 + k; b_asd = 1; b_as = D; b_b0 = L.sbar + ao; b_bs = E; b_len = E; b_scale = 2.0;
        } else if ((i -= ED) < ED) {  // s1bar[r][d]
            const int r = i / D, dd = i - r * D;
            b_out = L.s1b + i; b_a0 = L.sbar + r * E; b_as = 1; b_b0 = L.Vg + dd * E; b_bsd = 1; b_bs = 1; b_len = E; b_scale = 2.0;
        }
    }
    // E: out = ((sm[o0] + sm[o1]) + sm[o2]) + sm[o3]
    int e_out = -1, e_o0 = L.zero, e_o1 = L.zero, e_o2 = L.zero, e_o3 = L.zero;
    {
        int i = tid;
        if (i < E) {
            e_out = L.mxb + i; e_o0 = L.mbar + i; e_o1 = L.am + i; e_o2 = L.rg + i; e_sd2 = 1;
        } else if ((i -= E) < E * E) {
            const int r = i / E, c = i - r * E;
            e_out = L.sxb + i; e_o0 = L.sbar + i; e_o1 = L.am + D + tri(r, c); e_o2 = L.s1b + r * D + c; e_o3 = L.rg + E + i; e_sd3 = 1;
        } else if ((i -= E * E) < E * U) {
            const int r = i / U, u = i - r * U;
            e_out = L.Bb + i; e_o0 = e_o1 = L.am + D + tri(r, E + u); e_o2 = L.s1b + r * D + E + u;
        } else if ((i -= E * U) < UU) {
            const int u = i / U, v2 = i - u * U;
            e_out = L.sub + i; e_o0 = L.am + D + tri(E + u, E + v2);
        }
    }
    // H: wave 0: dW[u][j] / db[u]; wave 1: mbar; waves 2..: sbar
    int h_u = tid / E, h_j = tid - h_u * E;                    // tid < U E: dW[h_u][h_j]
    int h_i = tid - 128, h_r = h_i / E, h_c = h_i - h_r * E;   // 128 <= tid < 128 + E E: sbar[h_r][h_c]
    // C: output e_, terms kg0 .. kg0 + nok - 1 of the record product
    int c_ = tid / NOUT, e_ = tid - c_ * NOUT;
    const bool mv = c_ < NCHK;
    int kg0 = c_ * nper, nok = mv ? max(0, min(nper, NTERM - kg0)) : 0;
    // (a slot past a thread's share reads entry 0 of the step's records -- always there -- against the zero coefficient
    // cvec[NTERM]: the requests of a step are unconditional, no branch per slot)
    // Buffer loads: resource = the records' base, per-lane BYTE offset from the table below, the step's base in the scalar offset
    // (with 64-bit pointers the 32 addresses of a step alone would take 64 registers and two VALU operations each).
    unsigned off[REV_RT];
#pragma unroll
    for (int k = 0; k < REV_RT; ++k) off[k] = (k < nok) ? 8u * (unsigned)rev_rec_off(a, kg0 + k, e_, recp, reco, NOUT, nI, NT2) : 0u;
    const __amdgpu_buffer_rsrc_t jres = buf_rsrc_uniform(a.jrec);
    const unsigned gstep_b = 8u * (unsigned)a.gstep;
    // A: two slots of the step's small data: s1 | M | V (tape), m_x | s_x (trajectory), the step's k_rev_local record (+ seeds)
    // (every slot requests UNCONDITIONALLY -- an unused slot re-reads the trajectory's first entry, a slot without seeds adds
    // its own value times 0: a branch around a request makes the compiler wait for the data at the branch's end, one memory
    // round trip per slot in the middle of phase C: 7 300 of a step's 14 800 cycles in the first build)
    const double* sd_p[2];
    const double* sd_q[2];
    long sd_st[2], sd_qst[2];
    double sd_f[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int i = tid + q * REV_NT;
        sd_p[q] = a.traj; sd_st[q] = 0;
        if (i < ED + E) { sd_p[q] = a.tape + D + nI + i; sd_st[q] = a.TS; }
        else if (i < 2 * ED + E) { sd_p[q] = a.tape + D + nI + ED + E + E * E + (i - ED - E); sd_st[q] = a.TS; }
        else if (i < 2 * ED + E + SE) { sd_p[q] = a.traj + (i - 2 * ED - E); sd_st[q] = SE; }
        else if (i < NSD) { sd_p[q] = a.loc + (i - (2 * ED + E + SE)); sd_st[q] = NLOC; }
        sd_q[q] = sd_p[q]; sd_qst[q] = sd_st[q]; sd_f[q] = 0.0;
        const int j = i - (2 * ED + E + SE);
        if (a.seeds && j >= 0 && j < SE) { sd_q[q] = a.seeds + j; sd_qst[q] = SE; sd_f[q] = 1.0; }
    }
    // Requested TWO steps ahead and combined only where they are stored (phase H of the step before their own): a value used
    // next to its request is a memory round trip on the chain -- 2 us of cold HBM per step.
#define SD_REQ(t_) do { nx[0] = sd_p[0][(long)(t_) * sd_st[0]]; nx[1] = sd_q[0][(long)(t_) * sd_qst[0]];                     \
                        nx[2] = sd_p[1][(long)(t_) * sd_st[1]]; nx[3] = sd_q[1][(long)(t_) * sd_qst[1]]; } while (0)

    // ---- prologue
    for (int e = tid; e < UE; e += REV_NT) sm[L.Wl + e] = a.Wp[e];
    if (tid == 0) {
        sm[L.one] = 1.0;
        sm[L.zero] = 0.0;
        sm[L.cvec + NTERM] = 0.0;
        sm[L.status] = 0.0;
    }
    for (int e = tid; e < SE; e += REV_NT) {   // cotangents behind the last step: the caller's seeds of state H (symmetrised), or zero
        double v0 = 0.0;
        if (a.seeds) {
            const double* sd = a.seeds + (long)H * SE;
            if (e < E) v0 = sd[e];
            else {
                const int i = (e - E) / E, j = (e - E) - i * E;
                v0 = 0.5 * (sd[E + i * E + j] + sd[E + j * E + i]);
            }
        }
        sm[L.mbar + e] = v0;
    }
    double rv[REV_RT], cu[4] = {0.0, 0.0, 0.0, 0.0}, nx[4] = {0.0, 0.0, 0.0, 0.0};   // cu: small data of the step before the current one; nx: of the one before that
#pragma unroll
    for (int k = 0; k < REV_RT; ++k) rv[k] = 0.0;
    if (H > 0) {
        SD_REQ(H - 1);
#pragma unroll
        for (int q = 0; q < 4; ++q) cu[q] = nx[q];
#pragma unroll
        for (int k = 0; k < REV_RT; ++k) rv[k] = buf_ld(jres, off[k], (unsigned)(H - 1) * gstep_b);
    }
    if (a.dbg && tid == 0) a.dbg[(long)H * 8] = wall_clock64();
    double Wacc = 0.0;   // thread u * E + j: dW[u][j]; thread U*E + u: db[u]
    bool bad = false;
    // A of the first step
    {
        const int sd00 = ((H - 1) & 1) ? L.sd2 : 0;
        if (tid < NSD) sm[sd00 + L.s1 + tid] = fma(sd_f[0], cu[1], cu[0]);
        if (tid + REV_NT < NSD) sm[sd00 + L.s1 + tid + REV_NT] = fma(sd_f[1], cu[3], cu[2]);
        if (H > 1) {
            SD_REQ(H - 2);
#pragma unroll
            for (int q = 0; q < 4; ++q) cu[q] = nx[q];
        }
        for (int i = tid + 2 * REV_NT; i < NSD && H > 0; i += REV_NT) {   // (wider models than two slots per thread hold)
            double v0;
            if (i < ED + E) v0 = a.tape[(long)(H - 1) * a.TS + D + nI + i];
            else if (i < 2 * ED + E) v0 = a.tape[(long)(H - 1) * a.TS + D + nI + ED + E + E * E + (i - ED - E)];
            else if (i < 2 * ED + E + SE) v0 = a.traj[(long)(H - 1) * SE + (i - 2 * ED - E)];
            else {
                const int j = i - (2 * ED + E + SE);
                v0 = a.loc[(long)(H - 1) * NLOC + j];
                if (a.seeds && j < SE) v0 += a.seeds[(long)(H - 1) * SE + j];
            }
            sm[sd00 + L.s1 + i] = v0;
        }
    }
    __syncthreads();

    for (int t = H - 1; t >= 0; --t) {
        int E = a.E, U = a.U, D = a.D, P = a.P;
        asm volatile("" : "+s"(E), "+s"(U), "+s"(D), "+s"(P));
        REV_DIMS
        asm volatile("" : "+v"(b_out), "+v"(b_a0), "+v"(b_as), "+v"(b_b0), "+v"(b_bs), "+v"(b_len), "+v"(b_init));
        const int sdo = (t & 1) ? L.sd2 : 0, sdn = (t & 1) ? 0 : L.sd2;   // this step's copy of the step data / where the next one's lands
        asm volatile("" : "+v"(e_out), "+v"(e_o0), "+v"(e_o1), "+v"(e_o2), "+v"(e_o3), "+v"(b_asd), "+v"(b_bsd), "+v"(e_sd2), "+v"(e_sd3));
        asm volatile("" : "+v"(h_u), "+v"(h_j), "+v"(h_i), "+v"(h_r), "+v"(h_c), "+v"(c_), "+v"(e_), "+v"(kg0), "+v"(nok));
        // ---- B
        REV_STAMP(0);
        if (b_out >= 0) sm[b_out] = fma(b_scale, rev_dot(sm, b_a0 + (b_asd ? sdo : 0), b_as, b_b0 + (b_bsd ? sdo : 0), b_bs, b_len), sm[b_init]);
        rev_barrier();
        REV_STAMP(1);
        // ---- C
        if (mv) {
            double acc = 0.0;
#pragma unroll
            for (int k = 0; k < REV_RT; ++k) acc = fma(sm[L.cvec + (k < nok ? kg0 + k : NTERM)], rv[k], acc);
            const double* base = a.jrec + (long)t * a.gstep;
            for (int k = REV_RT; k < nok; ++k)   // (models with more terms per thread than the prefetch holds)
                acc = fma(sm[L.cvec + kg0 + k], base[rev_rec_off(a, kg0 + k, e_, recp, reco, NOUT, nI, NT2)], acc);
            sm[L.part + c_ * NOUT + e_] = acc;
        }
        __builtin_amdgcn_sched_barrier(0);   // (the requests below must not move up past the products: they reuse the registers)
        REV_STAMP(2);
        // The next step's records, requested in four groups behind phases C, D, E and G: one CU takes in ~20 bytes per cycle
        // (5 600 cycles for a step's 108 kB -- measured with all 32 requests per thread in one burst here, the workgroup
        // standing at the barrier meanwhile); spread out, the memory pipeline works beside the links instead of before them.
        // Small data: two steps ahead.
#define RV_REQ(g_)                                                                                                              \
    if (t > 0) {                                                                                                                \
        _Pragma("unroll") for (int k = (g_) * (REV_RT / 4); k < ((g_) + 1) * (REV_RT / 4); ++k)                                 \
            rv[k] = buf_ld(jres, off[k], (unsigned)(t - 1) * gstep_b);                                                          \
    }
        if (t > 1) SD_REQ(t - 2);
        RV_REQ(0)
        rev_barrier();
        REV_STAMP(3);
        // ---- D
        if (tid < NOUT) {
            const double acc = rev_dot(sm, L.part + tid, NOUT, L.one, 0, NCHK);
            sm[L.am + tid] = acc;
            if (!(fabs(acc) <= 1.7e308)) bad = true;   // singular s + Lambda^2 or I + Lambda s somewhere in the step's records
        }
        RV_REQ(1)
        rev_barrier();
        REV_STAMP(4);
        // ---- E
        if (e_out >= 0) {
            const double g = sm[e_o3 + (e_sd3 ? sdo : 0)], g2 = sm[e_o2 + (e_sd2 ? sdo : 0)];
            sm[e_out] = ((sm[e_o0] + sm[e_o1]) + g2) + g;
            if (tid >= NOUT && !(fabs(g + g2) <= 1.7e308)) bad = true;   // (threads past D's: the step's k_rev_local record -- a singular I + S W in a reward term)
        }
        RV_REQ(2)
        rev_barrier();
        REV_STAMP(5);
        // ---- G
        if (tid < U) {
            const int u = tid;
            const double Cdbar = rev_dot(sm, sdo + L.T1 + u * E, 1, L.Bb + u, U, E);   // sum_i V0[i][u] (s_x^T Bb)[i][u], V0 = W^T
            const double Mbar_u = sm[L.am + E + u], Mu = sm[sdo + L.sqM + u], Cdu = sm[sdo + L.sqCd + u];
            double acc = Mbar_u * Cdu - Cdbar * Mu;
            double dd = -0.5 * Mbar_u * Mu - 0.5 * Cdbar * Cdu, suu = 0.0;
            for (int v2 = 0; v2 < U; ++v2) {
                const int uv = u * U + v2, vu = v2 * U + u;
                // (all operands first: one LDS round trip, not one per use)
                const double s_uv = sm[L.sub + uv], s_vu = sm[L.sub + vu];
                const double ee_uv = sm[sdo + L.see + uv], Ep_uv = sm[sdo + L.sqEp + uv], q_uv = sm[sdo + L.sqq + uv], Em_uv = sm[sdo + L.sqEm + uv];
                const double sd_uv = sm[sdo + L.sdm + uv], ss_uv = sm[sdo + L.ssm + uv], cd_uv = sm[sdo + L.cdm + uv], cs_uv = sm[sdo + L.csm + uv];
                const double ee_vu = sm[sdo + L.see + vu], Ep_vu = sm[sdo + L.sqEp + vu], q_vu = sm[sdo + L.sqq + vu], Em_vu = sm[sdo + L.sqEm + vu];
                const double sd_vu = sm[sdo + L.sdm + vu], ss_vu = sm[sdo + L.ssm + vu], S_uv = sm[sdo + L.sqS + uv], S_vu = sm[sdo + L.sqS + vu];
                const double D1 = ee_uv / 2.0 * (-(Ep_uv - q_uv) * sd_uv + (Em_uv - q_uv) * ss_uv);
                const double D2 = ee_vu / 2.0 * ((Ep_vu - q_vu) * sd_vu + (Em_vu - q_vu) * ss_vu);
                acc += s_uv * D1 + s_vu * D2;
                dd -= 0.5 * (s_uv * S_uv + s_vu * S_vu);
                const double sb = s_uv * (ee_uv / 2.0 * (Ep_uv * cd_uv + Em_uv * cs_uv));
                if (v2 == u) suu = sb;
                else sm[L.su0b + uv] = sb;
            }
            sm[L.mu0b + u] = acc;
            sm[L.su0b + u * U + u] = suu + dd;
        }
        RV_REQ(3)
        rev_barrier();
        REV_STAMP(6);
        // ---- H (+ A of the step before: disjoint arrays)
        if (tid < UE) {
            const double cbj = rev_dot(sm, sdo + L.sx + h_j, E, L.Bb + h_u, U, E);          // (s_x^T Bb)[j][u]
            double acc = fma(cbj, sm[sdo + L.sqCd + h_u], sm[L.mu0b + h_u] * sm[sdo + L.mx + h_j]);   // V0bar^T + mu0bar m_x^T
            for (int v2 = 0; v2 < U; ++v2)
                acc += sm[L.su0b + h_u * U + v2] * sm[sdo + L.T1 + v2 * E + h_j] + sm[L.su0b + v2 * U + h_u] * sm[sdo + L.T2 + v2 * E + h_j];
            Wacc += acc;
        } else if (tid < UE + U) {
            Wacc += sm[L.mu0b + tid - UE];
        } else if (tid >= 64 && tid < 64 + E) {
            const int i = tid - 64;
            sm[L.mbar + i] = sm[L.mxb + i] + rev_dot(sm, L.Wl + i, E, L.mu0b, 1, U);
        } else if (tid >= 128 && h_i < E * E) {
            double f[2];
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const int r = s2 ? h_c : h_r, c = s2 ? h_r : h_c;
                double acc = sm[L.sxb + r * E + c];
                for (int u = 0; u < U; ++u) {
                    acc = fma(sm[L.Bb + r * U + u], sm[L.Wl + u * E + c] * sm[sdo + L.sqCd + u], acc);   // Bb c^T, c = V0 diag(Cd)
                    acc = fma(sm[L.Wl + u * E + r], rev_dot(sm, L.su0b + u * U, 1, L.Wl + c, E, U), acc);   // W^T su0bar W
                }
                f[s2] = acc;
            }
            sm[L.sbar + h_i] = 0.5 * (f[0] + f[1]);
        }
        if (t > 0) {
            if (tid < NSD) sm[sdn + L.s1 + tid] = fma(sd_f[0], cu[1], cu[0]);
            if (tid + REV_NT < NSD) sm[sdn + L.s1 + tid + REV_NT] = fma(sd_f[1], cu[3], cu[2]);
#pragma unroll
            for (int q = 0; q < 4; ++q) cu[q] = nx[q];
            for (int i = tid + 2 * REV_NT; i < NSD; i += REV_NT) {   // (wider models than two slots per thread hold)
                double v0;
                if (i < ED + E) v0 = a.tape[(long)(t - 1) * a.TS + D + nI + i];
                else if (i < 2 * ED + E) v0 = a.tape[(long)(t - 1) * a.TS + D + nI + ED + E + E * E + (i - ED - E)];
                else if (i < 2 * ED + E + SE) v0 = a.traj[(long)(t - 1) * SE + (i - 2 * ED - E)];
                else {
                    const int j = i - (2 * ED + E + SE);
                    v0 = a.loc[(long)(t - 1) * NLOC + j];
                    if (a.seeds && j < SE) v0 += a.seeds[(long)(t - 1) * SE + j];
                }
                sm[sdn + L.s1 + i] = v0;
            }
        }
        rev_barrier();
        REV_STAMP(7);
    }
#undef REV_DIMS
#undef SD_REQ
#undef RV_REQ
    if (a.dbg && tid == 0) a.dbg[(long)H * 8 + 1] = wall_clock64();
    if (bad) sm[L.status] = 1.0;
    __syncthreads();
    if (tid < UE + U) a.out[tid] = Wacc;                       // dW [U][E] | db [U]
    if (tid == REV_NT - 1) a.out[UE + U] = sm[L.status];       // 0 fine; 1: a singular matrix somewhere (see D, E)
    for (int e = tid; e < SE; e += REV_NT) a.out[UE + U + 1 + e] = sm[L.mbar + e];   // d objective / d (m_0, S_0) (not part of the C ABI yet)
}

void launch_rev_chain(hipStream_t st, const RevArgs& a0) {
    RevArgs a = a0;
    static const bool stamps = getenv("PILCO_REV_STAMPS") != nullptr;
    static unsigned long long* dbg = nullptr;
    static int dbg_H = 0;
    if (stamps && a.H > 0 && a.H <= 4096) {
        if (!dbg || dbg_H < a.H) {
            if (dbg) (void)hipHostFree(dbg);
            (void)hipHostMalloc((void**)&dbg, sizeof(unsigned long long) * ((size_t)a.H * 8 + 2), hipHostMallocDefault);
            dbg_H = a.H;
        }
        a.dbg = dbg;
    }
    const size_t lds = sizeof(double) * rev_chain_lds_doubles(a.E, a.U, a.D, a.P);
    static const int nh = getenv("PILCO_REV_HELPERS") ? std::max(0, std::min(REV_NH, atoi(getenv("PILCO_REV_HELPERS")))) : REV_NH;   // (A/B)
    hipLaunchKernelGGL(k_rev_chain, dim3(1 + 8 * nh), dim3(REV_NT), lds, st, a);
    if (a.dbg) {   // developer aid: waits, and prints the phases' mean cycles
        (void)hipStreamSynchronize(st);
        double ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int t = 0; t < a.H; ++t)
            for (int q = 0; q < 8; ++q) {
                const unsigned long long prev = q ? dbg[t * 8 + q - 1] : (t ? dbg[(t - 1) * 8 + 7] : dbg[0]);
                ph[q] += (double)(dbg[t * 8 + q] - prev) / a.H;
            }
        const double cyc = (double)(dbg[(a.H - 1) * 8 + 7] - dbg[0]), us = (double)(dbg[a.H * 8 + 1] - dbg[a.H * 8]) / 100.0;
        fprintf(stderr, "[pilco rev] %d steps %.1f us (%.0f MHz): per step  top %.0f | B %.0f | C products %.0f | C requests + barrier %.0f | D %.0f | E %.0f | G %.0f | H %.0f cycles\n", a.H, us,
                cyc / us, ph[0], ph[1], ph[2], ph[3], ph[4], ph[5], ph[6], ph[7]);
    }
}

bool rev_chain_supported(int E, int U, int D) {
    const int NOUT = D + D * (D + 1) / 2, P = E * (E + 1) / 2;
    return U > 0 && D == E + U && NOUT <= REV_NT && U * E + U <= 64 && E <= 64 && 128 + E * E <= REV_NT &&
           P + E + 2 * E * D <= REV_NT && E + E * E + E * U + U * U <= REV_NT &&
           sizeof(double) * rev_chain_lds_doubles(E, U, D, P) <= 64 * 1024;
}

size_t rev_loc_doubles(int E, int U) { return (size_t)rev_loc_size(E, U); }
void launch_rev_local(hipStream_t st, int n, const RewardDev* rw, int E, int U, int H, const double* traj, const double* Wp, const double* bp,
                      const double* maxact, double* loc) {
    if (H <= 0) return;
    RevRewards rs;
    for (int k = 0; k < MAX_REWARD_TERMS; ++k) rs.rw[k] = (k < n) ? rw[k] : RewardDev{};
    const size_t lds = sizeof(double) * ((size_t)2 * (E + E * E) + 3 * E + std::max((size_t)4 * E * E, (size_t)2 * U * E + U) + 8);
    hipLaunchKernelGGL(k_rev_local, dim3(H), dim3(256), lds, st, n, rs, E, U, traj, Wp, bp, maxact, loc);
}

}  // namespace pilco
