// The analytic moment-matching step (T5-T19 of SURVEY.md section 2.2) as three
// gfx950 kernels per horizon step:
//
//   k_mm_prep  : per output pair (a,b): R_ab, det R_ab, Q_ab = R^{-1} s / 2 by a
//                register-resident Gauss-Jordan in one wave (column per lane,
//                v_readlane broadcasts, no LDS, no barriers); then the O(N D^2)
//                per-row vectors of Appendix B (u_i, p_i = 2 Q z_i | w_j, v_j)
//                written k-major so the pair kernel reads MFMA fragments with
//                128-byte segments; diagonal pairs also do the mean / input-output
//                covariance sums (mgpr.py:102-118).  One extra workgroup of the
//                same launch evaluates the reward of the current state
//                (rewards.py:32-39), off the step's critical path.
//   k_mm_pair  : the O(N^2) part (mgpr.py:120-144): exponent tile = A^T B on
//                v_mfma_f64_16x16x4_f64 with K = D+2 (u and v folded into the
//                contraction), table-driven fp64 exp, beta-weighted reduction and,
//                for a == b, the streamed iK tile.  No atomics: one partial per
//                tile, summed in a fixed order => bitwise reproducible.
//   k_glue     : one workgroup: tile-partial reduction, S assembly
//                (mgpr.py:145-147), propagate (pilco.py:147-149), controller +
//                joint Gaussian for the next step (controllers.py:13-58,
//                pilco.py:139-144).
#include "moment.h"

#include <type_traits>

namespace pilco {

typedef double d4 __attribute__((ext_vector_type(4)));

// local pair index -> outputs (a >= b); see the dealing order in moment.h
__device__ __forceinline__ void local_pair_ab(const MMWork& wk, int E, int pl, int& a, int& b) {
    const int kk = pl * wk.nranks + wk.rank;
    if (kk < E) {
        a = b = kk;
        return;
    }
    const int q = kk - E;
    a = 1;
    while (a * (a + 1) / 2 <= q) ++a;
    b = q - a * (a - 1) / 2;
}
__device__ __forceinline__ int pair_order_index(int E, int a, int b) {  // a >= b
    return (a == b) ? a : E + a * (a - 1) / 2 + b;
}

// Write-through (sc1) store: the 10 MB of per-step operands leave the XCD's L2 as they are
// produced instead of in the end-of-kernel write-back, which is what the next kernel waits on.
__device__ __forceinline__ void store_wt(double* p, double v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

#define DBG_STAMP(wk_, slot_, cond_)                                           \
    do {                                                                       \
        if ((wk_).dbg && (cond_)) (wk_).dbg[slot_] = wall_clock64();          \
    } while (0)

// Wave-wide sum in lane 63 with DPP row shifts / broadcasts (no LDS traffic, fixed order).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_add(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, ROW_MASK == 0xf);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, ROW_MASK == 0xf);
    return v + __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum_lane63(double v) {
    v = dpp_add<0x111, 0xf>(v);  // row_shr:1
    v = dpp_add<0x112, 0xf>(v);  // row_shr:2
    v = dpp_add<0x114, 0xf>(v);  // row_shr:4
    v = dpp_add<0x118, 0xf>(v);  // row_shr:8   -> lane 15 of every row holds the row sum
    v = dpp_add<0x142, 0xa>(v);  // row_bcast:15 into rows 1 and 3
    v = dpp_add<0x143, 0xc>(v);  // row_bcast:31 into rows 2 and 3 -> lane 63 holds the total
    return v;
}

__device__ __forceinline__ double readlane_f64(double v, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

// Table-driven fp64 exp for the pair kernel: with T = 2^FEXP_TB table entries, x = (T m + j) ln2/T + r,
// exp(x) = 2^m * tab[j] * (1 + r + r^2/2 + r^3/6 + r^4/24), |r| <= ln2/(2T): for T = 256 the dropped
// r^5/120 term is < 4e-17 (T = 64 keeps it).  n = rint(T x / ln2) comes out of the low mantissa
// bits of x*C + 1.5*2^52 (no cvt), 2^m is an integer add into the exponent field.
// Inputs below -700 are clamped (result ~1e-304 instead of 0); the exponents of
// this path are bounded above by log(var_a var_b).  The reduction uses a single
// ln2/T constant: its rounding contributes |x| * 1.1e-16 relative error, the same
// size as the rounding of the exponent x itself.  Split into three phases so that a
// wave keeps all its table reads in flight while it evaluates the polynomials.
#ifndef FEXP_TB
#define FEXP_TB 8    // log2 of the table size: 256 entries let the polynomial stop at degree 4 (|r| <= ln2/512)
#endif
#define FEXP_TN (1 << FEXP_TB)
#if FEXP_TB == 6
#define FEXP_C 92.332482616893656758       /* 64 / ln2 */
#define FEXP_LN2_64 0.010830424696249145   /* ln2 / 64 */
#else
#define FEXP_C 369.3299304675746       /* 256 / ln2 */
#define FEXP_LN2_64 0.0027076061740622863   /* ln2 / 256 */
#endif
#define FEXP_MAGIC 6755399441055744.0      /* 1.5 * 2^52 */

#ifndef PAIR_OPT
#define PAIR_OPT 0   // experiment bits (tools): 1 no inline asm, 2 no clamp, 4 no sched barriers
#endif
__device__ __forceinline__ double fexp_clamp(double x) {
#if PAIR_OPT & 2
    return x;
#elif PAIR_OPT & 1
    return fmax(x, -700.0);
#else
    double y;
    const double lo = -700.0;
    asm("v_max_f64 %0, %1, %2" : "=v"(y) : "v"(x), "s"(lo));  // one instruction: no canonicalising pre-max
    return y;
#endif
}
__device__ __forceinline__ double fexp_t(double x) { return fma(x, FEXP_C, FEXP_MAGIC); }
__device__ __forceinline__ double fexp_poly(double x, double t) {
    const double nf = t - FEXP_MAGIC;
    const double r = fma(nf, -FEXP_LN2_64, x);
#if FEXP_TB == 6
    double q = fma(r, 1.0 / 120.0, 1.0 / 24.0);
    q = fma(r, q, 1.0 / 6.0);
#else
    double q = fma(r, 1.0 / 24.0, 1.0 / 6.0);
#endif
    q = fma(r, q, 0.5);
    q = fma(r, q, 1.0);
    return r * q;
}
__device__ __forceinline__ double fexp_finish(double tv, double pm1, double t) {
    const double res = fma(tv, pm1, tv);
    const int lo = __double2loint(t) & ~(FEXP_TN - 1);
    int hi;
#if PAIR_OPT & 1
    hi = __double2hiint(res) + (lo << (20 - FEXP_TB));
#else
    asm("v_lshl_add_u32 %0, %1, %3, %2" : "=v"(hi) : "v"(lo), "v"(__double2hiint(res)), "n"(20 - FEXP_TB));  // exponent += n >> FEXP_TB
#endif
    return __hiloint2double(hi, __double2loint(res));
}
__device__ __forceinline__ double fexp(double x, const double* __restrict__ tab) {
    x = fexp_clamp(x);
    const double t = fexp_t(x);
    const double tv = tab[__double2loint(t) & (FEXP_TN - 1)];
    return fexp_finish(tv, fexp_poly(x, t), t);
}

// Pivoted Gauss-Jordan on an n x nc augmented matrix held in LDS (row-major,
// ld = nc), ping-ponging between two buffers: one barrier per pivot step.
// Called by the whole workgroup.  Returns the buffer holding [I | A^{-1} B];
// det = det(A) (valid in every thread).  General (slow) path.
__device__ double* gauss_jordan(double* G0, double* G1, int n, int nc, double& det) {
    double* cur = G0;
    double* nxt = G1;
    det = 1.0;
    for (int k = 0; k < n; ++k) {
        __syncthreads();
        int p = k;
        double best = fabs(cur[k * nc + k]);
        for (int r = k + 1; r < n; ++r) {
            const double v = fabs(cur[r * nc + k]);
            if (v > best) {
                best = v;
                p = r;
            }
        }
        const double piv = cur[p * nc + k];
        det *= (p == k) ? piv : -piv;
        for (int e = threadIdx.x; e < n * nc; e += blockDim.x) {
            const int r = e / nc, c = e - r * nc;
            const double pk = cur[p * nc + c] / piv;
            double val;
            if (r == k) {
                val = pk;
            } else {
                const int rs = (r == p) ? k : r;
                val = fma(-cur[rs * nc + k], pk, cur[rs * nc + c]);
            }
            nxt[e] = val;
        }
        double* tmp = cur;
        cur = nxt;
        nxt = tmp;
    }
    __syncthreads();
    return cur;
}

// 1/x to fp64 accuracy: hardware reciprocal estimate + two Newton steps (short dependency chain;
// the IEEE division sequence is ~3x longer and sits on the critical path of every pivot).
__device__ __forceinline__ double fast_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}

// Unpivoted Gauss-Jordan with the matrix in registers: lane c of ONE wave holds column c of the
// DT x 2DT augmented matrix [A | B].  Per pivot the multipliers (column k) are broadcast from lane k
// with v_readlane; no LDS, no barriers (an LDS-broadcast variant measured slower at DT = 12).  On return
// lanes DT..2DT-1 hold the columns of A^{-1} B.  For SPD / diagonally-similar-to-SPD systems (no pivoting).
template <int DT>
__device__ __forceinline__ double gj_wave(double (&a)[DT], double* colbuf, int lane) {
    double det = 1.0;
#pragma unroll
    for (int k = 0; k < DT; ++k) {
        double f[DT];
#pragma unroll
        for (int r = 0; r < DT; ++r) f[r] = readlane_f64(a[r], k);
        (void)colbuf;
        (void)lane;
        const double piv = f[k];
        det *= piv;
        const double pk = a[k] * fast_rcp(piv);
#pragma unroll
        for (int r = 0; r < DT; ++r)
            if (r != k) a[r] = fma(-f[r], pk, a[r]);
        a[k] = pk;
    }
    return det;
}

// ------------------------------------------------------------------ rewards
// Unpivoted Gauss-Jordan for symmetric positive definite systems on an n x nc augmented
// matrix in LDS (ping-pong buffers, one barrier per pivot, one element per thread when
// n*nc <= blockDim).  Returns the buffer holding [I | A^{-1} B]; det in every thread.
__device__ double* gauss_jordan_spd(double* G0, double* G1, int n, int nc, double& det) {
    double* cur = G0;
    double* nxt = G1;
    det = 1.0;
    for (int k = 0; k < n; ++k) {
        __syncthreads();
        const double piv = cur[k * nc + k];
        det *= piv;
        for (int e = threadIdx.x; e < n * nc; e += blockDim.x) {
            const int r = e / nc, c = e - r * nc;
            const double pk = cur[k * nc + c] / piv;
            nxt[e] = (r == k) ? pk : fma(-cur[r * nc + k], pk, cur[r * nc + c]);
        }
        double* tmp = cur;
        cur = nxt;
        nxt = tmp;
    }
    __syncthreads();
    return cur;
}

// exp(-scale q / 2) / sqrt(det(I + scale S W)),  q = d^T W (I + scale S W)^{-1} d,  d = m - t
// (rewards.py:32-48; scale 1 -> mean, scale 2 -> second moment).  ws: LDS scratch.
__device__ double exp_reward_moment(const RewardDev& rw, int E, double scale, const double* mx, const double* sx,
                                    double* ws) {
    const int t = threadIdx.x;
    double result;
    if (rw.rank >= 0) {
        // W = F F^T (symmetric PSD): q = y^T (I + scale F^T S F)^{-1} y with y = F^T d, and
        // det(I + scale S W) = det(I_r + scale F^T S F): an SPD r x r system, no pivoting needed.
        const int r = rw.rank;
        double* y = ws;              // [E]
        double* Fl = y + E;          // [E*E]  F staged in LDS
        double* SF = Fl + E * E;     // [E*E]
        double* A = SF + E * E;      // [E*E]
        double* slot = A + E * E + E * (E + 1);
        for (int e2 = t; e2 < E * r; e2 += blockDim.x) Fl[e2] = rw.F[e2];
        if (t < E) slot[2 + t] = mx[t] - rw.t[t];
        __syncthreads();
        const double* d = slot + 2;
        for (int k = t; k < r; k += blockDim.x) {
            double acc = 0.0;
            _Pragma("unroll 8") for (int e = 0; e < E; ++e) acc = fma(Fl[e * r + k], d[e], acc);
            y[k] = acc;
        }
        for (int e2 = t; e2 < E * r; e2 += blockDim.x) {
            const int e = e2 / r, k = e2 - e * r;
            double acc = 0.0;
            _Pragma("unroll 8") for (int f = 0; f < E; ++f) acc = fma(sx[e * E + f], Fl[f * r + k], acc);
            SF[e2] = acc;
        }
        __syncthreads();
        for (int e2 = t; e2 < r * r; e2 += blockDim.x) {
            const int k = e2 / r, l = e2 - k * r;
            double acc = 0.0;
            _Pragma("unroll 8") for (int e = 0; e < E; ++e) acc = fma(Fl[e * r + k], SF[e * r + l], acc);
            A[e2] = fma(scale, acc, (k == l) ? 1.0 : 0.0);
        }
        __syncthreads();
        // [A | y] -> A^{-1} y; A is SPD so no pivoting is needed
        double* G0 = SF;             // SF is dead from here on: reuse as the augmented matrix
        double* G1 = A + E * E;      // [E*(E+1)]
        const int nc = r + 1;
        for (int e2 = t; e2 < r * nc; e2 += blockDim.x) {
            const int k = e2 / nc, l = e2 - k * nc;
            G1[e2] = (l < r) ? A[k * r + l] : y[k];
        }
        __syncthreads();
        for (int e2 = t; e2 < r * nc; e2 += blockDim.x) G0[e2] = G1[e2];
        double det;
        const double* res = gauss_jordan_spd(G0, G1, r, nc, det);
        if (t == 0) {
            double q = 0.0;
            for (int k = 0; k < r; ++k) q = fma(y[k], res[k * nc + r], q);
            slot[0] = exp(-0.5 * scale * q) / sqrt(det);
        }
        __syncthreads();
        result = slot[0];
        __syncthreads();
    } else {
        // general W: aug = [(I + scale S W)^T | W^T] -> X^T, X = W (I + scale S W)^{-1}
        const int nc = 2 * E;
        double* G0 = ws;
        double* G1 = G0 + 2 * E * E;
        double* slot = G1 + 2 * E * E;
        for (int e = t; e < E * nc; e += blockDim.x) {
            const int r = e / nc, c = e - r * nc;
            double v;
            if (c < E) {
                double sw = 0.0;  // (S W)[c][r]
                _Pragma("unroll 8") for (int k = 0; k < E; ++k) sw = fma(sx[c * E + k], rw.W[k * E + r], sw);
                v = fma(scale, sw, (r == c) ? 1.0 : 0.0);
            } else {
                v = rw.W[(c - E) * E + r];
            }
            G0[e] = v;
        }
        double det;
        double* res = gauss_jordan(G0, G1, E, nc, det);
        if (t == 0) {
            double q = 0.0;
            for (int r = 0; r < E; ++r) {
                double acc = 0.0;
                _Pragma("unroll 8") for (int c = 0; c < E; ++c) acc = fma(res[c * nc + E + r], mx[c] - rw.t[c], acc);
                q = fma(mx[r] - rw.t[r], acc, q);
            }
            slot[0] = exp(-0.5 * scale * q) / sqrt(det);
        }
        __syncthreads();
        result = slot[0];
        __syncthreads();
    }
    return result;
}

__host__ __device__ inline size_t reward_lds_doubles(int E) { return (size_t)E + 4 * (size_t)E * E + (size_t)E * (E + 1) + (size_t)E + 16; }

// mean (and variance) of the combined reward at (mx, sx) held in LDS (rewards.py:19-81)
__device__ __forceinline__ void reward_eval(int n, const RewardDev* rws, int E, const double* mx, const double* sx, double* ws,
                            bool want_var, double& mu_out, double& var_out) {
    double mu = 0.0, var = 0.0;
    for (int i = 0; i < n; ++i) {
        const RewardDev& rw = rws[i];
        double m_i = 0.0, v_i = 0.0;
        if (rw.kind == PILCO_REWARD_EXPONENTIAL) {
            m_i = exp_reward_moment(rw, E, 1.0, mx, sx, ws);
            if (want_var) v_i = exp_reward_moment(rw, E, 2.0, mx, sx, ws) - m_i * m_i;
        } else {  // linear: rewards.py:58-61
            _Pragma("unroll 8") for (int k = 0; k < E; ++k) m_i = fma(mx[k], rw.W[k], m_i);
            if (want_var)
                for (int r = 0; r < E; ++r)
                    _Pragma("unroll 8") for (int c = 0; c < E; ++c) v_i = fma(rw.W[r] * sx[r * E + c], rw.W[c], v_i);
        }
        mu = fma(rw.coef, m_i, mu);
        var = fma(rw.coef * rw.coef, v_i, var);
    }
    mu_out = mu;
    var_out = var;
}

// ------------------------------------------------------------------ prep
// 512 threads per workgroup = the whole register file of one CU.  The 256 rows of the chunk are
// handled twice in parallel: threads 0..255 ("group 0") build the row-side operand, threads
// 256..511 ("group 1") the column-side operand; on a diagonal pair both operands are the same
// vectors, so group 0 writes both and group 1 does the mean / input-output covariance sums.
// Mean / input-output-covariance sums of local output al over row chunk chm (mgpr.py:99-118), by one spare workgroup
// of the prep launch: T = Lambda^-1 B^-1 Lambda^-1 = (s + Lambda^2)^-1 by a register Gauss-Jordan in wave 0 while the
// other threads already have their point in flight; lb_i = exp(-zeta_i^T T zeta_i / 2) beta_i; partial c g and c T h
// into mean_part[al][chm][1 + D].
template <int DT>
__device__ __forceinline__ void prep_mean_block(const MMModel& md, const MMWork& wk, int al, int chm, double* sm) {
    const int D = md.D, npad = md.npad;
    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);   // wave-uniform: the role branches below are scalar branches
    const int a = al * wk.nranks + wk.rank;   // global output: the owner of (a,a) owns output a
    double* s_m = sm;                  // [DT]
    double* s_ia = s_m + DT;           // [DT] 1 / l_a
    double* s_s = s_ia + DT;           // [DT*DT] input covariance (D x D, ld D)
    double* s_T = s_s + DT * DT;       // [DT*DT] ld DT
    double* s_sc = s_T + DT * DT;      // [4]
    double* red = s_sc + 4;            // 9 * (DT + 1)
    double* colbuf = red + 9 * (DT + 1);   // [2 DT]
    double* zst = colbuf + 2 * DT;         // [512][DT + 1] centred points of the chunk's first 512 rows
    double* bst = zst + 512 * (DT + 1);    // [512] their beta_a
    constexpr int LDZ = DT + 1;
    if (t < DT) {
        s_m[t] = (t < D) ? wk.in_m[t] : 0.0;
        s_ia[t] = (t < D) ? 1.0 / md.ls[a * D + t] : 0.0;
    }
    for (int e = t; e < D * D; e += 512) s_s[e] = wk.in_s[e];
    for (int e = t; e < DT * DT; e += 512) s_T[e] = 0.0;
    __syncthreads();
    const int rpc = npad / wk.NCHM;
    const int i_begin = chm * rpc, i_end = i_begin + rpc;
    if (w == 0) {
        // [B | I],  B = Lambda^-1 s Lambda^-1 + I; T = Lambda^-1 B^-1 Lambda^-1   (mgpr.py:103-111)
        const double var_a = md.var[a];
        double col[DT];
        const int c = lane;
#pragma unroll
        for (int r = 0; r < DT; ++r) {
            double v = 0.0;
            if (c < DT) {
                v = (r == c) ? 1.0 : 0.0;
                if (r < D && c < D) v = fma(s_s[r * D + c], s_ia[r] * s_ia[c], v);
            } else if (c < 2 * DT) {
                v = (c - DT == r) ? 1.0 : 0.0;
            }
            col[r] = v;
        }
        const double detB = gj_wave<DT>(col, colbuf, lane);
        if (c >= DT && c < DT + D) {
            const int cc = c - DT;
#pragma unroll
            for (int r = 0; r < DT; ++r)
                if (r < D) s_T[r * DT + cc] = col[r] * s_ia[r] * s_ia[cc];
        }
        if (lane == 0) s_sc[1] = var_a / sqrt(detB);
    } else {
        // the other seven waves stage the centred points of the chunk's first 512 rows meanwhile
        const int idx = (w - 1) * 64 + lane;   // 0..447
        for (int e = idx; e < 512 * D; e += 448) {
            const int d = e >> 9, r = e & 511;
            const int i = i_begin + r;
            zst[r * LDZ + d] = (i < md.n && i < i_end) ? md.Pt[(long)d * npad + i] - s_m[d] : 0.0;
        }
        for (int r = idx; r < 512; r += 448) bst[r] = (i_begin + r < i_end) ? md.beta[(long)a * npad + i_begin + r] : 0.0;
    }
    __syncthreads();
    double g = 0.0;
    double h[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d) h[d] = 0.0;
    for (int i = i_begin + t; i < i_end; i += 512) {   // lb_i = exp(-zeta^T T zeta / 2) beta_i   (mgpr.py:113)
        const bool staged = (i - i_begin) < 512;
        double zeta[DT];
#pragma unroll
        for (int d = 0; d < DT; ++d)
            zeta[d] = (d < D) ? (staged ? zst[(i - i_begin) * LDZ + d] : (i < md.n ? md.Pt[(long)d * npad + i] - s_m[d] : 0.0)) : 0.0;
        double tz[DT];
#pragma unroll
        for (int r = 0; r < DT; ++r) tz[r] = 0.0;
#pragma unroll
        for (int c = 0; c < DT; ++c) {
            double trow[DT];
#pragma unroll
            for (int r = 0; r < DT; ++r) trow[r] = s_T[c * DT + r];
#pragma unroll
            for (int r = 0; r < DT; ++r) tz[r] = fma(trow[r], zeta[c], tz[r]);
            if ((c & 1) == 1) __builtin_amdgcn_sched_barrier(0);
        }
        double q = 0.0;
#pragma unroll
        for (int r = 0; r < DT; ++r) q = fma(zeta[r], tz[r], q);
        const double lb = exp(-0.5 * q) * (staged ? bst[i - i_begin] : md.beta[(long)a * npad + i]);
        g += lb;
#pragma unroll
        for (int d = 0; d < DT; ++d) h[d] = fma(zeta[d], lb, h[d]);
    }
    g = wave_sum_lane63(g);
    if (lane == 63) red[w * (DT + 1)] = g;
#pragma unroll
    for (int d = 0; d < DT; ++d) {
        const double v = wave_sum_lane63(h[d]);
        if (lane == 63) red[w * (DT + 1) + 1 + d] = v;
    }
    __syncthreads();
    // block sums of g and h (fixed order), then the M and V contributions of this row chunk:
    // c g (mgpr.py:117) and c T h (mgpr.py:118, V = c tiL^T lb = c T sum_i zeta_i lb_i)
    double* hs = red + 8 * (DT + 1);  // [DT + 1]
    if (t < 1 + D) {
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < 8; ++k) acc += red[k * (DT + 1) + t];
        hs[t] = acc;
    }
    __syncthreads();
    if (t < 1 + D) {
        double v;
        if (t == 0) {
            v = s_sc[1] * hs[0];
        } else {
            double acc = 0.0;
            _Pragma("unroll 8") for (int k = 0; k < D; ++k) acc = fma(s_T[(t - 1) * DT + k], hs[1 + k], acc);
            v = s_sc[1] * acc;
        }
        wk.mean_part[((long)al * wk.NCHM + chm) * (1 + D) + t] = v;   // indexed by the LOCAL output number
    }
}

template <int DT>
__global__ __launch_bounds__(512) void k_mm_prep(MMModel md, MMWork wk, PrepReward pr) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    if ((int)blockIdx.x >= wk.PL) {
        // spare workgroups of the launch: first the mean parts (local output, row chunk), then the reward
        const int idx = ((int)blockIdx.x - wk.PL) * (int)gridDim.y + (int)blockIdx.y;
        const int nmean = wk.EL * wk.NCHM;
        const int slot = 64 + 2 * (blockIdx.y * gridDim.x + blockIdx.x);
        if (wk.dbg && threadIdx.x == 0 && slot < 958) wk.dbg[slot] = wall_clock64();
        if (idx < nmean) {
            prep_mean_block<DT>(md, wk, idx / wk.NCHM, idx % wk.NCHM, sm);
            if (wk.dbg && threadIdx.x == 0 && slot < 958) wk.dbg[slot + 1] = wall_clock64();
            return;
        }
        // mean reward of the current (pre-propagation) state (rewards.py:19-81, pilco.py:133)
        if (idx != nmean || pr.n <= 0) return;
        const int E = pr.E, t = threadIdx.x;
        double* mx = sm;              // [E]
        double* sx = mx + E;          // [E][E]
        double* ws = sx + E * E;      // reward_lds_doubles(E)
        if (t < E) mx[t] = pr.m_x[t];
        for (int e = t; e < E * E; e += blockDim.x) sx[e] = pr.s_x[e];
        __syncthreads();
        double mu, var;
        reward_eval(pr.n, pr.rw, E, mx, sx, ws, false, mu, var);
        if (t == 0) pr.reward[0] += mu;
        if (wk.dbg && t == 0 && slot < 958) wk.dbg[slot + 1] = wall_clock64();
        return;
    }
    const int D = md.D, npad = md.npad;
    double* s_m = sm;
    double* s_ia2 = s_m + DT;
    double* s_ib2 = s_ia2 + DT;
    double* s_s = s_ib2 + DT;          // [DT*DT] input covariance (D x D, ld D)
    double* s_Q = s_s + DT * DT;       // [DT*DT] ld DT
    double* s_sc = s_Q + DT * DT;      // [4] isdet
    double* colbuf = s_sc + 4;         // DT (unused by the readlane Gauss-Jordan)
    double* zst = colbuf + DT;         // [256][DT + 1] centred points of the first 256 rows
    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);   // wave-uniform: the role branches below are scalar branches
    const int grp = t >> 8, tl = t & 255;
    const int pl = blockIdx.x, ch = blockIdx.y;
    const bool dbg0 = (t == 0 && pl == 0 && ch == 0);
    DBG_STAMP(wk, 0, dbg0);
    if (wk.dbg && t == 0) wk.dbg[64 + 2 * (blockIdx.y * gridDim.x + blockIdx.x)] = wall_clock64();
    int a, b;
    local_pair_ab(wk, md.E, pl, a, b);
    if (t < DT) {
        double la = 1.0, lb = 1.0, mm = 0.0;
        if (t < D) {
            mm = wk.in_m[t];
            la = md.ls[a * D + t];
            lb = md.ls[b * D + t];
        }
        s_m[t] = mm;
        s_ia2[t] = (t < D) ? 1.0 / (la * la) : 0.0;
        s_ib2[t] = (t < D) ? 1.0 / (lb * lb) : 0.0;
    }
    for (int e = t; e < D * D; e += 512) s_s[e] = wk.in_s[e];
    for (int e = t; e < DT * DT; e += 512) s_Q[e] = 0.0;   // padded rows / columns of Q stay zero
    __syncthreads();
    DBG_STAMP(wk, 1, dbg0);
    const int rpc = npad / wk.NCH;
    const int i_begin = ch * rpc, i_end = i_begin + rpc;
    // The centred points of the chunk's first 256 rows are staged in LDS by the seven waves that do not run the
    // Gauss-Jordan, so their load latency (and the log of the signal variance) hides behind that phase.
    // side 0 (threads 0..255): x = zeta / la^2 -> row operand (2 Q z | u | 1); side 1: x = zeta / lb^2 -> column
    // operand (w | 1 | v).  A diagonal pair (a == b) is not special here: its mean part runs in prep_mean_block.
    const int side = grp;
    constexpr int LDZ = DT + 1;
    if (w != 0) {
        const int idx = (w - 1) * 64 + lane;   // 0..447
        for (int e = idx; e < 256 * D; e += 448) {
            const int d = e >> 8, r = e & 255;
            const int i = i_begin + r;
            zst[r * LDZ + d] = (i < md.n && i < i_end) ? md.Pt[(long)d * npad + i] - s_m[d] : 0.0;
        }
    }
    const double logvar = log(md.var[side ? b : a]);
    if (wk.abl & 2) {
        if (t == 0) s_sc[0] = 1.0;
    } else if (w == 0) {
        // [R | s],  R = s diag(la^-2 + lb^-2) + I        (mgpr.py:121-124,129); padded with identity
        double col[DT];
        const int c = lane;
#pragma unroll
        for (int r = 0; r < DT; ++r) {
            double v = 0.0;
            if (c < DT) {
                v = (r == c) ? 1.0 : 0.0;
                if (r < D && c < D) v = fma(s_s[r * D + c], s_ia2[c] + s_ib2[c], v);
            } else if (c < 2 * DT) {
                const int cc = c - DT;
                if (r < D && cc < D) v = s_s[r * D + cc];
            }
            col[r] = v;
        }
        const double det = gj_wave<DT>(col, colbuf, lane);
        if (c >= DT && c < DT + D) {
#pragma unroll
            for (int r = 0; r < DT; ++r)
                if (r < D) s_Q[r * DT + (c - DT)] = 0.5 * col[r];
        }
        if (lane == 0) {
            s_sc[0] = 1.0 / sqrt(det);
            if (ch == 0) wk.pair_isdet[pl] = s_sc[0];
        }
    }
    __syncthreads();
    DBG_STAMP(wk, 2, dbg0);
    const int KP = wk.KP;
    double* At = wk.At + (long)pl * KP * npad;
    double* Bt = wk.Bt + (long)pl * KP * npad;
    auto row = [&](const int i, const bool valid, const double (&zeta)[DT]) {
        // y = Q x by columns of the symmetric Q: DT independent accumulators, one wide LDS row
        // read per column step (no LDS latency on the FMA chains).
        const double* il2 = side ? s_ib2 : s_ia2;   // padding entries are zero
        double x[DT], y[DT];
        double kk = logvar;
#pragma unroll
        for (int d = 0; d < DT; ++d) {
            x[d] = zeta[d] * il2[d];
            kk = fma(-0.5 * zeta[d], x[d], kk);
            y[d] = 0.0;
        }
#pragma unroll
        for (int c = 0; c < DT; ++c) {
            double qrow[DT];
#pragma unroll
            for (int r = 0; r < DT; ++r) qrow[r] = s_Q[c * DT + r];
#pragma unroll
            for (int r = 0; r < DT; ++r) y[r] = fma(qrow[r], x[c], y[r]);
            if ((c & 1) == 1) __builtin_amdgcn_sched_barrier(0);  // keep at most two rows of Q in flight
        }
        double quad = 0.0;
#pragma unroll
        for (int r = 0; r < DT; ++r) quad = fma(x[r], y[r], quad);
        const double uv = valid ? (kk + quad) : 0.0;
        const double one = valid ? 1.0 : 0.0;
        if (side == 0) {
#pragma unroll
            for (int r = 0; r < DT; ++r)
                if (r < D) store_wt(&At[(long)r * npad + i], 2.0 * y[r]);   // 2 Q z_i (0 on padded rows)
            store_wt(&At[(long)D * npad + i], uv);                           // u_i
            store_wt(&At[(long)(D + 1) * npad + i], one);
            for (int k = D + 2; k < KP; ++k) store_wt(&At[(long)k * npad + i], 0.0);
        } else {
#pragma unroll
            for (int r = 0; r < DT; ++r)
                if (r < D) store_wt(&Bt[(long)r * npad + i], x[r]);         // w_j
            store_wt(&Bt[(long)D * npad + i], one);
            store_wt(&Bt[(long)(D + 1) * npad + i], uv);                     // v_j
            for (int k = D + 2; k < KP; ++k) store_wt(&Bt[(long)k * npad + i], 0.0);
        }
    };
    if (!(wk.abl & 4)) {
        if (i_begin + tl < i_end) {   // first row of this thread: centred point from the LDS stage
            const int i = i_begin + tl;
            double zeta[DT];
#pragma unroll
            for (int d = 0; d < DT; ++d) zeta[d] = (d < D) ? zst[tl * LDZ + d] : 0.0;
            row(i, i < md.n, zeta);
        }
        for (int i = i_begin + tl + 256; i < i_end; i += 256) {   // chunks longer than 256 rows
            const bool valid = i < md.n;
            double zeta[DT];
#pragma unroll
            for (int d = 0; d < DT; ++d) zeta[d] = (d < D && valid) ? md.Pt[(long)d * npad + i] - s_m[d] : 0.0;
            row(i, valid, zeta);
        }
    }
    DBG_STAMP(wk, 3, dbg0);
    DBG_STAMP(wk, 4, dbg0);
    if (wk.dbg && t == 0) wk.dbg[65 + 2 * (blockIdx.y * gridDim.x + blockIdx.x)] = wall_clock64();
}

size_t prep_lds_bytes(int DT) {
    const size_t pair_blk = (size_t)4 * DT + 2 * (size_t)DT * DT + 4 + 256 * (size_t)(DT + 1);
    const size_t mean_blk = (size_t)2 * DT + 2 * (size_t)DT * DT + 4 + 9 * (size_t)(DT + 1) + 2 * (size_t)DT + 512 * (size_t)(DT + 2);
    return sizeof(double) * std::max(pair_blk, mean_blk);
}

int mm_kp(int D) { return round_up(D + 2, 4); }

static int device_cus() {
    int cus = 256;
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    return cus;
}
// Row chunks per pair and per output mean part: as many as keep the whole prep grid resident in ONE round (the kernel
// needs the full register file: one workgroup per CU), each chunk a multiple of 64 rows.  One CU is left for the
// reward workgroup; the mean-part workgroups take what the pair workgroups leave (at least one chunk per output).
void mm_prep_chunks(int npad, int PL, int EL, int* nch_out, int* nchm_out) {
    const int cus = device_cus();
    const int nb = npad / 64;
    int nch = 1;
    while (nch * 2 <= nb && nb % (nch * 2) == 0 && PL * nch * 2 + EL + 1 <= cus) nch *= 2;
    int nchm = 1;
    while (nchm * 2 <= nb && nb % (nchm * 2) == 0 && nchm * 2 <= nch && PL * nch + EL * nchm * 2 + 1 <= cus) nchm *= 2;
    *nch_out = nch;
    *nchm_out = nchm;
}

void launch_mm_prep(hipStream_t st, const MMModel& md, const MMWork& wk, const PrepReward* pr) {
    PrepReward none{};
    const PrepReward& r = pr ? *pr : none;
    const int spare = wk.EL * wk.NCHM + (r.n > 0 ? 1 : 0);   // mean-part workgroups, then the reward workgroup
    dim3 grid(wk.PL + (spare + wk.NCH - 1) / wk.NCH, wk.NCH);
    const int D = md.D;
    const size_t lds_rw = r.n > 0 ? sizeof(double) * ((size_t)r.E + (size_t)r.E * r.E + reward_lds_doubles(r.E)) : 0;
#define PREP(DT_)                                                                                          \
    do {                                                                                                   \
        const size_t lds_ = std::max(prep_lds_bytes(DT_), lds_rw);                                         \
        static size_t configured_ = 48 * 1024;   /* beyond the default dynamic-LDS limit: opt in once */   \
        if (lds_ > configured_) {                                                                          \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_mm_prep<DT_>),                       \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_);              \
            configured_ = lds_;                                                                            \
        }                                                                                                  \
        hipLaunchKernelGGL((k_mm_prep<DT_>), grid, dim3(512), lds_, st, md, wk, r);                        \
    } while (0)
    if (D <= 4) PREP(4);
    else if (D <= 8) PREP(8);
    else if (D <= 12) PREP(12);
    else if (D <= 16) PREP(16);
    else if (D <= 24) PREP(24);
    else PREP(32);
#undef PREP
}

// ------------------------------------------------------------------ pair kernel, MFMA
// Work item of a workgroup: (local pair, 64-row tile, column block); the four
// waves take consecutive column sub-ranges of JW columns.  Per 16-column step a
// wave issues 4*KC MFMAs (four 16-row tiles) and 16 exps per lane.
//
// a != b : S_num += beta_a,i beta_b,j L_ij; 16 per-row accumulators, beta_a applied at the end.
// a == b : L_aa and (beta beta^T - iK_a) are symmetric, so only column steps at or right of the
//          64x64 diagonal block are evaluated (weight 2 right of it): half the exps and half the
//          iK stream.  S_num += (beta_i beta_j - iK_ij) L_ij, one accumulator per result register.
#ifndef PAIR_RT
#define PAIR_RT 2      // 16-row MFMA tiles per wave (rows per work item = 16 * PAIR_RT); 2 measured best
#endif
// ablation switches for kernel experiments (tools/): never defined in product builds
#ifndef PAIR_ABL
#define PAIR_ABL 0
#endif
#if PAIR_ABL == 2
#define PAIR_ABL_TAB(v) (1.0 + 1e-9 * (double)(__double2loint(tt[i]) & (FEXP_TN - 1)))
#else
#define PAIR_ABL_TAB(v) (v)
#endif
#if PAIR_ABL == 3
#define PAIR_ABL_MFMA(a_, b_, e_) (d4{e_[0] + a_ * b_, e_[1] - a_, e_[2] + b_, e_[3] * 0.5})
#else
#define PAIR_ABL_MFMA(a_, b_, e_) __builtin_amdgcn_mfma_f64_16x16x4f64(a_, b_, e_, 0, 0, 0)
#endif
#ifndef PAIR_PF
#define PAIR_PF 2      // operand prefetch distance in 16-column steps
#endif
#ifndef PAIR_MINW
#define PAIR_MINW 1    // __launch_bounds__ min waves per SIMD for the pair kernel
#endif
// Buffer loads for the hot loop: wave-uniform resource (base pointer) + 32-bit per-lane byte offset + scalar byte
// offset, i.e. no 64-bit address arithmetic in the VALU stream (the fp64 pipe is the bottleneck of this kernel).
typedef unsigned v2u __attribute__((ext_vector_type(2)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t buf_rsrc(const double* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(base), 0, 0x7fffffff, 0x00020000);   // raw, untyped
}
__device__ __forceinline__ double buf_ld(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    const v2u v = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
    return __hiloint2double((int)v.y, (int)v.x);
}

template <int KC, bool DIAG>
__device__ __forceinline__ double pair_wave(const double* __restrict__ At, const double* __restrict__ Bt,
                                            const double* __restrict__ beta_a, const double* __restrict__ beta_b,
                                            const double* __restrict__ iKa, const double* __restrict__ tab, int npad, int i0,
                                            int jbeg, int jend, int lane) {
    constexpr int NE = 4 * PAIR_RT;  // exponent values per lane per 16-column step
    static_assert(PAIR_PF == 2, "the column loop is unrolled over a two-slot operand ring");
    const int lr = lane >> 4, lc = lane & 15;
    double af[PAIR_RT][KC];
#pragma unroll
    for (int rt = 0; rt < PAIR_RT; ++rt)
#pragma unroll
        for (int c = 0; c < KC; ++c) af[rt][c] = At[(long)(4 * c + lr) * npad + i0 + 16 * rt + lc];
    double acc[NE];
    double bi[NE];
    unsigned ik_off[NE];
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        acc[i] = 0.0;
        const int row = i0 + 16 * (i >> 2) + lr + 4 * (i & 3);
        bi[i] = beta_a[row];
        ik_off[i] = ((unsigned)(row - i0) * (unsigned)npad + (unsigned)lc) * 8u;   // relative to row i0: < 32 rows
    }
    unsigned b_off[KC];
#pragma unroll
    for (int c = 0; c < KC; ++c) b_off[c] = ((unsigned)(4 * c + lr) * (unsigned)npad + (unsigned)lc) * 8u;
    const unsigned bb_off = (unsigned)lc * 8u;
    const __amdgpu_buffer_rsrc_t rB = buf_rsrc(Bt), rbeta = buf_rsrc(beta_b);
    const __amdgpu_buffer_rsrc_t rIK = buf_rsrc(DIAG ? iKa + (long)i0 * npad : Bt);
    if (DIAG && jbeg < i0) jbeg = i0;  // columns left of the diagonal block are mirrored by the transposed tile
    double total = 0.0;
    // software pipeline: the operands of the column step two ahead are requested while this one is
    // evaluated (a first touch of Bt / beta misses the XCD's L2: ~2 us, more than one step)
    double ring[2][KC + 1];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
#pragma unroll
        for (int c = 0; c <= KC; ++c) ring[p][c] = 0.0;
        if (jbeg + 16 * p < jend) {
#pragma unroll
            for (int c = 0; c < KC; ++c) ring[p][c] = buf_ld(rB, b_off[c], (unsigned)(jbeg + 16 * p) * 8u);
            ring[p][KC] = buf_ld(rbeta, bb_off, (unsigned)(jbeg + 16 * p) * 8u);
        }
    }
    // one 16-column step on ring slot rg; the slot is refilled with the operands of column step j0 + 32
    auto step = [&](double (&rg)[KC + 1], const int j0) {
        double bf[KC];
#pragma unroll
        for (int c = 0; c < KC; ++c) bf[c] = rg[c];
        const double bb = rg[KC];
        double ik[NE];
        if (DIAG) {
#pragma unroll
            for (int i = 0; i < NE; ++i) ik[i] = buf_ld(rIK, ik_off[i], (unsigned)j0 * 8u);
        }
        // exponent tiles: C/D layout of the f64 MFMA is col = lane & 15, row = (lane >> 4) + 4 * reg
        double x[NE], tt[NE], tv[NE], pm[NE];
#pragma unroll
        for (int rt = 0; rt < PAIR_RT; ++rt) {
            d4 e = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int c = 0; c < KC; ++c) e = PAIR_ABL_MFMA(af[rt][c], bf[c], e);
#pragma unroll
            for (int r = 0; r < 4; ++r) x[4 * rt + r] = e[r];
        }
        if (PAIR_ABL != 4 && j0 + 32 < jend) {
#pragma unroll
            for (int c = 0; c < KC; ++c) rg[c] = buf_ld(rB, b_off[c], (unsigned)(j0 + 32) * 8u);
            rg[KC] = buf_ld(rbeta, bb_off, (unsigned)(j0 + 32) * 8u);
        }
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            x[i] = fexp_clamp(x[i]);
            tt[i] = fexp_t(x[i]);
        }
#pragma unroll
        for (int i = 0; i < NE; ++i) tv[i] = PAIR_ABL_TAB(tab[__double2loint(tt[i]) & (FEXP_TN - 1)]);
#if !(PAIR_OPT & 4)
        __builtin_amdgcn_sched_barrier(0);
#endif
        // Horner stages across all NE elements at once: NE independent fp64 chains per wave
        double rr[NE];
#pragma unroll
        for (int i = 0; i < NE; ++i) rr[i] = fma(tt[i] - FEXP_MAGIC, -FEXP_LN2_64, x[i]);
#if FEXP_TB == 6
#pragma unroll
        for (int i = 0; i < NE; ++i) pm[i] = fma(rr[i], 1.0 / 120.0, 1.0 / 24.0);
#pragma unroll
        for (int i = 0; i < NE; ++i) pm[i] = fma(rr[i], pm[i], 1.0 / 6.0);
#else
#pragma unroll
        for (int i = 0; i < NE; ++i) pm[i] = fma(rr[i], 1.0 / 24.0, 1.0 / 6.0);
#endif
#pragma unroll
        for (int i = 0; i < NE; ++i) pm[i] = fma(rr[i], pm[i], 0.5);
#pragma unroll
        for (int i = 0; i < NE; ++i) pm[i] = fma(rr[i], pm[i], 1.0);
#pragma unroll
        for (int i = 0; i < NE; ++i) pm[i] = rr[i] * pm[i];
        __builtin_amdgcn_sched_barrier(0);
#if PAIR_ABL == 1
#pragma unroll
        for (int i = 0; i < NE; ++i) { tv[i] = x[i]; pm[i] = 0.0; tt[i] = 0.0; }
#define FEXP_FINISH(a_, b_, c_) (a_)
#else
#define FEXP_FINISH(a_, b_, c_) fexp_finish(a_, b_, c_)
#endif
        if (DIAG) {
            double st[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int i = 0; i < NE; ++i)
                st[i & 3] = fma(fma(bi[i], bb, -ik[i]), FEXP_FINISH(tv[i], pm[i], tt[i]), st[i & 3]);
            const double wgt = (j0 >= i0 + 16 * PAIR_RT) ? 2.0 : 1.0;
            total = fma(wgt, (st[0] + st[1]) + (st[2] + st[3]), total);
        } else {
#pragma unroll
            for (int i = 0; i < NE; ++i) acc[i] = fma(bb, FEXP_FINISH(tv[i], pm[i], tt[i]), acc[i]);
        }
    };
    for (int j0 = jbeg; j0 < jend; j0 += 32) {
        step(ring[0], j0);
        if (j0 + 16 < jend) step(ring[1], j0 + 16);
    }
    if (!DIAG) {
#pragma unroll
        for (int i = 0; i < NE; ++i) total = fma(bi[i], acc[i], total);
    }
    return total;
}

template <int KC>
__global__ __launch_bounds__(256, PAIR_MINW) void k_mm_pair_tiled(MMModel md, MMWork wk, int NJB) {
    __shared__ double red[4];
    __shared__ double tab[FEXP_TN];
    for (int e = threadIdx.x; e < FEXP_TN; e += blockDim.x) tab[e] = wk.exp_tab[e];
    __syncthreads();
    const int npad = md.npad;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int jb = blockIdx.x % NJB, ti = blockIdx.x / NJB, pl = blockIdx.y;
    int a, b;
    local_pair_ab(wk, md.E, pl, a, b);
    const bool diag = (a == b) && (md.iK != nullptr);
    const int KP = wk.KP;
    const double* At = wk.At + (long)pl * KP * npad;
    const double* Bt = wk.Bt + (long)pl * KP * npad;
    const double* beta_a = md.beta + (long)a * npad;
    const double* beta_b = md.beta + (long)b * npad;
    const int i0 = ti * 16 * PAIR_RT;
    const int JB = npad / NJB, JW = JB / 4;
    const int jbeg = jb * JB + w * JW;
    double t1;
    if (diag)
        t1 = pair_wave<KC, true>(At, Bt, beta_a, beta_b, md.iK + (long)a * npad * npad, tab, npad, i0, jbeg, jbeg + JW, lane);
    else
        t1 = pair_wave<KC, false>(At, Bt, beta_a, beta_b, nullptr, tab, npad, i0, jbeg, jbeg + JW, lane);
    for (int off = 32; off > 0; off >>= 1) t1 += __shfl_down(t1, off);
    if (lane == 0) red[w] = t1;
    __syncthreads();
    if (threadIdx.x == 0) {
        double* out = wk.pair_part + ((long)pl * wk.NT + ti * NJB + jb) * 2;
        out[0] = (red[0] + red[1]) + (red[2] + red[3]);
        out[1] = 0.0;  // the trace term is already folded into out[0]
    }
}

// Stream-K form of the same computation: the column steps of all local (pair, row tile)
// rows are laid out on one line and cut into sk_waves equal ranges, so that every resident
// wave does the same number of 16-column steps (no tail, no per-tile launch overhead).
// A range touches at most two pairs; each wave writes one partial per touched pair.
// first column step of wave w: the cost line (diagonal steps weigh sk_ud units, the others sk_uo)
// is cut into sk_waves equal parts; a step belongs to the wave in whose part it starts.
// floor(a / b) for 0 <= a < 2^52, 0 < b: one fp64 division and an exact integer fix-up (the emulated 64-bit integer
// division is ~10x slower, and these quotients sit at the head of the glue kernel's critical path)
__host__ __device__ inline long div_floor(long a, long b) {
    long q = (long)((double)a / (double)b);
    while (q * b > a) --q;
    while ((q + 1) * b <= a) ++q;
    return q;
}
__host__ __device__ inline int sk_boundary_of(int w, int waves, int nd_steps, int total, int ud, int uo) {
    const long Ud = (long)nd_steps * ud;
    const long C = Ud + (long)(total - nd_steps) * uo;
    if (w >= waves) return total;
    const long x = div_floor((long)w * C, waves);
    if (x <= Ud) return (int)div_floor(x + ud - 1, ud);
    return nd_steps + (int)div_floor(x - Ud + uo - 1, uo);
}
// inverse: the last wave whose first step is <= x  (boundary(w) <= x  <=>  floor(w C / waves) <= cost(x))
__host__ __device__ inline int sk_wave_of(long x, int waves, int nd_steps, int total, int ud, int uo) {
    const long Ud = (long)nd_steps * ud;
    const long C = Ud + (long)(total - nd_steps) * uo;
    const long cx = (x <= nd_steps) ? x * ud : Ud + (x - nd_steps) * uo;
    long w = div_floor((cx + 1) * waves + C - 1, C) - 1;
    if (w > waves - 1) w = waves - 1;
    return (int)w;
}
// waves holding partials of local pair k: first wave, its slot for this pair, last wave
__host__ __device__ inline void sk_pair_waves(int k, int waves, int nd, int tdiag, int toff, int total, int ud, int uo,
                                              int& wlo, int& fslot, int& whi) {
    const long S0 = (k < nd) ? (long)k * tdiag : (long)nd * tdiag + (long)(k - nd) * toff;
    const long S1 = S0 + ((k < nd) ? tdiag : toff);
    const int nd_steps = nd * tdiag;
    wlo = sk_wave_of(S0, waves, nd_steps, total, ud, uo);
    fslot = sk_boundary_of(wlo, waves, nd_steps, total, ud, uo) < S0 ? 1 : 0;   // a wave that starts before the pair holds it second
    whi = sk_wave_of(S1 - 1, waves, nd_steps, total, ud, uo);
}
__device__ __forceinline__ int sk_boundary(const MMWork& wk, int w) {
    return sk_boundary_of(w, wk.sk_waves, wk.sk_nd * wk.sk_tdiag, wk.sk_total, wk.sk_ud, wk.sk_uo);
}
// row stride of the pair-major partial array: the largest number of waves touching one local pair, rounded up to 4
int mm_sk_maxw(const MMWork& wk) {
    int m = 4;
    for (int k = 0; k < wk.PL; ++k) {
        int wlo, fs, whi;
        sk_pair_waves(k, wk.sk_waves, wk.sk_nd, wk.sk_tdiag, wk.sk_toff, wk.sk_total, wk.sk_ud, wk.sk_uo, wlo, fs, whi);
        m = std::max(m, whi - wlo + 1);
    }
    return (m + 3) / 4 * 4;
}
int mm_sk_boundary(int w, int waves, int nd_steps, int total, int ud, int uo) {
    return sk_boundary_of(w, waves, nd_steps, total, ud, uo);
}

template <int KC>
__global__ __launch_bounds__(256, PAIR_MINW) void k_mm_pair_sk(MMModel md, MMWork wk) {
    __shared__ double tab[FEXP_TN];
    for (int e = threadIdx.x; e < FEXP_TN; e += blockDim.x) tab[e] = wk.exp_tab[e];
    __syncthreads();
    const int npad = md.npad, lane = threadIdx.x & 63;
    // XCD-aware placement: workgroups are dealt round-robin over the 8 XCDs (own L2 each), so workgroup b takes
    // position (b % 8) * (blocks / 8) + b / 8 of the cost line: the waves of one XCD cover one contiguous eighth of it
    // and its L2 holds the operands of ~1/8 of the pairs instead of all of them.
    int bpos = blockIdx.x;
    if ((gridDim.x & 7) == 0) bpos = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const int w = __builtin_amdgcn_readfirstlane(bpos * 4 + (threadIdx.x >> 6));
    const int NS = npad / 16;
    const int KP = wk.KP;
    DBG_STAMP(wk, 16, w == 0 && lane == 0);
    const int nd_steps = wk.sk_nd * wk.sk_tdiag;
    int step = sk_boundary(wk, w);
    const int end = sk_boundary(wk, w + 1);
    double out0 = 0.0, out1 = 0.0, cur = 0.0;
    int p0 = -1, p1 = -1, cur_pl = -1;
    while (step < end) {
        int pl, q, ti, sidx, cnt;
        const bool dg = step < nd_steps;
        if (dg) {
            pl = step / wk.sk_tdiag;
            q = step - pl * wk.sk_tdiag;
            ti = 0;
            int c = NS;
            while (q >= c) {
                q -= c;
                ++ti;
                c -= PAIR_RT;
            }
            sidx = ti * PAIR_RT + q;
            cnt = c - q;
        } else {
            const int r = step - nd_steps;
            pl = wk.sk_nd + r / wk.sk_toff;
            q = r - (pl - wk.sk_nd) * wk.sk_toff;
            ti = q / NS;
            sidx = q - ti * NS;
            cnt = NS - sidx;
        }
        const int seg = (cnt < end - step) ? cnt : (end - step);
        if (pl != cur_pl) {
            if (cur_pl >= 0) {  // a range touches at most two pairs
                out0 = cur;
                p0 = cur_pl;
            }
            cur_pl = pl;
            cur = 0.0;
        }
        int a, b;
        local_pair_ab(wk, md.E, pl, a, b);
        const double* At = wk.At + (long)pl * KP * npad;
        const double* Bt = wk.Bt + (long)pl * KP * npad;
        const double* beta_a = md.beta + (long)a * npad;
        const double* beta_b = md.beta + (long)b * npad;
        const int i0 = ti * 16 * PAIR_RT, jbeg = sidx * 16, jend = jbeg + seg * 16;
        if (dg)
            cur += pair_wave<KC, true>(At, Bt, beta_a, beta_b, md.iK + (long)a * npad * npad, tab, npad, i0, jbeg, jend, lane);
        else
            cur += pair_wave<KC, false>(At, Bt, beta_a, beta_b, nullptr, tab, npad, i0, jbeg, jend, lane);
        step += seg;
    }
    if (cur_pl >= 0) {
        if (p0 < 0) {
            out0 = cur;
            p0 = cur_pl;
        } else {
            out1 = cur;
            p1 = cur_pl;
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        out0 += __shfl_down(out0, off);
        out1 += __shfl_down(out1, off);
    }
    if (lane == 0) {
        // pair-major layout sk_part[pair][slot], slot = wave - (first wave of the pair): the reader (k_glue, one
        // workgroup on the step's critical path) then needs no index arithmetic at all.  A wave that enters a pair from
        // a previous one IS that pair's first wave (slot 0); only the first touched pair needs the closed form.
        if (p0 >= 0) {
            const long S0 = (p0 < wk.sk_nd) ? (long)p0 * wk.sk_tdiag : (long)wk.sk_nd * wk.sk_tdiag + (long)(p0 - wk.sk_nd) * wk.sk_toff;
            const int wlo = sk_wave_of(S0, wk.sk_waves, nd_steps, wk.sk_total, wk.sk_ud, wk.sk_uo);
            wk.sk_part[(long)p0 * wk.sk_maxw + (w - wlo)] = out0;
        }
        if (p1 >= 0) wk.sk_part[(long)p1 * wk.sk_maxw] = out1;
    }
    DBG_STAMP(wk, 17, w == 0 && lane == 0);
    DBG_STAMP(wk, 18, w == wk.sk_waves - 1 && lane == 0);
    if (wk.dbg && lane == 0 && (w & 7) == 0) wk.dbg[1024 + (w >> 3)] = wall_clock64();  // end stamp of every 8th wave
}


// ------------------------------------------------------------------ adjoint of the pair sums
// d e_ij/d m = P (z_i + w_j) and d e_ij/d s = (P y)(P y)^T / 2, so the reverse
// pass needs, per pair, only   r_i = sum_j W_ij L_ij,   c_j = sum_i W_ij L_ij,   m_i = sum_j W_ij L_ij w_j
// with W = beta_a beta_b^T (- iK_a on the diagonal pair).  A wave owns 16*BWD_RT rows and sweeps a range
// of columns; the exponent tile is computed TRANSPOSED (column operand as MFMA A, row operand as B) so
// that the weighted tile W.L lands in the B-operand layout of a second MFMA that contracts it with
// [w_j | 1]: moments and row sums cost 4 MFMAs per 16x16 tile and no VALU reductions.  The column sums of
// an off-diagonal pair run along the lanes of a DPP row: four row_shr adds per result register, the four
// waves of a workgroup keep their partial columns in separate LDS slices that are summed in a fixed order
// (diagonal pairs: c = r by symmetry).
// rowmom[pl][js][16][npad]: d < D -> m_i[d], d = D -> r_i, per column split js;
// cpart[pl - E][row block][npad]: column sums over the rows of one workgroup.
#ifndef BWD_RT
#define BWD_RT 2
#endif
// Per-step constants of the reverse pass, computed once by spare workgroups of the k_mm_bwd_pair launch and read by
// k_mm_bwd_post / k_mm_bwd_fin:   head[h][D*D + D + 2]
//   output a (h = a):      T = (s + Lambda_a^2)^-1 | u = T Vbar_a | mu = Mbar_a - sum_b (Sbar_ab + Sbar_ba) M_b | c_a
//   pair pl (h = E + pl):  P = (I + Lambda_ab s)^-1 | lambda_ab | kappa = Shat_ab / sqrt(det R_ab) | 0
// M_b comes from the mean partials the prep kernel of the same step left in wk.mean_part.
__device__ void bwd_head(const MMModel& md, const MMWork& wk, const double* __restrict__ bars, int h,
                         double* __restrict__ head, double* sm) {
    const int D = md.D, E = md.E, t = threadIdx.x, nc = 2 * D, nI = D * D;
    double* G0 = sm;               // [D][2D]
    double* G1 = G0 + D * nc;      // [D][2D]
    double* lam = G1 + D * nc;     // [D]
    const double* Mbar = bars;
    const double* Sbar = bars + E;
    const double* Vbar = bars + E + E * E;
    double* o = head + (long)h * (nI + D + 2);
    int a = h, b = h;
    if (h >= E) local_pair_ab(wk, E, h - E, a, b);
    if (t < D) {
        const double la = md.ls[a * D + t], lb = md.ls[b * D + t];
        lam[t] = (h < E) ? la * la : 1.0 / (la * la) + 1.0 / (lb * lb);
    }
    __syncthreads();
    for (int e = t; e < D * nc; e += 256) {
        const int r = e / nc, c = e - r * nc;
        double v;
        if (c >= D) v = (c - D == r) ? 1.0 : 0.0;
        else if (h < E) v = wk.in_s[r * D + c] + (r == c ? lam[r] : 0.0);        // s + Lambda_a^2
        else v = lam[r] * wk.in_s[r * D + c] + (r == c ? 1.0 : 0.0);             // I + Lambda_ab s
        G0[e] = v;
    }
    double det;
    const double* G = gauss_jordan(G0, G1, D, nc, det);   // inverse in G[:, D:]
    if (t < nI) o[t] = G[(t / D) * nc + D + (t % D)];
    if (h < E) {
        if (t < D) {
            double acc = 0.0;
            for (int c = 0; c < D; ++c) acc = fma(G[t * nc + D + c], Vbar[c * E + a], acc);
            o[nI + t] = acc;
        }
        if (t == 64) {
            double mu = Mbar[a];
            for (int bb = 0; bb < E; ++bb) {
                double Mb = 0.0;
                for (int ch = 0; ch < wk.NCHM; ++ch) Mb += wk.mean_part[((long)bb * wk.NCHM + ch) * (1 + D)];
                mu -= (Sbar[a * E + bb] + Sbar[bb * E + a]) * Mb;
            }
            double lp = 1.0;
            for (int d = 0; d < D; ++d) lp *= md.ls[a * D + d];
            o[nI + D] = mu;
            o[nI + D + 1] = md.var[a] * lp / sqrt(det);
        }
    } else {
        if (t < D) o[nI + t] = lam[t];
        if (t == 64) {
            const double shat = (a == b) ? Sbar[a * E + a] : Sbar[a * E + b] + Sbar[b * E + a];
            o[nI + D] = shat / sqrt(det);   // det(I + Lambda s) = det(s Lambda + I) = det R_ab
            o[nI + D + 1] = 0.0;
        }
    }
}

template <int KC>
__global__ __launch_bounds__(256) void k_mm_bwd_pair(MMModel md, MMWork wk, double* __restrict__ rowmom,
                                                    double* __restrict__ cpart, int njs, const double* __restrict__ bars,
                                                    double* __restrict__ head) {
    __shared__ double tab[FEXP_TN];
    extern __shared__ __attribute__((aligned(16))) double csl[];   // [4][jw]  (head workgroups: Gauss-Jordan scratch)
    if ((int)blockIdx.y >= wk.PL) {   // spare workgroups: the step's D x D inverses, one per output / pair
        const int h = ((int)blockIdx.y - wk.PL) * (int)(gridDim.x * gridDim.z) + (int)(blockIdx.z * gridDim.x + blockIdx.x);
        if (h < md.E + wk.PL) bwd_head(md, wk, bars, h, head, csl);
        return;
    }
    for (int e = threadIdx.x; e < FEXP_TN; e += blockDim.x) tab[e] = wk.exp_tab[e];
    __syncthreads();
    const int npad = md.npad, D = md.D, E = md.E;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int lr = lane >> 4, lc = lane & 15;
    const int pl = blockIdx.y, js = blockIdx.z, rb = blockIdx.x;
    int a, b;
    local_pair_ab(wk, E, pl, a, b);
    const int KP = wk.KP;
    const double* At = wk.At + (long)pl * KP * npad;
    const double* Bt = wk.Bt + (long)pl * KP * npad;
    const double* beta_a = md.beta + (long)a * npad;
    const double* beta_b = md.beta + (long)b * npad;
    const bool diag = (a == b);
    const double* iKa = (diag && md.iK) ? md.iK + (long)a * npad * npad : nullptr;
    const int jw = npad / njs, jbeg = js * jw, jend = jbeg + jw;
    const int ibase = rb * 64 * BWD_RT + w * 16 * BWD_RT;
    double rf[BWD_RT][KC], brow[BWD_RT];
    int irow[BWD_RT];
#pragma unroll
    for (int rt = 0; rt < BWD_RT; ++rt) {
        const bool ok = ibase + 16 * rt < npad;                  // wave-uniform; rows past the padding weigh zero
        irow[rt] = ok ? ibase + 16 * rt + lc : lc;
        brow[rt] = ok ? beta_a[irow[rt]] : 0.0;
#pragma unroll
        for (int c = 0; c < KC; ++c) rf[rt][c] = At[(long)(4 * c + lr) * npad + irow[rt]];
    }
    // buffer loads: uniform resource + per-lane byte offset + scalar column offset (no 64-bit VALU address math)
    const __amdgpu_buffer_rsrc_t rB = buf_rsrc(Bt), rbeta = buf_rsrc(beta_b);
    const __amdgpu_buffer_rsrc_t rIK = buf_rsrc(iKa ? iKa + (long)jbeg * npad : Bt);
    unsigned cf_off[KC], a2_off[4], bc_off[4], ik_off[BWD_RT][4];
#pragma unroll
    for (int c = 0; c < KC; ++c) cf_off[c] = ((unsigned)(4 * c + lr) * (unsigned)npad + (unsigned)lc) * 8u;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        bc_off[r] = (unsigned)(lr + 4 * r) * 8u;
        // rows of the column operand contracted by the second product: w_j (d < D) and the ones (d = D); lanes past
        // that repeat row D: their result rows (d > D of rowmom) are never read
        a2_off[r] = ((unsigned)(lc <= D ? lc : D) * (unsigned)npad + (unsigned)(4 * r + lr)) * 8u;
#pragma unroll
        for (int rt = 0; rt < BWD_RT; ++rt) ik_off[rt][r] = ((unsigned)(lr + 4 * r) * (unsigned)npad + (unsigned)irow[rt]) * 8u;
    }
    d4 acc[BWD_RT];
#pragma unroll
    for (int rt = 0; rt < BWD_RT; ++rt) acc[rt] = d4{0.0, 0.0, 0.0, 0.0};
    double* myslice = csl + w * jw;
    // the column sweep, specialised at compile time (branches inside the loop would fence the scheduler between the
    // eight exp evaluations of a step): MODE 0 off-diagonal pair (column sums), 1 diagonal pair with the iK stream,
    // 2 diagonal pair without it (RBF policy GP)
    auto sweep = [&](auto mode_c) {
        constexpr int MODE = decltype(mode_c)::value;
        for (int j0 = jbeg; j0 < jend; j0 += 16) {
            double cf[KC], a2[4], bcol[4];
            const unsigned so = (unsigned)j0 * 8u;
    #pragma unroll
            for (int c = 0; c < KC; ++c) cf[c] = buf_ld(rB, cf_off[c], so);
    #pragma unroll
            for (int r = 0; r < 4; ++r) {
                bcol[r] = buf_ld(rbeta, bc_off[r], so);
                a2[r] = buf_ld(rB, a2_off[r], so);
            }
            double csum[4] = {0.0, 0.0, 0.0, 0.0};
    #pragma unroll
            for (int rt = 0; rt < BWD_RT; ++rt) {
                d4 e = {0.0, 0.0, 0.0, 0.0};
    #pragma unroll
                for (int c = 0; c < KC; ++c)
                    e = __builtin_amdgcn_mfma_f64_16x16x4f64(cf[c], rf[rt][c], e, 0, 0, 0);   // e[r]: i = irow, j = j0+lr+4r
                double wl[4];
    #pragma unroll
                for (int r = 0; r < 4; ++r) {
                    double wgt = brow[rt] * bcol[r];
                    if (MODE == 1) wgt -= buf_ld(rIK, ik_off[rt][r], (unsigned)(j0 - jbeg) * (unsigned)npad * 8u);   // iK symmetric: coalesced along the rows
                    wl[r] = wgt * fexp(e[r], tab);
                    csum[r] += wl[r];
                }
    #pragma unroll
                for (int r = 0; r < 4; ++r) acc[rt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2[r], wl[r], acc[rt], 0, 0, 0);
            }
            if (MODE == 0) {
    #pragma unroll
                for (int r = 0; r < 4; ++r) {
                    double v = csum[r];
                    v = dpp_add<0x111, 0xf>(v);
                    v = dpp_add<0x112, 0xf>(v);
                    v = dpp_add<0x114, 0xf>(v);
                    v = dpp_add<0x118, 0xf>(v);   // lane 15 of every DPP row: sum over the 16 rows i
                    if (lc == 15) myslice[j0 - jbeg + lr + 4 * r] = v;
                }
            }
        }
    };
    if (!diag) sweep(std::integral_constant<int, 0>{});
    else if (iKa) sweep(std::integral_constant<int, 1>{});
    else sweep(std::integral_constant<int, 2>{});
    double* out = rowmom + ((long)pl * njs + js) * 16 * npad;
#pragma unroll
    for (int rt = 0; rt < BWD_RT; ++rt)
        if (ibase + 16 * rt < npad) {
#pragma unroll
            for (int r = 0; r < 4; ++r) out[(long)(lr + 4 * r) * npad + irow[rt]] = acc[rt][r];
        }
    if (!diag) {
        __syncthreads();
        double* cp = cpart + ((long)(pl - wk.EL) * gridDim.x + rb) * npad + jbeg;
        for (int jj = threadIdx.x; jj < jw; jj += 256)
            cp[jj] = (csl[jj] + csl[jw + jj]) + (csl[2 * jw + jj] + csl[3 * jw + jj]);
    }
}

// Reverse of the mean part (mgpr.py:99-118) for output a, including the -M M^T term of S:
// with T = (s + Lambda_a^2)^-1, l_i = beta_i exp(-zeta_i^T T zeta_i / 2), g = sum l_i, h = sum l_i zeta_i,
// u = T Vbar_a, mu = Mbar_a - sum_b (Sbar_ab + Sbar_ba) M_b, q_i = mu + zeta_i . u:
//   mbar_a = c (T sum l_i q_i zeta_i - g u),
//   sbar_a = -phi T / 2 + c T (sum l_i q_i zeta_i zeta_i^T) T / 2 - c (u (T h)^T + (T h) u^T) / 2,  phi = c (mu g + Vbar_a . T h).
// M_b is read from the mean partials the prep kernel of the same step left in wk.mean_part.
// stage 1 (a workgroup of k_mm_bwd_post): sums over the 64-point blocks rc, rc + nrc, ..:  mpart[a][rc][D*D + 2D + 1]
__device__ void bwd_mean_partial(const MMModel& md, const MMWork& wk, const double* __restrict__ head, int a, int rc,
                                 int nrc, double* __restrict__ mpart, double* sm) {
    const int D = md.D, npad = md.npad, t = threadIdx.x;
    const int nI = D * D, LD = D | 1;
    double* T = sm;                 // [D][D]
    double* zs = T + nI;            // [64][LD]
    double* lv = zs + 64 * LD;      // [64]
    double* lq = lv + 64;           // [64]
    double* u = lq + 64;            // [D + 2]: u | mu | c_a
    const double* hd = head + (long)a * (nI + D + 2);
    for (int e = t; e < nI + D + 2; e += 256) (e < nI ? T[e] : u[e - nI]) = hd[e];
    __syncthreads();
    const double mu = u[D];
    double acc = 0.0;
    for (int blk = rc; blk < npad / 64; blk += nrc) {
        if (t < 64) {
            const int i = blk * 64 + t;
            double l = 0.0, q = 0.0;
            if (i < md.n) {
                double quad = 0.0;
                q = mu;
                for (int d = 0; d < D; ++d) zs[t * LD + d] = md.Pt[(long)d * npad + i] - wk.in_m[d];
                for (int r = 0; r < D; ++r) {
                    double tz = 0.0;
                    for (int c = 0; c < D; ++c) tz = fma(T[r * D + c], zs[t * LD + c], tz);
                    quad = fma(zs[t * LD + r], tz, quad);
                    q = fma(zs[t * LD + r], u[r], q);
                }
                l = exp(-0.5 * quad) * md.beta[(long)a * npad + i];
            } else {
                for (int d = 0; d < D; ++d) zs[t * LD + d] = 0.0;
            }
            lv[t] = l;
            lq[t] = l * q;
        }
        __syncthreads();
        if (t < nI) {
            const int d = t / D, e2 = t - d * D;
            _Pragma("unroll 4") for (int ii = 0; ii < 64; ++ii) acc = fma(lq[ii] * zs[ii * LD + d], zs[ii * LD + e2], acc);
        } else if (t < nI + D) {
            const int d = t - nI;
            _Pragma("unroll 4") for (int ii = 0; ii < 64; ++ii) acc = fma(lq[ii], zs[ii * LD + d], acc);
        } else if (t < nI + 2 * D) {
            const int d = t - nI - D;
            _Pragma("unroll 4") for (int ii = 0; ii < 64; ++ii) acc = fma(lv[ii], zs[ii * LD + d], acc);
        } else if (t == nI + 2 * D) {
            _Pragma("unroll 4") for (int ii = 0; ii < 64; ++ii) acc += lv[ii];
        }
        __syncthreads();
    }
    if (t <= nI + 2 * D) mpart[((long)a * nrc + rc) * (nI + 2 * D + 1) + t] = acc;
}

// stage 2 (a workgroup of k_mm_bwd_fin): out[a][D + D*D]
__device__ void bwd_mean_final(const MMModel& md, const double* __restrict__ bars, const double* __restrict__ head, int a,
                               int nrc, const double* __restrict__ mpart, double* __restrict__ out, double* sm) {
    const int D = md.D, E = md.E, t = threadIdx.x;
    const int nI = D * D;
    double* T = sm;                 // [D][D]
    double* u = T + nI;             // [D + 2]: u | mu | c_a
    double* sc = u + D + 2;         // [2]  phi
    double* Th = sc + 2;            // [D]
    double* red = Th + D;           // [nI + 2 D + 1]   H2q | wq | h | g
    double* TH = red + nI + 2 * D + 1;  // [D][D]
    const double* Vbar = bars + E + E * E;
    const double* hd = head + (long)a * (nI + D + 2);
    for (int e = t; e < nI + D + 2; e += 256) (e < nI ? T[e] : u[e - nI]) = hd[e];
    if (t <= nI + 2 * D) {
        double acc = 0.0;
        for (int c = 0; c < nrc; ++c) acc += mpart[((long)a * nrc + c) * (nI + 2 * D + 1) + t];   // fixed order
        red[t] = acc;
    }
    __syncthreads();
    const double mu = u[D], c_a = u[D + 1];
    const double* H2q = red;
    const double* wq = red + nI;
    const double* h = red + nI + D;
    const double g = red[nI + 2 * D];
    if (t < D) {
        double acc2 = 0.0;
        for (int c = 0; c < D; ++c) acc2 = fma(T[t * D + c], h[c], acc2);
        Th[t] = acc2;
    }
    if (t < nI) {
        const int r = t / D, c = t - r * D;
        double acc2 = 0.0;
        for (int k = 0; k < D; ++k) acc2 = fma(T[r * D + k], H2q[k * D + c], acc2);
        TH[t] = acc2;
    }
    __syncthreads();
    if (t == 0) {
        double vTh = 0.0;
        for (int d = 0; d < D; ++d) vTh = fma(Vbar[d * E + a], Th[d], vTh);
        sc[0] = c_a * (mu * g + vTh);
    }
    __syncthreads();
    const double phi = sc[0];
    double* o = out + (long)a * (D + nI);
    if (t < nI) {
        const int r = t / D, c = t - r * D;
        double acc2 = 0.0;
        for (int k = 0; k < D; ++k) acc2 = fma(TH[r * D + k], T[k * D + c], acc2);
        o[D + t] = -0.5 * phi * T[r * D + c] + 0.5 * c_a * acc2 - 0.5 * c_a * (u[r] * Th[c] + Th[r] * u[c]);
    } else if (t < nI + D) {
        const int r = t - nI;
        double tw = 0.0;
        for (int c = 0; c < D; ++c) tw = fma(T[r * D + c], wq[c], tw);
        o[r] = c_a * (tw - g * u[r]);
    }
}

// Per unordered pair and row chunk: partial sums of  N_ab = sum_i r_i,  A = sum_i (r_i z_i + c_i w_i)  (D),
// I = sum_i (r_i z_i z_i^T + c_i w_i w_i^T + z_i m_i^T + m_i z_i^T)  (D x D).   part[pl][chunk][1 + D + D*D]
constexpr int BWD_RC = 16;  // row chunks per pair / output
__global__ __launch_bounds__(256) void k_mm_bwd_post(MMModel md, MMWork wk, const double* __restrict__ rowmom,
                                                    const double* __restrict__ cpart, int njs, int nrb,
                                                    double* __restrict__ part, int nrc,
                                                    const double* __restrict__ head, double* __restrict__ mpart) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int npad = md.npad, D = md.D, E = md.E, t = threadIdx.x;
    const int pl = blockIdx.x, rc = blockIdx.y;
    if (pl >= wk.PL) {   // the last E workgroup columns: mean part of output pl - PL
        bwd_mean_partial(md, wk, head, pl - wk.PL, rc, nrc, mpart, sm);
        return;
    }
    int a, b;
    local_pair_ab(wk, E, pl, a, b);
    const double* mom0 = rowmom + (long)pl * njs * 16 * npad;
    const double* cp = (a != b) ? cpart + (long)(pl - wk.EL) * nrb * npad : nullptr;
    const int LD = D | 1;         // odd row stride: the (d, e) readers of one point spread over the banks
    double* zs = sm;              // [64][LD]
    double* ws = zs + 64 * LD;    // [64][LD]
    double* ms = ws + 64 * LD;    // [64][LD]
    double* rs = ms + 64 * LD;    // [64]
    double* cs = rs + 64;         // [64]
    double* ia = cs + 64;         // [D] 1 / l_a^2
    double* ib = ia + D;          // [D] 1 / l_b^2
    double* mm = ib + D;          // [D] input mean
    const int nI = D * D;
    const int nblk = npad / 64;
    if (t < D) {
        const double la = md.ls[a * D + t], lb = md.ls[b * D + t];
        ia[t] = 1.0 / (la * la);
        ib[t] = 1.0 / (lb * lb);
        mm[t] = wk.in_m[t];
    }
    double acc = 0.0;
    for (int blk = rc; blk < nblk; blk += nrc) {
        const int i0 = blk * 64;
        __syncthreads();
        for (int e = t; e < 64 * D; e += 256) {
            const int d = e >> 6, ii = e & 63;   // consecutive threads -> consecutive points: coalesced
            const int i = i0 + ii;
            const bool valid = i < md.n;
            const double zeta = valid ? md.Pt[(long)d * npad + i] - mm[d] : 0.0;
            zs[ii * LD + d] = zeta * ia[d];
            ws[ii * LD + d] = zeta * ib[d];
            double mv = 0.0;
            if (valid)
                for (int q = 0; q < njs; ++q) mv += mom0[((long)q * 16 + d) * npad + i];
            ms[ii * LD + d] = mv;
        }
        if (t < 64) {
            const int i = i0 + t;
            const bool valid = i < md.n;
            double r = 0.0, c = 0.0;
            if (valid) {
                for (int q = 0; q < njs; ++q) r += mom0[((long)q * 16 + D) * npad + i];
                if (cp)
                    for (int q = 0; q < nrb; ++q) c += cp[(long)q * npad + i];
                else
                    c = r;
            }
            rs[t] = r;
            cs[t] = c;
        }
        __syncthreads();
        if (t < nI) {
            const int d = t / D, e2 = t - d * D;
            _Pragma("unroll 4") for (int ii = 0; ii < 64; ++ii) {
                const double zd = zs[ii * LD + d], ze = zs[ii * LD + e2];
                acc = fma(rs[ii] * zd, ze, acc);
                acc = fma(cs[ii] * ws[ii * LD + d], ws[ii * LD + e2], acc);
                acc = fma(zd, ms[ii * LD + e2], acc);
                acc = fma(ms[ii * LD + d], ze, acc);
            }
        } else if (t < nI + D) {
            const int d = t - nI;
            _Pragma("unroll 4") for (int ii = 0; ii < 64; ++ii) acc = fma(rs[ii], zs[ii * LD + d], fma(cs[ii], ws[ii * LD + d], acc));
        } else if (t == nI + D) {
            _Pragma("unroll 4") for (int ii = 0; ii < 64; ++ii) acc += rs[ii];
        }
    }
    double* o = part + ((long)pl * nrc + rc) * (1 + D + nI);
    if (t < nI) o[1 + D + t] = acc;
    else if (t < nI + D) o[1 + (t - nI)] = acc;
    else if (t == nI + D) o[0] = acc;
}

// Per pair, with P = (I + Lambda s)^-1 and kappa = Shat_ab / sqrt(det R_ab) from the step's head record:
//   mbar += kappa P A,   sbar += kappa (P I P^T / 2 - N (P Lambda + Lambda P^T) / 4)      
// out[E + pl][D + D*D].  bars = (Mbar [E] | Sbar [E][E] | Vbar [D][E]) on the device.  Workgroups past the pairs
// finish the mean part of one output each.
__global__ __launch_bounds__(256) void k_mm_bwd_fin(MMModel md, MMWork wk, const double* __restrict__ part, int nrc,
                                                   const double* __restrict__ bars, const double* __restrict__ head,
                                                   const double* __restrict__ mpart, double* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int D = md.D, E = md.E, t = threadIdx.x, pl = blockIdx.x;
    if (pl >= wk.PL) {
        bwd_mean_final(md, bars, head, pl - wk.PL, nrc, mpart, out, sm);
        return;
    }
    const int nI = D * D, rec = 1 + D + nI;
    double* Pm = sm;               // [D][D]
    double* lam = Pm + nI;         // [D + 2]: lambda | kappa
    double* Iv = lam + D + 2;      // [rec]  summed partials (N | A | I)
    double* PI = Iv + rec;         // [D][D]
    const double* hd = head + (long)(E + pl) * (nI + D + 2);
    for (int e = t; e < nI + D + 2; e += 256) (e < nI ? Pm[e] : lam[e - nI]) = hd[e];
    for (int e = t; e < rec; e += 256) {
        double acc = 0.0;
        for (int c = 0; c < nrc; ++c) acc += part[((long)pl * nrc + c) * rec + e];   // fixed order
        Iv[e] = acc;
    }
    __syncthreads();
    const double kappa = lam[D];
    const double Nab = Iv[0];
    const double* Av = Iv + 1;
    const double* Im = Iv + 1 + D;
    if (t < nI) {
        const int r = t / D, c = t - r * D;
        double acc = 0.0;
        for (int k = 0; k < D; ++k) acc = fma(Pm[r * D + k], Im[k * D + c], acc);
        PI[t] = acc;
    }
    __syncthreads();
    double* o = out + (long)(E + pl) * (D + nI);
    if (t < nI) {
        const int r = t / D, c = t - r * D;
        double acc = 0.0;
        for (int k = 0; k < D; ++k) acc = fma(PI[r * D + k], Pm[c * D + k], acc);   // (P I P^T)[r][c]
        const double pl2 = Pm[r * D + c] * lam[c] + Pm[c * D + r] * lam[r];         // P Lambda + Lambda P^T
        o[D + t] = kappa * (0.5 * acc - 0.25 * Nab * pl2);
    } else if (t < nI + D) {
        const int r = t - nI;
        double acc = 0.0;
        for (int c = 0; c < D; ++c) acc = fma(Pm[r * D + c], Av[c], acc);
        o[r] = kappa * acc;
    }
}

void mm_bwd_geometry(int npad, int PL, int* njs, int* nrb) {
    *nrb = (npad + 64 * BWD_RT - 1) / (64 * BWD_RT);
    int q = 1;   // column splits: enough workgroups for a few balanced rounds of the chip
    while (q < 4 && (npad / 16) % (2 * q) == 0 && (long)*nrb * PL * q < 1536) q *= 2;
    *njs = q;
}

void launch_mm_bwd(hipStream_t st, const MMModel& md, const MMWork& wk, double* rowmom, double* cpart, double* part,
                   const double* bars, double* head, double* out) {
    const int P = wk.PL, E = md.E, D = md.D;
    int njs, nrb;
    mm_bwd_geometry(md.npad, P, &njs, &nrb);
    const int nhead = E + P, per_row = nrb * njs;
    dim3 grid(nrb, P + (nhead + per_row - 1) / per_row, njs);   // the rows past P hold the head workgroups
    const int LD = D | 1, nI = D * D;
    const size_t lds_pair = sizeof(double) * std::max((size_t)4 * (md.npad / njs), (size_t)4 * nI + D);
#define PB(K_) hipLaunchKernelGGL((k_mm_bwd_pair<K_>), grid, dim3(256), lds_pair, st, md, wk, rowmom, cpart, njs, bars, head)
    switch (wk.KP / 4) {
        case 1: PB(1); break;
        case 2: PB(2); break;
        case 3: PB(3); break;
        default: PB(4); break;
    }
#undef PB
    const int nrc = mm_bwd_rc(md.npad);
    double* mpart = part + (size_t)P * nrc * (1 + D + nI);
    const size_t lds_post = sizeof(double) * std::max((size_t)3 * 64 * LD + 128 + 3 * D, (size_t)nI + 64 * LD + 128 + D + 2);
    hipLaunchKernelGGL(k_mm_bwd_post, dim3(P + E, nrc), dim3(256), lds_post, st, md, wk, rowmom, cpart, njs, nrb, part, nrc,
                       head, mpart);
    const size_t lds_fin = sizeof(double) * ((size_t)3 * nI + 4 * D + 8);
    hipLaunchKernelGGL(k_mm_bwd_fin, dim3(P + E), dim3(256), lds_fin, st, md, wk, part, nrc, bars, head, mpart, out);
}
int mm_bwd_rc(int npad) { return std::min(BWD_RC, npad / 64); }

// ------------------------------------------------------------------ pair kernel, plain VALU
// Reference implementation of the same tile sums without matrix cores: one row
// per thread (256-row tile), 64 columns staged in LDS and read by broadcast.
template <int KPT>
__global__ __launch_bounds__(256) void k_mm_pair_valu(MMModel md, MMWork wk) {
    __shared__ double Bs[KPT][64];
    __shared__ double bbs[64];
    __shared__ double red[8];
    const int npad = md.npad;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int ncb = npad / 64;
    const int tj = blockIdx.x % ncb, ti = blockIdx.x / ncb, pl = blockIdx.y;
    int a, b;
    local_pair_ab(wk, md.E, pl, a, b);
    const bool diag = (a == b) && (md.iK != nullptr);
    const int KP = wk.KP;
    const double* At = wk.At + (long)pl * KP * npad;
    const double* Bt = wk.Bt + (long)pl * KP * npad;
    const int i = ti * 256 + t;
    const bool rowok = i < npad;
    double av[KPT];
#pragma unroll
    for (int k = 0; k < KPT; ++k) av[k] = (k < KP && rowok) ? At[(long)k * npad + i] : 0.0;
    const int j0 = tj * 64;
    for (int e = t; e < KPT * 64; e += 256) {
        const int k = e >> 6, j = e & 63;
        Bs[k][j] = (k < KP) ? Bt[(long)k * npad + j0 + j] : 0.0;
    }
    if (t < 64) bbs[t] = md.beta[(long)b * npad + j0 + t];
    __syncthreads();
    double s1 = 0.0, s2 = 0.0;
    if (rowok) {
        const double* iKrow = diag ? md.iK + ((long)a * npad + i) * npad + j0 : nullptr;
        for (int j = 0; j < 64; ++j) {
            double e = 0.0;
#pragma unroll
            for (int k = 0; k < KPT; ++k) e = fma(av[k], Bs[k][j], e);
            const double L = exp(e);
            s1 = fma(bbs[j], L, s1);
            if (diag) s2 = fma(iKrow[j], L, s2);
        }
        s1 *= md.beta[(long)a * npad + i];
    }
    for (int off = 32; off > 0; off >>= 1) {
        s1 += __shfl_down(s1, off);
        s2 += __shfl_down(s2, off);
    }
    if (lane == 0) {
        red[2 * w] = s1;
        red[2 * w + 1] = s2;
    }
    __syncthreads();
    if (t == 0) {
        double* out = wk.pair_part + ((long)pl * wk.NT + ti * ncb + tj) * 2;
        out[0] = ((red[0] + red[2]) + red[4]) + red[6];
        out[1] = ((red[1] + red[3]) + red[5]) + red[7];
    }
}

static int pair_njb(int npad, int PL) {
    const int nb = npad / 64;
    int njb = 1;
    while (njb * 2 <= nb && nb % (njb * 2) == 0 && (long)PL * (npad / (16 * PAIR_RT)) * njb < 1536) njb *= 2;
    const char* env = getenv("PILCO_PAIR_NJB");
    if (env) {
        const int v = atoi(env);
        if (v >= 1 && v <= nb && nb % v == 0) njb = v;
    }
    return njb;
}

int mm_pair_nt(int npad, int variant, int PL) {
    if (variant == 1) return ((npad + 255) / 256) * (npad / 64);
    if (variant == 2) {  // depends on npad only, so the summation order is the same for every rank count
        const int nb = npad / 64;
        int njb = 1;
        while (njb * 2 <= nb && nb % (njb * 2) == 0 && njb < 4) njb *= 2;
        return (npad / (16 * PAIR_RT)) * njb;
    }
    return (npad / (16 * PAIR_RT)) * pair_njb(npad, PL);
}

void mm_pair_sk_steps(int npad, int* tdiag, int* toff) {
    const int NS = npad / 16, NTI = npad / (16 * PAIR_RT);
    *toff = NTI * NS;
    *tdiag = NTI * NS - PAIR_RT * NTI * (NTI - 1) / 2;
}

template <int KC>
static int sk_capacity_of() {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_mm_pair_sk<KC>, 256, 0) != hipSuccess || nb <= 0) nb = 2;
    int dev = 0, cus = 256;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    return nb * cus * 4;
}

int mm_pair_sk_capacity(int KP) {
    const char* env = getenv("PILCO_SK_WAVES");
    if (env && atoi(env) >= 4) return atoi(env) / 4 * 4;
    switch (KP / 4) {
        case 1: return sk_capacity_of<1>();
        case 2: return sk_capacity_of<2>();
        case 3: return sk_capacity_of<3>();
        case 4: return sk_capacity_of<4>();
        case 5: return sk_capacity_of<5>();
        case 6: return sk_capacity_of<6>();
        case 7: return sk_capacity_of<7>();
        case 8: return sk_capacity_of<8>();
        default: return sk_capacity_of<9>();
    }
}

void launch_mm_pair(hipStream_t st, const MMModel& md, const MMWork& wk, int variant) {
    const int KP = wk.KP;
    if (variant == 1) {
        dim3 grid(((md.npad + 255) / 256) * (md.npad / 64), wk.PL);
#define PV(K_) hipLaunchKernelGGL((k_mm_pair_valu<K_>), grid, dim3(256), 0, st, md, wk)
        if (KP <= 4) PV(4);
        else if (KP <= 8) PV(8);
        else if (KP <= 12) PV(12);
        else if (KP <= 16) PV(16);
        else if (KP <= 24) PV(24);
        else PV(36);
#undef PV
        return;
    }
    if (variant == 2) {
        const int NJB = wk.NT / (md.npad / (16 * PAIR_RT));
        dim3 grid((md.npad / (16 * PAIR_RT)) * NJB, wk.PL);
#define PM(K_) hipLaunchKernelGGL((k_mm_pair_tiled<K_>), grid, dim3(256), 0, st, md, wk, NJB)
        switch (KP / 4) {
            case 1: PM(1); break;
            case 2: PM(2); break;
            case 3: PM(3); break;
            case 4: PM(4); break;
            case 5: PM(5); break;
            case 6: PM(6); break;
            case 7: PM(7); break;
            case 8: PM(8); break;
            default: PM(9); break;
        }
#undef PM
        return;
    }
    dim3 grid(wk.sk_waves / 4);
#define PS(K_) hipLaunchKernelGGL((k_mm_pair_sk<K_>), grid, dim3(256), 0, st, md, wk)
    switch (KP / 4) {
        case 1: PS(1); break;
        case 2: PS(2); break;
        case 3: PS(3); break;
        case 4: PS(4); break;
        case 5: PS(5); break;
        case 6: PS(6); break;
        case 7: PS(7); break;
        case 8: PS(8); break;
        default: PS(9); break;
    }
#undef PS
}

// ------------------------------------------------------------------ glue
// Workgroup 0 is the serial link of the step: everything it needs is pulled into LDS
// with one batch of loads, then (pack ->) assemble -> propagate -> controller -> joint.
// Workgroup 1 (rollouts with a reward) evaluates the reward of the PRE-propagation
// state concurrently (pilco.py:133); the state is double-buffered so it never races
// with workgroup 0's update.
struct GlueLds {
    double* mx;   // [nm]     current state mean
    double* sx;   // [nm*nm]  current state covariance
    double* mu;   // [nm]
    double* su;   // [nm*nm]
    double* cxu;  // [nm*nm]
    double* t1;   // [nm*nm]
    double* t2;   // [nm*nm]
    double* s1;   // [nm*nm]  s1 = [s_x, s_x c_xu] of the previous joint
    double* seg;  // [SEG]    this rank's packed results
    double* mp;   // [EL*NCH*(1+D)] mean partials
    double* misc; // [128]
};

static size_t glue_lds_doubles(int E, int D, int SEG, int mp) {
    const int nm = E > D ? E : D;
    const size_t tail = (size_t)SEG + (size_t)mp;
    const size_t rew = reward_lds_doubles(E);
    return (size_t)2 * nm + 6 * (size_t)nm * nm + 128 + (tail > rew ? tail : rew);
}
size_t glue_lds_bytes(int E, int D) { return sizeof(double) * glue_lds_doubles(E, D, 0, 0); }

// global -> LDS copy with all loads of a 1024-element chunk in flight before the first wait
__device__ __forceinline__ void bulk_load(double* dst, const double* __restrict__ src, int n) {
    for (int base = 0; base < n; base += 1024) {
        double v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int e = base + k * 256 + (int)threadIdx.x;
            v[k] = (e < n) ? src[e] : 0.0;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int e = base + k * 256 + (int)threadIdx.x;
            if (e < n) dst[e] = v[k];
        }
    }
}

// squash_sin on (mu[U], su[U][U]) in place; cdiag[u] = e_u exp(-s_uu/2) cos(m_u)   (controllers.py:13-36)
__device__ __forceinline__ void squash_inplace(const GlueLds& L, int U, const double* maxact, double* cdiag) {
    const int t = threadIdx.x;
    for (int e = t; e < U * U; e += blockDim.x) {
        const int u = e / U, v = e - u * U;
        const double du = L.su[u * U + u], dv = L.su[v * U + v];
        const double lq = -(du + dv) / 2.0;
        const double q = exp(lq);
        const double suv = L.su[e];
        const double val = (exp(lq + suv) - q) * cos(L.mu[u] - L.mu[v]) - (exp(lq - suv) - q) * cos(L.mu[u] + L.mu[v]);
        const double eu = maxact ? maxact[u] : 1.0, ev = maxact ? maxact[v] : 1.0;
        L.t2[e] = eu * ev * val / 2.0;
    }
    if (t < U) {
        const double eu = maxact ? maxact[t] : 1.0;
        const double ex = exp(-L.su[t * U + t] / 2.0);
        cdiag[t] = eu * ex * cos(L.mu[t]);
        L.misc[64 + t] = eu * ex * sin(L.mu[t]);
    }
    __syncthreads();
    for (int e = t; e < U * U; e += blockDim.x) L.su[e] = L.t2[e];
    if (t < U) L.mu[t] = L.misc[64 + t];
    __syncthreads();
}

// joint Gaussian of (x,u) from mx,sx,mu,su,cxu in LDS -> in_m, in_s, s1 (pilco.py:141-144)
__device__ __forceinline__ void write_joint(const GlueArgs& g, const GlueLds& L) {
    const int E = g.E, U = g.U, D = g.D, t = threadIdx.x;
    for (int e = t; e < E * U; e += blockDim.x) {  // sc = s_x c_xu  (E,U)
        const int r = e / U, u = e - r * U;
        double acc = 0.0;
        _Pragma("unroll 8") for (int k = 0; k < E; ++k) acc = fma(L.sx[r * E + k], L.cxu[k * U + u], acc);
        L.t1[e] = acc;
    }
    __syncthreads();
    if (t < D) g.wk.in_m[t] = (t < E) ? L.mx[t] : L.mu[t - E];
    for (int e = t; e < D * D; e += blockDim.x) {
        const int r = e / D, c = e - r * D;
        double v;
        if (r < E && c < E) v = L.sx[r * E + c];
        else if (r < E) v = L.t1[r * U + (c - E)];
        else if (c < E) v = L.t1[c * U + (r - E)];
        else v = L.su[(r - E) * U + (c - E)];
        g.wk.in_s[e] = v;
        if (r < E) g.s1[r * D + c] = v;
        if (g.tape) {
            double* rec = g.tape + (long)g.step * (D + D * D + E * D + E + E * E + D * E);
            rec[D + e] = v;
            if (r < E) rec[D + D * D + r * D + c] = v;
        }
    }
    if (g.tape && t < D) g.tape[(long)g.step * (D + D * D + E * D + E + E * E + D * E) + t] = (t < E) ? L.mx[t] : L.mu[t - E];
}

// Reduce the tile / stream-K partials of the local pairs and the row-chunk partials of the
// owned outputs into this rank's segment (LDS copy + global gather buffer).  Four lanes per
// pair sum fixed quarters of the partial list and are combined in a fixed tree.
// Round `base` of mm_pack (4 threads per pair), split in two so that the loads of the first round are ISSUED at the very
// start of the glue kernel, together with its other loads, and consumed after them (vmcnt is in order: one round trip).
// Only kernel arguments go into the addresses (closed-form wave ranges).
struct PackPre {
    double v[16];
    double isdet;
};
// stream-K partials of pair k live in sk_part[k][0 .. sk_maxw) in wave order (unused slots stay zero); lane gq of the
// pair's four lanes takes the quarter [gq * sk_maxw / 4, (gq + 1) * sk_maxw / 4): 16 contiguous doubles = one cache line
// at the usual sizes, addresses known from the thread index alone.
__device__ __forceinline__ void mm_pack_issue(const MMWork& wk, int base, PackPre& pp) {
    const int t = threadIdx.x;
    const int k = base + (t >> 2), gq = t & 3;
    pp.isdet = 0.0;
#pragma unroll
    for (int u = 0; u < 16; ++u) pp.v[u] = 0.0;
    if (k >= wk.PL) return;
    pp.isdet = wk.pair_isdet[k];
    if (wk.sk_waves > 0) {
        const int qw = wk.sk_maxw >> 2;
        const double* src = wk.sk_part + (long)k * wk.sk_maxw + gq * qw;
#pragma unroll
        for (int u = 0; u < 16; ++u)
            if (u < qw) pp.v[u] = src[u];
    }
}
__device__ __forceinline__ void mm_pack_sum(const MMWork& wk, int base, const PackPre& pp, double& s0, double& s1) {
    const int t = threadIdx.x;
    const int k = base + (t >> 2), gq = t & 3;
    s0 = 0.0;
    s1 = 0.0;
    if (k >= wk.PL) return;
    if (wk.sk_waves > 0) {
#pragma unroll
        for (int u = 0; u < 16; ++u) s0 += pp.v[u];
        const int qw = wk.sk_maxw >> 2;
        const double* src = wk.sk_part + (long)k * wk.sk_maxw + gq * qw;
        for (int u = 16; u < qw; ++u) s0 += src[u];   // few pairs spread over many waves
    } else {
        const double* part = wk.pair_part + (long)k * wk.NT * 2;
        const int q0 = (int)((long)wk.NT * gq / 4), q1 = (int)((long)wk.NT * (gq + 1) / 4);
        for (int q = q0; q < q1; ++q) {
            s0 += part[2 * q];
            s1 += part[2 * q + 1];
        }
    }
}

__device__ __forceinline__ void mm_pack(const MMWork& wk, int D, int E, const GlueLds& L, PackPre& pp) {
    const int t = threadIdx.x;
    double* seg = wk.gath + (long)wk.rank * wk.SEG;
    for (int base = 0; base < wk.PL; base += 64) {
        const int k = base + (t >> 2), gq = t & 3;
        if (base > 0) mm_pack_issue(wk, base, pp);
        double s0, s1;
        mm_pack_sum(wk, base, pp, s0, s1);
        s0 += __shfl_xor(s0, 1);
        s1 += __shfl_xor(s1, 1);
        s0 += __shfl_xor(s0, 2);
        s1 += __shfl_xor(s1, 2);
        if (k < wk.PL && gq == 0) {
            int a, b;
            local_pair_ab(wk, E, k, a, b);
            const double v = ((a == b) ? (s0 - s1) : s0) * pp.isdet;   // mgpr.py:144-145
            seg[k] = v;
            L.seg[k] = v;
        }
    }
    const int W1 = 1 + D;
    for (int e = t; e < wk.EL * W1; e += blockDim.x) {   // M_a and V_a: sums of the chunk contributions
        const int o = e / W1, idx = e - o * W1;
        double sum = 0.0;
        _Pragma("unroll 8") for (int ch = 0; ch < wk.NCHM; ++ch) sum += L.mp[(o * wk.NCHM + ch) * W1 + idx];
        seg[wk.OUTOFF + e] = sum;
        L.seg[wk.OUTOFF + e] = sum;
    }
    __syncthreads();
}

// packed results -> out_M [E], out_S [E][E], out_V [D][E]; also left in LDS (oM, oS, oV)
__device__ __forceinline__ void mm_assemble(const MMWork& wk, const double* src, const double* var, int D, int E, double* oM,
                            double* oS, double* oV) {
    const int t = threadIdx.x;
    for (int a = t; a < E; a += blockDim.x) {
        const double v = src[(a % wk.nranks) * wk.SEG + wk.OUTOFF + (a / wk.nranks) * (1 + D)];
        oM[a] = v;
        wk.out_M[a] = v;
    }
    for (int e = t; e < D * E; e += blockDim.x) {
        const int d = e / E, a = e - d * E;
        const double v = src[(a % wk.nranks) * wk.SEG + wk.OUTOFF + (a / wk.nranks) * (1 + D) + 1 + d];
        oV[e] = v;
        wk.out_V[e] = v;
    }
    __syncthreads();
    for (int e = t; e < E * E; e += blockDim.x) {
        const int a = e / E, b = e - a * E;
        const int hi = a > b ? a : b, lo = a > b ? b : a;
        const int kk = pair_order_index(E, hi, lo);
        double v = src[(kk % wk.nranks) * wk.SEG + kk / wk.nranks];
        if (a == b) v += var[a];                                   // mgpr.py:146
        v = fma(-oM[a], oM[b], v);                                 // mgpr.py:147
        oS[e] = v;
        wk.out_S[e] = v;
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void k_glue(GlueArgs g) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int E = g.E, D = g.D, U = g.U, t = threadIdx.x;
    const int nm = E > D ? E : D;
    int mp_n = (g.flags & GF_PACK) ? g.wk.EL * g.wk.NCHM * (1 + D) : 0;
    int seg_n = (g.flags & (GF_PACK | GF_ASSEMBLE)) ? g.wk.SEG * ((g.flags & GF_PACK) ? 1 : g.wk.nranks) : 0;
    if (g.flags & GF_RBF_POST) {  // this launch reduces the POLICY GP (inputs = state, outputs = controls)
        mp_n = g.pwk.EL * g.pwk.NCHM * (1 + E);
        seg_n = g.pwk.SEG;
    }
    GlueLds L;
    L.mx = sm;
    L.sx = L.mx + nm;
    L.mu = L.sx + nm * nm;
    L.su = L.mu + nm;
    L.cxu = L.su + nm * nm;
    L.t1 = L.cxu + nm * nm;
    L.t2 = L.t1 + nm * nm;
    L.s1 = L.t2 + nm * nm;
    L.misc = L.s1 + nm * nm;
    L.seg = L.misc + 128;
    L.mp = L.seg + seg_n;

    const bool dbg0 = (t == 0);
    const int dbo = (g.step == 0) ? 16 : 0;  // the initial glue of a rollout stamps slots 24..29
    if (blockIdx.x == 1) {
        // Workgroup 1: reward of the current (pre-propagation) state (rewards.py:19-81), evaluated
        // concurrently with workgroup 0; the state is double-buffered so there is no race.
        DBG_STAMP(g.wk, 20, dbg0);
        double* ws = L.seg;  // scratch: this workgroup uses none of the pack / assemble storage
        if (t < E) L.mx[t] = g.m_x[t];
        bulk_load(L.sx, g.s_x, E * E);
        __syncthreads();
        double mu, var;
        reward_eval(g.n_rewards, g.rw, E, L.mx, L.sx, ws, g.rew_out != nullptr, mu, var);
        if (t == 0) {
            if (g.rew_out) {
                g.rew_out[0] = mu;   // pilco_reward_eval: mean and variance
                g.rew_out[1] = var;
            } else {
                g.reward[0] += mu;   // rollout (pilco.py:133): single writer, stream ordered
            }
        }
        DBG_STAMP(g.wk, 21, dbg0);
        return;
    }

    DBG_STAMP(g.wk, 8 + dbo, dbg0);
    PackPre pp;   // first round of the pack: its loads are in flight together with the batch below
    if ((g.flags & GF_PACK) && !(g.wk.abl & 16)) mm_pack_issue(g.wk, 0, pp);
    // one batch of loads for everything the serial part reads
    if (g.flags & (GF_PROPAGATE | GF_TRAJ | GF_POLICY | GF_RBF_PRE)) {
        if (t < E) L.mx[t] = g.m_x[t];
        bulk_load(L.sx, g.s_x, E * E);
    }
    if (g.flags & GF_PROPAGATE) bulk_load(L.s1, g.s1, E * D);
    if (g.flags & GF_PACK) bulk_load(L.mp, g.wk.mean_part, mp_n);
    if (g.flags & GF_RBF_POST) bulk_load(L.mp, g.pwk.mean_part, mp_n);
    if ((g.flags & GF_ASSEMBLE) && !(g.flags & GF_PACK)) bulk_load(L.seg, g.wk.gath, seg_n);
    __syncthreads();

    DBG_STAMP(g.wk, 9 + dbo, dbg0);
    if ((g.flags & GF_PACK) && !(g.wk.abl & 16)) mm_pack(g.wk, D, E, L, pp);
    DBG_STAMP(g.wk, 10 + dbo, dbg0);
    if (g.flags & GF_ASSEMBLE) {
        // single rank: the LDS copy of the segment is the whole gather buffer
        mm_assemble(g.wk, L.seg, g.var, D, E, L.mu, L.su, L.cxu);  // oM -> mu, oS -> su, oV -> cxu
        if (g.tape && g.step >= 1) {
            double* rec = g.tape + (long)(g.step - 1) * (D + D * D + E * D + E + E * E + D * E) + D + D * D + E * D;
            if (t < E) rec[t] = L.mu[t];
            for (int e = t; e < E * E; e += blockDim.x) rec[E + e] = L.su[e];
            for (int e = t; e < D * E; e += blockDim.x) rec[E + E * E + e] = L.cxu[e];
        }
    }
    DBG_STAMP(g.wk, 11 + dbo, dbg0);
    if (g.flags & GF_PROPAGATE) {
        // t1 = s1 V (E,E); state += increment                      (pilco.py:147-149)
        for (int e = t; e < E * E; e += blockDim.x) {
            const int r = e / E, c = e - r * E;
            double acc = 0.0;
            _Pragma("unroll 8") for (int k = 0; k < D; ++k) acc = fma(L.s1[r * D + k], L.cxu[k * E + c], acc);
            L.t1[e] = acc;
        }
        __syncthreads();
        for (int e = t; e < E * E; e += blockDim.x) {
            const int r = e / E, c = e - r * E;
            const double v = ((L.su[e] + L.sx[e]) + L.t1[e]) + L.t1[c * E + r];
            L.t2[e] = v;
            g.s_out[e] = v;
        }
        if (t < E) {
            const double v = L.mu[t] + L.mx[t];
            L.misc[96 + t] = v;
            g.m_out[t] = v;
        }
        __syncthreads();
        for (int e = t; e < E * E; e += blockDim.x) L.sx[e] = L.t2[e];
        if (t < E) L.mx[t] = L.misc[96 + t];
        __syncthreads();
    }
    DBG_STAMP(g.wk, 12 + dbo, dbg0);
    if ((g.flags & GF_TRAJ) && g.traj) {
        double* dst = g.traj + (long)g.step * (E + E * E);
        if (t < E) dst[t] = L.mx[t];
        for (int e = t; e < E * E; e += blockDim.x) dst[E + e] = L.sx[e];
    }
    if (g.flags & GF_RBF_PRE) {  // RbfController: the state is the input of the policy GP (controllers.py:115-116)
        if (t < E) g.pwk.in_m[t] = L.mx[t];
        for (int e = t; e < E * E; e += blockDim.x) g.pwk.in_s[e] = L.sx[e];
    }
    if (g.flags & GF_POLICY) {
        if (g.pol_kind == PILCO_POLICY_RBF) {
            // mean-function-only GP: iK = 0, then S -= diag(var - 1e-6)      (controllers.py:116-117)
            PackPre pq;
            mm_pack_issue(g.pwk, 0, pq);
            mm_pack(g.pwk, E, U, L, pq);
            mm_assemble(g.pwk, L.seg, g.pvar, E, U, L.mu, L.su, L.cxu);   // M (U), S (U,U), V (E,U)
            if (t < U) L.su[t * U + t] -= g.pvar[t] - 1e-6;
            __syncthreads();
            if (g.squash) {
                double* cdiag = L.misc + 1;
                squash_inplace(L, U, g.maxact, cdiag);
                for (int e = t; e < E * U; e += blockDim.x) L.cxu[e] *= cdiag[e % U];
                __syncthreads();
            }
        }
        if (g.pol_kind == PILCO_POLICY_LINEAR) {
            // M = m W^T + b, S = W s W^T, V = W^T                  (controllers.py:52-54)
            bulk_load(L.t2, g.W, U * E);
            __syncthreads();
            if (t < U) {
                double acc = g.b[t];
                _Pragma("unroll 8") for (int k = 0; k < E; ++k) acc = fma(L.t2[t * E + k], L.mx[k], acc);
                L.mu[t] = acc;
            }
            for (int e = t; e < U * E; e += blockDim.x) {
                const int u = e / E, c = e - u * E;
                double acc = 0.0;
                _Pragma("unroll 8") for (int k = 0; k < E; ++k) acc = fma(L.t2[u * E + k], L.sx[k * E + c], acc);
                L.t1[e] = acc;  // W s
                L.cxu[c * U + u] = L.t2[e];
            }
            __syncthreads();
            for (int e = t; e < U * U; e += blockDim.x) {
                const int u = e / U, v = e - u * U;
                double acc = 0.0;
                _Pragma("unroll 8") for (int k = 0; k < E; ++k) acc = fma(L.t1[u * E + k], L.t2[v * E + k], acc);
                L.su[e] = acc;
            }
            __syncthreads();
            if (g.squash) {
                double* cdiag = L.misc + 1;  // [U]
                squash_inplace(L, U, g.maxact, cdiag);
                for (int e = t; e < E * U; e += blockDim.x) L.cxu[e] *= cdiag[e % U];   // V @ C, C diagonal
                __syncthreads();
            }
        }
        if (g.act_out) {
            if (t < U) g.act_out[t] = L.mu[t];
            for (int e = t; e < U * U; e += blockDim.x) g.act_out[U + e] = L.su[e];
            for (int e = t; e < E * U; e += blockDim.x) g.act_out[U + U * U + e] = L.cxu[e];
        } else {
            write_joint(g, L);
        }
    }
    DBG_STAMP(g.wk, 13 + dbo, dbg0);
}

__global__ void k_stamp(unsigned long long* dbg, int slot) {
    if (threadIdx.x == 0) dbg[slot] = wall_clock64();
}
void launch_stamp(hipStream_t st, unsigned long long* dbg, int slot) {
    hipLaunchKernelGGL(k_stamp, dim3(1), dim3(64), 0, st, dbg, slot);
}

void launch_glue(hipStream_t st, const GlueArgs& g, bool with_reward_block) {
    int mp_n = (g.flags & GF_PACK) ? g.wk.EL * g.wk.NCHM * (1 + g.D) : 0;
    int seg_n = (g.flags & (GF_PACK | GF_ASSEMBLE)) ? g.wk.SEG * ((g.flags & GF_PACK) ? 1 : g.wk.nranks) : 0;
    if (g.flags & GF_RBF_POST) {
        mp_n = g.pwk.EL * g.pwk.NCHM * (1 + g.E);
        seg_n = g.pwk.SEG;
    }
    const size_t lds = sizeof(double) * glue_lds_doubles(g.E, g.D, seg_n, mp_n);
    static size_t configured = 0;
    if (lds > configured) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_glue), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)lds);
        configured = lds;
    }
    hipLaunchKernelGGL(k_glue, dim3(with_reward_block ? 2 : 1), dim3(256), lds, st, g);
}

// ------------------------------------------------------------------ self test
// D = A B with A[i][k] = i + 1 + 100 k (16x4), B[k][j] = (k == 0) ? j + 1 : 0 so that
// D[i][j] = (i + 1)(j + 1): exposes both the operand and the result lane maps.
__global__ void k_selftest_mfma(double* out) {
    const int lane = threadIdx.x;
    const int i = lane & 15, k = lane >> 4;
    const double a = (double)(i + 1 + 100 * k);
    const double b = (k == 0) ? (double)((lane & 15) + 1) : 0.0;
    d4 acc = {0.0, 0.0, 0.0, 0.0};
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[lane * 4 + r] = acc[r];
}

// max relative deviation of the table-driven exp from the library exp over [-720, 8]
__global__ void k_selftest_fexp(const double* tab_g, double* out) {
    __shared__ double tab[FEXP_TN];
    for (int e = threadIdx.x; e < FEXP_TN; e += blockDim.x) tab[e] = tab_g[e];
    __syncthreads();
    double worst = 0.0;
    for (int i = threadIdx.x; i < 200000; i += blockDim.x) {
        const double x = -720.0 + 728.0 * ((double)i + 0.37) / 200000.0;
        const double ref = exp(fmax(x, -700.0));
        const double got = fexp(x, tab);
        // allowed: 1 ulp of the result + the |x| eps conditioning of the single-constant reduction
        const double rel = fabs(got - ref) / ref / (2.3e-16 + 1.2e-16 * fabs(x));
        worst = fmax(worst, rel);
    }
    for (int off = 32; off > 0; off >>= 1) worst = fmax(worst, __shfl_down(worst, off));
    if ((threadIdx.x & 63) == 0) out[threadIdx.x >> 6] = worst;
}

int mm_exp_table_size() { return FEXP_TN; }

int launch_selftest_mfma(hipStream_t st, double* dbuf, double* hbuf, const double* exp_tab) {
    hipLaunchKernelGGL(k_selftest_fexp, dim3(1), dim3(256), 0, st, exp_tab, dbuf);
    if (hipMemcpyAsync(hbuf, dbuf, 4 * sizeof(double), hipMemcpyDeviceToHost, st) != hipSuccess) return -1;
    if (hipStreamSynchronize(st) != hipSuccess) return -1;
    for (int w = 0; w < 4; ++w)
        if (!(hbuf[w] < 1.0)) return 100000;
    hipLaunchKernelGGL(k_selftest_mfma, dim3(1), dim3(64), 0, st, dbuf);
    if (hipMemcpyAsync(hbuf, dbuf, 256 * sizeof(double), hipMemcpyDeviceToHost, st) != hipSuccess) return -1;
    if (hipStreamSynchronize(st) != hipSuccess) return -1;
    for (int lane = 0; lane < 64; ++lane)
        for (int r = 0; r < 4; ++r) {
            const int row = (lane >> 4) + 4 * r, col = lane & 15;
            if (hbuf[lane * 4 + r] != (double)((row + 1) * (col + 1))) return 1 + lane * 4 + r;
        }
    return 0;
}

}  // namespace pilco
