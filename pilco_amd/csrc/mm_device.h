// Device-side helpers shared by the moment-matching translation units (prep.hip, pair.hip, bwd.hip, glue.hip):
// pair indexing, write-through stores, DPP reductions, the table-driven fp64 exp, register / LDS Gauss-Jordan,
// buffer loads and the reward evaluation.  Internal; gfx950 only.
#pragma once
#include "moment.h"

#include <type_traits>

namespace pilco {

typedef double d4 __attribute__((ext_vector_type(4)));

// row of output a in the (gathered) beta buffer / block of output a in this rank's iK (see MMModel)
// (one rank: the identity -- without two signed divisions in every prologue that asks)
__device__ __forceinline__ long mm_beta_row(const MMModel& md, int a) { return md.bW == 1 ? (long)a : (long)((a % md.bW) * md.bEL + a / md.bW); }
__device__ __forceinline__ long mm_ik_blk(const MMModel& md, int a) { return md.bW == 1 ? (long)a : (long)(a / md.bW); }

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

// ---- addressing of the exponent GEMM's operands (layout: MMWork::At / Wt in moment.h)
// Where pair (local index pl, outputs a >= b) finds them in global memory.
struct PairOps {
    const double* At;    // the pair's row block [KP][npad]
    const double* Wt;    // base of the column-operand allocation (MMWork::Wt): one buffer resource for every pair
    unsigned w0;         // BYTES from Wt to the pair's column block [KP][npad]
    unsigned v0;         // BYTES from Wt to the pair's v_j [npad]   (v0 > w0: vcol sits behind the blocks)
    int D;
};
// column block of a pair: its column output b -- or, in the fused head, the pair itself where a workgroup must read back what
// IT wrote within one launch (one-launch small step with the operands in memory: wk.fuse_pair != 0).  Kernels that run
// behind a head launch pass blk = b.
__device__ __forceinline__ int pair_col_block(const MMWork& wk, int pl, int b) { return wk.fuse_pair ? pl : b; }
__device__ __forceinline__ PairOps pair_ops(const MMWork& wk, int D, int npad, int pl, int blk) {
    PairOps po;
    po.At = wk.At + (long)pl * wk.KP * npad;
    po.Wt = wk.Wt;
    po.w0 = (unsigned)blk * (unsigned)wk.KP * (unsigned)npad * 8u;
    po.v0 = (unsigned)((wk.vcol - wk.Wt) + (long)pl * npad) * 8u;
    po.D = D;
    return po;
}
// row k of the column-side operand B = (w_j | 1 | v_j | 0..) of a pair, as a pointer (scalar code paths)
template <bool VSEP>
__device__ __forceinline__ const double* colop_row(const PairOps& po, int npad, int k) {
    return (const double*)((const char*)po.Wt + ((!VSEP && k == po.D + 1) ? po.v0 : po.w0 + (unsigned)k * (unsigned)npad * 8u));
}
// Who writes the w rows.  The E D rows of all column blocks (row b D + d = coordinate d of output b's block: w = zeta_d /
// l_bd^2, ONE multiplication from the centred point every operand workgroup has staged) are dealt over the local pairs,
// wt_rows_per_pair rows each: every pair workgroup stores the same number of rows.  (First version: the block of output b
// by the diagonal pair (b, b) -- 40 of 220 workgroups stored twice as much as the others and ended 1.7 us behind them.)
// Not dealt -- the first local pair with column b writes the whole block, pair_writes_wt -- when a pair would get more
// rows than the workgroup has slots for their constants (few pairs on many ranks), when a chunk is longer than the stage, and in the one-launch small step with
// its operands in memory, where every pair owns a block.
// (both decided on the host when the workspace is built -- MMWork::wt_R, wt_deal --: rows per pair = ceil(E D / PL); dealt when
// that fits the workgroup's DT slots and every row of a chunk is in its stage, npad / NCH <= 256)
__device__ __forceinline__ int wt_rows_per_pair(const MMWork& wk) { return wk.wt_R; }
__device__ __forceinline__ bool wt_rows_dealt(const MMWork& wk) { return !wk.fuse_pair && wk.wt_deal; }
// Does local pair pl = (a, b) write the column block of output b this step?  The FIRST local pair with that column does:
// (b, b) when this rank owns it (diagonal pairs come first in the dealing order), else the off-diagonal (a', b) with the
// smallest a' it owns.  (One-launch small step with operands in memory: every pair writes its own block.)
__device__ __forceinline__ bool pair_writes_wt(const MMWork& wk, int E, int a, int b) {
    if (wk.fuse_pair) return true;
    if (wk.nranks == 1 || b % wk.nranks == wk.rank) return a == b;
    for (int a2 = b + 1; a2 < a; ++a2)
        if ((E + a2 * (a2 - 1) / 2 + b) % wk.nranks == wk.rank) return false;
    return true;
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

// stamps INSIDE the per-point operand code (tools/head_phases.py): a developer build only (-DHEAD_ROW_STAMPS) -- the test
// for the stamp buffer in that code costs the one-launch small step 0.3 us (config 4: 1422 -> 1398 rollouts/s)
#ifdef HEAD_ROW_STAMPS
#define DBG_STAMP_ROW(wk_, slot_, cond_) DBG_STAMP(wk_, slot_, cond_)
#else
#define DBG_STAMP_ROW(wk_, slot_, cond_) do { } while (0)
#endif

// hipcc (ROCm 7.2) lets the destination registers of v_mfma_f64_16x16x4_f64 overlap its A / B source registers when
// the accumulator input is the inline constant 0 and a source dies at that instruction; the hardware then reads the
// source while it is already being overwritten (found in round 2: one exponent register of a reverse-sweep
// instantiation came out wrong).  Keeping the operands of such a first MFMA alive past it (an empty asm that "uses"
// them) rules the overlap out at no instruction cost; tools/mfma_overlap_check.py scans the generated code
// (tests/test_build_isa.py).
#define MFMA_KEEP_ALIVE(x_) asm volatile("" ::"v"(x_))
// ... tied to the MFMA's result, for places where the scheduler moves the plain form AHEAD of the MFMA (seen in the unrolled
// chains of the one-launch small step): the empty asm reads and "writes" the result, so it cannot come before the MFMA,
// and it uses both sources, so they stay alive until then.
#define MFMA_PIN(res_, a_, b_) asm volatile("" : "+v"(res_) : "v"(a_), "v"(b_))
// The same compiler does not always insert the wait states between a v_mfma_f64_16x16x4_f64 and a VALU read of its result
// (it does in the forward pair kernel: 7 + destination-pair-index slots; in the wide reverse-sweep instantiations a
// v_max read the LAST destination pair in the very next slot and got the accumulator from before the last k-step).
// MFMA_RESULT_FENCE(e): every reader of e comes after an s_nop that covers the longest of those distances.
// tools/mfma_hazard_check.py scans the generated code for such reads (tests/test_build_isa.py).
#define MFMA_RESULT_FENCE(e_) asm volatile("s_nop 10" : "+v"(e_))

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
#define FEXP_C ((double)FEXP_TN * 1.4426950408889634074)    /* T / ln2 (exact scaling of the rounded 1/ln2) */
#define FEXP_LN2_64 (0.69314718055994530942 / (double)FEXP_TN)   /* ln2 / T */
#define FEXP_DEG (FEXP_TB <= 6 ? 5 : FEXP_TB <= 10 ? 4 : 3)        /* polynomial degree: r^(deg+1)/(deg+1)! < 2^-54 */
#define FEXP_MAGIC 6755399441055744.0      /* 1.5 * 2^52 */

#if !defined(PILCO_DEV) && defined(PAIR_OPT)
#error "PAIR_OPT is a developer experiment: build with -DPILCO_DEV (tools/ only)"
#endif
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
    double q;
    if (FEXP_DEG == 5) {
        q = fma(r, 1.0 / 120.0, 1.0 / 24.0);
        q = fma(r, q, 1.0 / 6.0);
        q = fma(r, q, 0.5);
    } else if (FEXP_DEG == 4) {
        q = fma(r, 1.0 / 24.0, 1.0 / 6.0);
        q = fma(r, q, 0.5);
    } else {
        q = fma(r, 1.0 / 6.0, 0.5);
    }
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
// ... for a result that goes STRAIGHT into an MFMA operand.  gfx950 does not interlock "VALU writes a VGPR -> an MFMA reads
// it as SrcA / SrcB": the MFMA must come at least three issue slots behind the write (measured: tools/ubench_srcc_war.hip --
// next slot and one s_nop 0 read the OLD register, s_nop 1 is enough).  hipcc leaves those wait states behind its own VALU
// instructions but NOT behind an inline-asm statement: with the exponent insertion written as asm (fexp_finish) the moment
// product of the reverse sweep's off-diagonal pairs read a weight from before its exponent went in whenever the scheduler put
// the MFMA within two slots -- round 5's "wrong, run-to-run different sums behind a division" (the division only moved the
// schedule), and every instantiation's turn sooner or later.  Here the insertion is C: the same single v_lshl_add_u32, an
// instruction the compiler's hazard recognizer sees.  tools/mfma_hazard_check.py flags the pattern.
__device__ __forceinline__ double fexp_to_mfma(double x, const double* __restrict__ tab) {
    x = fexp_clamp(x);
    const double t = fexp_t(x);
    const double tv = tab[__double2loint(t) & (FEXP_TN - 1)];
    const double pm1 = fexp_poly(x, t);
    const double res = fma(tv, pm1, tv);
    const int lo = __double2loint(t) & ~(FEXP_TN - 1);
    const int hi = __double2hiint(res) + (lo << (20 - FEXP_TB));   // exponent += n >> FEXP_TB
    return __hiloint2double(hi, __double2loint(res));
}

// Pivoted Gauss-Jordan on an n x nc augmented matrix held in LDS (row-major,
// ld = nc), ping-ponging between two buffers: one barrier per pivot step.
// Called by the whole workgroup.  Returns the buffer holding [I | A^{-1} B];
// det = det(A) (valid in every thread).  General (slow) path.
__device__ inline double* gauss_jordan(double* G0, double* G1, int n, int nc, double& det) {
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

// 1/sqrt(x) to fp64 accuracy: hardware estimate + two Newton steps (x > 0); the library's sqrt and division are ~80 dependent
// operations of a lone wave on the head's serial path
__device__ __forceinline__ double fast_rsqrt(double x) {
    double r = __builtin_amdgcn_rsq(x);
    r = r * fma(-0.5 * x * r, r, 1.5);
    r = r * fma(-0.5 * x * r, r, 1.5);
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

// Quotient and remainder of a small non-negative index (e < 2^22) by a positive divisor: float reciprocal and a one-step fix-up,
// a third of the instructions of the compiler's signed 32-bit division (~30 dependent VALU operations, 0.1 us of a lone wave
// each) -- the link indexes its small matrices by flat thread indices in every phase, dozens of divisions on the step's serial path.
__device__ __forceinline__ int idiv_s(int e, int d, int& rem) {
    int q = (int)((float)e * __builtin_amdgcn_rcpf((float)d));   // (v_rcp_f32: 1 ulp, the fix-up below covers it)
    int r = e - q * d;
    q += (r >= d ? 1 : 0) - (r < 0 ? 1 : 0);
    rem = e - q * d;
    return q;
}

// ------------------------------------------------------------------ rewards
// Unpivoted Gauss-Jordan for symmetric positive definite systems on an n x nc augmented
// matrix in LDS (ping-pong buffers, one barrier per pivot, one element per thread when
// n*nc <= blockDim).  Returns the buffer holding [I | A^{-1} B]; det in every thread.
__device__ inline double* gauss_jordan_spd(double* G0, double* G1, int n, int nc, double& det) {
    double* cur = G0;
    double* nxt = G1;
    det = 1.0;
    // (the reward workgroup is the last of the head launch to finish: the element's row / column come from a cheap index
    // division made once, the pivot row is scaled by a reciprocal -- estimate + two Newton steps -- instead of ~30 dependent
    // operations of an IEEE division per element and pivot)
    int r0, c0;
    r0 = idiv_s((int)threadIdx.x, nc, c0);
    for (int k = 0; k < n; ++k) {
        __syncthreads();
        const double piv = cur[k * nc + k];
        det *= piv;
        const double rp = fast_rcp(piv);
        for (int e = threadIdx.x; e < n * nc; e += blockDim.x) {
            int r = r0, c = c0;
            if (e != (int)threadIdx.x) r = idiv_s(e, nc, c);
            const double pk = cur[k * nc + c] * rp;
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
__device__ inline double exp_reward_moment(const RewardDev& rw, int E, double scale, const double* mx, const double* sx,
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
            int k;
            const int e = idiv_s(e2, r, k);
            double acc = 0.0;
            _Pragma("unroll 8") for (int f = 0; f < E; ++f) acc = fma(sx[e * E + f], Fl[f * r + k], acc);
            SF[e2] = acc;
        }
        __syncthreads();
        for (int e2 = t; e2 < r * r; e2 += blockDim.x) {
            int l;
            const int k = idiv_s(e2, r, l);
            double acc = 0.0;
            _Pragma("unroll 8") for (int e = 0; e < E; ++e) acc = fma(Fl[e * r + k], SF[e * r + l], acc);
            A[e2] = fma(scale, acc, (k == l) ? 1.0 : 0.0);
        }
        __syncthreads();
        // [A | y] -> A^{-1} y; A is SPD so no pivoting is needed.  The augmented matrix is built in G1 and the elimination
        // ping-pongs from there into SF (dead since the barrier above): no copy and no barrier of its own for that
        double* G0 = SF;             // [E*(E+1)] fits E*E + ... (r <= E: r (r + 1) <= E*E + E: SF and the head of A)
        double* G1 = A + E * E;      // [E*(E+1)]
        const int nc = r + 1;
        for (int e2 = t; e2 < r * nc; e2 += blockDim.x) {
            int l;
            const int k = idiv_s(e2, nc, l);
            G1[e2] = (l < r) ? A[k * r + l] : y[k];
        }
        double det;
        const double* res = gauss_jordan_spd(G1, G0, r, nc, det);
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

// Buffer loads for the hot loop: wave-uniform resource (base pointer) + 32-bit per-lane byte offset + scalar byte
// offset, i.e. no 64-bit address arithmetic in the VALU stream (the fp64 pipe is the bottleneck of this kernel).
typedef unsigned v2u __attribute__((ext_vector_type(2)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t buf_rsrc(const double* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(base), 0, 0x7fffffff, 0x00020000);   // raw, untyped
}
// ... with the base pointer pinned to scalar registers (a resource the compiler has moved to vector registers costs a
// waterfall loop -- readfirstlane, compare, masked load -- per load)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t buf_rsrc_uniform(const double* base) {
    const unsigned long long p = (unsigned long long)(uintptr_t)base;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)p), hi = __builtin_amdgcn_readfirstlane((unsigned)(p >> 32));
    return buf_rsrc((const double*)(uintptr_t)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ double buf_ld(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    const v2u v = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
    return __hiloint2double((int)v.y, (int)v.x);
}

}  // namespace pilco
