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
#include "mm_device.h"

namespace pilco {

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
#if !defined(PILCO_DEV) && defined(PAIR_ABL)
#error "PAIR_ABL is a developer experiment: build with -DPILCO_DEV (tools/ only)"
#endif
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
template <int KC, bool DIAG, bool VSEP>
__device__ __forceinline__ double pair_wave(const double* __restrict__ At, const double* __restrict__ Bt,
                                            const double* __restrict__ vcol,
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
    const __amdgpu_buffer_rsrc_t rV = buf_rsrc(VSEP ? vcol : Bt);
    const __amdgpu_buffer_rsrc_t rIK = buf_rsrc(DIAG ? iKa + (long)i0 * npad : Bt);
    if (DIAG && jbeg < i0) jbeg = i0;  // columns left of the diagonal block are mirrored by the transposed tile
    double total = 0.0;
    // software pipeline: the operands of the column step two ahead are requested while this one is
    // evaluated (a first touch of Bt / beta misses the XCD's L2: ~2 us, more than one step)
    double ring[2][KC + 2];   // column operand fragments, beta_b,j and (VSEP) v_j
#pragma unroll
    for (int p = 0; p < 2; ++p) {
#pragma unroll
        for (int c = 0; c <= KC + 1; ++c) ring[p][c] = 0.0;
        if (jbeg + 16 * p < jend) {
#pragma unroll
            for (int c = 0; c < KC; ++c) ring[p][c] = buf_ld(rB, b_off[c], (unsigned)(jbeg + 16 * p) * 8u);
            ring[p][KC] = buf_ld(rbeta, bb_off, (unsigned)(jbeg + 16 * p) * 8u);
            if (VSEP) ring[p][KC + 1] = buf_ld(rV, bb_off, (unsigned)(jbeg + 16 * p) * 8u);
        }
    }
    // one 16-column step on ring slot rg; the slot is refilled with the operands of column step j0 + 32
    auto step = [&](double (&rg)[KC + 2], const int j0) {
        double bf[KC];
#pragma unroll
        for (int c = 0; c < KC; ++c) bf[c] = rg[c];
        const double bb = rg[KC];
        const double vv = rg[KC + 1];
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
            MFMA_KEEP_ALIVE(af[rt][0]);   // (first MFMA of the chain: constant-zero accumulator, see mm_device.h)
            MFMA_KEEP_ALIVE(bf[0]);
#pragma unroll
            for (int r = 0; r < 4; ++r) x[4 * rt + r] = VSEP ? e[r] + vv : e[r];   // C/D column = lane & 15: one v_j per lane
        }
        if (PAIR_ABL != 4 && j0 + 32 < jend) {
#pragma unroll
            for (int c = 0; c < KC; ++c) rg[c] = buf_ld(rB, b_off[c], (unsigned)(j0 + 32) * 8u);
            rg[KC] = buf_ld(rbeta, bb_off, (unsigned)(j0 + 32) * 8u);
            if (VSEP) rg[KC + 1] = buf_ld(rV, bb_off, (unsigned)(j0 + 32) * 8u);
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
        if (FEXP_DEG == 5) {
#pragma unroll
            for (int i = 0; i < NE; ++i) pm[i] = fma(rr[i], 1.0 / 120.0, 1.0 / 24.0);
#pragma unroll
            for (int i = 0; i < NE; ++i) pm[i] = fma(rr[i], pm[i], 1.0 / 6.0);
#pragma unroll
            for (int i = 0; i < NE; ++i) pm[i] = fma(rr[i], pm[i], 0.5);
        } else if (FEXP_DEG == 4) {
#pragma unroll
            for (int i = 0; i < NE; ++i) pm[i] = fma(rr[i], 1.0 / 24.0, 1.0 / 6.0);
#pragma unroll
            for (int i = 0; i < NE; ++i) pm[i] = fma(rr[i], pm[i], 0.5);
        } else {
#pragma unroll
            for (int i = 0; i < NE; ++i) pm[i] = fma(rr[i], 1.0 / 6.0, 0.5);
        }
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

template <int KC, bool VSEP>
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
    const double* beta_a = md.beta + mm_beta_row(md, a) * npad;
    const double* beta_b = md.beta + mm_beta_row(md, b) * npad;
    const int i0 = ti * 16 * PAIR_RT;
    const int JB = npad / NJB, JW = JB / 4;
    const int jbeg = jb * JB + w * JW;
    double t1;
    if (diag)
        t1 = pair_wave<KC, true, VSEP>(At, Bt, wk.vcol + (long)pl * npad, beta_a, beta_b, md.iK + mm_ik_blk(md, a) * npad * npad, tab, npad, i0, jbeg, jbeg + JW, lane);
    else
        t1 = pair_wave<KC, false, VSEP>(At, Bt, wk.vcol + (long)pl * npad, beta_a, beta_b, nullptr, tab, npad, i0, jbeg, jbeg + JW, lane);
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
void mm_sk_pair_waves(int k, int waves, int nd, int tdiag, int toff, int total, int ud, int uo, int* wlo, int* fslot, int* whi) {
    sk_pair_waves(k, waves, nd, tdiag, toff, total, ud, uo, *wlo, *fslot, *whi);
}

template <int KC, bool VSEP>
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
    if (wk.dbg && w == 0 && lane == 0) wk.dbg[32] = clock64();     // shader-clock counter beside the 100 MHz wall clock: tools/ derive the engine clock under this load
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
        const double* beta_a = md.beta + mm_beta_row(md, a) * npad;
        const double* beta_b = md.beta + mm_beta_row(md, b) * npad;
        const int i0 = ti * 16 * PAIR_RT, jbeg = sidx * 16, jend = jbeg + seg * 16;
        if (dg)
            cur += pair_wave<KC, true, VSEP>(At, Bt, wk.vcol + (long)pl * npad, beta_a, beta_b, md.iK + mm_ik_blk(md, a) * npad * npad, tab, npad, i0, jbeg, jend, lane);
        else
            cur += pair_wave<KC, false, VSEP>(At, Bt, wk.vcol + (long)pl * npad, beta_a, beta_b, nullptr, tab, npad, i0, jbeg, jend, lane);
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
        // slot-major layout sk_part[slot][pair], slot = wave - (first wave of the pair): the reader (the serial link)
        // gets coalesced loads with addresses from its thread index alone.  A wave that enters a pair from a previous one
        // IS that pair's first wave (slot 0); only the first touched pair needs the closed form.
        if (p0 >= 0) {
            const long S0 = (p0 < wk.sk_nd) ? (long)p0 * wk.sk_tdiag : (long)wk.sk_nd * wk.sk_tdiag + (long)(p0 - wk.sk_nd) * wk.sk_toff;
            const int wlo = sk_wave_of(S0, wk.sk_waves, nd_steps, wk.sk_total, wk.sk_ud, wk.sk_uo);
            wk.sk_part[(long)(w - wlo) * wk.sk_pls + p0] = out0;
        }
        if (p1 >= 0) wk.sk_part[p1] = out1;
    }
    DBG_STAMP(wk, 17, w == 0 && lane == 0);
    if (wk.dbg && w == 0 && lane == 0) wk.dbg[33] = clock64();
    DBG_STAMP(wk, 18, w == wk.sk_waves - 1 && lane == 0);
    if (wk.dbg && lane == 0 && (w & 7) == 0) wk.dbg[1024 + (w >> 3)] = wall_clock64();  // end stamp of every 8th wave
}


// ------------------------------------------------------------------ pair kernel, plain VALU
// Reference implementation of the same tile sums without matrix cores: one row
// per thread (256-row tile), 64 columns staged in LDS and read by broadcast.
template <int KPT>
__global__ __launch_bounds__(256) void k_mm_pair_valu(MMModel md, MMWork wk) {
    __shared__ double Bs[KPT][64];
    __shared__ double bbs[64];
    __shared__ double vs[64];
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
    if (t < 64) {
        bbs[t] = md.beta[mm_beta_row(md, b) * npad + j0 + t];
        vs[t] = wk.vsep ? wk.vcol[(long)pl * npad + j0 + t] : 0.0;
    }
    __syncthreads();
    double s1 = 0.0, s2 = 0.0;
    if (rowok) {
        const double* iKrow = diag ? md.iK + (mm_ik_blk(md, a) * npad + i) * npad + j0 : nullptr;
        for (int j = 0; j < 64; ++j) {
            double e = vs[j];
#pragma unroll
            for (int k = 0; k < KPT; ++k) e = fma(av[k], Bs[k][j], e);
            const double L = exp(e);
            s1 = fma(bbs[j], L, s1);
            if (diag) s2 = fma(iKrow[j], L, s2);
        }
        s1 *= md.beta[mm_beta_row(md, a) * npad + i];
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

template <int KC, bool VSEP>
static int sk_capacity_of() {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_mm_pair_sk<KC, VSEP>, 256, 0) != hipSuccess || nb <= 0) nb = 2;
    int dev = 0, cus = 256;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    return nb * cus * 4;
}

int mm_pair_sk_capacity(int KP, bool vsep) {
    const char* env = getenv("PILCO_SK_WAVES");
    if (env && atoi(env) >= 4) return atoi(env) / 4 * 4;
    switch (KP / 4) {
        case 1: return vsep ? sk_capacity_of<1, true>() : sk_capacity_of<1, false>();
        case 2: return vsep ? sk_capacity_of<2, true>() : sk_capacity_of<2, false>();
        case 3: return vsep ? sk_capacity_of<3, true>() : sk_capacity_of<3, false>();
        case 4: return vsep ? sk_capacity_of<4, true>() : sk_capacity_of<4, false>();
        case 5: return vsep ? sk_capacity_of<5, true>() : sk_capacity_of<5, false>();
        case 6: return vsep ? sk_capacity_of<6, true>() : sk_capacity_of<6, false>();
        case 7: return vsep ? sk_capacity_of<7, true>() : sk_capacity_of<7, false>();
        case 8: return vsep ? sk_capacity_of<8, true>() : sk_capacity_of<8, false>();
        default: return vsep ? sk_capacity_of<9, true>() : sk_capacity_of<9, false>();
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
#define PM(K_)                                                                                        \
    do {                                                                                              \
        if (wk.vsep) hipLaunchKernelGGL((k_mm_pair_tiled<K_, true>), grid, dim3(256), 0, st, md, wk, NJB);  \
        else hipLaunchKernelGGL((k_mm_pair_tiled<K_, false>), grid, dim3(256), 0, st, md, wk, NJB);         \
    } while (0)
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
#define PS(K_)                                                                                 \
    do {                                                                                       \
        if (wk.vsep) hipLaunchKernelGGL((k_mm_pair_sk<K_, true>), grid, dim3(256), 0, st, md, wk);   \
        else hipLaunchKernelGGL((k_mm_pair_sk<K_, false>), grid, dim3(256), 0, st, md, wk);          \
    } while (0)
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
    MFMA_KEEP_ALIVE(a);
    MFMA_KEEP_ALIVE(b);
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
