// Device code of the O(N^2) pair sums shared by the pair kernels (pair.hip) and the one-launch small step of the fused head
// (prep_device.h): the per-wave tile loop (pair_wave) and the closed forms of the stream-K work split.  Internal; gfx950 only.
#pragma once
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
#ifndef PAIR_FAIR
#define PAIR_FAIR 1    // the stream-K waves lower their issue priority as they advance (WaveProgress); 0: A/B builds
#endif
#ifndef PAIR_MINW
#define PAIR_MINW 1    // __launch_bounds__ min waves per SIMD for the pair kernel
#endif
// FENCE: an s_nop between the MFMA chain of a tile and the first VALU read of its result (MFMA_RESULT_FENCE, mm_device.h).
// The pair kernels do not need it (the compiler keeps 7 + p slots there); inside the fused head, under its register
// budget, the same source has been seen scheduled with reads of destination pairs 2 and 3 one slot early
// (tools/mfma_hazard_check.py, tests/test_build_isa.py), so that host asks for the fence.
// Operands in global memory (po: the layout of MMWork::At / Wt): every row of the column operand is reached through ONE buffer
// resource (base po.Wt) and a per-lane row offset computed once per call -- the w rows of the pair's column block, the ones,
// v_j, the zeros --, so the hot loop is what it was when every pair had its own copy of all KP rows.
// LDSOP (the one-launch step of small models, prep_device.h): the operands never went to memory -- At / Bt / vcol point into
// the workgroup's LDS, At as [k][lda] over the workgroup's own rows (row i0 is its row i0l), Bt as [k][ldb] over all columns
// (all KP rows materialised there); po is not used.
// The exp table of a pair kernel's workgroup, on its way from memory while the first tile's operands are requested: the
// kernel REQUESTS its entry (one per thread) at its very top and hands it to the first pair_wave call of each wave, which
// stores it and meets the other waves at the workgroup's barrier only after its own operand requests have left (the table is
// first read by the first exp, a whole MFMA chain later).  Loading table -> barrier -> operands in sequence was two memory
// round trips in a row at the start of every launch.
struct TabArrival {
    static constexpr int NV = (FEXP_TN + 255) / 256;   // entries per thread of a 256-thread workgroup
    double v[NV];    // this thread's table entries (requested, perhaps not yet arrived)
    double* lds;     // the table in LDS
    bool pending;    // not stored yet (wave-uniform)
    __device__ __forceinline__ void request(const double* __restrict__ g, double* l) {
#pragma unroll
        for (int k = 0; k < NV; ++k) v[k] = (FEXP_TN % 256 == 0 || threadIdx.x + 256 * k < FEXP_TN) ? g[threadIdx.x + 256 * k] : 0.0;
        lds = l;
        pending = true;
    }
    __device__ __forceinline__ void land() {
        if (pending) {
#pragma unroll
            for (int k = 0; k < NV; ++k)
                if (FEXP_TN % 256 == 0 || threadIdx.x + 256 * k < FEXP_TN) lds[threadIdx.x + 256 * k] = v[k];
            __syncthreads();
            pending = false;
        }
    }
};
// Progress of a stream-K wave through its share of the line, turned into its issue priority.  The SIMD's arbiter serves the
// OLDEST wave first: of the three waves a SIMD hosts the first finished after 24 us, the second after 38, the third after 50
// (tools/pair_waves.py) -- and a lone wave cannot keep the fp64 pipe busy (dependent MFMA -> exp -> accumulate chains), so the
// last quarter of every launch ran at two thirds of the rate.  A wave that is ahead now yields: priority 3 for the first
// quarter of its steps down to 0 for the last, so the waves of a SIMD stay within a quarter of each other and the pipe has
// work from all of them until the end (headline 409 -> 417 rollouts/s in one A/B call).  Finer slices -- the four
// priorities cycled 8, 16, 32 times per share -- measured WORSE (410, 409, 399): waves in lock-step want the matrix cores
// and the VALU at the same moments.
#ifndef PAIR_FAIR_LEVELS
#define PAIR_FAIR_LEVELS 4     // priority levels used (2..4): level = L - 1 - floor(L done / total)
#endif
struct WaveProgress {
    int done, total, level;
    __device__ __forceinline__ void tick(int steps) {
        done += steps;
        const int lv = (PAIR_FAIR_LEVELS - 1) - min(PAIR_FAIR_LEVELS - 1, (PAIR_FAIR_LEVELS * done) / max(total, 1));
        if (lv != level) {   // (wave-uniform)
            level = lv;
            if (lv == 3) __builtin_amdgcn_s_setprio(3);
            else if (lv == 2) __builtin_amdgcn_s_setprio(2);
            else if (lv == 1) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(0);
        }
    }
};
template <int KC, bool DIAG, bool VSEP, bool FENCE = false, bool LDSOP = false>
__device__ __forceinline__ double pair_wave(const PairOps& po, const double* __restrict__ At, const double* __restrict__ Bt,
                                            const double* __restrict__ vcol,
                                            const double* __restrict__ beta_a, const double* __restrict__ beta_b,
                                            const double* __restrict__ iKa, const double* __restrict__ tab, int npad, int i0,
                                            int jbeg, int jend, int lane, int lda = 0, int ldb = 0, int i0l = 0, TabArrival* ta = nullptr, WaveProgress* wp = nullptr) {
    constexpr int NE = 4 * PAIR_RT;  // exponent values per lane per 16-column step
    static_assert(PAIR_PF == 2, "the column loop is unrolled over a two-slot operand ring");
    const int lr = lane >> 4, lc = lane & 15;
    double af[PAIR_RT][KC];
#pragma unroll
    for (int rt = 0; rt < PAIR_RT; ++rt)
#pragma unroll
        for (int c = 0; c < KC; ++c) {
            if (LDSOP) af[rt][c] = At[(4 * c + lr) * lda + i0l + 16 * rt + lc];
            else af[rt][c] = po.At[(long)(4 * c + lr) * npad + i0 + 16 * rt + lc];
        }
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
    for (int c = 0; c < KC; ++c) {
        b_off[c] = ((unsigned)(4 * c + lr) * (unsigned)npad + (unsigned)lc) * 8u;   // relative to the pair's column block (po.w0 rides in the scalar offset)
        if (!VSEP && !LDSOP && 4 * c + lr == po.D + 1) b_off[c] = po.v0 - po.w0 + (unsigned)lc * 8u;   // the lanes of row D + 1: the pair's own v_j
    }
    const unsigned bb_off = (unsigned)lc * 8u;
    const unsigned sB = LDSOP ? 0u : po.w0, sV = LDSOP ? 0u : po.v0;   // scalar byte offsets of the column block / of v_j (VSEP) in the resource
    const __amdgpu_buffer_rsrc_t rB = buf_rsrc(LDSOP ? beta_b : po.Wt), rbeta = buf_rsrc(beta_b);   // (LDSOP: rB is never used)
    const __amdgpu_buffer_rsrc_t rIK = buf_rsrc(DIAG ? iKa + (long)i0 * npad : beta_b);
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
            for (int c = 0; c < KC; ++c) {
                if (LDSOP) ring[p][c] = Bt[(4 * c + lr) * ldb + jbeg + 16 * p + lc];
                else ring[p][c] = buf_ld(rB, b_off[c], sB + (unsigned)(jbeg + 16 * p) * 8u);
            }
            ring[p][KC] = buf_ld(rbeta, bb_off, (unsigned)(jbeg + 16 * p) * 8u);
            if (VSEP) {
                if (LDSOP) ring[p][KC + 1] = vcol[jbeg + 16 * p + lc];
                else ring[p][KC + 1] = buf_ld(rB, bb_off, sV + (unsigned)(jbeg + 16 * p) * 8u);
            }
        }
    }
    if (ta) ta->land();   // (the requests above are in flight; see TabArrival)
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
            if (FENCE) MFMA_RESULT_FENCE(e);
#pragma unroll
            for (int r = 0; r < 4; ++r) x[4 * rt + r] = VSEP ? e[r] + vv : e[r];   // C/D column = lane & 15: one v_j per lane
        }
        if (PAIR_ABL != 4 && j0 + 32 < jend) {
#pragma unroll
            for (int c = 0; c < KC; ++c) {
                if (LDSOP) rg[c] = Bt[(4 * c + lr) * ldb + j0 + 32 + lc];
                else rg[c] = buf_ld(rB, b_off[c], sB + (unsigned)(j0 + 32) * 8u);
            }
            rg[KC] = buf_ld(rbeta, bb_off, (unsigned)(j0 + 32) * 8u);
            if (VSEP) {
                if (LDSOP) rg[KC + 1] = vcol[j0 + 32 + lc];
                else rg[KC + 1] = buf_ld(rB, bb_off, sV + (unsigned)(j0 + 32) * 8u);
            }
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
        if (wp) wp->tick(2);
        step(ring[0], j0);
        if (j0 + 16 < jend) step(ring[1], j0 + 16);
    }
    if (!DIAG) {
#pragma unroll
        for (int i = 0; i < NE; ++i) total = fma(bi[i], acc[i], total);
    }
    return total;
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
// One wave's share of the stream-K line (see k_mm_pair_sk): the 16-column steps [boundary(w), boundary(w + 1)) of the
// cost line, touching at most two local pairs p0, p1 (-1: none) with the sums out0, out1 (before the wave reduction).
template <int KC, bool VSEP, bool FENCE = false>
__device__ __forceinline__ void sk_wave_range(const MMModel& md, const MMWork& wk, const double* __restrict__ tab, int w, int lane,
                                              double& out0, double& out1, int& p0, int& p1, TabArrival* ta = nullptr, bool fair = false) {
    const int npad = md.npad, NS = npad / 16;
    const int nd_steps = wk.sk_nd * wk.sk_tdiag;
    int step = sk_boundary(wk, w);
    const int end = sk_boundary(wk, w + 1);
    WaveProgress prog{0, end - step, -1};
    WaveProgress* wp = fair ? &prog : nullptr;
    out0 = 0.0;
    out1 = 0.0;
    p0 = -1;
    p1 = -1;
    double cur = 0.0;
    int cur_pl = -1;
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
        const PairOps po = pair_ops(wk, md.D, npad, pl, b);
        const double* beta_a = md.beta + mm_beta_row(md, a) * npad;
        const double* beta_b = md.beta + mm_beta_row(md, b) * npad;
        const int i0 = ti * 16 * PAIR_RT, jbeg = sidx * 16, jend = jbeg + seg * 16;
        if (dg)
            cur += pair_wave<KC, true, VSEP, FENCE>(po, nullptr, nullptr, nullptr, beta_a, beta_b, md.iK + mm_ik_blk(md, a) * npad * npad, tab, npad, i0, jbeg, jend, lane, 0, 0, 0, ta, wp);
        else
            cur += pair_wave<KC, false, VSEP, FENCE>(po, nullptr, nullptr, nullptr, beta_a, beta_b, nullptr, tab, npad, i0, jbeg, jend, lane, 0, 0, 0, ta, wp);
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
}

}  // namespace pilco
