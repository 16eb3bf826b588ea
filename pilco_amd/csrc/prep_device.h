// Device code of the per-step operand work (operands of the pair kernel, mean parts, reward) shared by k_mm_prep
// (prep.hip).  Internal; gfx950 only.
#pragma once
#include "glue_device.h"
#include "pair_device.h"

namespace pilco {

constexpr int PREP_TAB_DOUBLES = FEXP_TN + 8;   // LDS tail of the operand kernel: exp table + wave sums of the one-launch small step
__host__ __device__ constexpr size_t prep_region_doubles(int DT) {   // the operand work's own LDS region (pair / mean workgroups), before that tail
    // (the point stage of a pair workgroup, 256 (DT + 1), or -- one-launch small step with its operands in LDS, KP <= 16 --
    // the operands of 64 rows and 256 columns and v: KP (64 + 256) + 256)
    // (value-and-gradient form of that step, D <= 14: + z of the rows 16 x 64, beta_b 256, column-sum slices 8 x 64)
    const size_t stage = 256 * (size_t)(DT + 1),
                 ops = mm_kp(DT) <= 16 ? (size_t)mm_kp(DT) * (64 + 256) + 256 + (DT <= 14 ? 16 * 64 + 256 + 8 * 64 : 0) : 0;
    const size_t pair_blk = (size_t)4 * DT + 2 * (size_t)DT * DT + 4 + (stage > ops ? stage : ops);
    const size_t mean_blk = (size_t)2 * DT + 2 * (size_t)DT * DT + 4 + 9 * (size_t)(DT + 1) + 2 * (size_t)DT + 512 * (size_t)(DT + 2);
    return pair_blk > mean_blk ? pair_blk : mean_blk;
}

// 1 / l^2 of the w rows dealt to local pair pl (mm_device.h: wt_rows_per_pair), into the pair workgroup's `colbuf` [DT]
template <int DT>
__device__ __forceinline__ void prep_wt_constants(const MMModel& md, const MMWork& wk, int pl, double* sm) {
    const int R = wt_rows_per_pair(wk), t = threadIdx.x;
    if (wt_rows_dealt(wk) && t < R && pl * R + t < md.E * md.D) {
        const double l = md.ls[pl * R + t];
        sm[3 * DT + 2 * DT * DT + 4 + t] = 1.0 / (l * l);
    }
}

// 512 threads per workgroup = the whole register file of one CU.  The 256 rows of the chunk are
// handled twice in parallel: threads 0..255 ("group 0") build the row-side operand, threads
// 256..511 ("group 1") the column-side operand; on a diagonal pair both operands are the same
// vectors, so group 0 writes both and group 1 does the mean / input-output covariance sums.
// Mean / input-output-covariance sums of local output al over row chunk chm (mgpr.py:99-118), by one spare workgroup
// of the prep launch: T = Lambda^-1 B^-1 Lambda^-1 = (s + Lambda^2)^-1 by a register Gauss-Jordan in wave 0 while the
// other threads already have their point in flight; lb_i = exp(-zeta_i^T T zeta_i / 2) beta_i; partial c g and c T h
// into mean_part[al][chm][1 + D].
template <int DT, bool FUSED, int NTHR, bool PRE = false>
__device__ __forceinline__ void prep_mean_block(const MMModel& md, const MMWork& wk, int al, int chm, double* sm,
                                                const double* jm, const double* js,   // joint Gaussian in LDS (FUSED head only)
                                                const double la_t, const double var_a) {  // l_a[t] (t < D) and var_a, loaded by the caller
    const int D = md.D, npad = md.npad;
    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);   // wave-uniform: the role branches below are scalar branches
    const bool act = (NTHR == 512) || t < 512;   // a host workgroup wider than 512 threads: the extra waves only keep the barriers
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
    constexpr int LDZ = DT | 1;   // odd row stride: conflict-free LDS rows
    if constexpr (!PRE) {   // (PRE: k_mm_prep wrote the constants before the link, the link stored the joint Gaussian here)
        if (t < DT) {
            s_m[t] = (t < D) ? (FUSED ? jm[t] : wk.in_m[t]) : 0.0;
            s_ia[t] = (t < D) ? 1.0 / la_t : 0.0;
        }
        for (int e = t; e < D * D; e += 512) s_s[e] = FUSED ? js[e] : wk.in_s[e];
        for (int e = t; e < DT * DT; e += 512) s_T[e] = 0.0;
        __syncthreads();
    }
    const bool dbgm = (t == 0 && al == 0 && chm == 0);
    DBG_STAMP(wk, 40, dbgm);
    const int rpc = npad >> __builtin_ctz(wk.NCHM);
    const int i_begin = chm * rpc, i_end = i_begin + rpc;
    if (w == 0) {
        // [B | I],  B = Lambda^-1 s Lambda^-1 + I; T = Lambda^-1 B^-1 Lambda^-1   (mgpr.py:103-111)
        double col[DT];   // (branch-free, as the pair workgroups' build: all LDS reads in flight together)
        const int c = lane;
        const bool in_s = c < D;
        const double ic = in_s ? s_ia[c] : 0.0;
        double sv[DT], ir[DT];
#pragma unroll
        for (int r = 0; r < DT; ++r) {
            sv[r] = s_s[(in_s && r < D) ? r * D + c : 0];
            ir[r] = s_ia[r];   // (zero for r >= D)
        }
#pragma unroll
        for (int r = 0; r < DT; ++r) {
            const double sval = (in_s && r < D) ? sv[r] : 0.0;
            col[r] = c < DT ? fma(sval, ir[r] * ic, (r == c) ? 1.0 : 0.0) : ((c < 2 * DT && c - DT == r) ? 1.0 : 0.0);
        }
        __builtin_amdgcn_s_setprio(3);   // (the wave the other seven wait for)
        const double detB = gj_wave<DT>(col, colbuf, lane);
        __builtin_amdgcn_s_setprio(0);
        if (c >= DT && c < DT + D) {
            const int cc = c - DT;
#pragma unroll
            for (int r = 0; r < DT; ++r)
                if (r < D) s_T[r * DT + cc] = col[r] * s_ia[r] * s_ia[cc];
        }
        if (lane == 0) s_sc[1] = var_a * fast_rsqrt(detB);
    } else if (act) {
        // the other seven waves stage the centred points of the chunk's first 512 rows meanwhile
        // (all of a thread's requests first, then the stores: element after element the loop was twelve L2 round trips in a
        // row -- 3.3-3.7 us against the 2.6 us of the Gauss-Jordan it is meant to hide behind)
        const int idx = (w - 1) * 64 + lane;   // 0..447
        constexpr int NB = (512 * DT + 447) / 448;
        double sv[NB], bv[2];
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            const int e = idx + 448 * k, d = e >> 9, i = i_begin + (e & 511);
            sv[k] = (e < 512 * D && i < md.n && i < i_end) ? md.Pt[(long)d * npad + i] : 0.0;
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int r = idx + 448 * k;
            bv[k] = (r < 512 && i_begin + r < i_end) ? md.beta[mm_beta_row(md, a) * npad + i_begin + r] : 0.0;
        }
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            const int e = idx + 448 * k, d = e >> 9, r = e & 511, i = i_begin + r;
            if (e < 512 * D) zst[r * LDZ + d] = (i < md.n && i < i_end) ? sv[k] - s_m[d] : 0.0;
        }
#pragma unroll
        for (int k = 0; k < 2; ++k)
            if (idx + 448 * k < 512) bst[idx + 448 * k] = bv[k];
    }
    __syncthreads();
    DBG_STAMP(wk, 41, dbgm);
    double g = 0.0;
    double h[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d) h[d] = 0.0;
    // lb_i = exp(-zeta^T T zeta / 2) beta_i   (mgpr.py:113).  Rows are taken from the LDS stage only; a chunk longer
    // than 512 rows is staged in rounds (no pointer ever selects between LDS and global memory).
    for (int r0 = 0; r0 < rpc; r0 += 512) {
        if (r0 > 0) {
            __syncthreads();
            for (int e = t; act && e < 512 * D; e += 512) {
                const int d = e >> 9, r = e & 511;
                const int i = i_begin + r0 + r;
                zst[r * LDZ + d] = (i < md.n && i < i_end) ? md.Pt[(long)d * npad + i] - s_m[d] : 0.0;
            }
            if (act) bst[t] = (i_begin + r0 + t < i_end) ? md.beta[mm_beta_row(md, a) * npad + i_begin + r0 + t] : 0.0;
            __syncthreads();
        }
        if (act && i_begin + r0 + t < i_end) {
            double zeta[DT];
#pragma unroll
            for (int d = 0; d < DT; ++d) zeta[d] = (d < D) ? zst[t * LDZ + d] : 0.0;
            // q / 2 = zeta^T T zeta / 2 from the upper triangle of the symmetric T: sum_c zeta_c (T_cc zeta_c / 2 + sum_{r > c} T_cr zeta_r)
            // -- half the LDS reads (every thread reads the whole matrix by broadcast: the phase is bound by LDS return bandwidth)
            double hq = 0.0;
#pragma unroll
            for (int c = 0; c < DT; ++c) {
                double inner = 0.5 * s_T[c * DT + c] * zeta[c];
#pragma unroll
                for (int r = c + 1; r < DT; ++r) inner = fma(s_T[c * DT + r], zeta[r], inner);
                hq = fma(zeta[c], inner, hq);
            }
            const double lb = exp(-hq) * bst[t];
            g += lb;
#pragma unroll
            for (int d = 0; d < DT; ++d) h[d] = fma(zeta[d], lb, h[d]);
        }
    }
    DBG_STAMP(wk, 42, dbgm);
    // Block sums of g and h in a fixed order: every thread parks its 1 + D partial sums in the (now dead) stage, quantity-major
    // [1 + D][512]; wave w then owns quantities w, w + 8, ..: each of its lanes adds the eight waves' values of its lane position
    // (wave order), one DPP tree over the lanes finishes the sum.  (The first version ran 1 + D DPP trees in EVERY wave -- 200
    // VALU operations per wave on a phase that is pure latency -- and added the eight wave sums in a stage of its own.)
    double* hs = red + 8 * (DT + 1);  // [DT + 1]
    __syncthreads();                  // every thread has taken its point from the stage
    if (act) {
        zst[t] = g;
#pragma unroll
        for (int d = 0; d < DT; ++d)
            if (d < D) zst[(1 + d) * 512 + t] = h[d];
    }
    __syncthreads();
    for (int k = w; act && k < 1 + D; k += 8) {   // (wave-uniform)
        double acc = zst[k * 512 + lane];
#pragma unroll
        for (int j = 1; j < 8; ++j) acc += zst[k * 512 + 64 * j + lane];
        acc = wave_sum_lane63(acc);
        if (lane == 63) hs[k] = acc;
    }
    // then the M and V contributions of this row chunk: c g (mgpr.py:117) and c T h (mgpr.py:118, V = c tiL^T lb = c T sum_i zeta_i lb_i)
    DBG_STAMP(wk, 43, dbgm);
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
        store_wt(&wk.mean_part[((long)al * wk.NCHM + chm) * (1 + D) + t], v);   // indexed by the LOCAL output number
    }
    DBG_STAMP(wk, 44, dbgm);
}

// Value-and-gradient form of the one-launch small step (DESIGN.md sections 4.4, 9): the reverse sweep of bwd.hip for ONE
// workgroup = (local pair, 64 rows) x ALL columns, operands in LDS (Al [KP][64], Bl [KP][256], vl), eight waves = two 32-row
// groups x four column quarters.  Per 16 x 16 tile as k_mm_bwd_pair: exponent tile transposed (column operand as MFMA A),
// W.L as the A operand of the moment product with [w_j | 1] (rows along the result registers), column sums by DPP row sums
// into per-wave LDS slices.  Then the row side's epilogue G (header of bwd.hip) and -- the workgroup has the sums of ITS
// rows for every column -- the column side Gc = sum_j c_j [w_j | 1] [w_j | 1]^T right here (linear in c_j: the chunks of a
// pair add up), both on the matrix cores.  Out: two 16 x 16 blocks per workgroup (G | Gc) and N_ab's share for the link.
// Diagonal pairs: tiles at / right of the diagonal only (weight 2 right of it).
template <int DT>
__device__ __forceinline__ void small_sweep(const MMModel& md, const MMWork& wk, int pl, int item, int i_begin, int c_begin, int ncols, const double* Al, const double* Bl,
                                            const double* vl, const double* Zl, double* bbl, double* csl, double* tail, bool act) {
    constexpr int KC = mm_kp(DT) / 4;
    constexpr bool VSEP = mm_vsep(DT);
    const int D = md.D, npad = md.npad, t = threadIdx.x, lane = t & 63, lr = lane >> 4, lc = lane & 15;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const double* tab = tail;          // exp table (k_mm_prep loaded it at its head)
    int a, b;
    local_pair_ab(wk, md.E, pl, a, b);
    const bool diag = (a == b);
    const double* beta_a = md.beta + mm_beta_row(md, a) * npad;
    const double* beta_b = md.beta + mm_beta_row(md, b) * npad;
    const double* iKa = (diag && md.iK) ? md.iK + mm_ik_blk(md, a) * npad * npad : nullptr;
    if (act)
        for (int e = t; e < 256; e += 512) bbl[e] = e < npad ? beta_b[e] : 0.0;
    for (int e = t; e < 8 * 64; e += 512) csl[e] = 0.0;   // (a diagonal pair's waves skip the tiles left of the diagonal)
    __syncthreads();   // operands, z, beta_b in LDS
    const int rg = w >> 2, cq = w & 3;                       // row group (32 rows), column quarter
    const int i0l = 32 * rg, i0 = i_begin + i0l;
    const int cpw = ncols / 4, jb = c_begin + cpw * cq, je = jb + cpw;   // (ncols: 64, 128 or 256 -- whole 16-column tiles per wave)
    double rf[2][KC], brow[2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        brow[rt] = beta_a[i0 + 16 * rt + lc];
#pragma unroll
        for (int c = 0; c < KC; ++c) rf[rt][c] = Al[(4 * c + lr) * 64 + i0l + 16 * rt + lc];
    }
    const __amdgpu_buffer_rsrc_t rIK = buf_rsrc_uniform(iKa ? iKa : beta_a);
    unsigned ik_off[2][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) ik_off[rt][r] = ((unsigned)(lr + 4 * r) * (unsigned)npad + (unsigned)(i0 + 16 * rt + lc)) * 8u;
    const int dsel = lc <= D ? lc : D;
    d4 acc[2] = {d4{0.0, 0.0, 0.0, 0.0}, d4{0.0, 0.0, 0.0, 0.0}};
    double* myslice = csl + w * 64;
    auto sweep = [&](auto mode_c) {
        constexpr int MODE = decltype(mode_c)::value;   // 0 off-diagonal, 1 diagonal with iK, 2 diagonal without (RBF policy GP)
        // a wave has at most four column steps: ALL of its iK tiles are requested up front (one memory round trip for the
        // workgroup instead of one per step on this latency chain; 64 of the 256 registers a wave may use here)
        double ikv[4][2][4];
        if (MODE == 1) {
#pragma unroll
            for (int sidx = 0; sidx < 4; ++sidx) {
                const int j0 = jb + 16 * sidx;
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        ikv[sidx][rt][r] = (j0 < je && j0 >= i0) ? buf_ld(rIK, ik_off[rt][r], (unsigned)j0 * (unsigned)npad * 8u) : 0.0;
            }
        }
#pragma unroll
        for (int sidx = 0; sidx < 4; ++sidx) {
            const int j0 = jb + 16 * sidx;
            if (j0 >= je) break;
            if (MODE != 0 && j0 < i0) continue;   // (wave-uniform) both row tiles lie below this column tile's mirror
            double cf[KC], a2[4], bcol[4], vj[4];
#pragma unroll
            for (int c = 0; c < KC; ++c) cf[c] = Bl[(4 * c + lr) * 256 + j0 + lc];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                bcol[r] = bbl[j0 + lr + 4 * r];
                vj[r] = VSEP ? vl[j0 + lr + 4 * r] : 0.0;
                a2[r] = Bl[dsel * 256 + j0 + 4 * r + lr];
            }
            double csum[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                const int it = i0 + 16 * rt;
                const double om = j0 > it ? 2.0 : (j0 == it ? 1.0 : 0.0);
                d4 e = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int c = 0; c < KC; ++c) {
                    e = __builtin_amdgcn_mfma_f64_16x16x4f64(cf[c], rf[rt][c], e, 0, 0, 0);   // e[r]: i = lc, j = j0 + lr + 4 r
                    if (c == 0) MFMA_PIN(e, cf[0], rf[rt][0]);
                }
                MFMA_RESULT_FENCE(e);
                double wl[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double l = fexp(VSEP ? e[r] + vj[r] : e[r], tab);
                    if (MODE == 0) {
                        wl[r] = bcol[r] * l;
                        csum[r] = fma(brow[rt], l, csum[r]);
                    } else {
                        double wgt = brow[rt] * bcol[r];
                        if (MODE == 1) wgt -= ikv[sidx][rt][r];
                        wl[r] = (wgt * om) * l;
                        csum[r] += wl[r];
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    acc[rt] = __builtin_amdgcn_mfma_f64_16x16x4f64(wl[r], a2[r], acc[rt], 0, 0, 0);
                    MFMA_PIN(acc[rt], wl[r], a2[r]);   // (the step loop is unrolled: the first step's accumulator is the constant zero)
                }
            }
            // column sums over the 16 lanes of a DPP row (this path is latency-bound, not issue-bound: plain row shifts)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                double v = csum[r];
                v = dpp_add<0x111, 0xf>(v);
                v = dpp_add<0x112, 0xf>(v);
                v = dpp_add<0x114, 0xf>(v);
                v = dpp_add<0x118, 0xf>(v);
                if (lc == 15) myslice[j0 - jb + lr + 4 * r] = v;
            }
        }
    };
    if (act) {
        if (!diag) sweep(std::integral_constant<int, 0>{});
        else if (iKa) sweep(std::integral_constant<int, 1>{});
        else sweep(std::integral_constant<int, 2>{});
    }
    // ---- row side: G[d][e] = sum_i beta~_i [z_i | 1]_d [m_i + r_i z_i / 2 | r_i]_e over the wave's 32 rows
    d4 G = {0.0, 0.0, 0.0, 0.0}, Gc = {0.0, 0.0, 0.0, 0.0};
    const int rlane = (lane & 48) | (D & 15);
    if (act) {
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int il = i0l + 16 * rt + 4 * r + lr, i = i_begin + il;
                const bool valid = i < md.n;
                const double zt = (valid && lc <= D) ? Zl[lc * 64 + il] : 0.0;   // (row D of Zl: ones)
                const double bs = !valid ? 0.0 : (diag ? 1.0 : beta_a[i]);
                const double ri = __shfl(acc[rt][r], rlane);
                const double x = acc[rt][r];
                const double mp = bs == 0.0 ? 0.0 : (lc < D ? fma(0.5 * ri, zt, x) : x);
                const double za = zt * bs;
                G = __builtin_amdgcn_mfma_f64_16x16x4f64(za, mp, G, 0, 0, 0);
                MFMA_PIN(G, za, mp);
            }
    }
    __syncthreads();   // every wave's column-sum slice is complete
    // ---- column side: the workgroup's 256 columns over the eight waves (32 each); c_j = the two row groups' sums
    if (act) {
        const int cw8 = ncols / 8;   // columns per wave here: 8, 16 or 32
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (4 * k >= cw8) break;
            const int jl = cw8 * w + 4 * k + lr, j = c_begin + jl;   // column of this lane in K-step k
            const int q = jl / cpw;                                  // its column quarter (the slices are per quarter)
            double cj = 0.0, wt = 0.0;
            if (j < md.n) {
                const int jj = jl - cpw * q;
                cj = (csl[q * 64 + jj] + csl[(4 + q) * 64 + jj]) * (diag ? 1.0 : bbl[j]);
                wt = lc <= D ? Bl[lc * 256 + j] : 0.0;          // [w_j | 1] (row D of Bl: ones)
            }
            const double ca = cj * wt;
            Gc = __builtin_amdgcn_mfma_f64_16x16x4f64(ca, wt, Gc, 0, 0, 0);
            MFMA_PIN(Gc, ca, wt);
        }
    }
    __syncthreads();   // Bl is free: the eight waves' blocks are added there in wave order
    double* red = const_cast<double*>(Bl);   // [8][256]
    double* gout = wk.sw_gpart + ((long)pl * wk.NT + item) * 512;
    const int tN = ((D & 15) >> 2) * 64 + (D & 3) * 16 + (D & 15);   // entry (D, D) of a block
    for (int blk = 0; blk < 2; ++blk) {
        if (blk) __syncthreads();
        d4 v4 = blk ? Gc : G;
        MFMA_RESULT_FENCE(v4);
        if (act) {
#pragma unroll
            for (int r = 0; r < 4; ++r) red[w * 256 + r * 64 + lane] = v4[r];
        }
        __syncthreads();
        if (t < 256) {
            double v = red[t];
#pragma unroll
            for (int k = 1; k < 8; ++k) v += red[k * 256 + t];
            gout[blk * 256 + t] = v;
            if (blk == 0 && t == tN) {   // N_ab's share of this chunk, in the tile-partial layout the link packs
                double* o = wk.pair_part + ((long)pl * wk.NT + item) * 2;
                store_wt(o, v);
                store_wt(o + 1, 0.0);
            }
        }
    }
}

// The per-workgroup work of the operand launch AFTER the serial link: the operands of one (local pair, row chunk), or the
// mean part of one (local output, row chunk), or the reward -- selected by the item coordinates (bx, by) of a gx x gy item
// grid (k_mm_prep: its own block index; the persistent rollout kernel: a fixed item per workgroup).  NTHR: threads of
// the host workgroup; the work is laid out for 512, wider workgroups keep their extra waves idle between the barriers.
template <int DT, bool FUSED, int NTHR, bool FPAIR = false, bool PRE = false>
__device__ __forceinline__ void prep_work(const MMModel& md, const MMWork& wk, const PrepReward& pr, const GlueArgs& g, const GlueLds& L,
                                          double* sm_all, int glue_doubles, int bx, int by, int gx, int gy, int pair_a, int pair_b, double pre_la, double pre_lb,
                                          double pre_var) {
    double* sm = sm_all + (FUSED ? glue_doubles : 0);
    // the Gaussian this launch's operands are built for: the joint (x, u) the link assembled, or -- policy head of an
    // RbfController (GF_RBF_PRE) -- the state itself, the input of the policy GP.  (Integer offsets, not a pointer select.)
    const bool state_in = FUSED && (g.flags & GF_RBF_PRE);
    const double* jm = sm_all + (state_in ? 0 : L.o_js - L.nm);
    const double* js = sm_all + (state_in ? L.o_sx : L.o_js);
    if (bx >= wk.PL) {
        // spare workgroups of the launch: first the mean parts (local output, row chunk), then the reward
        const int idx = (bx - wk.PL) * gy + by;
        const int nmean = wk.EL * wk.NCHM;
        const int slot = 64 + 2 * (by * gx + bx);
        if (wk.dbg && threadIdx.x == 0 && slot < 958) wk.dbg[slot] = wall_clock64();
        if (idx < nmean) {
            prep_mean_block<DT, FUSED, NTHR, PRE>(md, wk, idx >> __builtin_ctz(wk.NCHM), idx & (wk.NCHM - 1), sm, jm, js, pre_la, pre_var);
            if (wk.dbg && threadIdx.x == 0 && slot < 958) wk.dbg[slot + 1] = wall_clock64();
            return;
        }
        // mean reward of the current (pre-propagation) state (rewards.py:19-81, pilco.py:133)
        if (idx != nmean || pr.n <= 0) return;
        const int E = pr.E, t = threadIdx.x;
        double* mx = sm;              // [E]
        double* sx = mx + E;          // [E][E]
        double* ws = sx + E * E;      // reward_lds_doubles(E)
        if (FUSED) {
            if (t < E) mx[t] = L.mx[t];
            for (int e = t; e < E * E; e += blockDim.x) sx[e] = L.sx[e];
        } else {
            if (t < E) mx[t] = pr.m_x[t];
            for (int e = t; e < E * E; e += blockDim.x) sx[e] = pr.s_x[e];
        }
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
    const bool act = (NTHR == 512) || t < 512;   // a host workgroup wider than 512 threads: the extra waves only keep the barriers
    const int grp = t >> 8, tl = t & 255;
    const int pl = bx, ch = by & (wk.NCH - 1), cs = by >> __builtin_ctz(wk.NCH);   // (NCH is a power of two: mm_prep_chunks)   // (cs: column split of the one-launch small step, 0 elsewhere)
    const bool dbg0 = (t == 0 && pl == 0 && ch == 0);
    DBG_STAMP(wk, 0, dbg0);
    if (wk.dbg && t == 0) wk.dbg[64 + 2 * (by * gx + bx)] = wall_clock64();
    if constexpr (!PRE) {   // (PRE: k_mm_prep wrote the constants before the link, the link stored the joint Gaussian here)
        if (t < DT) {
            double la = 1.0, lb = 1.0, mm = 0.0;
            if (t < D) {
                mm = FUSED ? jm[t] : wk.in_m[t];
                la = pre_la;
                lb = pre_lb;
            }
            s_m[t] = mm;
            s_ia2[t] = (t < D) ? 1.0 / (la * la) : 0.0;
            s_ib2[t] = (t < D) ? 1.0 / (lb * lb) : 0.0;
        }
        prep_wt_constants<DT>(md, wk, pl, sm);
        for (int e = t; e < D * D; e += 512) s_s[e] = FUSED ? js[e] : wk.in_s[e];
        for (int e = t; e < DT * DT; e += 512) s_Q[e] = 0.0;   // padded rows / columns of Q stay zero
        __syncthreads();
    }
    DBG_STAMP(wk, 1, dbg0);
    const int rpc = npad >> __builtin_ctz(wk.NCH);   // (NCH, NCHM, NCS are powers of two: mm_prep_chunks; a signed division is ~40 scalar operations on this serial path)
    const int i_begin = ch * rpc, i_end = i_begin + rpc;
    // The centred points of the chunk's first 256 rows are staged in LDS by the seven waves that do not run the
    // Gauss-Jordan, so their load latency (and the log of the signal variance) hides behind that phase.
    // side 0 (threads 0..255): x = zeta / la^2 -> row operand (2 Q z | u | 1); side 1: x = zeta / lb^2 -> column
    // operand (w | 1 | v).  A diagonal pair (a == b) is not special here: its mean part runs in prep_mean_block.
    const int side = grp;
    constexpr int LDZ = DT | 1;   // odd row stride: conflict-free LDS rows
    // (one-launch step of small models, below: the workgroup needs the column operand of ALL points -- the stage holds
    // points 0..255 = all of them, side 1 takes one point per thread, side 0 its rows of the chunk)
    const bool fpair = FPAIR && wk.fuse_pair;
    // ... and with 64 rows per workgroup and a contraction of at most 16 rows the operands stay in LDS (the stage's place,
    // once every thread has taken its point from it): Al [KP][64] | Bl [KP][npad] | vl [npad]
    const bool fplds = fpair && mm_kp(DT) <= 16 && rpc == 64;
    // ... and over column splits (MMWork::NCS): this workgroup's share of the pair's columns
    const int ncsg = (fplds && wk.NCS > 1) ? wk.NCS : 1, ncols = npad >> __builtin_ctz(ncsg), c_begin = cs * ncols, c_end = c_begin + ncols;
    double* Al = zst;
    double* Bl = Al + mm_kp(DT) * 64;
    double* vl = Bl + mm_kp(DT) * 256;
    // value-and-gradient form (wk.fuse_pair == 2, small_sweep below): z_i of the workgroup's rows [16][64] (row D: ones),
    // beta_b of all columns, per-wave column-sum slices
    const bool fsweep = fplds && DT <= 14 && wk.fuse_pair == 2;
    double* Zl = vl + 256;
    double* bbl = Zl + 16 * 64;
    double* csl = bbl + 256;
    const int st_begin = fpair ? 0 : i_begin, st_end = fpair ? npad : i_end;
    if (w != 0 && act) {
        const int idx = (w - 1) * 64 + lane;   // 0..447
        for (int e = idx; e < 256 * D; e += 448) {
            const int d = e >> 8, r = e & 255;
            const int i = st_begin + r;
            zst[r * LDZ + d] = (i < md.n && i < st_end) ? md.Pt[(long)d * npad + i] - s_m[d] : 0.0;
        }
        DBG_STAMP(wk, 7, t == 64 && pl == 0 && ch == 0);   // (the first staging wave is done)
    }
    const double logvar = pre_var;   // (log var of this thread's side: MMModel::lvar, loaded by the caller)
    if (MM_ABL(wk, 2)) {
        if (t == 0) s_sc[0] = 1.0;
    } else if (w == 0) {
        // [R | s],  R = s diag(la^-2 + lb^-2) + I        (mgpr.py:121-124,129); padded with identity
        // (branch-free: every lane reads its element of s -- or element 0 -- in ALL DT iterations and selects afterwards, so the DT
        // LDS reads leave back to back; with the two lane populations in divergent branches each iteration was two LDS round
        // trips of this lone wave: 1.4 us of the 2.6 us phase)
        double col[DT];
        const int c = lane;
        const int cc = c < DT ? c : c - DT;
        const bool in_s = cc < D && c < 2 * DT;
        const double lam = (c < DT && c < D) ? s_ia2[c] + s_ib2[c] : 0.0;
        double sv[DT];
#pragma unroll
        for (int r = 0; r < DT; ++r) sv[r] = s_s[(in_s && r < D) ? r * D + cc : 0];
#pragma unroll
        for (int r = 0; r < DT; ++r) {
            const double sval = (in_s && r < D) ? sv[r] : 0.0;
            col[r] = c < DT ? fma(sval, lam, (r == c) ? 1.0 : 0.0) : sval;   // lanes >= 2 DT: zeros
        }
        DBG_STAMP(wk, 5, dbg0);
        __builtin_amdgcn_s_setprio(3);   // (the wave the other seven wait for: ahead of the staging wave that shares its SIMD)
        const double det = gj_wave<DT>(col, colbuf, lane);
        __builtin_amdgcn_s_setprio(0);
        DBG_STAMP(wk, 6, dbg0);
        if (c >= DT && c < DT + D) {
#pragma unroll
            for (int r = 0; r < DT; ++r)
                if (r < D) s_Q[r * DT + (c - DT)] = 0.5 * col[r];
        }
        if (lane == 0) {
            s_sc[0] = fast_rsqrt(det);
            if (by == 0) store_wt(&wk.pair_isdet[pl], s_sc[0]);
        }
    }
    __syncthreads();
    DBG_STAMP(wk, 2, dbg0);
    const int KP = wk.KP;
    // what this workgroup writes to memory (layout: MMWork::At / Wt): the pair's (2 Q z_i | u_i) rows and v_j always; the
    // w rows of the pair's column block only when this pair is the block's writer this step (every pair with the same column
    // output computes the same w_j: P / E workgroups used to store identical values, half of the head's write-through traffic)
    const int pa_ = pair_a, pb_ = pair_b;   // (from the caller, found before the link)
    double* At = wk.At + (long)pl * KP * npad;
    double* Wb = wk.Wt + (long)pair_col_block(wk, pl, pb_) * KP * npad;
    double* vrow = wk.vcol + (long)pl * npad;
    const bool wt_dealt = wt_rows_dealt(wk);
    const bool wt_writer = !wt_dealt && pair_writes_wt(wk, md.E, pa_, pb_);
    const int wt_R = wt_rows_per_pair(wk);
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
        DBG_STAMP_ROW(wk, 34, dbg0);
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
        DBG_STAMP_ROW(wk, 35, dbg0);
        const double uv = valid ? (kk + quad) : 0.0;
        const double one = valid ? 1.0 : 0.0;
        if (fplds) {   // (workgroup-uniform) the same operands into LDS: nobody else reads them
            if (side == 0) {
                const int il = i - i_begin;
#pragma unroll
                for (int r = 0; r < DT; ++r)
                    if (r < D) Al[r * 64 + il] = 2.0 * y[r];
                Al[D * 64 + il] = uv;
                if (!wk.vsep) Al[(D + 1) * 64 + il] = one;
                for (int k = D + 2; k < KP; ++k) Al[k * 64 + il] = 0.0;
                if (fsweep) {   // z_i = zeta_i / l_a^2 (what side 0 calls x) and the ones, for the row side's epilogue
#pragma unroll
                    for (int r = 0; r < DT; ++r)
                        if (r < D) Zl[r * 64 + il] = x[r];
                    Zl[D * 64 + il] = one;
                }
            } else {
#pragma unroll
                for (int r = 0; r < DT; ++r)
                    if (r < D) Bl[r * 256 + i] = x[r];
                Bl[D * 256 + i] = one;
                if (wk.vsep) vl[i] = uv;
                else Bl[(D + 1) * 256 + i] = uv;
                for (int k = D + 2; k < KP; ++k) Bl[k * 256 + i] = 0.0;
            }
        } else if (side == 0) {
#pragma unroll
            for (int r = 0; r < DT; ++r)
                if (r < D) store_wt(&At[(long)r * npad + i], 2.0 * y[r]);   // 2 Q z_i (0 on padded rows)
            store_wt(&At[(long)D * npad + i], uv);                           // u_i   (the ones and the zero rows of A are model constants)
        } else {
            if (wt_writer) {   // (workgroup-uniform)
#pragma unroll
                for (int r = 0; r < DT; ++r)
                    if (r < D) store_wt(&Wb[(long)r * npad + i], x[r]);     // w_j
            }
            store_wt(&vrow[i], uv);   // v_j: row D + 1 of the contraction, or (vsep) added after the K = D + 1 contraction
            if (wt_dealt) {           // (workgroup-uniform) this pair's share of the w rows of ALL column blocks
                for (int q = 0; q < wt_R; ++q) {
                    const int rr = pl * wt_R + q;
                    if (rr >= md.E * D) break;
                    int dd;
                    const int bb = idiv_s(rr, D, dd);
                    store_wt(&wk.Wt[((long)bb * KP + dd) * npad + i], zst[(i - st_begin) * LDZ + dd] * colbuf[q]);
                }
            }
        }
    };
    if (fplds) {   // (workgroup-uniform) every thread takes its point from the stage, THEN the operands overwrite it
        const int r_begin = side ? c_begin : i_begin, r_end = side ? c_end : i_end;
        const bool has = act && r_begin + tl < r_end;
        const int i = r_begin + tl;
        double zeta[DT];
#pragma unroll
        for (int d = 0; d < DT; ++d) zeta[d] = (has && d < D) ? zst[(i - st_begin) * LDZ + d] : 0.0;
        __syncthreads();
        if (has) row(i, i < md.n, zeta);
    } else if (!MM_ABL(wk, 4) && act) {
        const int r_begin = (fpair && side) ? 0 : i_begin, r_end = (fpair && side) ? npad : i_end;   // this side's points
        if (r_begin + tl < r_end) {   // first row of this thread: centred point from the LDS stage
            const int i = r_begin + tl;
            double zeta[DT];
#pragma unroll
            for (int d = 0; d < DT; ++d) zeta[d] = (d < D) ? zst[(i - st_begin) * LDZ + d] : 0.0;
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
    // Small models (npad <= 256: BASELINE configs 4 and 5, every reference example): the pair sums of this workgroup's rows
    // follow right here instead of in a launch of their own -- the step is ONE launch.  At npad = 256 a pair's sums are 1 us
    // of the chip's fp64 pipe; as a second launch they cost 9.5 us (launch boundary, stream-K bookkeeping, first-touch
    // misses).  The workgroup holds rows [i_begin, i_end) of the row operand (side 0 wrote them) and -- side 1 has one
    // thread per point -- ALL npad columns of the column operand (every row chunk of the pair computes them: the same
    // values to the same addresses), so after a barrier its eight waves evaluate the (rows) x (all columns) block with
    // pair_wave, the arithmetic of the pair kernels, and publish ONE partial in the tile-partial layout the link packs.
    if (fsweep) {
        small_sweep<DT>(md, wk, pl, by, i_begin, c_begin, ncols, Al, Bl, vl, Zl, bbl, csl, sm + prep_region_doubles(DT), act);
    } else if (fpair) {   // (compiled into the single-rank fused heads only)
        constexpr int KCP = mm_kp(DT) / 4;
        constexpr bool VSP = mm_vsep(DT);
        const double* tab = sm + prep_region_doubles(DT);   // loaded at the head of the kernel (k_mm_prep), long ago
        double* wred = sm + prep_region_doubles(DT) + FEXP_TN;   // [8]
        __syncthreads();   // the operands (write-through stores) are visible to the whole workgroup
        int a, b;
        local_pair_ab(wk, md.E, pl, a, b);
        const bool diag = (a == b) && (md.iK != nullptr);
        const int nrg = rpc / (16 * PAIR_RT), ncs = max(1, 8 / nrg);   // row groups of 32 rows x column splits = waves at work
        double val = 0.0;
        if (act && w < nrg * ncs) {
            const int rg = w / ncs, cq = w - rg * ncs, ct = ncols / 16;   // (ncols = npad unless the pair's columns are split over workgroups)
            const int i0 = i_begin + 16 * PAIR_RT * rg, jb = c_begin + 16 * (ct * cq / ncs), je = c_begin + 16 * (ct * (cq + 1) / ncs);
            const double* beta_a = md.beta + mm_beta_row(md, a) * npad;
            const double* beta_b = md.beta + mm_beta_row(md, b) * npad;
            const double* iKa = diag ? md.iK + mm_ik_blk(md, a) * npad * npad : nullptr;
            const PairOps po = pair_ops(wk, D, npad, pl, pair_col_block(wk, pl, b));
            if (fplds && mm_kp(DT) <= 16) {
                if (diag) val = pair_wave<KCP, true, VSP, true, true>(po, Al, Bl, vl, beta_a, beta_b, iKa, tab, npad, i0, jb, je, lane, 64, 256, i0 - i_begin);
                else val = pair_wave<KCP, false, VSP, true, true>(po, Al, Bl, vl, beta_a, beta_b, nullptr, tab, npad, i0, jb, je, lane, 64, 256, i0 - i_begin);
            } else if (diag) {
                val = pair_wave<KCP, true, VSP, true>(po, nullptr, nullptr, nullptr, beta_a, beta_b, iKa, tab, npad, i0, jb, je, lane);
            } else {
                val = pair_wave<KCP, false, VSP, true>(po, nullptr, nullptr, nullptr, beta_a, beta_b, nullptr, tab, npad, i0, jb, je, lane);
            }
            for (int off = 32; off > 0; off >>= 1) val += __shfl_down(val, off);
        }
        if (act && lane == 0) wred[w] = val;
        __syncthreads();
        if (t == 0) {
            double* out = wk.pair_part + ((long)pl * wk.NT + by) * 2;
            store_wt(out, ((wred[0] + wred[1]) + (wred[2] + wred[3])) + ((wred[4] + wred[5]) + (wred[6] + wred[7])));
            store_wt(out + 1, 0.0);   // the trace term is already folded into out[0]
        }
    }
    DBG_STAMP(wk, 4, dbg0);
    if (wk.dbg && t == 0) wk.dbg[65 + 2 * (by * gx + bx)] = wall_clock64();
}

}  // namespace pilco
