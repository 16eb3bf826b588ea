// Reverse pass of one moment-matching step: k_mm_bwd_pair / _post / _fin.
#include "mm_device.h"

namespace pilco {

// ------------------------------------------------------------------ adjoint of the pair sums
// d e_ij/d m = P (z_i + w_j) and d e_ij/d s = (P y)(P y)^T / 2, so the reverse
// pass needs, per pair, only   r_i = sum_j W_ij L_ij,   c_j = sum_i W_ij L_ij,   m_i = sum_j W_ij L_ij w_j
// with W = beta_a beta_b^T (- iK_a on the diagonal pair).  A wave owns 16*BWD_RT rows and sweeps a range
// of columns; the exponent tile is computed TRANSPOSED (column operand as MFMA A, row operand as B) so
// that the weighted tile W.L lands in the B-operand layout of a second MFMA that contracts it with
// [w_j | 1]: moments and row sums cost 4 MFMAs per 16x16 tile and moment tile (NMT = ceil((D + 1) / 16) of them) and no
// VALU reductions.  The column sums of an off-diagonal pair run along the lanes of a DPP row: the wave parks its four
// result registers in a private LDS scratch and reads the previous step's back four at a time (two quad permutes finish
// the sum: 9 VALU ops per step instead of four row_shr adds per register); the four waves of a workgroup keep their
// column sums in separate LDS slices that are summed in a fixed order (diagonal pairs: c = r by symmetry).
// Off-diagonal pairs: the row side carries beta_b only, the column side beta_a only (k_mm_bwd_post applies the rest).
// rowmom[pl][js][16 NMT][npad]: d < D -> m_i[d], d = D -> r_i, per column split js;
// cpart[pl - E][row block][npad]: column sums over the rows of one workgroup.
#ifndef BWD_RT
#define BWD_RT 2
#endif
constexpr int BWD_CH = 64;   // columns staged per LDS chunk (one wave-wide row segment)
__host__ __device__ constexpr int bwd_tp(int kp) { return kp <= 16 ? 17 : kp + 1; }   // pitch of the staged column-major tile (doubles): operand rows + 1 (odd: conflict-free)
// sum_{q < n} base[q * stride] with the loads of a batch of B issued together (a plain loop serialises one global
// latency per term: these kernels are latency-bound); fixed summation order
template <int B>
__device__ __forceinline__ double sum_strided(const double* __restrict__ base, long stride, int n) {
    double acc = 0.0;
    for (int q0 = 0; q0 < n; q0 += B) {
        double v[B];
#pragma unroll
        for (int u = 0; u < B; ++u) v[u] = (q0 + u < n) ? base[(long)(q0 + u) * stride] : 0.0;
#pragma unroll
        for (int u = 0; u < B; ++u) acc += v[u];
    }
    return acc;
}

// Per-step constants of the reverse pass, computed once by spare workgroups of the k_mm_bwd_pair launch and read by
// k_mm_bwd_post / k_mm_bwd_fin:   head[h][D*D + D + 2]
//   output a (h = a):      T = (s + Lambda_a^2)^-1 | u = T Vbar_a | mu = Mbar_a - sum_b (Sbar_ab + Sbar_ba) M_b | c_a
//   pair pl (h = E + pl):  P = (I + Lambda_ab s)^-1 | lambda_ab | kappa = Shat_ab / sqrt(det R_ab) | 0
// M_b comes from the mean partials the prep kernel of the same step left in wk.mean_part.
// bars == nullptr (Jacobian tape, below): no cotangents -- u = 0, mu = 0 and kappa = 1 / sqrt(det R_ab).
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
    for (int e = t; e < nI; e += 256) o[e] = G[(e / D) * nc + D + (e % D)];
    if (h < E) {
        if (t < D) {
            double acc = 0.0;
            if (bars)
                for (int c = 0; c < D; ++c) acc = fma(G[t * nc + D + c], Vbar[c * E + a], acc);
            o[nI + t] = acc;
        }
        if (t == 64) {
            double mu = bars ? Mbar[a] : 0.0;
            for (int bb = 0; bars && bb < E; ++bb) {
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
            const double shat = !bars ? 1.0 : (a == b) ? Sbar[a * E + a] : Sbar[a * E + b] + Sbar[b * E + a];
            o[nI + D] = shat / sqrt(det);   // det(I + Lambda s) = det(s Lambda + I) = det R_ab
            o[nI + D + 1] = 0.0;
        }
    }
}

// NMT: 16-row tiles of the moment product (rows d = 0..D: w_j and the ones), NMT = ceil((D + 1) / 16); KC <= 4 keeps four
// waves per SIMD, wider contractions run at two
template <int KC, bool VSEP, int NMT>
#ifndef BWD_LBW
#define BWD_LBW 2
#endif
__global__ __launch_bounds__(256, (KC <= 4 && NMT == 1) ? 4 : BWD_LBW) void k_mm_bwd_pair(MMModel md, MMWork wk, double* __restrict__ rowmom,
                                                    double* __restrict__ cpart, int njs, const double* __restrict__ bars,
                                                    double* __restrict__ head, double* __restrict__ npart) {
    __shared__ double tab[FEXP_TN];
    __shared__ double nred[4];
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
    const double* beta_a = md.beta + mm_beta_row(md, a) * npad;
    const double* beta_b = md.beta + mm_beta_row(md, b) * npad;
    const bool diag = (a == b);
    const double* iKa = (diag && md.iK) ? md.iK + mm_ik_blk(md, a) * npad * npad : nullptr;
    // column range of this split: the npad / 16 column tiles dealt as evenly as they go (njs need not divide them)
    const int ct = npad / 16, jbeg = 16 * (int)((long)js * ct / njs), jend = 16 * (int)((long)(js + 1) * ct / njs);
    const int jw = jend - jbeg, jws = 16 * ((ct + njs - 1) / njs);   // jws: LDS slice stride (the widest split)
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
    // The column operands (KP rows of Bt, beta_b, v) are the same for the four waves of the workgroup: they are staged
    // once per BWD_CH columns through LDS (wave w fetches rows w, w + 4, .. as 512-byte row segments) into a
    // column-major tile T[j][BWD_TP] that serves both MFMA operand layouts -- cf (K = operand row, M = column) and its
    // transpose a2 (M = operand row, K = column) -- without bank conflicts (pitch 17 doubles); the next chunk is in
    // flight in registers while the current one is evaluated.  Only the iK stream of a diagonal pair stays a per-wave
    // buffer load.
    constexpr int KPc = 4 * KC, NR = KPc + (VSEP ? 2 : 1), NST = (NR + 3) / 4, BWD_TP = bwd_tp(KPc), SB = BWD_CH * BWD_TP + 2 * BWD_CH;
    const __amdgpu_buffer_rsrc_t rIK = buf_rsrc(iKa ? iKa + (long)jbeg * npad : Bt);
    const double* vsrc = VSEP ? wk.vcol + (long)pl * npad : Bt;
    unsigned ik_off[BWD_RT][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int rt = 0; rt < BWD_RT; ++rt) ik_off[rt][r] = ((unsigned)(lr + 4 * r) * (unsigned)npad + (unsigned)irow[rt]) * 8u;
    int dsel[NMT];                       // operand rows contracted by the second product: w_j (d < D), the ones (d = D);
#pragma unroll                           // lanes past that repeat row D: their result rows (d > D of rowmom) are never read
    for (int m = 0; m < NMT; ++m) dsel[m] = 16 * m + lc <= D ? 16 * m + lc : D;
    double* stg = csl + 4 * jws;         // [2][SB]
    auto stage_load = [&](int jc, double (&sg)[NST]) {
#pragma unroll
        for (int k = 0; k < NST; ++k) {
            const int row = w + 4 * k, col = jc + lane;   // wave-uniform row
            const double* src = row < KPc ? Bt + (long)row * npad : (row == KPc ? beta_b : vsrc);
            sg[k] = (row < NR && col < jend) ? src[col] : 0.0;
        }
    };
    auto stage_store = [&](double* buf, const double (&sg)[NST]) {
#pragma unroll
        for (int k = 0; k < NST; ++k) {
            const int row = w + 4 * k;
            if (row < KPc) buf[lane * BWD_TP + row] = sg[k];
            else if (row < NR) buf[BWD_CH * BWD_TP + (row - KPc) * BWD_CH + lane] = sg[k];
        }
    };
    d4 acc[BWD_RT][NMT];
#pragma unroll
    for (int rt = 0; rt < BWD_RT; ++rt)
#pragma unroll
        for (int m = 0; m < NMT; ++m) acc[rt][m] = d4{0.0, 0.0, 0.0, 0.0};
    double* myslice = csl + w * jws;
    // column-sum scratch of this wave: [4 result registers][64 lanes]; reader lane = (column jj = lane / 4, quarter q = lane % 4)
    // takes the four partials of lanes lc = 4 q .. 4 q + 3 of DPP row lr = jj % 4, register jj / 4
    double* scr = csl + 4 * jws + 2 * SB + w * 256;
    const int rd_jj = lane >> 2, rd_q = lane & 3;
    const double* rd_src = scr + (rd_jj >> 2) * 64 + (rd_jj & 3) * 16 + 4 * rd_q;
    int jprev = -1;
    auto flush_cols = [&](int jp) {
        double p = (rd_src[0] + rd_src[1]) + (rd_src[2] + rd_src[3]);
        p = dpp_add<0xB1, 0xf>(p);   // quad_perm [1,0,3,2]
        p = dpp_add<0x4E, 0xf>(p);   // quad_perm [2,3,0,1]: every lane of the quad holds the sum over the 16 rows
        if (rd_q == 0) myslice[jp - jbeg + rd_jj] = p;
    };
    // the column sweep, specialised at compile time (branches inside the loop would fence the scheduler between the
    // eight exp evaluations of a step): MODE 0 off-diagonal pair (column sums), 1 diagonal pair with the iK stream,
    // 2 diagonal pair without it (RBF policy GP)
    auto sweep = [&](auto mode_c) {
        constexpr int MODE = decltype(mode_c)::value;
        double sg[NST];
        stage_load(jbeg, sg);
        stage_store(stg, sg);
        __syncthreads();
        int cur = 0;
        for (int jc = jbeg; jc < jend; jc += BWD_CH) {
            const bool more = jc + BWD_CH < jend;
            if (more) stage_load(jc + BWD_CH, sg);
            const double* Tb = stg + cur * SB;
            const double* bS = Tb + BWD_CH * BWD_TP;
            const double* vS = bS + BWD_CH;
            const int nst = min(BWD_CH, jend - jc);
            for (int jl = 0; jl < nst; jl += 16) {
                const int j0 = jc + jl;
                double cf[KC], a2[NMT][4], bcol[4], vj[4];
    #pragma unroll
                for (int c = 0; c < KC; ++c) cf[c] = Tb[(jl + lc) * BWD_TP + 4 * c + lr];
    #pragma unroll
                for (int r = 0; r < 4; ++r) {
                    bcol[r] = bS[jl + lr + 4 * r];
                    vj[r] = VSEP ? vS[jl + lr + 4 * r] : 0.0;   // transposed tile: v_j runs along the result registers
#pragma unroll
                    for (int m = 0; m < NMT; ++m) a2[m][r] = Tb[(jl + 4 * r + lr) * BWD_TP + dsel[m]];
                }
                double csum[4] = {0.0, 0.0, 0.0, 0.0};
    #pragma unroll
                for (int rt = 0; rt < BWD_RT; ++rt) {
                    d4 e = {0.0, 0.0, 0.0, 0.0};
    #pragma unroll
                    for (int c = 0; c < KC; ++c)
                        e = __builtin_amdgcn_mfma_f64_16x16x4f64(cf[c], rf[rt][c], e, 0, 0, 0);   // e[r]: i = irow, j = j0+lr+4r
                    MFMA_KEEP_ALIVE(cf[0]);      // (the first MFMA of the chain has a constant-zero accumulator)
                    MFMA_KEEP_ALIVE(rf[rt][0]);
                    // this hipcc leaves the instantiations below without the wait states (the K = D + 1 ones with one moment
                    // tile get them: there the fence would only cost its 11 slots twice per step); tests/test_build_isa.py
                    // scans the generated code of ALL instantiations for reads that come too early
                    if (!VSEP || KC > 4 || NMT > 1) MFMA_RESULT_FENCE(e);
                    double wl[4];
    #pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (MODE == 0) {
                            // off-diagonal pair: W = beta_a beta_b^T is separable -- the row side carries beta_b,j only and
                            // the column side beta_a,i only (one multiply and one FMA instead of two multiplies and an
                            // add); k_mm_bwd_post applies the missing factor per row / per column
                            const double l = fexp(VSEP ? e[r] + vj[r] : e[r], tab);
                            wl[r] = bcol[r] * l;
                            csum[r] = fma(brow[rt], l, csum[r]);
                        } else {
                            double wgt = brow[rt] * bcol[r];
                            if (MODE == 1) wgt -= buf_ld(rIK, ik_off[rt][r], (unsigned)(j0 - jbeg) * (unsigned)npad * 8u);   // iK symmetric: coalesced along the rows
                            wl[r] = wgt * fexp(VSEP ? e[r] + vj[r] : e[r], tab);
                        }
                    }
    #pragma unroll
                    for (int m = 0; m < NMT; ++m)
    #pragma unroll
                        for (int r = 0; r < 4; ++r) acc[rt][m] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2[m][r], wl[r], acc[rt][m], 0, 0, 0);
                }
                if (MODE == 0) {
                    // column sums over the wave's 16 lanes of a DPP row, through a per-wave LDS scratch instead of four
                    // DPP row shifts per register (12 VALU ops each on the pipe this kernel is bound by): the partial sums
                    // of the PREVIOUS column step are read back four at a time, added and finished with two quad
                    // permutes (9 VALU ops per step instead of 48); LDS operations of one wave execute in order
#ifdef BWD_DPP_COLS
    #pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        double v = csum[r];
                        v = dpp_add<0x111, 0xf>(v);
                        v = dpp_add<0x112, 0xf>(v);
                        v = dpp_add<0x114, 0xf>(v);
                        v = dpp_add<0x118, 0xf>(v);
                        if (lc == 15) myslice[j0 - jbeg + lr + 4 * r] = v;
                    }
#else
                    if (jprev >= 0) flush_cols(jprev);
    #pragma unroll
                    for (int r = 0; r < 4; ++r) scr[r * 64 + lane] = csum[r];
                    jprev = j0;
#endif
                }
            }
            if (more) stage_store(stg + (cur ^ 1) * SB, sg);
            __syncthreads();
            cur ^= 1;
        }
        if (MODE == 0 && jprev >= 0) flush_cols(jprev);
    };
    if (!diag) sweep(std::integral_constant<int, 0>{});
    else if (iKa) sweep(std::integral_constant<int, 1>{});
    else sweep(std::integral_constant<int, 2>{});
    double* out = rowmom + ((long)pl * njs + js) * (16 * NMT) * npad;
#pragma unroll
    for (int rt = 0; rt < BWD_RT; ++rt)
        if (ibase + 16 * rt < npad) {
#pragma unroll
            for (int m = 0; m < NMT; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (16 * m + lr + 4 * r <= D) out[(long)(16 * m + lr + 4 * r) * npad + irow[rt]] = acc[rt][m][r];   // rows past D are never read
        }
    if (!diag) {
        __syncthreads();
        double* cp = cpart + ((long)(pl - wk.EL) * gridDim.x + rb) * npad + jbeg;
        for (int jj = threadIdx.x; jj < jw; jj += 256)
            cp[jj] = (csl[jj] + csl[jws + jj]) + (csl[2 * jws + jj] + csl[3 * jws + jj]);
    }
    if (npart) {
        // Jacobian tape: this workgroup's share of N_ab = sum_i r_i, in the [pair][tile][2] layout the serial link packs
        // tile partials from (tile = js * row blocks + rb); r_i is row D of the moment tile: lanes lr == D % 4, register D / 4
        const int rsel = (D & 15) >> 2, msel = D >> 4;
        double v = 0.0;
#pragma unroll
        for (int rt = 0; rt < BWD_RT; ++rt) {
            double x = 0.0;
#pragma unroll
            for (int m = 0; m < NMT; ++m)
                if (m == msel) x = rsel == 0 ? acc[rt][m][0] : rsel == 1 ? acc[rt][m][1] : rsel == 2 ? acc[rt][m][2] : acc[rt][m][3];
            v += (lr == (D & 3) && ibase + 16 * rt < npad) ? (diag ? x : x * brow[rt]) : 0.0;   // tiles past the padding carry garbage and are never stored
        }
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
        if (lane == 0) nred[w] = v;
        __syncthreads();
        if (threadIdx.x == 0) {
            double* o = npart + ((long)pl * (njs * (int)gridDim.x) + js * (int)gridDim.x + rb) * 2;
            o[0] = (nred[0] + nred[1]) + (nred[2] + nred[3]);
            o[1] = 0.0;
        }
    }
}

// Reverse of the mean part (mgpr.py:99-118) for output a, including the -M M^T term of S:
// with T = (s + Lambda_a^2)^-1, l_i = beta_i exp(-zeta_i^T T zeta_i / 2), g = sum l_i, h = sum l_i zeta_i,
// u = T Vbar_a, mu = Mbar_a - sum_b (Sbar_ab + Sbar_ba) M_b, q_i = mu + zeta_i . u:
//   mbar_a = c (T sum l_i q_i zeta_i - g u),
//   sbar_a = -phi T / 2 + c T (sum l_i q_i zeta_i zeta_i^T) T / 2 - c (u (T h)^T + (T h) u^T) / 2,  phi = c (mu g + Vbar_a . T h).
// M_b is read from the mean partials the prep kernel of the same step left in wk.mean_part.
// stage 1 (a workgroup of k_mm_bwd_post): sums over the 64-point blocks rc, rc + nrc, ..:  mpart[a][rc][D*D + 2D + 1]
template <int NA>   // sums per thread: D*D + 2 D + 1 <= NA * 256  (NA = 1: D <= 14, NA = 5: D <= 32)
__device__ void bwd_mean_partial(const MMModel& md, const double* __restrict__ in_m, const double* __restrict__ head, int a, int rc,
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
    const int ntot = nI + 2 * D + 1; // H2q [D][D] | wq [D] | h [D] | g
    double acc[NA];
#pragma unroll
    for (int k = 0; k < NA; ++k) acc[k] = 0.0;
    for (int blk = rc; blk < npad / 64; blk += nrc) {
        if (t < 64) {
            const int i = blk * 64 + t;
            double l = 0.0, q = 0.0;
            if (i < md.n) {
                double quad = 0.0;
                q = mu;
                for (int d0 = 0; d0 < D; d0 += 16) {
                    double pv[16];   // sixteen coordinates of the point in flight together
#pragma unroll
                    for (int d = 0; d < 16; ++d) pv[d] = (d0 + d < D) ? md.Pt[(long)(d0 + d) * npad + i] : 0.0;
#pragma unroll
                    for (int d = 0; d < 16; ++d)
                        if (d0 + d < D) zs[t * LD + d0 + d] = pv[d] - in_m[d0 + d];
                }
                for (int r = 0; r < D; ++r) {
                    double tz = 0.0;
                    for (int c = 0; c < D; ++c) tz = fma(T[r * D + c], zs[t * LD + c], tz);
                    quad = fma(zs[t * LD + r], tz, quad);
                    q = fma(zs[t * LD + r], u[r], q);
                }
                l = exp(-0.5 * quad) * md.beta[mm_beta_row(md, a) * npad + i];
            } else {
                for (int d = 0; d < D; ++d) zs[t * LD + d] = 0.0;
            }
            lv[t] = l;
            lq[t] = l * q;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < NA; ++k) {
            const int e = t + 256 * k;
            if (e >= ntot) break;
            double a2 = acc[k];
            if (e < nI) {
                const int d = e / D, e2 = e - d * D;
                _Pragma("unroll 4") for (int ii = 0; ii < 64; ++ii) a2 = fma(lq[ii] * zs[ii * LD + d], zs[ii * LD + e2], a2);
            } else if (e < nI + D) {
                const int d = e - nI;
                _Pragma("unroll 4") for (int ii = 0; ii < 64; ++ii) a2 = fma(lq[ii], zs[ii * LD + d], a2);
            } else if (e < nI + 2 * D) {
                const int d = e - nI - D;
                _Pragma("unroll 4") for (int ii = 0; ii < 64; ++ii) a2 = fma(lv[ii], zs[ii * LD + d], a2);
            } else {
                _Pragma("unroll 4") for (int ii = 0; ii < 64; ++ii) a2 += lv[ii];
            }
            acc[k] = a2;
        }
        __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < NA; ++k)
        if (t + 256 * k < ntot) mpart[((long)a * nrc + rc) * ntot + t + 256 * k] = acc[k];
}

// ---- Jacobian tape (cotangent-free form of the same reverse pass)
// Everything above that costs O(N^2) or O(N D^2) is independent of the cotangents: the pair sums (N, A, I) are, and the
// mean part depends on (mu, u) only through q_i = mu + zeta_i . u, i.e. through the moments of l_i up to third order.  A
// forward rollout that runs the reverse sweep INSTEAD of the forward pair kernel therefore gets the value (N_ab) and the
// complete Jacobian of the step's outputs with respect to (m, s) from one O(N^2) pass; the reverse sweep of the policy
// gradient is then a chain of small contractions with no device work at all (csrc/grad.hip).
// Moments with the point extended by a one, zeta~ = (zeta | 1): H(d, e, f) = sum_i l_i zeta~_d zeta~_e zeta~_f for
// d >= e >= f -- g = H(D,D,D), h_d = H(D,D,d), H2_de = H(D,d,e), H3_def.  mpart[a][rc][NS], NS = (D+1)(D+2)(D+3)/6.
__host__ __device__ inline int tri3(int d, int e, int f) { return d * (d + 1) * (d + 2) / 6 + e * (e + 1) / 2 + f; }   // d >= e >= f
__device__ __forceinline__ int tri3_any(int a, int b, int c) {
    const int hi = max(a, max(b, c)), lo = min(a, min(b, c));
    return tri3(hi, a + b + c - hi - lo, lo);
}
int mm_jac_ns(int D) { return (D + 1) * (D + 2) * (D + 3) / 6; }
constexpr int JAC_MAXA = 4;   // moment sums per thread: NS <= 4 * 256 (D <= 16)
__device__ void bwd_mean_moments(const MMModel& md, const double* __restrict__ in_m, const double* __restrict__ head, int a, int rc,
                                 int nrc, double* __restrict__ mpart, double* sm) {
    const int D = md.D, D1 = D + 1, npad = md.npad, t = threadIdx.x;
    const int nI = D * D, LD = D1 | 1, NS = D1 * (D1 + 1) * (D1 + 2) / 6;
    double* T = sm;                 // [D][D]
    double* zs = T + nI;            // [64][LD]   zeta | 1
    double* lv = zs + 64 * LD;      // [64]
    int* tri = (int*)(lv + 64);     // [NS] packed index triples
    const double* hd = head + (long)a * (nI + D + 2);
    for (int e = t; e < nI; e += 256) T[e] = hd[e];
    if (t < D1) {
        int off = t * (t + 1) * (t + 2) / 6;
        for (int e = 0; e <= t; ++e)
            for (int f = 0; f <= e; ++f) tri[off++] = t | (e << 8) | (f << 16);
    }
    __syncthreads();
    int pk[JAC_MAXA];
    double acc[JAC_MAXA];
#pragma unroll
    for (int k = 0; k < JAC_MAXA; ++k) {
        const int idx = t + 256 * k;
        pk[k] = idx < NS ? tri[idx] : -1;
        acc[k] = 0.0;
    }
    for (int blk = rc; blk < npad / 64; blk += nrc) {
        if (t < 64) {
            const int i = blk * 64 + t;
            double l = 0.0;
            if (i < md.n) {
                double quad = 0.0;
                {
                    double pv[16];
#pragma unroll
                    for (int d = 0; d < 16; ++d) pv[d] = (d < D) ? md.Pt[(long)d * npad + i] : 0.0;
#pragma unroll
                    for (int d = 0; d < 16; ++d)
                        if (d < D) zs[t * LD + d] = pv[d] - in_m[d];
                }
                for (int r = 0; r < D; ++r) {
                    double tz = 0.0;
                    for (int c = 0; c < D; ++c) tz = fma(T[r * D + c], zs[t * LD + c], tz);
                    quad = fma(zs[t * LD + r], tz, quad);
                }
                l = exp(-0.5 * quad) * md.beta[mm_beta_row(md, a) * npad + i];
            } else {
                for (int d = 0; d < D; ++d) zs[t * LD + d] = 0.0;
            }
            zs[t * LD + D] = 1.0;
            lv[t] = l;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < JAC_MAXA; ++k)
            if (pk[k] >= 0) {
                const int d = pk[k] & 255, e = (pk[k] >> 8) & 255, f = pk[k] >> 16;
                double a2 = acc[k];
                _Pragma("unroll 4") for (int ii = 0; ii < 64; ++ii)
                    a2 = fma(lv[ii] * zs[ii * LD + d], zs[ii * LD + e] * zs[ii * LD + f], a2);
                acc[k] = a2;
            }
        __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < JAC_MAXA; ++k)
        if (pk[k] >= 0) mpart[((long)a * nrc + rc) * NS + t + 256 * k] = acc[k];
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
    for (int e = t; e <= nI + 2 * D; e += 256)
        red[e] = sum_strided<16>(mpart + (long)a * nrc * (nI + 2 * D + 1) + e, nI + 2 * D + 1, nrc);   // fixed order
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
    for (int e = t; e < nI; e += 256) {
        const int r = e / D, c = e - r * D;
        double acc2 = 0.0;
        for (int k = 0; k < D; ++k) acc2 = fma(T[r * D + k], H2q[k * D + c], acc2);
        TH[e] = acc2;
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
    for (int e = t; e < nI; e += 256) {
        const int r = e / D, c = e - r * D;
        double acc2 = 0.0;
        for (int k = 0; k < D; ++k) acc2 = fma(TH[r * D + k], T[k * D + c], acc2);
        o[D + e] = -0.5 * phi * T[r * D + c] + 0.5 * c_a * acc2 - 0.5 * c_a * (u[r] * Th[c] + Th[r] * u[c]);
    }
    if (t < D) {
        const int r = t;
        double tw = 0.0;
        for (int c = 0; c < D; ++c) tw = fma(T[r * D + c], wq[c], tw);
        o[r] = c_a * (tw - g * u[r]);
    }
}

// Per unordered pair and row chunk: partial sums of  N_ab = sum_i r_i,  A = sum_i (r_i z_i + c_i w_i)  (D),
// I = sum_i (r_i z_i z_i^T + c_i w_i w_i^T + z_i m_i^T + m_i z_i^T)  (D x D).   part[pl][chunk][1 + D + D*D]
constexpr int BWD_RC = 8;   // row chunks per pair / output (16: 12 % slower in the batched form, more workgroup prologues)
// Batched form (Jacobian tape): blockIdx.z = horizon step; every per-step array advances by its stride and the input
// mean comes from the step's tape record (wk.in_m holds the LAST step's by then).  Unbatched: strides 0, in_m = nullptr.
struct BwdBatch {
    long rowmom, cpart, part, head;
    const double* in_m;
    long in_m_stride;
};
template <int NA>   // NA = 1: D <= 14 (one sum per thread everywhere), NA = 5: D <= 32
#ifndef POST_LB
#define POST_LB 8   // waves per SIMD of the narrow post kernel: latency-bound, 8 resident workgroups per CU (64 VGPRs, a few spills) beat 4 by 30 %
#endif
__global__ __launch_bounds__(256, NA == 1 ? POST_LB : 1) void k_mm_bwd_post(MMModel md, MMWork wk, const double* __restrict__ rowmom,
                                                    const double* __restrict__ cpart, int njs, int nrb,
                                                    double* __restrict__ part, int nrc,
                                                    const double* __restrict__ head, double* __restrict__ mpart, int jac, BwdBatch bb) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int npad = md.npad, D = md.D, E = md.E, t = threadIdx.x;
    const int pl = blockIdx.x, rc = blockIdx.y, z = blockIdx.z;
    rowmom += (long)z * bb.rowmom;
    cpart += (long)z * bb.cpart;
    part += (long)z * bb.part;
    mpart += (long)z * bb.part;
    head += (long)z * bb.head;
    const double* in_m = bb.in_m ? bb.in_m + (long)z * bb.in_m_stride : wk.in_m;
    if (pl >= wk.PL) {   // the last E workgroup columns: mean part of output pl - PL
        if (jac && NA == 1) bwd_mean_moments(md, in_m, head, pl - wk.PL, rc, nrc, mpart, sm);
        else bwd_mean_partial<NA>(md, in_m, head, pl - wk.PL, rc, nrc, mpart, sm);
        return;
    }
    int a, b;
    local_pair_ab(wk, E, pl, a, b);
    const int mrows = 16 * ((D + 16) / 16);   // rows of a moment block: 16 per moment tile of the sweep
    const double* mom0 = rowmom + (long)pl * njs * mrows * npad;
    const double* cp = (a != b) ? cpart + (long)(pl - wk.EL) * nrb * npad : nullptr;
    const double* beta_a = md.beta + mm_beta_row(md, a) * npad;
    const double* beta_b = md.beta + mm_beta_row(md, b) * npad;
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
        mm[t] = in_m[t];
    }
    const int NT2 = D * (D + 1) / 2, per = NT2 + D + 1;          // I (packed) | A | N
    const int NG = per <= 85 ? 3 : per <= 128 ? 2 : 1;           // thread groups sharing the points of a block
    const int grp = t / per, idx = t - grp * per;
    int pd = 0;
    while (idx < NT2 && (pd + 1) * (pd + 2) / 2 <= idx) ++pd;    // idx = pd (pd + 1) / 2 + pe, pe <= pd
    const int pe = idx < NT2 ? idx - pd * (pd + 1) / 2 : 0;
    constexpr int NP = NA == 1 ? 1 : 3;   // pair sums per thread
    double acc[NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) acc[k] = 0.0;
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
            if (valid) {
                mv = sum_strided<4>(mom0 + (long)d * npad + i, (long)mrows * npad, njs);
                if (cp) mv *= beta_a[i];   // off-diagonal pair: the sweep left beta_a,i out of the row side
            }
            ms[ii * LD + d] = mv;
        }
        if (t < 64) {
            const int i = i0 + t;
            const bool valid = i < md.n;
            double r = 0.0, c = 0.0;
            if (valid) {
                r = sum_strided<4>(mom0 + (long)D * npad + i, (long)mrows * npad, njs);
                if (cp) {
                    r *= beta_a[i];
                    c = sum_strided<8>(cp + i, npad, nrb) * beta_b[i];   // ... and beta_b,j out of the column side
                } else {
                    c = r;
                }
            }
            rs[t] = r;
            cs[t] = c;
        }
        __syncthreads();
        // I is symmetric: only d >= e2 is accumulated (NT2 entries), and the 64 points of the block are dealt over NG
        // thread groups whose partial sums are added in a fixed order at the end (wide inputs, per > 256: one group,
        // up to three entries per thread)
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const int ek = (NG > 1) ? idx : t + 256 * k;
            if ((NG > 1 && (k > 0 || grp >= NG)) || ek >= per) break;
            const int ii0 = (NG > 1) ? grp * 64 / NG : 0, ii1 = (NG > 1) ? (grp + 1) * 64 / NG : 64;
            double a2 = acc[k];
            if (ek < NT2) {
                int qd = pd, qe = pe;
                if (k > 0) {
                    qd = 0;
                    while ((qd + 1) * (qd + 2) / 2 <= ek) ++qd;
                    qe = ek - qd * (qd + 1) / 2;
                }
                for (int ii = ii0; ii < ii1; ++ii) {
                    const double zd = zs[ii * LD + qd], ze = zs[ii * LD + qe];
                    a2 = fma(rs[ii] * zd, ze, a2);
                    a2 = fma(cs[ii] * ws[ii * LD + qd], ws[ii * LD + qe], a2);
                    a2 = fma(zd, ms[ii * LD + qe], a2);
                    a2 = fma(ms[ii * LD + qd], ze, a2);
                }
            } else if (ek < NT2 + D) {
                const int d = ek - NT2;
                for (int ii = ii0; ii < ii1; ++ii) a2 = fma(rs[ii], zs[ii * LD + d], fma(cs[ii], ws[ii * LD + d], a2));
            } else {
                for (int ii = ii0; ii < ii1; ++ii) a2 += rs[ii];
            }
            acc[k] = a2;
        }
    }
    __syncthreads();
    double* o = part + ((long)pl * nrc + rc) * (1 + D + nI);
    auto put = [&](int e, double v) {
        if (e < NT2) {
            int qd = 0;
            while ((qd + 1) * (qd + 2) / 2 <= e) ++qd;
            const int qe = e - qd * (qd + 1) / 2;
            o[1 + D + qd * D + qe] = v;
            o[1 + D + qe * D + qd] = v;
        } else if (e < NT2 + D) {
            o[1 + (e - NT2)] = v;
        } else {
            o[0] = v;
        }
    };
    if (NG > 1) {
        double* red = zs;   // [NG][per]
        if (grp < NG) red[grp * per + idx] = acc[0];
        __syncthreads();
        if (t < per) {
            double v = red[t];
            for (int g2 = 1; g2 < NG; ++g2) v += red[g2 * per + t];
            put(t, v);
        }
    } else {
#pragma unroll
        for (int k = 0; k < NP; ++k)
            if (t + 256 * k < per) put(t + 256 * k, acc[k]);
    }
}

// Per pair, with P = (I + Lambda s)^-1 and kappa = Shat_ab / sqrt(det R_ab) from the step's head record:
//   mbar += kappa P A,   sbar += kappa (P I P^T / 2 - N (P Lambda + Lambda P^T) / 4)      
// out[E + pl][D + D*D].  bars = (Mbar [E] | Sbar [E][E] | Vbar [D][E]) on the device.  Workgroups past the pairs
// finish the mean part of one output each.
// The workgroup that finishes last (a device-scope counter) adds the E + P records up in their fixed order and writes the
// sum (mbar | sbar before symmetrisation) to `sum_out` -- pinned host memory the caller reads after the stream has
// drained: no device-to-host copy command, no host loop over the records.
__device__ void bwd_fin_pairs(const MMModel& md, const MMWork& wk, const double* __restrict__ part, int nrc,
                              const double* __restrict__ head, double* __restrict__ out, double* sm);

__global__ __launch_bounds__(256) void k_mm_bwd_fin(MMModel md, MMWork wk, const double* __restrict__ part, int nrc,
                                                   const double* __restrict__ bars, const double* __restrict__ head,
                                                   const double* __restrict__ mpart, double* out,
                                                   unsigned* __restrict__ done, double* __restrict__ sum_out) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    __shared__ int last;
    const int D = md.D, E = md.E, t = threadIdx.x, pl = blockIdx.x;
    if (pl >= wk.PL) bwd_mean_final(md, bars, head, pl - wk.PL, nrc, mpart, out, sm);
    else bwd_fin_pairs(md, wk, part, nrc, head, out, sm);
    __threadfence();
    __syncthreads();
    if (t == 0) last = (atomicAdd(done, 1u) == gridDim.x - 1) ? 1 : 0;
    __syncthreads();
    if (!last) return;
    __threadfence();
    const int rec = D + D * D;
    for (int e = t; e < rec; e += 256) sum_out[e] = sum_strided<16>(out + e, rec, E + wk.PL);
    if (t == 0) *done = 0u;
}

__device__ void bwd_fin_pairs(const MMModel& md, const MMWork& wk, const double* __restrict__ part, int nrc,
                              const double* __restrict__ head, double* __restrict__ out, double* sm) {
    const int D = md.D, E = md.E, t = threadIdx.x, pl = blockIdx.x;
    const int nI = D * D, rec = 1 + D + nI;
    double* Pm = sm;               // [D][D]
    double* lam = Pm + nI;         // [D + 2]: lambda | kappa
    double* Iv = lam + D + 2;      // [rec]  summed partials (N | A | I)
    double* PI = Iv + rec;         // [D][D]
    const double* hd = head + (long)(E + pl) * (nI + D + 2);
    for (int e = t; e < nI + D + 2; e += 256) (e < nI ? Pm[e] : lam[e - nI]) = hd[e];
    for (int e = t; e < rec; e += 256) {
        double acc = 0.0;
        acc = sum_strided<16>(part + (long)pl * nrc * rec + e, rec, nrc);   // fixed order
        Iv[e] = acc;
    }
    __syncthreads();
    const double kappa = lam[D];
    const double Nab = Iv[0];
    const double* Av = Iv + 1;
    const double* Im = Iv + 1 + D;
    for (int e = t; e < nI; e += 256) {
        const int r = e / D, c = e - r * D;
        double acc = 0.0;
        for (int k = 0; k < D; ++k) acc = fma(Pm[r * D + k], Im[k * D + c], acc);
        PI[e] = acc;
    }
    __syncthreads();
    double* o = out + (long)(E + pl) * (D + nI);
    for (int e = t; e < nI; e += 256) {
        const int r = e / D, c = e - r * D;
        double acc = 0.0;
        for (int k = 0; k < D; ++k) acc = fma(PI[r * D + k], Pm[c * D + k], acc);   // (P I P^T)[r][c]
        const double pl2 = Pm[r * D + c] * lam[c] + Pm[c * D + r] * lam[r];         // P Lambda + Lambda P^T
        o[D + e] = kappa * (0.5 * acc - 0.25 * Nab * pl2);
    }
    if (t < D) {
        const int r = t;
        double acc = 0.0;
        for (int c = 0; c < D; ++c) acc = fma(Pm[r * D + c], Av[c], acc);
        o[r] = kappa * acc;
    }
}

// ---- Jacobian tape, last stage: one record per pair and per output (layout: mm_jac_rec_size)
//   pair pl:   N_ab | g = rdet P A (D) | sym G, G = rdet (P I P^T / 2 - N (P Lambda + Lambda P^T) / 4),  rdet = 1 / sqrt(det R_ab)
//              => a cotangent Shat_ab of S_ab contributes  mbar += Shat g,  sbar += Shat sym G
//   output a:  dM/dm (D) | sym dM/ds | dV_k/dm (D x D: [k][r]) | sym dV_k/ds (D of them: [k][..])
//              => cotangents (mu_a, Vbar_a) contribute  mbar += mu dM/dm + sum_k Vbar_k dV_k/dm,  sbar likewise
// every symmetric D x D matrix is stored packed: (X + X^T) / 2 at [c (c + 1) / 2 + r], r <= c  (D (D + 1) / 2 doubles)
// and N_ab in the layout the pack stage of the serial link reads tile partials in (pair_n[pl][2], NT = 1).
__device__ void jac_fin_output(const MMModel& md, const double* __restrict__ head, int a, int nrc,
                               const double* __restrict__ mpart, double* __restrict__ rec, double* sm) {
    const int D = md.D, D1 = D + 1, t = threadIdx.x;
    const int nI = D * D, n3 = nI * D, NS = D1 * (D1 + 1) * (D1 + 2) / 6;
    double* T = sm;            // [D][D]
    double* Hs = T + nI;       // [NS]
    double* Th = Hs + NS;      // [D]
    double* TH = Th + D;       // [D][D]   T H2
    double* THT = TH + nI;     // [D][D]   T H2 T
    double* W3 = THT + nI;     // [D][D][D]  W3[k][d][e] = sum_f H3(d,e,f) T[f][k]
    double* Z3 = W3 + n3;      // [D][D][D]  Z3[k][r][e] = sum_d T[r][d] W3[k][d][e]
    const double* hd = head + (long)a * (nI + D + 2);
    for (int e = t; e < nI; e += 256) T[e] = hd[e];
    const double c = hd[nI + D + 1];
    for (int e = t; e < NS; e += 256) Hs[e] = sum_strided<16>(mpart + (long)a * nrc * NS + e, NS, nrc);   // fixed order
    __syncthreads();
    const double g = Hs[tri3(D, D, D)];
    if (t < D) {
        double acc = 0.0;
        for (int cc = 0; cc < D; ++cc) acc = fma(T[t * D + cc], Hs[tri3(D, D, cc)], acc);
        Th[t] = acc;
    }
    for (int e = t; e < nI; e += 256) {
        const int r = e / D, cc = e - r * D;
        double acc = 0.0;
        for (int k = 0; k < D; ++k) acc = fma(T[r * D + k], Hs[tri3(D, max(k, cc), min(k, cc))], acc);
        TH[e] = acc;
    }
    for (int e = t; e < n3; e += 256) {
        const int k = e / nI, d = (e / D) % D, e2 = e % D;
        double acc = 0.0;
        for (int f = 0; f < D; ++f) acc = fma(Hs[tri3_any(d, e2, f)], T[f * D + k], acc);
        W3[e] = acc;
    }
    __syncthreads();
    for (int e = t; e < nI; e += 256) {
        const int r = e / D, cc = e - r * D;
        double acc = 0.0;
        for (int k = 0; k < D; ++k) acc = fma(TH[r * D + k], T[k * D + cc], acc);
        THT[e] = acc;
    }
    for (int e = t; e < n3; e += 256) {
        const int k = e / nI, r = (e / D) % D, e2 = e % D;
        double acc = 0.0;
        for (int d = 0; d < D; ++d) acc = fma(T[r * D + d], W3[(k * D + d) * D + e2], acc);
        Z3[e] = acc;
    }
    __syncthreads();
    // full dV_k/ds into W3 (dead by now), then everything symmetric goes out packed: X -> (X + X^T) / 2 at [c (c + 1) / 2 + r],
    // r <= c (the caller symmetrises the sum anyway; half the bytes to download and to stream through the host)
    for (int e = t; e < n3; e += 256) {
        const int k = e / nI, r = (e / D) % D, cc = e % D;
        double acc = 0.0;
        for (int e2 = 0; e2 < D; ++e2) acc = fma(Z3[(k * D + r) * D + e2], T[e2 * D + cc], acc);
        W3[e] = -0.5 * c * Th[k] * T[r * D + cc] + 0.5 * c * acc - 0.5 * c * (T[r * D + k] * Th[cc] + Th[r] * T[cc * D + k]);
    }
    __syncthreads();
    const int NT2 = D * (D + 1) / 2;
    double* dMdm = rec;
    double* dMds = rec + D;
    double* dVdm = dMds + NT2;
    double* dVds = dVdm + nI;
    if (t < D) dMdm[t] = c * Th[t];
    for (int e = t; e < nI; e += 256) {
        const int r = e / D, cc = e - r * D;
        dVdm[e] = c * (THT[cc * D + r] - g * T[cc * D + r]);     // [k = r][row = cc]: (T H2 T)[cc][k] - g T[cc][k]
        if (r <= cc) {
            const double x = -0.5 * c * g * T[e] + 0.5 * c * THT[e], y = -0.5 * c * g * T[cc * D + r] + 0.5 * c * THT[cc * D + r];
            dMds[cc * (cc + 1) / 2 + r] = 0.5 * (x + y);
        }
    }
    for (int e = t; e < n3; e += 256) {
        const int k = e / nI, r = (e / D) % D, cc = e % D;
        if (r <= cc) dVds[(long)k * NT2 + cc * (cc + 1) / 2 + r] = 0.5 * (W3[(k * D + r) * D + cc] + W3[(k * D + cc) * D + r]);
    }
}

__global__ __launch_bounds__(256) void k_mm_jac_fin(MMModel md, MMWork wk, const double* __restrict__ part, int nrc,
                                                   const double* __restrict__ head, const double* __restrict__ mpart,
                                                   double* __restrict__ jrec, long jstride, BwdBatch bb) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int D = md.D, E = md.E, t = threadIdx.x, pl = blockIdx.x, z = blockIdx.y;
    const int nI = D * D, rec = 1 + D + nI, NT2 = D * (D + 1) / 2, recp = 1 + D + NT2;
    part += (long)z * bb.part;
    mpart += (long)z * bb.part;
    head += (long)z * bb.head;
    jrec += (long)z * jstride;
    if (pl >= wk.PL) {
        const int a = pl - wk.PL;
        jac_fin_output(md, head, a, nrc, mpart, jrec + (long)wk.PL * recp + (long)a * (D + NT2 + nI + D * NT2), sm);
        return;
    }
    double* Pm = sm;               // [D][D]
    double* lam = Pm + nI;         // [D + 2]: lambda | rdet
    double* Iv = lam + D + 2;      // [rec]  summed partials (N | A | I)
    double* PI = Iv + rec;         // [D][D]
    const double* hd = head + (long)(E + pl) * (nI + D + 2);
    for (int e = t; e < nI + D + 2; e += 256) (e < nI ? Pm[e] : lam[e - nI]) = hd[e];
    for (int e = t; e < rec; e += 256) Iv[e] = sum_strided<16>(part + (long)pl * nrc * rec + e, rec, nrc);   // fixed order
    __syncthreads();
    const double rdet = lam[D];
    const double Nab = Iv[0];
    const double* Av = Iv + 1;
    const double* Im = Iv + 1 + D;
    for (int e = t; e < nI; e += 256) {
        const int r = e / D, c = e - r * D;
        double acc = 0.0;
        for (int k = 0; k < D; ++k) acc = fma(Pm[r * D + k], Im[k * D + c], acc);
        PI[e] = acc;
    }
    __syncthreads();
    // G = rdet (P I P^T / 2 - N (P Lambda + Lambda P^T) / 4) into Pm's neighbour PI2, then packed symmetric
    double* Gf = PI + nI;          // [D][D]
    for (int e = t; e < nI; e += 256) {
        const int r = e / D, c = e - r * D;
        double acc = 0.0;
        for (int k = 0; k < D; ++k) acc = fma(PI[r * D + k], Pm[c * D + k], acc);   // (P I P^T)[r][c]
        const double pl2 = Pm[r * D + c] * lam[c] + Pm[c * D + r] * lam[r];         // P Lambda + Lambda P^T
        Gf[e] = rdet * (0.5 * acc - 0.25 * Nab * pl2);
    }
    __syncthreads();
    double* o = jrec + (long)pl * recp;
    if (t == 0) o[0] = Nab;
    for (int e = t; e < nI; e += 256) {
        const int r = e / D, c = e - r * D;
        if (r <= c) o[1 + D + c * (c + 1) / 2 + r] = 0.5 * (Gf[e] + Gf[c * D + r]);
    }
    if (t >= 256 - D) {
        const int r = t - (256 - D);
        double acc = 0.0;
        for (int c = 0; c < D; ++c) acc = fma(Pm[r * D + c], Av[c], acc);
        o[1 + r] = rdet * acc;
    }
}

size_t mm_jac_rec_size(int D, int E, int P) {
    const size_t NT2 = (size_t)D * (D + 1) / 2;
    return (size_t)P * (1 + D + NT2) + (size_t)E * (D + NT2 + (size_t)D * D + D * NT2);
}
size_t mm_jac_part_size(int D, int E, int P, int npad) {
    return (size_t)P * mm_bwd_rc(npad) * (1 + D + D * D) + (size_t)E * mm_bwd_rc(npad) * mm_jac_ns(D);
}

// the sweep kernel for this contraction width (KC = KP / 4 MFMA k-steps) and number of moment tiles
static void launch_bwd_pair(hipStream_t st, dim3 grid, size_t lds, const MMModel& md, const MMWork& wk, double* rowmom, double* cpart,
                            int njs, const double* bars, double* head, double* npart) {
    const int kc = wk.KP / 4, nmt = (md.D + 16) / 16;
    static bool lds_set = false;   // (the attribute is per function; set for every instantiation that may need > 64 KB)
#define PBL(K_, M_)                                                                                                                  \
    do {                                                                                                                             \
        if (wk.vsep) {                                                                                                               \
            if (lds > 65536) (void)hipFuncSetAttribute((const void*)k_mm_bwd_pair<K_, true, M_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            hipLaunchKernelGGL((k_mm_bwd_pair<K_, true, M_>), grid, dim3(256), lds, st, md, wk, rowmom, cpart, njs, bars, head, npart); \
        } else {                                                                                                                     \
            if (lds > 65536) (void)hipFuncSetAttribute((const void*)k_mm_bwd_pair<K_, false, M_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            hipLaunchKernelGGL((k_mm_bwd_pair<K_, false, M_>), grid, dim3(256), lds, st, md, wk, rowmom, cpart, njs, bars, head, npart); \
        }                                                                                                                            \
    } while (0)
    (void)lds_set;
    if (nmt == 1) {
        switch (kc) {
            case 1: PBL(1, 1); break;
            case 2: PBL(2, 1); break;
            case 3: PBL(3, 1); break;
            default: PBL(4, 1); break;
        }
    } else if (nmt == 2) {
        switch (kc) {
            case 5: PBL(5, 2); break;
            case 6: PBL(6, 2); break;
            case 7: PBL(7, 2); break;
            default: PBL(8, 2); break;
        }
    } else {
        PBL(9, 3);
    }
#undef PBL
}

void mm_bwd_geometry(int npad, int PL, int* njs, int* nrb);
size_t mm_jac_rowmom_size(int npad, int P) {
    int njs, nrb;
    mm_bwd_geometry(npad, P, &njs, &nrb);
    return (size_t)P * njs * 16 * npad;
}
size_t mm_jac_cpart_size(int npad, int P, int E) {
    int njs, nrb;
    mm_bwd_geometry(npad, P, &njs, &nrb);
    return (size_t)std::max(1, P - E) * nrb * npad;
}
size_t mm_jac_head_size(int D, int E, int P) { return (size_t)(E + P) * (D * D + D + 2); }
int mm_jac_nt(int npad, int P) {
    int njs, nrb;
    mm_bwd_geometry(npad, P, &njs, &nrb);
    return njs * nrb;
}

// Jacobian tape, per step (on the rollout's critical path): the reverse sweep in place of the forward pair kernel.  It
// leaves the step's row moments / column sums / head records in THIS step's buffers and N_ab as [P][mm_jac_nt][2] tile
// partials for the serial link.
void launch_mm_sweep(hipStream_t st, const MMModel& md, const MMWork& wk, double* rowmom, double* cpart, double* head,
                     double* npart) {
    const int P = wk.PL, E = md.E, D = md.D;
    int njs, nrb;
    mm_bwd_geometry(md.npad, P, &njs, &nrb);
    const int nhead = E + P, per_row = nrb * njs;
    dim3 grid(nrb, P + (nhead + per_row - 1) / per_row, njs);
    const int nI = D * D;
    const size_t lds_pair = sizeof(double) * std::max((size_t)4 * 16 * ((md.npad / 16 + njs - 1) / njs) + 2 * (BWD_CH * bwd_tp(wk.KP) + 2 * BWD_CH) + 4 * 256, (size_t)4 * nI + D);
    const double* bars = nullptr;
    launch_bwd_pair(st, grid, lds_pair, md, wk, rowmom, cpart, njs, bars, head, npart);
}

// Jacobian tape, once per rollout (off the critical path): sums, moments and records of ALL H steps in two launches --
// nothing of the forward chain waits for them, so they run at throughput instead of paying their latency per step.
// Per-step arrays: rowmom / cpart / head / part advance by the sizes above, in_m is the head of the step's tape record.
void launch_mm_jac_finish(hipStream_t st, const MMModel& md, const MMWork& wk, int H, const double* rowmom, const double* cpart,
                          const double* head, double* part, const double* tape, size_t tape_stride, double* jrec) {
    if (H <= 0) return;
    const int P = wk.PL, E = md.E, D = md.D;
    int njs, nrb;
    mm_bwd_geometry(md.npad, P, &njs, &nrb);
    const int LD = D | 1, nI = D * D, D1 = D + 1, LD1 = D1 | 1, NS = mm_jac_ns(D);
    const int nrc = mm_bwd_rc(md.npad);
    BwdBatch bb;
    bb.rowmom = (long)mm_jac_rowmom_size(md.npad, P);
    bb.cpart = (long)mm_jac_cpart_size(md.npad, P, wk.EL);   // (EL = E on one rank; the first EL local pairs are the diagonal ones)
    bb.part = (long)mm_jac_part_size(D, E, P, md.npad);
    bb.head = (long)mm_jac_head_size(D, E, P);
    bb.in_m = tape;
    bb.in_m_stride = (long)tape_stride;
    double* mpart = part + (size_t)P * nrc * (1 + D + nI);
    const size_t lds_post = sizeof(double) * std::max((size_t)3 * 64 * LD + 128 + 3 * D, (size_t)nI + 64 * LD1 + 64 + NS / 2 + 2);
    hipLaunchKernelGGL(k_mm_bwd_post<1>, dim3(P + E, nrc, H), dim3(256), lds_post, st, md, wk, rowmom, cpart, njs, nrb, part, nrc,
                       head, mpart, 1, bb);   // (the Jacobian tape serves D <= 14)
    const size_t lds_fin = sizeof(double) * std::max((size_t)4 * nI + 4 * D + 8, (size_t)3 * nI + NS + D + 2 * nI * D);
    hipLaunchKernelGGL(k_mm_jac_fin, dim3(P + E, H), dim3(256), lds_fin, st, md, wk, part, nrc, head, mpart, jrec,
                       (long)mm_jac_rec_size(D, E, P), bb);
}

void mm_bwd_geometry(int npad, int PL, int* njs, int* nrb) {
    *nrb = (npad + 64 * BWD_RT - 1) / (64 * BWD_RT);
    int q = 1;   // column splits: enough workgroups for a few balanced rounds of the chip
    while (q < 4 && (npad / 16) % (2 * q) == 0 && (long)*nrb * PL * q < 1536) q *= 2;
    if (const char* ev = getenv("PILCO_BWD_NJS")) q = std::max(1, std::min(atoi(ev), npad / 16));
    *njs = q;
}

void launch_mm_bwd(hipStream_t st, const MMModel& md, const MMWork& wk, double* rowmom, double* cpart, double* part,
                   const double* bars, double* head, double* out, unsigned* done, double* sum_out) {
    const int P = wk.PL, E = md.E, D = md.D;
    int njs, nrb;
    mm_bwd_geometry(md.npad, P, &njs, &nrb);
    const int nhead = E + P, per_row = nrb * njs;
    dim3 grid(nrb, P + (nhead + per_row - 1) / per_row, njs);   // the rows past P hold the head workgroups
    const int LD = D | 1, nI = D * D;
    const size_t lds_pair = sizeof(double) * std::max((size_t)4 * 16 * ((md.npad / 16 + njs - 1) / njs) + 2 * (BWD_CH * bwd_tp(wk.KP) + 2 * BWD_CH) + 4 * 256, (size_t)4 * nI + D);
    double* npart = nullptr;
    launch_bwd_pair(st, grid, lds_pair, md, wk, rowmom, cpart, njs, bars, head, npart);
    const int nrc = mm_bwd_rc(md.npad);
    double* mpart = part + (size_t)P * nrc * (1 + D + nI);
    const size_t lds_post = sizeof(double) * std::max((size_t)3 * 64 * LD + 128 + 3 * D, (size_t)nI + 64 * LD + 128 + D + 2);
    if (D <= 14)
        hipLaunchKernelGGL(k_mm_bwd_post<1>, dim3(P + E, nrc), dim3(256), lds_post, st, md, wk, rowmom, cpart, njs, nrb, part, nrc,
                           head, mpart, 0, BwdBatch{0, 0, 0, 0, nullptr, 0});
    else
        hipLaunchKernelGGL(k_mm_bwd_post<5>, dim3(P + E, nrc), dim3(256), lds_post, st, md, wk, rowmom, cpart, njs, nrb, part, nrc,
                           head, mpart, 0, BwdBatch{0, 0, 0, 0, nullptr, 0});
    const size_t lds_fin = sizeof(double) * ((size_t)3 * nI + 4 * D + 8);
    hipLaunchKernelGGL(k_mm_bwd_fin, dim3(P + E), dim3(256), lds_fin, st, md, wk, part, nrc, bars, head, mpart, out, done, sum_out);
}
int mm_bwd_rc(int npad) { return std::min(BWD_RC, npad / 64); }

}  // namespace pilco
