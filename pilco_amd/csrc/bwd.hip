// Reverse pass of one moment-matching step: k_mm_bwd_pair / _post / _fin.
#include "mm_device.h"
#include "rev_local.h"

namespace pilco {

// ------------------------------------------------------------------ adjoint of the pair sums
// d e_ij/d m = P (z_i + w_j) and d e_ij/d s = (P y)(P y)^T / 2, so the reverse pass needs, per
// pair, only the moments of W.L (W = beta_a beta_b^T, - iK_a on a diagonal pair) up to second order in y_ij = z_i + w_j:
//   N = sum_ij W_ij L_ij,   A = sum_ij W_ij L_ij y_ij,   I = sum_ij W_ij L_ij y_ij y_ij^T.
// With r_i = sum_j W_ij L_ij, c_j = sum_i W_ij L_ij and the row moments m_i = sum_j W_ij L_ij w_j:
//   A = sum_i r_i z_i + sum_j c_j w_j,   I = sum_i (r_i z_i z_i^T + z_i m_i^T + m_i z_i^T) + sum_j c_j w_j w_j^T.
// A wave owns 16*BWD_RT rows and sweeps a range of columns; the exponent tile is computed TRANSPOSED (column operand as
// MFMA A, row operand as B) so that the weighted tile W.L lands in the operand layout of a second MFMA that contracts it
// with [w_j | 1]: moments and row sums cost 4 MFMAs per 16x16 tile and moment tile (NMT = ceil((D + 1) / 16)) and no VALU
// reductions; the result M_i = [m_i | r_i] comes out with the ROWS along the result registers (tile as the A operand),
// which is the B-operand layout of the wave's epilogue: ONE more contraction over its rows,
//   G[d][e] = sum_i beta~_i [z_i | 1]_d [m_i + r_i z_i / 2 | r_i]_e      ((D + 1) x (D + 1), 4 BWD_RT NMT^2 MFMAs per wave),
// holds everything the row side contributes (I's row part = G + G^T on the D x D block, A's = column D, N = G[D][D]).
// The four waves' G are added in LDS and ONE (16 NMT)^2 block per workgroup goes to memory (gpart): 3.6 MB per sweep at
// C2u where rounds 1-3 wrote the row moments themselves (22 MB, read back by k_mm_bwd_post: 42 MB of the sweep's 206).
// The column sums run along the lanes of a DPP row: the wave parks its four result registers in a private LDS scratch and
// reads the previous step's back four at a time (two quad permutes finish the sum: 9 VALU ops per step instead of four
// row_shr adds per register); the four waves keep their column sums in separate LDS slices that are summed in a fixed
// order (cpart; k_mm_bwd_post contracts them with [w_j | 1] once per pair, not once per workgroup).
// Off-diagonal pairs: the row side carries beta_b only, the column side beta_a only (the epilogue / the post kernel apply
// the other factor).  DIAGONAL pairs (a == b: z == w, W and L symmetric) sweep only the tiles at or right of the diagonal:
// a tile strictly right of it stands for its mirror image too (weight 2), a tile on it counts once -- N, A and I are sums
// of a symmetric function of (i, j), so the generic row / column formulas above give the full sums from the upper half:
// half the exps and half the iK stream (84 -> 42 MB per sweep at C2u).
// gpart[pl][js][rb][NMT * NMT][256]: entry r * 64 + lane of block (m1, m2) = G[16 m1 + lane / 16 + 4 r][16 m2 + lane % 16];
// cpart[pl][row block][npad]: column sums over the rows of one workgroup.
#ifndef BWD_RT
#define BWD_RT 2
#endif
#ifndef BWD_FAIR
#define BWD_FAIR 1   // the sweep's workgroups lower their issue priority as they advance through their column range (measured:
#endif               // sweep 98.8 -> 96.9 us, three A/B repetitions in one call; rising priority: no gain; 0 = off)
constexpr int BWD_CH = 64;   // columns staged per LDS chunk (one wave-wide row segment)
constexpr int BWD_SCR_H = 40, BWD_SCR_R = 72, BWD_SCR_W = 4 * BWD_SCR_R;   // column-sum scratch of a wave (doubles): half / register strides, size
__host__ __device__ constexpr int bwd_tp(int kp) { return kp <= 16 ? 17 : kp + 1; }   // pitch of the staged column-major tile (doubles): operand rows + 1 (odd: conflict-free)
// sum_{q < n} base[q * stride] with the loads of a batch of B issued together (a plain loop serialises one global
// latency per term: these kernels are latency-bound); fixed summation order
template <int B>
__device__ __forceinline__ double sum_strided(const double* __restrict__ base, long stride, int n) {
    double acc = 0.0;
    for (int q0 = 0; q0 < n; q0 += B) {
        double v[B];
#pragma unroll
        for (int u = 0; u < B; ++u) v[u] = base[(long)min(q0 + u, n - 1) * stride];   // (unconditional request, clamped: n >= 1)
#pragma unroll
        for (int u = 0; u < B; ++u) acc += (q0 + u < n) ? v[u] : 0.0;
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
                         double* __restrict__ head, double* sm, const double* __restrict__ in_s = nullptr) {
    if (!in_s) in_s = wk.in_s;
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
        else if (h < E) v = in_s[r * D + c] + (r == c ? lam[r] : 0.0);        // s + Lambda_a^2
        else v = lam[r] * in_s[r * D + c] + (r == c ? 1.0 : 0.0);             // I + Lambda_ab s
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

// The step's D x D inverses as a launch of their own, batched over horizon steps (blockIdx.y; the input covariance from the
// step's tape record): the one-launch small step has no sweep launch whose spare workgroups could compute them.
__global__ __launch_bounds__(256) void k_mm_bwd_head(MMModel md, MMWork wk, double* __restrict__ head, long head_stride,
                                                     const double* __restrict__ in_s, long in_s_stride) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    bwd_head(md, wk, nullptr, blockIdx.x, head + (long)blockIdx.y * head_stride, sm, in_s + (long)blockIdx.y * in_s_stride);
}

// NMT: 16-row tiles of the moment product (rows d = 0..D: w_j and the ones), NMT = ceil((D + 1) / 16); KC <= 4 keeps four
// waves per SIMD, wider contractions run at two
template <int KC, bool VSEP, int NMT>
#ifndef BWD_LBW
#define BWD_LBW 2
#endif
__global__ __launch_bounds__(256, (KC <= 4 && NMT == 1) ? 4 : BWD_LBW) void k_mm_bwd_pair(MMModel md, MMWork wk, double* __restrict__ gpart,
                                                    double* __restrict__ cpart, int njs, const double* __restrict__ bars,
                                                    double* __restrict__ head, double* __restrict__ npart) {
    __shared__ double tab[FEXP_TN];
    extern __shared__ __attribute__((aligned(16))) double csl[];   // [4][jw]  (head workgroups: Gauss-Jordan scratch)
    kernarg_warm<(int)(sizeof(MMModel) + sizeof(MMWork)) + 56 + 64>();
    if ((int)blockIdx.y >= wk.PL) {   // spare workgroups: the step's D x D inverses, one per output / pair
        const int h = ((int)blockIdx.y - wk.PL) * (int)(gridDim.x * gridDim.z) + (int)(blockIdx.z * gridDim.x + blockIdx.x);
        if (h < md.E + wk.PL) bwd_head(md, wk, bars, h, head, csl);
        return;
    }
    for (int e = threadIdx.x; e < FEXP_TN; e += blockDim.x) tab[e] = wk.exp_tab[e];
    const int npad = md.npad, D = md.D, E = md.E;
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // (the wave index in scalar registers: the diagonal pairs' tile tests are scalar)
    const int lr = lane >> 4, lc = lane & 15;
    const int pl = blockIdx.y, js = blockIdx.z, rb = blockIdx.x;
    int a, b;
    local_pair_ab(wk, E, pl, a, b);
    const PairOps po = pair_ops(wk, D, npad, pl, b);   // operand layout: MMWork::At / Wt (moment.h)
    const double* beta_a = md.beta + mm_beta_row(md, a) * npad;
    const double* beta_b = md.beta + mm_beta_row(md, b) * npad;
    const bool diag = (a == b);
    const double* iKa = (diag && md.iK) ? md.iK + mm_ik_blk(md, a) * npad * npad : nullptr;
    // column range of this split: the npad / 16 column tiles dealt as evenly as they go (njs need not divide them)
    const int ct = npad / 16, jbeg = 16 * (int)((long)js * ct / njs), jend = 16 * (int)((long)(js + 1) * ct / njs);
    const int jw = jend - jbeg, jws = 16 * ((ct + njs - 1) / njs);   // jws: LDS slice stride (the widest split)
    const int rbase = rb * 64 * BWD_RT, ibase = rbase + w * 16 * BWD_RT;
    // a diagonal pair starts at the staged chunk that holds the workgroup's first row (nothing left of it is swept)
    const int jstart = diag ? min(jend, jbeg + BWD_CH * (max(0, rbase - jbeg) / BWD_CH)) : jbeg;
    if (diag)
        for (int e = threadIdx.x; e < 4 * jws; e += 256) csl[e] = 0.0;   // columns a wave skips keep a zero sum
    __syncthreads();
    double rf[BWD_RT][KC], brow[BWD_RT];
    int irow[BWD_RT];
#pragma unroll
    for (int rt = 0; rt < BWD_RT; ++rt) {
        const bool ok = ibase + 16 * rt < npad;                  // wave-uniform; rows past the padding weigh zero
        irow[rt] = ok ? ibase + 16 * rt + lc : lc;
        brow[rt] = ok ? beta_a[irow[rt]] : 0.0;
#pragma unroll
        for (int c = 0; c < KC; ++c) rf[rt][c] = po.At[(long)(4 * c + lr) * npad + irow[rt]];
    }
    // The column operands (KP rows of Bt, beta_b, v) are the same for the four waves of the workgroup: they are staged
    // once per BWD_CH columns through LDS (wave w fetches rows w, w + 4, .. as 512-byte row segments) into a
    // column-major tile T[j][BWD_TP] that serves both MFMA operand layouts -- cf (K = operand row, M = column) and its
    // transpose a2 (K = column, N = operand row) -- without bank conflicts (pitch 17 doubles); the next chunk is in
    // flight in registers while the current one is evaluated.  Only the iK stream of a diagonal pair stays a per-wave
    // buffer load.
    constexpr int KPc = 4 * KC, NR = KPc + (VSEP ? 2 : 1), NST = (NR + 3) / 4, BWD_TP = bwd_tp(KPc), SB = BWD_CH * BWD_TP + 2 * BWD_CH;
    const __amdgpu_buffer_rsrc_t rIK = buf_rsrc_uniform(iKa ? iKa + (long)jbeg * npad : po.Wt);
    const double* vsrc = (const double*)((const char*)po.Wt + po.v0);   // (staged only when VSEP: otherwise v_j is row D + 1 of the contraction)
    unsigned ik_voff[BWD_RT];   // (the 4 r part of the row index rides in the scalar offset: 2 registers instead of 8)
#pragma unroll
    for (int rt = 0; rt < BWD_RT; ++rt) ik_voff[rt] = ((unsigned)lr * (unsigned)npad + (unsigned)irow[rt]) * 8u;
    int dsel[NMT];                       // operand rows contracted by the second product: w_j (d < D), the ones (d = D);
#pragma unroll                           // lanes past that repeat row D: their result columns (d > D) are never read
    for (int m = 0; m < NMT; ++m) dsel[m] = 16 * m + lc <= D ? 16 * m + lc : D;
    double* stg = csl + 4 * jws;         // [2][SB]
    auto stage_load = [&](int jc, double (&sg)[NST]) {
#pragma unroll
        for (int k = 0; k < NST; ++k) {
            const int row = w + 4 * k, col = jc + lane;   // wave-uniform row
            const double* src = row < KPc ? colop_row<VSEP>(po, npad, row) : (row == KPc ? beta_b : vsrc);
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
    // acc[rt][m][r]: row i = 16 rt + 4 r + lane / 16 of the wave, moment column d = 16 m + lane % 16
    d4 acc[BWD_RT][NMT];
#pragma unroll
    for (int rt = 0; rt < BWD_RT; ++rt)
#pragma unroll
        for (int m = 0; m < NMT; ++m) acc[rt][m] = d4{0.0, 0.0, 0.0, 0.0};
    double* myslice = csl + w * jws;
    // column-sum scratch of this wave: the partial of (result register r, DPP row lr, lane lc of the row) sits at
    // r BWD_SCR_R + (lc / 8) BWD_SCR_H + 8 lr + lc % 8; reader lane = (column jj = lane / 4 = 4 r + lr, quarter q = lane % 4)
    // takes the partials of lanes lc = 2 q, 2 q + 1, 8 + 2 q, 9 + 2 q as two 16-byte reads.  The strides put the four DPP
    // rows of one 16-lane read group on the four quarters of the bank row and the two 8-lane runs of a store group on
    // different halves of the 32 store banks: no conflicts either way (the plain [r][lr][lc] layout read back two-way: lanes
    // l and l + 8 of a group on the same banks -- SQ_LDS_BANK_CONFLICT was twice the useful LDS cycles, profiles/r04_grad_pmc_summary.json)
    double* scr = csl + 4 * jws + 2 * SB + w * BWD_SCR_W;
    const int rd_jj = lane >> 2, rd_q = lane & 3;
    const double* rd_src = scr + (rd_jj >> 2) * BWD_SCR_R + (rd_jj & 3) * 8 + 2 * rd_q;
    double* wr_dst = scr + (lc >> 3) * BWD_SCR_H + lr * 8 + (lc & 7);
    int jprev = -1;
    auto flush_cols = [&](int jp) {
        double p = (rd_src[0] + rd_src[1]) + (rd_src[BWD_SCR_H] + rd_src[BWD_SCR_H + 1]);
        p = dpp_add<0xB1, 0xf>(p);   // quad_perm [1,0,3,2]
        p = dpp_add<0x4E, 0xf>(p);   // quad_perm [2,3,0,1]: every lane of the quad holds the sum over the 16 rows
        if (rd_q == 0) myslice[jp - jbeg + rd_jj] = p;
    };
    // the column sweep, specialised at compile time (branches inside the loop would fence the scheduler between the
    // eight exp evaluations of a step): MODE 0 off-diagonal pair, 1 diagonal pair with the iK stream,
    // 2 diagonal pair without it (RBF policy GP)
    auto sweep = [&](auto mode_c) {
        constexpr int MODE = decltype(mode_c)::value;
        if (jstart >= jend) return;   // (workgroup-uniform) a diagonal pair's workgroup entirely left of the diagonal
        // the iK tiles of a diagonal pair are requested ONE column step ahead, into registers of their own (a load consumed in the
        // step that issues it lands in registers the next MFMA chain wants: the wave then waits for the L2 round trip first)
        double ikn[BWD_RT][4];
        auto ik_request = [&](int j0n) {
#pragma unroll
            for (int rt = 0; rt < BWD_RT; ++rt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    ikn[rt][r] = buf_ld(rIK, ik_voff[rt], ((unsigned)(j0n - jbeg) + 4u * (unsigned)r) * (unsigned)npad * 8u);
        };
        // (a wave whose rows lie past the padding -- npad not a multiple of the 128-row block -- sweeps nothing, and the tile it
        // would ask for starts below the last row of iK: the last output's block ends there.  Found in round 5 by the
        // contraction-depth test: a memory fault at npad = 192 with seven outputs, latent since the symmetric sweep of round 4.)
        if (MODE == 1 && ibase < npad) ik_request(jstart < ibase ? jstart + 16 * ((ibase - jstart) / 16) : jstart);   // the wave's first swept tile
        double sg[NST];
        stage_load(jstart, sg);
        stage_store(stg, sg);
        __syncthreads();
        int cur = 0;
        for (int jc = jstart; jc < jend; jc += BWD_CH) {
#if BWD_FAIR
            {   // issue priority by progress through the column range (see WaveProgress in pair_device.h): the workgroup's four waves
                // step together (one barrier per chunk), so this is the workgroup's priority against the others on its SIMDs.
                // (Quartiles by comparison with scalar thresholds: a signed division here -- the compiler's float-reciprocal
                // sequence -- made the K = 8 instantiation compute wrong sums, found by the seeded shape sweep of the GPU tests.)
                const int done4 = 4 * (jc - jstart), len = jend - jstart;
#if BWD_FAIR == 2   // (the variant of the advisor's question: the signed division that once went with wrong sums at K = 8)
                const int q = done4 / len;
#else
                const int q = (done4 >= len ? 1 : 0) + (done4 >= 2 * len ? 1 : 0) + (done4 >= 3 * len ? 1 : 0);
#endif
                if (q == 0) __builtin_amdgcn_s_setprio(3);
                else if (q == 1) __builtin_amdgcn_s_setprio(2);
                else if (q == 2) __builtin_amdgcn_s_setprio(1);
                else __builtin_amdgcn_s_setprio(0);
            }
#endif
            const bool more = jc + BWD_CH < jend;
            if (more) stage_load(jc + BWD_CH, sg);
            const double* Tb = stg + cur * SB;
            const double* bS = Tb + BWD_CH * BWD_TP;
            const double* vS = bS + BWD_CH;
            const int nst = min(BWD_CH, jend - jc);
            for (int jl = 0; jl < nst; jl += 16) {
                const int j0 = jc + jl;
                if (MODE != 0 && j0 < ibase) continue;   // (wave-uniform) both row tiles lie below this column tile's mirror
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
                if (MODE == 0) {
    #pragma unroll
                    for (int r = 0; r < 4; ++r)
    #pragma unroll
                        for (int m = 0; m < NMT; ++m) a2[m][r] *= bcol[r];
                }
                double csum[4] = {0.0, 0.0, 0.0, 0.0};
    #pragma unroll
                for (int rt = 0; rt < BWD_RT; ++rt) {
                    // diagonal pair: tile weight 2 right of the diagonal tile (it stands for its mirror image too), 1 on it, 0 left of it
                    const int i0 = ibase + 16 * rt;
                    const double om = j0 > i0 ? 2.0 : (j0 == i0 ? 1.0 : 0.0);
                    // VSEP: v_j, which runs along the result registers of the transposed tile, is the chain's initial accumulator
                    // (the first MFMA reads it as its C operand: no add per exponent afterwards)
                    d4 e = {VSEP ? vj[0] : 0.0, VSEP ? vj[1] : 0.0, VSEP ? vj[2] : 0.0, VSEP ? vj[3] : 0.0};
    #pragma unroll
                    for (int c = 0; c < KC; ++c)
                        e = __builtin_amdgcn_mfma_f64_16x16x4f64(cf[c], rf[rt][c], e, 0, 0, 0);   // e[r]: i = irow, j = j0+lr+4r
                    MFMA_KEEP_ALIVE(cf[0]);      // (not VSEP: the first MFMA of the chain has a constant-zero accumulator)
                    MFMA_KEEP_ALIVE(rf[rt][0]);
                    // this hipcc does not reliably leave the wait states between the chain and the first VALU read of its result
                    // (since v_j became the chain's accumulator the exp's clamp reads it directly, in every instantiation);
                    // tests/test_build_isa.py scans the generated code of ALL instantiations for reads that come too early
                    MFMA_RESULT_FENCE(e);
                    double wl[4];
    #pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (MODE == 0) {
                            // off-diagonal pair: W = beta_a beta_b^T is separable -- the row side carries beta_b,j only and
                            // the column side beta_a,i only (one multiply and one FMA instead of two multiplies and an
                            // add); the epilogue / k_mm_bwd_post apply the missing factor per row / per column
                            // (beta_b,j rides in the moment product's column operand a2, scaled once per column step)
                            const double l = fexp_to_mfma(e[r], tab);   // (straight into the moment product's A operand: mm_device.h)
                            wl[r] = l;
                            csum[r] = fma(brow[rt], l, csum[r]);
                        } else {
                            double wgt = brow[rt] * bcol[r];
                            if (MODE == 1) wgt -= ikn[rt][r];   // iK symmetric: coalesced along the rows (requested a step ago)
                            wl[r] = (wgt * om) * fexp(e[r], tab);
                            csum[r] += wl[r];
                        }
                    }
    #pragma unroll
                    for (int m = 0; m < NMT; ++m)
    #pragma unroll
                        for (int r = 0; r < 4; ++r) acc[rt][m] = __builtin_amdgcn_mfma_f64_16x16x4f64(wl[r], a2[m][r], acc[rt][m], 0, 0, 0);
                }
                if (MODE == 1 && j0 + 16 < jend) ik_request(j0 + 16);   // (every use of this step's tile is behind us)
                // column sums over the wave's 16 lanes of a DPP row, through a per-wave LDS scratch instead of four
                // DPP row shifts per register (12 VALU ops each on the pipe this kernel is bound by): the partial sums
                // of the PREVIOUS column step are read back four at a time, added and finished with two quad
                // permutes (9 VALU ops per step instead of 48); LDS operations of one wave execute in order
                if (jprev >= 0) flush_cols(jprev);
    #pragma unroll
                for (int r = 0; r < 4; ++r) wr_dst[r * BWD_SCR_R] = csum[r];
                jprev = j0;
            }
            if (more) stage_store(stg + (cur ^ 1) * SB, sg);
            __syncthreads();
            cur ^= 1;
        }
        if (jprev >= 0) flush_cols(jprev);
    };
    if (!diag) sweep(std::integral_constant<int, 0>{});
    else if (iKa) sweep(std::integral_constant<int, 1>{});
    else sweep(std::integral_constant<int, 2>{});
    // ---- epilogue 1: the wave's rows contracted with [z_i | 1]   (header comment: G)
    // zt[m][r]: lane (d = 16 m + lane % 16, i = 4 r + lane / 16) -- the A-operand layout, and the layout acc holds M_i in
    double zm[NMT], zi[NMT];
#pragma unroll
    for (int m = 0; m < NMT; ++m) {
        const int d = 16 * m + lc;
        const double la = d < D ? md.ls[a * D + d] : 1.0;
        zm[m] = d < D ? wk.in_m[d] : 0.0;
        zi[m] = 1.0 / (la * la);
    }
    d4 G[NMT][NMT];
#pragma unroll
    for (int m1 = 0; m1 < NMT; ++m1)
#pragma unroll
        for (int m2 = 0; m2 < NMT; ++m2) G[m1][m2] = d4{0.0, 0.0, 0.0, 0.0};
    const int rlane = (lane & 48) | (D & 15);   // the lane of this DPP row that holds r_i (moment column D)
#pragma unroll
    for (int rt = 0; rt < BWD_RT; ++rt) {
        if (ibase + 16 * rt >= npad) continue;   // (wave-uniform) tiles past the padding carry garbage
        double zt[NMT][4], bs[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = ibase + 16 * rt + 4 * r + lr;
            const bool valid = i < md.n;
            bs[r] = !valid ? 0.0 : (diag ? 1.0 : beta_a[i]);
#pragma unroll
            for (int m = 0; m < NMT; ++m) {
                const int d = 16 * m + lc;
                zt[m][r] = (valid && d < D) ? (md.Pt[(long)d * npad + i] - zm[m]) * zi[m] : ((valid && d == D) ? 1.0 : 0.0);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            double ri = 0.0;
#pragma unroll
            for (int m = 0; m < NMT; ++m)
                if (m == (D >> 4)) ri = __shfl(acc[rt][m][r], rlane);
            double mp[NMT];
#pragma unroll
            for (int m = 0; m < NMT; ++m) {
                const int d = 16 * m + lc;
                const double x = acc[rt][m][r];
                mp[m] = bs[r] == 0.0 ? 0.0 : (d < D ? fma(0.5 * ri, zt[m][r], x) : x);   // rows past n: nothing (their sums may be anything)
            }
#pragma unroll
            for (int m1 = 0; m1 < NMT; ++m1)
#pragma unroll
                for (int m2 = 0; m2 < NMT; ++m2) {
                    const double za = zt[m1][r] * bs[r];
                    G[m1][m2] = __builtin_amdgcn_mfma_f64_16x16x4f64(za, mp[m2], G[m1][m2], 0, 0, 0);
                    MFMA_KEEP_ALIVE(za);         // (the first MFMA of each chain has a constant-zero accumulator)
                    MFMA_KEEP_ALIVE(mp[m2]);
                }
        }
    }
    // ---- epilogue 2: column sums of the workgroup's rows (fixed order over the four waves)
    __syncthreads();
    {
        double* cp = cpart + ((long)pl * gridDim.x + rb) * npad + jbeg;
        for (int jj = threadIdx.x; jj < jw; jj += 256)
            cp[jj] = (csl[jj] + csl[jws + jj]) + (csl[2 * jws + jj] + csl[3 * jws + jj]);
    }
    // ---- epilogue 3: the four waves' G added in LDS (block by block: the staging area holds 4 x 256 doubles), one block
    // per workgroup to memory; N_ab's share for the serial link of a value-and-gradient rollout is entry (D, D)
    double* red = stg;   // [4][256]  (2 SB + 4 * BWD_SCR_W >= 1024 doubles)
    double* gout = gpart + (((long)pl * njs + js) * gridDim.x + rb) * (NMT * NMT * 256);
    const int tN = ((D & 15) >> 2) * 64 + (D & 3) * 16 + (D & 15);   // entry of (d, e) = (D, D) inside its block
#pragma unroll
    for (int m1 = 0; m1 < NMT; ++m1)
#pragma unroll
        for (int m2 = 0; m2 < NMT; ++m2) {
            __syncthreads();
            MFMA_RESULT_FENCE(G[m1][m2]);
#pragma unroll
            for (int r = 0; r < 4; ++r) red[w * 256 + r * 64 + lane] = G[m1][m2][r];
            __syncthreads();
            const int t = threadIdx.x;
            const double v = (red[t] + red[256 + t]) + (red[512 + t] + red[768 + t]);
            gout[(m1 * NMT + m2) * 256 + t] = v;
            if (npart && m1 == (D >> 4) && m2 == (D >> 4) && t == tN) {
                // Jacobian tape: this workgroup's share of N_ab, in the [pair][tile][2] layout the serial link packs tile
                // partials from (tile = js * row blocks + rb)
                double* o = npart + ((long)pl * (njs * (int)gridDim.x) + js * (int)gridDim.x + rb) * 2;
                o[0] = v;
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
// The moments on the matrix cores (round 6; the fused finish k_mm_jac_rec).  The VALU form of rounds 3-5 read FOUR LDS words per
// multiply-add -- 9.3 M wave-wide LDS reads per rollout at C2u, 60 us of the LDS pipes of the whole chip and 145 us in practice.
// As a product:  C[(d, e)][f] = sum_i (l_i zeta~_d zeta~_e) zeta~_f,  rows = the D1 (D1 + 1) / 2 pairs d >= e (16 per tile),
// columns f < 16, K = the points (4 per v_mfma_f64_16x16x4_f64): 4 LDS reads feed 1024 multiply-adds; entries f <= e are the
// moments (the others are computed for nothing).  l_i itself: the quadratic form's rows dealt over the four waves (256
// threads per 64-point block instead of 64).
constexpr int JAC_MT = 8;   // row tiles: D1 (D1 + 1) / 2 <= 128 pairs (D <= 14)
__device__ void bwd_mean_moments_mfma(const MMModel& md, const double* __restrict__ in_m, const double* __restrict__ head, int a, int rc,
                                      int nrc, double* __restrict__ mpart, double* sm) {
    const int D = md.D, D1 = D + 1, npad = md.npad, t = threadIdx.x, lane = t & 63, w = t >> 6, lr = lane >> 4, lc = lane & 15;
    const int nI = D * D, LD = 17, NS = D1 * (D1 + 1) * (D1 + 2) / 6, NPAIR = D1 * (D1 + 1) / 2, ntile = (NPAIR + 15) / 16;
    double* T = sm;                 // [D][16]    rows zero-padded: the quadratic form runs over 16 columns without a condition
    double* zs = T + 16 * 16;       // [64][17]   zeta (zeros past D) | 1 at column D   (odd row length: conflict-free column reads)
    double* lv = zs + 64 * LD;      // [64]
    double* qp = lv + 64;           // [4][64]  partial quadratic forms; later [4][256] for the waves' tiles
    int* ptab = (int*)(qp + 4 * 256);   // [16 JAC_MT] pair row -> d | e << 8, or -1
    const double* hd = head + (long)a * (nI + D + 2);
    {
        const int r = t >> 4, c = t & 15;   // 256 threads: one entry of the padded [16][16] each
        T[t] = (r < D && c < D) ? hd[r * D + c] : 0.0;
    }
    if (t < 16 * JAC_MT) {
        int d = 0;
        while ((d + 1) * (d + 2) / 2 <= t) ++d;
        ptab[t] = t < NPAIR ? (d | ((t - d * (d + 1) / 2) << 8)) : -1;
    }
    __syncthreads();
    // this lane's pair row in every tile: (d, e), d >= e, or -1
    int pd[JAC_MT], pe[JAC_MT];
#pragma unroll
    for (int m = 0; m < JAC_MT; ++m) {
        const int q = ptab[16 * m + lc];
        pd[m] = q < 0 ? -1 : (q & 255);
        pe[m] = q < 0 ? 0 : (q >> 8);
    }
    d4 C[JAC_MT];
#pragma unroll
    for (int m = 0; m < JAC_MT; ++m) C[m] = d4{0.0, 0.0, 0.0, 0.0};
    __syncthreads();
    // (requests are UNCONDITIONAL, from clamped addresses -- the padding of Pt / beta is there to be read.  Requesting the NEXT
    // block's coordinates a block ahead was tried: no faster, 32 more registers.  No run-time index into the register copy of
    // the point either: `pv[r]` with the wave's row r compiled into a ladder of 14 branches per row -- 8 000 cycles per block)
    const double* beta_a = md.beta + mm_beta_row(md, a) * npad;
    double mloc[16];
#pragma unroll
    for (int d = 0; d < 16; ++d) mloc[d] = in_m[min(d, D - 1)];
    for (int blk = rc; blk < npad / 64; blk += nrc) {
        // ---- l_i of the block's 64 points: thread (point lane, wave w) takes the rows r = w, w + 4, .. of zeta^T T zeta
        const int i = blk * 64 + lane;
        double pv[16];
#pragma unroll
        for (int d = 0; d < 16; ++d) pv[d] = md.Pt[(unsigned)(min(d, D - 1) * npad + i)];
        const double bcur = beta_a[(unsigned)i];
#pragma unroll
        for (int d = 0; d < 16; ++d) pv[d] = (d < D && i < md.n) ? pv[d] - mloc[d] : 0.0;
        if (w == 0) {
#pragma unroll
            for (int d = 0; d < 16; ++d) zs[lane * LD + d] = (d == D) ? 1.0 : pv[d];   // (d == D: a select, not a branch)
        }
        __syncthreads();
        double part = 0.0;
        for (int r = w; r < D; r += 4) {
            double tz = 0.0;
#pragma unroll
            for (int c = 0; c < 16; ++c) tz = fma(T[r * 16 + c], pv[c], tz);
            part = fma(zs[lane * LD + r], tz, part);
        }
        qp[w * 64 + lane] = part;
        __syncthreads();
        if (w == 0) {
            const double quad = (qp[lane] + qp[64 + lane]) + (qp[128 + lane] + qp[192 + lane]);
            lv[lane] = (i < md.n) ? exp(-0.5 * quad) * bcur : 0.0;
        }
        __syncthreads();
        // ---- the block's 16 k-steps of 4 points dealt over the waves
        for (int ks = w; ks < 16; ks += 4) {
            const int ii = 4 * ks + lr;
            const double li = lv[ii];
            const double bf = (lc < D1) ? zs[ii * LD + lc] : 0.0;
#pragma unroll
            for (int m = 0; m < JAC_MT; ++m)
                if (m < ntile) {
                    const double av = (pd[m] >= 0) ? li * zs[ii * LD + pd[m]] * zs[ii * LD + pe[m]] : 0.0;
                    C[m] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bf, C[m], 0, 0, 0);
                    MFMA_KEEP_ALIVE(av);          // (the first MFMA of each chain has a constant-zero accumulator)
                    MFMA_KEEP_ALIVE(bf);
                }
        }
        __syncthreads();   // (zs / lv / qp are rewritten by the next block)
    }
    // ---- the four waves' tiles added in a fixed order; entry (pair (d, e), f <= e) -> mpart[tri3(d, e, f)]
    double* red = qp;   // [4][256]
#pragma unroll
    for (int m = 0; m < JAC_MT; ++m)
        if (m < ntile) {
            __syncthreads();
            MFMA_RESULT_FENCE(C[m]);
#pragma unroll
            for (int r = 0; r < 4; ++r) red[w * 256 + r * 64 + lane] = C[m][r];
            __syncthreads();
            const double v = (red[t] + red[256 + t]) + (red[512 + t] + red[768 + t]);
            const int row = ((t >> 4) & 3) + 4 * (t >> 6), f = t & 15, q = ptab[16 * m + row];
            if (q >= 0) {
                const int d = q & 255, e = q >> 8;
                if (f <= e) mpart[((long)a * nrc + rc) * NS + tri3(d, e, f)] = v;
            }
        }
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

// Per unordered pair and column chunk rc: a share of  N_ab | A (D) | I (D x D)  (header comment of the sweep):
// the share of the sweep's G blocks it adds up (blocks rc, rc + nrc, .. of the pair's njs * nrb, fixed order) and the
// column side of its 64-column blocks (rc, rc + nrc, ..), contracted on the matrix cores:
//   Gc[d][e] = sum_j c_j [w_j | 1]_d [w_j | 1]_e,   c_j = (beta_b,j) * sum over the row blocks of cpart   (K = columns).
//   N = G[D][D],  A_d = G[d][D] + Gc[d][D],  I_de = G[d][e] + G[e][d] + (Gc[d][e] + Gc[e][d]) / 2.      part[pl][chunk][1 + D + D*D]
#ifdef JAC_STAMPS   // developer build: phase stamps of pair workgroup (0, step 0) of the fused finish (tools/jac_phases.py)
#define JAC_STAMP(wk_, slot_, cond_) DBG_STAMP(wk_, slot_, cond_)
#else
#define JAC_STAMP(wk_, slot_, cond_) do { } while (0)
#endif
constexpr int BWD_RC = 8;   // chunks per pair / output (16: 12 % slower in the batched form, more workgroup prologues)
// Batched form (Jacobian tape): blockIdx.z = horizon step; every per-step array advances by its stride and the input
// mean comes from the step's tape record (wk.in_m holds the LAST step's by then).  Unbatched: strides 0, in_m = nullptr.
struct BwdBatch {
    long gpart, cpart, part, head;
    const double* in_m;
    long in_m_stride;
};
// cjl != nullptr (the fused finish: ONE workgroup per pair and step, nrc = 1): LDS for npad column coefficients.  The workgroup
// then works for memory-level parallelism instead of per-block round trips -- every column's nrb sums and beta requested in
// one coalesced burst (a thread per column) beside the G blocks' 32 requests, and the MFMA loop's points four blocks at a
// time: ~8 us per workgroup where the block-by-block form needs ~50 (2 us of memory latency per 64-column block).
template <int NMT>
__device__ void bwd_pair_post(const MMModel& md, const MMWork& wk, const double* __restrict__ in_m, const double* __restrict__ gpart,
                              const double* __restrict__ cpart, int njs, int nrb, double* o /* [1 + D + D*D]: global, or LDS behind sm's 6 * 256 (NMT = 1) */,
                              int nrc, int pl, int rc, double* sm, double* cjl = nullptr) {
    const int npad = md.npad, D = md.D, E = md.E, t = threadIdx.x;
    const int lane = t & 63, w = t >> 6, lr = lane >> 4, lc = lane & 15;
    constexpr int NB2 = NMT * NMT, GW = 16 * NMT;
    int a, b;
    local_pair_ab(wk, E, pl, a, b);
    const bool diag = (a == b);
    double* Gs = sm;                 // [GW][GW]  this chunk's share of G
    double* Gc = Gs + GW * GW;       // [GW][GW]  ... and of the column side
    double* red = Gc + GW * GW;      // [4][256]
    if (njs == 0) {
        // one-launch small step (prep_device.h: small_sweep): every chunk-workgroup of the pair left TWO blocks, G and Gc (the
        // column side is already contracted); nrb = chunks per pair.  Chunk rc of nrc adds its share, as below.
        const double* g0s = gpart + (long)pl * nrb * 512;
        if (t < 256) {
            const int cnt = (nrb - rc + nrc - 1) / nrc;
            const int d = ((t >> 4) & 3) + 4 * (t >> 6), e = t & 15;
            Gs[d * GW + e] = rc < nrb ? sum_strided<4>(g0s + (long)rc * 512 + t, (long)nrc * 512, cnt) : 0.0;
            Gc[d * GW + e] = rc < nrb ? sum_strided<4>(g0s + (long)rc * 512 + 256 + t, (long)nrc * 512, cnt) : 0.0;
        }
        __syncthreads();
        const int nI2 = D * D;
        double* o2 = o;
        for (int e2 = t; e2 < 1 + D + nI2; e2 += 256) {
            double v;
            if (e2 == 0) {
                v = Gs[D * GW + D];
            } else if (e2 <= D) {
                v = Gs[(e2 - 1) * GW + D] + Gc[(e2 - 1) * GW + D];
            } else {
                const int d = (e2 - 1 - D) / D, e = (e2 - 1 - D) - d * D;
                v = (Gs[d * GW + e] + Gs[e * GW + d]) + 0.5 * (Gc[d * GW + e] + Gc[e * GW + d]);
            }
            o2[e2] = v;
        }
        return;
    }
    // (i) the sweep's blocks
    const int nparts = njs * nrb;
    const double* g0 = gpart + (long)pl * nparts * (NB2 * 256);
    const double* cp = cpart + (long)pl * nrb * npad;
    const double* beta_b = md.beta + mm_beta_row(md, b) * npad;
    // per-lane scale / shift of the column operand (requested first: everything below is independent of it)
    double wm[NMT], wi[NMT];
#pragma unroll
    for (int m = 0; m < NMT; ++m) {
        const int d = 16 * m + lc;
        const double lbr = md.ls[b * D + min(d, D - 1)], imr = in_m[min(d, D - 1)];
        const double lb = d < D ? lbr : 1.0;
        wm[m] = d < D ? imr : 0.0;
        wi[m] = 1.0 / (lb * lb);
    }
    if (cjl && NMT == 1) {
        // ONE memory round trip for the sweep's G blocks (up to 32 per thread) AND the column coefficients (nrb sums + beta for
        // four columns per thread): a workgroup is a chain of memory latencies -- with the two bursts one behind the other,
        // then four bursts of point coordinates, it was eight round trips of ~3 us under load (36 us per workgroup at three per CU)
        double gq[32];
        const int cntg = (nparts - rc + nrc - 1) / nrc;
        const double* gb = g0 + (long)rc * 256 + t;
#pragma unroll
        for (int u = 0; u < 32; ++u) gq[u] = gb[(unsigned)(min(u, cntg - 1) * nrc * 256)];   // (32-bit indices: base in SGPRs, one VALU op per address)
        for (int j0 = 0; j0 < npad; j0 += 4 * 256) {   // column coefficients c_j = (sum over the row blocks of cpart) * beta_b,j
            // (unconditional requests from clamped addresses, see bwd_mean_moments_mfma; the first eight row blocks of all four
            // columns and beta requested TOGETHER, behind the G blocks: with a wait per column the phase was five round trips,
            // 13 of the workgroup's 31 us -- tools/jac_phases.py)
            double cq[4][8], bj[4], cs[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int jc = min(j0 + q * 256 + t, npad - 1);
#pragma unroll
                for (int u = 0; u < 8; ++u) cq[q][u] = cp[(unsigned)(min(u, nrb - 1) * npad + jc)];
                bj[q] = beta_b[(unsigned)jc];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                cs[q] = 0.0;
#pragma unroll
                for (int u = 0; u < 8; ++u) cs[q] += (u < nrb) ? cq[q][u] : 0.0;
            }
            for (int k0 = 8; k0 < nrb; k0 += 8)   // (more than eight row blocks: N > 1024)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int jc = min(j0 + q * 256 + t, npad - 1);
                    double c8[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) c8[u] = cp[(unsigned)(min(k0 + u, nrb - 1) * npad + jc)];
#pragma unroll
                    for (int u = 0; u < 8; ++u) cs[q] += (k0 + u < nrb) ? c8[u] : 0.0;
                }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int j = j0 + q * 256 + t;
                if (j < npad) cjl[j] = (j < md.n) ? cs[q] * (diag ? 1.0 : bj[q]) : 0.0;
            }
        }
        double v = 0.0;
#pragma unroll
        for (int u = 0; u < 32; ++u) v += (u < cntg) ? gq[u] : 0.0;          // fixed order
        for (int u = 32; u < cntg; ++u) v += gb[(long)u * nrc * 256];        // (more column splits than the burst holds)
        Gs[(((t >> 4) & 3) + 4 * (t >> 6)) * GW + (t & 15)] = v;
    } else {
#pragma unroll
        for (int blk = 0; blk < NB2; ++blk) {
            const double v = sum_strided<4>(g0 + (long)rc * (NB2 * 256) + blk * 256 + t, (long)nrc * (NB2 * 256), (nparts - rc + nrc - 1) / nrc);
            const int d = 16 * (blk / NMT) + ((t >> 4) & 3) + 4 * (t >> 6), e = 16 * (blk % NMT) + (t & 15);
            Gs[d * GW + e] = v;
        }
    }
    // (ii) the column side
    d4 C[NMT][NMT];
#pragma unroll
    for (int m1 = 0; m1 < NMT; ++m1)
#pragma unroll
        for (int m2 = 0; m2 < NMT; ++m2) C[m1][m2] = d4{0.0, 0.0, 0.0, 0.0};
    if (cjl) {
        JAC_STAMP(wk, 49, pl == 0 && blockIdx.y == 0 && t == 0);
        __syncthreads();   // (cjl complete)
        JAC_STAMP(wk, 50, pl == 0 && blockIdx.y == 0 && t == 0);
        constexpr int UB = NMT == 1 ? 8 : 1;   // 64-column blocks whose points are requested together
        // Lane (lr, lc) takes the FOUR CONSECUTIVE columns 16 w + 4 lr + r of a block for coordinate lc (the order of the
        // contraction index is free): 32 contiguous bytes per lane, whole cache lines per quad of lanes.  With column
        // 16 w + 4 r + lr -- 8 bytes per lane, sixteen lines a quarter used per request -- this loop was 15 of the workgroup's
        // 31 us (tools/jac_phases.py).  No selects: a padded column's coefficient is 0 in cjl, a padded block's is zeroed here,
        // and the operand row is one fma (scale 0 for lanes past D, offset 1 on lane D).
        double wsc[NMT], wof[NMT];
#pragma unroll
        for (int m = 0; m < NMT; ++m) {
            const int d = 16 * m + lc;
            wsc[m] = d < D ? wi[m] : 0.0;
            wof[m] = d < D ? -wm[m] * wi[m] : (d == D ? 1.0 : 0.0);
        }
        const int nblk = npad / 64;
        for (int blk0 = rc; blk0 < nblk; blk0 += nrc * UB) {
            double wt[UB][NMT][4], cj[UB][4];
#pragma unroll
            for (int ub = 0; ub < UB; ++ub) {
                const int blk = blk0 + ub * nrc, jb = min(blk, nblk - 1) * 64 + 16 * w + 4 * lr;
                const double live = blk < nblk ? 1.0 : 0.0;
                const double2 c01 = *reinterpret_cast<const double2*>(cjl + jb), c23 = *reinterpret_cast<const double2*>(cjl + jb + 2);
                cj[ub][0] = c01.x * live;
                cj[ub][1] = c01.y * live;
                cj[ub][2] = c23.x * live;
                cj[ub][3] = c23.y * live;
#pragma unroll
                for (int m = 0; m < NMT; ++m) {
                    const int d = 16 * m + lc;
                    const double2* pp = reinterpret_cast<const double2*>(md.Pt + (unsigned)(min(d, D - 1) * npad + jb));   // (npad and jb are multiples of 4: 16-byte aligned)
                    const double2 p01 = pp[0], p23 = pp[1];
                    wt[ub][m][0] = fma(p01.x, wsc[m], wof[m]);
                    wt[ub][m][1] = fma(p01.y, wsc[m], wof[m]);
                    wt[ub][m][2] = fma(p23.x, wsc[m], wof[m]);
                    wt[ub][m][3] = fma(p23.y, wsc[m], wof[m]);
                }
            }
#pragma unroll
            for (int ub = 0; ub < UB; ++ub)
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int m1 = 0; m1 < NMT; ++m1)
#pragma unroll
                        for (int m2 = 0; m2 < NMT; ++m2) {
                            const double ca = cj[ub][r] * wt[ub][m1][r];
                            C[m1][m2] = __builtin_amdgcn_mfma_f64_16x16x4f64(ca, wt[ub][m2][r], C[m1][m2], 0, 0, 0);
                            MFMA_KEEP_ALIVE(ca);
                            MFMA_KEEP_ALIVE(wt[ub][m2][r]);
                        }
        }
    } else
    for (int blk = rc; blk < npad / 64; blk += nrc) {
        double wt[NMT][4], cj[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int j = blk * 64 + 16 * w + 4 * r + lr;
            const bool valid = j < md.n;
            cj[r] = valid ? sum_strided<8>(cp + j, npad, nrb) * (diag ? 1.0 : beta_b[j]) : 0.0;   // the sweep left beta_b,j out of the column side
#pragma unroll
            for (int m = 0; m < NMT; ++m) {
                const int d = 16 * m + lc;
                wt[m][r] = (valid && d < D) ? (md.Pt[(long)d * npad + j] - wm[m]) * wi[m] : ((valid && d == D) ? 1.0 : 0.0);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int m1 = 0; m1 < NMT; ++m1)
#pragma unroll
                for (int m2 = 0; m2 < NMT; ++m2) {
                    const double ca = cj[r] * wt[m1][r];
                    C[m1][m2] = __builtin_amdgcn_mfma_f64_16x16x4f64(ca, wt[m2][r], C[m1][m2], 0, 0, 0);
                    MFMA_KEEP_ALIVE(ca);          // (the first MFMA of each chain has a constant-zero accumulator)
                    MFMA_KEEP_ALIVE(wt[m2][r]);
                }
    }
    JAC_STAMP(wk, 51, cjl && pl == 0 && blockIdx.y == 0 && t == 0);
#pragma unroll
    for (int m1 = 0; m1 < NMT; ++m1)
#pragma unroll
        for (int m2 = 0; m2 < NMT; ++m2) {
            __syncthreads();
            MFMA_RESULT_FENCE(C[m1][m2]);
#pragma unroll
            for (int r = 0; r < 4; ++r) red[w * 256 + r * 64 + lane] = C[m1][m2][r];
            __syncthreads();
            const int d = 16 * m1 + ((t >> 4) & 3) + 4 * (t >> 6), e = 16 * m2 + (t & 15);
            Gc[d * GW + e] = (red[t] + red[256 + t]) + (red[512 + t] + red[768 + t]);
        }
    __syncthreads();
    // (iii) N | A | I
    const int nI = D * D;
    for (int e2 = t; e2 < 1 + D + nI; e2 += 256) {
        double v;
        if (e2 == 0) {
            v = Gs[D * GW + D];
        } else if (e2 <= D) {
            const int d = e2 - 1;
            v = Gs[d * GW + D] + Gc[d * GW + D];
        } else {
            const int d = (e2 - 1 - D) / D, e = (e2 - 1 - D) - d * D;
            v = (Gs[d * GW + e] + Gs[e * GW + d]) + 0.5 * (Gc[d * GW + e] + Gc[e * GW + d]);
        }
        o[e2] = v;
    }
}

template <int NA, int NMT>   // NA = 1: D <= 14 (one mean-part sum per thread), NA = 5: D <= 32; NMT = ceil((D + 1) / 16)
#ifndef POST_LB
#define POST_LB 8   // waves per SIMD of the narrow post kernel: latency-bound, 8 resident workgroups per CU
#endif
__global__ __launch_bounds__(256, NA == 1 ? POST_LB : 1) void k_mm_bwd_post(MMModel md, MMWork wk, const double* __restrict__ gpart,
                                                    const double* __restrict__ cpart, int njs, int nrb,
                                                    double* __restrict__ part, int nrc,
                                                    const double* __restrict__ head, double* __restrict__ mpart, BwdBatch bb) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int pl = blockIdx.x, rc = blockIdx.y, z = blockIdx.z, D0 = md.D;
    gpart += (long)z * bb.gpart;
    cpart += (long)z * bb.cpart;
    part += (long)z * bb.part;
    mpart += (long)z * bb.part;
    head += (long)z * bb.head;
    const double* in_m = bb.in_m ? bb.in_m + (long)z * bb.in_m_stride : wk.in_m;
    if (pl >= wk.PL) {   // the last E workgroup columns: mean part of output pl - PL
        bwd_mean_partial<NA>(md, in_m, head, pl - wk.PL, rc, nrc, mpart, sm);
        return;
    }
    bwd_pair_post<NMT>(md, wk, in_m, gpart, cpart, njs, nrb, part + ((long)pl * nrc + rc) * (1 + D0 + D0 * D0), nrc, pl, rc, sm);
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

// The record of one pair from its sums Iv = (N | A | I) and the step's head (Pm = P = (I + Lambda s)^-1, lam = lambda | rdet),
// all in LDS and complete (the caller has synchronised); PI: 2 D^2 doubles of scratch.  Layout: comment above.
__device__ __forceinline__ void jac_pair_record(int D, const double* Pm, const double* lam, const double* Iv, double* PI, double* __restrict__ o) {
    const int t = threadIdx.x, nI = D * D;
    const double rdet = lam[D];
    const double Nab = Iv[0];
    const double* Av = Iv + 1;
    const double* Im = Iv + 1 + D;
    const int r0 = t / D, c0 = t - r0 * D;   // (D <= 14 on this path: D*D <= 256, one entry per thread, one division)
    for (int e = t; e < nI; e += 256) {
        const int r = r0, c = c0;
        double acc = 0.0;
        for (int k = 0; k < D; ++k) acc = fma(Pm[r * D + k], Im[k * D + c], acc);
        PI[e] = acc;
    }
    __syncthreads();
    // G = rdet (P I P^T / 2 - N (P Lambda + Lambda P^T) / 4), then packed symmetric
    double* Gf = PI + nI;          // [D][D]
    for (int e = t; e < nI; e += 256) {
        const int r = r0, c = c0;
        double acc = 0.0;
        for (int k = 0; k < D; ++k) acc = fma(PI[r * D + k], Pm[c * D + k], acc);   // (P I P^T)[r][c]
        const double pl2 = Pm[r * D + c] * lam[c] + Pm[c * D + r] * lam[r];         // P Lambda + Lambda P^T
        Gf[e] = rdet * (0.5 * acc - 0.25 * Nab * pl2);
    }
    __syncthreads();
    if (t == 0) o[0] = Nab;
    for (int e = t; e < nI; e += 256) {
        const int r = r0, c = c0;
        if (r <= c) o[1 + D + c * (c + 1) / 2 + r] = 0.5 * (Gf[e] + Gf[c * D + r]);
    }
    if (t >= 256 - D) {
        const int r = t - (256 - D);
        double acc = 0.0;
        for (int c = 0; c < D; ++c) acc = fma(Pm[r * D + c], Av[c], acc);
        o[1 + r] = rdet * acc;
    }
}

__global__ __launch_bounds__(256) void k_mm_jac_fin(MMModel md, MMWork wk, const double* __restrict__ part, int nrc,
                                                   const double* __restrict__ head, const double* __restrict__ mpart,
                                                   double* __restrict__ jrec, long jstride, BwdBatch bb, int pl0, RevLocalArgs rl) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    if (rl.loc && (int)blockIdx.x == (int)gridDim.x - 1) {   // the extra workgroup of the step (device reverse chain, rev_local.h)
        rev_local_step(rl.n, rl.rs, rl.E, rl.U, rl.traj, rl.Wp, rl.bp, rl.maxact, rl.loc, blockIdx.y, sm);
        return;
    }
    const int D = md.D, pl = blockIdx.x + pl0, z = blockIdx.y;   // pl0 = PL: the output records only (the fused finish has written the pairs')
    const int nI = D * D, NT2 = D * (D + 1) / 2, recp = 1 + D + NT2;
    part += (long)z * bb.part;
    mpart += (long)z * bb.part;
    head += (long)z * bb.head;
    jrec += (long)z * jstride;
    const int a = pl - wk.PL;   // (launched with pl0 = PL: the output records; the pairs' come from k_mm_jac_rec)
    jac_fin_output(md, head, a, nrc, mpart, jrec + (long)wk.PL * recp + (long)a * (D + NT2 + nI + D * NT2), sm);
}

// Fused finish of the Jacobian tape (round 6): ONE workgroup per (pair, step) adds the sweep's G blocks, contracts the column
// sums with [w_j | 1] on the matrix cores and writes the pair's record -- the partial sums never leave LDS -- beside the
// E x nrc workgroups per step that take the moments of the mean part (their records: k_mm_jac_fin with pl0 = P on the
// outputs only).  Rounds 3-5 cut a pair into nrc chunk-workgroups (k_mm_bwd_post), wrote their N | A | I to memory and ran
// a third launch per chunk of steps (k_mm_jac_fin) over them: 20 800 + 2 600 latency-bound workgroups per rollout at C2u,
// 0.58 ms behind the chain; one pass over the sweep's 7.2 MB per step is what the work needs.
#ifndef JAC_REC_LB
#define JAC_REC_LB 3   // workgroups per CU: a workgroup is a chain of memory round trips (2: 4.92 ms per value + gradient at C2u, 3: 4.90, 4: 93 spilled registers)
#endif
__global__ __launch_bounds__(256, JAC_REC_LB) void k_mm_jac_rec(MMModel md, MMWork wk, const double* __restrict__ gpart, const double* __restrict__ cpart,
                                                       int njs, int nrb, int nrc /* chunks of the MEAN part's points */, const double* __restrict__ head, double* __restrict__ mpart,
                                                       double* __restrict__ jrec, long jstride, BwdBatch bb) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int D = md.D, E = md.E, t = threadIdx.x, z = blockIdx.y;
    const int nI = D * D, rec = 1 + D + nI, NT2 = D * (D + 1) / 2, recp = 1 + D + NT2;
    gpart += (long)z * bb.gpart;
    cpart += (long)z * bb.cpart;
    mpart += (long)z * bb.part;
    head += (long)z * bb.head;
    jrec += (long)z * jstride;
    const double* in_m = bb.in_m + (long)z * bb.in_m_stride;
    if ((int)blockIdx.x >= wk.PL) {   // mean part of output a, point blocks rc, rc + nrc, ..
        const int q = (int)blockIdx.x - wk.PL, a = q / nrc, rc = q - a * nrc;
        bwd_mean_moments_mfma(md, in_m, head, a, rc, nrc, mpart, sm);
        return;
    }
    const int pl = blockIdx.x;
    JAC_STAMP(wk, 48, pl == 0 && z == 0 && t == 0);
    double* Iv = sm + 6 * 256;     // [rec]  behind bwd_pair_post's Gs | Gc | red
    double* Pm = Iv + rec;         // [D][D]
    double* lam = Pm + nI;         // [D + 2]
    double* PI = lam + D + 2;      // [2][D][D]
    const double* hd = head + (long)(E + pl) * (nI + D + 2);
    for (int e = t; e < nI + D + 2; e += 256) (e < nI ? Pm[e] : lam[e - nI]) = hd[e];
    bwd_pair_post<1>(md, wk, in_m, gpart, cpart, njs, nrb, Iv, 1, pl, 0, sm, sm + ((6 * 256 + rec + 3 * nI + D + 2 + 1) & ~1));   // cjl [npad] behind PI, 16-byte aligned
    __syncthreads();
    JAC_STAMP(wk, 52, pl == 0 && z == 0 && t == 0);
    jac_pair_record(D, Pm, lam, Iv, PI, jrec + (long)pl * recp);
    JAC_STAMP(wk, 53, pl == 0 && z == 0 && t == 0);
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

void mm_bwd_geometry(int npad, int Pg, int* njs, int* nrb);
size_t mm_bwd_gpart_size(int npad, int P, int D) {   // one (16 NMT)^2 block per sweep workgroup (sized for the most column splits: 4)
    int njs, nrb;
    mm_bwd_geometry(npad, 1, &njs, &nrb);
    const int nmt = (D + 16) / 16;
    const size_t small = npad <= 256 ? (size_t)16 * 512 : 0;   // one-launch small step: up to 16 workgroups per pair, two blocks each
    return (size_t)std::max(1, P) * std::max((size_t)4 * nrb * nmt * nmt * 256, small);
}
size_t mm_bwd_cpart_size(int npad, int P) {
    int njs, nrb;
    mm_bwd_geometry(npad, 1, &njs, &nrb);
    return (size_t)std::max(1, P) * nrb * npad;
}
size_t mm_jac_rowmom_size(int npad, int P) { return mm_bwd_gpart_size(npad, P, 1); }   // (the Jacobian tape serves D <= 14: one block)
size_t mm_jac_cpart_size(int npad, int P, int) { return mm_bwd_cpart_size(npad, P); }
size_t mm_jac_head_size(int D, int E, int P) { return (size_t)(E + P) * (D * D + D + 2); }
int mm_jac_nt(int npad, int Pg) {   // Pg: pairs of the WHOLE model
    int njs, nrb;
    mm_bwd_geometry(npad, Pg, &njs, &nrb);
    return njs * nrb;
}

// Jacobian tape, per step (on the rollout's critical path): the reverse sweep in place of the forward pair kernel.  It
// leaves the step's row moments / column sums / head records in THIS step's buffers and N_ab as [P][mm_jac_nt][2] tile
// partials for the serial link.
void launch_mm_sweep(hipStream_t st, const MMModel& md, const MMWork& wk, double* rowmom, double* cpart, double* head,
                     double* npart) {
    const int P = wk.PL, E = md.E, D = md.D;
    int njs, nrb;
    mm_bwd_geometry(md.npad, md.E * (md.E + 1) / 2, &njs, &nrb);   // (the model's pairs, not this rank's: the split of a pair's sums is the same on every rank count)
    const int nhead = E + P, per_row = nrb * njs;
    dim3 grid(nrb, P + (nhead + per_row - 1) / per_row, njs);
    const int nI = D * D;
    const size_t lds_pair = sizeof(double) * std::max((size_t)4 * 16 * ((md.npad / 16 + njs - 1) / njs) + 2 * (BWD_CH * bwd_tp(wk.KP) + 2 * BWD_CH) + 4 * BWD_SCR_W, (size_t)4 * nI + D);
    const double* bars = nullptr;
    launch_bwd_pair(st, grid, lds_pair, md, wk, rowmom, cpart, njs, bars, head, npart);
}

// Jacobian tape, once per rollout (off the critical path): sums, moments and records of ALL H steps in two launches --
// nothing of the forward chain waits for them, so they run at throughput instead of paying their latency per step.
// Per-step arrays: rowmom / cpart / head / part advance by the sizes above, in_m is the head of the step's tape record.
void launch_mm_jac_finish(hipStream_t st, const MMModel& md, const MMWork& wk, int H, const double* rowmom, const double* cpart,
                          double* head, double* part, const double* tape, size_t tape_stride, double* jrec, int small_nch, const RevLocalArgs* rlp) {
    if (H <= 0) return;
    RevLocalArgs rl{};
    if (rlp) rl = *rlp;
    const int P = wk.PL, E = md.E, D = md.D;
    int njs, nrb;
    mm_bwd_geometry(md.npad, md.E * (md.E + 1) / 2, &njs, &nrb);   // (the model's pairs, not this rank's: the split of a pair's sums is the same on every rank count)
    if (small_nch > 0) {   // the steps ran as one launch each (small_sweep): two blocks per chunk-workgroup, and the inverses are still to be made
        njs = 0;
        nrb = small_nch;
    }
    const int nI = D * D, NS = mm_jac_ns(D);
    const int nrc = mm_bwd_rc(md.npad);
    BwdBatch bb;
    bb.gpart = (long)mm_jac_rowmom_size(md.npad, P);
    bb.cpart = (long)mm_jac_cpart_size(md.npad, P, wk.EL);   // (EL = E on one rank; the first EL local pairs are the diagonal ones)
    bb.part = (long)mm_jac_part_size(D, E, P, md.npad);
    bb.head = (long)mm_jac_head_size(D, E, P);
    bb.in_m = tape;
    bb.in_m_stride = (long)tape_stride;
    double* mpart = part + (size_t)P * nrc * (1 + D + nI);
    if (small_nch > 0)
        hipLaunchKernelGGL(k_mm_bwd_head, dim3(E + P, H), dim3(256), sizeof(double) * ((size_t)4 * nI + D), st, md, wk, head, bb.head, tape + D,
                           (long)tape_stride);
    const size_t lds_fin = sizeof(double) * std::max(std::max((size_t)4 * nI + 4 * D + 8, (size_t)3 * nI + NS + D + 2 * nI * D),
                                                     rl.loc ? rev_local_lds_doubles(rl.E, rl.U) : (size_t)0);
    const long jstride = (long)mm_jac_rec_size(D, E, P);
    {
        // the mean part's points in nrcm chunks per output and step: a workgroup's set-up and its epilogue (five tiles through
        // LDS, scattered stores) cost as much as four 64-point blocks -- 8 chunks of 2 blocks (the split finish's cut) spent
        // 22 us per workgroup at three per CU
        const int nrcm = std::max(1, std::min(2, md.npad / 64));
        const size_t lds_rec = sizeof(double) * std::max((size_t)6 * 256 + (1 + D + nI) + 3 * nI + D + 2 + 1 + md.npad, (size_t)256 + 64 * 17 + 64 + 4 * 256 + 8 * JAC_MT + 2);
        hipLaunchKernelGGL(k_mm_jac_rec, dim3(P + E * nrcm, H), dim3(256), lds_rec, st, md, wk, rowmom, cpart, njs, nrb, nrcm, head, mpart, jrec,
                           jstride, bb);
        hipLaunchKernelGGL(k_mm_jac_fin, dim3(E + (rl.loc ? 1 : 0), H), dim3(256), lds_fin, st, md, wk, part, nrcm, head, mpart, jrec, jstride, bb, P, rl);
    }
}

// Pg: the pairs of the WHOLE model, E (E + 1) / 2 -- never the local count of a rank: the column split decides how a pair's
// sums are cut and added, and a sharded value-and-gradient rollout is bit-identical to the single-rank one because of that.
void mm_bwd_geometry(int npad, int Pg, int* njs, int* nrb) {
    *nrb = (npad + 64 * BWD_RT - 1) / (64 * BWD_RT);
    int q = 1;   // column splits: enough workgroups for a few balanced rounds of the chip
    while (q < 4 && (npad / 16) % (2 * q) == 0 && (long)*nrb * Pg * q < 1536) q *= 2;
    if (const char* ev = getenv("PILCO_BWD_NJS")) q = std::max(1, std::min(atoi(ev), npad / 16));
    *njs = q;
}

void launch_mm_bwd(hipStream_t st, const MMModel& md, const MMWork& wk, double* rowmom, double* cpart, double* part,
                   const double* bars, double* head, double* out, unsigned* done, double* sum_out) {
    const int P = wk.PL, E = md.E, D = md.D;
    int njs, nrb;
    mm_bwd_geometry(md.npad, md.E * (md.E + 1) / 2, &njs, &nrb);   // (the model's pairs, not this rank's: the split of a pair's sums is the same on every rank count)
    const int nhead = E + P, per_row = nrb * njs;
    dim3 grid(nrb, P + (nhead + per_row - 1) / per_row, njs);   // the rows past P hold the head workgroups
    const int LD = D | 1, nI = D * D;
    const size_t lds_pair = sizeof(double) * std::max((size_t)4 * 16 * ((md.npad / 16 + njs - 1) / njs) + 2 * (BWD_CH * bwd_tp(wk.KP) + 2 * BWD_CH) + 4 * BWD_SCR_W, (size_t)4 * nI + D);
    double* npart = nullptr;
    launch_bwd_pair(st, grid, lds_pair, md, wk, rowmom, cpart, njs, bars, head, npart);
    const int nrc = mm_bwd_rc(md.npad);
    double* mpart = part + (size_t)P * nrc * (1 + D + nI);
    const int nmt = (D + 16) / 16;
    const size_t lds_post = sizeof(double) * std::max((size_t)2 * 256 * nmt * nmt + 4 * 256, (size_t)nI + 64 * LD + 128 + D + 2);
    const BwdBatch nob{0, 0, 0, 0, nullptr, 0};
#define PPOST(NA_, M_)                                                                                                          \
    hipLaunchKernelGGL((k_mm_bwd_post<NA_, M_>), dim3(P + E, nrc), dim3(256), lds_post, st, md, wk, rowmom, cpart, njs, nrb, part, nrc, \
                       head, mpart, nob)
    if (D <= 14) PPOST(1, 1);
    else if (nmt == 1) PPOST(5, 1);
    else if (nmt == 2) PPOST(5, 2);
    else PPOST(5, 3);
#undef PPOST
    const size_t lds_fin = sizeof(double) * ((size_t)3 * nI + 4 * D + 8);
    hipLaunchKernelGGL(k_mm_bwd_fin, dim3(P + E), dim3(256), lds_fin, st, md, wk, part, nrc, bars, head, mpart, out, done, sum_out);
}
int mm_bwd_rc(int npad) { return std::min(BWD_RC, npad / 64); }

}  // namespace pilco
