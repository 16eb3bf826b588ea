// GPRFITC training objective on the device: the log marginal likelihood of the FITC sparse GP and its gradient w.r.t.
// the kernel hyper-parameters AND the inducing inputs of every output.
//
// Reference: SMGPR builds one gpflow.models.GPRFITC per output, each with its OWN trainable inducing inputs Z
// (pilco/models/smgpr.py:16-22); MGPR.optimize hands model.training_loss and ALL its trainable variables -- Z included
// -- to SciPy through GPflow's autodiff (pilco/models/mgpr.py:47-75).  GPflow's arithmetic is third-party code
// (gpflow/models/sgpr.py, GPRFITC.common_terms / fitc_log_marginal_likelihood, v2.1; the tests hold a torch restatement):
//   Kuu = K(Z,Z) + 1e-6 I = Luu Luu^T,  V = Luu^-1 Kuf,  nu = var - colsum(V^2) + sn2,  B = I + V diag(1/nu) V^T = L L^T,
//   f = -1/2 sum y^2/nu + 1/2 |L^-1 V (y/nu)|^2 - N/2 log 2pi - 1/2 sum log nu - sum log diag L.
// No priors: smgpr.py creates its kernels without any.
//
// Gradient (hand-derived; torch autograd of the restated objective is the test oracle).  With Sigma = Qff + diag(nu),
// a = Sigma^-1 y, G = 1/2 (a a^T - Sigma^-1), g = diag(G), A = Kuu^-1 Kuf:
//   df/dKuf = 2 A (G - diag g),   df/dKuu = -A (G - diag g) A^T,   df/dvar += sum g,   df/dsn2 = sum g,
// and with U = V diag(nu)^-1/2, P = L^-1 U:  Sigma^-1 = D (I - P^T P) D  (D = diag(nu)^-1/2),  A Sigma^-1 = (L^-1 Luu^-1)^T P D.
// Everything O(M^2 N) is an f64-MFMA GEMM of csrc/linalg.hip; the kernel derivatives are weighted reductions over the
// (m, n) grid with K recomputed on the fly.  All outputs are batched (z = output).
#include "ctx.h"
#include "mm_device.h"   // d4, the MFMA guards

namespace pilco {

// per column n < N (thread per column, coalesced over n): t_n = sum_m P_mn gamma_m, cn_n = sum_m P_mn^2, then
//   ytil = y / (sn G),  a_n = (ytil - t_n) / (sn G),  g_n = 1/2 (a_n^2 - (1 - cn_n) / (sn2 G^2))
__global__ __launch_bounds__(256) void k_fitc_cols(const double* __restrict__ P, const double* __restrict__ gam, const double* __restrict__ G,
                                                   const double* __restrict__ y, const double* __restrict__ noise, int M, int mpad,
                                                   int N, int npad, double* __restrict__ a_out, double* __restrict__ g_out) {
    // 64 columns per workgroup, the rows dealt over its four waves (a thread per column walking all M rows: 80 workgroups on
    // the chip and 25 round trips per thread at M = 200, N = 5000 -- 30 us)
    __shared__ double red[2][4][64];
    const int b = blockIdx.y, lane = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + lane, nc = min(n, npad - 1);
    const double* Pb = P + (long)b * mpad * npad;
    const double* gb = gam + (long)b * mpad;
    double tn = 0.0, cn = 0.0;
    for (int m0 = q; m0 < M; m0 += 4 * 8) {   // eight rows requested together
        double pv[8], gm[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int m = min(m0 + 4 * u, M - 1);
            pv[u] = Pb[(long)m * npad + nc];
            gm[u] = gb[m];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (m0 + 4 * u < M) {
                tn = fma(pv[u], gm[u], tn);
                cn = fma(pv[u], pv[u], cn);
            }
    }
    red[0][q][lane] = tn;
    red[1][q][lane] = cn;
    __syncthreads();
    if (q != 0 || n >= npad) return;
    double av = 0.0, gv = 0.0;
    if (n < N) {
        tn = (red[0][0][lane] + red[0][1][lane]) + (red[0][2][lane] + red[0][3][lane]);
        cn = (red[1][0][lane] + red[1][1][lane]) + (red[1][2][lane] + red[1][3][lane]);
        const double sn2 = noise[b], sn = sqrt(sn2), Gn = G[(long)b * npad + n];
        tn /= sn;                                          // gam holds gamma * sn (= Am^-1 r0)
        const double il = 1.0 / (sn * Gn);                 // diag(nu)^-1/2
        av = (y[(long)b * npad + n] * il - tn) * il;
        gv = 0.5 * (av * av - (1.0 - cn) * il * il);
    }
    a_out[(long)b * npad + n] = av;
    g_out[(long)b * npad + n] = gv;
}

// c_m = sum_n A'_mn G_n a_n   (A = A' o G, column scaling); one wave per row
__global__ __launch_bounds__(256) void k_fitc_c(const double* __restrict__ Ap, const double* __restrict__ G, const double* __restrict__ a,
                                                int mpad, int N, int npad, double* __restrict__ c) {
    const int b = blockIdx.y, lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= mpad) return;
    const double* row = Ap + ((long)b * mpad + m) * npad;
    const double* Gb = G + (long)b * npad;
    const double* ab = a + (long)b * npad;
    double s = 0.0;
    for (int n0 = lane; n0 < N; n0 += 64 * 8) {   // eight requests per array in flight (one at a time: a latency chain of N / 64 round trips)
        double rv[8], gv[8], av[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int n = min(n0 + 64 * u, npad - 1);   // (clamped: the padding is there to be read; its terms are masked below)
            rv[u] = row[n];
            gv[u] = Gb[n];
            av[u] = ab[n];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) s = fma((n0 + 64 * u < N) ? rv[u] * gv[u] : 0.0, av[u], s);
    }
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
    if (lane == 0) c[(long)b * mpad + m] = s;
}

// DKuf_mn = c_m a_n - T3_mn / G_n - 2 A'_mn G_n g_n   (in place of T3);   W2g_mn = (1/2 T3_mn / G_n + A'_mn G_n g_n) G_n  (in place of P)
__global__ __launch_bounds__(256) void k_fitc_combine(double* __restrict__ T3, double* __restrict__ P, const double* __restrict__ Ap,
                                                      const double* __restrict__ c, const double* __restrict__ a, const double* __restrict__ g,
                                                      const double* __restrict__ G, int M, int mpad, int N, int npad) {
    const int b = blockIdx.z, m = blockIdx.y;
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= npad) return;
    const long idx = ((long)b * mpad + m) * npad + n;
    double dk = 0.0, w2 = 0.0;
    if (m < M && n < N) {
        const double Gn = G[(long)b * npad + n], gn = g[(long)b * npad + n];
        const double t3 = T3[idx] / Gn, ag = Ap[idx] * Gn * gn;
        dk = c[(long)b * mpad + m] * a[(long)b * npad + n] - t3 - 2.0 * ag;
        w2 = (0.5 * t3 + ag) * Gn;
    }
    T3[idx] = dk;
    P[idx] = w2;
}

// Kernel-derivative reductions of one row m of a weight matrix DK (mpad x ldk) against K(z_m, p_n), n < N:
//   w = (DK_mn + cc * c_m c_n) K_mn,   out[m] = ( sum_n w (p_nd - z_md) / l_d^2  [D] | sum_n w (z_md - p_nd)^2 / l_d^3  [D] | sum_n w / var )
// Zt: [D][mpad] of this output; Pt: [D][ldp] (stride sPt per output, 0 = shared data X).  One workgroup per (row, output).
// The per-dimension arrays have the compile-time width DT >= D (round 2 indexed [32]-arrays with the run-time D: 1.3 KB of
// scratch memory per thread, 709 us per launch at M = 200, N = 5000 -- 45 % of a FITC objective evaluation).
constexpr int FT_MAXD = 32;
template <int DT>
__global__ __launch_bounds__(256) void k_fitc_kgrad(const double* __restrict__ DK, int ldk, const double* __restrict__ Zt, int mpad,
                                                    const double* __restrict__ Pt, int ldp, long sPt, int N, int D,
                                                    const double* __restrict__ ls, const double* __restrict__ var, const double* __restrict__ c,
                                                    double cc, double* __restrict__ out) {
    __shared__ double red[4][2 * FT_MAXD + 1];
    const int b = blockIdx.y, m = blockIdx.x, t = threadIdx.x;
    const double* Zb = Zt + (long)b * D * mpad;
    const double* Pb = Pt + (long)b * sPt;
    double z[DT], il[DT], accz[DT], accl[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d) {
        z[d] = (d < D) ? Zb[(long)d * mpad + m] : 0.0;
        il[d] = (d < D) ? 1.0 / ls[b * D + d] : 0.0;
        accz[d] = 0.0;
        accl[d] = 0.0;
    }
    const double v = var[b];
    const double cm = c ? c[(long)b * mpad + m] : 0.0;
    double accv = 0.0;
    const double* row = DK + ((long)b * mpad + m) * ldk;
    for (int n = t; n < N; n += 256) {
        double df[DT];
        double r2 = 0.0;
#pragma unroll
        for (int d = 0; d < DT; ++d)
            if (d < D) {
                df[d] = (Pb[(long)d * ldp + n] - z[d]) * il[d];
                r2 = fma(df[d], df[d], r2);
            } else {
                df[d] = 0.0;
            }
        double wgt = row[n];
        if (c) wgt = fma(cc * cm, c[(long)b * mpad + n], wgt);
        const double w = wgt * v * exp(-0.5 * r2);
#pragma unroll
        for (int d = 0; d < DT; ++d)
            if (d < D) {
                accz[d] = fma(w, df[d] * il[d], accz[d]);
                accl[d] = fma(w, df[d] * df[d] * il[d], accl[d]);
            }
        accv += w;
    }
#pragma unroll
    for (int d = 0; d < 2 * DT + 1; ++d) {
        // slot d of the output: accz | accl | accv / v  (slots D .. 2 D - 1 hold accl: the run-time D decides where a register lands)
        double s = 0.0;
        if (d < DT) s = accz[d];
        else if (d < 2 * DT) s = accl[d - DT];
        else s = accv / v;
        for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
        const int slot = d < DT ? d : (d < 2 * DT ? D + (d - DT) : 2 * D);
        const bool live = d < DT ? d < D : (d < 2 * DT ? d - DT < D : true);
        if (live && (t & 63) == 0) red[t >> 6][slot] = s;
    }
    __syncthreads();
    if (t < 2 * D + 1) out[((long)b * mpad + m) * (2 * FT_MAXD + 1) + t] = (red[0][t] + red[1][t]) + (red[2][t] + red[3][t]);
}
template <typename... Args>
static void launch_fitc_kgrad(int D, dim3 grid, hipStream_t st, Args... args) {
    if (D <= 4) hipLaunchKernelGGL(k_fitc_kgrad<4>, grid, dim3(256), 0, st, args...);
    else if (D <= 8) hipLaunchKernelGGL(k_fitc_kgrad<8>, grid, dim3(256), 0, st, args...);
    else if (D <= 12) hipLaunchKernelGGL(k_fitc_kgrad<12>, grid, dim3(256), 0, st, args...);
    else if (D <= 16) hipLaunchKernelGGL(k_fitc_kgrad<16>, grid, dim3(256), 0, st, args...);
    else if (D <= 24) hipLaunchKernelGGL(k_fitc_kgrad<24>, grid, dim3(256), 0, st, args...);
    else hipLaunchKernelGGL(k_fitc_kgrad<32>, grid, dim3(256), 0, st, args...);
}

// The same reductions on the matrix cores (round 6; D <= 14).  The VALU kernel above re-reads a point's D coordinates for every
// row m it meets -- 880 MB of L2 traffic and a load-latency chain per thread: 350 us for the 10^7 kernel evaluations of config 4,
// a tenth of what the arithmetic needs.  With p~ = p / l, z~ = z / l the tile of squared distances is a product,
//     r2[n][m] = sum_k A1[n][k] B1[k][m],   A1 = (p~_n | |p~_n|^2 | 1),   B1 = (-2 z~_m | 1 | |z~_m|^2)     (K = D + 2),
// computed TRANSPOSED (points along the result registers) so that w = DK_mn var exp(-r2 / 2) lands in the A-operand layout of
// a second product with the points' coordinates,
//     out[m][c] = sum_n w_mn B2[n][c],   B2 = (p_n | 1 | 0..) and (p_n^2 | 0..):   WP = sum w p,  rs = sum w,  WP2 = sum w p^2,
// from which  sum w (p - z) / l^2 = (WP - z rs) / l^2  and  sum w (z - p)^2 / l^3 = (WP2 - 2 z WP + z^2 rs) / l^3.
// One workgroup per (16-row tile, output, slice of the points); its four waves take the 16-point tiles of the slice in turn and
// add their results in a fixed order.  Output: out[slice][b][mpad][2 FT_MAXD + 1], the row format of the kernel above.
constexpr int FT_NSPLIT = 4;
template <int KS>   // k-steps of the distance product: 4 KS >= D + 2
__global__ __launch_bounds__(256) void k_fitc_kgrad_mfma(const double* __restrict__ DK, int ldk, const double* __restrict__ Zt, int mpad,
                                                         const double* __restrict__ Pt, int ldp, long sPt, int N, int M, int D,
                                                         const double* __restrict__ ls, const double* __restrict__ var,
                                                         const double* __restrict__ c, double cc, double* __restrict__ out, int E) {
    __shared__ double red[4][2][256];
    const int b = blockIdx.y, m0 = blockIdx.x * 16, sp = blockIdx.z, nsp = gridDim.z;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, lr = lane >> 4, lc = lane & 15;
    const double* Zb = Zt + (long)b * D * mpad;
    const double* Pb = Pt + (long)b * sPt;
    const double v = var[b];
    // B1 fragments (constant over the points): row k = 4 q + lr, column m = m0 + lc (clamped: rows past M weigh zero below)
    const int mc = min(m0 + lc, mpad - 1);
    double zt2 = 0.0;
    for (int d = 0; d < D; ++d) {
        const double zd = Zb[(long)d * mpad + mc] / ls[b * D + d];
        zt2 = fma(zd, zd, zt2);
    }
    double b1[KS], il1[KS];
#pragma unroll
    for (int q = 0; q < KS; ++q) {
        const int k = 4 * q + lr, kc = min(k, D - 1);
        const double il = 1.0 / ls[b * D + kc];
        il1[q] = k < D ? il : 0.0;                                   // A1's scale for this lane's coordinate
        b1[q] = k < D ? -2.0 * Zb[(long)kc * mpad + mc] * il : (k == D ? 1.0 : (k == D + 1 ? zt2 : 0.0));
    }
    const double cm = c ? c[(long)b * mpad + mc] : 0.0;
    const bool mok = m0 + lc < M;
    const int cB = min(lc, D - 1);                                     // B2's coordinate row of this lane
    d4 C0 = d4{0.0, 0.0, 0.0, 0.0}, C1 = d4{0.0, 0.0, 0.0, 0.0};
    // this workgroup's slice of the 16-point tiles, dealt over the waves
    const int ntile = (N + 15) / 16, per = (ntile + nsp - 1) / nsp, t0 = sp * per, t1 = min(ntile, t0 + per);
    const double* row = DK + ((long)b * mpad + mc) * ldk;
    for (int tl = t0 + w; tl < t1; tl += 4) {
        const int n0 = 16 * tl;
        // A1: row n = n0 + lc, column k = 4 q + lr  (requests unconditional, clamped into the padding)
        const int nA = min(n0 + lc, ldp - 1);
        double a1[KS], sq = 0.0;
#pragma unroll
        for (int q = 0; q < KS; ++q) {
            const int kc = min(4 * q + lr, D - 1);
            a1[q] = Pb[(long)kc * ldp + nA] * il1[q];
            sq = fma(a1[q], a1[q], sq);
        }
        // weights and B2: point n = n0 + 4 r + lr
        double wg[4], pb[4], cn[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int nB = min(n0 + 4 * r + lr, ldp - 1);
            wg[r] = row[min(n0 + 4 * r + lr, ldk - 1)];
            pb[r] = Pb[(long)cB * ldp + nB];
            cn[r] = c ? c[(long)b * mpad + min(nB, mpad - 1)] : 0.0;
        }
        // |p~_n|^2: the four k-groups of a point sit in lanes lc, lc + 16, lc + 32, lc + 48
        sq += __shfl_xor(sq, 16);
        sq += __shfl_xor(sq, 32);
#pragma unroll
        for (int q = 0; q < KS; ++q) {
            const int k = 4 * q + lr;
            if (k == D) a1[q] = sq;
            else if (k == D + 1) a1[q] = 1.0;
        }
        d4 ST = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int q = 0; q < KS; ++q) {
            ST = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[q], b1[q], ST, 0, 0, 0);
            MFMA_KEEP_ALIVE(a1[q]);
            MFMA_KEEP_ALIVE(b1[q]);
        }
        MFMA_RESULT_FENCE(ST);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool ok = mok && n0 + 4 * r + lr < N;
            double wt = wg[r];
            if (c) wt = fma(cc * cm, cn[r], wt);
            const double wv = ok ? wt * v * exp(-0.5 * fmax(ST[r], 0.0)) : 0.0;
            const double b20 = lc < D ? pb[r] : (lc == D ? 1.0 : 0.0), b21 = lc < D ? pb[r] * pb[r] : 0.0;
            C0 = __builtin_amdgcn_mfma_f64_16x16x4f64(wv, b20, C0, 0, 0, 0);
            C1 = __builtin_amdgcn_mfma_f64_16x16x4f64(wv, b21, C1, 0, 0, 0);
            MFMA_KEEP_ALIVE(wv);
            MFMA_KEEP_ALIVE(b20);
            MFMA_KEEP_ALIVE(b21);
        }
    }
    MFMA_RESULT_FENCE(C0);
    MFMA_RESULT_FENCE(C1);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        red[w][0][r * 64 + lane] = C0[r];
        red[w][1][r * 64 + lane] = C1[r];
    }
    __syncthreads();
    // element t of a tile: row m = (t >> 4) & 3 + 4 (t >> 6), column t & 15
    double* s0 = &red[0][0][0];   // (reused: sums of the four waves)
    const double v0 = (red[0][0][t] + red[1][0][t]) + (red[2][0][t] + red[3][0][t]);
    const double v1 = (red[0][1][t] + red[1][1][t]) + (red[2][1][t] + red[3][1][t]);
    __syncthreads();
    s0[t] = v0;
    s0[256 + t] = v1;
    __syncthreads();
    const int mr = ((t >> 4) & 3) + 4 * (t >> 6), d = t & 15, m = m0 + mr;   // thread -> (row, coordinate)
    const int PW = 2 * FT_MAXD + 1;
    if (m < M) {
        double* o = out + (((long)sp * E + b) * mpad + m) * PW;
        const int ti = (mr & 3) * 16 + (mr >> 2) * 64;   // index of (row mr, column 0) in the tile layout
        const double rs = s0[ti + D];
        if (d < D) {
            const double il = 1.0 / ls[b * D + d], z = Zb[(long)d * mpad + m], wp = s0[ti + d], wp2 = s0[256 + ti + d];
            o[d] = il * il * (wp - z * rs);
            o[D + d] = il * il * il * ((wp2 - 2.0 * z * wp) + z * z * rs);
        } else if (d == D) {
            o[2 * D] = rs / v;
        }
    }
}
template <typename... Args>
static void launch_fitc_kgrad_mfma(int D, dim3 grid, hipStream_t st, Args... args) {
    if (D + 2 <= 8) hipLaunchKernelGGL(k_fitc_kgrad_mfma<2>, grid, dim3(256), 0, st, args...);
    else if (D + 2 <= 12) hipLaunchKernelGGL(k_fitc_kgrad_mfma<3>, grid, dim3(256), 0, st, args...);
    else hipLaunchKernelGGL(k_fitc_kgrad_mfma<4>, grid, dim3(256), 0, st, args...);
}

// sums for the value: out[b] = (sum (y/G)^2, sum log G, sum g)
__global__ __launch_bounds__(1024) void k_fitc_sums(const double* __restrict__ y, const double* __restrict__ G, const double* __restrict__ g, int N,
                                                    int npad, double* __restrict__ out) {
    // (one workgroup per output: 1024 threads -- with 256 a thread's 20 dependent iterations of a division and a log were 16 us)
    __shared__ double red[16][3];
    const int b = blockIdx.x, t = threadIdx.x;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    for (int n = t; n < N; n += 1024) {
        const double Gn = G[(long)b * npad + n], yy = y[(long)b * npad + n] / Gn;
        s0 = fma(yy, yy, s0);
        s1 += log(Gn);
        s2 += g[(long)b * npad + n];
    }
    for (int off = 32; off > 0; off >>= 1) {
        s0 += __shfl_down(s0, off);
        s1 += __shfl_down(s1, off);
        s2 += __shfl_down(s2, off);
    }
    if ((t & 63) == 0) { red[t >> 6][0] = s0; red[t >> 6][1] = s1; red[t >> 6][2] = s2; }
    __syncthreads();
    if (t < 3) {
        double v = 0.0;
        for (int w = 0; w < 16; ++w) v += red[w][t];   // fixed order
        out[b * 3 + t] = v;
    }
}

// The two reductions' per-row sums folded into what the caller gets (one workgroup per output, fixed orders):
//   res[e] = ( d f / d lengthscale [D] | d f / d var | d f / d sn2 | gam . gam | d f / d Z [M][D] ),   sg = sums[3 e + 2] = sum_n g_n
// -- 1.7 kB per output cross to the host instead of the (slices + 1) x M x 65 partial rows (6.6 MB at config 4, pageable).
__global__ __launch_bounds__(256) void k_fitc_kgrad_fin(const double* __restrict__ uf, int nsp, const double* __restrict__ uu, const double* __restrict__ sums,
                                                        const double* __restrict__ gam, int E, int M, int mpad, int D, double* __restrict__ res) {
    __shared__ double red[4];
    const int e = blockIdx.x, t = threadIdx.x, PW = 2 * FT_MAXD + 1;
    const long RS = (long)D + 3 + (long)M * D;
    double* o = res + (long)e * RS;
    const double* u1 = uu + (long)e * mpad * PW;
    for (int i = blockIdx.y * 256 + t; i < M * D; i += 256 * gridDim.y) {   // (blockIdx.y: slices of d f / d Z; the sums below in slice 0)
        const int m = i / D, d = i - m * D;
        double a = 0.0;
        for (int q = 0; q < nsp; ++q) a += uf[(((long)q * E + e) * mpad + m) * PW + d];
        o[D + 3 + i] = -(a + 2.0 * u1[(long)m * PW + d]);
    }
    if (blockIdx.y != 0) return;
    const double sg = sums[3 * e + 2];
    // lengthscales (slots D + d) and the variance (slot 2 D): 16 threads per slot, rows dealt over them, fixed-order tree
    // (one thread per slot walked M (slices + 1) strided loads one after the other: 290 us for ten workgroups)
    for (int s0 = 0; s0 <= D; s0 += 16) {
        const int sl = s0 + (t >> 4), gi = t & 15, slot = D + min(sl, D);
        double acc = 0.0;
        for (int m0 = gi; m0 < M; m0 += 16 * 4) {   // four rows' (slices + 1) requests together; the sums in the order of the plain loop
            double v[4][FT_NSPLIT + 1];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int m = min(m0 + 16 * k, M - 1);
#pragma unroll
                for (int q = 0; q < FT_NSPLIT; ++q) v[k][q] = uf[(((long)min(q, nsp - 1) * E + e) * mpad + m) * PW + slot];
                v[k][FT_NSPLIT] = u1[(long)m * PW + slot];
            }
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (m0 + 16 * k < M) {
                    double a = 0.0;
#pragma unroll
                    for (int q = 0; q < FT_NSPLIT; ++q) a += (q < nsp) ? v[k][q] : 0.0;
                    acc += a + v[k][FT_NSPLIT];
                }
        }
        for (int off = 8; off > 0; off >>= 1) acc += __shfl_down(acc, off, 16);
        if (gi == 0 && sl <= D) o[sl] = (sl < D) ? -acc : -(sg + acc);
    }
    if (t == D + 1) o[D + 1] = -sg;
    double g2 = 0.0;
    for (int m = t; m < M; m += 256) g2 = fma(gam[(long)e * mpad + m], gam[(long)e * mpad + m], g2);
    for (int off = 32; off > 0; off >>= 1) g2 += __shfl_down(g2, off);
    if ((t & 63) == 0) red[t >> 6] = g2;
    __syncthreads();
    if (t == 0) o[D + 2] = (red[0] + red[1]) + (red[2] + red[3]);
}

}  // namespace pilco

using namespace pilco;

// The objective for a batch of E outputs given as compact arrays (all outputs of the slot on one rank; the outputs a rank
// owns when the model is sharded): ls [E][D], var [E], noise [E], Yt [E][Npad], Z_all [E][M][D] (host).
static int fitc_nlml_batch(pilco_ctx* ctx, Slot& s, int E, const double* o_ls, const double* o_var, const double* o_noise, const double* o_Yt,
                           const double* Z_all, int M, double* nlml, double* grad_hyp, double* grad_Z) {
    HIPCHK(hipSetDevice(ctx->device));
    const int D = s.D, N = s.N, Np = s.Npad, Mp = round_up(M, NB);
    const size_t mm = (size_t)Mp * Mp, mn = (size_t)Mp * Np;
    hipStream_t st = ctx->st;
    // the slot's FITC buffers are scratch here; whatever factorisation the slot held is invalidated
    s.factor_valid = false;
    s.user_factors = false;
    ENSURE(s.K, E * mm);
    ENSURE(s.Linv, E * mm);
    ENSURE(s.Kmn, E * mn);      // A' = Luu^-T Vb
    ENSURE(s.V2, E * mn);       // Kuf -> V -> Vb
    ENSURE(s.Am, E * mm);
    ENSURE(s.AmInv, E * mm);
    ENSURE(s.iAt, E * mm);
    ENSURE(s.ksplit_ws, (size_t)FITC_KSPLIT * E * mm);
    ENSURE(s.iK, E * mm);       // DKuu
    ENSURE(s.G, (size_t)E * Np);
    ENSURE(s.Tscr, std::max(std::max(E * mm, (size_t)E * Mp * ((size_t)(Np + 63) / 64)), (size_t)E * Mp * (2 * FT_MAXD + 1) * (FT_NSPLIT + 1)));
    ENSURE(s.ft_P, E * mn);
    ENSURE(s.ft_T3, E * mn);
    ENSURE(s.ft_Z, (size_t)E * D * Mp + (size_t)E * M * D);
    ENSURE(s.vec, (size_t)E * (4 * (size_t)std::max(Mp, Np) + 8) + 8 + (size_t)E * ((size_t)D + 3 + (size_t)M * D));
    double* Zt = s.ft_Z.p;                       // [E][D][Mp]
    double* Zraw = Zt + (size_t)E * D * Mp;      // [E][M][D] staging
    const long sZ = (long)D * Mp;
    double* Kuu = s.K.p;
    double* V = s.V2.p;
    const bool want_grad = grad_hyp || grad_Z;
    double* r0 = s.vec.p;                                    // [E][Mp]  Vb (y / G)
    double* gam = r0 + (size_t)E * Mp;                       // [E][Mp]  AmInv r0 = gamma sn   (gamma = L^-1 U ytil)
    double* av = gam + (size_t)E * Mp;                       // [E][Np]
    double* gv = av + (size_t)E * Np;                        // [E][Np]
    double* cv = gv + (size_t)E * Np;                        // [E][Mp]
    double* sums = cv + (size_t)E * Mp;                      // [E][3] + logdet [E]
    const size_t RS = (size_t)D + 3 + (size_t)M * D;         // per output: d ls | d var | d sn2 | gam . gam | d Z
    double* res = sums + 4 * (size_t)E + 4;                  // [E][RS]
    // per-row partial sums of the two kernel-derivative reductions: Kuf's in nsp slices of the points, then Kuu's
    const bool kg_mfma = D <= 14 && getenv("PILCO_FITC_KGRAD_VALU") == nullptr;
    const int nsp = kg_mfma ? (N >= 2048 ? FT_NSPLIT : 1) : 1;
    double* part_uf = s.Tscr.p;
    double* part_uu = part_uf + (size_t)nsp * E * Mp * (2 * FT_MAXD + 1);
    // everything between the upload of Z and the downloads is a fixed launch sequence (~80 launches at M = 200): one graph
    std::vector<unsigned long long> key;
    for (const DevBuf* b : {&s.K, &s.Linv, &s.Kmn, &s.V2, &s.Am, &s.AmInv, &s.iAt, &s.ksplit_ws, &s.iK, &s.G, &s.Tscr, &s.ft_P,
                            &s.ft_T3, &s.ft_Z, &s.vec, &s.Xt})
        key.push_back((unsigned long long)(uintptr_t)b->p);
    for (const void* q : {(const void*)o_ls, (const void*)o_var, (const void*)o_noise, (const void*)o_Yt, (const void*)ctx->d_info})
        key.push_back((unsigned long long)(uintptr_t)q);
    for (int v : {E, M, Mp, N, Np, D, want_grad ? 1 : 0, nsp, kg_mfma ? 1 : 0}) key.push_back((unsigned long long)v);
    auto chain = [&]() -> int {
    launch_transpose_points(st, Zraw, M, D, Zt, Mp, E, (long)M * D, (long)D * Mp);
    HIPCHK(hipMemsetAsync(ctx->d_info, 0, sizeof(int) * 64, st));
    launch_gram(st, Zt, Mp, M, Zt, Mp, M, D, o_ls, o_var, E, Kuu, Mp, Mp, 2, nullptr, 1e-6, sZ, sZ);
    launch_gram(st, Zt, Mp, M, s.Xt.p, Np, N, D, o_ls, o_var, E, s.Kmn.p, Mp, Np, 0, nullptr, 0.0, sZ, 0);
    launch_potrf(st, Kuu, Mp, E, s.Linv.p, ctx->d_info, true);
    launch_trtri(st, Kuu, Mp, E, s.Linv.p, s.Tscr.p, (long)mm);
    GemmDesc g{};
    g.A = s.Linv.p; g.lda = Mp; g.sA = (long)mm;            // V = Luu^-1 Kuf
    g.B = s.Kmn.p; g.ldb = Np; g.sB = (long)mn;
    g.C = V; g.ldc = Np; g.sC = (long)mn;
    g.M = Mp; g.N = Np; g.K = Mp; g.alpha = 1.0; g.beta = 0.0; g.k_mode = 3;
    launch_gemm(st, g, false, false, E);
    launch_fitc_scale_rhs(st, V, Mp, Np, E, o_var, o_noise, s.G.p, o_Yt, s.Tscr.p, r0);   // G = sqrt(nu) / sn, V <- Vb = V / G, r0 = Vb (y / G)
    g = GemmDesc{};                                          // Am = Vb Vb^T + sn2 I = sn2 B
    g.A = V; g.lda = Np; g.sA = (long)mn;
    g.B = V; g.ldb = Np; g.sB = (long)mn;
    g.C = s.Am.p; g.ldc = Mp; g.sC = (long)mm;
    g.M = Mp; g.N = Mp; g.K = Np; g.alpha = 1.0; g.beta = 0.0;
    g.ksplit = FITC_KSPLIT; g.split_ws = s.ksplit_ws.p;
    g.tile_mode = 2;   // V V^T is symmetric: lower tiles + mirror images
    launch_gemm(st, g, false, true, E);
    launch_add_diag(st, s.Am.p, Mp, E, o_noise);
    launch_potrf(st, s.Am.p, Mp, E, s.AmInv.p, ctx->d_info + 32, true);  // Am = sn L
    launch_trtri(st, s.Am.p, Mp, E, s.AmInv.p, s.Tscr.p, (long)mm);
    g = GemmDesc{};                                          // iAt = Am^-1 Luu^-1 = L^-1 Luu^-1 / sn
    g.A = s.AmInv.p; g.lda = Mp; g.sA = (long)mm;
    g.B = s.Linv.p; g.ldb = Mp; g.sB = (long)mm;
    g.C = s.iAt.p; g.ldc = Mp; g.sC = (long)mm;
    g.M = Mp; g.N = Mp; g.K = Mp; g.alpha = 1.0; g.beta = 0.0; g.k_mode = 4;
    launch_gemm(st, g, false, false, E);
    launch_matvec(st, s.AmInv.p, Mp, E, r0, gam, false);
    launch_logdet(st, s.Am.p, Mp, M, E, sums + 3 * E);
    if (want_grad) {
        g = GemmDesc{};                                      // P = L^-1 U = AmInv Vb
        g.A = s.AmInv.p; g.lda = Mp; g.sA = (long)mm;
        g.B = V; g.ldb = Np; g.sB = (long)mn;
        g.C = s.ft_P.p; g.ldc = Np; g.sC = (long)mn;
        g.M = Mp; g.N = Np; g.K = Mp; g.alpha = 1.0; g.beta = 0.0; g.k_mode = 3;
        launch_gemm(st, g, false, false, E);
        hipLaunchKernelGGL(k_fitc_cols, dim3((Np + 63) / 64, E), dim3(256), 0, st, s.ft_P.p, gam, s.G.p, o_Yt, o_noise, M, Mp, N, Np, av, gv);
        g = GemmDesc{};                                      // A' = Luu^-T Vb   (A = Kuu^-1 Kuf = A' o G)
        g.A = s.Linv.p; g.lda = Mp; g.sA = (long)mm;
        g.B = V; g.ldb = Np; g.sB = (long)mn;
        g.C = s.Kmn.p; g.ldc = Np; g.sC = (long)mn;
        g.M = Mp; g.N = Np; g.K = Mp; g.alpha = 1.0; g.beta = 0.0;
        launch_gemm(st, g, true, false, E);
        hipLaunchKernelGGL(k_fitc_c, dim3((Mp + 3) / 4, E), dim3(256), 0, st, s.Kmn.p, s.G.p, av, Mp, N, Np, cv);
        g = GemmDesc{};                                      // T3 = iAt^T P   (A Sigma^-1 = T3 / G)
        g.A = s.iAt.p; g.lda = Mp; g.sA = (long)mm;
        g.B = s.ft_P.p; g.ldb = Np; g.sB = (long)mn;
        g.C = s.ft_T3.p; g.ldc = Np; g.sC = (long)mn;
        g.M = Mp; g.N = Np; g.K = Mp; g.alpha = 1.0; g.beta = 0.0;
        launch_gemm(st, g, true, false, E);
        hipLaunchKernelGGL(k_fitc_combine, dim3((Np + 255) / 256, Mp, E), dim3(256), 0, st, s.ft_T3.p, s.ft_P.p, s.Kmn.p, cv, av, gv, s.G.p, M,
                           Mp, N, Np);
        g = GemmDesc{};                                      // DKuu + 1/2 c c^T = W2g A'^T
        g.A = s.ft_P.p; g.lda = Np; g.sA = (long)mn;
        g.B = s.Kmn.p; g.ldb = Np; g.sB = (long)mn;
        g.C = s.iK.p; g.ldc = Mp; g.sC = (long)mm;
        g.M = Mp; g.N = Mp; g.K = Np; g.alpha = 1.0; g.beta = 0.0;
        g.ksplit = FITC_KSPLIT; g.split_ws = s.ksplit_ws.p;
        launch_gemm(st, g, false, true, E);
        if (kg_mfma) {
            launch_fitc_kgrad_mfma(D, dim3((M + 15) / 16, E, nsp), st, (const double*)s.ft_T3.p, Np, (const double*)Zt, Mp, (const double*)s.Xt.p, Np,
                                   0L, N, M, D, o_ls, o_var, (const double*)nullptr, 0.0, part_uf, E);
            launch_fitc_kgrad_mfma(D, dim3((M + 15) / 16, E, 1), st, (const double*)s.iK.p, Mp, (const double*)Zt, Mp, (const double*)Zt, Mp, sZ, M, M, D,
                                   o_ls, o_var, (const double*)cv, -0.5, part_uu, E);
        } else {
            launch_fitc_kgrad(D, dim3(M, E), st, (const double*)s.ft_T3.p, Np, (const double*)Zt, Mp, (const double*)s.Xt.p, Np, 0L, N, D,
                              o_ls, o_var, (const double*)nullptr, 0.0, part_uf);
            launch_fitc_kgrad(D, dim3(M, E), st, (const double*)s.iK.p, Mp, (const double*)Zt, Mp, (const double*)Zt, Mp, sZ, M, D, o_ls,
                              o_var, (const double*)cv, -0.5, part_uu);
        }
    } else {
        HIPCHK(hipMemsetAsync(gv, 0, sizeof(double) * (size_t)E * Np, st));
    }
    hipLaunchKernelGGL(k_fitc_sums, dim3(E), dim3(1024), 0, st, o_Yt, s.G.p, gv, N, Np, sums);
    if (want_grad)
        hipLaunchKernelGGL(k_fitc_kgrad_fin, dim3(E, 8), dim3(256), 0, st, (const double*)part_uf, nsp, (const double*)part_uu, (const double*)sums,
                           (const double*)gam, E, M, Mp, D, res);
    return PILCO_OK;
    };
    // everything the host needs in ONE pinned block: res [E][RS] | sums [4 E] | gam [E][Mp] (value-only calls) | noise [E] | info (64 ints)
    const size_t n_pin = (size_t)E * RS + 4 * (size_t)E + (size_t)E * Mp + E + 32 + 8 + (size_t)E * M * D;
    if (ctx->pin_cap < n_pin) {
        if (ctx->pin) (void)hipHostFree(ctx->pin);
        ctx->pin = nullptr;
        ctx->pin_cap = 0;
        HIPCHK(hipHostMalloc((void**)&ctx->pin, sizeof(double) * n_pin, hipHostMallocDefault));
        ctx->pin_cap = n_pin;
    }
    {   // the inducing inputs go up from pinned memory (a pageable source is a staged, blocking copy)
        double* h_z = ctx->pin + (n_pin - (size_t)E * M * D);
        memcpy(h_z, Z_all, sizeof(double) * (size_t)E * M * D);
        HIPCHK(hipMemcpyAsync(Zraw, h_z, sizeof(double) * (size_t)E * M * D, hipMemcpyHostToDevice, st));
    }
    if (int r = run_chain_graph(ctx, s.g_fitc_nlml, key, chain)) return r;
    double* h_res = ctx->pin;
    double* h_s = h_res + (size_t)E * RS;
    double* h_g = h_s + 4 * (size_t)E;
    double* h_n = h_g + (size_t)E * Mp;
    int* info = (int*)(h_n + E);
    if (want_grad) HIPCHK(hipMemcpyAsync(h_res, res, sizeof(double) * (size_t)E * RS, hipMemcpyDeviceToHost, st));
    else HIPCHK(hipMemcpyAsync(h_g, gam, sizeof(double) * (size_t)E * Mp, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(h_s, sums, sizeof(double) * 4 * E, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(h_n, o_noise, sizeof(double) * E, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(info, ctx->d_info, sizeof(int) * 64, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    HIPCHK(hipGetLastError());
    const double* hs = h_s;
    const double* hn = h_n;
    for (int e = 0; e < std::min(E, 32); ++e)
        if (info[e] != 0 || info[32 + e] != 0) {
            ctx->not_pd = e;
            return fail(ctx, PILCO_E_NOT_PD, "FITC objective: Cholesky failed for output " + std::to_string(e));
        }
    for (int e = 0; e < E; ++e) {
        const double sn2 = hn[e];
        double g2 = 0.0;
        if (want_grad) g2 = h_res[(size_t)e * RS + D + 2];
        else
            for (int m = 0; m < M; ++m) g2 += h_g[(size_t)e * Mp + m] * h_g[(size_t)e * Mp + m];
        const double f = -0.5 * hs[3 * e] / sn2 + 0.5 * g2 / sn2 - 0.5 * N * std::log(2.0 * M_PI) -
                         0.5 * (N * std::log(sn2) + 2.0 * hs[3 * e + 1]) - (hs[3 * E + e] - 0.5 * M * std::log(sn2));
        nlml[e] = -f;
        if (!want_grad) continue;
        const double* r = h_res + (size_t)e * RS;
        if (grad_hyp) {
            for (int d = 0; d < D; ++d) grad_hyp[(size_t)e * (D + 2) + d] = r[d];
            grad_hyp[(size_t)e * (D + 2) + D] = r[D];
            grad_hyp[(size_t)e * (D + 2) + D + 1] = r[D + 1];
        }
        if (grad_Z) memcpy(grad_Z + (size_t)e * M * D, r + D + 3, sizeof(double) * (size_t)M * D);
    }
    return PILCO_OK;
}

extern "C" int pilco_gp_fitc_nlml(pilco_ctx* ctx, int slot, const double* Z_all, int M, double* nlml, double* grad_hyp, double* grad_Z) {
    if (int r = check_slot(ctx, slot)) return r;
    Slot& s = ctx->slot[slot];
    if (!s.has_data || !s.has_hyp) return fail(ctx, PILCO_E_STATE, "fitc_nlml needs set_data and set_hyp first");
    if (!Z_all || M <= 0 || !nlml) return fail(ctx, PILCO_E_SHAPE, "fitc_nlml: bad arguments");
    if (s.D > FT_MAXD) return fail(ctx, PILCO_E_SHAPE, "fitc_nlml: D > 32");
    HIPCHK(hipSetDevice(ctx->device));
    const int E = s.E, D = s.D;
    if (ctx->nranks == 1) return fitc_nlml_batch(ctx, s, E, s.ls.p, s.var.p, s.noise.p, s.Yt.p, Z_all, M, nlml, grad_hyp, grad_Z);
    // Several ranks: sharded by output like the factorisation -- every output's GPRFITC model is its own problem
    // with its own inducing inputs (smgpr.py:16-22).  This rank evaluates the outputs a = rank, rank + W, ... from compacted
    // copies of their hyper-parameters and targets and writes them at their global places; one ncclAllGather completes every
    // rank's arrays when a communicator is attached, otherwise the other ranks' entries are NaN and the caller combines.
    OwnView o{};
    if (int r = prepare_own(ctx, s, o)) return r;
    const int EL = o.EL, W = o.W, rank = o.rank;
    const size_t per = 1 + (size_t)(D + 2) + (size_t)M * D;   // nlml | d hyp | d Z of one output
    const double nan = std::nan("");
    for (int a = 0; a < E; ++a) {
        nlml[a] = nan;
        if (grad_hyp) for (int k = 0; k < D + 2; ++k) grad_hyp[(size_t)a * (D + 2) + k] = nan;
        if (grad_Z) for (size_t k = 0; k < (size_t)M * D; ++k) grad_Z[(size_t)a * M * D + k] = nan;
    }
    std::vector<double> Zo((size_t)std::max(EL, 1) * M * D), no(std::max(EL, 1)), gho((size_t)std::max(EL, 1) * (D + 2)), gzo((size_t)std::max(EL, 1) * M * D);
    for (int al = 0; al < EL; ++al) memcpy(&Zo[(size_t)al * M * D], Z_all + (size_t)(al * W + rank) * M * D, sizeof(double) * M * D);
    int bad = -1, agreed = -1;
    if (EL > 0) {
        const int r = fitc_nlml_batch(ctx, s, EL, o.ls, o.var, o.noise, o.Yt, Zo.data(), M, no.data(), grad_hyp ? gho.data() : nullptr, grad_Z ? gzo.data() : nullptr);
        if (r == PILCO_E_NOT_PD && ctx->not_pd >= 0) bad = ctx->not_pd = ctx->not_pd * W + rank;   // local -> global output index
        else if (r) return r;
    }
    // the failure of one rank's output is every rank's failure, agreed on BEFORE the all-gather below (the peers would wait in it)
    if (int r = agree_not_pd(ctx, W, bad, &agreed)) return r;
    if (agreed >= 0) {
        ctx->not_pd = agreed;
        return fail(ctx, PILCO_E_NOT_PD, "FITC objective: Cholesky failed for output " + std::to_string(agreed));
    }
    std::vector<double> own((size_t)o.ELcap * per, 0.0);
    for (int al = 0; al < EL; ++al) {
        const int a = al * W + rank;
        nlml[a] = no[al];
        own[(size_t)al * per] = no[al];
        if (grad_hyp) {
            memcpy(grad_hyp + (size_t)a * (D + 2), &gho[(size_t)al * (D + 2)], sizeof(double) * (D + 2));
            memcpy(&own[(size_t)al * per + 1], &gho[(size_t)al * (D + 2)], sizeof(double) * (D + 2));
        }
        if (grad_Z) {
            memcpy(grad_Z + (size_t)a * M * D, &gzo[(size_t)al * M * D], sizeof(double) * M * D);
            memcpy(&own[(size_t)al * per + 1 + D + 2], &gzo[(size_t)al * M * D], sizeof(double) * M * D);
        }
    }
    if (ctx->comm) {
        const size_t blk = (size_t)o.ELcap * per;
        ENSURE(s.vec, (size_t)(W + 1) * blk);
        std::vector<double> all((size_t)W * blk);
        HIPCHK(hipMemcpyAsync(s.vec.p + (size_t)W * blk, own.data(), sizeof(double) * blk, hipMemcpyHostToDevice, ctx->st));
        ncclResult_t r = ncclAllGather(s.vec.p + (size_t)W * blk, s.vec.p, blk, ncclDouble, ctx->comm, ctx->st);
        if (r != ncclSuccess) return fail(ctx, PILCO_E_RCCL, std::string("ncclAllGather(fitc_nlml): ") + ncclGetErrorString(r));
        HIPCHK(hipMemcpyAsync(all.data(), s.vec.p, sizeof(double) * W * blk, hipMemcpyDeviceToHost, ctx->st));
        HIPCHK(hipStreamSynchronize(ctx->st));
        for (int a = 0; a < E; ++a) {
            const double* src = all.data() + ((size_t)(a % W) * o.ELcap + a / W) * per;
            nlml[a] = src[0];
            if (grad_hyp) memcpy(grad_hyp + (size_t)a * (D + 2), src + 1, sizeof(double) * (D + 2));
            if (grad_Z) memcpy(grad_Z + (size_t)a * M * D, src + 1 + D + 2, sizeof(double) * M * D);
        }
    }
    return PILCO_OK;
}
