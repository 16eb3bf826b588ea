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

namespace pilco {

// per column n < N (thread per column, coalesced over n): t_n = sum_m P_mn gamma_m, cn_n = sum_m P_mn^2, then
//   ytil = y / (sn G),  a_n = (ytil - t_n) / (sn G),  g_n = 1/2 (a_n^2 - (1 - cn_n) / (sn2 G^2))
__global__ __launch_bounds__(256) void k_fitc_cols(const double* __restrict__ P, const double* __restrict__ gam, const double* __restrict__ G,
                                                   const double* __restrict__ y, const double* __restrict__ noise, int M, int mpad,
                                                   int N, int npad, double* __restrict__ a_out, double* __restrict__ g_out) {
    const int b = blockIdx.y;
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= npad) return;
    double av = 0.0, gv = 0.0;
    if (n < N) {
        const double* Pb = P + (long)b * mpad * npad;
        double tn = 0.0, cn = 0.0;
        for (int m = 0; m < M; ++m) {
            const double p = Pb[(long)m * npad + n];
            tn = fma(p, gam[(long)b * mpad + m], tn);
            cn = fma(p, p, cn);
        }
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
    double s = 0.0;
    for (int n = lane; n < N; n += 64) s = fma(row[n] * G[(long)b * npad + n], a[(long)b * npad + n], s);
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

// sums for the value: out[b] = (sum (y/G)^2, sum log G, sum g)
__global__ __launch_bounds__(256) void k_fitc_sums(const double* __restrict__ y, const double* __restrict__ G, const double* __restrict__ g, int N,
                                                   int npad, double* __restrict__ out) {
    __shared__ double red[4][3];
    const int b = blockIdx.x, t = threadIdx.x;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    for (int n = t; n < N; n += 256) {
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
    if (t < 3) out[b * 3 + t] = (red[0][t] + red[1][t]) + (red[2][t] + red[3][t]);
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
    ENSURE(s.Tscr, std::max(E * mm, (size_t)E * Mp * (2 * FT_MAXD + 1) * 2));
    ENSURE(s.ft_P, E * mn);
    ENSURE(s.ft_T3, E * mn);
    ENSURE(s.ft_Z, (size_t)E * D * Mp + (size_t)E * M * D);
    ENSURE(s.vec, (size_t)E * (4 * (size_t)std::max(Mp, Np) + 8));
    double* Zt = s.ft_Z.p;                       // [E][D][Mp]
    double* Zraw = Zt + (size_t)E * D * Mp;      // [E][M][D] staging
    HIPCHK(hipMemcpyAsync(Zraw, Z_all, sizeof(double) * (size_t)E * M * D, hipMemcpyHostToDevice, st));
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
    double* part_uf = s.Tscr.p;
    double* part_uu = part_uf + (size_t)E * Mp * (2 * FT_MAXD + 1);
    // everything between the upload of Z and the downloads is a fixed launch sequence (~80 launches at M = 200): one graph
    std::vector<unsigned long long> key;
    for (const DevBuf* b : {&s.K, &s.Linv, &s.Kmn, &s.V2, &s.Am, &s.AmInv, &s.iAt, &s.ksplit_ws, &s.iK, &s.G, &s.Tscr, &s.ft_P,
                            &s.ft_T3, &s.ft_Z, &s.vec, &s.Xt})
        key.push_back((unsigned long long)(uintptr_t)b->p);
    for (const void* q : {(const void*)o_ls, (const void*)o_var, (const void*)o_noise, (const void*)o_Yt, (const void*)ctx->d_info})
        key.push_back((unsigned long long)(uintptr_t)q);
    for (int v : {E, M, Mp, N, Np, D, want_grad ? 1 : 0}) key.push_back((unsigned long long)v);
    auto chain = [&]() -> int {
    for (int e = 0; e < E; ++e) launch_transpose_points(st, Zraw + (size_t)e * M * D, M, D, Zt + (size_t)e * D * Mp, Mp);
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
    launch_fitc_scale(st, V, Mp, Np, E, o_var, o_noise, s.G.p);     // G = sqrt(nu) / sn, V <- Vb = V / G
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
    launch_fitc_rhs(st, V, s.G.p, o_Yt, Mp, Np, E, r0);
    launch_matvec(st, s.AmInv.p, Mp, E, r0, gam, false);
    launch_logdet(st, s.Am.p, Mp, M, E, sums + 3 * E);
    if (want_grad) {
        g = GemmDesc{};                                      // P = L^-1 U = AmInv Vb
        g.A = s.AmInv.p; g.lda = Mp; g.sA = (long)mm;
        g.B = V; g.ldb = Np; g.sB = (long)mn;
        g.C = s.ft_P.p; g.ldc = Np; g.sC = (long)mn;
        g.M = Mp; g.N = Np; g.K = Mp; g.alpha = 1.0; g.beta = 0.0; g.k_mode = 3;
        launch_gemm(st, g, false, false, E);
        hipLaunchKernelGGL(k_fitc_cols, dim3((Np + 255) / 256, E), dim3(256), 0, st, s.ft_P.p, gam, s.G.p, o_Yt, o_noise, M, Mp, N, Np, av, gv);
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
        launch_fitc_kgrad(D, dim3(M, E), st, (const double*)s.ft_T3.p, Np, (const double*)Zt, Mp, (const double*)s.Xt.p, Np, 0L, N, D,
                          o_ls, o_var, (const double*)nullptr, 0.0, part_uf);
        launch_fitc_kgrad(D, dim3(M, E), st, (const double*)s.iK.p, Mp, (const double*)Zt, Mp, (const double*)Zt, Mp, sZ, M, D, o_ls,
                          o_var, (const double*)cv, -0.5, part_uu);
    } else {
        HIPCHK(hipMemsetAsync(gv, 0, sizeof(double) * (size_t)E * Np, st));
    }
    hipLaunchKernelGGL(k_fitc_sums, dim3(E), dim3(256), 0, st, o_Yt, s.G.p, gv, N, Np, sums);
    return PILCO_OK;
    };
    if (int r = run_chain_graph(ctx, s.g_fitc_nlml, key, chain)) return r;
    std::vector<double> hz;
    if (want_grad) {
        hz.resize((size_t)2 * E * Mp * (2 * FT_MAXD + 1));
        HIPCHK(hipMemcpyAsync(hz.data(), part_uf, sizeof(double) * hz.size(), hipMemcpyDeviceToHost, st));
    }
    std::vector<double> hs(4 * (size_t)E), hg((size_t)E * Mp), hn(E);
    int info[64];
    HIPCHK(hipMemcpyAsync(hs.data(), sums, sizeof(double) * 4 * E, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(hg.data(), gam, sizeof(double) * (size_t)E * Mp, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(hn.data(), o_noise, sizeof(double) * E, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(info, ctx->d_info, sizeof(int) * 64, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    HIPCHK(hipGetLastError());
    for (int e = 0; e < std::min(E, 32); ++e)
        if (info[e] != 0 || info[32 + e] != 0) {
            ctx->not_pd = e;
            return fail(ctx, PILCO_E_NOT_PD, "FITC objective: Cholesky failed for output " + std::to_string(e));
        }
    const int PW = 2 * FT_MAXD + 1;
    for (int e = 0; e < E; ++e) {
        const double sn2 = hn[e];
        double g2 = 0.0;
        for (int m = 0; m < M; ++m) g2 += hg[(size_t)e * Mp + m] * hg[(size_t)e * Mp + m];
        const double f = -0.5 * hs[3 * e] / sn2 + 0.5 * g2 / sn2 - 0.5 * N * std::log(2.0 * M_PI) -
                         0.5 * (N * std::log(sn2) + 2.0 * hs[3 * e + 1]) - (hs[3 * E + e] - 0.5 * M * std::log(sn2));
        nlml[e] = -f;
        if (!want_grad) continue;
        const double* uf = hz.data() + (size_t)e * Mp * PW;
        const double* uu = hz.data() + (size_t)(E + e) * Mp * PW;
        const double sg = hs[3 * e + 2];
        if (grad_hyp) {
            for (int d = 0; d < D; ++d) {
                double acc = 0.0;
                for (int m = 0; m < M; ++m) acc += uf[(size_t)m * PW + D + d] + uu[(size_t)m * PW + D + d];
                grad_hyp[(size_t)e * (D + 2) + d] = -acc;
            }
            double accv = sg;
            for (int m = 0; m < M; ++m) accv += uf[(size_t)m * PW + 2 * D] + uu[(size_t)m * PW + 2 * D];
            grad_hyp[(size_t)e * (D + 2) + D] = -accv;
            grad_hyp[(size_t)e * (D + 2) + D + 1] = -sg;
        }
        if (grad_Z)
            for (int m = 0; m < M; ++m)
                for (int d = 0; d < D; ++d)
                    grad_Z[((size_t)e * M + m) * D + d] = -(uf[(size_t)m * PW + d] + 2.0 * uu[(size_t)m * PW + d]);
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
