// Reverse mode: the VJP of one moment-matching step (device) and the native reverse sweep of a rollout
// (DESIGN.md section 9).
#include "ctx.h"

extern "C" {

// Vector-Jacobian product of one moment-matching step (the reverse of pilco_gp_predict):
// given cotangents Mbar (1,E), Sbar (E,E), Vbar (D,E) returns mbar (1,D) and the symmetric sbar (D,D).
// Entirely on the device (k_mm_bwd_pair / _post / _fin; the mean part rides in extra workgroups of _post / _fin);
// the E + P contribution records are summed in a fixed order by the last workgroup to finish.  Single rank, exact or sparse model, D <= 32 (the forward path's limit).
int pilco_gp_predict_vjp(pilco_ctx* ctx, int slot, const double* m, const double* s_in, const double* Mbar,
                         const double* Sbar, const double* Vbar, double* mbar, double* sbar) {
    if (int r = check_slot(ctx, slot)) return r;
    Slot& s = ctx->slot[slot];
    if (!s.factor_valid) return fail(ctx, PILCO_E_STATE, "predict_vjp: no current factorisation");
    if (!m || !s_in || !Mbar || !Sbar || !Vbar || !mbar || !sbar) return fail(ctx, PILCO_E_SHAPE, "predict_vjp: null pointer");
    if (ctx->nranks != 1) return fail(ctx, PILCO_E_STATE, "predict_vjp: single rank only");
    const int D = s.D, E = s.E, npad = s.npad;
    if (D > MAX_D) return fail(ctx, PILCO_E_SHAPE, "predict_vjp: D <= 32 in this build");
    HIPCHK(hipSetDevice(ctx->device));
    if (int r = build_work(ctx, s)) return r;
    const int P = s.wk.PL;
    // ---- device: operands (prep), reverse pair sweep, mean part, per-pair / per-output contributions
    const int rec = D + D * D, nb = E + E * E + D * E;
    int njs, nrb;
    mm_bwd_geometry(npad, E * (E + 1) / 2, &njs, &nrb);
    ENSURE(s.bwd_mom, mm_bwd_gpart_size(npad, P, D));   // one (16 NMT)^2 block per sweep workgroup
    ENSURE(s.bwd_cp, mm_bwd_cpart_size(npad, P));
    ENSURE(s.bwd_part, (size_t)(P + E) * mm_bwd_rc(npad) * (1 + rec + D));   // pair partials, then mean partials
    ENSURE(s.bwd_out, (size_t)(E + P) * rec + (size_t)(E + P) * (D * D + D + 2));   // contributions | head records
    if (!s.bwd_cnt.p) {
        ENSURE(s.bwd_cnt, 2);
        HIPCHK(hipMemsetAsync(s.bwd_cnt.p, 0, 2 * sizeof(double), ctx->st));
    }
    const size_t n_in = (size_t)D + D * D + nb, n_out = (size_t)rec;
    if (ctx->pin_cap < n_in + n_out) {
        if (ctx->pin) (void)hipHostFree(ctx->pin);
        ctx->pin = nullptr;
        ctx->pin_cap = 0;
        HIPCHK(hipHostMalloc((void**)&ctx->pin, sizeof(double) * (n_in + n_out), hipHostMallocDefault));
        ctx->pin_cap = n_in + n_out;
    }
    double* hin = ctx->pin;
    const double* po = ctx->pin + n_in;
    memcpy(hin, m, sizeof(double) * D);
    memcpy(hin + D, s_in, sizeof(double) * D * D);
    memcpy(hin + D + D * D, Mbar, sizeof(double) * E);
    memcpy(hin + D + D * D + E, Sbar, sizeof(double) * E * E);
    memcpy(hin + D + D * D + E + E * E, Vbar, sizeof(double) * D * E);
    HIPCHK(hipMemcpyAsync(s.wk.in_m, hin, sizeof(double) * n_in, hipMemcpyHostToDevice, ctx->st));   // in_m | in_s | bars are contiguous
    const double* bars = s.wk.in_s + D * D;
    const MMModel md = model_of(s);
    launch_mm_prep(ctx->st, md, s.wk);
    // the last workgroup of k_mm_bwd_fin writes the summed record straight into the pinned host buffer
    launch_mm_bwd(ctx->st, md, s.wk, s.bwd_mom.p, s.bwd_cp.p, s.bwd_part.p, bars, s.bwd_out.p + (size_t)(E + P) * rec, s.bwd_out.p,
                  (unsigned*)s.bwd_cnt.p, ctx->pin + n_in);
    HIPCHK(hipStreamSynchronize(ctx->st));
    HIPCHK(hipGetLastError());
    // ---- host: symmetrise
    const double* acc = po;
    for (int e = 0; e < rec; ++e)
        if (!std::isfinite(acc[e])) return fail(ctx, PILCO_E_NOT_PD, "predict_vjp: singular s + Lambda^2 or I + Lambda s");
    for (int d = 0; d < D; ++d) mbar[d] = acc[d];
    for (int r = 0; r < D; ++r)
        for (int c = 0; c < D; ++c) sbar[(size_t)r * D + c] = 0.5 * (acc[D + (size_t)r * D + c] + acc[D + (size_t)c * D + r]);
    return PILCO_OK;
}

// ------------------------------------------------------------------ native reverse sweep (policy gradient)
}  // extern "C"

namespace {

typedef std::vector<double> vec;

// inverse and determinant of a small dense matrix (partial pivoting); false if singular
bool inv_small(const double* A, int n, vec& inv, double& det) {
    vec a(A, A + (size_t)n * n);
    inv.assign((size_t)n * n, 0.0);
    for (int i = 0; i < n; ++i) inv[(size_t)i * n + i] = 1.0;
    det = 1.0;
    for (int k = 0; k < n; ++k) {
        int p = k;
        for (int r = k + 1; r < n; ++r)
            if (std::fabs(a[(size_t)r * n + k]) > std::fabs(a[(size_t)p * n + k])) p = r;
        const double piv = a[(size_t)p * n + k];
        if (piv == 0.0 || !std::isfinite(piv)) return false;
        if (p != k) {
            for (int c = 0; c < n; ++c) {
                std::swap(a[(size_t)p * n + c], a[(size_t)k * n + c]);
                std::swap(inv[(size_t)p * n + c], inv[(size_t)k * n + c]);
            }
            det = -det;
        }
        det *= piv;
        const double ip = 1.0 / piv;
        for (int c = 0; c < n; ++c) {
            a[(size_t)k * n + c] *= ip;
            inv[(size_t)k * n + c] *= ip;
        }
        for (int r = 0; r < n; ++r) {
            if (r == k) continue;
            const double f = a[(size_t)r * n + k];
            if (f == 0.0) continue;
            for (int c = 0; c < n; ++c) {
                a[(size_t)r * n + c] -= f * a[(size_t)k * n + c];
                inv[(size_t)r * n + c] -= f * inv[(size_t)k * n + c];
            }
        }
    }
    return true;
}

// squash_sin (controllers.py:13-36) forward quantities and its vector-Jacobian product (derivatives as in gSin.m:50-74)
struct Squash {
    int U;
    vec M, Cd, S, q, Ep, Em, dm, sm, ee;
    void fwd(const vec& mu0, const vec& su0, const vec& e) {
        U = (int)mu0.size();
        M.resize(U); Cd.resize(U);
        S.resize((size_t)U * U); q = Ep = Em = dm = sm = ee = S;
        for (int u = 0; u < U; ++u) {
            const double ex = std::exp(-su0[(size_t)u * U + u] / 2.0);
            M[u] = e[u] * ex * std::sin(mu0[u]);
            Cd[u] = e[u] * ex * std::cos(mu0[u]);
        }
        for (int u = 0; u < U; ++u)
            for (int v = 0; v < U; ++v) {
                const size_t k = (size_t)u * U + v;
                const double lq = -(su0[(size_t)u * U + u] + su0[(size_t)v * U + v]) / 2.0;
                q[k] = std::exp(lq);
                Ep[k] = std::exp(lq + su0[k]);
                Em[k] = std::exp(lq - su0[k]);
                dm[k] = mu0[u] - mu0[v];
                sm[k] = mu0[u] + mu0[v];
                ee[k] = e[u] * e[v];
                S[k] = ee[k] / 2.0 * ((Ep[k] - q[k]) * std::cos(dm[k]) - (Em[k] - q[k]) * std::cos(sm[k]));
            }
    }
    void vjp(const double* Mbar, const double* Sbar, const double* Cdbar, vec& mubar, vec& subar) const {
        mubar.assign(U, 0.0);
        subar.assign((size_t)U * U, 0.0);
        for (int u = 0; u < U; ++u) {
            double acc = Mbar[u] * Cd[u] - Cdbar[u] * M[u];
            double dd = -0.5 * Mbar[u] * M[u] - 0.5 * Cdbar[u] * Cd[u];
            for (int v = 0; v < U; ++v) {
                const size_t uv = (size_t)u * U + v, vu = (size_t)v * U + u;
                const double D1 = ee[uv] / 2.0 * (-(Ep[uv] - q[uv]) * std::sin(dm[uv]) + (Em[uv] - q[uv]) * std::sin(sm[uv]));
                const double D2 = ee[vu] / 2.0 * ((Ep[vu] - q[vu]) * std::sin(dm[vu]) + (Em[vu] - q[vu]) * std::sin(sm[vu]));
                acc += Sbar[uv] * D1 + Sbar[vu] * D2;
                dd -= 0.5 * (Sbar[uv] * S[uv] + Sbar[vu] * S[vu]);
                subar[uv] = Sbar[uv] * (ee[uv] / 2.0 * (Ep[uv] * std::cos(dm[uv]) + Em[uv] * std::cos(sm[uv])));
            }
            mubar[u] = acc;
            subar[(size_t)u * U + u] += dd;
        }
    }
};

// d muR / d m, d muR / d S of the reward terms (rewards.py:19-81; formulas of reward.m:47-50), accumulated into dm, dS
bool reward_grad(const pilco_reward_term* rw, int n_rw, int E, const double* m, const double* S, vec& dm, vec& dS) {
    vec A((size_t)E * E), Ai, iSpW((size_t)E * E), d(E), v(E);
    for (int k = 0; k < n_rw; ++k) {
        const double c = rw[k].coef;
        if (rw[k].kind == PILCO_REWARD_LINEAR) {
            for (int i = 0; i < E; ++i) dm[i] += c * rw[k].W[i];
            continue;
        }
        const double* W = rw[k].W;
        auto Wv = [&](int i, int j) { return W ? W[(size_t)i * E + j] : (i == j ? 1.0 : 0.0); };
        for (int i = 0; i < E; ++i) {
            d[i] = m[i] - (rw[k].t ? rw[k].t[i] : 0.0);
            for (int j = 0; j < E; ++j) {
                double acc = (i == j) ? 1.0 : 0.0;
                for (int l = 0; l < E; ++l) acc += S[(size_t)i * E + l] * Wv(l, j);
                A[(size_t)i * E + j] = acc;   // I + S W
            }
        }
        double det;
        if (!inv_small(A.data(), E, Ai, det)) return false;
        for (int i = 0; i < E; ++i)
            for (int j = 0; j < E; ++j) {
                double acc = 0.0;
                for (int l = 0; l < E; ++l) acc += Wv(i, l) * Ai[(size_t)l * E + j];
                iSpW[(size_t)i * E + j] = acc;   // W (I + S W)^-1
            }
        double quad = 0.0;
        for (int i = 0; i < E; ++i) {
            double acc = 0.0;
            for (int j = 0; j < E; ++j) acc += iSpW[(size_t)i * E + j] * d[j];
            v[i] = acc;          // iSpW d
            quad += d[i] * acc;
        }
        const double muR = std::exp(-0.5 * quad) / std::sqrt(det);
        vec dTi(E, 0.0);   // d^T iSpW
        for (int j = 0; j < E; ++j)
            for (int i = 0; i < E; ++i) dTi[j] += d[i] * iSpW[(size_t)i * E + j];
        for (int j = 0; j < E; ++j) dm[j] -= c * muR * dTi[j];
        // dS = muR (iSpW d d^T - I) iSpW / 2, symmetrised
        vec T((size_t)E * E);
        for (int i = 0; i < E; ++i)
            for (int j = 0; j < E; ++j) T[(size_t)i * E + j] = 0.5 * muR * (v[i] * dTi[j] - iSpW[(size_t)i * E + j]);
        for (int i = 0; i < E; ++i)
            for (int j = 0; j < E; ++j) dS[(size_t)i * E + j] += c * 0.5 * (T[(size_t)i * E + j] + T[(size_t)j * E + i]);
    }
    return true;
}

// ---- small dense matrices for the host side of the reverse sweep
struct Mat {
    int r = 0, c = 0;
    vec d;
    Mat() {}
    Mat(int r_, int c_) : r(r_), c(c_), d((size_t)r_ * c_, 0.0) {}
    double& operator()(int i, int j) { return d[(size_t)i * c + j]; }
    double operator()(int i, int j) const { return d[(size_t)i * c + j]; }
};
Mat mm(const Mat& A, const Mat& B) {        // A B
    Mat C(A.r, B.c);
    for (int i = 0; i < A.r; ++i)
        for (int k = 0; k < A.c; ++k) {
            const double a = A(i, k);
            for (int j = 0; j < B.c; ++j) C(i, j) += a * B(k, j);
        }
    return C;
}
Mat mmT(const Mat& A, const Mat& B) {       // A B^T
    Mat C(A.r, B.r);
    for (int i = 0; i < A.r; ++i)
        for (int j = 0; j < B.r; ++j) {
            double acc = 0.0;
            for (int k = 0; k < A.c; ++k) acc += A(i, k) * B(j, k);
            C(i, j) = acc;
        }
    return C;
}
Mat Tmm(const Mat& A, const Mat& B) {       // A^T B
    Mat C(A.c, B.c);
    for (int k = 0; k < A.r; ++k)
        for (int i = 0; i < A.c; ++i) {
            const double a = A(k, i);
            for (int j = 0; j < B.c; ++j) C(i, j) += a * B(k, j);
        }
    return C;
}

// Adjoint of a policy inside the sweep: fwd gives the pre-squash action moments (mu0 (U), su0 (U,U), V0 (E,U)) at
// (m_x, s_x); vjp takes their cotangents, accumulates the parameter gradients and returns the cotangents of (m_x, s_x).
struct PolicyAdj {
    virtual ~PolicyAdj() {}
    virtual bool fwd(const double* m_x, const double* s_x, vec& mu0, vec& su0, vec& V0) = 0;
    virtual void vjp(const double* m_x, const double* s_x, const vec& mu0b, const vec& su0b, const vec& V0b, vec& mxb, vec& sxb) = 0;
};

// LinearController (controllers.py:46-58): mu0 = W m + b, su0 = W s W^T, V0 = W^T
struct LinearAdj : PolicyAdj {
    int E, U;
    const double* W;
    const double* b;
    vec Wbar, bbar;
    LinearAdj(int E_, int U_, const double* W_, const double* b_) : E(E_), U(U_), W(W_), b(b_), Wbar((size_t)U_ * E_, 0.0), bbar(U_, 0.0) {}
    bool fwd(const double* m_x, const double* s_x, vec& mu0, vec& su0, vec& V0) override {
        vec WS((size_t)U * E);
        for (int u = 0; u < U; ++u) {
            double acc = b[u];
            for (int i = 0; i < E; ++i) acc += W[(size_t)u * E + i] * m_x[i];
            mu0[u] = acc;
            for (int j = 0; j < E; ++j) {
                double a2 = 0.0;
                for (int i = 0; i < E; ++i) a2 += W[(size_t)u * E + i] * s_x[(size_t)i * E + j];
                WS[(size_t)u * E + j] = a2;
            }
        }
        for (int u = 0; u < U; ++u)
            for (int v = 0; v < U; ++v) {
                double acc = 0.0;
                for (int j = 0; j < E; ++j) acc += WS[(size_t)u * E + j] * W[(size_t)v * E + j];
                su0[(size_t)u * U + v] = acc;
            }
        for (int i = 0; i < E; ++i)
            for (int u = 0; u < U; ++u) V0[(size_t)i * U + u] = W[(size_t)u * E + i];
        return true;
    }
    void vjp(const double* m_x, const double* s_x, const vec& mu0b, const vec& su0b, const vec& V0b, vec& mxb, vec& sxb) override {
        // Wbar += V0b^T + mu0b m_x^T + su0b W s_x^T + su0b^T W s_x
        vec T1((size_t)U * E), T2((size_t)U * E);
        for (int u = 0; u < U; ++u)
            for (int j = 0; j < E; ++j) {
                double a1 = 0.0, a2 = 0.0;
                for (int i = 0; i < E; ++i) {
                    a1 += W[(size_t)u * E + i] * s_x[(size_t)j * E + i];     // (W s_x^T)[u][j]
                    a2 += W[(size_t)u * E + i] * s_x[(size_t)i * E + j];     // (W s_x)[u][j]
                }
                T1[(size_t)u * E + j] = a1;
                T2[(size_t)u * E + j] = a2;
            }
        for (int u = 0; u < U; ++u) {
            bbar[u] += mu0b[u];
            for (int j = 0; j < E; ++j) {
                double acc = V0b[(size_t)j * U + u] + mu0b[u] * m_x[j];
                for (int v = 0; v < U; ++v)
                    acc += su0b[(size_t)u * U + v] * T1[(size_t)v * E + j] + su0b[(size_t)v * U + u] * T2[(size_t)v * E + j];
                Wbar[(size_t)u * E + j] += acc;
            }
        }
        for (int i = 0; i < E; ++i) {
            double acc = 0.0;
            for (int u = 0; u < U; ++u) acc += W[(size_t)u * E + i] * mu0b[u];
            mxb[i] += acc;                                                   // W^T mu0b
            for (int j = 0; j < E; ++j) {
                double a2 = 0.0;
                for (int u = 0; u < U; ++u)
                    for (int v = 0; v < U; ++v) a2 += W[(size_t)u * E + i] * su0b[(size_t)u * U + v] * W[(size_t)v * E + j];
                sxb[(size_t)i * E + j] += a2;                                // W^T su0b W
            }
        }
    }
};

// RbfController = deterministic GP (controllers.py:80-121 calling mgpr.py:91-149 with iK = 0, unit signal variance,
// S -= diag(var - 1e-6)): the same line-by-line reverse as pilco_amd/adjoint.py (rbf_policy_fwd / rbf_policy_vjp), which
// is checked against autograd on the CPU (tests/test_oracle.py); beta = (K + noise I)^-1 Y and the Gram matrices do not
// change along the sweep, so their cotangent is accumulated and pushed through the solve once at the end.
struct RbfAdj : PolicyAdj {
    int n, d, U;
    Mat X, Y, ls;
    vec noise;
    std::vector<Mat> K, Ainv;
    Mat beta;                         // (U, n)
    Mat Xbar, lsbar, bbsum;           // accumulated cotangents of X (direct part), ls (direct part) and beta
    // per-step cache
    Mat zeta;
    vec Mv;
    struct MeanC { Mat T, G; vec ex, q; };
    struct PairC { Mat z, w, Ri, Q, zQ, wQ, L; double r, val; };
    std::vector<MeanC> mean;
    std::vector<PairC> pair;
    vec s_cur;
    bool init(int n_, int d_, int U_, const double* Xp, const double* Yp, const double* lsp, const double* nz) {
        n = n_; d = d_; U = U_;
        X = Mat(n, d); Y = Mat(n, U); ls = Mat(U, d);
        X.d.assign(Xp, Xp + (size_t)n * d);
        Y.d.assign(Yp, Yp + (size_t)n * U);
        ls.d.assign(lsp, lsp + (size_t)U * d);
        noise.assign(nz, nz + U);
        beta = Mat(U, n); Xbar = Mat(n, d); lsbar = Mat(U, d); bbsum = Mat(U, n);
        K.resize(U); Ainv.resize(U);
        for (int a = 0; a < U; ++a) {
            K[a] = Mat(n, n);
            Mat A(n, n);
            for (int i = 0; i < n; ++i)
                for (int j = 0; j < n; ++j) {
                    double q = 0.0;
                    for (int k = 0; k < d; ++k) {
                        const double t = (X(i, k) - X(j, k)) / ls(a, k);
                        q += t * t;
                    }
                    K[a](i, j) = std::exp(-0.5 * q);
                    A(i, j) = K[a](i, j) + (i == j ? noise[a] : 0.0);   // FakeGPR likelihood variance, controllers.py:67-77
                }
            double det;
            Ainv[a] = Mat(n, n);
            if (!inv_small(A.d.data(), n, Ainv[a].d, det)) return false;
            for (int i = 0; i < n; ++i) {
                double acc = 0.0;
                for (int j = 0; j < n; ++j) acc += Ainv[a](i, j) * Y(j, a);
                beta(a, i) = acc;
            }
        }
        mean.resize(U);
        pair.resize((size_t)U * U);
        return true;
    }
    bool fwd(const double* m_x, const double* s_x, vec& mu0, vec& su0, vec& V0) override {
        s_cur.assign(s_x, s_x + (size_t)d * d);
        zeta = Mat(n, d);
        for (int i = 0; i < n; ++i)
            for (int k = 0; k < d; ++k) zeta(i, k) = X(i, k) - m_x[k];
        Mv.assign(U, 0.0);
        for (int a = 0; a < U; ++a) {
            MeanC& mc = mean[a];
            Mat A(d, d);
            double sumlog = 0.0;
            for (int i = 0; i < d; ++i) {
                for (int j = 0; j < d; ++j) A(i, j) = s_x[(size_t)i * d + j] + (i == j ? ls(a, i) * ls(a, i) : 0.0);
                sumlog += std::log(ls(a, i));
            }
            double det;
            mc.T = Mat(d, d);
            if (!inv_small(A.d.data(), d, mc.T.d, det) || !(det > 0.0)) return false;
            mc.G = mm(zeta, mc.T);
            const double logc = -0.5 * (std::log(det) - 2.0 * sumlog);
            mc.ex.assign(n, 0.0);
            mc.q.assign(n, 0.0);
            double Msum = 0.0;
            for (int i = 0; i < n; ++i) {
                double h = 0.0;
                for (int k = 0; k < d; ++k) h += zeta(i, k) * mc.G(i, k);
                mc.ex[i] = std::exp(-0.5 * h + logc);
                mc.q[i] = beta(a, i) * mc.ex[i];
                Msum += mc.q[i];
            }
            Mv[a] = Msum;
            mu0[a] = Msum;
            for (int k = 0; k < d; ++k) {
                double acc = 0.0;
                for (int i = 0; i < n; ++i) acc += mc.G(i, k) * mc.q[i];
                V0[(size_t)k * U + a] = acc;
            }
        }
        for (int a = 0; a < U; ++a)
            for (int b = 0; b < U; ++b) {
                PairC& pc = pair[(size_t)a * U + b];
                pc.z = Mat(n, d); pc.w = Mat(n, d);
                vec ka(n, 0.0), kb(n, 0.0);
                Mat R(d, d);
                for (int i = 0; i < n; ++i)
                    for (int k = 0; k < d; ++k) {
                        pc.z(i, k) = zeta(i, k) / (ls(a, k) * ls(a, k));
                        pc.w(i, k) = zeta(i, k) / (ls(b, k) * ls(b, k));
                        ka[i] -= 0.5 * zeta(i, k) * pc.z(i, k);
                        kb[i] -= 0.5 * zeta(i, k) * pc.w(i, k);
                    }
                for (int i = 0; i < d; ++i)
                    for (int j = 0; j < d; ++j)
                        R(i, j) = s_x[(size_t)i * d + j] * (1.0 / (ls(a, j) * ls(a, j)) + 1.0 / (ls(b, j) * ls(b, j))) + (i == j ? 1.0 : 0.0);
                double det;
                pc.Ri = Mat(d, d);
                if (!inv_small(R.d.data(), d, pc.Ri.d, det) || !(det > 0.0)) return false;
                Mat sM(d, d);
                sM.d.assign(s_x, s_x + (size_t)d * d);
                pc.Q = mm(pc.Ri, sM);
                for (double& v : pc.Q.d) v *= 0.5;
                pc.zQ = mm(pc.z, pc.Q);
                pc.wQ = mm(pc.w, pc.Q);
                vec uu(n), vv(n);
                for (int i = 0; i < n; ++i) {
                    double a1 = ka[i], a2 = kb[i];
                    for (int k = 0; k < d; ++k) {
                        a1 += pc.zQ(i, k) * pc.z(i, k);
                        a2 += pc.wQ(i, k) * pc.w(i, k);
                    }
                    uu[i] = a1;
                    vv[i] = a2;
                }
                pc.L = mmT(pc.zQ, pc.w);
                double val = 0.0;
                for (int i = 0; i < n; ++i)
                    for (int j = 0; j < n; ++j) {
                        const double l = std::exp(uu[i] + vv[j] + 2.0 * pc.L(i, j));
                        pc.L(i, j) = l;
                        val += beta(a, i) * l * beta(b, j);
                    }
                pc.r = 1.0 / std::sqrt(det);
                pc.val = val;
                su0[(size_t)a * U + b] = val * pc.r - Mv[a] * Mv[b] + (a == b ? 1e-6 : 0.0);   // + var - (var - 1e-6), controllers.py:117
            }
        return true;
    }
    void vjp(const double*, const double*, const vec& mu0b, const vec& su0b, const vec& V0b, vec& mxb, vec& sxb) override {
        vec Mbar(mu0b);
        Mat zb(n, d), sb(d, d), ibar(U, d), bb(U, n);
        Mat sM(d, d);
        sM.d = s_cur;
        for (int a = 0; a < U; ++a)
            for (int b = 0; b < U; ++b) {
                const double g = su0b[(size_t)a * U + b];
                if (g == 0.0) continue;
                const PairC& pc = pair[(size_t)a * U + b];
                Mbar[a] -= g * Mv[b];
                Mbar[b] -= g * Mv[a];
                const double valb = g * pc.r, ldb = -0.5 * g * pc.val * pc.r;   // ldb: cotangent of log det R
                Mat Eb(n, n);
                vec ub(n, 0.0), vb(n, 0.0);
                for (int i = 0; i < n; ++i)
                    for (int j = 0; j < n; ++j) {
                        const double l = pc.L(i, j);
                        bb(a, i) += valb * l * beta(b, j);
                        bb(b, j) += valb * l * beta(a, i);
                        const double e = valb * beta(a, i) * beta(b, j) * l;
                        Eb(i, j) = e;
                        ub[i] += e;
                        vb[j] += e;
                    }
                Mat zQb = mm(Eb, pc.w);          // 2 Eb w + ub z
                Mat wb = Tmm(Eb, pc.zQ);         // 2 Eb^T zQ + vb wQ
                Mat zb_(n, d), wQb(n, d);
                for (int i = 0; i < n; ++i)
                    for (int k = 0; k < d; ++k) {
                        zQb(i, k) = 2.0 * zQb(i, k) + ub[i] * pc.z(i, k);
                        wb(i, k) = 2.0 * wb(i, k) + vb[i] * pc.wQ(i, k);
                        zb_(i, k) = ub[i] * pc.zQ(i, k);
                        wQb(i, k) = vb[i] * pc.w(i, k);
                    }
                const Mat t1 = mmT(zQb, pc.Q), t2 = mmT(wQb, pc.Q);
                Mat Qb = Tmm(pc.z, zQb);
                const Mat Qb2 = Tmm(pc.w, wQb);
                for (size_t e = 0; e < Qb.d.size(); ++e) Qb.d[e] += Qb2.d[e];
                for (int i = 0; i < n; ++i)
                    for (int k = 0; k < d; ++k) {
                        const double zbk = zb_(i, k) + t1(i, k) - 0.5 * ub[i] * zeta(i, k);
                        const double wbk = wb(i, k) + t2(i, k) - 0.5 * vb[i] * zeta(i, k);
                        const double ia = 1.0 / (ls(a, k) * ls(a, k)), ib = 1.0 / (ls(b, k) * ls(b, k));
                        zb(i, k) += -0.5 * ub[i] * pc.z(i, k) - 0.5 * vb[i] * pc.w(i, k) + zbk * ia + wbk * ib;
                        ibar(a, k) += zbk * zeta(i, k);
                        ibar(b, k) += wbk * zeta(i, k);
                    }
                // Rb = -Ri^T Qb Q^T + ldb Ri^T;  sb += Ri^T Qb / 2 + Rb diag(ia + ib);  lam = colsum(s o Rb)
                const Mat RiTQb = Tmm(pc.Ri, Qb);
                Mat Rb = mmT(RiTQb, pc.Q);
                for (int i = 0; i < d; ++i)
                    for (int j = 0; j < d; ++j) Rb(i, j) = -Rb(i, j) + ldb * pc.Ri(j, i);
                for (int i = 0; i < d; ++i)
                    for (int j = 0; j < d; ++j) {
                        const double lamj = 1.0 / (ls(a, j) * ls(a, j)) + 1.0 / (ls(b, j) * ls(b, j));
                        sb(i, j) += 0.5 * RiTQb(i, j) + Rb(i, j) * lamj;
                    }
                for (int j = 0; j < d; ++j) {
                    double lam = 0.0;
                    for (int i = 0; i < d; ++i) lam += sM(i, j) * Rb(i, j);
                    ibar(a, j) += lam;
                    ibar(b, j) += lam;
                }
            }
        for (int a = 0; a < U; ++a) {
            const MeanC& mc = mean[a];
            vec qb(n), hb(n);
            double logcb = 0.0;
            Mat Gb(n, d);
            for (int i = 0; i < n; ++i) {
                double acc = Mbar[a];
                for (int k = 0; k < d; ++k) acc += mc.G(i, k) * V0b[(size_t)k * U + a];
                qb[i] = acc;
                hb[i] = acc * mc.q[i];
                logcb += hb[i];
                bb(a, i) += acc * mc.ex[i];
                for (int k = 0; k < d; ++k) {
                    Gb(i, k) = mc.q[i] * V0b[(size_t)k * U + a] - 0.5 * hb[i] * zeta(i, k);
                    zb(i, k) -= 0.5 * hb[i] * mc.G(i, k);
                }
            }
            const Mat zadd = mmT(Gb, mc.T);          // Gb T^T
            for (size_t e = 0; e < zb.d.size(); ++e) zb.d[e] += zadd.d[e];
            const Mat Tb = Tmm(zeta, Gb);
            const Mat TtTb = Tmm(mc.T, Tb);
            const Mat Ab2 = mmT(TtTb, mc.T);         // T^T Tb T^T
            for (int i = 0; i < d; ++i)
                for (int j = 0; j < d; ++j) {
                    const double ab = -0.5 * logcb * mc.T(i, j) - Ab2(i, j);
                    sb(i, j) += ab;
                    if (i == j) lsbar(a, i) += logcb / ls(a, i) + 2.0 * ls(a, i) * ab;
                }
        }
        for (int a = 0; a < U; ++a)
            for (int k = 0; k < d; ++k) lsbar(a, k) += ibar(a, k) * (-2.0 / (ls(a, k) * ls(a, k) * ls(a, k)));
        for (int i = 0; i < n; ++i)
            for (int k = 0; k < d; ++k) {
                Xbar(i, k) += zb(i, k);
                mxb[k] -= zb(i, k);
            }
        for (size_t e = 0; e < sxb.size(); ++e) sxb[e] += sb.d[e];
        for (size_t e = 0; e < bbsum.d.size(); ++e) bbsum.d[e] += bb.d[e];
    }
    // push the accumulated cotangent of beta through beta = (K + noise I)^-1 Y (once, after the sweep)
    void finish(double* dX, double* dY, double* dls) {
        Mat Xb = Xbar, lb = lsbar, Yb(n, U);
        for (int a = 0; a < U; ++a) {
            vec g(n, 0.0);
            for (int i = 0; i < n; ++i)
                for (int j = 0; j < n; ++j) g[i] += Ainv[a](i, j) * bbsum(a, j);      // Ainv symmetric
            Mat Wk(n, n);
            for (int i = 0; i < n; ++i) {
                Yb(i, a) = g[i];
                for (int j = 0; j < n; ++j) Wk(i, j) = -g[i] * beta(a, j) * K[a](i, j);
            }
            for (int i = 0; i < n; ++i)
                for (int j = 0; j < n; ++j) {
                    const double wsym = Wk(i, j) + Wk(j, i);
                    for (int k = 0; k < d; ++k) {
                        const double df = X(i, k) - X(j, k);
                        Xb(i, k) -= wsym * df / (ls(a, k) * ls(a, k));
                        lb(a, k) += 0.5 * wsym * df * df / (ls(a, k) * ls(a, k) * ls(a, k));
                    }
                }
        }
        memcpy(dX, Xb.d.data(), sizeof(double) * n * d);
        memcpy(dY, Yb.d.data(), sizeof(double) * n * U);
        memcpy(dls, lb.d.data(), sizeof(double) * U * d);
    }
};

// The reverse sweep common to both policies: propagate (pilco.py:147-149), joint Gaussian (:141-144), squash
// (controllers.py:13-36), rewards (rewards.py:19-81); the moment-matching adjoint of every step on the device.
// Host contraction of one step's Jacobian records (bwd.hip: k_mm_jac_fin) with the cotangents of the step's outputs:
// what pilco_gp_predict_vjp computes on the device, without touching the device.  M (E): the step's GP means (tape).
// The step streams its records (108 kB at C2u) once, cache-cold (the device wrote them): one core reads them at ~11 GB/s --
// 10 us per step, three quarters of the host sweep.  pf: distance (doubles) to the records the NEXT call will stream (the
// step before this one; 0: none): every line is requested one whole record ahead of its use.
void jac_vjp(const double* __restrict__ jr, int D, int E, const double* M, const double* Mbar, const double* Sbar, const double* Vbar,
             double* mbar, double* sbar, vec& acc, long pf) {
    auto axpy = [pf](double* __restrict__ y, const double c, const double* __restrict__ x, const int n) {
        if (pf != 0)
            for (int e = 0; e < n; e += 8) __builtin_prefetch(x + e + pf, 0, 1);
        for (int e = 0; e < n; ++e) y[e] += c * x[e];
    };
    const int nI = D * D, NT2 = D * (D + 1) / 2, recp = 1 + D + NT2, P = E * (E + 1) / 2;
    acc.assign((size_t)D + NT2, 0.0);
    double* __restrict__ am = acc.data();
    // pairs in the dealing order: (0,0) .. (E-1,E-1), (1,0), (2,0), (2,1), ...
    int pl = 0;
    auto add_pair = [&](int a, int b) {
        const double shat = (a == b) ? Sbar[(size_t)a * E + a] : Sbar[(size_t)a * E + b] + Sbar[(size_t)b * E + a];
        const double* __restrict__ r = jr + (size_t)pl * recp + 1;
        axpy(am, shat, r, D + NT2);   // (a zero cotangent adds zeros: same bits as skipping it, and the prefetch still runs)
        ++pl;
    };
    for (int a = 0; a < E; ++a) add_pair(a, a);
    for (int a = 1; a < E; ++a)
        for (int b = 0; b < a; ++b) add_pair(a, b);
    const double* jo = jr + (size_t)P * recp;
    const size_t reco = (size_t)D + NT2 + nI + (size_t)D * NT2;
    for (int a = 0; a < E; ++a) {
        double mu = Mbar[a];
        for (int b = 0; b < E; ++b) mu -= (Sbar[(size_t)a * E + b] + Sbar[(size_t)b * E + a]) * M[b];
        const double* __restrict__ r = jo + (size_t)a * reco;
        axpy(am, mu, r, D + NT2);            // dM/dm | sym dM/ds are contiguous
        const double* __restrict__ dVdm = r + D + NT2;
        const double* __restrict__ dVds = dVdm + nI;
        for (int k = 0; k < D; ++k) {
            const double vb = Vbar[(size_t)k * E + a];
            axpy(am, vb, dVdm + (size_t)k * D, D);
            axpy(am + D, vb, dVds + (size_t)k * NT2, NT2);
        }
    }
    for (int d = 0; d < D; ++d) mbar[d] = am[d];
    for (int c = 0; c < D; ++c)
        for (int r = 0; r <= c; ++r) sbar[(size_t)r * D + c] = sbar[(size_t)c * D + r] = am[D + (size_t)c * (c + 1) / 2 + r];
}

// In two halves, so that several value-and-gradient rollouts (the lanes of pilco_rollout_grad_batch) can be in flight at once:
// rollout_grad_begin enqueues the forward half -- with `defer` it returns without waiting for anything -- and
// rollout_grad_finish waits for the records chunk by chunk while it runs the host's reverse sweep.
struct GradCall {
    bool jac = false;
    const double *traj = nullptr, *tape = nullptr, *jrec = nullptr, *reward_later = nullptr;
    size_t JS = 0;
    vec traj_v, tape_v;
    std::chrono::steady_clock::time_point tm0, tm1;
};
int rollout_grad_begin(pilco_ctx* ctx, const pilco_policy* policy, const pilco_reward_term* rewards, int n_rewards, const double* m0,
                       const double* S0, int H, double* reward, GradCall& gc, bool defer) {
    const int E = policy->state_dim, U = policy->control_dim, D = E + U;
    const size_t TS = (size_t)D + D * D + (size_t)E * D + E + (size_t)E * E + (size_t)D * E;
    // Forward half.  Jacobian tape (default): one O(N^2) sweep per step gives the value and the step's Jacobian records,
    // the reverse sweep below is host algebra only.  PILCO_GRAD_MODE=0 / pilco_set_grad_mode(ctx, 0): plain tape, and
    // the O(N^2) adjoint of every step on the device again (pilco_gp_predict_vjp) -- the two agree to rounding.
    bool jac = ctx->grad_mode != 0;
    gc.tm0 = std::chrono::steady_clock::now();
    vec mH(E), SH((size_t)E * E);
    if (jac) {
        const int r = rollout_jtape(ctx, policy, rewards, n_rewards, m0, S0, H, reward, &gc.traj, &gc.tape, &gc.jrec, &gc.JS,
                                    defer ? &gc.reward_later : nullptr);
        if (r == PILCO_JAC_TOO_LARGE) jac = false;
        else if (r) return r;
    }
    if (!jac) {   // (runs to completion here: nothing of it overlaps with other lanes)
        gc.traj_v.resize((size_t)(H + 1) * (E + E * E));
        gc.tape_v.resize(std::max<size_t>(1, (size_t)H * TS));
        if (int r = pilco_rollout_tape(ctx, policy, rewards, n_rewards, m0, S0, H, mH.data(), SH.data(), reward, gc.traj_v.data(), gc.tape_v.data()))
            return r;
        gc.traj = gc.traj_v.data();
        gc.tape = gc.tape_v.data();
    }
    gc.jac = jac;
    gc.tm1 = std::chrono::steady_clock::now();
    return PILCO_OK;
}
int rollout_grad_finish(pilco_ctx* ctx, const pilco_policy* policy, const pilco_reward_term* rewards, int n_rewards, int H, double* reward,
                        PolicyAdj& pol, pilco_seed_fn seed_fn, void* seed_user, GradCall& gc) {
    const int E = policy->state_dim, U = policy->control_dim, D = E + U;
    const size_t TS = (size_t)D + D * D + (size_t)E * D + E + (size_t)E * E + (size_t)D * E;
    const bool jac = gc.jac;
    const bool timing = getenv("PILCO_GRAD_TIMING") != nullptr;   // developer aid: forward / reverse split on stderr
    if (gc.reward_later) {   // deferred begin: the reward, the trajectory, the tape and the last chunk of records
        if (int r = rollout_jtape_wait(ctx, H - 1)) return r;
        if (H <= 0) HIPCHK(hipStreamSynchronize(ctx->st));
        *reward = *gc.reward_later;
        gc.tm1 = std::chrono::steady_clock::now();
    }
    const double *traj = gc.traj, *tape = gc.tape, *jrec = gc.jrec;
    const size_t JS = gc.JS;
    const auto tm0 = gc.tm0, tm1 = gc.tm1;
    vec e(U);
    for (int u = 0; u < U; ++u) e[u] = policy->max_action[u];
    vec mbar(E, 0.0), sbar((size_t)E * E, 0.0);
    // An objective beyond the additive reward (Safe-PILCO's multiplicative risk term, any function of the state
    // trajectory): the caller turns the trajectory into cotangent seeds d objective / d (m_t, s_t), t = 0..H, and the
    // sweep adds them where the reward's own cotangents enter -- what TensorFlow's reverse mode does for whatever
    // the reference's training_loss contains (pilco/models/pilco.py:47-50, safe_pilco_extension/safe_pilco.py:29-50).
    const size_t SE = (size_t)E + (size_t)E * E;
    vec seeds;
    if (seed_fn) {
        seeds.assign((size_t)(H + 1) * SE, 0.0);
        seed_fn(seed_user, H, E, traj, seeds.data());
        for (size_t q = 0; q < seeds.size(); ++q)
            if (!std::isfinite(seeds[q])) return fail(ctx, PILCO_E_SHAPE, "rollout_grad: the seed callback returned a non-finite cotangent");
        const double* sd = &seeds[(size_t)H * SE];
        for (int i = 0; i < E; ++i) mbar[i] = sd[i];
        for (int i = 0; i < E; ++i)
            for (int j = 0; j < E; ++j) sbar[(size_t)i * E + j] = 0.5 * (sd[E + (size_t)i * E + j] + sd[E + (size_t)j * E + i]);
    }
    vec G((size_t)E * E), Vb((size_t)D * E), s1bar((size_t)E * D), mjb(D), sjb((size_t)D * D), mxb(E), sxb((size_t)E * E);
    vec Bb((size_t)E * U), sub((size_t)U * U), mu0(U), su0((size_t)U * U), V0((size_t)E * U), V0b((size_t)E * U), cb((size_t)E * U), Cdbar(U);
    vec mu0b, su0b, rm(E), rS((size_t)E * E), jacc;
    Squash sq;
    for (int t = H - 1; t >= 0; --t) {
        const double* m_x = &traj[(size_t)t * (E + E * E)];
        const double* s_x = m_x + E;
        const double* rec = &tape[(size_t)t * TS];
        const double* m_j = rec;
        const double* s_j = rec + D;
        const double* s1 = rec + D + D * D;                                      // (E, D)
        const double* Mgp = rec + D + D * D + (size_t)E * D;                     // (E)   GP means of the step
        const double* V = rec + D + D * D + (size_t)E * D + E + (size_t)E * E;   // (D, E)
        // propagate (pilco.py:147-149): M_x = M + m_x, S_x = S + s_x + s1 V + (s1 V)^T
        for (int i = 0; i < E; ++i)
            for (int j = 0; j < E; ++j) G[(size_t)i * E + j] = sbar[(size_t)i * E + j] + sbar[(size_t)j * E + i];
        for (int d = 0; d < D; ++d)
            for (int j = 0; j < E; ++j) {
                double acc = 0.0;
                for (int i = 0; i < E; ++i) acc += s1[(size_t)i * D + d] * G[(size_t)i * E + j];
                Vb[(size_t)d * E + j] = acc;                                 // s1^T G
            }
        for (int i = 0; i < E; ++i)
            for (int d = 0; d < D; ++d) {
                double acc = 0.0;
                for (int j = 0; j < E; ++j) acc += G[(size_t)i * E + j] * V[(size_t)d * E + j];
                s1bar[(size_t)i * D + d] = acc;                              // G V^T
            }
        mxb = mbar;
        sxb = sbar;
        if (jac) {
            if (int r = rollout_jtape_wait(ctx, t)) return r;
            jac_vjp(jrec + (size_t)t * JS, D, E, Mgp, mbar.data(), sbar.data(), Vb.data(), mjb.data(), sjb.data(), jacc,
                    (t > 0 && t - 1 >= ctx->jwait_from) ? -(long)JS : 0);   // (only records that have arrived)
            for (int q = 0; q < D * D; ++q)
                if (!std::isfinite(sjb[q])) return fail(ctx, PILCO_E_NOT_PD, "rollout_grad: singular s + Lambda^2 or I + Lambda s");
        } else if (int r = pilco_gp_predict_vjp(ctx, PILCO_SLOT_DYNAMICS, m_j, s_j, mbar.data(), sbar.data(), Vb.data(), mjb.data(), sjb.data())) {
            return r;
        }
        // joint Gaussian (pilco.py:141-144)
        for (int i = 0; i < E; ++i) mxb[i] += mjb[i];
        for (int i = 0; i < E; ++i)
            for (int j = 0; j < E; ++j) sxb[(size_t)i * E + j] += sjb[(size_t)i * D + j] + s1bar[(size_t)i * D + j];
        for (int i = 0; i < E; ++i)
            for (int u = 0; u < U; ++u)
                Bb[(size_t)i * U + u] = sjb[(size_t)i * D + E + u] + sjb[(size_t)(E + u) * D + i] + s1bar[(size_t)i * D + E + u];
        for (int u = 0; u < U; ++u)
            for (int v = 0; v < U; ++v) sub[(size_t)u * U + v] = sjb[(size_t)(E + u) * D + E + v];
        // controller: (mu0, su0, V0) -> squash_sin -> (m_u, s_u, c = V0 diag(Cd))   (controllers.py:46-58,108-121)
        if (!pol.fwd(m_x, s_x, mu0, su0, V0)) return fail(ctx, PILCO_E_NOT_PD, "rollout_grad: singular matrix in the policy");
        sq.fwd(mu0, su0, e);
        for (int i = 0; i < E; ++i)
            for (int j = 0; j < E; ++j) {
                double acc = 0.0;
                for (int u = 0; u < U; ++u) acc += Bb[(size_t)i * U + u] * V0[(size_t)j * U + u] * sq.Cd[u];   // Bb c^T
                sxb[(size_t)i * E + j] += acc;
            }
        for (int i = 0; i < E; ++i)
            for (int u = 0; u < U; ++u) {
                double acc = 0.0;
                for (int l = 0; l < E; ++l) acc += s_x[(size_t)l * E + i] * Bb[(size_t)l * U + u];
                cb[(size_t)i * U + u] = acc;                                 // s_x^T Bb
            }
        for (int u = 0; u < U; ++u) {
            double acc = 0.0;
            for (int i = 0; i < E; ++i) {
                acc += V0[(size_t)i * U + u] * cb[(size_t)i * U + u];
                V0b[(size_t)i * U + u] = cb[(size_t)i * U + u] * sq.Cd[u];
            }
            Cdbar[u] = acc;
        }
        sq.vjp(&mjb[E], sub.data(), Cdbar.data(), mu0b, su0b);
        pol.vjp(m_x, s_x, mu0b, su0b, V0b, mxb, sxb);
        // reward of the pre-propagation state (pilco.py:133)
        std::fill(rm.begin(), rm.end(), 0.0);
        std::fill(rS.begin(), rS.end(), 0.0);
        if (!reward_grad(rewards, n_rewards, E, m_x, s_x, rm, rS)) return fail(ctx, PILCO_E_NOT_PD, "rollout_grad: singular I + S W in the reward");
        for (int i = 0; i < E; ++i) mxb[i] += rm[i];
        for (int i = 0; i < E * E; ++i) sxb[i] += rS[i];
        if (seed_fn) {
            const double* sd = &seeds[(size_t)t * SE];
            for (int i = 0; i < E; ++i) mxb[i] += sd[i];
            for (int i = 0; i < E * E; ++i) sxb[i] += sd[E + i];
        }
        mbar = mxb;
        for (int i = 0; i < E; ++i)
            for (int j = 0; j < E; ++j) sbar[(size_t)i * E + j] = 0.5 * (sxb[(size_t)i * E + j] + sxb[(size_t)j * E + i]);
    }
    if (timing) {
        const auto tm2 = std::chrono::steady_clock::now();
        fprintf(stderr, "[pilco grad] forward (device + download) %.3f ms, reverse sweep %.3f ms, mode %d\n",
                std::chrono::duration<double, std::milli>(tm1 - tm0).count(), std::chrono::duration<double, std::milli>(tm2 - tm1).count(), jac ? 1 : 0);
    }
    return PILCO_OK;
}
int rollout_grad_impl(pilco_ctx* ctx, const pilco_policy* policy, const pilco_reward_term* rewards, int n_rewards,
                      const double* m0, const double* S0, int H, double* reward, PolicyAdj& pol, pilco_seed_fn seed_fn, void* seed_user) {
    GradCall gc;
    if (int r = rollout_grad_begin(ctx, policy, rewards, n_rewards, m0, S0, H, reward, gc, false)) return r;
    return rollout_grad_finish(ctx, policy, rewards, n_rewards, H, reward, pol, seed_fn, seed_user, gc);
}

// LinearController: the reverse chain on the device (rev.hip).  begin enqueues the forward half, the records' finish and --
// unless the caller has cotangent seeds to add -- the chain itself, and returns without waiting; finish waits (with seeds:
// trajectory -> callback -> upload -> chain -> wait).  PILCO_JAC_TOO_LARGE from begin: nothing was enqueued, the caller takes
// the host chain.
bool dev_chain_applies(const pilco_ctx* ctx, const pilco_policy* policy) {
    const int E = policy->state_dim, U = policy->control_dim;
    return ctx->dev_chain && ctx->grad_mode != 0 && E + U <= 14 && rev_chain_supported(E, U, E + U);
}
int rollout_grad_dev_begin(pilco_ctx* ctx, const pilco_policy* policy, const pilco_reward_term* rewards, int n_rewards, const double* m0,
                           const double* S0, int H, bool seeds, JtapeDev& dev) {
    dev.seeds = seeds;
    double r_unused = 0.0;
    const double *t0 = nullptr, *t1 = nullptr, *t2 = nullptr;
    size_t js = 0;
    return rollout_jtape(ctx, policy, rewards, n_rewards, m0, S0, H, &r_unused, &t0, &t1, &t2, &js, nullptr, &dev);
}
int rollout_grad_dev_finish(pilco_ctx* ctx, JtapeDev& dev, const pilco_policy* policy, int H, pilco_seed_fn seed_fn, void* seed_user,
                            double* reward, double* dW, double* db) {
    const int E = policy->state_dim, U = policy->control_dim;
    if (int r = rollout_jtape_dev_finish(ctx, dev, H, E, seed_fn, seed_user)) return r;
    *reward = *dev.h_reward;
    memcpy(dW, dev.h_out, sizeof(double) * (size_t)U * E);
    memcpy(db, dev.h_out + (size_t)U * E, sizeof(double) * (size_t)U);
    return PILCO_OK;
}

int check_grad_args(pilco_ctx* ctx, const pilco_policy* policy, const pilco_reward_term* rewards, int n_rewards, int kind) {
    if (policy->kind != kind || !policy->squash || policy->control_dim <= 0)
        return fail(ctx, PILCO_E_SHAPE, "rollout_grad: squashed LinearController (pilco_rollout_grad) or RbfController (pilco_rollout_grad_rbf) only");
    for (int k = 0; k < n_rewards; ++k)
        if (rewards[k].kind != PILCO_REWARD_EXPONENTIAL && rewards[k].kind != PILCO_REWARD_LINEAR)
            return fail(ctx, PILCO_E_SHAPE, "rollout_grad: unknown reward term");
    return PILCO_OK;
}

}  // namespace

extern "C" {

// Value and gradient of the rollout reward w.r.t. a LinearController's (W, b): what TensorFlow's reverse mode through
// the tf.while_loop gives the reference (pilco/models/pilco.py:85-90,126-135).  Forward rollout with a tape on the
// device, then the reverse sweep: the O(N^2) adjoint of every moment-matching step on the device
// (pilco_gp_predict_vjp), the O(D^3) links here on the host in C++.  dW (U,E), db (U).
int pilco_rollout_grad_seeded(pilco_ctx* ctx, const pilco_policy* policy, const pilco_reward_term* rewards, int n_rewards,
                              const double* m0, const double* S0, int H, pilco_seed_fn seed_fn, void* seed_user, double* reward,
                              double* dW, double* db) {
    if (!ctx) return PILCO_E_SHAPE;
    if (!policy || !m0 || !S0 || !reward || !dW || !db || H < 0) return fail(ctx, PILCO_E_SHAPE, "rollout_grad: bad arguments");
    if (int r = check_grad_args(ctx, policy, rewards, n_rewards, PILCO_POLICY_LINEAR)) return r;
    if (dev_chain_applies(ctx, policy)) {
        JtapeDev dev;
        const int r = rollout_grad_dev_begin(ctx, policy, rewards, n_rewards, m0, S0, H, seed_fn != nullptr, dev);
        if (r == PILCO_OK) return rollout_grad_dev_finish(ctx, dev, policy, H, seed_fn, seed_user, reward, dW, db);
        if (r != PILCO_JAC_TOO_LARGE) return r;
    }
    LinearAdj pol(policy->state_dim, policy->control_dim, policy->W, policy->b);
    if (int r = rollout_grad_impl(ctx, policy, rewards, n_rewards, m0, S0, H, reward, pol, seed_fn, seed_user)) return r;
    memcpy(dW, pol.Wbar.data(), sizeof(double) * pol.Wbar.size());
    memcpy(db, pol.bbar.data(), sizeof(double) * pol.bbar.size());
    return PILCO_OK;
}
int pilco_rollout_grad(pilco_ctx* ctx, const pilco_policy* policy, const pilco_reward_term* rewards, int n_rewards,
                       const double* m0, const double* S0, int H, double* reward, double* dW, double* db) {
    return pilco_rollout_grad_seeded(ctx, policy, rewards, n_rewards, m0, S0, H, nullptr, nullptr, reward, dW, db);
}

// The same for an RbfController whose GP lives in PILCO_SLOT_POLICY: gradients w.r.t. the centres Xp (bf,E), the
// targets Yp (bf,U) and the lengthscales lsp (U,E) (controllers.py:80-129); the caller passes the host copies of the
// policy parameters it uploaded with pilco_gp_set_data / _set_hyp (noisep (U): the FakeGPR likelihood variance).
int pilco_rollout_grad_rbf_seeded(pilco_ctx* ctx, const pilco_policy* policy, const pilco_reward_term* rewards, int n_rewards,
                                  const double* m0, const double* S0, int H, const double* Xp, const double* Yp, const double* lsp,
                                  const double* noisep, int bf, pilco_seed_fn seed_fn, void* seed_user, double* reward, double* dX,
                                  double* dY, double* dls) {
    if (!ctx) return PILCO_E_SHAPE;
    if (!policy || !m0 || !S0 || !reward || !Xp || !Yp || !lsp || !noisep || !dX || !dY || !dls || H < 0 || bf <= 0)
        return fail(ctx, PILCO_E_SHAPE, "rollout_grad_rbf: bad arguments");
    if (int r = check_grad_args(ctx, policy, rewards, n_rewards, PILCO_POLICY_RBF)) return r;
    RbfAdj pol;
    if (!pol.init(bf, policy->state_dim, policy->control_dim, Xp, Yp, lsp, noisep))
        return fail(ctx, PILCO_E_NOT_PD, "rollout_grad_rbf: K + noise I of the policy is singular");
    if (int r = rollout_grad_impl(ctx, policy, rewards, n_rewards, m0, S0, H, reward, pol, seed_fn, seed_user)) return r;
    pol.finish(dX, dY, dls);
    return PILCO_OK;
}
int pilco_rollout_grad_rbf(pilco_ctx* ctx, const pilco_policy* policy, const pilco_reward_term* rewards, int n_rewards,
                           const double* m0, const double* S0, int H, const double* Xp, const double* Yp, const double* lsp,
                           const double* noisep, int bf, double* reward, double* dX, double* dY, double* dls) {
    return pilco_rollout_grad_rbf_seeded(ctx, policy, rewards, n_rewards, m0, S0, H, Xp, Yp, lsp, noisep, bf, nullptr, nullptr, reward, dX,
                                         dY, dls);
}

// B value-and-gradient rollouts of ONE dynamics model in flight together: the restarts of optimize_policy (pilco.py:94-107
// runs them one after the other; every restart is an L-BFGS-B walk of its own, and their evaluations are independent).
// Lanes as in pilco_rollout_batch (contexts of their own that borrow this context's model); every lane's forward half
// (steps, batched finish, downloads) is enqueued on its stream before the first wait, then the host's reverse sweeps run
// lane by lane -- lane i's sweep while lanes i+1.. are still on the device.  Every lane runs exactly the launch sequence
// and the host arithmetic of its solo call: results are bit-identical to pilco_rollout_grad / pilco_rollout_grad_rbf.
// LinearController lanes: policies[i].W / .b; dW (B, U, E), db (B, U), reward (B); m0 (B, E), S0 (B, E, E).
// (seed_fn, seed_users [B]: an objective beyond the additive reward, as pilco_rollout_grad_seeded -- the callback runs once per lane,
// in lane order, with seed_users[i], when lane i's trajectory has arrived and before its reverse sweep)
int pilco_rollout_grad_batch_seeded(pilco_ctx* ctx, int B, const pilco_policy* policies, const pilco_reward_term* rewards, int n_rewards,
                                    const double* m0, const double* S0, int H, pilco_seed_fn seed_fn, void* const* seed_users,
                                    double* reward, double* dW, double* db) {
    if (!ctx) return PILCO_E_SHAPE;
    if (B <= 0 || B > 64 || !policies || !m0 || !S0 || !reward || !dW || !db || H < 0) return fail(ctx, PILCO_E_SHAPE, "rollout_grad_batch: bad arguments");
    if (ctx->nranks != 1 || ctx->comm) return fail(ctx, PILCO_E_STATE, "rollout_grad_batch: single rank only (shard OR batch)");
    if (!ctx->slot[0].factor_valid) return fail(ctx, PILCO_E_STATE, "rollout_grad_batch: dynamics model has no current factorisation");
    for (int i = 0; i < B; ++i)
        if (int r = check_grad_args(ctx, &policies[i], rewards, n_rewards, PILCO_POLICY_LINEAR)) return r;
    std::vector<pilco_ctx*> lane;
    LanesGuard lanes_guard{ctx};
    if (int r = rollout_lanes(ctx, B, lane, "rollout_grad_batch")) return r;
    const int E = policies[0].state_dim, U = policies[0].control_dim;
    std::vector<GradCall> gc((size_t)B);
    int err = PILCO_OK, begun = 0;
    if (dev_chain_applies(ctx, &policies[0])) {   // every lane's chain on the device: all lanes enqueued before the first wait
        std::vector<JtapeDev> dv((size_t)B);
        bool fallback = false;
        for (int i = 0; i < B && !err; ++i) {
            const int r = rollout_grad_dev_begin(lane[i], &policies[i], rewards, n_rewards, m0 + (size_t)i * E, S0 + (size_t)i * E * E, H,
                                                 seed_fn != nullptr, dv[i]);
            if (r == PILCO_JAC_TOO_LARGE && i == 0) {   // (the same answer for every lane: same model, same horizon)
                fallback = true;
                break;
            }
            if (r) {
                err = r;
                if (i > 0) ctx->err = "lane " + std::to_string(i) + ": " + lane[i]->err;
                break;
            }
            ++begun;
        }
        if (!fallback) {
            for (int i = 0; i < begun; ++i) {
                const int r = rollout_grad_dev_finish(lane[i], dv[i], &policies[i], H, seed_fn, (seed_fn && seed_users) ? seed_users[i] : nullptr,
                                                      reward + i, dW + (size_t)i * U * E, db + (size_t)i * U);
                if (r && !err) {
                    err = r;
                    if (i > 0) ctx->err = "lane " + std::to_string(i) + ": " + lane[i]->err;
                }
            }
            if (err)
                for (int i = 0; i < B; ++i) (void)hipStreamSynchronize(lane[i]->st);
            return err;
        }
    }
    for (int i = 0; i < B && !err; ++i, ++begun) {
        err = rollout_grad_begin(lane[i], &policies[i], rewards, n_rewards, m0 + (size_t)i * E, S0 + (size_t)i * E * E, H, reward + i, gc[i], true);
        if (err && i > 0) ctx->err = "lane " + std::to_string(i) + ": " + lane[i]->err;
        if (err) break;
    }
    for (int i = 0; i < begun; ++i) {
        LinearAdj pol(E, U, policies[i].W, policies[i].b);
        const int r = rollout_grad_finish(lane[i], &policies[i], rewards, n_rewards, H, reward + i, pol, seed_fn,
                                          (seed_fn && seed_users) ? seed_users[i] : nullptr, gc[i]);
        if (r && !err) {
            err = r;
            if (i > 0) ctx->err = "lane " + std::to_string(i) + ": " + lane[i]->err;
        }
        if (!r) {
            memcpy(dW + (size_t)i * U * E, pol.Wbar.data(), sizeof(double) * (size_t)U * E);
            memcpy(db + (size_t)i * U, pol.bbar.data(), sizeof(double) * (size_t)U);
        }
    }
    if (err)   // a lane that failed after others had begun: nothing may stay in flight behind the caller's back
        for (int i = 0; i < B; ++i) (void)hipStreamSynchronize(lane[i]->st);
    return err;
}

int pilco_rollout_grad_batch(pilco_ctx* ctx, int B, const pilco_policy* policies, const pilco_reward_term* rewards, int n_rewards,
                             const double* m0, const double* S0, int H, double* reward, double* dW, double* db) {
    return pilco_rollout_grad_batch_seeded(ctx, B, policies, rewards, n_rewards, m0, S0, H, nullptr, nullptr, reward, dW, db);
}

// RbfController lanes: lane i's policy GP (centres Xp (B, bf, E), targets Yp (B, bf, U), lengthscales lsp (B, U, E), likelihood
// variances noisep (B, U); unit signal variance, controllers.py:92-93) is uploaded to and factorised in PILCO_SLOT_POLICY of
// lane i's context by this call -- INCLUDING lane 0, this context: whatever the caller had in its policy slot is replaced by
// lane 0's controller.  dX (B, bf, E), dY (B, bf, U), dls (B, U, E).
int pilco_rollout_grad_rbf_batch_seeded(pilco_ctx* ctx, int B, const pilco_policy* policies, const pilco_reward_term* rewards, int n_rewards,
                                        const double* m0, const double* S0, int H, const double* Xp, const double* Yp, const double* lsp,
                                        const double* noisep, int bf, pilco_seed_fn seed_fn, void* const* seed_users, double* reward,
                                        double* dX, double* dY, double* dls) {
    if (!ctx) return PILCO_E_SHAPE;
    if (B <= 0 || B > 64 || !policies || !m0 || !S0 || !reward || !Xp || !Yp || !lsp || !noisep || !dX || !dY || !dls || H < 0 || bf <= 0)
        return fail(ctx, PILCO_E_SHAPE, "rollout_grad_rbf_batch: bad arguments");
    if (ctx->nranks != 1 || ctx->comm) return fail(ctx, PILCO_E_STATE, "rollout_grad_rbf_batch: single rank only (shard OR batch)");
    if (!ctx->slot[0].factor_valid) return fail(ctx, PILCO_E_STATE, "rollout_grad_rbf_batch: dynamics model has no current factorisation");
    for (int i = 0; i < B; ++i)
        if (int r = check_grad_args(ctx, &policies[i], rewards, n_rewards, PILCO_POLICY_RBF)) return r;
    std::vector<pilco_ctx*> lane;
    LanesGuard lanes_guard{ctx};
    if (int r = rollout_lanes(ctx, B, lane, "rollout_grad_rbf_batch")) return r;
    const int E = policies[0].state_dim, U = policies[0].control_dim;
    const size_t nX = (size_t)bf * E, nY = (size_t)bf * U, nL = (size_t)U * E;
    std::vector<double> ones((size_t)U, 1.0);
    std::vector<GradCall> gc((size_t)B);
    std::vector<RbfAdj> pol((size_t)B);
    int err = PILCO_OK, begun = 0;
    for (int i = 0; i < B; ++i, ++begun) {
        pilco_ctx* l = lane[i];
        err = pilco_gp_set_data(l, PILCO_SLOT_POLICY, Xp + i * nX, Yp + i * nY, bf, E, U);
        if (!err) err = pilco_gp_set_hyp(l, PILCO_SLOT_POLICY, lsp + i * nL, ones.data(), noisep + (size_t)i * U);
        if (!err) err = pilco_gp_factorize(l, PILCO_SLOT_POLICY);
        if (!err && !pol[i].init(bf, E, U, Xp + i * nX, Yp + i * nY, lsp + i * nL, noisep + (size_t)i * U))
            err = fail(l, PILCO_E_NOT_PD, "rollout_grad_rbf_batch: K + noise I of the policy is singular");
        if (!err) err = rollout_grad_begin(l, &policies[i], rewards, n_rewards, m0 + (size_t)i * E, S0 + (size_t)i * E * E, H, reward + i, gc[i], true);
        if (err) {
            if (i > 0) ctx->err = "lane " + std::to_string(i) + ": " + l->err;
            break;
        }
    }
    for (int i = 0; i < begun; ++i) {
        const int r = rollout_grad_finish(lane[i], &policies[i], rewards, n_rewards, H, reward + i, pol[i], seed_fn,
                                          (seed_fn && seed_users) ? seed_users[i] : nullptr, gc[i]);
        if (r && !err) {
            err = r;
            if (i > 0) ctx->err = "lane " + std::to_string(i) + ": " + lane[i]->err;
        }
        if (!r) pol[i].finish(dX + i * nX, dY + i * nY, dls + i * nL);
    }
    if (err)
        for (int i = 0; i < B; ++i) (void)hipStreamSynchronize(lane[i]->st);
    return err;
}
int pilco_rollout_grad_rbf_batch(pilco_ctx* ctx, int B, const pilco_policy* policies, const pilco_reward_term* rewards, int n_rewards,
                                 const double* m0, const double* S0, int H, const double* Xp, const double* Yp, const double* lsp,
                                 const double* noisep, int bf, double* reward, double* dX, double* dY, double* dls) {
    return pilco_rollout_grad_rbf_batch_seeded(ctx, B, policies, rewards, n_rewards, m0, S0, H, Xp, Yp, lsp, noisep, bf, nullptr, nullptr,
                                               reward, dX, dY, dls);
}

}  // extern "C"
