// C ABI of libpilco_hip.so (see include/pilco_hip.h): context, GP slots,
// factorisation drivers, single moment-matching step, rollout, sharding.
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>

#include "moment.h"

using namespace pilco;

namespace {

struct Slot {
    int N = 0, D = 0, E = 0, M = 0;  // data size, input dim, outputs, inducing points (0 = exact)
    int Npad = 0;                    // padded N
    int n = 0, npad = 0;             // points the moment matching runs over (N or M) and padding
    bool has_data = false, has_hyp = false, factor_valid = false, user_factors = false, iK_null = false;
    bool ignore_iK = false;  // policy slot: RbfController evaluates with iK zeroed (controllers.py:116)
    DevBuf bwd_mom, bwd_cp, bwd_part, bwd_out; // reverse-pass scratch
    DevBuf Xt, Yt, Zt, ls, var, noise;         // Yt: [E][Npad]
    DevBuf K, Linv, iK, invD, beta, Tscr, vec; // factorisation
    DevBuf Kmn, V2, Am, AmInv, AmD, iAt, G;    // FITC extras
    // moment-matching workspace
    DevBuf w_in, w_At, w_Bt, w_small, w_part, w_gath, w_out;
    MMWork wk{};
    bool wk_valid = false;
    int wk_variant = -1;
    std::vector<int> pair_owner;  // [P]
};

}  // namespace

struct pilco_ctx {
    int device = 0;
    hipStream_t st = nullptr;
    std::string err;
    int not_pd = -1;
    int variant = 0;
    int rank = 0, nranks = 1;
    ncclComm_t comm = nullptr;
    Slot slot[2];
    int* d_info = nullptr;
    DevBuf state;   // m_x, s_x, s1, reward, act_out, rew_out
    DevBuf params;  // policy + reward parameters
    DevBuf traj;
    DevBuf tape;
    DevBuf selftest;
    DevBuf exp_tab;  // 2^(j/n), j = 0..n-1, n = mm_exp_table_size()
    unsigned long long* dbg = nullptr;
    // cached hipGraph of one rollout (single-rank): replayed while the plan key is unchanged
    hipGraphExec_t graph = nullptr;
    std::vector<unsigned long long> graph_key;
    bool use_graph = true;
    bool graph_rccl_failed = false;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::vector<hipEvent_t> pair_events;
    double* pin = nullptr;   // pinned host staging buffer of the reverse pass (truly asynchronous small copies)
    size_t pin_cap = 0;
};

namespace {

int fail(pilco_ctx* c, int code, const std::string& msg) {
    if (c) c->err = msg;
    return code;
}

#define HIPCHK(call)                                                                                       \
    do {                                                                                                   \
        hipError_t e_ = (call);                                                                            \
        if (e_ != hipSuccess)                                                                              \
            return fail(ctx, PILCO_E_HIP, std::string(#call) + ": " + hipGetErrorString(e_));              \
    } while (0)

#define ENSURE(buf, count)                                                                                 \
    do {                                                                                                   \
        if ((buf).ensure(count) != hipSuccess) return fail(ctx, PILCO_E_ALLOC, "hipMalloc failed: " #buf); \
    } while (0)

int check_slot(pilco_ctx* ctx, int slot) {
    if (!ctx) return PILCO_E_SHAPE;
    if (slot < 0 || slot > 1) return fail(ctx, PILCO_E_SHAPE, "slot must be 0 (dynamics) or 1 (policy)");
    return PILCO_OK;
}

// pair p = a(a+1)/2 + b; dealing order: diagonal pairs first, then the rest
std::vector<int> deal_pairs(int E, int nranks) {
    const int P = E * (E + 1) / 2;
    std::vector<int> owner(P, 0);
    int k = 0;
    for (int a = 0; a < E; ++a) owner[a * (a + 1) / 2 + a] = (k++) % nranks;
    for (int a = 0; a < E; ++a)
        for (int b = 0; b < a; ++b) owner[a * (a + 1) / 2 + b] = (k++) % nranks;
    return owner;
}

int build_work(pilco_ctx* ctx, Slot& s) {
    if (s.wk_valid && s.wk_variant == ctx->variant) return PILCO_OK;
    const int E = s.E, D = s.D, npad = s.npad;
    const int P = E * (E + 1) / 2;
    const int W = ctx->nranks, rank = ctx->rank;
    s.pair_owner = deal_pairs(E, W);
    const int PLcap = (P + W - 1) / W, ELcap = (E + W - 1) / W;
    MMWork& wk = s.wk;
    wk.PL = (rank < P) ? (P - rank + W - 1) / W : 0;
    wk.EL = (rank < E) ? (E - rank + W - 1) / W : 0;
    wk.P = P;
    wk.KP = mm_kp(D);
    mm_prep_chunks(npad, std::max(wk.PL, 1), wk.EL, &wk.NCH, &wk.NCHM);
    wk.NT = mm_pair_nt(npad, ctx->variant, std::max(wk.PL, 1));
    wk.OUTOFF = PLcap;
    wk.SEG = PLcap + ELcap * (1 + D);
    wk.rank = rank;
    wk.nranks = W;
    wk.abl = getenv("PILCO_ABL") ? atoi(getenv("PILCO_ABL")) : 0;
    const int PLa = std::max(wk.PL, 1);
    // stream-K geometry (variant 0)
    wk.sk_waves = 0;
    wk.sk_maxw = 0;
    if (ctx->variant == 0 && wk.PL > 0) {
        int tdiag, toff;
        mm_pair_sk_steps(npad, &tdiag, &toff);
        // local pairs kk = pl*W + rank are diagonal (kk < E) first; they stream iK unless it is absent
        int nd = 0;
        const bool no_iK = s.iK_null || s.ignore_iK;
        if (!no_iK)
            for (int pl = 0; pl < wk.PL; ++pl)
                if (pl * W + rank < E) ++nd;
        if (no_iK) tdiag = toff;
        const long T = (long)nd * tdiag + (long)(wk.PL - nd) * toff;
        int waves = mm_pair_sk_capacity(wk.KP);
        if ((long)waves > T) waves = (int)std::max<long>(4, (T + 3) / 4 * 4);
        wk.sk_waves = waves;
        wk.sk_total = (int)T;
        wk.sk_nd = nd;
        wk.sk_tdiag = tdiag;
        wk.sk_toff = toff;
        wk.sk_ud = 5;   // measured: a diagonal step (iK stream + one more FMA per element) costs ~5/4 of an off-diagonal one
        wk.sk_uo = 4;
        if (const char* env = getenv("PILCO_SK_UNITS")) {
            int a = 0, b = 0;
            if (sscanf(env, "%d,%d", &a, &b) == 2 && a > 0 && b > 0) { wk.sk_ud = a; wk.sk_uo = b; }
        }
        wk.sk_maxw = mm_sk_maxw(wk);
    }
    ENSURE(s.w_in, (size_t)D + D * D + E + E * E + D * E);   // m | s | cotangents (Mbar | Sbar | Vbar) of the reverse pass
    ENSURE(s.w_At, (size_t)PLa * wk.KP * npad);
    ENSURE(s.w_Bt, (size_t)PLa * wk.KP * npad);
    ENSURE(s.w_small, (size_t)PLa + (size_t)E * wk.NCHM * (1 + D));
    ENSURE(s.w_part, std::max((size_t)PLa * wk.NT * 2, (size_t)PLa * std::max(wk.sk_maxw, 4)));
    ENSURE(s.w_gath, (size_t)W * wk.SEG);
    ENSURE(s.w_out, (size_t)E + E * E + D * E);
    wk.in_m = s.w_in.p;
    wk.in_s = s.w_in.p + D;
    wk.At = s.w_At.p;
    wk.Bt = s.w_Bt.p;
    wk.pair_isdet = s.w_small.p;
    wk.mean_part = wk.pair_isdet + PLa;
    wk.pair_part = s.w_part.p;
    wk.sk_part = s.w_part.p;
    wk.gath = s.w_gath.p;
    wk.out_M = s.w_out.p;
    wk.out_S = wk.out_M + E;
    wk.out_V = wk.out_S + E * E;
    wk.exp_tab = ctx->exp_tab.p;
    wk.dbg = ctx->dbg;
    HIPCHK(hipMemsetAsync(s.w_gath.p, 0, sizeof(double) * W * wk.SEG, ctx->st));
    if (wk.sk_waves > 0) HIPCHK(hipMemsetAsync(s.w_part.p, 0, sizeof(double) * PLa * wk.sk_maxw, ctx->st));   // slots no wave writes
    s.wk_valid = true;
    s.wk_variant = ctx->variant;
    return PILCO_OK;
}

// W (E x E, symmetric PSD) = F F^T with F (E x rank) from a cyclic Jacobi eigen-decomposition.
// Returns rank, or -1 when W is not symmetric PSD (the general pivoted device path is used then).
int psd_factor(const double* W, int E, std::vector<double>& F) {
    double scale = 0.0;
    for (int i = 0; i < E * E; ++i) scale = std::max(scale, std::fabs(W[i]));
    if (scale == 0.0) { F.clear(); return 0; }
    for (int i = 0; i < E; ++i)
        for (int j = 0; j < i; ++j)
            if (std::fabs(W[i * E + j] - W[j * E + i]) > 1e-13 * scale) return -1;
    std::vector<double> A(W, W + E * E), V(E * E, 0.0);
    for (int i = 0; i < E; ++i) V[i * E + i] = 1.0;
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0.0;
        for (int i = 0; i < E; ++i)
            for (int j = 0; j < i; ++j) off += A[i * E + j] * A[i * E + j];
        if (off <= 1e-32 * scale * scale) break;
        for (int p = 0; p < E; ++p)
            for (int q = p + 1; q < E; ++q) {
                const double apq = A[p * E + q];
                if (apq == 0.0) continue;
                const double theta = (A[q * E + q] - A[p * E + p]) / (2.0 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                const double c = 1.0 / std::sqrt(t * t + 1.0), sn = t * c;
                for (int k = 0; k < E; ++k) {
                    const double akp = A[k * E + p], akq = A[k * E + q];
                    A[k * E + p] = c * akp - sn * akq;
                    A[k * E + q] = sn * akp + c * akq;
                }
                for (int k = 0; k < E; ++k) {
                    const double apk = A[p * E + k], aqk = A[q * E + k];
                    A[p * E + k] = c * apk - sn * aqk;
                    A[q * E + k] = sn * apk + c * aqk;
                }
                for (int k = 0; k < E; ++k) {
                    const double vkp = V[k * E + p], vkq = V[k * E + q];
                    V[k * E + p] = c * vkp - sn * vkq;
                    V[k * E + q] = sn * vkp + c * vkq;
                }
            }
    }
    double lmax = 0.0;
    for (int i = 0; i < E; ++i) lmax = std::max(lmax, A[i * E + i]);
    for (int i = 0; i < E; ++i)
        if (A[i * E + i] < -1e-12 * std::max(lmax, scale)) return -1;
    std::vector<int> keep;
    for (int i = 0; i < E; ++i)
        if (A[i * E + i] > 1e-15 * lmax) keep.push_back(i);
    const int r = (int)keep.size();
    F.assign((size_t)E * std::max(r, 1), 0.0);
    for (int k = 0; k < r; ++k) {
        const double sq = std::sqrt(A[keep[k] * E + keep[k]]);
        for (int e = 0; e < E; ++e) F[(size_t)e * r + k] = V[e * E + keep[k]] * sq;
    }
    return r;
}

// marshal reward terms into a host staging vector; pointers are patched relative to dev_base
int stage_rewards(pilco_ctx* ctx, const pilco_reward_term* rw, int n_rw, int E, std::vector<double>& hp, size_t& off,
                  const double* dev_base, RewardDev* out) {
    for (int i = 0; i < n_rw; ++i) {
        out[i].kind = rw[i].kind;
        out[i].coef = rw[i].coef;
        out[i].F = nullptr;
        out[i].rank = -1;
        if (!rw[i].W) return fail(ctx, PILCO_E_SHAPE, "reward: W is required");
        if (rw[i].kind == PILCO_REWARD_EXPONENTIAL) {
            hp.resize(std::max(hp.size(), off + (size_t)2 * E * E + E));
            memcpy(&hp[off], rw[i].W, sizeof(double) * E * E);
            out[i].W = dev_base + off; off += (size_t)E * E;
            if (rw[i].t) memcpy(&hp[off], rw[i].t, sizeof(double) * E);
            else std::fill(hp.begin() + off, hp.begin() + off + E, 0.0);
            out[i].t = dev_base + off; off += E;
            std::vector<double> F;
            const int r = psd_factor(rw[i].W, E, F);
            out[i].rank = r;
            if (r > 0) {
                memcpy(&hp[off], F.data(), sizeof(double) * E * r);
                out[i].F = dev_base + off;
            } else if (r == 0) {
                out[i].F = dev_base + off;
            }
            off += (size_t)E * E;
        } else if (rw[i].kind == PILCO_REWARD_LINEAR) {
            hp.resize(std::max(hp.size(), off + (size_t)E));
            memcpy(&hp[off], rw[i].W, sizeof(double) * E);
            out[i].W = dev_base + off; off += E;
            out[i].t = out[i].W;
        } else {
            return fail(ctx, PILCO_E_SHAPE, "reward: unknown kind");
        }
    }
    return PILCO_OK;
}

MMModel model_of(const Slot& s) {
    MMModel md{};
    md.Pt = s.M > 0 ? s.Zt.p : s.Xt.p;
    md.ls = s.ls.p;
    md.var = s.var.p;
    md.beta = s.beta.p;
    md.iK = (s.iK_null || s.ignore_iK) ? nullptr : s.iK.p;
    md.n = s.n;
    md.npad = s.npad;
    md.D = s.D;
    md.E = s.E;
    return md;
}

int all_gather_segments(pilco_ctx* ctx, Slot& s) {
    if (ctx->nranks == 1 && !ctx->comm) return PILCO_OK;
    if (!ctx->comm) return fail(ctx, PILCO_E_STATE, "sharded context without communicator: use pilco_gp_shard_pack / pilco_gp_shard_finish");
    double* base = s.wk.gath;
    ncclResult_t r = ncclAllGather(base + (size_t)ctx->rank * s.wk.SEG, base, s.wk.SEG, ncclDouble, ctx->comm, ctx->st);
    if (r != ncclSuccess) return fail(ctx, PILCO_E_RCCL, std::string("ncclAllGather: ") + ncclGetErrorString(r));
    return PILCO_OK;
}

// ---- exact GP: Gram -> Cholesky -> L^{-1} -> iK = L^{-T} L^{-1}, beta = L^{-T} (L^{-1} y)
int factorize_exact(pilco_ctx* ctx, Slot& s) {
    const int E = s.E, npad = s.Npad, nblk = npad / NB;
    const size_t mat = (size_t)npad * npad;
    ENSURE(s.K, E * mat);
    ENSURE(s.Linv, E * mat);
    ENSURE(s.iK, E * mat);
    ENSURE(s.invD, (size_t)E * nblk * NB * NB);
    ENSURE(s.beta, (size_t)E * npad);
    ENSURE(s.vec, (size_t)E * npad);
    hipStream_t st = ctx->st;
    HIPCHK(hipMemsetAsync(ctx->d_info, 0, sizeof(int) * 64, st));
    launch_gram(st, s.Xt.p, npad, s.N, s.Xt.p, npad, s.N, s.D, s.ls.p, s.var.p, E, s.K.p, npad, npad, 1, s.noise.p, 0.0);
    launch_potrf(st, s.K.p, npad, E, s.invD.p, ctx->d_info);
    launch_trtri(st, s.K.p, npad, E, s.invD.p, s.Linv.p, s.iK.p, (long)mat);   // iK is free until the next GEMM
    GemmDesc g{};
    g.A = s.Linv.p; g.lda = npad; g.sA = (long)mat;
    g.B = s.Linv.p; g.ldb = npad; g.sB = (long)mat;
    g.C = s.iK.p; g.ldc = npad; g.sC = (long)mat;
    g.M = npad; g.N = npad; g.K = npad; g.alpha = 1.0; g.beta = 0.0; g.tile_mode = 0; g.k_mode = 1;
    launch_gemm(st, g, true, false, E);
    launch_clear_padding(st, s.iK.p, npad, s.N, E);
    launch_matvec(st, s.Linv.p, npad, E, s.Yt.p, s.vec.p, false);
    launch_matvec(st, s.Linv.p, npad, E, s.vec.p, s.beta.p, true);
    int info[64];
    HIPCHK(hipMemcpyAsync(info, ctx->d_info, sizeof(int) * std::min(E, 64), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    for (int a = 0; a < std::min(E, 64); ++a)
        if (info[a] != 0) {
            ctx->not_pd = a;
            return fail(ctx, PILCO_E_NOT_PD, "Cholesky failed: K + noise*I of output " + std::to_string(a) +
                                                 " is not positive definite (pivot " + std::to_string(info[a]) + ")");
        }
    s.n = s.N;
    s.npad = npad;
    s.iK_null = false;
    return PILCO_OK;
}

}  // namespace

// sparse FITC factorisation lives in fitc.hip
int pilco_factorize_fitc(pilco_ctx* ctx, void* slot_ptr);

extern "C" {

int pilco_abi_version(void) { return PILCO_HIP_ABI_VERSION; }

int pilco_ctx_create(int device, pilco_ctx** out) {
    if (!out) return PILCO_E_SHAPE;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return PILCO_E_HIP;
    if (device < 0 || device >= count) return PILCO_E_SHAPE;
    if (hipSetDevice(device) != hipSuccess) return PILCO_E_HIP;
    pilco_ctx* ctx = new pilco_ctx();
    ctx->device = device;
    ctx->slot[PILCO_SLOT_POLICY].ignore_iK = true;
    if (hipStreamCreateWithFlags(&ctx->st, hipStreamNonBlocking) != hipSuccess ||
        hipMalloc(&ctx->d_info, 64 * sizeof(int)) != hipSuccess || hipEventCreate(&ctx->ev0) != hipSuccess ||
        hipEventCreate(&ctx->ev1) != hipSuccess) {
        delete ctx;
        return PILCO_E_HIP;
    }
    {
        double tab[256];
        const int tn = mm_exp_table_size();
        for (int j = 0; j < tn; ++j) tab[j] = std::exp2((double)j / (double)tn);
        if (ctx->exp_tab.ensure(256) != hipSuccess ||
            hipMemcpy(ctx->exp_tab.p, tab, sizeof(tab), hipMemcpyHostToDevice) != hipSuccess) {
            delete ctx;
            return PILCO_E_HIP;
        }
    }
    if (const char* eg = getenv("PILCO_NO_GRAPH")) ctx->use_graph = (atoi(eg) == 0);
    const char* env = getenv("PILCO_PAIR_KERNEL");
    if (env) ctx->variant = (atoi(env) >= 0 && atoi(env) <= 2) ? atoi(env) : 0;
    *out = ctx;
    return PILCO_OK;
}

int pilco_ctx_destroy(pilco_ctx* ctx) {
    if (!ctx) return PILCO_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->st);
    if (ctx->graph) (void)hipGraphExecDestroy(ctx->graph);
    if (ctx->comm) ncclCommDestroy(ctx->comm);
    for (Slot& s : ctx->slot) {
        for (DevBuf* b : {&s.Xt, &s.Yt, &s.Zt, &s.ls, &s.var, &s.noise, &s.K, &s.Linv, &s.iK, &s.invD, &s.beta, &s.Tscr,
                          &s.vec, &s.Kmn, &s.V2, &s.bwd_mom, &s.bwd_cp, &s.bwd_part, &s.bwd_out, &s.Am, &s.AmInv, &s.AmD, &s.iAt, &s.G, &s.w_in, &s.w_At, &s.w_Bt, &s.w_small,
                          &s.w_part, &s.w_gath, &s.w_out})
            b->release();
    }
    ctx->state.release();
    ctx->params.release();
    ctx->traj.release();
    ctx->tape.release();
    ctx->selftest.release();
    ctx->exp_tab.release();
    for (hipEvent_t e : ctx->pair_events) (void)hipEventDestroy(e);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    if (ctx->d_info) (void)hipFree(ctx->d_info);
    if (ctx->pin) (void)hipHostFree(ctx->pin);
    if (ctx->st) (void)hipStreamDestroy(ctx->st);
    delete ctx;
    return PILCO_OK;
}

const char* pilco_last_error(const pilco_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }
int pilco_last_not_pd_output(const pilco_ctx* ctx) { return ctx ? ctx->not_pd : -1; }

int pilco_set_pair_kernel(pilco_ctx* ctx, int variant) {
    if (!ctx || variant < 0 || variant > 2) return PILCO_E_SHAPE;
    ctx->variant = variant;
    return PILCO_OK;
}

int pilco_selftest(pilco_ctx* ctx) {
    if (!ctx) return PILCO_E_SHAPE;
    HIPCHK(hipSetDevice(ctx->device));
    ENSURE(ctx->selftest, 256);
    double h[256];
    const int r = launch_selftest_mfma(ctx->st, ctx->selftest.p, h, ctx->exp_tab.p);
    if (r < 0) return fail(ctx, PILCO_E_HIP, "selftest launch failed");
    if (r == 100000) return fail(ctx, PILCO_E_STATE, "table-driven exp deviates from the library exp by more than 2 ulp");
    if (r > 0) return fail(ctx, PILCO_E_STATE, "f64 MFMA fragment layout differs from the assumed map at element " + std::to_string(r - 1));
    return PILCO_OK;
}

int pilco_gp_set_data(pilco_ctx* ctx, int slot, const double* X, const double* Y, int N, int D, int E) {
    if (int r = check_slot(ctx, slot)) return r;
    if (!X || !Y || N <= 0 || D <= 0 || E <= 0) return fail(ctx, PILCO_E_SHAPE, "set_data: bad sizes");
    if (D > MAX_D) return fail(ctx, PILCO_E_SHAPE, "set_data: GP input dimension > 32 not supported by this build");
    if (E > MAX_D) return fail(ctx, PILCO_E_SHAPE, "set_data: more than 32 outputs not supported by this build");
    HIPCHK(hipSetDevice(ctx->device));
    Slot& s = ctx->slot[slot];
    const bool reshape = (D != s.D || E != s.E);
    s.N = N; s.D = D; s.E = E;
    s.Npad = round_up(N, NB);
    if (reshape) { s.has_hyp = false; s.M = 0; }
    ENSURE(s.Xt, (size_t)D * s.Npad);
    ENSURE(s.Yt, (size_t)E * s.Npad);
    // stage through a scratch device buffer: X (N,D) -> Xt [D][Npad]; Y (N,E) -> Yt [E][Npad]
    const size_t need = (size_t)N * std::max(D, E);
    ENSURE(s.vec, std::max(need, (size_t)E * s.Npad));
    HIPCHK(hipMemcpyAsync(s.vec.p, X, sizeof(double) * N * D, hipMemcpyHostToDevice, ctx->st));
    launch_transpose_points(ctx->st, s.vec.p, N, D, s.Xt.p, s.Npad);
    HIPCHK(hipStreamSynchronize(ctx->st));
    HIPCHK(hipMemcpyAsync(s.vec.p, Y, sizeof(double) * N * E, hipMemcpyHostToDevice, ctx->st));
    launch_transpose_points(ctx->st, s.vec.p, N, E, s.Yt.p, s.Npad);
    HIPCHK(hipStreamSynchronize(ctx->st));
    s.has_data = true;
    s.factor_valid = false;
    s.user_factors = false;
    if (s.M == 0) { s.n = N; s.npad = s.Npad; }
    s.wk_valid = false;
    return PILCO_OK;
}

int pilco_gp_set_hyp(pilco_ctx* ctx, int slot, const double* lengthscales, const double* variance, const double* noise) {
    if (int r = check_slot(ctx, slot)) return r;
    Slot& s = ctx->slot[slot];
    if (!s.has_data) return fail(ctx, PILCO_E_STATE, "set_hyp before set_data");
    if (!lengthscales || !variance || !noise) return fail(ctx, PILCO_E_SHAPE, "set_hyp: null pointer");
    for (int i = 0; i < s.E * s.D; ++i)
        if (!(lengthscales[i] > 0.0)) return fail(ctx, PILCO_E_SHAPE, "set_hyp: lengthscales must be positive");
    for (int i = 0; i < s.E; ++i)
        if (!(variance[i] > 0.0) || !(noise[i] >= 0.0)) return fail(ctx, PILCO_E_SHAPE, "set_hyp: variance must be positive, noise non-negative");
    HIPCHK(hipSetDevice(ctx->device));
    ENSURE(s.ls, (size_t)s.E * s.D);
    ENSURE(s.var, (size_t)s.E);
    ENSURE(s.noise, (size_t)s.E);
    HIPCHK(hipMemcpyAsync(s.ls.p, lengthscales, sizeof(double) * s.E * s.D, hipMemcpyHostToDevice, ctx->st));
    HIPCHK(hipMemcpyAsync(s.var.p, variance, sizeof(double) * s.E, hipMemcpyHostToDevice, ctx->st));
    HIPCHK(hipMemcpyAsync(s.noise.p, noise, sizeof(double) * s.E, hipMemcpyHostToDevice, ctx->st));
    HIPCHK(hipStreamSynchronize(ctx->st));
    s.has_hyp = true;
    s.factor_valid = false;
    s.user_factors = false;
    return PILCO_OK;
}

int pilco_gp_set_inducing(pilco_ctx* ctx, int slot, const double* Z, int M) {
    if (int r = check_slot(ctx, slot)) return r;
    Slot& s = ctx->slot[slot];
    if (!s.has_data) return fail(ctx, PILCO_E_STATE, "set_inducing before set_data");
    if (M < 0 || (M > 0 && !Z)) return fail(ctx, PILCO_E_SHAPE, "set_inducing: bad arguments");
    HIPCHK(hipSetDevice(ctx->device));
    s.M = M;
    s.factor_valid = false;
    s.user_factors = false;
    s.wk_valid = false;
    if (M == 0) {
        s.n = s.N;
        s.npad = s.Npad;
        return PILCO_OK;
    }
    s.n = M;
    s.npad = round_up(M, NB);
    ENSURE(s.Zt, (size_t)s.D * s.npad);
    ENSURE(s.vec, std::max((size_t)M * s.D, (size_t)s.E * std::max(s.Npad, s.npad)));
    HIPCHK(hipMemcpyAsync(s.vec.p, Z, sizeof(double) * M * s.D, hipMemcpyHostToDevice, ctx->st));
    launch_transpose_points(ctx->st, s.vec.p, M, s.D, s.Zt.p, s.npad);
    HIPCHK(hipStreamSynchronize(ctx->st));
    return PILCO_OK;
}

int pilco_gp_gram(pilco_ctx* ctx, int slot, const double* X1, int N1, const double* X2, int N2, double* out) {
    if (int r = check_slot(ctx, slot)) return r;
    Slot& s = ctx->slot[slot];
    if (!s.has_hyp) return fail(ctx, PILCO_E_STATE, "gram before set_hyp");
    if (!X1 || N1 <= 0 || !out) return fail(ctx, PILCO_E_SHAPE, "gram: bad arguments");
    if (!X2) { X2 = X1; N2 = N1; }
    if (N2 <= 0) return fail(ctx, PILCO_E_SHAPE, "gram: bad arguments");
    HIPCHK(hipSetDevice(ctx->device));
    const int p1 = round_up(N1, NB), p2 = round_up(N2, NB), D = s.D, E = s.E;
    DevBuf raw, t1, t2, K;
    auto cleanup = [&]() { raw.release(); t1.release(); t2.release(); K.release(); };
    if (raw.ensure((size_t)std::max(N1, N2) * D) != hipSuccess || t1.ensure((size_t)D * p1) != hipSuccess ||
        t2.ensure((size_t)D * p2) != hipSuccess || K.ensure((size_t)E * p1 * p2) != hipSuccess) {
        cleanup();
        return fail(ctx, PILCO_E_ALLOC, "gram: hipMalloc failed");
    }
    hipError_t e = hipMemcpyAsync(raw.p, X1, sizeof(double) * N1 * D, hipMemcpyHostToDevice, ctx->st);
    launch_transpose_points(ctx->st, raw.p, N1, D, t1.p, p1);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->st);
    if (e == hipSuccess) e = hipMemcpyAsync(raw.p, X2, sizeof(double) * N2 * D, hipMemcpyHostToDevice, ctx->st);
    launch_transpose_points(ctx->st, raw.p, N2, D, t2.p, p2);
    launch_gram(ctx->st, t1.p, p1, N1, t2.p, p2, N2, D, s.ls.p, s.var.p, E, K.p, p1, p2, 0, nullptr, 0.0);
    if (e == hipSuccess)
        e = hipMemcpy2DAsync(out, sizeof(double) * N2, K.p, sizeof(double) * p2, sizeof(double) * N2, (size_t)N1, hipMemcpyDeviceToHost, ctx->st);
    // the E matrices are strided by p1*p2 on the device and N1*N2 on the host
    for (int a = 1; a < E && e == hipSuccess; ++a)
        e = hipMemcpy2DAsync(out + (size_t)a * N1 * N2, sizeof(double) * N2, K.p + (size_t)a * p1 * p2, sizeof(double) * p2,
                             sizeof(double) * N2, (size_t)N1, hipMemcpyDeviceToHost, ctx->st);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->st);
    cleanup();
    if (e != hipSuccess) return fail(ctx, PILCO_E_HIP, std::string("gram: ") + hipGetErrorString(e));
    return PILCO_OK;
}

int pilco_gp_factorize(pilco_ctx* ctx, int slot) {
    if (int r = check_slot(ctx, slot)) return r;
    Slot& s = ctx->slot[slot];
    if (!s.has_data || !s.has_hyp) return fail(ctx, PILCO_E_STATE, "factorize needs set_data and set_hyp first");
    if (s.factor_valid && !s.user_factors) return PILCO_OK;
    HIPCHK(hipSetDevice(ctx->device));
    ctx->not_pd = -1;
    int r = (s.M > 0) ? pilco_factorize_fitc(ctx, &s) : factorize_exact(ctx, s);
    if (r != PILCO_OK) return r;
    s.factor_valid = true;
    s.user_factors = false;
    s.wk_valid = false;
    return PILCO_OK;
}

int pilco_gp_nlml(pilco_ctx* ctx, int slot, double* nlml, double* grad) {
    if (int r = check_slot(ctx, slot)) return r;
    Slot& s = ctx->slot[slot];
    if (!s.has_data || !s.has_hyp) return fail(ctx, PILCO_E_STATE, "nlml needs set_data and set_hyp first");
    if (s.M > 0) return fail(ctx, PILCO_E_STATE, "nlml: the FITC training objective is not built; exact GP only");
    if (!nlml) return fail(ctx, PILCO_E_SHAPE, "nlml: null pointer");
    HIPCHK(hipSetDevice(ctx->device));
    if (!s.factor_valid || s.user_factors) {
        s.factor_valid = false;
        if (int r = pilco_gp_factorize(ctx, slot)) return r;
    }
    const int E = s.E, D = s.D, npad = s.Npad, N = s.N;
    // after factorize_exact: s.K holds L, s.iK the inverse, s.beta = alpha, s.Yt the targets
    const size_t npart = (size_t)E * (npad / NB) * 34;
    ENSURE(s.vec, std::max((size_t)E * npad * 2, npart + (size_t)E * (D + 2) + 2 * (size_t)E));
    double* d_part = s.vec.p;
    double* d_grad = d_part + npart;
    double* d_logdet = d_grad + (size_t)E * (D + 2);
    launch_logdet(ctx->st, s.K.p, npad, N, E, d_logdet);
    if (grad) launch_nlml_grad(ctx->st, s.Xt.p, npad, N, D, s.ls.p, s.var.p, s.iK.p, s.beta.p, E, d_part, d_grad);
    std::vector<double> hl(E), hb((size_t)E * npad), hy((size_t)E * npad);
    HIPCHK(hipMemcpyAsync(hl.data(), d_logdet, sizeof(double) * E, hipMemcpyDeviceToHost, ctx->st));
    HIPCHK(hipMemcpyAsync(hb.data(), s.beta.p, sizeof(double) * E * npad, hipMemcpyDeviceToHost, ctx->st));
    HIPCHK(hipMemcpyAsync(hy.data(), s.Yt.p, sizeof(double) * E * npad, hipMemcpyDeviceToHost, ctx->st));
    if (grad) HIPCHK(hipMemcpyAsync(grad, d_grad, sizeof(double) * E * (D + 2), hipMemcpyDeviceToHost, ctx->st));
    HIPCHK(hipStreamSynchronize(ctx->st));
    HIPCHK(hipGetLastError());
    for (int a = 0; a < E; ++a) {
        double ya = 0.0;
        for (int i = 0; i < N; ++i) ya += hy[(size_t)a * npad + i] * hb[(size_t)a * npad + i];
        nlml[a] = 0.5 * ya + hl[a] + 0.5 * N * std::log(2.0 * M_PI);
    }
    return PILCO_OK;
}

int pilco_gp_num_points(const pilco_ctx* ctx, int slot) {
    if (!ctx || slot < 0 || slot > 1) return -1;
    return ctx->slot[slot].n;
}

int pilco_gp_get_factors(pilco_ctx* ctx, int slot, double* iK, double* beta) {
    if (int r = check_slot(ctx, slot)) return r;
    Slot& s = ctx->slot[slot];
    if (!s.factor_valid) return fail(ctx, PILCO_E_STATE, "get_factors: no current factorisation");
    HIPCHK(hipSetDevice(ctx->device));
    const int n = s.n, npad = s.npad, E = s.E;
    if (iK) {
        if (s.iK_null) {
            memset(iK, 0, sizeof(double) * (size_t)E * n * n);
        } else {
            for (int a = 0; a < E; ++a)
                HIPCHK(hipMemcpy2DAsync(iK + (size_t)a * n * n, sizeof(double) * n, s.iK.p + (size_t)a * npad * npad,
                                        sizeof(double) * npad, sizeof(double) * n, (size_t)n, hipMemcpyDeviceToHost, ctx->st));
        }
    }
    if (beta)
        HIPCHK(hipMemcpy2DAsync(beta, sizeof(double) * n, s.beta.p, sizeof(double) * npad, sizeof(double) * n, (size_t)E,
                                hipMemcpyDeviceToHost, ctx->st));
    HIPCHK(hipStreamSynchronize(ctx->st));
    return PILCO_OK;
}

int pilco_gp_set_factors(pilco_ctx* ctx, int slot, const double* iK, const double* beta) {
    if (int r = check_slot(ctx, slot)) return r;
    Slot& s = ctx->slot[slot];
    if (!s.has_data || !s.has_hyp) return fail(ctx, PILCO_E_STATE, "set_factors needs set_data and set_hyp first");
    if (!beta) return fail(ctx, PILCO_E_SHAPE, "set_factors: beta is required");
    HIPCHK(hipSetDevice(ctx->device));
    const int n = s.n, npad = s.npad, E = s.E;
    ENSURE(s.beta, (size_t)E * npad);
    HIPCHK(hipMemsetAsync(s.beta.p, 0, sizeof(double) * E * npad, ctx->st));
    HIPCHK(hipMemcpy2DAsync(s.beta.p, sizeof(double) * npad, beta, sizeof(double) * n, sizeof(double) * n, (size_t)E,
                            hipMemcpyHostToDevice, ctx->st));
    if (iK) {
        ENSURE(s.iK, (size_t)E * npad * npad);
        HIPCHK(hipMemsetAsync(s.iK.p, 0, sizeof(double) * E * npad * npad, ctx->st));
        for (int a = 0; a < E; ++a)
            HIPCHK(hipMemcpy2DAsync(s.iK.p + (size_t)a * npad * npad, sizeof(double) * npad, iK + (size_t)a * n * n,
                                    sizeof(double) * n, sizeof(double) * n, (size_t)n, hipMemcpyHostToDevice, ctx->st));
    }
    HIPCHK(hipStreamSynchronize(ctx->st));
    s.iK_null = (iK == nullptr);
    s.factor_valid = true;
    s.user_factors = true;
    s.wk_valid = false;  // the stream-K geometry depends on whether an iK stream exists
    return PILCO_OK;
}

int pilco_gp_predict(pilco_ctx* ctx, int slot, const double* m, const double* s_in, double* M, double* S, double* V) {
    if (int r = check_slot(ctx, slot)) return r;
    Slot& s = ctx->slot[slot];
    if (!s.factor_valid) return fail(ctx, PILCO_E_STATE, "predict: no current factorisation (call pilco_gp_factorize)");
    if (!m || !s_in || !M || !S || !V) return fail(ctx, PILCO_E_SHAPE, "predict: null pointer");
    HIPCHK(hipSetDevice(ctx->device));
    if (int r = build_work(ctx, s)) return r;
    const int D = s.D, E = s.E;
    HIPCHK(hipMemcpyAsync(s.wk.in_m, m, sizeof(double) * D, hipMemcpyHostToDevice, ctx->st));
    HIPCHK(hipMemcpyAsync(s.wk.in_s, s_in, sizeof(double) * D * D, hipMemcpyHostToDevice, ctx->st));
    const MMModel md = model_of(s);
    if (s.wk.PL > 0) {
        launch_mm_prep(ctx->st, md, s.wk);
        launch_mm_pair(ctx->st, md, s.wk, ctx->variant);
    }
    GlueArgs g{};
    g.E = E; g.D = D; g.U = 0;
    g.wk = s.wk;
    g.var = s.var.p;
    if (ctx->nranks == 1 && !ctx->comm) {
        g.flags = GF_PACK | GF_ASSEMBLE;
        launch_glue(ctx->st, g);
    } else {
        g.flags = GF_PACK;
        launch_glue(ctx->st, g);
        if (int r = all_gather_segments(ctx, s)) return r;
        g.flags = GF_ASSEMBLE;
        launch_glue(ctx->st, g);
    }
    HIPCHK(hipMemcpyAsync(M, s.wk.out_M, sizeof(double) * E, hipMemcpyDeviceToHost, ctx->st));
    HIPCHK(hipMemcpyAsync(S, s.wk.out_S, sizeof(double) * E * E, hipMemcpyDeviceToHost, ctx->st));
    HIPCHK(hipMemcpyAsync(V, s.wk.out_V, sizeof(double) * D * E, hipMemcpyDeviceToHost, ctx->st));
    HIPCHK(hipStreamSynchronize(ctx->st));
    HIPCHK(hipGetLastError());
    return PILCO_OK;
}

}  // extern "C"

// ------------------------------------------------------------------ rollout
namespace {

struct RolloutPlan {
    GlueArgs g{};
    double* st[2] = {nullptr, nullptr};  // double-buffered state: m_x[E] | s_x[E*E]
    int E = 0, D = 0, U = 0;
};

int setup_rollout(pilco_ctx* ctx, const pilco_policy* pol, const pilco_reward_term* rw, int n_rw, int H, bool want_traj,
                  RolloutPlan& plan) {
    Slot& s = ctx->slot[0];
    if (!s.factor_valid) return fail(ctx, PILCO_E_STATE, "rollout: dynamics model has no current factorisation");
    if (!pol) return fail(ctx, PILCO_E_SHAPE, "rollout: null policy");
    const int E = s.E, D = s.D, U = D - E;
    if (pol->state_dim != E || pol->control_dim != U || U < 0)
        return fail(ctx, PILCO_E_SHAPE, "rollout: policy dims do not match the model (state_dim must be E, control_dim D-E)");
    if (pol->kind == PILCO_POLICY_NONE && U != 0) return fail(ctx, PILCO_E_SHAPE, "rollout: policy NONE needs D == E");
    if (pol->kind == PILCO_POLICY_LINEAR && (U == 0 || !pol->W || !pol->b)) return fail(ctx, PILCO_E_SHAPE, "rollout: linear policy needs W, b and control_dim > 0");
    if (pol->kind == PILCO_POLICY_RBF) {
        Slot& ps = ctx->slot[PILCO_SLOT_POLICY];
        if (!ps.factor_valid) return fail(ctx, PILCO_E_STATE, "rollout: RBF policy slot has no current factorisation");
        if (ps.D != E || ps.E != U || U == 0) return fail(ctx, PILCO_E_SHAPE, "rollout: RBF policy GP must map state_dim -> control_dim");
        if (ctx->nranks != 1) return fail(ctx, PILCO_E_STATE, "rollout: the RBF policy is evaluated unsharded; use one rank");
        if (int r = build_work(ctx, ps)) return r;
    }
    if (pol->kind < 0 || pol->kind > 2) return fail(ctx, PILCO_E_SHAPE, "rollout: unknown policy kind");
    if (n_rw < 0 || n_rw > MAX_REWARD_TERMS || (n_rw > 0 && !rw)) return fail(ctx, PILCO_E_SHAPE, "rollout: 0..4 reward terms supported");
    if (int r = build_work(ctx, s)) return r;
    // state: 2 x (m_x[E] s_x[E*E]) | s1[E*D] | reward[1]
    const size_t n_state = 2 * ((size_t)E + E * E) + (size_t)E * D + 1 + 8;
    ENSURE(ctx->state, n_state);
    // params: W[U*E] b[U] maxact[U] then per reward W[E*E] t[E] F[E*E]
    const size_t n_par = (size_t)U * E + 2 * U + (size_t)MAX_REWARD_TERMS * (2 * E * E + E) + 8;
    ENSURE(ctx->params, n_par);
    if (want_traj) ENSURE(ctx->traj, (size_t)(H + 1) * (E + E * E));
    std::vector<double> hp(n_par, 0.0);
    size_t off = 0;
    GlueArgs& g = plan.g;
    g = GlueArgs{};
    g.E = E; g.D = D; g.U = U;
    g.wk = s.wk;
    g.var = s.var.p;
    plan.st[0] = ctx->state.p;
    plan.st[1] = ctx->state.p + (E + E * E);
    g.s1 = ctx->state.p + 2 * (E + E * E);
    g.reward = g.s1 + (size_t)E * D;
    g.traj = want_traj ? ctx->traj.p : nullptr;
    g.pol_kind = pol->kind;
    g.squash = pol->squash;
    if (pol->kind == PILCO_POLICY_RBF) {
        g.pwk = ctx->slot[PILCO_SLOT_POLICY].wk;
        g.pvar = ctx->slot[PILCO_SLOT_POLICY].var.p;
        for (int u = 0; u < U; ++u) hp[off + u] = pol->max_action ? pol->max_action[u] : 1.0;
        g.maxact = ctx->params.p + off; off += U;
    }
    if (pol->kind == PILCO_POLICY_LINEAR) {
        memcpy(&hp[off], pol->W, sizeof(double) * U * E);
        g.W = ctx->params.p + off; off += (size_t)U * E;
        memcpy(&hp[off], pol->b, sizeof(double) * U);
        g.b = ctx->params.p + off; off += U;
        for (int u = 0; u < U; ++u) hp[off + u] = pol->max_action ? pol->max_action[u] : 1.0;
        g.maxact = ctx->params.p + off; off += U;
    }
    g.n_rewards = n_rw;
    g.rew_out = nullptr;
    if (int r = stage_rewards(ctx, rw, n_rw, E, hp, off, ctx->params.p, g.rw)) return r;
    if (hp.size() > n_par) return fail(ctx, PILCO_E_ALLOC, "rollout: parameter staging overflow");
    HIPCHK(hipMemcpyAsync(ctx->params.p, hp.data(), sizeof(double) * n_par, hipMemcpyHostToDevice, ctx->st));
    HIPCHK(hipStreamSynchronize(ctx->st));  // hp is a local vector
    plan.E = E; plan.D = D; plan.U = U;
    return PILCO_OK;
}

// enqueue one full rollout on the stream (initial state already in plan.st[0]); the final
// state ends up in plan.st[H & 1].  The reward of state t (pilco.py:133) is evaluated by the
// second workgroup of the glue launch that turns state t into state t+1.
int enqueue_rollout(pilco_ctx* ctx, RolloutPlan& plan, int H, std::vector<hipEvent_t>* pair_ev) {
    Slot& s = ctx->slot[0];
    const MMModel md = model_of(s);
    const int E = plan.E;
    GlueArgs g = plan.g;
    const bool rew = g.n_rewards > 0 && !(s.wk.abl & 8);
    HIPCHK(hipMemsetAsync(g.reward, 0, sizeof(double), ctx->st));
    g.step = 0;
    g.m_x = plan.st[0];
    g.s_x = plan.st[0] + E;
    g.m_out = nullptr;
    g.s_out = nullptr;
    const bool rbf = (g.pol_kind == PILCO_POLICY_RBF);
    Slot& ps = ctx->slot[PILCO_SLOT_POLICY];
    const MMModel pmd = rbf ? model_of(ps) : MMModel{};
    // RBF policy (controllers.py:108-121): the glue that produced the state hands it to the policy GP
    // (GF_RBF_PRE), the policy's moment matching runs as its own prep/pair, a second glue squashes and
    // builds the joint Gaussian (GF_RBF_POST | GF_POLICY).
    auto policy_stage = [&](GlueArgs& ga) {
        launch_mm_prep(ctx->st, pmd, ps.wk);
        launch_mm_pair(ctx->st, pmd, ps.wk, ctx->variant);
        const int keep = ga.flags;
        ga.flags = GF_RBF_POST | GF_POLICY;
        launch_glue(ctx->st, ga);
        ga.flags = keep;
    };
    g.flags = GF_TRAJ | (H > 0 ? (rbf ? GF_RBF_PRE : GF_POLICY) : 0);
    launch_glue(ctx->st, g);
    if (rbf && H > 0) policy_stage(g);
    size_t evi = 0;
    for (int t = 0; t < H; ++t) {
        if (s.wk.PL > 0) {
            PrepReward pr{};
            if (rew) {   // reward of state t rides in a spare workgroup of this step's prep launch
                pr.n = g.n_rewards;
                pr.E = E;
                for (int i = 0; i < g.n_rewards; ++i) pr.rw[i] = g.rw[i];
                pr.m_x = plan.st[t & 1];
                pr.s_x = plan.st[t & 1] + E;
                pr.reward = g.reward;
            }
            launch_mm_prep(ctx->st, md, s.wk, rew ? &pr : nullptr);
            if (ctx->dbg && (s.wk.abl & 64)) launch_stamp(ctx->st, ctx->dbg, 30);
            if (pair_ev) HIPCHK(hipEventRecord((*pair_ev)[evi++], ctx->st));
            launch_mm_pair(ctx->st, md, s.wk, ctx->variant);
            if (pair_ev) HIPCHK(hipEventRecord((*pair_ev)[evi++], ctx->st));
        }
        g.step = t + 1;
        g.m_x = plan.st[t & 1];
        g.s_x = plan.st[t & 1] + E;
        g.m_out = plan.st[(t + 1) & 1];
        g.s_out = plan.st[(t + 1) & 1] + E;
        const bool more = t + 1 < H;
        const int tail = GF_ASSEMBLE | GF_PROPAGATE | GF_TRAJ | (more ? (rbf ? GF_RBF_PRE : GF_POLICY) : 0);
        if (ctx->nranks == 1 && !ctx->comm) {
            g.flags = GF_PACK | tail;
        } else {
            g.flags = GF_PACK;
            launch_glue(ctx->st, g);
            if (int r = all_gather_segments(ctx, s)) return r;
            g.flags = tail;
        }
        launch_glue(ctx->st, g, rew && s.wk.PL == 0);   // (a rank without pairs keeps the reward in the glue launch)
        if (rbf && more) {  // the policy stage reads the NEW state
            g.m_x = g.m_out;
            g.s_x = g.s_out;
            policy_stage(g);
        }
    }
    return PILCO_OK;
}

// Run one rollout: replay the cached hipGraph when the launch sequence is unchanged
// (same buffers, sizes, horizon, policy / reward structure), otherwise (re)capture it.
int run_rollout(pilco_ctx* ctx, RolloutPlan& plan, int H) {
    Slot& s = ctx->slot[0];
    // With a communicator the captured graph contains the ncclAllGather nodes (RCCL supports stream
    // capture); if capture or instantiation fails the rollout falls back to eager launches for good.
    const bool sharded = (ctx->nranks != 1 || ctx->comm);
    if (!ctx->use_graph || (sharded && (!ctx->comm || ctx->graph_rccl_failed)) || (ctx->dbg && !getenv("PILCO_DBG_GRAPH")))
        return enqueue_rollout(ctx, plan, H, nullptr);
    const GlueArgs& g = plan.g;
    std::vector<unsigned long long> key = {
        (unsigned long long)H, (unsigned long long)g.pol_kind, (unsigned long long)g.n_rewards, (unsigned long long)g.squash,
        (unsigned long long)ctx->variant, (unsigned long long)(uintptr_t)plan.st[0], (unsigned long long)(uintptr_t)g.s1,
        (unsigned long long)(uintptr_t)g.traj, (unsigned long long)(uintptr_t)g.tape, (unsigned long long)(uintptr_t)g.W, (unsigned long long)(uintptr_t)g.maxact,
        (unsigned long long)(uintptr_t)s.w_part.p, (unsigned long long)(uintptr_t)s.w_At.p, (unsigned long long)(uintptr_t)s.w_Bt.p,
        (unsigned long long)(uintptr_t)s.w_small.p, (unsigned long long)(uintptr_t)s.w_gath.p, (unsigned long long)(uintptr_t)s.w_out.p,
        (unsigned long long)(uintptr_t)s.w_in.p, (unsigned long long)(uintptr_t)s.beta.p,
        (unsigned long long)(uintptr_t)s.iK.p, (unsigned long long)s.iK_null, (unsigned long long)(uintptr_t)s.Xt.p,
        (unsigned long long)(uintptr_t)s.Zt.p, (unsigned long long)(uintptr_t)s.ls.p, (unsigned long long)s.n,
        (unsigned long long)s.wk.sk_waves, (unsigned long long)s.wk.NT, (unsigned long long)s.wk.NCH, (unsigned long long)s.wk.NCHM, (unsigned long long)s.wk.abl,
        (unsigned long long)(uintptr_t)ctx->slot[1].w_part.p, (unsigned long long)(uintptr_t)ctx->slot[1].w_At.p,
        (unsigned long long)(uintptr_t)ctx->slot[1].beta.p, (unsigned long long)(uintptr_t)ctx->slot[1].Xt.p,
        (unsigned long long)ctx->slot[1].n,
        (unsigned long long)ctx->slot[1].wk.sk_waves, (unsigned long long)(uintptr_t)ctx->slot[1].w_small.p,
        (unsigned long long)(uintptr_t)ctx->slot[1].w_in.p, (unsigned long long)(uintptr_t)ctx->slot[1].ls.p};
    for (int i = 0; i < g.n_rewards; ++i) {
        key.push_back((unsigned long long)g.rw[i].kind);
        key.push_back((unsigned long long)(long long)g.rw[i].rank);
        key.push_back((unsigned long long)(uintptr_t)g.rw[i].W);
        unsigned long long cbits;
        memcpy(&cbits, &g.rw[i].coef, sizeof(cbits));
        key.push_back(cbits);
    }
    if (!ctx->graph || key != ctx->graph_key) {
        if (ctx->graph) {
            (void)hipGraphExecDestroy(ctx->graph);
            ctx->graph = nullptr;
        }
        // warm the per-kernel one-time host configuration outside the capture
        if (int r = enqueue_rollout(ctx, plan, H > 0 ? 1 : 0, nullptr)) return r;
        HIPCHK(hipStreamSynchronize(ctx->st));
        hipGraph_t graph = nullptr;
        HIPCHK(hipStreamBeginCapture(ctx->st, hipStreamCaptureModeThreadLocal));
        const int rc = enqueue_rollout(ctx, plan, H, nullptr);
        hipError_t e = hipStreamEndCapture(ctx->st, &graph);
        if (rc != PILCO_OK) {
            if (graph) (void)hipGraphDestroy(graph);
            if (sharded) {
                ctx->graph_rccl_failed = true;
                (void)hipGetLastError();
                return -1;
            }
            return rc;
        }
        if (e != hipSuccess) {
            if (sharded) {  // not fatal: run this and all later sharded rollouts eagerly
                ctx->graph_rccl_failed = true;
                (void)hipGetLastError();
                return -1;
            }
            return fail(ctx, PILCO_E_HIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(e));
        }
        e = hipGraphInstantiate(&ctx->graph, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        if (e != hipSuccess) {
            ctx->graph = nullptr;
            if (sharded) {
                ctx->graph_rccl_failed = true;
                (void)hipGetLastError();
                return -1;
            }
            return fail(ctx, PILCO_E_HIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(e));
        }
        ctx->graph_key = key;
        // the warm-up rollout above overwrote the initial state: the caller re-uploads it (see callers)
        return -1;
    }
    HIPCHK(hipGraphLaunch(ctx->graph, ctx->st));
    return PILCO_OK;
}

}  // namespace

extern "C" {

int pilco_rollout(pilco_ctx* ctx, const pilco_policy* policy, const pilco_reward_term* rewards, int n_rewards,
                  const double* m0, const double* S0, int H, double* mH, double* SH, double* reward, double* traj) {
    if (!ctx) return PILCO_E_SHAPE;
    if (!m0 || !S0 || !mH || !SH || !reward || H < 0) return fail(ctx, PILCO_E_SHAPE, "rollout: bad arguments");
    HIPCHK(hipSetDevice(ctx->device));
    RolloutPlan plan;
    if (int r = setup_rollout(ctx, policy, rewards, n_rewards, H, traj != nullptr, plan)) return r;
    const int E = plan.E;
    for (int attempt = 0; attempt < 2; ++attempt) {
        HIPCHK(hipMemcpyAsync(plan.st[0], m0, sizeof(double) * E, hipMemcpyHostToDevice, ctx->st));
        HIPCHK(hipMemcpyAsync(plan.st[0] + E, S0, sizeof(double) * E * E, hipMemcpyHostToDevice, ctx->st));
        const int r = run_rollout(ctx, plan, H);
        if (r == -1) continue;  // graph was just (re)captured: upload the state again and replay it
        if (r != PILCO_OK) return r;
        break;
    }
    HIPCHK(hipMemcpyAsync(mH, plan.st[H & 1], sizeof(double) * E, hipMemcpyDeviceToHost, ctx->st));
    HIPCHK(hipMemcpyAsync(SH, plan.st[H & 1] + E, sizeof(double) * E * E, hipMemcpyDeviceToHost, ctx->st));
    HIPCHK(hipMemcpyAsync(reward, plan.g.reward, sizeof(double), hipMemcpyDeviceToHost, ctx->st));
    if (traj)
        HIPCHK(hipMemcpyAsync(traj, ctx->traj.p, sizeof(double) * (size_t)(H + 1) * (E + E * E), hipMemcpyDeviceToHost, ctx->st));
    HIPCHK(hipStreamSynchronize(ctx->st));
    HIPCHK(hipGetLastError());
    return PILCO_OK;
}

int pilco_propagate(pilco_ctx* ctx, const pilco_policy* policy, const double* m_x, const double* s_x, double* M_x, double* S_x) {
    double r = 0.0;
    return pilco_rollout(ctx, policy, nullptr, 0, m_x, s_x, 1, M_x, S_x, &r, nullptr);
}

int pilco_policy_action(pilco_ctx* ctx, const pilco_policy* policy, const double* m, const double* s_in, double* M, double* S, double* V) {
    if (!ctx) return PILCO_E_SHAPE;
    if (!policy || !m || !s_in || !M || !S || !V) return fail(ctx, PILCO_E_SHAPE, "policy_action: null pointer");
    if (policy->kind != PILCO_POLICY_LINEAR && policy->kind != PILCO_POLICY_RBF) return fail(ctx, PILCO_E_SHAPE, "policy_action: policy kind must be LINEAR or RBF");
    HIPCHK(hipSetDevice(ctx->device));
    const int E = policy->state_dim, U = policy->control_dim;
    if (E <= 0 || U <= 0 || E > MAX_D || U > MAX_D) return fail(ctx, PILCO_E_SHAPE, "policy_action: bad dims");
    if (policy->kind == PILCO_POLICY_RBF) {
        Slot& ps = ctx->slot[PILCO_SLOT_POLICY];
        if (!ps.factor_valid) return fail(ctx, PILCO_E_STATE, "policy_action: RBF policy slot has no current factorisation");
        if (ps.D != E || ps.E != U) return fail(ctx, PILCO_E_SHAPE, "policy_action: RBF policy GP must map state_dim -> control_dim");
        if (int r = build_work(ctx, ps)) return r;
        const size_t n_st = (size_t)E + E * E + U + (U + U * U + (size_t)E * U);
        ENSURE(ctx->state, n_st + 8);
        std::vector<double> h(n_st, 0.0);
        memcpy(&h[0], m, sizeof(double) * E);
        memcpy(&h[E], s_in, sizeof(double) * E * E);
        size_t off = (size_t)E + E * E;
        GlueArgs g{};
        g.E = E; g.D = E + U; g.U = U;
        g.m_x = ctx->state.p;
        g.s_x = ctx->state.p + E;
        for (int u = 0; u < U; ++u) h[off + u] = policy->max_action ? policy->max_action[u] : 1.0;
        g.maxact = ctx->state.p + off; off += U;
        g.act_out = ctx->state.p + off;
        g.pol_kind = PILCO_POLICY_RBF;
        g.squash = policy->squash;
        g.pwk = ps.wk;
        g.pvar = ps.var.p;
        HIPCHK(hipMemcpyAsync(ctx->state.p, h.data(), sizeof(double) * n_st, hipMemcpyHostToDevice, ctx->st));
        HIPCHK(hipMemcpyAsync(ps.wk.in_m, m, sizeof(double) * E, hipMemcpyHostToDevice, ctx->st));
        HIPCHK(hipMemcpyAsync(ps.wk.in_s, s_in, sizeof(double) * E * E, hipMemcpyHostToDevice, ctx->st));
        const MMModel pmd = model_of(ps);
        launch_mm_prep(ctx->st, pmd, ps.wk);
        launch_mm_pair(ctx->st, pmd, ps.wk, ctx->variant);
        g.flags = GF_RBF_POST | GF_POLICY;
        launch_glue(ctx->st, g);
        std::vector<double> o((size_t)U + U * U + (size_t)E * U);
        HIPCHK(hipMemcpyAsync(o.data(), g.act_out, sizeof(double) * o.size(), hipMemcpyDeviceToHost, ctx->st));
        HIPCHK(hipStreamSynchronize(ctx->st));
        HIPCHK(hipGetLastError());
        memcpy(M, &o[0], sizeof(double) * U);
        memcpy(S, &o[U], sizeof(double) * U * U);
        memcpy(V, &o[(size_t)U + U * U], sizeof(double) * E * U);
        return PILCO_OK;
    }
    if (!policy->W || !policy->b) return fail(ctx, PILCO_E_SHAPE, "policy_action: linear policy needs W and b");
    const size_t n_state = (size_t)E + E * E + (size_t)U * E + 2 * U + (U + U * U + (size_t)E * U);
    ENSURE(ctx->state, n_state + 8);
    std::vector<double> h(n_state, 0.0);
    memcpy(&h[0], m, sizeof(double) * E);
    memcpy(&h[E], s_in, sizeof(double) * E * E);
    size_t off = (size_t)E + E * E;
    GlueArgs g{};
    g.E = E; g.D = E + U; g.U = U;
    g.m_x = ctx->state.p;
    g.s_x = g.m_x + E;
    memcpy(&h[off], policy->W, sizeof(double) * U * E);
    g.W = ctx->state.p + off; off += (size_t)U * E;
    memcpy(&h[off], policy->b, sizeof(double) * U);
    g.b = ctx->state.p + off; off += U;
    for (int u = 0; u < U; ++u) h[off + u] = policy->max_action ? policy->max_action[u] : 1.0;
    g.maxact = ctx->state.p + off; off += U;
    g.act_out = ctx->state.p + off;
    g.pol_kind = PILCO_POLICY_LINEAR;
    g.squash = policy->squash;
    g.flags = GF_POLICY;
    HIPCHK(hipMemcpyAsync(ctx->state.p, h.data(), sizeof(double) * n_state, hipMemcpyHostToDevice, ctx->st));
    launch_glue(ctx->st, g);
    std::vector<double> o((size_t)U + U * U + (size_t)E * U);
    HIPCHK(hipMemcpyAsync(o.data(), g.act_out, sizeof(double) * o.size(), hipMemcpyDeviceToHost, ctx->st));
    HIPCHK(hipStreamSynchronize(ctx->st));
    HIPCHK(hipGetLastError());
    memcpy(M, &o[0], sizeof(double) * U);
    memcpy(S, &o[U], sizeof(double) * U * U);
    memcpy(V, &o[(size_t)U + U * U], sizeof(double) * E * U);
    return PILCO_OK;
}

int pilco_reward_eval(pilco_ctx* ctx, const pilco_reward_term* rewards, int n_rewards, int state_dim, const double* m,
                      const double* s_in, double* muR, double* sR) {
    if (!ctx) return PILCO_E_SHAPE;
    if (!rewards || n_rewards <= 0 || n_rewards > MAX_REWARD_TERMS || !m || !s_in || !muR || !sR || state_dim <= 0 || state_dim > MAX_D)
        return fail(ctx, PILCO_E_SHAPE, "reward_eval: bad arguments");
    HIPCHK(hipSetDevice(ctx->device));
    const int E = state_dim;
    const size_t n = (size_t)E + E * E + (size_t)n_rewards * (2 * E * E + E) + 2;
    ENSURE(ctx->state, n + 8);
    std::vector<double> h(n, 0.0);
    memcpy(&h[0], m, sizeof(double) * E);
    memcpy(&h[E], s_in, sizeof(double) * E * E);
    size_t off = (size_t)E + E * E;
    GlueArgs g{};
    g.E = E; g.D = E; g.U = 0;
    g.m_x = ctx->state.p;
    g.s_x = ctx->state.p + E;
    g.n_rewards = n_rewards;
    if (int r = stage_rewards(ctx, rewards, n_rewards, E, h, off, ctx->state.p, g.rw)) return r;
    if (h.size() < off + 2) h.resize(off + 2, 0.0);
    if (h.size() + 8 > ctx->state.cap) return fail(ctx, PILCO_E_ALLOC, "reward_eval: staging overflow");
    g.rew_out = ctx->state.p + off;
    g.flags = 0;  // workgroup 0 idles; workgroup 1 evaluates mean and variance
    HIPCHK(hipMemcpyAsync(ctx->state.p, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice, ctx->st));
    launch_glue(ctx->st, g, true);
    double o[2];
    HIPCHK(hipMemcpyAsync(o, g.rew_out, sizeof(double) * 2, hipMemcpyDeviceToHost, ctx->st));
    HIPCHK(hipStreamSynchronize(ctx->st));
    HIPCHK(hipGetLastError());
    *muR = o[0];
    *sR = o[1];
    return PILCO_OK;
}

int pilco_rollout_timed(pilco_ctx* ctx, const pilco_policy* policy, const pilco_reward_term* rewards, int n_rewards,
                        const double* m0, const double* S0, int H, int reps, double* mH, double* SH, double* reward,
                        float* ms_total, float* ms_pair, int* n_pair_launches) {
    if (!ctx) return PILCO_E_SHAPE;
    if (!m0 || !S0 || !mH || !SH || !reward || H < 0 || reps <= 0 || !ms_total) return fail(ctx, PILCO_E_SHAPE, "rollout_timed: bad arguments");
    HIPCHK(hipSetDevice(ctx->device));
    RolloutPlan plan;
    if (int r = setup_rollout(ctx, policy, rewards, n_rewards, H, false, plan)) return r;
    const int E = plan.E;
    ENSURE(ctx->selftest, (size_t)E + E * E + 256);
    double* init = ctx->selftest.p + 256;  // device copy of (m0, S0) so that the timed region has no host traffic
    HIPCHK(hipMemcpyAsync(init, m0, sizeof(double) * E, hipMemcpyHostToDevice, ctx->st));
    HIPCHK(hipMemcpyAsync(init + E, S0, sizeof(double) * E * E, hipMemcpyHostToDevice, ctx->st));
    HIPCHK(hipStreamSynchronize(ctx->st));
    {   // make sure the graph exists before the timed region
        HIPCHK(hipMemcpyAsync(plan.st[0], init, sizeof(double) * (E + E * E), hipMemcpyDeviceToDevice, ctx->st));
        const int r = run_rollout(ctx, plan, H);
        if (r != PILCO_OK && r != -1) return r;
        HIPCHK(hipStreamSynchronize(ctx->st));
    }
    HIPCHK(hipEventRecord(ctx->ev0, ctx->st));
    for (int rep = 0; rep < reps; ++rep) {
        HIPCHK(hipMemcpyAsync(plan.st[0], init, sizeof(double) * (E + E * E), hipMemcpyDeviceToDevice, ctx->st));
        int r = run_rollout(ctx, plan, H);
        if (r == -1) {  // only possible when the sharded capture fell back to eager mode: redo this rollout
            HIPCHK(hipMemcpyAsync(plan.st[0], init, sizeof(double) * (E + E * E), hipMemcpyDeviceToDevice, ctx->st));
            r = run_rollout(ctx, plan, H);
        }
        if (r != PILCO_OK) return r;
    }
    HIPCHK(hipEventRecord(ctx->ev1, ctx->st));
    HIPCHK(hipEventSynchronize(ctx->ev1));
    HIPCHK(hipEventElapsedTime(ms_total, ctx->ev0, ctx->ev1));
    if (ms_pair) {
        // second pass with an event pair around every pair-kernel launch (perturbs the total, so timed separately)
        const size_t need = (size_t)2 * H;
        while (ctx->pair_events.size() < need) {
            hipEvent_t e;
            HIPCHK(hipEventCreate(&e));
            ctx->pair_events.push_back(e);
        }
        HIPCHK(hipMemcpyAsync(plan.st[0], init, sizeof(double) * (E + E * E), hipMemcpyDeviceToDevice, ctx->st));
        if (int r = enqueue_rollout(ctx, plan, H, &ctx->pair_events)) return r;
        HIPCHK(hipStreamSynchronize(ctx->st));
        float tot = 0.f;
        int cnt = 0;
        if (ctx->slot[0].wk.PL > 0)
            for (int t = 0; t < H; ++t) {
                float ms = 0.f;
                HIPCHK(hipEventElapsedTime(&ms, ctx->pair_events[2 * t], ctx->pair_events[2 * t + 1]));
                tot += ms;
                ++cnt;
            }
        *ms_pair = tot;
        if (n_pair_launches) *n_pair_launches = cnt;
    }
    HIPCHK(hipMemcpyAsync(mH, plan.st[H & 1], sizeof(double) * E, hipMemcpyDeviceToHost, ctx->st));
    HIPCHK(hipMemcpyAsync(SH, plan.st[H & 1] + E, sizeof(double) * E * E, hipMemcpyDeviceToHost, ctx->st));
    HIPCHK(hipMemcpyAsync(reward, plan.g.reward, sizeof(double), hipMemcpyDeviceToHost, ctx->st));
    HIPCHK(hipStreamSynchronize(ctx->st));
    HIPCHK(hipGetLastError());
    return PILCO_OK;
}

int pilco_factorize_timed(pilco_ctx* ctx, int slot, int reps, float* ms_each) {
    if (int r = check_slot(ctx, slot)) return r;
    if (reps <= 0 || !ms_each) return fail(ctx, PILCO_E_SHAPE, "factorize_timed: bad arguments");
    Slot& s = ctx->slot[slot];
    s.factor_valid = false;
    if (int r = pilco_gp_factorize(ctx, slot)) return r;  // warm-up, allocations
    HIPCHK(hipEventRecord(ctx->ev0, ctx->st));
    for (int i = 0; i < reps; ++i) {
        s.factor_valid = false;
        if (int r = pilco_gp_factorize(ctx, slot)) return r;
    }
    HIPCHK(hipEventRecord(ctx->ev1, ctx->st));
    HIPCHK(hipEventSynchronize(ctx->ev1));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    *ms_each = ms / reps;
    return PILCO_OK;
}

// ------------------------------------------------------------------ multi-GPU
int pilco_comm_unique_id(void* id128) {
    if (!id128) return PILCO_E_SHAPE;
    static_assert(sizeof(ncclUniqueId) <= PILCO_COMM_ID_BYTES, "ncclUniqueId larger than the ABI slot");
    ncclUniqueId id;
    if (ncclGetUniqueId(&id) != ncclSuccess) return PILCO_E_RCCL;
    memset(id128, 0, PILCO_COMM_ID_BYTES);
    memcpy(id128, &id, sizeof(id));
    return PILCO_OK;
}

int pilco_shard_set(pilco_ctx* ctx, int rank, int nranks) {
    if (!ctx || nranks <= 0 || rank < 0 || rank >= nranks) return fail(ctx, PILCO_E_SHAPE, "shard_set: bad rank / nranks");
    ctx->rank = rank;
    ctx->nranks = nranks;
    for (Slot& s : ctx->slot) s.wk_valid = false;
    return PILCO_OK;
}

int pilco_comm_init(pilco_ctx* ctx, const void* id128, int rank, int nranks) {
    if (!ctx || !id128) return PILCO_E_SHAPE;
    if (int r = pilco_shard_set(ctx, rank, nranks)) return r;
    HIPCHK(hipSetDevice(ctx->device));
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    ncclResult_t r = ncclCommInitRank(&ctx->comm, nranks, id, rank);
    if (r != ncclSuccess) {
        ctx->comm = nullptr;
        return fail(ctx, PILCO_E_RCCL, std::string("ncclCommInitRank: ") + ncclGetErrorString(r));
    }
    return PILCO_OK;
}

int pilco_shard_owner_of_pair(const pilco_ctx* ctx, int pair_index) {
    if (!ctx) return -1;
    const Slot& s = ctx->slot[0];
    if (pair_index < 0 || pair_index >= (int)s.pair_owner.size()) return -1;
    return s.pair_owner[pair_index];
}

int pilco_debug_blocks(pilco_ctx* ctx, unsigned long long* out, int n) {
    if (!ctx || !ctx->dbg || !out || n <= 0 || n > 4032) return PILCO_E_SHAPE;
    HIPCHK(hipStreamSynchronize(ctx->st));
    HIPCHK(hipMemcpy(out, ctx->dbg + 64, n * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return PILCO_OK;
}

int pilco_debug_timestamps(pilco_ctx* ctx, unsigned long long* out32) {  // 64 slots
    if (!ctx) return PILCO_E_SHAPE;
    HIPCHK(hipSetDevice(ctx->device));
    if (!ctx->dbg) {
        HIPCHK(hipMalloc(&ctx->dbg, 4096 * sizeof(unsigned long long)));
        HIPCHK(hipMemset(ctx->dbg, 0, 32 * sizeof(unsigned long long)));
        HIPCHK(hipMemset(ctx->dbg + 5, 0xff, sizeof(unsigned long long)));
        HIPCHK(hipMemset(ctx->dbg + 22, 0xff, sizeof(unsigned long long)));
        for (Slot& s : ctx->slot) s.wk_valid = false;
    }
    if (out32) {
        HIPCHK(hipStreamSynchronize(ctx->st));
        HIPCHK(hipMemcpy(out32, ctx->dbg, 64 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        HIPCHK(hipMemset(ctx->dbg, 0, 32 * sizeof(unsigned long long)));
        HIPCHK(hipMemset(ctx->dbg + 5, 0xff, sizeof(unsigned long long)));
        HIPCHK(hipMemset(ctx->dbg + 22, 0xff, sizeof(unsigned long long)));
    }
    return PILCO_OK;
}

// ---- pure host functions of the ownership / gather-buffer layout (no GPU needed)
int pilco_shard_plan(int E, int D, int nranks, int rank, int* out5) {
    if (E <= 0 || D <= 0 || nranks <= 0 || rank < 0 || rank >= nranks || !out5) return PILCO_E_SHAPE;
    const int P = E * (E + 1) / 2;
    const int PLcap = (P + nranks - 1) / nranks, ELcap = (E + nranks - 1) / nranks;
    out5[0] = (rank < P) ? (P - rank + nranks - 1) / nranks : 0;  // local pairs
    out5[1] = (rank < E) ? (E - rank + nranks - 1) / nranks : 0;  // owned outputs
    out5[2] = PLcap + ELcap * (1 + D);                            // SEG: doubles per rank in the gather buffer
    out5[3] = PLcap;                                              // OUTOFF: offset of the output records
    out5[4] = P;
    return PILCO_OK;
}
// index into the gathered buffer [nranks][SEG] of the value of pair (a,b), a >= b
int pilco_shard_pair_slot(int E, int D, int nranks, int a, int b) {
    if (a < b) { const int t = a; a = b; b = t; }
    if (b < 0 || a >= E) return -1;
    int plan[5];
    if (pilco_shard_plan(E, D, nranks, 0, plan) != PILCO_OK) return -1;
    const int kk = (a == b) ? a : E + a * (a - 1) / 2 + b;
    return (kk % nranks) * plan[2] + kk / nranks;
}
// index of M_a in the gathered buffer (V_a[0..D) follows)
int pilco_shard_output_slot(int E, int D, int nranks, int a) {
    if (a < 0 || a >= E) return -1;
    int plan[5];
    if (pilco_shard_plan(E, D, nranks, 0, plan) != PILCO_OK) return -1;
    return (a % nranks) * plan[2] + plan[3] + (a / nranks) * (1 + D);
}

// ---- host-mediated exchange: the caller moves the segments between the ranks
int pilco_gp_shard_pack(pilco_ctx* ctx, int slot, const double* m, const double* s_in, double* segment) {
    if (int r = check_slot(ctx, slot)) return r;
    Slot& s = ctx->slot[slot];
    if (!s.factor_valid) return fail(ctx, PILCO_E_STATE, "shard_pack: no current factorisation");
    if (!m || !s_in || !segment) return fail(ctx, PILCO_E_SHAPE, "shard_pack: null pointer");
    HIPCHK(hipSetDevice(ctx->device));
    if (int r = build_work(ctx, s)) return r;
    const int D = s.D, E = s.E;
    HIPCHK(hipMemcpyAsync(s.wk.in_m, m, sizeof(double) * D, hipMemcpyHostToDevice, ctx->st));
    HIPCHK(hipMemcpyAsync(s.wk.in_s, s_in, sizeof(double) * D * D, hipMemcpyHostToDevice, ctx->st));
    const MMModel md = model_of(s);
    if (s.wk.PL > 0) {
        launch_mm_prep(ctx->st, md, s.wk);
        launch_mm_pair(ctx->st, md, s.wk, ctx->variant);
    }
    GlueArgs g{};
    g.E = E; g.D = D; g.U = 0;
    g.wk = s.wk;
    g.var = s.var.p;
    g.flags = GF_PACK;
    launch_glue(ctx->st, g);
    HIPCHK(hipMemcpyAsync(segment, s.wk.gath + (size_t)ctx->rank * s.wk.SEG, sizeof(double) * s.wk.SEG, hipMemcpyDeviceToHost, ctx->st));
    HIPCHK(hipStreamSynchronize(ctx->st));
    HIPCHK(hipGetLastError());
    return PILCO_OK;
}

int pilco_gp_shard_finish(pilco_ctx* ctx, int slot, const double* gathered, double* M, double* S, double* V) {
    if (int r = check_slot(ctx, slot)) return r;
    Slot& s = ctx->slot[slot];
    if (!s.wk_valid) return fail(ctx, PILCO_E_STATE, "shard_finish before shard_pack");
    if (!gathered || !M || !S || !V) return fail(ctx, PILCO_E_SHAPE, "shard_finish: null pointer");
    HIPCHK(hipSetDevice(ctx->device));
    const int D = s.D, E = s.E;
    HIPCHK(hipMemcpyAsync(s.wk.gath, gathered, sizeof(double) * (size_t)ctx->nranks * s.wk.SEG, hipMemcpyHostToDevice, ctx->st));
    GlueArgs g{};
    g.E = E; g.D = D; g.U = 0;
    g.wk = s.wk;
    g.var = s.var.p;
    g.flags = GF_ASSEMBLE;
    launch_glue(ctx->st, g);
    HIPCHK(hipMemcpyAsync(M, s.wk.out_M, sizeof(double) * E, hipMemcpyDeviceToHost, ctx->st));
    HIPCHK(hipMemcpyAsync(S, s.wk.out_S, sizeof(double) * E * E, hipMemcpyDeviceToHost, ctx->st));
    HIPCHK(hipMemcpyAsync(V, s.wk.out_V, sizeof(double) * D * E, hipMemcpyDeviceToHost, ctx->st));
    HIPCHK(hipStreamSynchronize(ctx->st));
    HIPCHK(hipGetLastError());
    return PILCO_OK;
}

int pilco_comm_rank(const pilco_ctx* ctx) { return ctx ? ctx->rank : -1; }
int pilco_comm_size(const pilco_ctx* ctx) { return ctx ? ctx->nranks : -1; }

// Vector-Jacobian product of one moment-matching step (the reverse of pilco_gp_predict):
// given cotangents Mbar (1,E), Sbar (E,E), Vbar (D,E) returns mbar (1,D) and the symmetric sbar (D,D).
// Entirely on the device (k_mm_bwd_pair / _post / _fin; the mean part rides in extra workgroups of _post / _fin); the
// host only sums the E + P contribution records.  Single rank, exact or sparse model, D <= 14.
int pilco_gp_predict_vjp(pilco_ctx* ctx, int slot, const double* m, const double* s_in, const double* Mbar,
                         const double* Sbar, const double* Vbar, double* mbar, double* sbar) {
    if (int r = check_slot(ctx, slot)) return r;
    Slot& s = ctx->slot[slot];
    if (!s.factor_valid) return fail(ctx, PILCO_E_STATE, "predict_vjp: no current factorisation");
    if (!m || !s_in || !Mbar || !Sbar || !Vbar || !mbar || !sbar) return fail(ctx, PILCO_E_SHAPE, "predict_vjp: null pointer");
    if (ctx->nranks != 1) return fail(ctx, PILCO_E_STATE, "predict_vjp: single rank only");
    const int D = s.D, E = s.E, npad = s.npad;
    if (D + 2 > 16) return fail(ctx, PILCO_E_SHAPE, "predict_vjp: D <= 14 in this build");
    HIPCHK(hipSetDevice(ctx->device));
    if (int r = build_work(ctx, s)) return r;
    const int P = s.wk.PL;
    // ---- device: operands (prep), reverse pair sweep, mean part, per-pair / per-output contributions
    const int rec = D + D * D, nb = E + E * E + D * E;
    int njs, nrb;
    mm_bwd_geometry(npad, P, &njs, &nrb);
    ENSURE(s.bwd_mom, (size_t)P * njs * 16 * npad);
    ENSURE(s.bwd_cp, (size_t)std::max(1, P - E) * nrb * npad);
    ENSURE(s.bwd_part, (size_t)(P + E) * mm_bwd_rc(npad) * (1 + rec + D));   // pair partials, then mean partials
    ENSURE(s.bwd_out, (size_t)(E + P) * rec + (size_t)(E + P) * (D * D + D + 2));   // contributions | head records
    const size_t n_in = (size_t)D + D * D + nb, n_out = (size_t)(E + P) * rec;
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
    launch_mm_bwd(ctx->st, md, s.wk, s.bwd_mom.p, s.bwd_cp.p, s.bwd_part.p, bars, s.bwd_out.p + (size_t)(E + P) * rec, s.bwd_out.p);
    HIPCHK(hipMemcpyAsync(ctx->pin + n_in, s.bwd_out.p, sizeof(double) * n_out, hipMemcpyDeviceToHost, ctx->st));
    HIPCHK(hipStreamSynchronize(ctx->st));
    HIPCHK(hipGetLastError());
    // ---- host: sum the E + P records in a fixed order, symmetrise
    std::vector<double> acc(rec, 0.0);
    for (int k = 0; k < E + P; ++k)
        for (int e = 0; e < rec; ++e) acc[e] += po[(size_t)k * rec + e];
    for (int e = 0; e < rec; ++e)
        if (!std::isfinite(acc[e])) return fail(ctx, PILCO_E_NOT_PD, "predict_vjp: singular s + Lambda^2 or I + Lambda s");
    for (int d = 0; d < D; ++d) mbar[d] = acc[d];
    for (int r = 0; r < D; ++r)
        for (int c = 0; c < D; ++c) sbar[(size_t)r * D + c] = 0.5 * (acc[D + (size_t)r * D + c] + acc[D + (size_t)c * D + r]);
    return PILCO_OK;
}

// pilco_rollout that also records, for every step t < H, the joint Gaussian (m, s, s1) handed to
// the dynamics GP and its outputs (M, S, V): tape [H][D + D*D + E*D + E + E*E + D*E].  The reverse
// sweep of the policy gradient replays these records (pilco_amd/adjoint.py).
int pilco_rollout_tape(pilco_ctx* ctx, const pilco_policy* policy, const pilco_reward_term* rewards, int n_rewards,
                       const double* m0, const double* S0, int H, double* mH, double* SH, double* reward, double* traj,
                       double* tape) {
    if (!ctx) return PILCO_E_SHAPE;
    if (!m0 || !S0 || !mH || !SH || !reward || !tape || H < 0) return fail(ctx, PILCO_E_SHAPE, "rollout_tape: bad arguments");
    HIPCHK(hipSetDevice(ctx->device));
    RolloutPlan plan;
    if (int r = setup_rollout(ctx, policy, rewards, n_rewards, H, traj != nullptr, plan)) return r;
    const int E = plan.E, D = plan.D;
    const size_t TS = (size_t)D + D * D + (size_t)E * D + E + (size_t)E * E + (size_t)D * E;
    ENSURE(ctx->tape, std::max<size_t>(1, (size_t)H * TS));
    plan.g.tape = ctx->tape.p;
    HIPCHK(hipMemcpyAsync(plan.st[0], m0, sizeof(double) * E, hipMemcpyHostToDevice, ctx->st));
    HIPCHK(hipMemcpyAsync(plan.st[0] + E, S0, sizeof(double) * E * E, hipMemcpyHostToDevice, ctx->st));
    if (int r = enqueue_rollout(ctx, plan, H, nullptr)) return r;
    HIPCHK(hipMemcpyAsync(mH, plan.st[H & 1], sizeof(double) * E, hipMemcpyDeviceToHost, ctx->st));
    HIPCHK(hipMemcpyAsync(SH, plan.st[H & 1] + E, sizeof(double) * E * E, hipMemcpyDeviceToHost, ctx->st));
    HIPCHK(hipMemcpyAsync(reward, plan.g.reward, sizeof(double), hipMemcpyDeviceToHost, ctx->st));
    if (traj) HIPCHK(hipMemcpyAsync(traj, ctx->traj.p, sizeof(double) * (size_t)(H + 1) * (E + E * E), hipMemcpyDeviceToHost, ctx->st));
    if (H > 0) HIPCHK(hipMemcpyAsync(tape, ctx->tape.p, sizeof(double) * (size_t)H * TS, hipMemcpyDeviceToHost, ctx->st));
    HIPCHK(hipStreamSynchronize(ctx->st));
    HIPCHK(hipGetLastError());
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

}  // namespace

extern "C" {

// Value and gradient of the rollout reward w.r.t. a LinearController's (W, b): what TensorFlow's reverse mode through
// the tf.while_loop gives the reference (pilco/models/pilco.py:85-90,126-135).  Forward rollout with a tape on the
// device, then the reverse sweep: the O(N^2) adjoint of every moment-matching step on the device
// (pilco_gp_predict_vjp), the O(D^3) links (propagate pilco.py:147-149, joint Gaussian :141-144, controller + squash
// controllers.py:13-58, rewards rewards.py:19-81) here on the host in C++.  dW (U,E), db (U).
int pilco_rollout_grad(pilco_ctx* ctx, const pilco_policy* policy, const pilco_reward_term* rewards, int n_rewards,
                       const double* m0, const double* S0, int H, double* reward, double* dW, double* db) {
    if (!ctx) return PILCO_E_SHAPE;
    if (!policy || !m0 || !S0 || !reward || !dW || !db || H < 0) return fail(ctx, PILCO_E_SHAPE, "rollout_grad: bad arguments");
    if (policy->kind != PILCO_POLICY_LINEAR || !policy->squash || policy->control_dim <= 0)
        return fail(ctx, PILCO_E_SHAPE, "rollout_grad: squashed LinearController only (other policies: pilco_rollout_tape + pilco_gp_predict_vjp)");
    for (int k = 0; k < n_rewards; ++k)
        if (rewards[k].kind != PILCO_REWARD_EXPONENTIAL && rewards[k].kind != PILCO_REWARD_LINEAR)
            return fail(ctx, PILCO_E_SHAPE, "rollout_grad: unknown reward term");
    const int E = policy->state_dim, U = policy->control_dim, D = E + U;
    const size_t TS = (size_t)D + D * D + (size_t)E * D + E + (size_t)E * E + (size_t)D * E;
    vec mH(E), SH((size_t)E * E), traj((size_t)(H + 1) * (E + E * E)), tape(std::max<size_t>(1, (size_t)H * TS));
    if (int r = pilco_rollout_tape(ctx, policy, rewards, n_rewards, m0, S0, H, mH.data(), SH.data(), reward, traj.data(), tape.data()))
        return r;
    const double* W = policy->W;
    const double* b = policy->b;
    vec e(U);
    for (int u = 0; u < U; ++u) e[u] = policy->max_action[u];
    vec mbar(E, 0.0), sbar((size_t)E * E, 0.0), Wbar((size_t)U * E, 0.0), bbar(U, 0.0);
    vec G((size_t)E * E), Vb((size_t)D * E), s1bar((size_t)E * D), mjb(D), sjb((size_t)D * D), mxb(E), sxb((size_t)E * E);
    vec Bb((size_t)E * U), sub((size_t)U * U), mu0(U), su0((size_t)U * U), WS((size_t)U * E), cb((size_t)E * U), Cdbar(U);
    vec mu0b, su0b, rm(E), rS((size_t)E * E), T1((size_t)U * E), T2((size_t)U * E);
    Squash sq;
    for (int t = H - 1; t >= 0; --t) {
        const double* m_x = &traj[(size_t)t * (E + E * E)];
        const double* s_x = m_x + E;
        const double* rec = &tape[(size_t)t * TS];
        const double* m_j = rec;
        const double* s_j = rec + D;
        const double* s1 = rec + D + D * D;                           // (E, D)
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
        if (int r = pilco_gp_predict_vjp(ctx, PILCO_SLOT_DYNAMICS, m_j, s_j, mbar.data(), sbar.data(), Vb.data(), mjb.data(), sjb.data()))
            return r;
        // joint Gaussian (pilco.py:141-144)
        for (int i = 0; i < E; ++i) mxb[i] += mjb[i];
        for (int i = 0; i < E; ++i)
            for (int j = 0; j < E; ++j) sxb[(size_t)i * E + j] += sjb[(size_t)i * D + j] + s1bar[(size_t)i * D + j];
        for (int i = 0; i < E; ++i)
            for (int u = 0; u < U; ++u)
                Bb[(size_t)i * U + u] = sjb[(size_t)i * D + E + u] + sjb[(size_t)(E + u) * D + i] + s1bar[(size_t)i * D + E + u];
        for (int u = 0; u < U; ++u)
            for (int v = 0; v < U; ++v) sub[(size_t)u * U + v] = sjb[(size_t)(E + u) * D + E + v];
        // controller (controllers.py:46-58): mu0 = W m + b, su0 = W s W^T, c = W^T diag(Cd)
        for (int u = 0; u < U; ++u) {
            double acc = b[u];
            for (int i = 0; i < E; ++i) acc += W[(size_t)u * E + i] * m_x[i];
            mu0[u] = acc;
            for (int j = 0; j < E; ++j) {
                double a2 = 0.0;
                for (int i = 0; i < E; ++i) a2 += W[(size_t)u * E + i] * s_x[(size_t)i * E + j];
                WS[(size_t)u * E + j] = a2;                                  // W s
            }
        }
        for (int u = 0; u < U; ++u)
            for (int v = 0; v < U; ++v) {
                double acc = 0.0;
                for (int j = 0; j < E; ++j) acc += WS[(size_t)u * E + j] * W[(size_t)v * E + j];
                su0[(size_t)u * U + v] = acc;
            }
        sq.fwd(mu0, su0, e);
        for (int i = 0; i < E; ++i)
            for (int j = 0; j < E; ++j) {
                double acc = 0.0;
                for (int u = 0; u < U; ++u) acc += Bb[(size_t)i * U + u] * W[(size_t)u * E + j] * sq.Cd[u];   // Bb c^T, c = W^T diag(Cd)
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
            for (int i = 0; i < E; ++i) acc += W[(size_t)u * E + i] * cb[(size_t)i * U + u];
            Cdbar[u] = acc;
        }
        sq.vjp(&mjb[E], sub.data(), Cdbar.data(), mu0b, su0b);
        // Wbar += diag(Cd) cb^T + mu0b m_x^T + su0b W s_x^T + su0b^T W s_x
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
                double acc = sq.Cd[u] * cb[(size_t)j * U + u] + mu0b[u] * m_x[j];
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
        // reward of the pre-propagation state (pilco.py:133)
        std::fill(rm.begin(), rm.end(), 0.0);
        std::fill(rS.begin(), rS.end(), 0.0);
        if (!reward_grad(rewards, n_rewards, E, m_x, s_x, rm, rS)) return fail(ctx, PILCO_E_NOT_PD, "rollout_grad: singular I + S W in the reward");
        for (int i = 0; i < E; ++i) mxb[i] += rm[i];
        for (int i = 0; i < E * E; ++i) sxb[i] += rS[i];
        mbar = mxb;
        for (int i = 0; i < E; ++i)
            for (int j = 0; j < E; ++j) sbar[(size_t)i * E + j] = 0.5 * (sxb[(size_t)i * E + j] + sxb[(size_t)j * E + i]);
    }
    memcpy(dW, Wbar.data(), sizeof(double) * U * E);
    memcpy(db, bbar.data(), sizeof(double) * U);
    return PILCO_OK;
}

}  // extern "C"

// ---- sparse GP: FITC factorisation of pilco/models/smgpr.py:24-45 on the device.
// With L = chol(Kmm + 1e-6 I), V = L^{-1} Kmn / G, Am = chol(V V^T + sn2 I), At = L Am:
//   beta = L^{-T} (Am Am^T)^{-1} (V/G) y,   iK = Kmm^{-1} - sn2 At^{-T} At^{-1}.
// Explicit triangular inverses turn every solve into an MFMA GEMM / mat-vec.
int pilco_factorize_fitc(pilco_ctx* ctx, void* slot_ptr) {
    Slot& s = *static_cast<Slot*>(slot_ptr);
    const int E = s.E, Mp = s.npad, Np = s.Npad, nblk = Mp / NB;
    const size_t mm = (size_t)Mp * Mp, mn = (size_t)Mp * Np;
    ENSURE(s.K, E * mm);        // Kmm -> L
    ENSURE(s.Linv, E * mm);
    ENSURE(s.iK, E * mm);
    ENSURE(s.invD, (size_t)E * nblk * NB * NB);
    ENSURE(s.Kmn, E * mn);      // Kmn -> V
    ENSURE(s.Am, E * mm);
    ENSURE(s.AmInv, E * mm);
    ENSURE(s.AmD, (size_t)E * nblk * NB * NB);
    ENSURE(s.iAt, E * mm);
    ENSURE(s.G, (size_t)E * Np);
    ENSURE(s.beta, (size_t)E * Mp);
    ENSURE(s.Tscr, (size_t)E * Mp * Mp);
    ENSURE(s.vec, (size_t)E * std::max(Mp, Np) * 2);
    hipStream_t st = ctx->st;
    HIPCHK(hipMemsetAsync(ctx->d_info, 0, sizeof(int) * 64, st));
    // smgpr.py:27-28: Kmm = K(Z) + 1e-6 I, Kmn = K(Z, X)
    launch_gram(st, s.Zt.p, Mp, s.M, s.Zt.p, Mp, s.M, s.D, s.ls.p, s.var.p, E, s.K.p, Mp, Mp, 2, nullptr, 1e-6);
    launch_gram(st, s.Zt.p, Mp, s.M, s.Xt.p, Np, s.N, s.D, s.ls.p, s.var.p, E, s.Kmn.p, Mp, Np, 0, nullptr, 0.0);
    launch_potrf(st, s.K.p, Mp, E, s.invD.p, ctx->d_info);                      // smgpr.py:29
    launch_trtri(st, s.K.p, Mp, E, s.invD.p, s.Linv.p, s.Tscr.p, (long)Mp * Mp);
    GemmDesc g{};
    // V = L^{-1} Kmn  (smgpr.py:30) -- out of place into vec? Kmn is (Mp, Np): use iAt-sized scratch is too small, so
    // write V into a second Kmn-sized buffer: reuse s.Am? no (Mp x Mp).  V goes to s.Kmn2 = s.vec is too small -> allocate.
    DevBuf& Vb = s.V2;
    ENSURE(Vb, E * mn);
    g = GemmDesc{};
    g.A = s.Linv.p; g.lda = Mp; g.sA = (long)mm;
    g.B = s.Kmn.p; g.ldb = Np; g.sB = (long)mn;
    g.C = Vb.p; g.ldc = Np; g.sC = (long)mn;
    g.M = Mp; g.N = Np; g.K = Mp; g.alpha = 1.0; g.beta = 0.0; g.k_mode = 3;
    launch_gemm(st, g, false, false, E);
    launch_fitc_scale(st, Vb.p, Mp, Np, E, s.var.p, s.noise.p, s.G.p);          // smgpr.py:31-33
    // Am = chol(V V^T + sn2 I)  (smgpr.py:34-35)
    g = GemmDesc{};
    g.A = Vb.p; g.lda = Np; g.sA = (long)mn;
    g.B = Vb.p; g.ldb = Np; g.sB = (long)mn;
    g.C = s.Am.p; g.ldc = Mp; g.sC = (long)mm;
    g.M = Mp; g.N = Mp; g.K = Np; g.alpha = 1.0; g.beta = 0.0;
    launch_gemm(st, g, false, true, E);
    launch_add_diag(st, s.Am.p, Mp, E, s.noise.p);
    launch_potrf(st, s.Am.p, Mp, E, s.AmD.p, ctx->d_info + 32);
    launch_trtri(st, s.Am.p, Mp, E, s.AmD.p, s.AmInv.p, s.Tscr.p, (long)Mp * Mp);
    // iAt = (L Am)^{-1} = Am^{-1} L^{-1}  (smgpr.py:36-37)
    g = GemmDesc{};
    g.A = s.AmInv.p; g.lda = Mp; g.sA = (long)mm;
    g.B = s.Linv.p; g.ldb = Mp; g.sB = (long)mm;
    g.C = s.iAt.p; g.ldc = Mp; g.sC = (long)mm;
    g.M = Mp; g.N = Mp; g.K = Mp; g.alpha = 1.0; g.beta = 0.0; g.k_mode = 4;
    launch_gemm(st, g, false, false, E);
    // beta = L^{-T} Am^{-T} Am^{-1} (V/G) y  (smgpr.py:38-42)
    double* r0 = s.vec.p;
    double* r1 = s.vec.p + (size_t)E * Mp;
    launch_fitc_rhs(st, Vb.p, s.G.p, s.Yt.p, Mp, Np, E, r0);
    launch_matvec(st, s.AmInv.p, Mp, E, r0, r1, false);
    launch_matvec(st, s.AmInv.p, Mp, E, r1, r0, true);
    launch_matvec(st, s.Linv.p, Mp, E, r0, s.beta.p, true);
    // iK = Kmm^{-1} - sn2 iAt^T iAt  (smgpr.py:43-44)
    g = GemmDesc{};
    g.A = s.Linv.p; g.lda = Mp; g.sA = (long)mm;
    g.B = s.Linv.p; g.ldb = Mp; g.sB = (long)mm;
    g.C = s.iK.p; g.ldc = Mp; g.sC = (long)mm;
    g.M = Mp; g.N = Mp; g.K = Mp; g.alpha = 1.0; g.beta = 0.0; g.k_mode = 1;
    launch_gemm(st, g, true, false, E);
    g.A = s.iAt.p;
    g.B = s.iAt.p;
    g.alpha = -1.0; g.alpha_vec = s.noise.p; g.beta = 1.0; g.k_mode = 1;
    launch_gemm(st, g, true, false, E);
    launch_clear_padding(st, s.iK.p, Mp, s.M, E);
    int info[64];
    HIPCHK(hipMemcpyAsync(info, ctx->d_info, sizeof(int) * 64, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    for (int a = 0; a < std::min(E, 32); ++a)
        if (info[a] != 0 || info[32 + a] != 0) {
            ctx->not_pd = a;
            return fail(ctx, PILCO_E_NOT_PD, "FITC Cholesky failed for output " + std::to_string(a));
        }
    s.n = s.M;
    s.iK_null = false;
    return PILCO_OK;
}
