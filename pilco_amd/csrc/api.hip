// C ABI of libpilco_hip.so (see include/pilco_hip.h): context, GP slots, factorisation drivers, the single
// moment-matching step and introspection.  Rollouts: rollout.hip; reverse mode: grad.hip; sharding: shard.hip.
#include "ctx.h"

int fail(pilco_ctx* c, int code, const std::string& msg) {
    if (c) c->err = msg;
    return code;
}

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
    wk.vsep = mm_vsep(D) ? 1 : 0;
    mm_prep_chunks(npad, std::max(wk.PL, 1), wk.EL, &wk.NCH, &wk.NCHM);
    if (W > 1) {   // the chunks of an output's mean sums decide how they are added up: taken from the WHOLE model's counts, so
        int nch_g;  // that M and V come out with the same bits on any rank count (the operand chunks NCH carry no sums)
        mm_prep_chunks(npad, P, E, &nch_g, &wk.NCHM);
        wk.NCHM = std::min(wk.NCHM, wk.NCH);
    }
    wk.NT = mm_pair_nt(npad, ctx->variant, std::max(wk.PL, 1));
    wk.OUTOFF = PLcap;
    wk.SEG = PLcap + ELcap * (1 + D);
    wk.rank = rank;
    wk.nranks = W;
#ifdef PILCO_DEV
    wk.abl = getenv("PILCO_ABL") ? atoi(getenv("PILCO_ABL")) : 0;   // experiment switches: developer builds (tools/) only
#else
    wk.abl = 0;
#endif
    const int PLa = std::max(wk.PL, 1);
    // stream-K geometry (variant 0)
    wk.sk_waves = 0;
    wk.sk_maxw = 0;
    wk.sk_pls = 0;
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
        int waves = mm_pair_sk_capacity(wk.KP, wk.vsep != 0);
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
        wk.sk_pls = round_up(wk.PL, 16);
    }
    ENSURE(s.w_in, (size_t)D + D * D + E + E * E + D * E);   // m | s | cotangents (Mbar | Sbar | Vbar) of the reverse pass
    // operands of the exponent GEMM (layout: MMWork::At / Wt): per pair (2 Q z_i | u_i) and v_j; per column output the w rows
    // (models of at most 256 points: a block per local pair too, for the one-launch small step with operands in memory);
    // the ones (valid mask) and the zeros, constants of the model, are written here
    const size_t wt_blocks = npad <= 256 ? (size_t)std::max(E, PLa) : (size_t)E;
    ENSURE(s.w_At, (size_t)PLa * wk.KP * npad);
    ENSURE(s.w_Wt, (wt_blocks * wk.KP + PLa) * npad);
    const size_t n_small = (size_t)PLa + (size_t)E * wk.NCHM * (1 + D);
    ENSURE(s.w_small, 2 * n_small);
    ENSURE(s.w_part, std::max((size_t)PLa * wk.NT * 2, (size_t)std::max(wk.sk_pls, 16) * std::max(wk.sk_maxw, 4)));
    ENSURE(s.w_fpart, (size_t)2 * PLa * wk.NCH * 4 * 2);   // (up to four column splits per row chunk)
    wk.fuse_pair = 0;
    wk.wt_R = wt_rows_per_pair_of(E, D, wk.PL);
    wk.wt_deal = (wk.wt_R <= mm_prep_dt(D) && npad / wk.NCH <= 256) ? 1 : 0;
    wk.NCS = 1;
    wk.share_cu = 0;
    ENSURE(s.w_gath, (size_t)W * wk.SEG);
    ENSURE(s.w_out, (size_t)E + E * E + D * E);
    wk.in_m = s.w_in.p;
    wk.in_s = s.w_in.p + D;
    wk.At = s.w_At.p;
    wk.Wt = s.w_Wt.p;
    wk.vcol = wk.Wt + wt_blocks * wk.KP * npad;
    // the constant rows of every block: zeros everywhere, then the ones (the valid mask) in row D + 1 of A (not vsep) and row D
    // of B -- one small launch (rounds 4-5: an upload, a stream synchronisation and PL + E device-to-device copies on every
    // workspace rebuild, i.e. on every policy-slot update of an optimisation)
    HIPCHK(hipMemsetAsync(wk.At, 0, sizeof(double) * (size_t)PLa * wk.KP * npad, ctx->st));
    HIPCHK(hipMemsetAsync(wk.Wt, 0, sizeof(double) * (wt_blocks * wk.KP + PLa) * npad, ctx->st));
    launch_const_rows(ctx->st, wk.vsep ? nullptr : wk.At + (size_t)(D + 1) * npad, PLa, wk.Wt + (size_t)D * npad, (int)wt_blocks, (long)wk.KP * npad, npad,
                      s.n);
    wk.pair_isdet = s.w_small.p;
    wk.mean_part = wk.pair_isdet + PLa;
    s.alt_isdet = s.w_small.p + n_small;
    s.alt_mean = s.alt_isdet + PLa;
    wk.pair_part = s.w_part.p;
    wk.sk_part = s.w_part.p;
    wk.gath = s.w_gath.p;
    wk.out_M = s.w_out.p;
    wk.out_S = wk.out_M + E;
    wk.out_V = wk.out_S + E * E;
    wk.exp_tab = ctx->exp_tab.p;
    wk.dbg = ctx->dbg;
    HIPCHK(hipMemsetAsync(s.w_gath.p, 0, sizeof(double) * W * wk.SEG, ctx->st));
    if (wk.sk_waves > 0) HIPCHK(hipMemsetAsync(s.w_part.p, 0, sizeof(double) * wk.sk_pls * wk.sk_maxw, ctx->st));   // slots no wave writes
    s.wk_valid = true;
    s.wk_variant = ctx->variant;
    return PILCO_OK;
}

MMModel model_of(const Slot& s) {
    MMModel md{};
    md.Pt = s.M > 0 ? s.Zt.p : s.Xt.p;
    md.ls = s.ls.p;
    md.var = s.var.p;
    md.lvar = s.var.p + s.E;
    md.beta = s.beta.p;
    md.iK = (s.iK_null || s.ignore_iK) ? nullptr : s.iK.p;
    md.bW = s.shW > 0 ? s.shW : 1;
    md.bEL = s.shW > 1 ? s.shEL : s.E;
    md.n = s.n;
    md.npad = s.npad;
    md.D = s.D;
    md.E = s.E;
    return md;
}

// in-process exchange (pilco_rollout_group): every member waits for all segments to be complete, copies the peers'
// segments into its own gather buffer (device-to-device / peer-to-peer), and nobody starts the next step before all
// copies are done
static int group_exchange(pilco_ctx* ctx, Slot& s) {
    PeerGroup& grp = *ctx->group;
    const int slot_idx = (int)(&s - &ctx->slot[0]);
    HIPCHK(hipStreamSynchronize(ctx->st));
    if (!grp.arrive_and_wait()) return fail(ctx, PILCO_E_STATE, "group exchange: another member failed");
    const size_t seg = (size_t)s.wk.SEG;
    for (int j = 0; j < (int)grp.ctxs.size(); ++j) {
        if (j == ctx->rank) continue;
        const double* src = grp.ctxs[j]->slot[slot_idx].wk.gath + (size_t)j * seg;
        HIPCHK(hipMemcpyAsync(s.wk.gath + (size_t)j * seg, src, sizeof(double) * seg, hipMemcpyDefault, ctx->st));
    }
    HIPCHK(hipStreamSynchronize(ctx->st));
    if (!grp.arrive_and_wait()) return fail(ctx, PILCO_E_STATE, "group exchange: another member failed");
    return PILCO_OK;
}

int all_gather_segments(pilco_ctx* ctx, Slot& s) {
    if (ctx->nranks == 1 && !ctx->comm) return PILCO_OK;
    if (ctx->group) return group_exchange(ctx, s);
    if (!ctx->comm) return fail(ctx, PILCO_E_STATE, "sharded context without communicator: use pilco_gp_shard_pack / pilco_gp_shard_finish");
    double* base = s.wk.gath;
    ncclResult_t r = ncclAllGather(base + (size_t)ctx->rank * s.wk.SEG, base, s.wk.SEG, ncclDouble, ctx->comm, ctx->st);
    if (r != ncclSuccess) return fail(ctx, PILCO_E_RCCL, std::string("ncclAllGather: ") + ncclGetErrorString(r));
    return PILCO_OK;
}

// ---- which outputs this rank factorises, and their hyper-parameters / targets compacted for the batched kernels
int prepare_own(pilco_ctx* ctx, Slot& s, OwnView& o) {
    o.W = ctx->nranks;
    o.rank = ctx->rank;
    if (&s == &ctx->slot[PILCO_SLOT_POLICY]) {   // the RbfController's GP is never sharded: every rank factorises all of it
        o.W = 1;                                 // (bf <= a few hundred points) and evaluates it inside its own link
        o.rank = 0;
    }
    const int E = s.E, D = s.D, Np = s.Npad;
    o.ELcap = (E + o.W - 1) / o.W;
    o.EL = (o.rank < E) ? (E - o.rank + o.W - 1) / o.W : 0;
    if (o.W == 1) {
        o.ls = s.ls.p; o.var = s.var.p; o.noise = s.noise.p; o.Yt = s.Yt.p;
        return PILCO_OK;
    }
    const int ELa = std::max(o.EL, 1);
    ENSURE(s.own, (size_t)ELa * (D + 2) + (size_t)ELa * Np);
    double* pls = s.own.p;
    double* pvar = pls + (size_t)ELa * D;
    double* pnz = pvar + ELa;
    double* pY = pnz + ELa;
    if (o.EL > 0) {   // rows rank, rank + W, ... of the full arrays (strided device-to-device copies)
        hipStream_t st = ctx->st;
        HIPCHK(hipMemcpy2DAsync(pls, sizeof(double) * D, s.ls.p + (size_t)o.rank * D, sizeof(double) * D * o.W, sizeof(double) * D, o.EL, hipMemcpyDeviceToDevice, st));
        HIPCHK(hipMemcpy2DAsync(pvar, sizeof(double), s.var.p + o.rank, sizeof(double) * o.W, sizeof(double), o.EL, hipMemcpyDeviceToDevice, st));
        HIPCHK(hipMemcpy2DAsync(pnz, sizeof(double), s.noise.p + o.rank, sizeof(double) * o.W, sizeof(double), o.EL, hipMemcpyDeviceToDevice, st));
        HIPCHK(hipMemcpy2DAsync(pY, sizeof(double) * Np, s.Yt.p + (size_t)o.rank * Np, sizeof(double) * Np * o.W, sizeof(double) * Np, o.EL, hipMemcpyDeviceToDevice, st));
    }
    o.ls = pls; o.var = pvar; o.noise = pnz; o.Yt = pY;
    return PILCO_OK;
}

// beta rows of the other ranks: one ncclAllGather per MODEL (not per step).  Without a communicator the rows arrive
// later through pilco_group_sync_model (contexts of one process) and the factorisation stays incomplete until then.
static int gather_beta(pilco_ctx* ctx, Slot& s, const OwnView& o, int npad) {
    s.shW = o.W; s.shEL = o.ELcap; s.shOwn = o.EL; s.shRank = o.rank;
    s.beta_complete = true;
    if (o.W == 1) return PILCO_OK;
    if (ctx->comm) {
        const size_t blk = (size_t)o.ELcap * npad;
        ncclResult_t r = ncclAllGather(s.beta.p + (size_t)o.rank * blk, s.beta.p, blk, ncclDouble, ctx->comm, ctx->st);
        if (r != ncclSuccess) return fail(ctx, PILCO_E_RCCL, std::string("ncclAllGather(beta): ") + ncclGetErrorString(r));
        HIPCHK(hipStreamSynchronize(ctx->st));
        return PILCO_OK;
    }
    s.beta_complete = false;
    return PILCO_OK;
}

// A failed Cholesky on one rank must fail on EVERY rank of a communicator, before any of them enters the next collective
// (the all-gather of beta / of the objective): otherwise the peers hang in it, or pair it with the failing rank's retry.
// bad = the failing GLOBAL output index of this rank (-1: none); returns the largest index any rank reports (-1: none).
int agree_not_pd(pilco_ctx* ctx, int W, int bad, int* agreed) {
    *agreed = bad;
    if (W <= 1 || !ctx->comm) return PILCO_OK;
    int* d = ctx->d_info + 64;   // (words 0..63 belong to the factorisations' kernels)
    const int mine = bad + 1;    // 0 = fine
    HIPCHK(hipMemcpyAsync(d, &mine, sizeof(int), hipMemcpyHostToDevice, ctx->st));
    ncclResult_t r = ncclAllReduce(d, d + 1, 1, ncclInt, ncclMax, ctx->comm, ctx->st);
    if (r != ncclSuccess) return fail(ctx, PILCO_E_RCCL, std::string("ncclAllReduce(not positive definite): ") + ncclGetErrorString(r));
    int all = 0;
    HIPCHK(hipMemcpyAsync(&all, d + 1, sizeof(int), hipMemcpyDeviceToHost, ctx->st));
    HIPCHK(hipStreamSynchronize(ctx->st));
    *agreed = all - 1;
    return PILCO_OK;
}

// ---- exact GP: Gram -> Cholesky -> L^{-1} -> iK = L^{-T} L^{-1}, beta = L^{-T} (L^{-1} y), for the outputs this rank owns
int factorize_exact(pilco_ctx* ctx, Slot& s) {
    OwnView o{};
    if (int r = prepare_own(ctx, s, o)) return r;
    const int EL = o.EL, ELa = std::max(EL, 1), npad = s.Npad;
    const size_t mat = (size_t)npad * npad;
    ENSURE(s.K, ELa * mat);
    ENSURE(s.Linv, ELa * mat);
    ENSURE(s.iK, ELa * mat);
    ENSURE(s.beta, (size_t)o.W * o.ELcap * npad);
    ENSURE(s.vec, (size_t)ELa * npad);
    hipStream_t st = ctx->st;
    double* beta_own = s.beta.p + (size_t)o.rank * o.ELcap * npad;
    // ~50 launches whose sequence depends only on the sizes: replayed as ONE graph launch (mgpr.py:81-89 is paid per
    // evaluation of optimize_models' objective; eager, the launches' host time was 20 % of the whole on a slower host)
    const std::vector<unsigned long long> key = {(unsigned long long)(uintptr_t)s.K.p, (unsigned long long)(uintptr_t)s.Linv.p,
        (unsigned long long)(uintptr_t)s.iK.p, (unsigned long long)(uintptr_t)s.beta.p,
        (unsigned long long)(uintptr_t)s.vec.p, (unsigned long long)(uintptr_t)s.Xt.p, (unsigned long long)(uintptr_t)o.ls,
        (unsigned long long)(uintptr_t)o.var, (unsigned long long)(uintptr_t)o.noise, (unsigned long long)(uintptr_t)o.Yt,
        (unsigned long long)(uintptr_t)ctx->d_info, (unsigned long long)npad, (unsigned long long)s.N, (unsigned long long)s.D,
        (unsigned long long)EL, (unsigned long long)o.W, (unsigned long long)o.rank, (unsigned long long)o.ELcap};
    auto chain = [&]() -> int {
    HIPCHK(hipMemsetAsync(ctx->d_info, 0, sizeof(int) * 64, st));
    if (o.W > 1) HIPCHK(hipMemsetAsync(s.beta.p, 0, sizeof(double) * (size_t)o.W * o.ELcap * npad, st));
    if (EL > 0) {
        launch_gram(st, s.Xt.p, npad, s.N, s.Xt.p, npad, s.N, s.D, o.ls, o.var, EL, s.K.p, npad, npad, 1, o.noise, 0.0);
        launch_potrf(st, s.K.p, npad, EL, s.Linv.p, ctx->d_info, false);   // (nobody below reads L^-1 above its diagonal tiles: not zeroed)
        launch_trtri(st, s.K.p, npad, EL, s.Linv.p, s.iK.p, (long)mat);    // iK is free until the next GEMM
        GemmDesc g{};
        g.A = s.Linv.p; g.lda = npad; g.sA = (long)mat;
        g.B = s.Linv.p; g.ldb = npad; g.sB = (long)mat;
        g.C = s.iK.p; g.ldc = npad; g.sC = (long)mat;
        g.M = npad; g.N = npad; g.K = npad; g.alpha = 1.0; g.beta = 0.0; g.k_mode = 1;
        g.tile_mode = 2;   // iK = Linv^T Linv is symmetric: the tiles on and below the diagonal are computed, their mirror images stored (the same sums, term by term, as when both halves were computed)
        launch_gemm(st, g, true, false, EL);
        launch_clear_padding(st, s.iK.p, npad, s.N, EL);
        launch_matvec(st, s.Linv.p, npad, EL, o.Yt, s.vec.p, false);
        launch_matvec(st, s.Linv.p, npad, EL, s.vec.p, beta_own, true);
    }
    return PILCO_OK;
    };
    if (int r = run_chain_graph(ctx, s.g_fact, key, chain)) return r;
    int info[64];
    HIPCHK(hipMemcpyAsync(info, ctx->d_info, sizeof(int) * std::min(ELa, 64), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    int bad = -1, pivot = 0, agreed = -1;
    for (int al = 0; al < std::min(EL, 64) && bad < 0; ++al)
        if (info[al] != 0) bad = al * o.W + o.rank, pivot = info[al];
    if (int r = agree_not_pd(ctx, o.W, bad, &agreed)) return r;
    if (agreed >= 0) {   // the same output named, the same error raised on every rank
        ctx->not_pd = agreed;
        return fail(ctx, PILCO_E_NOT_PD, "Cholesky failed: K + noise*I of output " + std::to_string(agreed) + " is not positive definite" +
                                             (agreed == bad ? " (pivot " + std::to_string(pivot) + ")" : std::string(" (on another rank)")));
    }
    s.n = s.N;
    s.npad = npad;
    s.iK_null = false;
    return gather_beta(ctx, s, o, npad);
}


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
        hipMalloc(&ctx->d_info, 72 * sizeof(int)) != hipSuccess || hipEventCreate(&ctx->ev0) != hipSuccess ||
        hipEventCreate(&ctx->ev1) != hipSuccess) {
        delete ctx;
        return PILCO_E_HIP;
    }
    {
        const int tn = mm_exp_table_size();
        std::vector<double> tab(tn);
        for (int j = 0; j < tn; ++j) tab[j] = std::exp2((double)j / (double)tn);
        if (ctx->exp_tab.ensure(tn) != hipSuccess ||
            hipMemcpy(ctx->exp_tab.p, tab.data(), sizeof(double) * tn, hipMemcpyHostToDevice) != hipSuccess) {
            delete ctx;
            return PILCO_E_HIP;
        }
    }
    if (const char* eg = getenv("PILCO_NO_GRAPH")) ctx->use_graph = (atoi(eg) == 0);
    if (const char* ei = getenv("PILCO_INLINE_POLICY")) ctx->inline_policy = (atoi(ei) != 0);
    if (const char* ef = getenv("PILCO_FUSED")) ctx->fused = (atoi(ef) != 0);
    if (const char* ef = getenv("PILCO_SMALL_STEP")) ctx->fuse_small = (atoi(ef) != 0);
    if (const char* ef = getenv("PILCO_HOST_CHAIN")) ctx->dev_chain = (atoi(ef) == 0);
    if (const char* env = getenv("PILCO_PAIR_KERNEL")) {   // a choice like pilco_set_pair_kernel's: pilco_shard_set / _comm_init leave it alone
        ctx->variant = (atoi(env) >= 0 && atoi(env) <= 2) ? atoi(env) : 0;
#ifndef PILCO_DEV
        if (ctx->variant == 1) ctx->variant = 0;   // (the plain-VALU cross-check kernel is not in this build)
#endif
        ctx->variant_user = true;
    }
    *out = ctx;
    return PILCO_OK;
}

int pilco_ctx_destroy(pilco_ctx* ctx) {
    if (!ctx) return PILCO_OK;
    for (pilco_ctx* lane : ctx->lanes) (void)pilco_ctx_destroy(lane);   // (their borrowed model buffers are not freed there)
    ctx->lanes.clear();
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->st);
    (void)peer_detach(ctx);
    for (auto& ge : ctx->graph_cache) (void)hipGraphExecDestroy(ge.second);
    for (Slot& sl : ctx->slot)
        for (ChainGraph* cg : {&sl.g_fact, &sl.g_fitc, &sl.g_fitc_nlml}) chain_graph_release(*cg);
    if (ctx->comm) ncclCommDestroy(ctx->comm);
    for (Slot& s : ctx->slot) {
        for (DevBuf* b : {&s.Xt, &s.Yt, &s.Zt, &s.ls, &s.var, &s.noise, &s.K, &s.Linv, &s.iK, &s.beta, &s.Tscr, &s.ksplit_ws,
                          &s.vec, &s.Kmn, &s.V2, &s.bwd_mom, &s.bwd_cp, &s.bwd_part, &s.bwd_out, &s.bwd_cnt, &s.jac_rowmom, &s.jac_cpart, &s.jac_head, &s.jac_part, &s.jac_np, &s.own, &s.Am, &s.AmInv, &s.iAt, &s.G, &s.w_in, &s.w_At, &s.w_Wt, &s.w_small, &s.w_fpart,
                          &s.w_part, &s.w_gath, &s.w_out, &s.ft_P, &s.ft_T3, &s.ft_Z})
            b->release();
    }
    ctx->state.release();
    ctx->params.release();
    ctx->traj.release();
    ctx->tape.release();
    ctx->jrec.release();
    ctx->jgath.release();
    ctx->revloc.release();
    ctx->revseeds.release();
    ctx->revmat.release();
    ctx->selftest.release();
    ctx->exp_tab.release();
    for (hipEvent_t e : ctx->pair_events) (void)hipEventDestroy(e);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    if (ctx->d_info) (void)hipFree(ctx->d_info);
    if (ctx->pin) (void)hipHostFree(ctx->pin);
    if (ctx->pin_io) (void)hipHostFree(ctx->pin_io);
    if (ctx->jpin) (void)hipHostFree(ctx->jpin);
    for (hipEvent_t e : ctx->jwait_ev)
        if (e) (void)hipEventDestroy(e);
    if (ctx->st) (void)hipStreamDestroy(ctx->st);
    delete ctx;
    return PILCO_OK;
}

const char* pilco_last_error(const pilco_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }
int pilco_last_not_pd_output(const pilco_ctx* ctx) { return ctx ? ctx->not_pd : -1; }

int pilco_set_pair_kernel(pilco_ctx* ctx, int variant) {
    if (!ctx || variant < 0 || variant > 2) return PILCO_E_SHAPE;
#ifndef PILCO_DEV
    if (variant == 1) return fail(ctx, PILCO_E_STATE, "set_pair_kernel: the plain-VALU cross-check kernel exists in -DPILCO_DEV builds only");
#endif
    ctx->variant = variant;
    ctx->variant_user = true;
    return PILCO_OK;
}

int pilco_debug_poison(pilco_ctx* ctx, int slot, int which) {
    if (int r = check_slot(ctx, slot)) return r;
    Slot& s = ctx->slot[slot];
    DevBuf* b = which == 0 ? &s.Linv : which == 1 ? &s.iK : which == 2 ? &s.beta : nullptr;
    if (!b) return fail(ctx, PILCO_E_SHAPE, "debug_poison: which = 0 (L^-1), 1 (iK), 2 (beta)");
    HIPCHK(hipSetDevice(ctx->device));
    if (b->p && b->cap) {
        HIPCHK(hipMemset(b->p, 0xff, sizeof(double) * b->cap));   // all-ones words: NaN
        HIPCHK(hipStreamSynchronize(nullptr));                    // (a device memset is ordered on the null stream only; the context's stream is non-blocking)
    }
    return PILCO_OK;
}

int pilco_set_reverse_chain(pilco_ctx* ctx, int on_device) {
    if (!ctx) return PILCO_E_SHAPE;
    ctx->dev_chain = (on_device != 0);
    return PILCO_OK;
}

int pilco_set_fused_step(pilco_ctx* ctx, int on) {
    if (!ctx) return PILCO_E_SHAPE;
    ctx->fused = (on != 0);
    return PILCO_OK;
}

int pilco_set_small_step(pilco_ctx* ctx, int on) {
    if (!ctx) return PILCO_E_SHAPE;
    ctx->fuse_small = (on != 0);
    return PILCO_OK;
}

int pilco_set_grad_mode(pilco_ctx* ctx, int mode) {
    if (!ctx) return PILCO_E_SHAPE;
    ctx->grad_mode = mode;
    return PILCO_OK;
}

int pilco_set_inline_policy(pilco_ctx* ctx, int on) {
    if (!ctx) return PILCO_E_SHAPE;
    ctx->inline_policy = (on != 0);
    return PILCO_OK;
}

int pilco_set_use_graph(pilco_ctx* ctx, int on) {
    if (!ctx) return PILCO_E_SHAPE;
    ctx->use_graph = (on != 0);
    return PILCO_OK;
}

int pilco_set_pair_timing(pilco_ctx* ctx, int on) {
    if (!ctx) return PILCO_E_SHAPE;
    ctx->time_pairs = (on != 0);
    ctx->timed_pairs = 0;
    return PILCO_OK;
}

int pilco_get_pair_timing(pilco_ctx* ctx, float* ms_pair, int* n_pair_launches) {
    if (!ctx) return PILCO_E_SHAPE;
    if (!ms_pair || !n_pair_launches) return fail(ctx, PILCO_E_SHAPE, "get_pair_timing: null pointer");
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipStreamSynchronize(ctx->st));
    float tot = 0.f;
    for (int t = 0; t < ctx->timed_pairs; ++t) {
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, ctx->pair_events[2 * t], ctx->pair_events[2 * t + 1]));
        tot += ms;
    }
    *ms_pair = tot;
    *n_pair_launches = ctx->timed_pairs;
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
    ENSURE(s.var, (size_t)2 * s.E);   // var | log var (MMModel::lvar)
    ENSURE(s.noise, (size_t)s.E);
    std::vector<double> lv((size_t)s.E);
    for (int i = 0; i < s.E; ++i) lv[i] = std::log(variance[i]);
    HIPCHK(hipMemcpyAsync(s.ls.p, lengthscales, sizeof(double) * s.E * s.D, hipMemcpyHostToDevice, ctx->st));
    HIPCHK(hipMemcpyAsync(s.var.p, variance, sizeof(double) * s.E, hipMemcpyHostToDevice, ctx->st));
    HIPCHK(hipMemcpyAsync(s.var.p + s.E, lv.data(), sizeof(double) * s.E, hipMemcpyHostToDevice, ctx->st));
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
    if (s.M > 0) return fail(ctx, PILCO_E_STATE, "nlml: exact GP only (the sparse objective is pilco_gp_fitc_nlml)");
    if (!nlml) return fail(ctx, PILCO_E_SHAPE, "nlml: null pointer");
    HIPCHK(hipSetDevice(ctx->device));
    if (!s.factor_valid || s.user_factors) {
        s.factor_valid = false;
        if (int r = pilco_gp_factorize(ctx, slot)) return r;
    }
    // Sharded by output like the factorisation it rests on: rank r evaluates the outputs a = r, r + W, ... it
    // owns -- their L, iK, alpha and compacted hyper-parameters are exactly what factorize_exact left on this rank -- and
    // writes them at their global places; one ncclAllGather of (D + 3) doubles per output completes every rank's arrays.
    // Without a communicator the other ranks' entries are NaN and the caller combines (pilco_amd._lib.group_nlml).
    const int E = s.E, D = s.D, npad = s.Npad, N = s.N;
    const int W = s.shW > 0 ? s.shW : 1, rank = W > 1 ? s.shRank : 0;
    const int EL = W > 1 ? s.shOwn : E, ELcap = W > 1 ? s.shEL : E, ELa = std::max(EL, 1);
    const double* ls = s.ls.p;
    const double* var = s.var.p;
    const double* Yt = s.Yt.p;
    const double* beta = s.beta.p;
    if (W > 1) {   // the compacted copies prepare_own made for the owned outputs: [EL][D] | [EL] | [EL] | [EL][Npad]
        ls = s.own.p;
        var = s.own.p + (size_t)ELa * D;
        Yt = s.own.p + (size_t)ELa * (D + 2);
        beta = s.beta.p + (size_t)rank * ELcap * npad;
    }
    const int W3 = D + 3;   // per output: nlml | gradient (D + 2)
    // after factorize_exact: s.K holds L, s.iK the inverse, beta = alpha, Yt the targets (owned outputs, compact)
    const size_t npart = (size_t)ELa * (npad / NB) * (npad / NB) * 34;   // [EL][tiles^2][NLML_MAXD + 2] partial sums of the gradient
    ENSURE(s.vec, std::max((size_t)ELa * npad * 2, npart + (size_t)ELa * (D + 2) + 2 * (size_t)ELa + (size_t)(W + 1) * ELcap * W3));
    double* d_part = s.vec.p;
    double* d_grad = d_part + npart;
    double* d_logdet = d_grad + (size_t)ELa * (D + 2);
    double* d_gath = d_logdet + 2 * (size_t)ELa;   // [W][ELcap][W3] when a communicator completes the result
    // what comes back: gradient [EL][D + 2] | log-determinant sums [EL] | data-fit terms y . beta [EL], ONE copy into pinned
    // memory (rounds 1-4: y and beta themselves came back, 2 x 82 KB into pageable vectors, and were multiplied here)
    const size_t nback = (size_t)ELa * (D + 4);
    if (ctx->pin_io_cap < nback) {
        if (ctx->pin_io) (void)hipHostFree(ctx->pin_io);
        ctx->pin_io = nullptr;
        ctx->pin_io_cap = 0;
        HIPCHK(hipHostMalloc((void**)&ctx->pin_io, sizeof(double) * nback, hipHostMallocDefault));
        ctx->pin_io_cap = nback;
    }
    const double* hg = ctx->pin_io;
    const double* hl = hg + (size_t)ELa * (D + 2);
    const double* hyb = hl + ELa;
    if (EL > 0) {
        launch_logdet(ctx->st, s.K.p, npad, N, EL, d_logdet, Yt, beta);
        if (grad) launch_nlml_grad(ctx->st, s.Xt.p, npad, N, D, ls, var, s.iK.p, beta, EL, d_part, d_grad);
        if (grad) {
            HIPCHK(hipMemcpyAsync(ctx->pin_io, d_grad, sizeof(double) * nback, hipMemcpyDeviceToHost, ctx->st));
        } else {
            HIPCHK(hipMemcpyAsync(ctx->pin_io + (size_t)ELa * (D + 2), d_logdet, sizeof(double) * 2 * ELa, hipMemcpyDeviceToHost, ctx->st));
        }
    }
    HIPCHK(hipStreamSynchronize(ctx->st));
    HIPCHK(hipGetLastError());
    const double nan = std::nan("");
    for (int a = 0; a < E; ++a) {
        nlml[a] = nan;
        if (grad)
            for (int k = 0; k < D + 2; ++k) grad[(size_t)a * (D + 2) + k] = nan;
    }
    std::vector<double> own((size_t)ELcap * W3, 0.0);
    for (int al = 0; al < EL; ++al) {
        const int a = al * W + rank;
        nlml[a] = 0.5 * hyb[al] + hl[al] + 0.5 * N * std::log(2.0 * M_PI);
        own[(size_t)al * W3] = nlml[a];
        if (grad)
            for (int k = 0; k < D + 2; ++k) {
                grad[(size_t)a * (D + 2) + k] = hg[(size_t)al * (D + 2) + k];
                own[(size_t)al * W3 + 1 + k] = hg[(size_t)al * (D + 2) + k];
            }
    }
    if (W > 1 && ctx->comm) {
        const size_t blk = (size_t)ELcap * W3;
        std::vector<double> all((size_t)W * blk);
        HIPCHK(hipMemcpyAsync(d_gath + (size_t)W * blk, own.data(), sizeof(double) * blk, hipMemcpyHostToDevice, ctx->st));
        ncclResult_t r = ncclAllGather(d_gath + (size_t)W * blk, d_gath, blk, ncclDouble, ctx->comm, ctx->st);
        if (r != ncclSuccess) return fail(ctx, PILCO_E_RCCL, std::string("ncclAllGather(nlml): ") + ncclGetErrorString(r));
        HIPCHK(hipMemcpyAsync(all.data(), d_gath, sizeof(double) * W * blk, hipMemcpyDeviceToHost, ctx->st));
        HIPCHK(hipStreamSynchronize(ctx->st));
        for (int a = 0; a < E; ++a) {
            const double* src = all.data() + ((size_t)(a % W) * ELcap + a / W) * W3;
            nlml[a] = src[0];
            if (grad)
                for (int k = 0; k < D + 2; ++k) grad[(size_t)a * (D + 2) + k] = src[1 + k];
        }
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
    if (s.shW > 1) {   // sharded: beta of every output (once complete), iK of no other rank's outputs
        if (iK) return fail(ctx, PILCO_E_STATE, "get_factors: a sharded context holds iK of its own outputs only");
        if (!s.beta_complete) return fail(ctx, PILCO_E_STATE, "get_factors: beta of the other ranks has not arrived yet");
        if (beta)
            for (int a = 0; a < E; ++a)
                HIPCHK(hipMemcpyAsync(beta + (size_t)a * n, s.beta.p + ((size_t)(a % s.shW) * s.shEL + a / s.shW) * npad, sizeof(double) * n,
                                      hipMemcpyDeviceToHost, ctx->st));
        HIPCHK(hipStreamSynchronize(ctx->st));
        return PILCO_OK;
    }
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
    const int W = ctx->nranks, rank = ctx->rank, ELcap = (E + W - 1) / W;
    const int EL = (rank < E) ? (E - rank + W - 1) / W : 0;
    ENSURE(s.beta, (size_t)W * ELcap * npad);
    HIPCHK(hipMemsetAsync(s.beta.p, 0, sizeof(double) * (size_t)W * ELcap * npad, ctx->st));
    for (int a = 0; a < E; ++a)      // row of output a in the [W][ELcap][npad] layout (plain [E][npad] for one rank)
        HIPCHK(hipMemcpyAsync(s.beta.p + ((size_t)(a % W) * ELcap + a / W) * npad, beta + (size_t)a * n, sizeof(double) * n,
                              hipMemcpyHostToDevice, ctx->st));
    std::vector<double> sym;   // must outlive the asynchronous copies below
    if (iK) {
        // The pair kernel visits only the column steps at / right of the diagonal block of a diagonal pair (weight 2):
        // that equals sum_ij iK_ij L_ij only for symmetric iK.  L_aa is symmetric, so sum iK o L = sum sym(iK) o L
        // EXACTLY: a caller-supplied asymmetric iK (predict_given_factorizations accepts any, mgpr.py:91,143-144) is
        // replaced by its symmetric part, which leaves the reference's result unchanged.
        bool asym = false;
        for (int a = 0; a < E && !asym; ++a) {
            const double* A = iK + (size_t)a * n * n;
            for (int i = 0; i < n && !asym; ++i)
                for (int j = 0; j < i; ++j)
                    if (A[(size_t)i * n + j] != A[(size_t)j * n + i]) { asym = true; break; }
        }
        if (asym) {
            sym.resize((size_t)E * n * n);
            for (int a = 0; a < E; ++a) {
                const double* A = iK + (size_t)a * n * n;
                double* Sm = sym.data() + (size_t)a * n * n;
                for (int i = 0; i < n; ++i)
                    for (int j = 0; j < n; ++j) Sm[(size_t)i * n + j] = 0.5 * (A[(size_t)i * n + j] + A[(size_t)j * n + i]);
            }
            iK = sym.data();
        }
        ENSURE(s.iK, (size_t)std::max(EL, 1) * npad * npad);
        HIPCHK(hipMemsetAsync(s.iK.p, 0, sizeof(double) * (size_t)std::max(EL, 1) * npad * npad, ctx->st));
        for (int al = 0; al < EL; ++al) {   // this rank keeps the blocks of its own outputs a = al W + rank
            const int a = al * W + rank;
            HIPCHK(hipMemcpy2DAsync(s.iK.p + (size_t)al * npad * npad, sizeof(double) * npad, iK + (size_t)a * n * n,
                                    sizeof(double) * n, sizeof(double) * n, (size_t)n, hipMemcpyHostToDevice, ctx->st));
        }
    }
    HIPCHK(hipStreamSynchronize(ctx->st));
    s.shW = W; s.shEL = ELcap; s.shOwn = EL; s.shRank = rank;
    s.beta_complete = true;
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
    if (!s.beta_complete) return fail(ctx, PILCO_E_STATE, "predict: beta of the other ranks is missing (attach a communicator before factorising, or pilco_group_sync_model)");
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

// developer aid (tools/): raw copy of a device work buffer of a slot -- 0 At, 1 Bt, 2 reverse-pass row moments, 3 column sums, 4 beta
int pilco_debug_buffer(pilco_ctx* ctx, int slot, int which, double* out, long n) {
    if (int r = check_slot(ctx, slot)) return r;
    Slot& s = ctx->slot[slot];
    const double* src = which == 0 ? s.wk.At : which == 1 ? s.wk.Wt : which == 2 ? s.bwd_mom.p : which == 3 ? s.bwd_cp.p : s.beta.p;
    if (!src || !out || n <= 0) return fail(ctx, PILCO_E_SHAPE, "debug_buffer: bad arguments");
    HIPCHK(hipStreamSynchronize(ctx->st));
    HIPCHK(hipMemcpy(out, src, sizeof(double) * n, hipMemcpyDeviceToHost));
    return PILCO_OK;
}

int pilco_debug_blocks(pilco_ctx* ctx, unsigned long long* out, int n) {
    if (!ctx || !ctx->dbg || !out || n <= 0 || n > PILCO_DBG_WORDS - 64) return PILCO_E_SHAPE;
    HIPCHK(hipStreamSynchronize(ctx->st));
    HIPCHK(hipMemcpy(out, ctx->dbg + 64, n * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return PILCO_OK;
}

int pilco_debug_timestamps(pilco_ctx* ctx, unsigned long long* out32) {  // 64 slots
    if (!ctx) return PILCO_E_SHAPE;
    HIPCHK(hipSetDevice(ctx->device));
    if (!ctx->dbg) {
        HIPCHK(hipMalloc(&ctx->dbg, PILCO_DBG_WORDS * sizeof(unsigned long long)));
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

}  // extern "C"

// ---- sparse GP: FITC factorisation of pilco/models/smgpr.py:24-45 on the device.
// With L = chol(Kmm + 1e-6 I), V = L^{-1} Kmn / G, Am = chol(V V^T + sn2 I), At = L Am:
//   beta = L^{-T} (Am Am^T)^{-1} (V/G) y,   iK = Kmm^{-1} - sn2 At^{-T} At^{-1}.
// Explicit triangular inverses turn every solve into an MFMA GEMM / mat-vec.
int pilco_factorize_fitc(pilco_ctx* ctx, void* slot_ptr) {
    Slot& s = *static_cast<Slot*>(slot_ptr);
    OwnView o{};
    if (int r = prepare_own(ctx, s, o)) return r;
    const int E = std::max(o.EL, 1), EL = o.EL, Mp = s.npad, Np = s.Npad;   // E: batch of OWNED outputs
    const size_t mm = (size_t)Mp * Mp, mn = (size_t)Mp * Np;
    ENSURE(s.K, E * mm);        // Kmm -> L
    ENSURE(s.Linv, E * mm);
    ENSURE(s.iK, E * mm);
    ENSURE(s.Kmn, E * mn);      // Kmn -> V
    ENSURE(s.Am, E * mm);
    ENSURE(s.AmInv, E * mm);
    ENSURE(s.iAt, E * mm);
    ENSURE(s.G, (size_t)E * Np);
    ENSURE(s.beta, (size_t)o.W * o.ELcap * Mp);
    ENSURE(s.Tscr, (size_t)E * Mp * std::max((size_t)Mp, (size_t)(Np + 63) / 64));   // trtri scratch | the right-hand side's per-block partials
    ENSURE(s.vec, (size_t)E * std::max(Mp, Np) * 2);
    DevBuf& Vb = s.V2;
    ENSURE(Vb, E * mn);
    ENSURE(s.ksplit_ws, (size_t)FITC_KSPLIT * E * Mp * Mp);
    hipStream_t st = ctx->st;
    double* beta_own = s.beta.p + (size_t)o.rank * o.ELcap * Mp;
    if (EL == 0) {
        HIPCHK(hipMemsetAsync(ctx->d_info, 0, sizeof(int) * 64, st));
        if (o.W > 1) HIPCHK(hipMemsetAsync(s.beta.p, 0, sizeof(double) * (size_t)o.W * o.ELcap * Mp, st));
        int agreed0 = -1;
        if (int r = agree_not_pd(ctx, o.W, -1, &agreed0)) return r;   // (a rank without outputs still takes part in the agreement)
        if (agreed0 >= 0) {
            ctx->not_pd = agreed0;
            return fail(ctx, PILCO_E_NOT_PD, "FITC Cholesky failed for output " + std::to_string(agreed0));
        }
        s.n = s.M;
        s.iK_null = false;
        return gather_beta(ctx, s, o, Mp);
    }
    // one graph launch for the whole sequence (smgpr.py:24-45 is paid per evaluation of the sparse model's objective)
    std::vector<unsigned long long> key;
    for (const DevBuf* b : {&s.K, &s.Linv, &s.iK, &s.Kmn, &s.Am, &s.AmInv, &s.iAt, &s.G, &s.beta, &s.Tscr, &s.vec, &s.V2,
                            &s.ksplit_ws, &s.Zt, &s.Xt})
        key.push_back((unsigned long long)(uintptr_t)b->p);
    for (const void* q : {(const void*)o.ls, (const void*)o.var, (const void*)o.noise, (const void*)o.Yt, (const void*)ctx->d_info})
        key.push_back((unsigned long long)(uintptr_t)q);
    for (int v : {Mp, Np, s.M, s.N, s.D, EL, o.W, o.rank, o.ELcap}) key.push_back((unsigned long long)v);
    auto chain = [&]() -> int {
    HIPCHK(hipMemsetAsync(ctx->d_info, 0, sizeof(int) * 64, st));
    if (o.W > 1) HIPCHK(hipMemsetAsync(s.beta.p, 0, sizeof(double) * (size_t)o.W * o.ELcap * Mp, st));
    // smgpr.py:27-28: Kmm = K(Z) + 1e-6 I, Kmn = K(Z, X)
    launch_gram(st, s.Zt.p, Mp, s.M, s.Zt.p, Mp, s.M, s.D, o.ls, o.var, E, s.K.p, Mp, Mp, 2, nullptr, 1e-6);
    launch_gram(st, s.Zt.p, Mp, s.M, s.Xt.p, Np, s.N, s.D, o.ls, o.var, E, s.Kmn.p, Mp, Np, 0, nullptr, 0.0);
    launch_potrf(st, s.K.p, Mp, E, s.Linv.p, ctx->d_info, true);                // smgpr.py:29
    launch_trtri(st, s.K.p, Mp, E, s.Linv.p, s.Tscr.p, (long)Mp * Mp);
    GemmDesc g{};
    // V = L^{-1} Kmn  (smgpr.py:30) -- out of place into vec? Kmn is (Mp, Np): use iAt-sized scratch is too small, so
    // write V into a second Kmn-sized buffer: reuse s.Am? no (Mp x Mp).  V goes to s.Kmn2 = s.vec is too small -> allocate.
    g = GemmDesc{};
    g.A = s.Linv.p; g.lda = Mp; g.sA = (long)mm;
    g.B = s.Kmn.p; g.ldb = Np; g.sB = (long)mn;
    g.C = Vb.p; g.ldc = Np; g.sC = (long)mn;
    g.M = Mp; g.N = Np; g.K = Mp; g.alpha = 1.0; g.beta = 0.0; g.k_mode = 3;
    launch_gemm(st, g, false, false, E);
    double* r0 = s.vec.p;
    double* r1 = s.vec.p + (size_t)E * Mp;
    launch_fitc_scale_rhs(st, Vb.p, Mp, Np, E, o.var, o.noise, s.G.p, o.Yt, s.Tscr.p, r0);   // smgpr.py:31-33, and r0 = Vb (y / G) (smgpr.py:40) in the same pass
    // Am = chol(V V^T + sn2 I)  (smgpr.py:34-35)
    g = GemmDesc{};
    g.A = Vb.p; g.lda = Np; g.sA = (long)mn;
    g.B = Vb.p; g.ldb = Np; g.sB = (long)mn;
    g.C = s.Am.p; g.ldc = Mp; g.sC = (long)mm;
    g.M = Mp; g.N = Mp; g.K = Np; g.alpha = 1.0; g.beta = 0.0;
    g.ksplit = FITC_KSPLIT; g.split_ws = s.ksplit_ws.p;
    g.tile_mode = 2;   // V V^T is symmetric: lower tiles + mirror images
    launch_gemm(st, g, false, true, E);
    launch_add_diag(st, s.Am.p, Mp, E, o.noise);
    launch_potrf(st, s.Am.p, Mp, E, s.AmInv.p, ctx->d_info + 32, true);
    launch_trtri(st, s.Am.p, Mp, E, s.AmInv.p, s.Tscr.p, (long)Mp * Mp);
    // iAt = (L Am)^{-1} = Am^{-1} L^{-1}  (smgpr.py:36-37)
    g = GemmDesc{};
    g.A = s.AmInv.p; g.lda = Mp; g.sA = (long)mm;
    g.B = s.Linv.p; g.ldb = Mp; g.sB = (long)mm;
    g.C = s.iAt.p; g.ldc = Mp; g.sC = (long)mm;
    g.M = Mp; g.N = Mp; g.K = Mp; g.alpha = 1.0; g.beta = 0.0; g.k_mode = 4;
    launch_gemm(st, g, false, false, E);
    // beta = L^{-T} Am^{-T} Am^{-1} (V/G) y  (smgpr.py:38-42)
    launch_matvec(st, s.AmInv.p, Mp, E, r0, r1, false);
    launch_matvec(st, s.AmInv.p, Mp, E, r1, r0, true);
    launch_matvec(st, s.Linv.p, Mp, E, r0, beta_own, true);
    // iK = Kmm^{-1} - sn2 iAt^T iAt  (smgpr.py:43-44)
    g = GemmDesc{};
    g.A = s.Linv.p; g.lda = Mp; g.sA = (long)mm;
    g.B = s.Linv.p; g.ldb = Mp; g.sB = (long)mm;
    g.C = s.iK.p; g.ldc = Mp; g.sC = (long)mm;
    g.M = Mp; g.N = Mp; g.K = Mp; g.alpha = 1.0; g.beta = 0.0; g.k_mode = 1;
    launch_gemm(st, g, true, false, E);
    g.A = s.iAt.p;
    g.B = s.iAt.p;
    g.alpha = -1.0; g.alpha_vec = o.noise; g.beta = 1.0; g.k_mode = 1;
    launch_gemm(st, g, true, false, E);
    launch_clear_padding(st, s.iK.p, Mp, s.M, E);
    return PILCO_OK;
    };
    if (int r = run_chain_graph(ctx, s.g_fitc, key, chain)) return r;
    int info[64];
    HIPCHK(hipMemcpyAsync(info, ctx->d_info, sizeof(int) * 64, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    int bad = -1, agreed = -1;
    for (int al = 0; al < std::min(EL, 32) && bad < 0; ++al)
        if (info[al] != 0 || info[32 + al] != 0) bad = al * o.W + o.rank;
    if (int r = agree_not_pd(ctx, o.W, bad, &agreed)) return r;
    if (agreed >= 0) {
        ctx->not_pd = agreed;
        return fail(ctx, PILCO_E_NOT_PD, "FITC Cholesky failed for output " + std::to_string(agreed));
    }
    s.n = s.M;
    s.iK_null = false;
    return gather_beta(ctx, s, o, Mp);
}

