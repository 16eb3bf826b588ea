// Multi-GPU: RCCL communicator, pair ownership / gather-buffer layout, host-mediated exchange (BASELINE config 3).
#include "ctx.h"

#include <thread>

namespace pilco {
// The flag wait as a launch of its own (ranks sharing a GPU): one workgroup, thread r watches rank r's flag.
__global__ void k_peer_wait(unsigned long long* area, int k, int W, int spin) {
    const int t = threadIdx.x;
    if (t >= W) return;
    const unsigned long long epoch = __hip_atomic_load(area, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) + (unsigned long long)k + 1ULL;
    const unsigned long long* flag = area + 8 + (int)(epoch & 1ULL) * W + t;
    int it = 0;
    while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < epoch) {
        if (++it > spin) {
            __hip_atomic_store(area + 1, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            break;
        }
        __builtin_amdgcn_s_sleep(8);
    }
}
}  // namespace pilco
void launch_peer_wait(hipStream_t st, unsigned long long* area, int k, int W, int spin) {
    hipLaunchKernelGGL(pilco::k_peer_wait, dim3(1), dim3(64), 0, st, area, k, W, spin);
}

// ------------------------------------------------------------------ peer exchange (include/pilco_hip.h)
static int peer_alloc_local(pilco_ctx* ctx) {
    PeerXch& x = ctx->xq;
    if (x.local && x.W == ctx->nranks) return PILCO_OK;
    if (int r = peer_detach(ctx)) return r;
    if (ctx->nranks < 2) return fail(ctx, PILCO_E_STATE, "peer exchange: shard_set(rank, nranks >= 2) first");
    HIPCHK(hipSetDevice(ctx->device));
    x.W = ctx->nranks;
    x.cap = 4096;
    const size_t bytes = sizeof(unsigned long long) * xq_area_words(x.W, x.cap);
    HIPCHK(hipExtMallocWithFlags((void**)&x.local, bytes, hipDeviceMallocFinegrained));
    HIPCHK(hipMemset(x.local, 0, bytes));
    HIPCHK(hipStreamSynchronize(nullptr));   // (ordered on the null stream only; the kernels that poll this area run on non-blocking streams)
    HIPCHK(hipHostMalloc((void**)&x.pin, sizeof(unsigned long long) * 256, hipHostMallocDefault));
    memset(x.pin, 0, sizeof(unsigned long long) * 256);
    x.epoch = 0;
    x.ring = 0;
    return PILCO_OK;
}
static int peer_finish_attach(pilco_ctx* ctx, bool share_gpu) {
    PeerXch& x = ctx->xq;
    HIPCHK(hipMalloc((void**)&x.d_peers, sizeof(unsigned long long*) * x.W));
    HIPCHK(hipMemcpy(x.d_peers, x.mapped.data(), sizeof(unsigned long long*) * x.W, hipMemcpyHostToDevice));
    x.wait_kernel = share_gpu;
    x.ready = true;
    for (Slot& s : ctx->slot) s.wk_valid = false;
    return PILCO_OK;
}
int peer_detach(pilco_ctx* ctx) {
    PeerXch& x = ctx->xq;
    if (!x.local && x.mapped.empty()) return PILCO_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->st);
    if (x.members) {
        // in-process group: the other members store segments and flags straight into this context's area.  Before it is
        // freed they are taken off the exchange (their queued work drained, ready cleared, graphs with the exchange baked
        // in dropped): their next rollout runs the collective / host-mediated path or reports the missing exchange -- it
        // does not write into freed memory.  (Peers in OTHER processes hold their own hipIpc mapping of the area, which
        // keeps it alive on their side until they detach.)
        for (pilco_ctx*& m : *x.members) {
            if (m == ctx) { m = nullptr; continue; }
            if (!m) continue;
            (void)hipSetDevice(m->device);
            (void)hipStreamSynchronize(m->st);
            m->xq.ready = false;
            for (auto& ge : m->graph_cache) (void)hipGraphExecDestroy(ge.second);
            m->graph_cache.clear();
            m->graph = nullptr;
        }
        (void)hipSetDevice(ctx->device);
    }
    for (size_t j = 0; j < x.mapped.size(); ++j)
        if (j < x.opened.size() && x.opened[j] && x.mapped[j]) (void)hipIpcCloseMemHandle(x.mapped[j]);
    if (x.d_peers) (void)hipFree(x.d_peers);
    if (x.local) (void)hipFree(x.local);
    if (x.pin) (void)hipHostFree(x.pin);
    // graphs captured with the exchange baked in must not be replayed
    for (auto& ge : ctx->graph_cache) (void)hipGraphExecDestroy(ge.second);
    ctx->graph_cache.clear();
    ctx->graph = nullptr;
    x = PeerXch{};
    return PILCO_OK;
}

extern "C" {

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
    if (ctx->rank != rank || ctx->nranks != nranks) (void)peer_detach(ctx);
    if (ctx->rank != rank || ctx->nranks != nranks)
        for (Slot& s : ctx->slot) s.factor_valid = false;   // each rank factorises only the outputs it owns: the layout changed
    ctx->rank = rank;
    ctx->nranks = nranks;
    // several ranks: the pair kernel whose summation order does not depend on how many pairs a rank holds (variant 2), so that
    // a sharded rollout is bit-identical across rank counts -- and to the single-rank run under the same kernel -- without the
    // caller asking for it; pilco_set_pair_kernel keeps the last word (stream-K: faster per rank, bits depend on the split)
    if (!ctx->variant_user) ctx->variant = nranks > 1 ? 2 : 0;
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

// ---- pure host functions of the stream-K work split (no GPU needed): first step of wave w, and the waves holding the
// partials of local pair k (first wave, slot of the pair in it, last wave) -- the closed forms the pair kernel uses
int pilco_debug_sk_boundary(int w, int waves, int nd, int tdiag, int toff, int ud, int uo, int n_pairs) {
    if (waves <= 0 || nd < 0 || n_pairs < nd || tdiag <= 0 || toff <= 0 || ud <= 0 || uo <= 0) return -1;
    return mm_sk_boundary(w, waves, nd * tdiag, nd * tdiag + (n_pairs - nd) * toff, ud, uo);
}
int pilco_debug_sk_pair_waves(int k, int waves, int nd, int tdiag, int toff, int ud, int uo, int n_pairs, int* out3) {
    if (!out3 || waves <= 0 || nd < 0 || n_pairs < nd || k < 0 || k >= n_pairs || tdiag <= 0 || toff <= 0 || ud <= 0 || uo <= 0)
        return PILCO_E_SHAPE;
    mm_sk_pair_waves(k, waves, nd, tdiag, toff, nd * tdiag + (n_pairs - nd) * toff, ud, uo, &out3[0], &out3[1], &out3[2]);
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
    if (!s.beta_complete) return fail(ctx, PILCO_E_STATE, "shard_pack: beta of the other ranks is missing (pilco_group_sync_model)");
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

// After every context of an in-process group has factorised its own outputs: copy each rank's beta rows to all the
// others (what ncclAllGather does inside pilco_gp_factorize when a communicator is attached).
int pilco_group_sync_model(pilco_ctx** ctxs, int n, int slot) {
    if (!ctxs || n <= 0 || !ctxs[0]) return PILCO_E_SHAPE;
    pilco_ctx* c0 = ctxs[0];
    if (slot < 0 || slot > 1) return fail(c0, PILCO_E_SHAPE, "group_sync_model: bad slot");
    for (int i = 0; i < n; ++i) {
        if (!ctxs[i] || ctxs[i]->nranks != n || ctxs[i]->rank != i) return fail(c0, PILCO_E_STATE, "group_sync_model: context i must be shard_set(i, n)");
        Slot& s = ctxs[i]->slot[slot];
        if (!s.factor_valid || s.shW != n || s.shRank != i) return fail(c0, PILCO_E_STATE, "group_sync_model: rank " + std::to_string(i) + " has not factorised under this layout");
        if (hipSetDevice(ctxs[i]->device) != hipSuccess || hipStreamSynchronize(ctxs[i]->st) != hipSuccess) return fail(c0, PILCO_E_HIP, "group_sync_model: sync failed");
    }
    for (int i = 0; i < n; ++i) {
        pilco_ctx* ctx = ctxs[i];
        Slot& s = ctx->slot[slot];
        HIPCHK(hipSetDevice(ctx->device));
        const size_t blk = (size_t)s.shEL * s.npad;
        for (int j = 0; j < n; ++j) {
            if (j == i) continue;
            Slot& sj = ctxs[j]->slot[slot];
            if (sj.npad != s.npad || sj.shEL != s.shEL) return fail(c0, PILCO_E_STATE, "group_sync_model: the ranks hold different models");
            HIPCHK(hipMemcpyAsync(s.beta.p + (size_t)j * blk, sj.beta.p + (size_t)j * blk, sizeof(double) * blk, hipMemcpyDefault, ctx->st));
        }
        HIPCHK(hipStreamSynchronize(ctx->st));
        s.beta_complete = true;
    }
    return PILCO_OK;
}

// The same exchange over any host transport (ranks in different processes without a communicator): every rank exports
// the beta rows it computed ([ELcap][npad] doubles, ELcap = ceil(E / nranks), npad = pilco_gp_beta_rows's second result),
// the caller all-gathers them in rank order and every rank imports the [nranks][ELcap][npad] block.
int pilco_gp_beta_rows(pilco_ctx* ctx, int slot, int* elcap, int* npad) {
    if (int r = check_slot(ctx, slot)) return r;
    Slot& s = ctx->slot[slot];
    if (!s.factor_valid) return fail(ctx, PILCO_E_STATE, "beta_rows: no current factorisation");
    if (elcap) *elcap = s.shW > 1 ? s.shEL : s.E;
    if (npad) *npad = s.npad;
    return PILCO_OK;
}
int pilco_gp_beta_export(pilco_ctx* ctx, int slot, double* own_rows) {
    if (int r = check_slot(ctx, slot)) return r;
    Slot& s = ctx->slot[slot];
    if (!own_rows) return fail(ctx, PILCO_E_SHAPE, "beta_export: null pointer");
    if (!s.factor_valid || s.shW != ctx->nranks || s.shRank != ctx->rank) return fail(ctx, PILCO_E_STATE, "beta_export: factorise under the current shard layout first");
    HIPCHK(hipSetDevice(ctx->device));
    const size_t blk = (size_t)(s.shW > 1 ? s.shEL : s.E) * s.npad;
    HIPCHK(hipMemcpyAsync(own_rows, s.beta.p + (size_t)(s.shW > 1 ? s.shRank : 0) * blk, sizeof(double) * blk, hipMemcpyDeviceToHost, ctx->st));
    HIPCHK(hipStreamSynchronize(ctx->st));
    return PILCO_OK;
}
int pilco_gp_beta_import(pilco_ctx* ctx, int slot, const double* all_rows) {
    if (int r = check_slot(ctx, slot)) return r;
    Slot& s = ctx->slot[slot];
    if (!all_rows) return fail(ctx, PILCO_E_SHAPE, "beta_import: null pointer");
    if (!s.factor_valid || s.shW != ctx->nranks || s.shRank != ctx->rank) return fail(ctx, PILCO_E_STATE, "beta_import: factorise under the current shard layout first");
    HIPCHK(hipSetDevice(ctx->device));
    const size_t blk = (size_t)(s.shW > 1 ? s.shEL : s.E) * s.npad;
    for (int j = 0; j < s.shW; ++j) {
        if (j == s.shRank) continue;   // the own rows stay as computed
        HIPCHK(hipMemcpyAsync(s.beta.p + (size_t)j * blk, all_rows + (size_t)j * blk, sizeof(double) * blk, hipMemcpyHostToDevice, ctx->st));
    }
    HIPCHK(hipStreamSynchronize(ctx->st));
    s.beta_complete = true;
    return PILCO_OK;
}

// One sharded rollout over n contexts of THIS process (rank i = ctxs[i], any devices): every context runs its own
// pilco_rollout on a host thread; the per-step exchange is done by peer copies between host barriers.  It drives exactly
// the launch sequence the RCCL path runs (PACK launch, exchange, tail launch) with the collective swapped for copies,
// so the sharded rollout can be validated on a single GPU.  Outputs: rank 0's; *mismatch = 1 if any other rank ended
// with a different bit pattern.
int pilco_rollout_group(pilco_ctx** ctxs, int n, const pilco_policy* policy, const pilco_reward_term* rewards, int n_rewards,
                        const double* m0, const double* S0, int H, double* mH, double* SH, double* reward, double* traj,
                        int* mismatch) {
    if (!ctxs || n <= 0 || !ctxs[0]) return PILCO_E_SHAPE;
    pilco_ctx* c0 = ctxs[0];
    if (!policy || !mH || !SH || !reward) return fail(c0, PILCO_E_SHAPE, "rollout_group: null pointer");
    for (int i = 0; i < n; ++i)
        if (!ctxs[i] || ctxs[i]->nranks != n || ctxs[i]->rank != i || ctxs[i]->comm)
            return fail(c0, PILCO_E_STATE, "rollout_group: context i must be shard_set(i, n) and have no communicator");
    const int E = policy->state_dim;
    auto grp = std::make_shared<PeerGroup>();
    grp->ctxs.assign(ctxs, ctxs + n);
    for (int i = 0; i < n; ++i) ctxs[i]->group = grp;
    const size_t nt = traj ? (size_t)(H + 1) * (E + (size_t)E * E) : 0;
    std::vector<std::vector<double>> om(n, std::vector<double>(E)), os(n, std::vector<double>((size_t)E * E)), orw(n, std::vector<double>(1)),
        otr(n, std::vector<double>(nt));
    std::vector<int> rc(n, PILCO_OK);
    std::vector<std::thread> th;
    for (int i = 0; i < n; ++i)
        th.emplace_back([&, i] {
            rc[i] = pilco_rollout(ctxs[i], policy, rewards, n_rewards, m0, S0, H, om[i].data(), os[i].data(), orw[i].data(),
                                  traj ? otr[i].data() : nullptr);
            if (rc[i] != PILCO_OK) grp->fail_all();
        });
    for (auto& t : th) t.join();
    for (int i = 0; i < n; ++i) ctxs[i]->group.reset();
    for (int i = 0; i < n; ++i)
        if (rc[i] != PILCO_OK) {
            if (i != 0) c0->err = "rank " + std::to_string(i) + ": " + ctxs[i]->err;
            return rc[i];
        }
    int mm = 0;
    for (int i = 1; i < n; ++i)
        if (memcmp(om[i].data(), om[0].data(), sizeof(double) * E) || memcmp(os[i].data(), os[0].data(), sizeof(double) * E * E) ||
            memcmp(orw[i].data(), orw[0].data(), sizeof(double)) || (nt && memcmp(otr[i].data(), otr[0].data(), sizeof(double) * nt)))
            mm = 1;
    if (mismatch) *mismatch = mm;
    memcpy(mH, om[0].data(), sizeof(double) * E);
    memcpy(SH, os[0].data(), sizeof(double) * (size_t)E * E);
    *reward = orw[0][0];
    if (traj) memcpy(traj, otr[0].data(), sizeof(double) * nt);
    return PILCO_OK;
}

// Value and gradient of one sharded rollout over n contexts of THIS process (see pilco_rollout_group): every context runs
// pilco_rollout_grad on a host thread; the per-step exchange of the forward chain and the one all-gather of the per-pair
// Jacobian records are done by copies between host barriers.  Outputs of every rank: reward [n], dW [n][U*E], db [n][U].
int pilco_rollout_grad_group(pilco_ctx** ctxs, int n, const pilco_policy* policy, const pilco_reward_term* rewards, int n_rewards,
                             const double* m0, const double* S0, int H, double* reward, double* dW, double* db) {
    if (!ctxs || n <= 0 || !ctxs[0]) return PILCO_E_SHAPE;
    pilco_ctx* c0 = ctxs[0];
    if (!policy || !reward || !dW || !db) return fail(c0, PILCO_E_SHAPE, "rollout_grad_group: null pointer");
    for (int i = 0; i < n; ++i)
        if (!ctxs[i] || ctxs[i]->nranks != n || ctxs[i]->rank != i || ctxs[i]->comm)
            return fail(c0, PILCO_E_STATE, "rollout_grad_group: context i must be shard_set(i, n) and have no communicator");
    const int E = policy->state_dim, U = policy->control_dim;
    auto grp = std::make_shared<PeerGroup>();
    grp->ctxs.assign(ctxs, ctxs + n);
    for (int i = 0; i < n; ++i) ctxs[i]->group = grp;
    std::vector<int> rc(n, PILCO_OK);
    std::vector<std::thread> th;
    for (int i = 0; i < n; ++i)
        th.emplace_back([&, i] {
            rc[i] = pilco_rollout_grad(ctxs[i], policy, rewards, n_rewards, m0, S0, H, reward + i, dW + (size_t)i * U * E, db + (size_t)i * U);
            if (rc[i] != PILCO_OK) grp->fail_all();
        });
    for (auto& t : th) t.join();
    for (int i = 0; i < n; ++i) ctxs[i]->group.reset();
    for (int i = 0; i < n; ++i)
        if (rc[i] != PILCO_OK) {
            if (i != 0) c0->err = "rank " + std::to_string(i) + ": " + ctxs[i]->err;
            return rc[i];
        }
    return PILCO_OK;
}

// ... and for an RbfController (pilco_rollout_grad_rbf on every context; the policy GP is not sharded: every rank evaluates
// all of it inside its link kernel).  Outputs of every rank: reward [n], dX [n][bf*E], dY [n][bf*U], dls [n][U*E].
int pilco_rollout_grad_rbf_group(pilco_ctx** ctxs, int n, const pilco_policy* policy, const pilco_reward_term* rewards, int n_rewards,
                                 const double* m0, const double* S0, int H, const double* Xp, const double* Yp, const double* lsp,
                                 const double* noisep, int bf, double* reward, double* dX, double* dY, double* dls) {
    if (!ctxs || n <= 0 || !ctxs[0]) return PILCO_E_SHAPE;
    pilco_ctx* c0 = ctxs[0];
    if (!policy || !reward || !dX || !dY || !dls) return fail(c0, PILCO_E_SHAPE, "rollout_grad_rbf_group: null pointer");
    for (int i = 0; i < n; ++i)
        if (!ctxs[i] || ctxs[i]->nranks != n || ctxs[i]->rank != i || ctxs[i]->comm)
            return fail(c0, PILCO_E_STATE, "rollout_grad_rbf_group: context i must be shard_set(i, n) and have no communicator");
    const int E = policy->state_dim, U = policy->control_dim;
    auto grp = std::make_shared<PeerGroup>();
    grp->ctxs.assign(ctxs, ctxs + n);
    for (int i = 0; i < n; ++i) ctxs[i]->group = grp;
    std::vector<int> rc(n, PILCO_OK);
    std::vector<std::thread> th;
    for (int i = 0; i < n; ++i)
        th.emplace_back([&, i] {
            rc[i] = pilco_rollout_grad_rbf(ctxs[i], policy, rewards, n_rewards, m0, S0, H, Xp, Yp, lsp, noisep, bf, reward + i,
                                           dX + (size_t)i * bf * E, dY + (size_t)i * bf * U, dls + (size_t)i * U * E);
            if (rc[i] != PILCO_OK) grp->fail_all();
        });
    for (auto& t : th) t.join();
    for (int i = 0; i < n; ++i) ctxs[i]->group.reset();
    for (int i = 0; i < n; ++i)
        if (rc[i] != PILCO_OK) {
            if (i != 0) c0->err = "rank " + std::to_string(i) + ": " + ctxs[i]->err;
            return rc[i];
        }
    return PILCO_OK;
}

// ------------------------------------------------------------------ peer exchange (include/pilco_hip.h)
int pilco_peer_export(pilco_ctx* ctx, void* handle64) {
    if (!ctx || !handle64) return PILCO_E_SHAPE;
    static_assert(sizeof(hipIpcMemHandle_t) <= PILCO_PEER_HANDLE_BYTES, "hipIpcMemHandle_t larger than the ABI slot");
    if (int r = peer_alloc_local(ctx)) return r;
    hipIpcMemHandle_t h;
    HIPCHK(hipIpcGetMemHandle(&h, ctx->xq.local));
    memset(handle64, 0, PILCO_PEER_HANDLE_BYTES);
    memcpy(handle64, &h, sizeof(h));
    return PILCO_OK;
}

int pilco_peer_attach(pilco_ctx* ctx, const void* handles, int share_gpu) {
    if (!ctx || !handles) return PILCO_E_SHAPE;
    PeerXch& x = ctx->xq;
    if (!x.local || x.W != ctx->nranks) return fail(ctx, PILCO_E_STATE, "peer_attach: pilco_peer_export first");
    if (x.ready) return PILCO_OK;
    HIPCHK(hipSetDevice(ctx->device));
    x.mapped.assign(x.W, nullptr);
    x.opened.assign(x.W, 0);
    for (int j = 0; j < x.W; ++j) {
        if (j == ctx->rank) {
            x.mapped[j] = x.local;
            continue;
        }
        hipIpcMemHandle_t h;
        memcpy(&h, (const char*)handles + (size_t)j * PILCO_PEER_HANDLE_BYTES, sizeof(h));
        void* q = nullptr;
        hipError_t e = hipIpcOpenMemHandle(&q, h, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            const std::string msg = std::string("peer_attach: hipIpcOpenMemHandle(rank ") + std::to_string(j) + "): " + hipGetErrorString(e);
            (void)peer_detach(ctx);
            return fail(ctx, PILCO_E_HIP, msg);
        }
        x.mapped[j] = (unsigned long long*)q;
        x.opened[j] = 1;
    }
    return peer_finish_attach(ctx, share_gpu != 0);
}

int pilco_group_peer_attach(pilco_ctx** ctxs, int n) {
    if (!ctxs || n < 2 || !ctxs[0]) return PILCO_E_SHAPE;
    pilco_ctx* c0 = ctxs[0];
    bool share = false;
    for (int i = 0; i < n; ++i) {
        if (!ctxs[i] || ctxs[i]->nranks != n || ctxs[i]->rank != i) return fail(c0, PILCO_E_STATE, "group_peer_attach: context i must be shard_set(i, n)");
        for (int j = 0; j < i; ++j) share = share || ctxs[j]->device == ctxs[i]->device;
    }
    for (int i = 0; i < n; ++i) {
        if (int r = peer_detach(ctxs[i])) return r;
        if (int r = peer_alloc_local(ctxs[i])) {
            if (i) c0->err = ctxs[i]->err;
            return r;
        }
    }
    for (int i = 0; i < n; ++i) {
        pilco_ctx* ctx = ctxs[i];
        HIPCHK(hipSetDevice(ctx->device));
        ctx->xq.mapped.assign(n, nullptr);
        ctx->xq.opened.assign(n, 0);
        for (int j = 0; j < n; ++j) {
            ctx->xq.mapped[j] = ctxs[j]->xq.local;
            if (ctxs[j]->device != ctx->device) {
                hipError_t e = hipDeviceEnablePeerAccess(ctxs[j]->device, 0);
                if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) return fail(c0, PILCO_E_HIP, std::string("hipDeviceEnablePeerAccess: ") + hipGetErrorString(e));
                (void)hipGetLastError();
            }
        }
        if (int r = peer_finish_attach(ctx, share)) {
            if (i) c0->err = ctx->err;
            return r;
        }
    }
    auto members = std::make_shared<std::vector<pilco_ctx*>>(ctxs, ctxs + n);
    for (int i = 0; i < n; ++i) ctxs[i]->xq.members = members;
    return PILCO_OK;
}

int pilco_peer_detach(pilco_ctx* ctx) { return ctx ? peer_detach(ctx) : PILCO_E_SHAPE; }
int pilco_peer_attached(const pilco_ctx* ctx) { return (ctx && ctx->xq.ready) ? 1 : 0; }

int pilco_comm_rank(const pilco_ctx* ctx) { return ctx ? ctx->rank : -1; }
int pilco_comm_size(const pilco_ctx* ctx) { return ctx ? ctx->nranks : -1; }
// ranks RCCL itself reports for the attached communicator (ncclCommCount); 0: no communicator; -1: RCCL error
int pilco_comm_count(const pilco_ctx* ctx) {
    if (!ctx || !ctx->comm) return 0;
    int n = 0;
    return ncclCommCount(ctx->comm, &n) == ncclSuccess ? n : -1;
}

}  // extern "C"
