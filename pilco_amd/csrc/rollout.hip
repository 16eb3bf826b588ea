// Rollouts (PILCO.predict / propagate, pilco/models/pilco.py:118-153): plan, launch sequence, hipGraph capture
// and replay, and the policy / reward evaluation entry points.
#include "ctx.h"
#include <chrono>

namespace {

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

}  // namespace

int setup_rollout(pilco_ctx* ctx, const pilco_policy* pol, const pilco_reward_term* rw, int n_rw, int H, bool want_traj,
                  RolloutPlan& plan) {
    Slot& s = ctx->slot[0];
    if (!s.factor_valid) return fail(ctx, PILCO_E_STATE, "rollout: dynamics model has no current factorisation");
    if (!s.beta_complete) return fail(ctx, PILCO_E_STATE, "rollout: beta of the other ranks is missing (attach a communicator before factorising, or pilco_group_sync_model)");
    if (s.shW != ctx->nranks || s.shRank != ctx->rank) return fail(ctx, PILCO_E_STATE, "rollout: the model was factorised under a different rank layout; factorise again");
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
        if (ctx->nranks != 1 && !(ctx->inline_policy && rbf_inline_lds_doubles(E, U, ps.n) > 0))
            return fail(ctx, PILCO_E_STATE, "rollout: with several ranks an RbfController must be small enough for the inline evaluation (pilco_set_inline_policy)");
        if (int r = build_work(ctx, ps)) return r;
    }
    if (pol->kind < 0 || pol->kind > 2) return fail(ctx, PILCO_E_SHAPE, "rollout: unknown policy kind");
    if (n_rw < 0 || n_rw > MAX_REWARD_TERMS || (n_rw > 0 && !rw)) return fail(ctx, PILCO_E_SHAPE, "rollout: 0..4 reward terms supported");
    if (int r = build_work(ctx, s)) return r;
    // state: 2 x (m_x[E] s_x[E*E]) | s1[E*D] | reward[1]
    const size_t n_state = 2 * ((size_t)E + E * E) + 2 * (size_t)E * D + 1 + 8;
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
    plan.s1b[0] = g.s1;
    plan.s1b[1] = g.s1 + (size_t)E * D;
    g.reward = g.s1 + 2 * (size_t)E * D;
    g.traj = want_traj ? ctx->traj.p : nullptr;
    g.pol_kind = pol->kind;
    g.squash = pol->squash;
    if (pol->kind == PILCO_POLICY_RBF) {
        g.pwk = ctx->slot[PILCO_SLOT_POLICY].wk;
        g.pvar = ctx->slot[PILCO_SLOT_POLICY].var.p;
        g.pmd = model_of(ctx->slot[PILCO_SLOT_POLICY]);
        g.pol_lds = rbf_inline_lds_doubles(E, U, ctx->slot[PILCO_SLOT_POLICY].n);
        g.pol_inline = (ctx->inline_policy && g.pol_lds > 0) ? 1 : 0;
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
    // the policy / reward parameters of consecutive rollouts are usually the same bytes (bench loop, restarts,
    // compute_reward after an optimiser step): upload only when they changed
    if (ctx->params_dev != ctx->params.p || ctx->params_host.size() != n_par ||
        memcmp(ctx->params_host.data(), hp.data(), sizeof(double) * n_par) != 0) {
        HIPCHK(hipMemcpyAsync(ctx->params.p, hp.data(), sizeof(double) * n_par, hipMemcpyHostToDevice, ctx->st));
        HIPCHK(hipStreamSynchronize(ctx->st));  // hp is a local vector
        ctx->params_host = hp;
        ctx->params_dev = ctx->params.p;
    }
    plan.E = E; plan.D = D; plan.U = U;
    return PILCO_OK;
}

// The fused heads need the serial link's and the operand kernel's LDS side by side in one workgroup: wide models
// (D > 24 with many outputs) exceed the CU's 160 KB and run the three-kernel step instead (same results).
// The answer must be the SAME ON EVERY RANK (it selects between the peer exchange and the collective path, and ranks
// that disagree wait for each other forever): it is computed from the model dimensions and the rank COUNT only -- the
// geometry of rank 0, which holds the largest share of pairs and outputs under the round-robin dealing.
static bool fused_heads_fit(pilco_ctx* ctx, const RolloutPlan& plan) {
    const Slot& s = ctx->slot[0];
    const bool rbf = plan.g.pol_kind == PILCO_POLICY_RBF;
    GlueArgs gl = plan.g;
    if (ctx->nranks > 1) {
        const int W = ctx->nranks, P = s.E * (s.E + 1) / 2;
        gl.wk.PL = (P + W - 1) / W;
        gl.wk.EL = (s.E + W - 1) / W;
        mm_prep_chunks(s.npad, std::max(gl.wk.PL, 1), gl.wk.EL, &gl.wk.NCH, &gl.wk.NCHM);
    } else if (s.wk.PL <= 0) {
        return true;
    }
    const bool rbf_k = rbf && !gl.pol_inline;   // the policy GP as launches of its own (an inline policy is part of the link)
    gl.flags = GF_TRAJ | GF_POLICY | GF_PACK | GF_ASSEMBLE | GF_PROPAGATE | (rbf_k ? (GF_RBF_PRE | GF_RBF_POST) : 0);
    const int rew_E = plan.g.n_rewards > 0 ? plan.E : 0;
    bool fits = mm_fused_head_fits(model_of(s), rew_E, gl);
    if (fits && rbf_k) fits = mm_fused_head_fits(model_of(ctx->slot[PILCO_SLOT_POLICY]), rew_E, gl);
    return fits;
}

// The peer exchange carries a rollout when it is attached, the model is sharded, the policy is not an RbfController
// (its GP is not sharded) and the segments fit the exchange slots; otherwise the RCCL / group path runs.
static bool peer_rollout_applies(pilco_ctx* ctx, const RolloutPlan& plan, int H) {
    const Slot& s = ctx->slot[0];
    // (an RbfController rides along when it is evaluated inside the link: every rank evaluates the whole, unsharded policy)
    return ctx->xq.ready && ctx->nranks > 1 && ctx->xq.W == ctx->nranks && H > 0 &&
           (plan.g.pol_kind != PILCO_POLICY_RBF || plan.g.pol_inline) &&
           s.wk.SEG <= ctx->xq.cap && !plan.g.tape && !plan.jrec && fused_heads_fit(ctx, plan);
}
// Host side of a rollout's exchanges: the epoch base goes up before the rollout's launches (outside any graph: the
// value changes per replay), `n` exchanges are accounted for afterwards.
static int xq_begin(pilco_ctx* ctx) {
    PeerXch& x = ctx->xq;
    unsigned long long* slot = x.pin + (x.ring++ & 127u);   // a ring: rollouts may be queued without a host sync in between
    *slot = x.epoch;
    HIPCHK(hipMemcpyAsync(x.local, slot, sizeof(unsigned long long), hipMemcpyHostToDevice, ctx->st));
    return PILCO_OK;
}

// enqueue one full rollout on the stream (initial state already in plan.st[0]); the final
// state ends up in plan.st[H & 1].  The reward of state t (pilco.py:133) is evaluated by the
// second workgroup of the glue launch that turns state t into state t+1.
static int enqueue_rollout_steps(pilco_ctx* ctx, RolloutPlan& plan, int H, std::vector<hipEvent_t>* pair_ev);
// Jacobian tape: the sums, moments and records of steps [t0, t1) in two launches behind the chain (rollout_jtape enqueues
// them per chunk of steps, last steps first, each followed by its download: the host's reverse sweep works on one chunk
// while the device finishes the next)
// Does a value-and-gradient rollout of this plan run its steps as the one-launch small step (small_sweep)?  Returns the
// workgroups per pair (row chunks x column splits) or 0.  (The same conditions enqueue_rollout_steps applies, plus those of its fused-head branch.)
static bool fused_heads_fit(pilco_ctx* ctx, const RolloutPlan& plan);
static int device_cus_of(int device) {
    static int cached[64] = {};
    int& c = cached[device & 63];
    if (c == 0) {
        hipDeviceProp_t prop;
        c = (hipGetDeviceProperties(&prop, device) == hipSuccess) ? prop.multiProcessorCount : 256;
    }
    return c;
}
// Column splits of the one-launch small step (MMWork::NCS): with 64-row workgroups whose operands stay in LDS, a pair's
// columns are dealt over up to four workgroups as long as the whole launch stays ONE round of the chip (one workgroup per
// CU) -- a small model with few outputs would otherwise leave most CUs idle while 40 of them work through the pair sums.
static int small_col_splits(pilco_ctx* ctx, const Slot& s, bool rew) {
    if (s.npad / s.wk.NCH != 64 || s.wk.KP > 16) return 1;
    static const int forced = getenv("PILCO_SMALL_NCS") ? atoi(getenv("PILCO_SMALL_NCS")) : 0;   // (tools: A/B)
    const int cus = device_cus_of(ctx->device), spare = s.wk.EL * s.wk.NCHM + (rew ? 1 : 0);
    int ncs = 1;
    while (ncs * 2 <= s.wk.NCH && ncs * 2 <= 4 && s.wk.PL * s.wk.NCH * ncs * 2 + spare <= cus) ncs *= 2;
    if (forced > 0 && forced <= s.wk.NCH && (forced & (forced - 1)) == 0 && forced <= 4) ncs = forced;
    return ncs;
}
static int jac_small_chunks(pilco_ctx* ctx, const RolloutPlan& plan, int H) {
    const Slot& s = ctx->slot[0];
    const int D = s.D, dtk = D <= 4 ? 4 : D <= 6 ? 6 : D <= 8 ? 8 : D <= 10 ? 10 : D == 11 ? 11 : D <= 12 ? 12 : D <= 14 ? 14 : D <= 16 ? 16 : 32;
    const bool rbf = plan.g.pol_kind == PILCO_POLICY_RBF;
    if (!ctx->fuse_small || !ctx->fused || ctx->nranks != 1 || ctx->comm || s.wk.PL <= 0 || H <= 0 || ctx->time_pairs || MM_ABL(s.wk, 255)) return 0;
    if (D > 14 || s.npad > 256 || s.npad / s.wk.NCH != 64 || s.wk.KP != mm_kp(dtk) || (s.wk.vsep != 0) != mm_vsep(dtk) || s.wk.KP > 16) return 0;
    if (rbf && !plan.g.pol_inline) return 0;
    if (!fused_heads_fit(ctx, plan)) return 0;
    return s.wk.NCH * small_col_splits(ctx, s, plan.g.n_rewards > 0 && !MM_ABL(s.wk, 8));
}
static void jac_finish_range(pilco_ctx* ctx, const RolloutPlan& plan, int t0, int t1, const RevLocalArgs* rl = nullptr) {
    Slot& s = ctx->slot[0];
    hipStream_t st = ctx->st;
    const int D = plan.D, E = plan.E, P = s.wk.PL;
    const size_t TS = (size_t)D + D * D + (size_t)E * D + E + (size_t)E * E + (size_t)D * E;
    const size_t o = (size_t)t0;
    launch_mm_jac_finish(st, model_of(s), s.wk, t1 - t0, s.jac_rowmom.p + o * mm_jac_rowmom_size(s.npad, P),
                         s.jac_cpart.p + o * mm_jac_cpart_size(s.npad, P, s.wk.EL), s.jac_head.p + o * mm_jac_head_size(D, E, P),
                         s.jac_part.p + o * mm_jac_part_size(D, E, P, s.npad), plan.g.tape + o * TS, TS, plan.jrec + o * plan.jstride,
                         plan.jsmall, rl);
}
int enqueue_rollout(pilco_ctx* ctx, RolloutPlan& plan, int H, std::vector<hipEvent_t>* pair_ev) {
    return enqueue_rollout_steps(ctx, plan, H, pair_ev);
}

static int enqueue_rollout_steps(pilco_ctx* ctx, RolloutPlan& plan, int H, std::vector<hipEvent_t>* pair_ev) {
    Slot& s = ctx->slot[0];
    const MMModel md = model_of(s);
    const int E = plan.E;
    GlueArgs g = plan.g;
    const bool rew = g.n_rewards > 0 && !MM_ABL(s.wk, 8);
    HIPCHK(hipMemsetAsync(g.reward, 0, sizeof(double), ctx->st));
    g.step = 0;
    g.m_x = plan.st[0];
    g.s_x = plan.st[0] + E;
    g.m_out = nullptr;
    g.s_out = nullptr;
    const bool rbf = (g.pol_kind == PILCO_POLICY_RBF);
    // Jacobian tape (bwd.hip): the dynamics step runs the reverse sweep in place of the forward pair kernel; the serial
    // link packs N_ab from the per-workgroup partials the sweep leaves in the tile-partial layout
    const bool jac = plan.jrec != nullptr;
    MMWork wk0 = s.wk;
    if (jac) {
        wk0.sk_waves = 0;
        wk0.NT = mm_jac_nt(s.npad, s.wk.P);
        wk0.pair_part = s.jac_np.p;
        g.wk = wk0;
    }
    const size_t j_rm = jac ? mm_jac_rowmom_size(s.npad, s.wk.PL) : 0, j_cp = jac ? mm_jac_cpart_size(s.npad, s.wk.PL, s.wk.EL) : 0,
                 j_hd = jac ? mm_jac_head_size(s.D, s.E, s.wk.PL) : 0;
    auto dyn_pairs = [&](const MMWork& w, int t) {
        if (jac)
            launch_mm_sweep(ctx->st, md, w, s.jac_rowmom.p + (size_t)t * j_rm, s.jac_cpart.p + (size_t)t * j_cp,
                            s.jac_head.p + (size_t)t * j_hd, s.jac_np.p);
        else
            launch_mm_pair(ctx->st, md, w, ctx->variant);
    };
    bool fits = fused_heads_fit(ctx, plan);
    // An RbfController evaluated inside the link (GlueArgs::pol_inline) makes the step the LinearController's: two launches.
    const bool inl = rbf && g.pol_inline && ctx->fused && fits && ctx->nranks == 1 && !ctx->comm && s.wk.PL > 0 && H > 0;
    const bool inl_peer = rbf && g.pol_inline && peer_rollout_applies(ctx, plan, H);   // sharded rollout over the peer exchange
    // Several ranks: the policy GP is never sharded (every rank holds all of it) and its own launches would deal its pairs over
    // the ranks, so it is evaluated INSIDE the link -- by the fused head over the peer exchange, or (value-and-gradient
    // rollouts: the Jacobian tape runs the three-kernel step with the all-gather) by the link kernel itself.
    const bool inl_link = rbf && g.pol_inline && !inl && !inl_peer && (ctx->nranks != 1 || ctx->comm);
    if (rbf && (ctx->nranks != 1 || ctx->comm) && !inl_peer && !inl_link)
        return fail(ctx, PILCO_E_STATE, "rollout: several ranks run an RbfController only with the inline policy (at most 256 basis functions, see pilco_set_inline_policy)");
    if (rbf && !inl && !inl_peer && !inl_link && g.pol_inline) {   // not this time (three-kernel step, ...): the policy GP gets its own launches
        g.pol_inline = 0;
        plan.g.pol_inline = 0;
        fits = fused_heads_fit(ctx, plan);
    }
    if (ctx->fused && fits && (!rbf || inl) && ctx->nranks == 1 && !ctx->comm && s.wk.PL > 0 && H > 0) {
        // Fused head: launch h = 0..H-1 is [serial link producing state h and its joint Gaussian | operands of step h],
        // followed by the pair kernel of step h; one plain glue launch closes the rollout.  What the link reads
        // (previous step's pair_isdet / mean_part / s1 / state) and what the same launch writes alternate between two
        // buffer sets, because the workgroups of one launch are not ordered.
        MMWork wkb[2] = {wk0, wk0};
        wkb[1].pair_isdet = s.alt_isdet;
        wkb[1].mean_part = s.alt_mean;
        // Small models: the pair sums ride in the head launch (prep_device.h) -- one launch per step.  The head instantiated for
        // this input dimension carries the pair arithmetic of ONE contraction depth (that of D = DT); the workgroup's rows must
        // be whole 32-row groups and one thread per point must cover the columns.
        const int dtk = s.D <= 4 ? 4 : s.D <= 6 ? 6 : s.D <= 8 ? 8 : s.D <= 10 ? 10 : s.D == 11 ? 11 : s.D <= 12 ? 12 : s.D <= 14 ? 14 : s.D <= 16 ? 16 : 32;
        // Value-and-gradient rollouts (the Jacobian tape) take the same road when the workgroup's operands stay in LDS (64-row
        // chunks, KP <= 16; the tape serves D <= 14): the pair workgroups run the reverse sweep of their block (small_sweep).
        const bool small_ok = ctx->fuse_small && s.npad <= 256 && (s.npad / wk0.NCH) % 32 == 0 && wk0.KP == mm_kp(dtk) &&
                              (wk0.vsep != 0) == mm_vsep(dtk) && !pair_ev && !MM_ABL(s.wk, 255);
        // value-and-gradient rollouts: the PLAN has decided (jac_small_chunks sized the tape and the finish for it); the step
        // geometry seen here must agree, or the finish would read the tape in the wrong layout
        if (jac && plan.jsmall > 0 && !small_ok) return fail(ctx, PILCO_E_STATE, "rollout: the planned one-launch small step does not fit the step's geometry");
        const bool small = jac ? (plan.jsmall > 0) : (small_ok && ctx->variant == 0);
        if (small)
            for (int k = 0; k < 2; ++k) {
                wkb[k].fuse_pair = jac ? 2 : 1;
                wkb[k].sk_waves = 0;           // the link packs tile partials: one per (pair, row chunk)
                wkb[k].NCS = small_col_splits(ctx, s, rew);
                wkb[k].share_cu = ctx->share_cu;
                wkb[k].NT = wk0.NCH * wkb[k].NCS;
                wkb[k].pair_part = s.w_fpart.p + (size_t)k * std::max(wk0.PL, 1) * wk0.NCH * 4 * 2;
            }
        size_t evi = 0;
        for (int h = 0; h < H; ++h) {
            GlueArgs gh = g;
            gh.step = h;
            gh.dbg_off = h > 0 ? 48 : 0;              // fused heads stamp slots 56..61 (the closing k_glue keeps 8..13)
            gh.wk = wkb[(h + 1) & 1];                 // read side: written by launch h - 1
            gh.flags = GF_TRAJ | GF_POLICY | (h > 0 ? (GF_PACK | GF_ASSEMBLE | GF_PROPAGATE) : 0);
            gh.m_x = plan.st[h > 0 ? (h - 1) & 1 : 0];
            gh.s_x = gh.m_x + E;
            gh.m_out = h > 0 ? plan.st[h & 1] : nullptr;
            gh.s_out = h > 0 ? plan.st[h & 1] + E : nullptr;
            gh.s1 = plan.s1b[(h + 1) & 1];
            gh.s1_out = plan.s1b[h & 1];
            PrepReward pr{};
            if (rew) {   // reward of state h (pilco.py:133), from the link's LDS copy of the state
                pr.n = g.n_rewards;
                pr.E = E;
                for (int i = 0; i < g.n_rewards; ++i) pr.rw[i] = g.rw[i];
                pr.reward = g.reward;
            }
            if (small && jac) {
                MMWork wh = wkb[h & 1];
                wh.sw_gpart = s.jac_rowmom.p + (size_t)h * j_rm;
                launch_mm_prep(ctx->st, md, wh, rew ? &pr : nullptr, &gh);
                continue;
            }
            launch_mm_prep(ctx->st, md, wkb[h & 1], rew ? &pr : nullptr, &gh);
            if (small) continue;
            if (pair_ev) HIPCHK(hipEventRecord((*pair_ev)[evi++], ctx->st));
            dyn_pairs(wkb[h & 1], h);
            if (pair_ev) HIPCHK(hipEventRecord((*pair_ev)[evi++], ctx->st));
        }
        GlueArgs gf = g;
        gf.step = H;
        gf.wk = wkb[(H - 1) & 1];
        gf.flags = GF_PACK | GF_ASSEMBLE | GF_PROPAGATE | GF_TRAJ;
        gf.m_x = plan.st[(H - 1) & 1];
        gf.s_x = gf.m_x + E;
        gf.m_out = plan.st[H & 1];
        gf.s_out = gf.m_out + E;
        gf.s1 = plan.s1b[(H - 1) & 1];
        gf.s1_out = nullptr;
        launch_glue(ctx->st, gf);
        return PILCO_OK;
    }
    Slot& ps = ctx->slot[PILCO_SLOT_POLICY];
    const MMModel pmd = rbf ? model_of(ps) : MMModel{};
    if (peer_rollout_applies(ctx, plan, H)) {
        // Sharded rollout with the peer exchange (GlueArgs::xq): per step
        //   head  [wait for the W flags of exchange h - 1, segments from the own area -> assemble / propagate / controller
        //          / joint, redundantly in every workgroup | operands of step h]        (ranks without pairs: a plain k_glue)
        //   pairs of this rank
        //   push  [pack this rank's segment, store it into every rank's area, raise the flag there]   (one workgroup)
        // No host involvement and no collective launch per step; the state and s1 alternate between two buffers as in the
        // single-rank fused path.  The reward of state h is taken by head h (k_glue launches: by the launch that
        // propagates state h, from its pre-propagation copy), i.e. in the same order on every rank.
        PeerXch& x = ctx->xq;
        {   // executed now (not being captured into a graph): this rollout's epoch base goes up ahead of its launches
            hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
            HIPCHK(hipStreamIsCapturing(ctx->st, &cs));
            if (cs == hipStreamCaptureStatusNone) {
                if (int r = xq_begin(ctx)) return r;
                x.epoch += (unsigned long long)H;
            }
        }
        const int spin = 2000000;   // ~2 s of polling before a wait gives up
        auto with_xq = [&](GlueArgs& ga, int k) {
            ga.xq = x.local;
            ga.xq_k = k;
            ga.xq_W = x.W;
            ga.xq_cap = x.cap;
            ga.xq_spin = x.wait_kernel ? 1000 : spin;   // behind a wait launch the flags are already up
        };
        PrepReward pr{};
        if (rew) {
            pr.n = g.n_rewards;
            pr.E = E;
            for (int i = 0; i < g.n_rewards; ++i) pr.rw[i] = g.rw[i];
            pr.reward = g.reward;
        }
        size_t evi = 0;
        for (int h = 0; h < H; ++h) {
            GlueArgs gh = g;
            gh.step = h;
            gh.wk = s.wk;
            gh.flags = GF_TRAJ | GF_POLICY | (h > 0 ? (GF_ASSEMBLE | GF_PROPAGATE) : 0);
            gh.m_x = plan.st[h > 0 ? (h - 1) & 1 : 0];
            gh.s_x = gh.m_x + E;
            gh.m_out = h > 0 ? plan.st[h & 1] : nullptr;
            gh.s_out = h > 0 ? plan.st[h & 1] + E : nullptr;
            gh.s1 = plan.s1b[(h + 1) & 1];
            gh.s1_out = plan.s1b[h & 1];
            if (h > 0) {
                with_xq(gh, h - 1);
                if (x.wait_kernel) launch_peer_wait(ctx->st, x.local, h - 1, x.W, spin);
            }
            if (s.wk.PL > 0) {
                launch_mm_prep(ctx->st, md, s.wk, rew ? &pr : nullptr, &gh);
                if (pair_ev) HIPCHK(hipEventRecord((*pair_ev)[evi++], ctx->st));
                launch_mm_pair(ctx->st, md, s.wk, ctx->variant);
                if (pair_ev) HIPCHK(hipEventRecord((*pair_ev)[evi++], ctx->st));
            } else {
                launch_glue(ctx->st, gh, rew && h > 0);   // its reward workgroup takes the pre-propagation state h - 1
            }
            GlueArgs gp = g;
            gp.step = h + 1;
            gp.wk = s.wk;
            gp.flags = GF_PACK;
            with_xq(gp, h);
            gp.xq_peers = x.d_peers;
            launch_glue(ctx->st, gp);
        }
        GlueArgs gf = g;
        gf.step = H;
        gf.wk = s.wk;
        gf.flags = GF_ASSEMBLE | GF_PROPAGATE | GF_TRAJ;
        gf.m_x = plan.st[(H - 1) & 1];
        gf.s_x = gf.m_x + E;
        gf.m_out = plan.st[H & 1];
        gf.s_out = gf.m_out + E;
        gf.s1 = plan.s1b[(H - 1) & 1];
        gf.s1_out = nullptr;
        with_xq(gf, H - 1);
        if (x.wait_kernel) launch_peer_wait(ctx->st, x.local, H - 1, x.W, spin);
        launch_glue(ctx->st, gf, rew && s.wk.PL == 0);
        return PILCO_OK;
    }
    if (ctx->fused && fits && rbf && ctx->nranks == 1 && !ctx->comm && s.wk.PL > 0 && ps.wk.PL > 0 && H > 0) {
        // Fused heads with an RbfController (controllers.py:108-121): the policy is a moment-matching GP of its own, so a
        // step is two head + pair rounds and the serial link splits in two:
        //   policy head   [pack / assemble / propagate of step h - 1 -> state h | operands of the POLICY GP at state h]
        //   policy pairs
        //   dynamics head [reduce the policy GP, S -= diag(var - 1e-6), squash, joint Gaussian | operands of step h]
        //   dynamics pairs
        // four launches per step instead of six (two of them single-workgroup glue launches).  What a head's link reads
        // was written by EARLIER launches and what its prep part writes belongs to the other GP: no double buffering
        // beyond the state's.
        size_t evi = 0;
        for (int h = 0; h < H; ++h) {
            GlueArgs ga = g;
            ga.step = h;
            ga.dbg_off = h > 0 ? 48 : 0;
            ga.flags = GF_TRAJ | GF_RBF_PRE | (h > 0 ? (GF_PACK | GF_ASSEMBLE | GF_PROPAGATE) : 0);
            ga.m_x = plan.st[h > 0 ? (h - 1) & 1 : 0];
            ga.s_x = ga.m_x + E;
            ga.m_out = h > 0 ? plan.st[h & 1] : nullptr;
            ga.s_out = h > 0 ? plan.st[h & 1] + E : nullptr;
            PrepReward pr{};
            if (rew) {   // reward of state h (pilco.py:133), from the link's LDS copy of the state
                pr.n = g.n_rewards;
                pr.E = E;
                for (int i = 0; i < g.n_rewards; ++i) pr.rw[i] = g.rw[i];
                pr.reward = g.reward;
            }
            launch_mm_prep(ctx->st, pmd, ps.wk, rew ? &pr : nullptr, &ga);
            launch_mm_pair(ctx->st, pmd, ps.wk, ctx->variant);
            GlueArgs gc = g;
            gc.step = h;
            gc.flags = GF_RBF_POST | GF_POLICY;
            gc.m_x = plan.st[h & 1];
            gc.s_x = gc.m_x + E;
            gc.m_out = nullptr;
            gc.s_out = nullptr;
            launch_mm_prep(ctx->st, md, s.wk, nullptr, &gc);
            if (pair_ev) HIPCHK(hipEventRecord((*pair_ev)[evi++], ctx->st));
            dyn_pairs(s.wk, h);
            if (pair_ev) HIPCHK(hipEventRecord((*pair_ev)[evi++], ctx->st));
        }
        GlueArgs gf = g;
        gf.step = H;
        gf.flags = GF_PACK | GF_ASSEMBLE | GF_PROPAGATE | GF_TRAJ;
        gf.m_x = plan.st[(H - 1) & 1];
        gf.s_x = gf.m_x + E;
        gf.m_out = plan.st[H & 1];
        gf.s_out = gf.m_out + E;
        launch_glue(ctx->st, gf);
        return PILCO_OK;
    }
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
    const bool rbf_l = rbf && !inl_link;   // the policy GP as launches of its own (an inline policy is part of the link kernel)
    g.flags = GF_TRAJ | (H > 0 ? (rbf_l ? GF_RBF_PRE : GF_POLICY) : 0);
    launch_glue(ctx->st, g);
    if (rbf_l && H > 0) policy_stage(g);
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
            if (ctx->dbg && MM_ABL(s.wk, 64)) launch_stamp(ctx->st, ctx->dbg, 30);
            if (pair_ev) HIPCHK(hipEventRecord((*pair_ev)[evi++], ctx->st));
            dyn_pairs(s.wk, t);
            if (pair_ev) HIPCHK(hipEventRecord((*pair_ev)[evi++], ctx->st));
        }
        g.step = t + 1;
        g.m_x = plan.st[t & 1];
        g.s_x = plan.st[t & 1] + E;
        g.m_out = plan.st[(t + 1) & 1];
        g.s_out = plan.st[(t + 1) & 1] + E;
        const bool more = t + 1 < H;
        const int tail = GF_ASSEMBLE | GF_PROPAGATE | GF_TRAJ | (more ? (rbf_l ? GF_RBF_PRE : GF_POLICY) : 0);
        if (ctx->nranks == 1 && !ctx->comm) {
            g.flags = GF_PACK | tail;
        } else {
            g.flags = GF_PACK;
            launch_glue(ctx->st, g);
            if (int r = all_gather_segments(ctx, s)) return r;
            g.flags = tail;
        }
        launch_glue(ctx->st, g, rew && s.wk.PL == 0);   // (a rank without pairs keeps the reward in the glue launch)
        if (rbf_l && more) {  // the policy stage reads the NEW state
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
    const bool peer = peer_rollout_applies(ctx, plan, H);   // no collective nodes: captured like a single-rank rollout
    const bool sharded = (ctx->nranks != 1 || ctx->comm) && !peer;
    if (ctx->time_pairs) {   // measurement mode (pilco_set_pair_timing): eager, an event pair around every O(N^2) launch
        while (ctx->pair_events.size() < (size_t)2 * std::max(H, 1)) {
            hipEvent_t e;
            HIPCHK(hipEventCreate(&e));
            ctx->pair_events.push_back(e);
        }
        ctx->timed_pairs = 0;
        if (int r = enqueue_rollout(ctx, plan, H, &ctx->pair_events)) return r;
        ctx->timed_pairs = (s.wk.PL > 0) ? H : 0;
        return PILCO_OK;
    }
    if (!ctx->use_graph || (sharded && (!ctx->comm || ctx->graph_rccl_failed)) || (ctx->dbg && !getenv("PILCO_DBG_GRAPH")))
        return enqueue_rollout(ctx, plan, H, nullptr);
    const GlueArgs& g = plan.g;
    std::vector<unsigned long long> key = {
        (unsigned long long)H, (unsigned long long)g.pol_kind, (unsigned long long)g.n_rewards, (unsigned long long)g.squash,
        (unsigned long long)ctx->variant, (unsigned long long)ctx->fused + 2ull * (unsigned long long)ctx->fuse_small + 4ull * (unsigned long long)(ctx->share_cu != 0), (unsigned long long)(uintptr_t)plan.st[0], (unsigned long long)(uintptr_t)g.s1,
        (unsigned long long)(uintptr_t)g.traj, (unsigned long long)(uintptr_t)g.tape, (unsigned long long)(uintptr_t)g.W, (unsigned long long)(uintptr_t)g.maxact,
        (unsigned long long)(uintptr_t)s.w_part.p, (unsigned long long)(uintptr_t)s.w_At.p, (unsigned long long)(uintptr_t)s.w_Wt.p,
        (unsigned long long)(uintptr_t)s.w_small.p, (unsigned long long)(uintptr_t)s.w_gath.p, (unsigned long long)(uintptr_t)s.w_out.p,
        (unsigned long long)(uintptr_t)s.w_in.p, (unsigned long long)(uintptr_t)s.beta.p,
        (unsigned long long)(uintptr_t)s.iK.p, (unsigned long long)s.iK_null, (unsigned long long)(uintptr_t)s.Xt.p,
        (unsigned long long)(uintptr_t)s.Zt.p, (unsigned long long)(uintptr_t)s.ls.p, (unsigned long long)s.n,
        (unsigned long long)s.wk.sk_waves, (unsigned long long)s.wk.NT, (unsigned long long)s.wk.NCH, (unsigned long long)s.wk.NCHM, (unsigned long long)s.wk.KP, (unsigned long long)s.wk.abl,
        (unsigned long long)(uintptr_t)ctx->slot[1].w_part.p, (unsigned long long)(uintptr_t)ctx->slot[1].w_At.p,
        (unsigned long long)(uintptr_t)ctx->slot[1].beta.p, (unsigned long long)(uintptr_t)ctx->slot[1].Xt.p,
        (unsigned long long)ctx->slot[1].n,
        (unsigned long long)ctx->slot[1].wk.sk_waves, (unsigned long long)(uintptr_t)ctx->slot[1].w_small.p,
        (unsigned long long)(uintptr_t)ctx->slot[1].w_in.p, (unsigned long long)(uintptr_t)ctx->slot[1].ls.p,
        (unsigned long long)(peer ? 1 : 0), (unsigned long long)(uintptr_t)ctx->xq.local, (unsigned long long)ctx->nranks, (unsigned long long)ctx->rank,
        (unsigned long long)(g.pol_inline && ctx->inline_policy ? 1 : 0), (unsigned long long)(uintptr_t)ctx->slot[1].var.p,
        (unsigned long long)(uintptr_t)plan.jrec, (unsigned long long)plan.jstride, (unsigned long long)(uintptr_t)s.jac_rowmom.p,
        (unsigned long long)(uintptr_t)s.jac_cpart.p, (unsigned long long)(uintptr_t)s.jac_part.p, (unsigned long long)(uintptr_t)s.jac_head.p,
        (unsigned long long)(uintptr_t)s.jac_np.p};
    for (int i = 0; i < g.n_rewards; ++i) {
        key.push_back((unsigned long long)g.rw[i].kind);
        key.push_back((unsigned long long)(long long)g.rw[i].rank);
        key.push_back((unsigned long long)(uintptr_t)g.rw[i].W);
        unsigned long long cbits;
        memcpy(&cbits, &g.rw[i].coef, sizeof(cbits));
        key.push_back(cbits);
    }
    // a few instantiated graphs are kept (value rollouts and tape rollouts of an optimiser alternate): find this key
    for (size_t i = 0; i < ctx->graph_cache.size(); ++i)
        if (ctx->graph_cache[i].first == key) {
            if (i != 0) std::swap(ctx->graph_cache[i], ctx->graph_cache[0]);   // most recently used first
            ctx->graph = ctx->graph_cache[0].second;
            ctx->graph_key = key;
            if (peer) {
                if (int r = xq_begin(ctx)) return r;
                ctx->xq.epoch += (unsigned long long)H;
            }
            HIPCHK(hipGraphLaunch(ctx->graph, ctx->st));
            return PILCO_OK;
        }
    {
        ctx->graph = nullptr;
        if (ctx->graph_cache.size() >= 4) {   // evict the least recently used
            (void)hipGraphExecDestroy(ctx->graph_cache.back().second);
            ctx->graph_cache.pop_back();
        }
        // warm the per-kernel one-time host configuration outside the capture (a one-step rollout: with the peer
        // exchange attached every rank runs it, so it is a complete exchange of its own epoch)
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
        ctx->graph_cache.insert(ctx->graph_cache.begin(), std::make_pair(key, ctx->graph));
        // the warm-up rollout above overwrote the initial state: the caller re-uploads it (see callers)
        return -1;
    }
}


extern "C" {

}  // extern "C"

// pilco_rollout in two halves, so that several rollouts (the lanes of pilco_rollout_batch) can be in flight at once:
// rollout_begin enqueues everything -- upload of (m0, S0), the rollout, the downloads into pinned memory -- and returns;
// rollout_end waits for the stream and hands the results out.
struct RolloutCall {
    RolloutPlan plan;
    size_t nst = 0;
    double* pin_out = nullptr;
    bool peer = false;
};
static int rollout_begin(pilco_ctx* ctx, const pilco_policy* policy, const pilco_reward_term* rewards, int n_rewards, const double* m0,
                         const double* S0, int H, double* traj, RolloutCall& rc) {
    HIPCHK(hipSetDevice(ctx->device));
    RolloutPlan& plan = rc.plan;
    if (int r = setup_rollout(ctx, policy, rewards, n_rewards, H, traj != nullptr, plan)) return r;
    const int E = plan.E;
    // one pinned staging area: [m0 | S0] up in ONE asynchronous copy, [m_H | S_H] and the reward down in two, one host
    // synchronisation at the end (pageable buffers would cost a blocking staging copy per call)
    const size_t nst = (size_t)E + (size_t)E * E;
    if (ctx->pin_io_cap < 2 * nst + 8) {
        if (ctx->pin_io) (void)hipHostFree(ctx->pin_io);
        ctx->pin_io = nullptr;
        ctx->pin_io_cap = 0;
        HIPCHK(hipHostMalloc((void**)&ctx->pin_io, sizeof(double) * (2 * nst + 8), hipHostMallocDefault));
        ctx->pin_io_cap = 2 * nst + 8;
    }
    double* pin_in = ctx->pin_io;
    double* pin_out = ctx->pin_io + nst;
    memcpy(pin_in, m0, sizeof(double) * E);
    memcpy(pin_in + E, S0, sizeof(double) * E * E);
    for (int attempt = 0; attempt < 2; ++attempt) {
        HIPCHK(hipMemcpyAsync(plan.st[0], pin_in, sizeof(double) * nst, hipMemcpyHostToDevice, ctx->st));
        const int r = run_rollout(ctx, plan, H);
        if (r == -1) continue;  // graph was just (re)captured: upload the state again and replay it
        if (r != PILCO_OK) return r;
        break;
    }
    HIPCHK(hipMemcpyAsync(pin_out, plan.st[H & 1], sizeof(double) * nst, hipMemcpyDeviceToHost, ctx->st));
    HIPCHK(hipMemcpyAsync(pin_out + nst, plan.g.reward, sizeof(double), hipMemcpyDeviceToHost, ctx->st));
    if (traj)
        HIPCHK(hipMemcpyAsync(traj, ctx->traj.p, sizeof(double) * (size_t)(H + 1) * (E + E * E), hipMemcpyDeviceToHost, ctx->st));
    rc.peer = peer_rollout_applies(ctx, plan, H);
    if (rc.peer) HIPCHK(hipMemcpyAsync(ctx->xq.pin + 128, ctx->xq.local + 1, sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->st));
    rc.nst = nst;
    rc.pin_out = pin_out;
    return PILCO_OK;
}
static int rollout_end(pilco_ctx* ctx, RolloutCall& rc, double* mH, double* SH, double* reward) {
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipStreamSynchronize(ctx->st));
    HIPCHK(hipGetLastError());
    if (rc.peer && ctx->xq.pin[128] != 0ULL) {   // a flag wait gave up: some rank never delivered that exchange
        const unsigned long long ep = ctx->xq.pin[128];
        (void)hipMemsetAsync(ctx->xq.local + 1, 0, sizeof(unsigned long long), ctx->st);
        return fail(ctx, PILCO_E_STATE, "rollout: peer exchange " + std::to_string(ep) + " timed out on rank " + std::to_string(ctx->rank) +
                                            " (the ranks must make the same sequence of rollout calls)");
    }
    const int E = rc.plan.E;
    memcpy(mH, rc.pin_out, sizeof(double) * E);
    memcpy(SH, rc.pin_out + E, sizeof(double) * E * E);
    *reward = rc.pin_out[rc.nst];
    return PILCO_OK;
}

// A lane of pilco_rollout_batch: a context of its own whose dynamics slot borrows the parent's model.  (Re)pointed at the
// parent's current buffers before every batch; a changed geometry drops the lane's workspace and graphs.
static int lane_sync_model(pilco_ctx* parent, pilco_ctx* lane) {
    const Slot& p = parent->slot[0];
    Slot& l = lane->slot[0];
    const bool same = l.N == p.N && l.D == p.D && l.E == p.E && l.M == p.M && l.Npad == p.Npad && l.n == p.n && l.npad == p.npad &&
                      l.iK_null == p.iK_null && l.Xt.p == p.Xt.p && l.Zt.p == p.Zt.p && l.beta.p == p.beta.p && l.iK.p == p.iK.p &&
                      l.ls.p == p.ls.p && l.var.p == p.var.p;
    l.N = p.N; l.D = p.D; l.E = p.E; l.M = p.M; l.Npad = p.Npad; l.n = p.n; l.npad = p.npad;
    l.has_data = p.has_data; l.has_hyp = p.has_hyp; l.factor_valid = p.factor_valid; l.user_factors = p.user_factors;
    l.iK_null = p.iK_null; l.ignore_iK = p.ignore_iK;
    l.shW = p.shW; l.shEL = p.shEL; l.shOwn = p.shOwn; l.shRank = p.shRank; l.beta_complete = p.beta_complete;
    l.Xt.borrow(p.Xt); l.Yt.borrow(p.Yt); l.Zt.borrow(p.Zt); l.ls.borrow(p.ls); l.var.borrow(p.var); l.noise.borrow(p.noise);
    l.beta.borrow(p.beta); l.iK.borrow(p.iK);
    lane->variant = parent->variant;
    lane->fused = parent->fused;
    lane->use_graph = parent->use_graph;
    lane->fuse_small = parent->fuse_small;
    lane->inline_policy = parent->inline_policy;
    lane->grad_mode = parent->grad_mode;
    lane->dev_chain = parent->dev_chain;
    if (!same) {
        l.wk_valid = false;
        for (auto& ge : lane->graph_cache) (void)hipGraphExecDestroy(ge.second);
        lane->graph_cache.clear();
        lane->graph = nullptr;
    }
    return PILCO_OK;
}

// The B lanes of a batch call: lane 0 is the context itself, lanes 1.. are created on first use and pointed at its model.
int rollout_lanes(pilco_ctx* ctx, int B, std::vector<pilco_ctx*>& lane, const char* who) {
    while ((int)ctx->lanes.size() < B - 1) {
        pilco_ctx* l = nullptr;
        if (int r = pilco_ctx_create(ctx->device, &l)) return fail(ctx, r, std::string(who) + ": could not create a lane context");
        l->is_lane = true;
        ctx->lanes.push_back(l);
    }
    lane.resize((size_t)B);
    for (int i = 0; i < B; ++i) {
        lane[i] = i == 0 ? ctx : ctx->lanes[i - 1];
        lane[i]->share_cu = B > 1 ? 1 : 0;   // (the caller resets lane 0's when the batch is over: rollout_lanes_done)
        if (i > 0) {
            HIPCHK(hipStreamSynchronize(lane[i]->st));
            if (int r = lane_sync_model(ctx, lane[i])) return r;
        }
    }
    // the parent's model must be complete in memory before another stream reads it
    HIPCHK(hipStreamSynchronize(ctx->st));
    return PILCO_OK;
}

void rollout_lanes_done(pilco_ctx* ctx) { ctx->share_cu = 0; }

extern "C" {

int pilco_rollout(pilco_ctx* ctx, const pilco_policy* policy, const pilco_reward_term* rewards, int n_rewards,
                  const double* m0, const double* S0, int H, double* mH, double* SH, double* reward, double* traj) {
    if (!ctx) return PILCO_E_SHAPE;
    if (!m0 || !S0 || !mH || !SH || !reward || H < 0) return fail(ctx, PILCO_E_SHAPE, "rollout: bad arguments");
    RolloutCall rc;
    if (int r = rollout_begin(ctx, policy, rewards, n_rewards, m0, S0, H, traj, rc)) return r;
    return rollout_end(ctx, rc, mH, SH, reward);
}

// B independent rollouts of ONE model in flight together (multi-start policy search, several initial states; the restart
// loop of pilco.py:96-110 evaluates its candidates one after the other).  Lane 0 is this context; lanes 1..B-1 are
// contexts of their own -- stream, per-step workspace, state, policy parameters, cached graph -- that borrow this context's
// model (X, hyper-parameters, beta, iK are not copied).  All B graph replays are enqueued before the first wait, so the
// serial head of one lane's step (a chain of latencies that leaves the chip idle) runs under the pair kernels of the
// others.  Every lane runs exactly the launch sequence of pilco_rollout: its result is bit-identical to its solo run.
int pilco_rollout_batch(pilco_ctx* ctx, int B, const pilco_policy* policies, const pilco_reward_term* rewards, int n_rewards,
                        const double* m0, const double* S0, int H, double* mH, double* SH, double* reward) {
    if (!ctx) return PILCO_E_SHAPE;
    if (B <= 0 || B > 64 || !policies || !m0 || !S0 || !mH || !SH || !reward || H < 0) return fail(ctx, PILCO_E_SHAPE, "rollout_batch: bad arguments");
    if (ctx->nranks != 1 || ctx->comm) return fail(ctx, PILCO_E_STATE, "rollout_batch: single rank only (shard OR batch)");
    for (int i = 0; i < B; ++i)
        if (policies[i].kind == PILCO_POLICY_RBF) return fail(ctx, PILCO_E_SHAPE, "rollout_batch: RbfController lanes are not supported (one policy GP slot per context)");
    HIPCHK(hipSetDevice(ctx->device));
    const Slot& s = ctx->slot[0];
    if (!s.factor_valid) return fail(ctx, PILCO_E_STATE, "rollout_batch: dynamics model has no current factorisation");
    const int E = s.E;
    const size_t nst = (size_t)E + (size_t)E * E;
    std::vector<RolloutCall> rc((size_t)B);
    std::vector<pilco_ctx*> lane;
    LanesGuard lanes_guard{ctx};
    if (int r = rollout_lanes(ctx, B, lane, "rollout_batch")) return r;
    int rc_err = PILCO_OK;
    int begun = 0;
    for (int i = 0; i < B; ++i, ++begun) {
        rc_err = rollout_begin(lane[i], &policies[i], rewards, n_rewards, m0 + (size_t)i * E, S0 + (size_t)i * E * E, H, nullptr, rc[i]);
        if (rc_err) {
            if (i > 0) ctx->err = "lane " + std::to_string(i) + ": " + lane[i]->err;
            break;
        }
    }
    for (int i = 0; i < begun; ++i) {
        const int r = rollout_end(lane[i], rc[i], mH + (size_t)i * E, SH + (size_t)i * E * E, reward + i);
        if (r && !rc_err) {
            rc_err = r;
            if (i > 0) ctx->err = "lane " + std::to_string(i) + ": " + lane[i]->err;
        }
    }
    (void)nst;
    return rc_err;
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
    for (int attempt = 0; attempt < 2; ++attempt) {   // replayed as a hipGraph like pilco_rollout (the tape pointer is part of the graph key)
        HIPCHK(hipMemcpyAsync(plan.st[0], m0, sizeof(double) * E, hipMemcpyHostToDevice, ctx->st));
        HIPCHK(hipMemcpyAsync(plan.st[0] + E, S0, sizeof(double) * E * E, hipMemcpyHostToDevice, ctx->st));
        const int r = run_rollout(ctx, plan, H);
        if (r == -1) continue;
        if (r != PILCO_OK) return r;
        break;
    }
    HIPCHK(hipMemcpyAsync(mH, plan.st[H & 1], sizeof(double) * E, hipMemcpyDeviceToHost, ctx->st));
    HIPCHK(hipMemcpyAsync(SH, plan.st[H & 1] + E, sizeof(double) * E * E, hipMemcpyDeviceToHost, ctx->st));
    HIPCHK(hipMemcpyAsync(reward, plan.g.reward, sizeof(double), hipMemcpyDeviceToHost, ctx->st));
    if (traj) HIPCHK(hipMemcpyAsync(traj, ctx->traj.p, sizeof(double) * (size_t)(H + 1) * (E + E * E), hipMemcpyDeviceToHost, ctx->st));
    if (H > 0) HIPCHK(hipMemcpyAsync(tape, ctx->tape.p, sizeof(double) * (size_t)H * TS, hipMemcpyDeviceToHost, ctx->st));
    HIPCHK(hipStreamSynchronize(ctx->st));
    HIPCHK(hipGetLastError());
    return PILCO_OK;
}

}  // extern "C"

// Value-and-gradient rollout, forward half: pilco_rollout_tape with every dynamics step run as the reverse sweep
// (launch_mm_jac), so that the step's value and its Jacobian records come out of ONE O(N^2) pass; trajectory, tape and
// records land in pinned host memory for the host-side reverse sweep (grad.hip).  Single rank; D <= 14 (wider inputs: PILCO_JAC_TOO_LARGE, the caller
// falls back to the plain tape + per-step device adjoint, which has the forward path's D <= 32).
int rollout_jtape(pilco_ctx* ctx, const pilco_policy* policy, const pilco_reward_term* rewards, int n_rewards, const double* m0,
                  const double* S0, int H, double* reward, const double** traj, const double** tape, const double** jrec, size_t* jstride,
                  const double** reward_later, JtapeDev* dev) {
    HIPCHK(hipSetDevice(ctx->device));
    // Several ranks (round 3): every rank sweeps ITS pairs (k_mm_bwd_pair is per-pair independent; the mean-part records of
    // all E outputs are cheap and computed everywhere), the per-step exchange of the forward chain is the sharded
    // rollout's own (RCCL all-gather, or host-mediated inside pilco_rollout_grad_group), and after the batched finish the
    // per-pair records are all-gathered ONCE -- records, not sums: the host sweep below adds them in the single-rank order.
    const int W = ctx->nranks;
    const bool sharded = (W != 1 || ctx->comm);
    if (sharded && !ctx->comm && !ctx->group)
        return fail(ctx, PILCO_E_STATE, "rollout_grad: a sharded context needs a communicator (pilco_comm_init) or pilco_rollout_grad_group");
    RolloutPlan plan;
    if (int r = setup_rollout(ctx, policy, rewards, n_rewards, H, true, plan)) return r;
    Slot& s = ctx->slot[0];
    const int E = plan.E, D = plan.D, P = s.wk.PL, npad = s.npad;
    if (D > 14) return PILCO_JAC_TOO_LARGE;   // third-moment records and their LDS working set are sized for D <= 14
    const size_t TS = (size_t)D + D * D + (size_t)E * D + E + (size_t)E * E + (size_t)D * E;
    const size_t JS = mm_jac_rec_size(D, E, P), NTJ = (size_t)(H + 1) * (E + (size_t)E * E);
    // every step keeps its own sweep output until the batched finish (nothing on the chain waits for a buffer): at C2u
    // 29 MB per step -- HBM is 288 GB; a rollout that would need more than PILCO_JAC_GB (default 32) falls back
    const size_t Hn = (size_t)std::max(H, 1);
    const int ELc = s.wk.EL;   // owned outputs = diagonal pairs held here (E on one rank)
    const size_t per_step = mm_jac_rowmom_size(npad, P) + mm_jac_cpart_size(npad, P, ELc) + mm_jac_head_size(D, E, P) + mm_jac_part_size(D, E, P, npad);
    double cap_gb = 32.0;
    if (const char* ev = getenv("PILCO_JAC_GB")) cap_gb = atof(ev);
    if ((double)per_step * 8.0 * (double)Hn > cap_gb * 1e9) return PILCO_JAC_TOO_LARGE;
    ENSURE(s.jac_rowmom, Hn * mm_jac_rowmom_size(npad, P));
    ENSURE(s.jac_cpart, Hn * mm_jac_cpart_size(npad, P, ELc));
    ENSURE(s.jac_head, Hn * mm_jac_head_size(D, E, P));
    ENSURE(s.jac_part, Hn * mm_jac_part_size(D, E, P, npad));
    ENSURE(s.jac_np, (size_t)2 * std::max(P, 1) * mm_jac_nt(npad, s.wk.P));
    ENSURE(ctx->tape, std::max<size_t>(1, (size_t)H * TS));
    ENSURE(ctx->jrec, std::max<size_t>(1, (size_t)H * JS));
    // sharded: the host sweep reads GLOBAL records [P_all pair records | E output records] per step, assembled on the host
    // from every rank's pair records (all-gathered) and this rank's own output records
    const int NT2 = D * (D + 1) / 2, recp = 1 + D + NT2, Pall = E * (E + 1) / 2, PLcap = (Pall + W - 1) / W;
    const size_t reco = (size_t)D + NT2 + (size_t)D * D + (size_t)D * NT2;
    const size_t JSg = sharded ? (size_t)Pall * recp + (size_t)E * reco : JS;
    // one rank's block: per step its pair records (padded to PLcap) and the E output records (every rank WITH pairs computes
    // them all; a rank without pairs runs no sweep at all, so the readers take them from rank 0, which always has pairs)
    const size_t gstep = (size_t)PLcap * recp + (size_t)E * reco;
    const size_t gblk = (size_t)std::max(H, 1) * gstep;
    const size_t SEd = (size_t)E + (size_t)E * E;
    if (dev) {
        dev->n_seeds = (size_t)(H + 1) * SEd;
        dev->n_out = (size_t)plan.U * E + plan.U + 1 + SEd + 2;
    }
    const size_t need = dev ? NTJ + 8 + dev->n_seeds + dev->n_out + 8
                            : NTJ + (size_t)H * TS + (size_t)H * JSg + 8 + (sharded ? (size_t)W * gblk : 0);
    if (ctx->jpin_cap < need) {
        if (ctx->jpin) (void)hipHostFree(ctx->jpin);
        ctx->jpin = nullptr;
        ctx->jpin_cap = 0;
        HIPCHK(hipHostMalloc((void**)&ctx->jpin, sizeof(double) * need, hipHostMallocDefault));
        ctx->jpin_cap = need;
    }
    double* h_traj = ctx->jpin;
    double* h_tape = h_traj + NTJ;
    double* h_jrec = h_tape + (size_t)H * TS;
    plan.g.tape = ctx->tape.p;
    // one rank: k_mm_jac_fin writes the records straight into the pinned host buffer (device-visible): their 4.3 MB cross
    // PCIe while the kernel runs instead of as four copies that hold the stream between the chunks of the finish
    const bool jdirect = !dev && !sharded && getenv("PILCO_JAC_COPY") == nullptr;
    plan.jrec = jdirect ? h_jrec : ctx->jrec.p;
    plan.jstride = JS;
    plan.jsmall = sharded ? 0 : jac_small_chunks(ctx, plan, H);
    double* h_misc = dev ? h_traj + NTJ : h_jrec + (size_t)H * JSg;
    double* h_all = h_misc + 8;                          // sharded: [W][H][PLcap * recp | E * reco]
    for (int attempt = 0; attempt < 2; ++attempt) {
        HIPCHK(hipMemcpyAsync(plan.st[0], m0, sizeof(double) * E, hipMemcpyHostToDevice, ctx->st));
        HIPCHK(hipMemcpyAsync(plan.st[0] + E, S0, sizeof(double) * E * E, hipMemcpyHostToDevice, ctx->st));
        const int r = run_rollout(ctx, plan, H);
        if (r == -1) continue;
        if (r != PILCO_OK) return r;
        break;
    }
    if (!dev) HIPCHK(hipMemcpyAsync(h_misc, plan.g.reward, sizeof(double), hipMemcpyDeviceToHost, ctx->st));
    if (dev) {
        // ---- the reverse chain on the device (rev.hip): nothing but the reward, the gradient -- and, for a caller with
        // cotangent seeds, the trajectory -- crosses to the host
        if (!ctx->jwait_ev[0]) HIPCHK(hipEventCreateWithFlags(&ctx->jwait_ev[0], hipEventDisableTiming));
        if (dev->seeds) {
            HIPCHK(hipMemcpyAsync(h_traj, ctx->traj.p, sizeof(double) * NTJ, hipMemcpyDeviceToHost, ctx->st));
            HIPCHK(hipEventRecord(ctx->jwait_ev[0], ctx->st));   // the host turns the trajectory into seeds while the finish runs
        }
        RevArgs& ra = dev->ra;
        ra = RevArgs{};
        ra.E = E; ra.U = plan.U; ra.D = D; ra.H = H; ra.P = Pall;
        ra.W = 1; ra.gblk = 0; ra.gstep = (long)JS; ra.out_off = (long)P * recp;
        ra.jrec = ctx->jrec.p;
        ENSURE(ctx->revloc, std::max<size_t>(1, (size_t)H * rev_loc_doubles(E, plan.U)));
        const RevLocalArgs rl = rev_local_args(plan.g.n_rewards, plan.g.rw, E, plan.U, ctx->traj.p, plan.g.W, plan.g.b, plan.g.maxact, ctx->revloc.p);
        if (H > 0) {
            jac_finish_range(ctx, plan, 0, H, &rl);   // (the trajectory-only quantities of the chain ride in its last launch)
            if (sharded) {   // every rank's pair records, all-gathered ONCE; the chain reads them where they land
                ENSURE(ctx->jgath, (size_t)(W + 1) * gblk);
                double* own = ctx->jgath.p + (size_t)W * gblk;
                HIPCHK(hipMemsetAsync(own, 0, sizeof(double) * gblk, ctx->st));
                if (P > 0) {
                    HIPCHK(hipMemcpy2DAsync(own, sizeof(double) * gstep, ctx->jrec.p, sizeof(double) * JS, sizeof(double) * P * recp, (size_t)H,
                                            hipMemcpyDeviceToDevice, ctx->st));
                    HIPCHK(hipMemcpy2DAsync(own + (size_t)PLcap * recp, sizeof(double) * gstep, ctx->jrec.p + (size_t)P * recp, sizeof(double) * JS,
                                            sizeof(double) * E * reco, (size_t)H, hipMemcpyDeviceToDevice, ctx->st));
                }
                if (ctx->comm) {
                    ncclResult_t r = ncclAllGather(own, ctx->jgath.p, gblk, ncclDouble, ctx->comm, ctx->st);
                    if (r != ncclSuccess) return fail(ctx, PILCO_E_RCCL, std::string("ncclAllGather(jacobian records): ") + ncclGetErrorString(r));
                } else {   // contexts of one process (pilco_rollout_grad_group): take the peers' blocks between two host barriers
                    HIPCHK(hipStreamSynchronize(ctx->st));
                    std::shared_ptr<PeerGroup> grp = ctx->group;
                    if (!grp->arrive_and_wait()) return fail(ctx, PILCO_E_STATE, "rollout_grad: another rank of the group failed");
                    // (on THIS context's stream: a device-to-device hipMemcpy is ordered on the null stream only and need not have
                    // finished when it returns -- the chain below, on a non-blocking stream, read blocks that were still being
                    // copied once in ten runs; the peers may reuse their blocks after the second barrier, so the copies are
                    // waited for in front of it)
                    for (int j = 0; j < W; ++j) {
                        pilco_ctx* pj = grp->ctxs[j];
                        HIPCHK(hipMemcpyAsync(ctx->jgath.p + (size_t)j * gblk, pj->jgath.p + (size_t)W * gblk, sizeof(double) * gblk, hipMemcpyDeviceToDevice,
                                              ctx->st));
                    }
                    HIPCHK(hipStreamSynchronize(ctx->st));
                    if (!grp->arrive_and_wait()) return fail(ctx, PILCO_E_STATE, "rollout_grad: another rank of the group failed");
                }
                ra.jrec = ctx->jgath.p;
                ra.W = W; ra.gblk = (long)gblk; ra.gstep = (long)gstep; ra.out_off = (long)PLcap * recp;
            }
        }
        ra.traj = ctx->traj.p;
        ra.tape = ctx->tape.p;
        ra.TS = (long)TS;
        ra.loc = ctx->revloc.p;
        ENSURE(ctx->revmat, std::max<size_t>(1, (size_t)H * rev_mat_doubles(E, plan.U, D)));
        ra.amat = ctx->revmat.p;
        ra.seeds = nullptr;
        ra.reward_dev = plan.g.reward;
        ra.Wp = plan.g.W;
        dev->h_seeds = h_misc + 8;
        double* h_out = dev->h_seeds + dev->n_seeds;
        ra.out = h_out;
        dev->h_out = h_out;
        dev->h_traj = h_traj;
        dev->h_reward = h_out + ((size_t)plan.U * E + plan.U) + 1 + (size_t)(E + Pall);   // (written by the chain kernel)
        if (!dev->seeds) launch_rev_chain(ctx->st, ra);
        HIPCHK(hipGetLastError());
        return PILCO_OK;
    }
    HIPCHK(hipMemcpyAsync(h_traj, ctx->traj.p, sizeof(double) * NTJ, hipMemcpyDeviceToHost, ctx->st));
    // the records come down in chunks, LAST steps first, an event behind each: the host's reverse sweep starts on the
    // last steps while the earlier ones are still on their way (rollout_jtape_wait)
    ctx->jwait_from = H;
    ctx->jwait_next = 0;
    ctx->jwait_n = 0;
    if (H > 0 && sharded) {
        jac_finish_range(ctx, plan, 0, H);
        HIPCHK(hipMemcpyAsync(h_tape, ctx->tape.p, sizeof(double) * (size_t)H * TS, hipMemcpyDeviceToHost, ctx->st));
        ENSURE(ctx->jgath, (size_t)(W + 1) * gblk);
        double* own = ctx->jgath.p + (size_t)W * gblk;
        HIPCHK(hipMemsetAsync(own, 0, sizeof(double) * gblk, ctx->st));
        if (P > 0) {   // this rank's records of every step, compacted: [H][PLcap pair records | E output records]
            HIPCHK(hipMemcpy2DAsync(own, sizeof(double) * gstep, ctx->jrec.p, sizeof(double) * JS, sizeof(double) * P * recp, (size_t)H,
                                    hipMemcpyDeviceToDevice, ctx->st));
            HIPCHK(hipMemcpy2DAsync(own + (size_t)PLcap * recp, sizeof(double) * gstep, ctx->jrec.p + (size_t)P * recp, sizeof(double) * JS,
                                    sizeof(double) * E * reco, (size_t)H, hipMemcpyDeviceToDevice, ctx->st));
        }
        if (ctx->comm) {
            ncclResult_t r = ncclAllGather(own, ctx->jgath.p, gblk, ncclDouble, ctx->comm, ctx->st);
            if (r != ncclSuccess) return fail(ctx, PILCO_E_RCCL, std::string("ncclAllGather(jacobian records): ") + ncclGetErrorString(r));
            HIPCHK(hipMemcpyAsync(h_all, ctx->jgath.p, sizeof(double) * (size_t)W * gblk, hipMemcpyDeviceToHost, ctx->st));
            HIPCHK(hipStreamSynchronize(ctx->st));
        } else {   // contexts of one process (pilco_rollout_grad_group): read the peers' blocks between two host barriers
            HIPCHK(hipStreamSynchronize(ctx->st));
            std::shared_ptr<PeerGroup> grp = ctx->group;
            if (!grp->arrive_and_wait()) return fail(ctx, PILCO_E_STATE, "rollout_grad: another rank of the group failed");
            for (int j = 0; j < W; ++j) {
                pilco_ctx* pj = grp->ctxs[j];
                HIPCHK(hipMemcpy(h_all + (size_t)j * gblk, pj->jgath.p + (size_t)W * gblk, sizeof(double) * gblk, hipMemcpyDeviceToHost));
            }
            if (!grp->arrive_and_wait()) return fail(ctx, PILCO_E_STATE, "rollout_grad: another rank of the group failed");
        }
        HIPCHK(hipGetLastError());
        for (int t = 0; t < H; ++t) {   // global record of step t: pair kk of the dealing order lives on rank kk % W as its pair kk / W
            double* dst = h_jrec + (size_t)t * JSg;
            for (int kk = 0; kk < Pall; ++kk)
                memcpy(dst + (size_t)kk * recp, h_all + (size_t)(kk % W) * gblk + (size_t)t * gstep + (size_t)(kk / W) * recp, sizeof(double) * recp);
            memcpy(dst + (size_t)Pall * recp, h_all + (size_t)t * gstep + (size_t)PLcap * recp, sizeof(double) * E * reco);   // rank 0's
        }
        ctx->jwait_n = 0;
        ctx->jwait_from = 0;
    } else if (H > 0) {
        HIPCHK(hipMemcpyAsync(h_tape, ctx->tape.p, sizeof(double) * (size_t)H * TS, hipMemcpyDeviceToHost, ctx->st));
        // chunks in the order the reverse sweep consumes them, SHRINKING towards step 0.  Measured at C2u (gpurun_out/r03/
        // chunks*.log): the device finishes a step's records in ~11 us, the host sweeps one in ~9 us, and every chunk costs
        // both sides a fixed ~40-60 us (two launches, a copy, an event wait) -- four chunks of 16/12/8/4 fortieths: 6.03 ->
        // 5.97 ms; six chunks (8,8,8,8,4,4): 6.14 ms; finishing the early chunks on a second stream WHILE the chain runs:
        // 6.09 ms with one fork, 7.4 ms with three (the chain's kernels lose what the finish gains)
        static const int parts[4] = {16, 12, 8, 4};   // fortieths of H
        const int nch = std::min(H, 4);
        int t1 = H, used = 0;
        for (int k = 0; k < nch; ++k) {
            used += parts[k];
            const int t0 = (k == nch - 1) ? 0 : std::min(t1 - 1, std::max(0, H - (int)((long)used * H / 40)));   // steps [t0, t1), never empty
            jac_finish_range(ctx, plan, t0, t1);
            if (!jdirect)
                HIPCHK(hipMemcpyAsync(h_jrec + (size_t)t0 * JS, ctx->jrec.p + (size_t)t0 * JS, sizeof(double) * (size_t)(t1 - t0) * JS,
                                      hipMemcpyDeviceToHost, ctx->st));
            if (!ctx->jwait_ev[k]) HIPCHK(hipEventCreateWithFlags(&ctx->jwait_ev[k], hipEventDisableTiming));
            HIPCHK(hipEventRecord(ctx->jwait_ev[k], ctx->st));
            ctx->jwait_t0[k] = t0;
            t1 = t0;
        }
        ctx->jwait_n = nch;
        if (reward_later) {   // a lane of a batch: everything is enqueued, the caller waits when it gets to this lane
            *reward_later = h_misc;
            *traj = h_traj;
            *tape = h_tape;
            *jrec = h_jrec;
            *jstride = JSg;
            return PILCO_OK;
        }
        if (int r = rollout_jtape_wait(ctx, H - 1)) return r;   // reward, trajectory, tape and the last chunk are on the host
    } else {
        HIPCHK(hipStreamSynchronize(ctx->st));
    }
    HIPCHK(hipGetLastError());
    *reward = h_misc[0];
    *traj = h_traj;
    *tape = h_tape;
    *jrec = h_jrec;
    *jstride = JSg;
    return PILCO_OK;
}

// Device reverse chain, second half: with seeds, wait for the trajectory, let the caller turn it into cotangent seeds, upload
// them and launch the chain; then wait for the gradient.  dev.h_out / dev.h_reward are valid on PILCO_OK.
int rollout_jtape_dev_finish(pilco_ctx* ctx, JtapeDev& dev, int H, int E, jtape_seed_fn seed_fn, void* seed_user) {
    HIPCHK(hipSetDevice(ctx->device));
    if (dev.seeds) {
        HIPCHK(hipEventSynchronize(ctx->jwait_ev[0]));
        std::fill(dev.h_seeds, dev.h_seeds + dev.n_seeds, 0.0);
        seed_fn(seed_user, H, E, dev.h_traj, dev.h_seeds);
        for (size_t q = 0; q < dev.n_seeds; ++q)
            if (!std::isfinite(dev.h_seeds[q])) {
                (void)hipStreamSynchronize(ctx->st);
                return fail(ctx, PILCO_E_SHAPE, "rollout_grad: the seed callback returned a non-finite cotangent");
            }
        ENSURE(ctx->revseeds, dev.n_seeds);
        HIPCHK(hipMemcpyAsync(ctx->revseeds.p, dev.h_seeds, sizeof(double) * dev.n_seeds, hipMemcpyHostToDevice, ctx->st));
        dev.ra.seeds = ctx->revseeds.p;
        launch_rev_chain(ctx->st, dev.ra);
    }
    HIPCHK(hipStreamSynchronize(ctx->st));
    HIPCHK(hipGetLastError());
    const double status = dev.h_out[(size_t)dev.ra.U * dev.ra.E + dev.ra.U];
    if (status == 1.0) return fail(ctx, PILCO_E_NOT_PD, "rollout_grad: singular s + Lambda^2 or I + Lambda s");
    if (status != 0.0) return fail(ctx, PILCO_E_NOT_PD, "rollout_grad: singular I + S W in the reward");
    return PILCO_OK;
}

// Block until the records of step t (and everything enqueued before them) are on the host.
int rollout_jtape_wait(pilco_ctx* ctx, int t) {
    while (t < ctx->jwait_from && ctx->jwait_next < ctx->jwait_n) {
        const int k = ctx->jwait_next++;
        static const bool timing = getenv("PILCO_GRAD_TIMING") != nullptr;   // developer aid: how long the host waited for chunk k
        const auto w0 = std::chrono::steady_clock::now();
        HIPCHK(hipEventSynchronize(ctx->jwait_ev[k]));
        if (timing)
            fprintf(stderr, "[pilco grad] chunk %d (steps >= %d): waited %.3f ms (asked for step %d)\n", k, ctx->jwait_t0[k],
                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - w0).count(), t);
        ctx->jwait_from = ctx->jwait_t0[k];
    }
    return PILCO_OK;
}

