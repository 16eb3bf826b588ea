// Internal state of libpilco_hip.so shared by its host-side translation units (api.hip, rollout.hip, shard.hip,
// grad.hip): the context and GP-slot structs, the error / allocation macros and the cross-file helpers.
#pragma once
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <cmath>
#include <cstdint>

#include "moment.h"

using namespace pilco;

// A fixed launch sequence (factorisation, training objective) kept as an instantiated hipGraph: `key` names everything the
// enqueued launches depend on (buffer addresses, sizes); another key -> captured again.  failed: capture or instantiation
// did not work once -- the sequence stays on eager launches.
struct ChainGraph {
    std::vector<unsigned long long> key;
    hipGraphExec_t exec = nullptr;
    bool failed = false;
};

struct Slot {
    ChainGraph g_fact, g_fitc, g_fitc_nlml;   // exact / FITC factorisation, the FITC training objective (the exact objective = the factorisation's graph + two eager launches)
    int N = 0, D = 0, E = 0, M = 0;  // data size, input dim, outputs, inducing points (0 = exact)
    int Npad = 0;                    // padded N
    int n = 0, npad = 0;             // points the moment matching runs over (N or M) and padding
    bool has_data = false, has_hyp = false, factor_valid = false, user_factors = false, iK_null = false;
    bool ignore_iK = false;  // policy slot: RbfController evaluates with iK zeroed (controllers.py:116)
    DevBuf bwd_mom, bwd_cp, bwd_part, bwd_out, bwd_cnt; // reverse-pass scratch (bwd_cnt: the finished-workgroups counter)
    DevBuf jac_rowmom, jac_cpart, jac_head, jac_part, jac_np;   // Jacobian tape: per-step sweep outputs [H][..], N_ab tile partials
    DevBuf Xt, Yt, Zt, ls, var, noise;         // Yt: [E][Npad]
    DevBuf K, Linv, iK, beta, Tscr, vec;       // factorisation
    DevBuf ksplit_ws;                          // partial products of the split-K GEMMs of the FITC path (GemmDesc::split_ws)
    DevBuf Kmn, V2, Am, AmInv, iAt, G;         // FITC extras
    DevBuf ft_P, ft_T3, ft_Z;                  // FITC training objective (fitc_train.hip)
    // sharded factorisation (8e): this rank factorises only its outputs a = rank, rank + shW, ...; `own` holds their
    // hyper-parameters and targets compacted ([EL][D] | [EL] | [EL] | [EL][Npad]); beta is all-gathered afterwards
    DevBuf own;
    int shW = 1, shEL = 0, shOwn = 0, shRank = 0;   // ranks, outputs per rank (capacity), outputs owned, rank at factorisation time
    bool beta_complete = true;                      // false until the other ranks' beta rows have arrived
    // moment-matching workspace
    DevBuf w_in, w_At, w_Wt, w_small, w_part, w_gath, w_out;
    DevBuf w_fpart;   // one-launch step of small models: [2][PL][NCH][2] pair partials (the head reads one copy, writes the other)
    MMWork wk{};
    double* alt_isdet = nullptr;   // second copies of pair_isdet / mean_part: the fused head reads one set (previous step)
    double* alt_mean = nullptr;    // while its prep part writes the other
    bool wk_valid = false;
    int wk_variant = -1;
    std::vector<int> pair_owner;  // [P]
};


// Several contexts of ONE process exchanging their per-step segments through peer copies (pilco_rollout_group): a
// reusable host barrier shared by the group; `failed` releases everybody when one member hits an error.
struct PeerGroup {
    std::vector<struct pilco_ctx*> ctxs;
    std::mutex mu;
    std::condition_variable cv;
    int waiting = 0;
    unsigned long generation = 0;
    bool failed = false;
    bool arrive_and_wait() {   // false when the group has failed
        std::unique_lock<std::mutex> lk(mu);
        if (failed) return false;
        const unsigned long gen = generation;
        if (++waiting == (int)ctxs.size()) {
            waiting = 0;
            ++generation;
            cv.notify_all();
            return true;
        }
        cv.wait(lk, [&] { return generation != gen || failed; });
        return !failed;
    }
    void fail_all() {
        std::lock_guard<std::mutex> lk(mu);
        failed = true;
        cv.notify_all();
    }
};

// Peer exchange of the per-step segments without a collective call (GlueArgs::xq, moment.h): this rank's exchange area
// and every rank's area as mapped on this device (opened from hipIpc handles for ranks in other processes, plain
// pointers for contexts of this process).
struct PeerXch {
    int W = 0, cap = 0;                            // ranks, doubles per segment slot
    unsigned long long* local = nullptr;           // fine-grained device memory, xq_area_words(W, cap) words
    std::vector<unsigned long long*> mapped;       // [W]; mapped[rank] == local
    std::vector<char> opened;                      // mapped[j] came from hipIpcOpenMemHandle
    unsigned long long** d_peers = nullptr;        // device copy of `mapped`
    unsigned long long epoch = 0;                  // exchanges enqueued so far: the same number on every rank
    unsigned long long* pin = nullptr;             // pinned: ring [128] of epoch-base uploads; [128] error-word download
    unsigned ring = 0;
    bool ready = false, wait_kernel = false;       // wait_kernel: ranks share a GPU -> the flag wait gets a launch of its own
    // contexts of ONE process attached together (pilco_group_peer_attach) hold plain pointers into each other's areas:
    // the shared membership list lets a member that detaches (or is destroyed) take the others off the exchange first
    std::shared_ptr<std::vector<struct pilco_ctx*>> members;
};

constexpr int PILCO_DBG_WORDS = 16384;   // 64 phase slots | block / wave stamps (tools/)
struct pilco_ctx {
    int device = 0;
    hipStream_t st = nullptr;
    std::string err;
    int not_pd = -1;
    int variant = 0;
    bool variant_user = false;   // pilco_set_pair_kernel was called: the caller's choice stands (otherwise: stream-K on one rank, the tiled kernel -- whose sums do not depend on the rank count -- on several)
    int rank = 0, nranks = 1;
    ncclComm_t comm = nullptr;
    std::shared_ptr<PeerGroup> group;   // set only while pilco_rollout_group runs
    PeerXch xq;
    Slot slot[2];
    int* d_info = nullptr;
    DevBuf state;   // m_x, s_x, s1, reward, act_out, rew_out
    DevBuf params;  // policy + reward parameters
    DevBuf traj;
    DevBuf tape;
    DevBuf jrec;             // Jacobian tape: [H][mm_jac_rec_size] records of a value-and-gradient rollout
    DevBuf jgath;            // sharded value-and-gradient rollout: [W + 1][H][PLcap * recp] pair records (own block last) for the all-gather
    DevBuf revloc, revseeds, revmat; // device reverse chain (rev.hip): per-step trajectory-only quantities [H][rev_loc_doubles]; the caller's cotangent seeds [H + 1][E + E*E]
    bool dev_chain = true;   // LinearController gradients: the reverse chain runs on the device (false: the host chain of rounds 1-5, kept for the RbfController and as a cross-check)
    double* jpin = nullptr;  // pinned host copy of (traj | tape | jrec) for the host-side reverse sweep
    size_t jpin_cap = 0;
    hipEvent_t jwait_ev[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // behind the chunks of the records' download (last steps first)
    int jwait_t0[8] = {0, 0, 0, 0, 0, 0, 0, 0}, jwait_n = 0, jwait_next = 0, jwait_from = 0;   // first step of chunk k; steps >= jwait_from are on the host
    DevBuf selftest;
    DevBuf exp_tab;  // 2^(j/n), j = 0..n-1, n = mm_exp_table_size()
    unsigned long long* dbg = nullptr;   // [PILCO_DBG_WORDS] developer stamps (pilco_debug_timestamps allocates it)
    // cached hipGraph of one rollout (single-rank): replayed while the plan key is unchanged
    hipGraphExec_t graph = nullptr;
    std::vector<unsigned long long> graph_key;
    std::vector<std::pair<std::vector<unsigned long long>, hipGraphExec_t>> graph_cache;   // most recently used first (<= 4)
    bool use_graph = true;
    bool inline_policy = true;   // an RbfController small enough is evaluated inside the link (2 launches per step instead of 4)
    bool fused = true;   // fused head: the serial link of step t runs inside the prep launch of step t+1 (2 launches per step)
    bool fuse_small = true;   // ... and, for models of at most 256 points, the pair sums too: ONE launch per step (prep_device.h)
    bool graph_rccl_failed = false;
    int grad_mode = 1;   // pilco_rollout_grad*: 1 = Jacobian tape (one O(N^2) sweep per step), 0 = tape + per-step device adjoint
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::vector<hipEvent_t> pair_events;
    bool time_pairs = false;   // pilco_set_pair_timing: rollouts run eagerly with an event pair around every O(N^2) launch
    int timed_pairs = 0;       // launches bracketed by the last such rollout
    double* pin = nullptr;   // pinned host staging buffer of the reverse pass (truly asynchronous small copies)
    size_t pin_cap = 0;
    double* pin_io = nullptr;   // pinned staging of pilco_rollout's inputs / results (one copy each way, no pageable detours)
    size_t pin_io_cap = 0;
    const double* params_dev = nullptr;
    std::vector<double> params_host;   // what ctx->params holds on the device: identical parameters are not uploaded again
    // pilco_rollout_batch: lanes 1.. of a batch are contexts of their own (stream, workspace, state, graph cache) whose
    // dynamics slot BORROWS this context's model buffers; owned and destroyed by this context
    std::vector<pilco_ctx*> lanes;
    bool is_lane = false;
    int share_cu = 0;   // > 0 while this context runs as one of several lanes of a batch call (heads launch their two-per-CU build)
};

int fail(pilco_ctx* c, int code, const std::string& msg);
int agree_not_pd(pilco_ctx* ctx, int W, int bad, int* agreed);   // api.hip: a not-positive-definite failure made collective

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

// Run `enqueue` -- a callable that ONLY enqueues work on ctx->st and returns a status -- as the cached graph cg.  The first
// call with a key runs the launches eagerly (its results are this call's results; one-time host configuration of the kernels
// happens here) and then captures the same sequence for the calls that follow; a context with graphs off, or with the
// developer stamps armed, always launches eagerly.
inline void chain_graph_release(ChainGraph& cg) {
    if (cg.exec) (void)hipGraphExecDestroy(cg.exec);
    cg.exec = nullptr;
    cg.key.clear();
}
// a chain that falls back to eager launches says so once per process (a silent fallback is a silent slow-down)
inline void chain_graph_note() {
    static bool said = false;
    if (!said) fprintf(stderr, "libpilco_hip: a launch chain could not be captured as a hipGraph; it stays on eager launches\n");
    said = true;
}
template <class F>
int run_chain_graph(pilco_ctx* ctx, ChainGraph& cg, const std::vector<unsigned long long>& key, F&& enqueue) {
    if (!ctx->use_graph || ctx->dbg || cg.failed) return enqueue();
    if (cg.exec && cg.key == key) {
        HIPCHK(hipGraphLaunch(cg.exec, ctx->st));
        return PILCO_OK;
    }
    chain_graph_release(cg);
    if (int r = enqueue()) return r;   // this call's work, eagerly
    hipGraph_t graph = nullptr;
    if (hipStreamBeginCapture(ctx->st, hipStreamCaptureModeThreadLocal) != hipSuccess) {
        (void)hipGetLastError();
        cg.failed = true;
        chain_graph_note();
        return PILCO_OK;
    }
    const int rc = enqueue();
    const hipError_t e = hipStreamEndCapture(ctx->st, &graph);
    if (rc != PILCO_OK || e != hipSuccess || !graph || hipGraphInstantiate(&cg.exec, graph, nullptr, nullptr, 0) != hipSuccess) {
        (void)hipGetLastError();
        cg.exec = nullptr;
        cg.failed = true;   // eager from now on (the work of this call is already enqueued)
        chain_graph_note();
    } else {
        cg.key = key;
    }
    if (graph) (void)hipGraphDestroy(graph);
    return PILCO_OK;
}

// api.hip
// which outputs this rank factorises / trains, and their hyper-parameters / targets compacted for the batched kernels
struct OwnView {
    int W, rank, EL, ELcap;
    const double *ls, *var, *noise, *Yt;   // [EL][D], [EL], [EL], [EL][Npad]
};
int prepare_own(pilco_ctx* ctx, Slot& s, OwnView& o);
int check_slot(pilco_ctx* ctx, int slot);
int build_work(pilco_ctx* ctx, Slot& s);          // (re)builds the per-slot step workspace and its geometry
MMModel model_of(const Slot& s);
int all_gather_segments(pilco_ctx* ctx, Slot& s);
int pilco_factorize_fitc(pilco_ctx* ctx, void* slot_ptr);

// rollout.hip
struct RolloutPlan {
    GlueArgs g{};
    double* st[2] = {nullptr, nullptr};  // double-buffered state: m_x[E] | s_x[E*E]
    double* s1b[2] = {nullptr, nullptr}; // double-buffered [s_x, s_x c_xu] (fused head)
    int E = 0, D = 0, U = 0;
    double* jrec = nullptr;              // Jacobian tape (bwd.hip): the dynamics step runs launch_mm_jac and writes jrec[t]
    size_t jstride = 0;
    int jsmall = 0;                      // > 0: the steps of this value-and-gradient rollout run as the one-launch small step (chunks per pair)
};
constexpr int PILCO_JAC_TOO_LARGE = -77;   // rollout_jtape: the per-step buffers would exceed the cap (caller falls back)
// Device reverse chain of a LinearController's gradient: what rollout_jtape(.., dev) leaves enqueued / staged.
struct JtapeDev {
    bool seeds = false;                 // the caller has cotangent seeds to add: the chain kernel is launched by rollout_jtape_dev_finish, behind their upload
    RevArgs ra{};
    size_t n_seeds = 0, n_out = 0;
    double* h_seeds = nullptr;          // pinned staging [H + 1][E + E*E]
    const double* h_out = nullptr;      // pinned: dW | db | status | d / d (m_0, S_0), written by the chain kernel
    const double* h_traj = nullptr;     // pinned trajectory (seeds only; valid once jwait_ev[0] has passed)
    const double* h_reward = nullptr;   // pinned reward (valid with h_out)
};
typedef void (*jtape_seed_fn)(void* user, int H, int E, const double* traj, double* seeds);
int rollout_jtape_dev_finish(pilco_ctx* ctx, JtapeDev& dev, int H, int E, jtape_seed_fn seed_fn, void* seed_user);
int rollout_jtape_wait(pilco_ctx* ctx, int t);   // blocks until the records of step t have arrived
// forward rollout with the tape and the Jacobian records of every step, downloaded into pinned memory (grad.hip)
int rollout_jtape(pilco_ctx* ctx, const pilco_policy* policy, const pilco_reward_term* rewards, int n_rewards, const double* m0,
                  const double* S0, int H, double* reward, const double** traj, const double** tape, const double** jrec, size_t* jstride,
                  const double** reward_later = nullptr,    // reward_later: return without waiting (one rank); *reward_later is valid once rollout_jtape_wait(ctx, H - 1) has returned
                  JtapeDev* dev = nullptr);                 // dev: records stay on the device, the reverse chain runs there (LinearController); returns without waiting
int rollout_lanes(pilco_ctx* ctx, int B, std::vector<pilco_ctx*>& lane, const char* who);   // lanes of a batch call (rollout.hip)
void rollout_lanes_done(pilco_ctx* ctx);   // ... and after it: the context is on its own again
struct LanesGuard {
    pilco_ctx* c;
    ~LanesGuard() { rollout_lanes_done(c); }
};
int setup_rollout(pilco_ctx* ctx, const pilco_policy* pol, const pilco_reward_term* rw, int n_rw, int H, bool want_traj,
                  RolloutPlan& plan);
int enqueue_rollout(pilco_ctx* ctx, RolloutPlan& plan, int H, std::vector<hipEvent_t>* pair_ev);
int run_rollout(pilco_ctx* ctx, RolloutPlan& plan, int H);
// shard.hip: peer exchange
void launch_peer_wait(hipStream_t st, unsigned long long* area, int k, int W, int spin);
int peer_detach(pilco_ctx* ctx);
