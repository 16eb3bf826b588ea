/*
 * pilco_hip.h -- C ABI of libpilco_hip.so, the MI355X (gfx950) implementation of
 * the PILCO moment-matching rollout and GP-factorisation hot path.
 *
 * The reference (nrontsis/PILCO) has no FFI layer: its boundary is the Python
 * method surface of pilco.models.MGPR / SMGPR / PILCO.  Each entry point below
 * names the reference method (file:line under /root/reference) whose arithmetic
 * it replaces; pilco_amd/ (Python, ctypes) re-creates that method surface on top
 * of this header, and INTEGRATION.md shows the stub a maintainer of the
 * reference would add.
 *
 * Conventions: every matrix is row-major float64 in caller-owned HOST memory
 * unless a parameter is documented as a device pointer; the library owns all
 * device memory.  A context is bound to one GPU and must be used from one host
 * thread at a time.  Every function returns a pilco_status (0 = OK) and never
 * aborts; pilco_last_error() gives the message for the last failure.
 */
#ifndef PILCO_HIP_H
#define PILCO_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

/* 2: round 6 -- the round-4 persistent-kernel entry points are gone (since round 5) and the launch-structure knobs,
 * the sharding-plan introspection and pilco_rollout_tape are declared in pilco_hip_dev.h (still exported). */
#define PILCO_HIP_ABI_VERSION 2

typedef struct pilco_ctx pilco_ctx;

typedef enum pilco_status {
    PILCO_OK = 0,
    PILCO_E_SHAPE = 1,   /* inconsistent sizes / null pointer */
    PILCO_E_NOT_PD = 2,  /* Cholesky failed: the reference raises InvalidArgumentError (tests/test_cascade.py:22) */
    PILCO_E_HIP = 3,     /* HIP runtime error */
    PILCO_E_RCCL = 4,    /* RCCL error */
    PILCO_E_STATE = 5,   /* call order violated (e.g. predict before set_data) */
    PILCO_E_ALLOC = 6
} pilco_status;

/* ------------------------------------------------------------------ context */
int pilco_abi_version(void);
/* device: HIP device ordinal.  One context per process per GPU. */
int pilco_ctx_create(int device, pilco_ctx** out);
int pilco_ctx_destroy(pilco_ctx* ctx);
const char* pilco_last_error(const pilco_ctx* ctx);
/* index of the output whose Gram matrix was not positive definite (-1 if none) */
int pilco_last_not_pd_output(const pilco_ctx* ctx);
/* checks the f64 MFMA fragment layout assumptions on the device; 0 = OK */
int pilco_selftest(pilco_ctx* ctx);

/* ------------------------------------------------------------------ GP model
 * Slot 0 is the dynamics model (MGPR / SMGPR); slot 1 is the RBF policy
 * (RbfController is an MGPR subclass, pilco/controllers.py:80-129).            */
#define PILCO_SLOT_DYNAMICS 0
#define PILCO_SLOT_POLICY 1

/* MGPR.__init__/set_data (pilco/models/mgpr.py:18-45): X (N,D), Y (N,E).
 * N may change between calls; any cached factorisation is invalidated. */
int pilco_gp_set_data(pilco_ctx* ctx, int slot, const double* X, const double* Y, int N, int D, int E);
/* kernel.lengthscales (E,D), kernel.variance (E), likelihood.variance (E)
 * (properties at pilco/models/mgpr.py:170-186). Invalidates the factorisation. */
int pilco_gp_set_hyp(pilco_ctx* ctx, int slot, const double* lengthscales, const double* variance, const double* noise);
/* SMGPR inducing inputs Z (M,D) of model 0, used for every output
 * (pilco/models/smgpr.py:20-22,47-52).  M = 0 switches back to the exact GP. */
int pilco_gp_set_inducing(pilco_ctx* ctx, int slot, const double* Z, int M);
/* MGPR.K(X1, X2) (pilco/models/mgpr.py:154-157): out (E,N1,N2). X2 may be NULL (= X1). */
int pilco_gp_gram(pilco_ctx* ctx, int slot, const double* X1, int N1, const double* X2, int N2, double* out);
/* MGPR.calculate_factorizations (pilco/models/mgpr.py:81-89) or, when inducing
 * inputs are set, SMGPR.calculate_factorizations (pilco/models/smgpr.py:24-45).
 * Result stays on the device and is cached until data / hyper-parameters change. */
int pilco_gp_factorize(pilco_ctx* ctx, int slot);
/* Negative log marginal likelihood of every output at the current hyper-parameters and its gradient
 * w.r.t. (lengthscales[D], kernel variance, noise variance): what gpflow's GPR.training_loss and its
 * autodiff provide to MGPR.optimize (pilco/models/mgpr.py:47-75), without the prior terms (added on
 * the host).  nlml (E), grad (E, D+2) may be NULL.  Exact GP only.
 * Several ranks: sharded by output like the factorisation under it -- a rank evaluates the outputs it owns; with a
 * communicator attached one ncclAllGather of (D + 3) doubles per output completes the arrays on every rank, without one the
 * other ranks' entries come back NaN and the caller combines them (contexts of one process: pilco_amd._lib.group_nlml). */
int pilco_gp_nlml(pilco_ctx* ctx, int slot, double* nlml, double* grad);
/* The same for the sparse model: negative log marginal likelihood of gpflow's GPRFITC (one model per output, each with its
 * OWN inducing inputs: pilco/models/smgpr.py:16-22) and its gradient w.r.t. (lengthscales[D], kernel variance, noise
 * variance) and w.r.t. the inducing inputs -- what MGPR.optimize's SciPy loop receives for an SMGPR
 * (pilco/models/mgpr.py:47-75; no priors, smgpr.py sets none).  Z_all (E,M,D); nlml (E); grad_hyp (E,D+2) and
 * grad_Z (E,M,D) may be NULL.  Uses the slot's data and hyper-parameters; invalidates its cached factorisation. 
 * Several ranks: sharded by output exactly like pilco_gp_nlml (own outputs evaluated, all-gather with a communicator, NaN for
 * the other ranks' outputs without one). */
int pilco_gp_fitc_nlml(pilco_ctx* ctx, int slot, const double* Z_all, int M, double* nlml, double* grad_hyp, double* grad_Z);
/* number of points the moment-matching runs over: N (exact) or M (sparse) */
int pilco_gp_num_points(const pilco_ctx* ctx, int slot);
/* download iK (E,n,n) and beta (E,n); either may be NULL */
int pilco_gp_get_factors(pilco_ctx* ctx, int slot, double* iK, double* beta);
/* upload caller-supplied factors for predict_given_factorizations(m,s,iK,beta)
 * (pilco/models/mgpr.py:91); iK may be NULL meaning all-zero (RbfController,
 * pilco/controllers.py:116). */
int pilco_gp_set_factors(pilco_ctx* ctx, int slot, const double* iK, const double* beta);
/* MGPR.predict_given_factorizations with the factors held on the device
 * (pilco/models/mgpr.py:91-149): m (1,D), s (D,D) -> M (1,E), S (E,E), V (D,E).
 * pilco_gp_predict = predict_on_noisy_inputs (mgpr.py:77-79) with the
 * factorisation cached instead of recomputed. */
int pilco_gp_predict(pilco_ctx* ctx, int slot, const double* m, const double* s, double* M, double* S, double* V);

/* ------------------------------------------------------------------ rollout */
typedef enum pilco_policy_kind {
    PILCO_POLICY_NONE = 0,   /* control_dim == 0 */
    PILCO_POLICY_LINEAR = 1, /* LinearController (pilco/controllers.py:39-63) */
    PILCO_POLICY_RBF = 2     /* RbfController (pilco/controllers.py:80-129); GP in PILCO_SLOT_POLICY */
} pilco_policy_kind;

typedef struct pilco_policy {
    int kind;
    int state_dim;            /* E of the dynamics model */
    int control_dim;          /* D - E */
    const double* W;          /* linear: (control_dim, state_dim) */
    const double* b;          /* linear: (1, control_dim) */
    const double* max_action; /* (control_dim) scale of squash_sin (controllers.py:13-36); NULL = ones */
    int squash;               /* 1 = apply squash_sin (the reference always does inside propagate) */
} pilco_policy;

typedef enum pilco_reward_kind {
    PILCO_REWARD_EXPONENTIAL = 1, /* ExponentialReward (pilco/rewards.py:7-51) */
    PILCO_REWARD_LINEAR = 2       /* LinearReward (pilco/rewards.py:53-61) */
} pilco_reward_kind;

typedef struct pilco_reward_term {
    int kind;
    double coef;     /* CombinedRewards coefficient (pilco/rewards.py:64-81); 1 for a single reward */
    const double* W; /* exponential: (E,E); linear: (E) */
    const double* t; /* exponential: (1,E) target; NULL = zeros */
} pilco_reward_term;

/* PILCO.predict (pilco/models/pilco.py:118-136) = H x [reward(m,s); propagate
 * (pilco.py:138-153)].  m0 (1,E), S0 (E,E) -> mH (1,E), SH (E,E), reward (sum of
 * the mean rewards of the PRE-propagation states).  traj (optional, may be
 * NULL): (H+1) x (E + E*E) doubles, the state after every step.
 * The factorisation of slot 0 (and slot 1 for an RBF policy) must be current. */
int pilco_rollout(pilco_ctx* ctx, const pilco_policy* policy, const pilco_reward_term* rewards, int n_rewards,
                  const double* m0, const double* S0, int H, double* mH, double* SH, double* reward, double* traj);
/* B independent rollouts of ONE model in flight together on one GPU: policies [B] (NONE / LINEAR; each lane may have its
 * own parameters), m0 [B][E], S0 [B][E][E] -> mH [B][E], SH [B][E][E], reward [B].  What multi-start policy search and the
 * evaluation of several initial states need (the reference evaluates its restarts one after the other, pilco.py:96-110).
 * Lanes 1..B-1 are internal contexts (own stream, workspace, state, cached graph) that borrow this context's model -- no
 * copy of X / beta / iK; all B graph replays are enqueued before the first wait, so the serial head of one lane's step
 * runs under the pair kernels of the others.  Every lane runs exactly pilco_rollout's launch sequence: its result is
 * BIT-IDENTICAL to its solo run.  Single rank.  More than 4 lanes overlap only if the HIP runtime has as many hardware
 * queues (GPU_MAX_HW_QUEUES, set before the runtime starts). */
int pilco_rollout_batch(pilco_ctx* ctx, int B, const pilco_policy* policies, const pilco_reward_term* rewards, int n_rewards,
                        const double* m0, const double* S0, int H, double* mH, double* SH, double* reward);
/* PILCO.propagate (pilco/models/pilco.py:138-153): one step, no reward. */
int pilco_propagate(pilco_ctx* ctx, const pilco_policy* policy, const double* m_x, const double* s_x, double* M_x, double* S_x);
/* controller.compute_action(m, s, squash) -> M (1,U), S (U,U), V (E,U)
 * (pilco/controllers.py:46-58,108-121) and reward.compute_reward(m,s) ->
 * muR, sR (pilco/rewards.py:19-51,58-61,73-81), evaluated on the device. */
int pilco_policy_action(pilco_ctx* ctx, const pilco_policy* policy, const double* m, const double* s, double* M, double* S, double* V);
int pilco_reward_eval(pilco_ctx* ctx, const pilco_reward_term* rewards, int n_rewards, int state_dim, const double* m, const double* s, double* muR, double* sR);

/* ------------------------------------------------------------------ reverse mode
 * The reference differentiates training_loss with TensorFlow's autodiff (pilco/models/pilco.py:85-90);
 * here the adjoint of the moment-matching step is hand-derived.
 * pilco_gp_predict_vjp: cotangents Mbar (1,E), Sbar (E,E), Vbar (D,E) of pilco_gp_predict's outputs ->
 * mbar (1,D), sbar (D,D, symmetric) at the input (m, s).  Single rank, D <= 32. */
int pilco_gp_predict_vjp(pilco_ctx* ctx, int slot, const double* m, const double* s, const double* Mbar,
                         const double* Sbar, const double* Vbar, double* mbar, double* sbar);
/* Cotangent seeds for objectives beyond the additive reward (SafePILCO's multiplicative risk term,
 * safe_pilco_extension/safe_pilco.py:29-50; any function of the state trajectory): after the forward pass the library
 * calls seed_fn(user, H, E, traj, seeds) with traj [H+1][E + E*E] (m_t | s_t, t = 0..H; state t < H is the
 * pre-propagation state the reward of step t sees) and seeds [H+1][E + E*E] zero-filled; the callback writes
 * d objective / d (m_t, s_t) and the reverse sweep adds them where the reward's own cotangents enter.  *reward is the
 * additive reward only: the caller adds its own term.  What TensorFlow's reverse mode through training_loss gives the
 * reference for whatever predict() returns (pilco/models/pilco.py:47-50, 85-90). */
typedef void (*pilco_seed_fn)(void* user, int H, int E, const double* traj, double* seeds);
int pilco_rollout_grad_seeded(pilco_ctx* ctx, const pilco_policy* policy, const pilco_reward_term* rewards, int n_rewards,
                              const double* m0, const double* S0, int H, pilco_seed_fn seed_fn, void* seed_user, double* reward,
                              double* dW, double* db);
int pilco_rollout_grad_rbf_seeded(pilco_ctx* ctx, const pilco_policy* policy, const pilco_reward_term* rewards, int n_rewards,
                                  const double* m0, const double* S0, int H, const double* Xp, const double* Yp, const double* lsp,
                                  const double* noisep, int bf, pilco_seed_fn seed_fn, void* seed_user, double* reward, double* dX,
                                  double* dY, double* dls);
/* Value and gradient of the rollout reward w.r.t. a squashed LinearController's parameters, dW (U,E) and db (U):
 * training_loss + TensorFlow reverse mode in the reference (pilco/models/pilco.py:47-50,85-90).  Forward rollout
 * with a tape, then the reverse sweep with the moment-matching adjoint of every step on the device and the O(D^3)
 * links (propagate, joint Gaussian, controller + squash, rewards) in native host code. */
int pilco_rollout_grad(pilco_ctx* ctx, const pilco_policy* policy, const pilco_reward_term* rewards, int n_rewards,
                       const double* m0, const double* S0, int H, double* reward, double* dW, double* db);
/* The same for an RbfController (policy GP in PILCO_SLOT_POLICY, pilco/controllers.py:80-129): gradients w.r.t. the
 * centres dX (bf,E), the targets dY (bf,U) and the lengthscales dls (U,E).  Xp, Yp, lsp, noisep (U): host copies of the
 * policy parameters that were uploaded with pilco_gp_set_data / pilco_gp_set_hyp (noise = FakeGPR likelihood variance). */
int pilco_rollout_grad_rbf(pilco_ctx* ctx, const pilco_policy* policy, const pilco_reward_term* rewards, int n_rewards,
                           const double* m0, const double* S0, int H, const double* Xp, const double* Yp, const double* lsp,
                           const double* noisep, int bf, double* reward, double* dX, double* dY, double* dls);
/* B value-and-gradient rollouts of ONE dynamics model in flight together -- the restarts of PILCO.optimize_policy
 * (pilco/models/pilco.py:94-107 runs them one after the other; each restart is an L-BFGS-B walk of its own, so the evaluations
 * of different restarts are independent).  Lanes as in pilco_rollout_batch; every lane's result is BIT-IDENTICAL to its solo
 * pilco_rollout_grad call.  LinearController lanes: policies[i].W / .b; m0 (B,E), S0 (B,E,E); reward (B), dW (B,U,E), db (B,U). */
int pilco_rollout_grad_batch(pilco_ctx* ctx, int B, const pilco_policy* policies, const pilco_reward_term* rewards, int n_rewards,
                             const double* m0, const double* S0, int H, double* reward, double* dW, double* db);
/* RbfController lanes: lane i's policy GP -- centres Xp (B,bf,E), targets Yp (B,bf,U), lengthscales lsp (B,U,E), likelihood
 * variances noisep (B,U), unit signal variance (pilco/controllers.py:92-93) -- is uploaded to and factorised in
 * PILCO_SLOT_POLICY of lane i's context BY THIS CALL, lane 0 = this context included: the caller's policy slot holds lane 0's
 * controller afterwards.  reward (B), dX (B,bf,E), dY (B,bf,U), dls (B,U,E); bit-identical to the solo pilco_rollout_grad_rbf. */
int pilco_rollout_grad_rbf_batch(pilco_ctx* ctx, int B, const pilco_policy* policies, const pilco_reward_term* rewards, int n_rewards,
                                 const double* m0, const double* S0, int H, const double* Xp, const double* Yp, const double* lsp,
                                 const double* noisep, int bf, double* reward, double* dX, double* dY, double* dls);
/* ... with an objective beyond the additive reward (Safe-PILCO's risk term, host-evaluated reward terms), as
 * pilco_rollout_grad_seeded: seed_fn is called once per lane, in lane order, with seed_users[i] (seed_users may be NULL), when
 * lane i's trajectory has arrived and before its reverse sweep; reward[i] is lane i's additive reward alone. */
int pilco_rollout_grad_batch_seeded(pilco_ctx* ctx, int B, const pilco_policy* policies, const pilco_reward_term* rewards, int n_rewards,
                                    const double* m0, const double* S0, int H, pilco_seed_fn seed_fn, void* const* seed_users,
                                    double* reward, double* dW, double* db);
int pilco_rollout_grad_rbf_batch_seeded(pilco_ctx* ctx, int B, const pilco_policy* policies, const pilco_reward_term* rewards, int n_rewards,
                                        const double* m0, const double* S0, int H, const double* Xp, const double* Yp, const double* lsp,
                                        const double* noisep, int bf, pilco_seed_fn seed_fn, void* const* seed_users, double* reward,
                                        double* dX, double* dY, double* dls);

/* (timing and introspection entry points -- HIP-event timers, in-kernel phase stamps, raw work buffers, the stream-K split --
 * are developer aids of the same library, declared in include/pilco_hip_dev.h: not part of the boundary a binding needs.) */

/* ------------------------------------------------------------------ multi-GPU
 * One process per GPU.  The P = E(E+1)/2 output pairs (and with them the
 * outputs' factorisations) are dealt round-robin over the ranks; one
 * ncclAllGather (RCCL over xGMI) per horizon step reassembles (M, S, V).
 * id: 128 opaque bytes produced on rank 0 and broadcast by the host launcher. */
#define PILCO_COMM_ID_BYTES 128
int pilco_comm_unique_id(void* id128);
int pilco_comm_init(pilco_ctx* ctx, const void* id128, int rank, int nranks);
/* sharding without a communicator: the caller moves the bytes (host fake
 * all-gather for tests; gloo fallback).  Pairs are dealt round-robin in the order (0,0),(1,1),..,(E-1,E-1),(1,0),(2,0),(2,1),..;
 * output a belongs to the owner of (a,a)  (the plan's introspection functions: include/pilco_hip_dev.h). */
int pilco_shard_set(pilco_ctx* ctx, int rank, int nranks);
/* One sharded moment-matching step with the exchange done by the caller: shard_pack runs this
 * rank's pairs and returns its SEG doubles; after an all-gather by any transport, shard_finish
 * assembles (M, S, V) from the [nranks][SEG] buffer.  pilco_gp_predict / pilco_rollout do the
 * same with ncclAllGather when a communicator is attached. */
int pilco_gp_shard_pack(pilco_ctx* ctx, int slot, const double* m, const double* s, double* segment);
int pilco_gp_shard_finish(pilco_ctx* ctx, int slot, const double* gathered, double* M, double* S, double* V);
/* Sharded factorisation: with nranks > 1 pilco_gp_factorize computes and stores K / L^-1 / iK only for the outputs this
 * rank owns (a = rank, rank + nranks, ...; memory and work / nranks, no communication) and then needs every rank's beta
 * rows: one ncclAllGather per MODEL when a communicator is attached, or -- contexts of one process -- this call, after
 * all of them have factorised. */
int pilco_group_sync_model(pilco_ctx** ctxs, int n, int slot);
/* The same beta exchange over any host transport (ranks in different processes, no communicator): pilco_gp_beta_rows
 * gives the block shape (elcap = ceil(E / nranks) rows of npad doubles per rank), every rank exports the rows it
 * computed, the caller all-gathers the blocks in rank order, every rank imports [nranks][elcap][npad]. */
int pilco_gp_beta_rows(pilco_ctx* ctx, int slot, int* elcap, int* npad);
int pilco_gp_beta_export(pilco_ctx* ctx, int slot, double* own_rows);
int pilco_gp_beta_import(pilco_ctx* ctx, int slot, const double* all_rows);
/* The whole sharded rollout over n contexts of ONE process (context i = rank i of n; several contexts may share a GPU):
 * every context runs pilco_rollout on its own host thread and the per-step exchange is done with peer copies between
 * host barriers instead of ncclAllGather -- the same launch sequence as the RCCL path, so the multi-rank rollout can be
 * validated without a multi-GPU node (host-synchronised per step: a test vehicle, not a fast path).
 * Outputs are rank 0's; *mismatch (may be NULL) = 1 if another rank finished with a different bit pattern. */
int pilco_rollout_group(pilco_ctx** ctxs, int n, const pilco_policy* policy, const pilco_reward_term* rewards, int n_rewards,
                        const double* m0, const double* S0, int H, double* mH, double* SH, double* reward, double* traj,
                        int* mismatch);
/* Value and gradient of ONE sharded rollout (pilco_rollout_grad, LinearController) over n contexts of this process, as
 * pilco_rollout_group does the forward rollout: every rank sweeps its own pairs, the per-pair Jacobian records are
 * all-gathered once after the horizon (between processes pilco_rollout_grad does that itself with ncclAllGather when a
 * communicator is attached), every rank runs the same host reverse sweep on the same records.  reward [n], dW [n][U*E],
 * db [n][U]: every rank's results (bit-identical to one another). */
int pilco_rollout_grad_group(pilco_ctx** ctxs, int n, const pilco_policy* policy, const pilco_reward_term* rewards, int n_rewards,
                             const double* m0, const double* S0, int H, double* reward, double* dW, double* db);
/* The same for an RbfController (pilco_rollout_grad_rbf on every context; pilco/controllers.py:80-129).  The policy GP is not
 * sharded -- every rank holds all of it in PILCO_SLOT_POLICY and evaluates it inside its link kernel (inline policy: at most
 * 256 basis functions).  reward [n], dX [n][bf*E], dY [n][bf*U], dls [n][U*E]: every rank's results, bit-identical to one
 * another and to the single-rank pilco_rollout_grad_rbf (the split of every pair's sums does not depend on the rank count). */
int pilco_rollout_grad_rbf_group(pilco_ctx** ctxs, int n, const pilco_policy* policy, const pilco_reward_term* rewards, int n_rewards,
                                 const double* m0, const double* S0, int H, const double* Xp, const double* Yp, const double* lsp,
                                 const double* noisep, int bf, double* reward, double* dX, double* dY, double* dls);
/* Peer exchange: the per-step all-gather without a collective call.  Every rank owns an exchange area in its GPU's
 * memory; after its pair kernel a rank stores its segment straight into EVERY rank's area (over xGMI between GPUs) and
 * raises its flag there; the next step's head waits for the W flags of that exchange and reads the segments from its
 * own memory.  No host involvement per step, no RCCL launch (~20 us) on the critical path; the whole sharded rollout is
 * captured in one hipGraph.  Used by pilco_rollout whenever it is attached (linear / no policy; otherwise the RCCL path).
 *   ranks in different processes: pilco_peer_export on every rank (64 opaque bytes = a hipIpcMemHandle_t), all-gather the
 *   handles by any host transport, pilco_peer_attach(handles[nranks][64]).  share_gpu != 0 when ranks share a GPU
 *   (oversubscribed tests): the flag wait then runs as a one-workgroup launch of its own so that waiting kernels cannot
 *   keep the awaited ones off the machine.
 *   contexts of one process: pilco_group_peer_attach(ctxs, n).
 * A wait that gives up (~2 s) makes pilco_rollout return PILCO_E_STATE instead of hanging.  All ranks must make the same
 * sequence of pilco_rollout calls.  pilco_shard_set / pilco_comm_init detach. */
#define PILCO_PEER_HANDLE_BYTES 64
int pilco_peer_export(pilco_ctx* ctx, void* handle64);
int pilco_peer_attach(pilco_ctx* ctx, const void* handles, int share_gpu);
int pilco_group_peer_attach(pilco_ctx** ctxs, int n);
int pilco_peer_detach(pilco_ctx* ctx);
int pilco_comm_rank(const pilco_ctx* ctx);
int pilco_comm_size(const pilco_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* PILCO_HIP_H */
