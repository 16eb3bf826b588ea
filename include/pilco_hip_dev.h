/* pilco_hip_dev.h -- developer / measurement entry points of libpilco_hip.so.
 *
 * NOT part of the drop-in boundary (include/pilco_hip.h): nothing a binding of the reference's MGPR / PILCO surface needs.
 * bench.py (roofline: HIP-event timers around the O(N^2) launches), tools/ (phase stamps, raw work buffers) and the tests of
 * the stream-K split use them.  Same library, same context handle. */
#ifndef PILCO_HIP_DEV_H
#define PILCO_HIP_DEV_H
#include "pilco_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ launch-structure knobs (A/B and cross-check tests; none changes what is computed) */
/* pair-kernel variant: 0 = MFMA, stream-K work split (default, fastest; run-to-run bitwise stable);
 * 1 = plain-VALU tiled reference; 2 = MFMA tiled (bits also independent of the number of ranks).
 * All three agree to rounding. */
int pilco_set_pair_kernel(pilco_ctx* ctx, int variant);
/* Launch structure of a rollout step.  1 (default): "fused head" -- the serial link of step t (reduce the pair sums,
 * assemble (M,S,V), propagate, controller, joint Gaussian: mgpr.py:143-149, pilco.py:139-149) runs redundantly inside every
 * workgroup of step t+1's operand kernel, two launches per horizon step.  0: separate link kernel, three launches per
 * step (always used with more than one rank unless the peer exchange is attached).  Both produce bitwise identical results. */
int pilco_set_fused_step(pilco_ctx* ctx, int on);
/* Models of at most 256 points (every example of the reference; the inducing points of a sparse model: smgpr.py:47-52): 1
 * (default; PILCO_SMALL_STEP=0 in the environment starts with 0) = with the fused step, the operand launch's pair workgroups
 * also evaluate their pair sums, so a horizon step is ONE launch; 0 = the pair sums keep their own launch.  Same arithmetic
 * per element; the two split the sums differently, so results agree to rounding and each is bitwise repeatable. */
int pilco_set_small_step(pilco_ctx* ctx, int on);
/* How pilco_rollout_grad / pilco_rollout_grad_rbf obtain the moment-matching adjoint: 1 (default) = Jacobian tape -- the
 * forward rollout runs the reverse sweep in place of the forward pair kernel, so that ONE O(N^2) pass per step yields the
 * step's value and its Jacobian, and the reverse sweep is host algebra on the downloaded records; 0 = plain tape, then the
 * O(N^2) adjoint of every step on the device again (pilco_gp_predict_vjp).  Same gradient up to rounding. */
int pilco_set_grad_mode(pilco_ctx* ctx, int mode);
/* The reverse chain of pilco_rollout_grad* for a LinearController -- H dependent steps of small contractions over the Jacobian
 * records (propagate / joint / controller + squash / reward adjoints: pilco.py:141-149, controllers.py:13-58, rewards.py:19-81):
 * 1 (default; PILCO_HOST_CHAIN=1 in the environment starts with 0) = on the device, one workgroup walking the records in HBM
 * (csrc/rev.hip); 0 = the host chain of rounds 1-5 (csrc/grad.hip), which is what an RbfController always gets.  Same
 * formulas, different summation orders: they agree to rounding, each is bitwise repeatable. */
int pilco_set_reverse_chain(pilco_ctx* ctx, int on_device);
/* 1 (default; PILCO_NO_GRAPH=1 in the environment starts with 0): a rollout's launch sequence is captured once into a
 * hipGraph and replayed while the plan is unchanged; 0: every rollout is enqueued launch by launch.  Same results. */
int pilco_set_use_graph(pilco_ctx* ctx, int on);
/* RbfController inside a rollout.  1 (default; PILCO_INLINE_POLICY=0 starts with 0): a policy GP that is small enough
 * (bf <= 256 basis functions, state_dim <= 16, control_dim <= 4, U(U+1)/2 bf^2 <= 16384) is evaluated INSIDE the serial
 * link of the step -- by the workgroups that run the link anyway, from the state in LDS -- so that a step is two launches
 * (head, pair sums) as with a LinearController; 0: the policy GP gets its own operand and pair launch (four launches per
 * step), which is also what larger policies always get.  Same formulas (controllers.py:108-121 -> mgpr.py:99-149 with
 * iK = 0); the summation order differs, results agree to rounding (~1e-13), each mode is bitwise repeatable. */
int pilco_set_inline_policy(pilco_ctx* ctx, int on);

/* pilco_rollout that also returns, per step, the joint Gaussian handed to the dynamics GP (m (D), s (D,D), s1 (E,D)) and its
 * outputs (M (E), S (E,E), V (D,E)): the tape of the plain-tape gradient mode (pilco_set_grad_mode 0) and of tests. */
int pilco_rollout_tape(pilco_ctx* ctx, const pilco_policy* policy, const pilco_reward_term* rewards, int n_rewards,
                       const double* m0, const double* S0, int H, double* mH, double* SH, double* reward, double* traj,
                       double* tape);

/* ------------------------------------------------------------------ sharding plan / exchange introspection */
int pilco_shard_owner_of_pair(const pilco_ctx* ctx, int pair_index);
/* Layout of the per-step exchange (pure host functions, usable without a GPU).
 * out5 = {local pairs, owned outputs, SEG (doubles per rank), OUTOFF, P}; pairs are dealt
 * round-robin in the order (0,0),(1,1),..,(E-1,E-1),(1,0),(2,0),(2,1),.. and output a belongs
 * to the owner of (a,a).  *_slot return indices into the gathered buffer [nranks][SEG]. */
int pilco_shard_plan(int E, int D, int nranks, int rank, int* out5);
int pilco_shard_pair_slot(int E, int D, int nranks, int a, int b);
int pilco_shard_output_slot(int E, int D, int nranks, int a);
int pilco_peer_attached(const pilco_ctx* ctx);
/* the number of ranks RCCL ITSELF reports for the attached communicator (ncclCommCount); 0 = none attached, -1 = RCCL error */
int pilco_comm_count(const pilco_ctx* ctx);

/* ------------------------------------------------------------------ timing / introspection */
/* Time `reps` back-to-back rollouts with HIP events on the library's stream.
 * ms_total: wall time of the timed region; ms_pair: summed duration of the
 * pair kernel launches inside it (its own event pairs); n_pair_launches: count. */
int pilco_rollout_timed(pilco_ctx* ctx, const pilco_policy* policy, const pilco_reward_term* rewards, int n_rewards,
                        const double* m0, const double* S0, int H, int reps, double* mH, double* SH, double* reward,
                        float* ms_total, float* ms_pair, int* n_pair_launches);
/* Measurement aid for ANY rollout entry (pilco_rollout, pilco_rollout_tape, pilco_rollout_grad*): while on, every rollout
 * is enqueued launch by launch with a HIP-event pair around each launch of the step's O(N^2) kernel (the forward pair
 * kernel, or the reverse sweep of a value-and-gradient rollout).  pilco_get_pair_timing synchronises the stream and returns
 * the summed duration and the number of those launches of the LAST rollout.  Perturbs the rollout's total time: never
 * switched on inside a timed region. */
int pilco_set_pair_timing(pilco_ctx* ctx, int on);
int pilco_get_pair_timing(pilco_ctx* ctx, float* ms_pair, int* n_pair_launches);
/* Developer aid: the first call (out32 may be NULL) switches on phase timestamps inside the
 * prep / glue kernels (100 MHz wall clock); later calls copy the 32 slots of the last launch. */
int pilco_debug_timestamps(pilco_ctx* ctx, unsigned long long* out32);
/* per-workgroup (start, end) stamps of the last prep launch, n values (developer aid) */
int pilco_debug_blocks(pilco_ctx* ctx, unsigned long long* out, int n);
/* developer aid: raw copy of a work buffer (0 row operands At, 1 column operands Wt | vcol, 2 reverse-pass row moments, 3 column sums, 4 beta) */
int pilco_debug_buffer(pilco_ctx* ctx, int slot, int which, double* out, long n);
/* test aid: fills a factorisation buffer of the slot (0 = L^-1, 1 = iK, 2 = beta) with NaN bit patterns, so that a test can show
 * which parts of it the next factorisation really writes before anybody reads them */
int pilco_debug_poison(pilco_ctx* ctx, int slot, int which);
/* Stream-K work split of the pair kernel (pure host functions, no GPU): the column steps of nd diagonal pairs (tdiag
 * steps each, cost ud) and n_pairs - nd off-diagonal pairs (toff steps, cost uo) lie on one line cut into `waves` equal
 * cost ranges.  pilco_debug_sk_boundary: first step of wave w (w = waves: the total).  pilco_debug_sk_pair_waves:
 * out3 = (first wave holding a partial of pair k, slot of the pair in that wave (0/1), last such wave). */
int pilco_debug_sk_boundary(int w, int waves, int nd, int tdiag, int toff, int ud, int uo, int n_pairs);
int pilco_debug_sk_pair_waves(int k, int waves, int nd, int tdiag, int toff, int ud, int uo, int n_pairs, int* out3);
/* Time `reps` factorisations (invalidating the cache each time): ms per factorisation. */
int pilco_factorize_timed(pilco_ctx* ctx, int slot, int reps, float* ms_each);

#ifdef __cplusplus
}
#endif
#endif /* PILCO_HIP_DEV_H */
