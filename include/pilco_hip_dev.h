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
