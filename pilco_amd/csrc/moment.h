// Device-side descriptors of the moment-matching step (internal).
#pragma once
#include "common.h"

namespace pilco {

constexpr int MAX_REWARD_TERMS = 4;
// Experiment switches (skip a stage of the step to time the rest) exist only in developer builds of tools/
// (-DPILCO_DEV); in the product library the tests are compile-time false and the stages cannot be removed.
#ifdef PILCO_DEV
#define MM_ABL(wk_, bit_) (((wk_).abl & (bit_)) != 0)
#else
#define MM_ABL(wk_, bit_) false
#endif
constexpr int MAX_D = 32;  // GP input dimension supported by the register-tiled kernels

// One GP as the moment-matching kernels see it (all device pointers).
struct MMModel {
    const double* Pt;    // [D][npad]  points, transposed (training inputs X, inducing Z, or RBF centres)
    const double* ls;    // [E][D]
    const double* var;   // [E]
    const double* lvar;  // [E]  log var (the operand rows carry log var_a - ...: taken on the host when the hyper-parameters arrive, not by every workgroup of every step)
    const double* beta;  // [E][npad]  zero padded
    const double* iK;    // [E][npad][npad] zero padded, or nullptr (== 0: RbfController, controllers.py:116)
    int n, npad, D, E;
    // Sharded factorisation: rank r owns the outputs a = r, r + bW, ...; beta is the all-gathered [bW][bEL][npad] buffer
    // (row of output a: (a % bW) * bEL + a / bW), iK holds only the OWNED outputs (block a / bW).  bW = 1: plain [E] layout.
    int bW, bEL;
};

// Per-slot workspace of one step.
struct MMWork {
    double* in_m;        // [D]      input mean  (joint state-action mean)
    double* in_s;        // [D][D]   input covariance
    // Operands of the exponent GEMM e_ij = sum_k A[k][i] B[k][j], K = KP rows: A = (2 Q z_i | u_i | 1 | 0..), B = (w_j | 1 | v_j | 0..).
    // Only what depends on the step AND the pair is written per pair and step: p_i = 2 Q z_i, u_i (At) and v_j (vcol).
    // w_j = (x_j - m) / l_b^2 depends on the pair's COLUMN output b alone: one block per output (Wt), written by the first
    // local pair with that column -- E blocks instead of P (round 5: the head's write-through traffic 10.8 -> 6.2 MB per step
    // at C2).  The ones (the valid mask: 1 for points < n) and the zero rows of the padded contraction are constants of the
    // model: they sit in the rows of the blocks where the contraction expects them and are written once, when the workspace
    // is built.  A reader addresses both operands as before (block base + k npad); only the lanes that hold row D + 1 of B
    // are pointed at the pair's v_j instead (one select per pair_wave call, PairOps in mm_device.h).
    double* At;          // [PL][KP][npad]  rows 0..D: (2 Q z_i | u_i) of the pair, written every step; row D + 1: ones (not vsep); beyond: zeros
    double* Wt;          // [blocks][KP][npad]  rows 0..D-1: w_j of the column block, written every step by the block's writer; row D: ones;
                         //   beyond: zeros (row D + 1 is never read: v_j lives in vcol).  Block = the column output b (one-launch small
                         //   step with its operands in memory: the local pair)
    double* vcol;        // [PL][npad]  v_j of every pair, in the SAME allocation as Wt (behind its blocks: one buffer resource reaches
                         //   both): row D + 1 of B or -- vsep (D + 2 = 1 mod 4: K = D + 1) -- added on the VALU after the contraction
    int vsep;
    int wt_R, wt_deal;   // the w rows of all column blocks dealt over the local pairs, wt_R rows each (wt_deal: 1 when they are dealt,
                         // mm_device.h: wt_rows_dealt) -- taken on the host: a signed division is ~40 dependent operations of the head's serial path
    double* pair_isdet;  // [PL]     1/sqrt(det R_ab)
    double* mean_part;   // [EL][NCHM][1+D]  per row chunk: c_a g and c_a T_a h (contributions to M_a, V_a)
    double* pair_part;   // [PL][NT][2]
    double* gath;        // [nranks][SEG]  packed per-rank results (all-gather buffer)
    double* out_M;       // [E]
    double* out_S;       // [E][E]
    double* out_V;       // [D][E]
    // Ownership is closed-form: pairs are dealt round-robin in the order
    // (0,0),(1,1),..,(E-1,E-1),(1,0),(2,0),(2,1),(3,0).. : order index kk -> rank kk % nranks,
    // local index kk / nranks; output a belongs to the owner of (a,a).
    // stream-K decomposition of the MFMA pair kernel (variant 0)
    double* sk_part;     // [sk_maxw][sk_pls] stream-K partials, slot-major: slot = wave - first wave of the pair (unused slots stay zero)
    int sk_waves, sk_total, sk_nd, sk_tdiag, sk_toff, sk_maxw;
    int sk_pls;          // row stride of the slot-major partial array (local pairs rounded up to 16)
    unsigned long long* dbg;  // optional [32] phase timestamps (100 MHz wall clock) of the last prep / glue launch
    int abl;             // experiment switches (PILCO_ABL, tools only; 0 in product use)
    int sk_ud, sk_uo;    // cost units of a diagonal / off-diagonal column step (diagonal steps also stream iK)
    int fuse_pair;       // small models: the operand launch's pair workgroups also evaluate their pair sums (pair_part [PL][NCH][2]); no pair launch.  2: ... as the reverse sweep (value-and-gradient rollouts)
    double* sw_gpart;    // fuse_pair == 2: this step's [PL][NT][2][256] blocks (G | Gc per workgroup)
    int share_cu;        // the head launches its 128-register build (two workgroups per CU): this context is a lane of a batch call
    int NCS;             // fuse_pair with LDS-resident operands: column splits per (pair, row chunk) -- the launch spreads a small model over
                         // the CUs it would leave idle (grid y = NCH * NCS, NT = NCH * NCS partials per pair); 1 elsewhere
    const double* exp_tab;    // [n] 2^(j/n), n = mm_exp_table_size(), for the table-driven fp64 exp of the pair kernel
    int PL, EL, P, KP, NCH, NCHM, NT, SEG, OUTOFF, rank, nranks;  // NCH / NCHM: row chunks of the pair / mean-part prep workgroups; OUTOFF: offset of the output records inside a segment
};

struct RewardDev {
    int kind;
    double coef;
    const double* W;  // device
    const double* t;  // device (never null: zeros if the caller passed NULL)
    const double* F;  // device [E][rank]: W = F F^T (symmetric PSD W), or nullptr
    int rank;         // >= 0: factored fast path; -1: general pivoted path
};

// Reward of the rollout's current state, evaluated by one spare workgroup of the dynamics prep launch (the state it
// reads was written by the previous glue kernel; nothing on the step's critical path waits for it).
struct PrepReward {
    int n;                // number of terms; 0 = no reward workgroup
    int E;                // state dimension
    RewardDev rw[MAX_REWARD_TERMS];
    const double* m_x;    // [E]
    const double* s_x;    // [E][E]
    double* reward;       // [1] accumulator (single writer, stream ordered)
};

enum GlueFlags {
    GF_PACK = 1,       // reduce tile partials of the local pairs/outputs into gath[rank]
    GF_ASSEMBLE = 2,   // gath -> out_M, out_S, out_V
    GF_PROPAGATE = 4,  // state <- state + GP increment (pilco.py:147-149)
    GF_TRAJ = 8,       // store state into traj[step]
    GF_POLICY = 32,    // controller + joint Gaussian -> in_m, in_s, s1 (pilco.py:139-144)
    GF_RBF_POST = 64,  // RBF policy: S -= diag(var - 1e-6), squash, joint (controllers.py:116-121)
    GF_RBF_PRE = 128   // RBF policy: copy the state into the policy slot's input
};

struct GlueArgs {
    int flags;
    int E;  // state dim
    int D;  // GP input dim = E + U
    int U;
    MMWork wk;          // dynamics slot workspace
    const double* var;  // dynamics kernel variances [E]
    MMWork pwk;         // policy slot workspace (RBF policy only)
    const double* pvar; // policy kernel variances [U]
    // rollout state
    const double* m_x;  // [E]     current state (read)
    const double* s_x;  // [E][E]
    double* m_out;      // [E]     next state (written by GF_PROPAGATE; the other half of the double buffer)
    double* s_out;      // [E][E]
    double* s1;      // [E][D]  = [s_x, s_x c_xu], kept for propagate (read by GF_PROPAGATE)
    double* s1_out;  // where the new joint's [s_x, s_x c_xu] goes; nullptr = s1 (the fused head double-buffers it: every
                     // workgroup reads s1 while the writer workgroup stores the next one)
    double* reward;  // [1]
    double* traj;    // [(H+1)][E + E*E] or nullptr
    double* tape;    // [H][D + D*D + E*D + E + E*E + D*E] joint (m, s, s1) and GP outputs (M, S, V) of every step, or nullptr
    int step;
    int dbg_off;     // developer aid: slot offset of this launch's phase stamps (0 = default)
    // RbfController evaluated INSIDE the link (glue_device.h: rbf_policy_inline) instead of by an operand + pair launch of its
    // own: the policy GP's model (centres, lengthscales, targets' beta; iK = 0, controllers.py:116) and its LDS scratch
    MMModel pmd;
    int pol_inline;  // 1: inline evaluation (small policy GPs: pol_lds > 0 doubles of extra LDS behind the link's region)
    int pol_lds;
    // policy
    int pol_kind;
    const double* W;       // [U][E]
    const double* b;       // [U]
    const double* maxact;  // [U]
    int squash;
    int n_rewards;
    RewardDev rw[MAX_REWARD_TERMS];
    // direct outputs of pilco_policy_action / pilco_reward_eval (optional)
    double* act_out;  // [U + U*U + E*U]
    double* rew_out;  // [2] mean, variance: set only by pilco_reward_eval
    // peer exchange (SURVEY 8e without a collective call): every rank owns an exchange area (see PeerArea) that its peers
    // write into directly -- over xGMI when the ranks sit on different GPUs.  A GF_PACK launch with xq_peers set stores
    // its segment into every rank's area and then raises its flag there; a GF_ASSEMBLE launch with xq set waits for the
    // W flags of exchange xq_k of this rollout and takes the segments from its own area.
    unsigned long long* xq;         // this rank's exchange area, or nullptr
    unsigned long long** xq_peers;  // device array [W]: every rank's area as mapped on this device, or nullptr
    int xq_k;                       // exchange number within the rollout: epoch = area[0] + xq_k + 1
    int xq_W;
    int xq_cap;                     // doubles per segment slot
    int xq_spin;                    // bound of the flag wait (iterations of ~1 us): on expiry the area's error word is set
};

// Layout of an exchange area, in 8-byte words:  [0] epoch base of the current rollout (uploaded by the owner's host)
// [1] error word (0 = fine; else the epoch a wait gave up on)   [8 + s * W + r] flag of rank r in slot s (s = epoch & 1):
// the last epoch rank r has delivered there   [xq_data_off(W) + (s * W + r) * cap + i] segment of rank r in slot s.
// Two slots suffice: a rank delivers epoch e + 2 only after it has seen every flag of epoch e + 1, which the others raise
// after they have finished reading epoch e.
__host__ __device__ inline int xq_data_off(int W) { return (8 + 2 * W + 7) & ~7; }
__host__ __device__ inline size_t xq_area_words(int W, int cap) { return (size_t)xq_data_off(W) + (size_t)2 * W * cap; }

// fused != nullptr: the "fused head" -- every workgroup first runs the serial link of the PREVIOUS step (glue_body with
// *fused: pack / assemble / propagate / controller / joint) redundantly and takes the joint Gaussian from its own LDS;
// fused->wk carries the buffers that link READS (previous step's partials), wk the ones this launch WRITES.
bool mm_fused_head_fits(const MMModel& md, int reward_E, const GlueArgs& ga);
void launch_mm_prep(hipStream_t st, const MMModel& md, const MMWork& wk, const PrepReward* pr = nullptr,
                    const GlueArgs* fused = nullptr);
size_t glue_lds_doubles_for(const GlueArgs& g);
// LDS doubles the inline evaluation of an RbfController with bf centres needs (0: too large for the inline path)
int rbf_inline_lds_doubles(int state_dim, int control_dim, int bf);
// variant 0 = MFMA stream-K, 1 = VALU (tiled), 2 = MFMA tiled (bits independent of the rank count)
void launch_mm_pair(hipStream_t st, const MMModel& md, const MMWork& wk, int variant);
// stream-K geometry: resident waves of the MFMA pair kernel for this KP, and the per-pair step counts
int mm_pair_sk_capacity(int KP, bool vsep);
void mm_pair_sk_steps(int npad, int* tdiag, int* toff);
int mm_sk_boundary(int w, int waves, int nd_steps, int total, int ud, int uo);
int mm_sk_maxw(const MMWork& wk);
void mm_sk_pair_waves(int k, int waves, int nd, int tdiag, int toff, int total, int ud, int uo, int* wlo, int* fslot, int* whi);   // needs the sk_* geometry fields and PL
void launch_glue(hipStream_t st, const GlueArgs& g, bool with_reward_block = false);
size_t glue_lds_bytes(int E, int D);
// tile-partial counts per pair for a variant (NT) and the number of row chunks of the prep kernel
int mm_pair_nt(int npad, int variant, int PL);
void mm_prep_chunks(int npad, int PL, int EL, int* nch, int* nchm);
inline int wt_rows_per_pair_of(int E, int D, int PL) { return (E * D + (PL > 1 ? PL : 1) - 1) / (PL > 1 ? PL : 1); }
int mm_prep_dt(int D);   // the operand kernel's instantiation (DT >= D) for this input dimension
// the contraction stops at K = D + 1 and v_j is added after it when D + 2 = 1 (mod 4) (saves a whole MFMA k-step)
__host__ __device__ constexpr bool mm_vsep(int D) { return (D + 2) % 4 == 1; }
__host__ __device__ constexpr int mm_kp(int D) { return mm_vsep(D) ? D + 1 : (D + 2 + 3) / 4 * 4; }
void launch_stamp(hipStream_t st, unsigned long long* dbg, int slot);
void launch_const_rows(hipStream_t st, double* a_row, int nA, double* b_row, int nB, long bs, int npad, int n);   // glue.hip: valid-mask rows of the operand blocks
// reverse pass of one moment-matching step (single rank, D <= 32; the step's prep kernel must precede it on st):
// scratch: rowmom [P][njs][16 ceil((D + 1) / 16)][npad], cpart [P - E][nrb][npad] (njs, nrb from mm_bwd_geometry) and
// part [P][mm_bwd_rc][1 + D + D*D] + [E][mm_bwd_rc][D*D + 2D + 1]; bars = (Mbar | Sbar | Vbar) on the device,
// out [E + P][D + D*D] = per output / per pair contributions (mbar | sbar); their sum in a fixed order goes to
// sum_out [D + D*D] (device-visible, normally pinned host memory); done: a zeroed device counter (left zeroed)
// head [E + P][D*D + D + 2]: the step's D x D inverses (see bwd_head)
void launch_mm_bwd(hipStream_t st, const MMModel& md, const MMWork& wk, double* rowmom, double* cpart, double* part,
                   const double* bars, double* head, double* out, unsigned* done, double* sum_out);
void mm_bwd_geometry(int npad, int Pg, int* njs, int* nrb);   // Pg: pairs of the whole model (rank-count independent split)
// Jacobian tape (bwd.hip).  Per step: launch_mm_sweep runs the reverse sweep in place of the forward pair kernel and
// leaves rowmom / cpart / head in the step's own buffers (sizes below) and N_ab as npart [P][mm_jac_nt][2] tile partials
// for the serial link; once per rollout launch_mm_jac_finish turns the H steps' buffers into the records
// jrec [H][mm_jac_rec_size] (part: [H][mm_jac_part_size] scratch; tape: the rollout tape, whose records start with m_j).
void launch_mm_sweep(hipStream_t st, const MMModel& md, const MMWork& wk, double* rowmom, double* cpart, double* head,
                     double* npart);
// small_nch > 0: the steps ran as the one-launch small step (chunks per pair = small_nch): gpart holds two blocks per
// chunk-workgroup, cpart is unused and the inverses (head) are made here
void launch_mm_jac_finish(hipStream_t st, const MMModel& md, const MMWork& wk, int H, const double* rowmom, const double* cpart,
                          double* head, double* part, const double* tape, size_t tape_stride, double* jrec, int small_nch = 0,
                          const struct RevLocalArgs* rl = nullptr);   // rl: the reverse chain's per-step local quantities ride in the last launch (one more workgroup per step)
size_t mm_jac_rec_size(int D, int E, int P);
size_t mm_jac_part_size(int D, int E, int P, int npad);
size_t mm_jac_rowmom_size(int npad, int P);
size_t mm_bwd_gpart_size(int npad, int P, int D);
size_t mm_bwd_cpart_size(int npad, int P);
size_t mm_jac_cpart_size(int npad, int P, int E);
size_t mm_jac_head_size(int D, int E, int P);
int mm_jac_nt(int npad, int Pg);
int mm_jac_ns(int D);
int mm_bwd_rc(int npad);
// The reverse chain of the policy gradient on the device (rev.hip).
struct RevRewards {
    RewardDev rw[MAX_REWARD_TERMS];
};
struct RevLocalArgs {       // k_rev_local / the extra workgroup of k_mm_jac_fin: trajectory-only quantities of every step (rev_local.h)
    int n, E, U;
    RevRewards rs;
    const double *traj, *Wp, *bp, *maxact;
    double* loc;            // [H][rev_loc_doubles]; nullptr: nothing to do
};
struct RevArgs {
    int E, U, D, H, P;      // P = E (E + 1) / 2: the pairs of the WHOLE model
    // Jacobian records: step t starts at jrec + t * gstep; pair kk of the dealing order ((0,0) .. (E-1,E-1), (1,0), (2,0), (2,1), ..)
    // lives in rank kk % W's block as its pair kk / W: + (kk % W) * gblk + (kk / W) * recp; the E output records at + out_off
    // (rank 0's).  One rank: W = 1, gstep = mm_jac_rec_size, out_off = P * recp.
    const double* jrec;
    int W;
    long gblk, gstep, out_off;
    const double* traj;     // [H + 1][E + E*E]
    const double* tape;     // [H][TS]: m_j | s_j | s1 (E,D) | M (E) | S (E,E) | V (D,E)
    long TS;
    const double* loc;      // [H][rev_loc_doubles]  (k_rev_local: reward gradients, controller / squash forward quantities)
    const double* seeds;    // [H + 1][E + E*E] cotangent seeds of the caller's objective, or nullptr
    const double* Wp;       // LinearController W (U,E)
    const double* reward_dev;   // the rollout's reward on the device: handed out with the gradient (out[..]) instead of a copy of its own in front of the finish
    double* amat;           // [H][rev_mat_doubles]: every step's reverse map [A; B] by columns | r | flags  (k_rev_step -> k_rev_chain)
    double* out;            // [U*E + U + 1 + E + E(E+1)/2 + 1]: dW | db | status (0 fine) | d / d (m_0, S_0 packed) | reward   (device-visible)
};
bool rev_chain_supported(int E, int U, int D);
size_t rev_loc_doubles(int E, int U);
size_t rev_mat_doubles(int E, int U, int D);
RevLocalArgs rev_local_args(int n, const RewardDev* rw, int E, int U, const double* traj, const double* Wp, const double* bp, const double* maxact,
                            double* loc);
void launch_rev_chain(hipStream_t st, const RevArgs& a);
int mm_exp_table_size();   // entries of the 2^(j/n) table the pair kernels were built for
int launch_selftest_mfma(hipStream_t st, double* dbuf, double* hbuf, const double* exp_tab);

}  // namespace pilco
