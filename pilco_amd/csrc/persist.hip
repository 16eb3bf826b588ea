// k_rollout_persist: the whole H-step rollout (PILCO.predict, pilco/models/pilco.py:118-136 -- the reference's
// tf.while_loop) as ONE launch.  One workgroup of 12 waves per CU stays resident for all H steps and runs, per step, the
// three phases the launch sequence of rollout.hip runs as two kernels:
//
//   link      (glue_body, glue_device.h)  every workgroup, redundantly: reduce the pair sums of step h - 1, assemble
//             (M, S, V), propagate, controller, joint Gaussian -- the state and [s_x, s_x c_xu] stay in this workgroup's LDS;
//   operands  (prep_work, prep_device.h)  workgroup b owns item b of the operand grid for the whole rollout: the operands of
//             one (pair, row chunk), the mean part of one (output, row chunk), or the reward;
//   pairs     (sk_wave_range, pair_device.h)  wave w owns range w of the stream-K cost line -- the SAME decomposition as
//             k_mm_pair_sk, so every partial sum, and with it every bit of the result, equals the launch sequence's.
//
// Phase order between workgroups comes from flags in global memory, not from kernel boundaries:
//   ready[h][pair][chunk]  raised by the operand workgroup; polled by the waves whose range touches that pair;
//   done[h][workgroup]     raised when a workgroup has finished step h (its operand item AND its waves); the next link
//                          waits for all of them.
// ONE wave per workgroup does the polling, on dense flag arrays (a few cache lines per poll).
// No cache maintenance: a device-wide release / acquire (buffer_wbl2 / buffer_inv sc1) costs ~10 us on this chip
// (profiles/r03_ubench_sync.txt), a flag round ~1.4 us.  Instead everything a step hands to other workgroups is written
// with write-through (sc1) stores followed by s_waitcnt vmcnt(0) before the flag, and is read, after the flag, with
// ordinary loads from addresses that NOBODY has loaded before in this launch: every step has its own operand / partial /
// flag buffers (PersistArgs strides), so neither a CU's L1 nor an XCD's L2 can hold a stale line, and the column operands
// keep their 32-fold reuse out of the L2.  tools/ubench_coherence.hip exercises exactly this protocol (including 16
// writers of different XCDs per cache line): profiles/r03_ubench_coherence.txt.
// Every wait is bounded by the wall clock: a workgroup that is not resident (the GPU shared with another process) or a
// lost flag makes the launch give up -- the abort word is set, every later wait returns at once, the host reports it and
// the rollout is repeated on the launch-sequence path (rollout.hip).  Nothing can hang.
#include "pair_device.h"
#include "prep_device.h"

namespace pilco {

constexpr int PERSIST_THREADS = 768;   // 12 waves = 3 per SIMD: the occupancy the pair tiles are tuned for (156-163 VGPRs)

typedef unsigned long long u64;

__device__ __forceinline__ u64 pflag_ld(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void pflag_st(u64* p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Spin (write-through loads, no cache maintenance) until *p shows this launch's epoch.  false: the launch has given up.
__device__ __forceinline__ bool pflag_wait(const u64* p, const PersistArgs& a) {
    if (pflag_ld(p) == a.epoch) return true;
    const u64 t0 = wall_clock64();
    for (int it = 1;; ++it) {
        if (pflag_ld(p) == a.epoch) return true;
        if ((it & 31) == 0) {
            if (pflag_ld(a.ctl) == a.epoch) return false;   // somebody gave up: nobody waits any more
            if (wall_clock64() - t0 > a.timeout_ticks) {
                pflag_st(a.ctl, a.epoch);
                return false;
            }
        }
        __builtin_amdgcn_s_sleep(1);
    }
}
// this thread's write-through stores have reached memory (s_waitcnt vmcnt(0); workgroup scope: no cache maintenance)
__device__ __forceinline__ void stores_done() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); }
// nothing below may be hoisted above the flag that was just seen (compiler ordering; the hardware issues loads in order)
__device__ __forceinline__ void after_flag() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); }

// The kernel's arguments are ~1.5 KB of wave-uniform values.  Read through the plain by-value parameter, every one of
// them is loop invariant: the compiler loads them all ahead of the step loop and, with ~100 SGPRs, spills hundreds of
// them (SGPR -> VGPR lanes -> scratch; the first build carried 1 KB of scratch per lane and its reloads sat on the
// serial path of every step).  kargs() hands out the kernel-argument segment behind an offset the compiler cannot see
// through (an s_mov of 0 in volatile asm): loads stay scalar (constant address space), stay inside the phase that calls
// kargs(), and die with it.
__device__ __forceinline__ const PersistArgs& kargs() {
    unsigned off;
    asm volatile("s_mov_b32 %0, 0" : "=s"(off));
    typedef const char __attribute__((address_space(4)))* cptr4;
    cptr4 p = (cptr4)__builtin_amdgcn_kernarg_segment_ptr();
    return *(const PersistArgs*)(p + off);
}

// PK: the controller code compiled into the link (glue_body<PK, true>: 0 none, 3 LinearController) -- the kernel walks
// its whole instruction stream once per step, and the CU pair's 64 KB instruction cache is the resource to fit.
template <int DT, int KC, bool VSEP, int PK>
__global__ __launch_bounds__(PERSIST_THREADS) void k_rollout_persist(PersistArgs a_unused) {
    (void)a_unused;   // (declares the layout of the kernel-argument segment; read through kargs())
    extern __shared__ __attribute__((aligned(16))) double sm_all[];
    __shared__ double tab[FEXP_TN];
    __shared__ double s_pre[2 * MAX_D + 2];   // this workgroup's item: lengthscales of its outputs a and b, their signal variances
    __shared__ int s_geo[16];                 // bx, by, first wave, waves, .., has item, interleaved; [8..13] ready-flag ranges of the 3 wave groups
    const int t = threadIdx.x;
    {   // ---- once: exp table, this workgroup's place on the stream-K line and its item of the operand phase
        const PersistArgs& a = kargs();
        const MMWork& wk0 = a.g.wk;
        const int nwg = gridDim.x, b = blockIdx.x;
        for (int e = t; e < FEXP_TN; e += PERSIST_THREADS) tab[e] = wk0.exp_tab[e];
        // the XCD-aware order of k_mm_pair_sk: workgroups are dealt round-robin over the 8 XCDs; the waves of one XCD
        // cover one contiguous eighth of the line
        int bpos = b;
        if ((nwg & 7) == 0) bpos = (b & 7) * (nwg >> 3) + (b >> 3);
        const int wpw = (wk0.sk_waves + nwg - 1) / nwg;   // waves per workgroup that carry a range (<= 12)
        const int gy = wk0.NCH;
        const int bx = b / gy, by = b - bx * gy;
        const bool has_item = bx < a.nitems_x;
        const bool spare_wg = bx >= wk0.PL;
        const int spare_idx = (bx - wk0.PL) * gy + by;
        const bool mean_wg = spare_wg && spare_idx < wk0.EL * wk0.NCHM;
        int oa = 0, ob = 0;
        if (!spare_wg) local_pair_ab(wk0, a.md.E, bx, oa, ob);
        else if (mean_wg) oa = ob = (spare_idx / wk0.NCHM) * wk0.nranks + wk0.rank;
        const bool has_model = has_item && (!spare_wg || mean_wg);
        if (t < MAX_D) {
            s_pre[t] = (has_model && t < a.md.D) ? a.md.ls[oa * a.md.D + t] : 1.0;
            s_pre[MAX_D + t] = (has_model && t < a.md.D) ? a.md.ls[ob * a.md.D + t] : 1.0;
        }
        if (t < 2) s_pre[2 * MAX_D + t] = has_model ? a.md.var[t ? ob : oa] : 1.0;
        if (t == 0) {
            // Which waves of the stream-K line are this workgroup's?  Consecutive waves of the line do the same kind of
            // work (a diagonal pair streams iK, an off-diagonal one does not): a CU that held twelve consecutive waves would
            // be all-streaming or all-compute, and the streaming CUs finish ~20 us after the others (measured).  The launch
            // sequence's kernel puts three 4-wave workgroups from three distant places of its XCD's eighth on every CU; the
            // same here: wave group j (waves 4j .. 4j+3) of the workgroup at place q of XCD x takes the four consecutive waves
            // of virtual workgroup q + (workgroups per XCD) * j of that XCD.  (Lines shorter than 12 waves per workgroup:
            // consecutive waves.)
            const bool full = ((nwg & 7) == 0) && wk0.sk_waves == nwg * 12;
            s_geo[7] = full ? 1 : 0;
            int fl[6] = {0, 0, 0, 0, 0, 0};
            const int nd_steps = wk0.sk_nd * wk0.sk_tdiag;
            for (int j = 0; j < 3; ++j) {
                int w_first, w_last;   // [w_first, w_last) of wave group j
                if (full) {
                    const int x = b & 7, q = b >> 3, per = nwg >> 3;
                    w_first = (x * 3 * per + q + per * j) * 4;
                    w_last = w_first + 4;
                } else {
                    w_first = j == 0 ? bpos * wpw : 0;
                    w_last = j == 0 ? min(w_first + wpw, wk0.sk_waves) : 0;
                }
                // ready flags the group depends on: the pairs between its first and its last step ([pair][chunk], dense)
                if (w_first < w_last) {
                    const int s_lo = sk_boundary(wk0, w_first), s_hi = sk_boundary(wk0, w_last) - 1;
                    const int p_lo = s_lo < nd_steps ? s_lo / wk0.sk_tdiag : wk0.sk_nd + (s_lo - nd_steps) / wk0.sk_toff;
                    const int p_hi = s_hi < nd_steps ? s_hi / wk0.sk_tdiag : wk0.sk_nd + (s_hi - nd_steps) / wk0.sk_toff;
                    if (s_hi >= s_lo) {
                        fl[2 * j] = p_lo * gy;
                        fl[2 * j + 1] = (p_hi + 1) * gy;
                    }
                }
            }
            s_geo[0] = bx;
            s_geo[1] = by;
            s_geo[2] = full ? ((b & 7) * 3 * (nwg >> 3) + (b >> 3)) * 4 : bpos * wpw;   // first wave of group 0
            s_geo[3] = full ? 12 : max(min(bpos * wpw + wpw, wk0.sk_waves) - bpos * wpw, 0);
            s_geo[6] = has_item ? 1 : 0;
            for (int k = 0; k < 6; ++k) s_geo[8 + k] = fl[k];
        }
    }
    __syncthreads();
    const int H = kargs().H;
    // developer stamps (100 MHz wall clock) of one step into the control words: [8..15] workgroup 0, [16..23] the last workgroup
    // with an item (mean part / reward), [24..31] the last workgroup
#define PSTAMP(k_)                                                                                                      \
    do {                                                                                                                \
        if (t == 0 && (h == hs || (h == hs + 1 && (k_) == 0))) {                                                       \
            const int sl_ = blockIdx.x == 0 ? 8 : ((int)blockIdx.x == kargs().nitems_x * kargs().g.wk.NCH - 1 ? 16 : ((int)blockIdx.x == (int)gridDim.x - 1 ? 24 : -1)); \
            if (sl_ >= 0) kargs().ctl[sl_ + (k_) + (h == hs + 1 ? 6 : 0)] = wall_clock64();                               \
        }                                                                                                               \
    } while (0)
    const int hs = H > 4 ? 3 : H - 1;
    for (int h = 0; h <= H; ++h) {
        const bool closing = (h == H);
        if (closing && blockIdx.x != 0) break;   // the closing link (state H) is workgroup 0's alone
        {   // ---- all workgroups have finished step h - 1 (its partial sums, mean parts and 1/sqrt(det R) are in memory), then
            // the serial link: (pack / assemble / propagate of step h - 1 ->) state h -> controller -> joint Gaussian of step h
            const PersistArgs& a = kargs();
            const int nwg = gridDim.x;
            if (h > 0) {
                // ONE wave per workgroup watches the (dense) done flags -- thousands of lanes polling write-through loads
                // flood the fabric and slow the very stores they wait for (tools/ubench_sync.hip, "pair flags")
                if (t < 64)
                    for (int k = t; k < nwg; k += 64) (void)pflag_wait(a.done + (size_t)(h - 1) * nwg + k, a);
                after_flag();
                __syncthreads();
            }
            PSTAMP(0);
            const MMWork& wk0 = a.g.wk;
            const int E = a.g.E;
            GlueArgs g = a.g;
            g.step = h;
            g.dbg_off = h > 0 ? 48 : 0;
            if (h > 0) {   // read side: the buffers step h - 1 wrote
                g.wk.pair_isdet = wk0.pair_isdet + (size_t)(h - 1) * a.sSmall;
                g.wk.mean_part = wk0.mean_part + (size_t)(h - 1) * a.sSmall;
                g.wk.sk_part = wk0.sk_part + (size_t)(h - 1) * a.sPart;
            }
            g.flags = GF_TRAJ | (closing ? 0 : GF_POLICY) | (h > 0 ? (GF_PACK | GF_ASSEMBLE | GF_PROPAGATE) : 0);
            g.m_x = a.st[h > 0 ? (h - 1) & 1 : 0];
            g.s_x = g.m_x + E;
            g.m_out = h > 0 ? a.st[h & 1] : nullptr;
            g.s_out = h > 0 ? a.st[h & 1] + E : nullptr;
            g.s1 = a.s1b[(h + 1) & 1];
            g.s1_out = closing ? nullptr : a.s1b[h & 1];
            g.lds_state = h > 0 ? 2 : 1;
            GlueLds L;
            glue_lds_carve(g, sm_all, L);
            glue_body<PK, true>(g, L, blockIdx.x == 0);
            PSTAMP(1);
        }
        if (closing) break;
        {   // ---- operands of step h (write side: this step's own buffers)
            const PersistArgs& a = kargs();
            const MMWork& wk0 = a.g.wk;
            const int bx = s_geo[0], by = s_geo[1];
            if (s_geo[6]) {
                GlueArgs g = a.g;   // (prep_work reads flags only: the operands are built for the joint Gaussian in LDS)
                g.flags = GF_TRAJ | GF_POLICY | GF_PACK | GF_ASSEMBLE | GF_PROPAGATE;
                GlueLds L;
                glue_lds_carve(g, sm_all, L);
                MMWork wkh = wk0;
                wkh.At = wk0.At + (size_t)h * a.sAt;
                wkh.Bt = wk0.Bt + (size_t)h * a.sBt;
                wkh.vcol = wk0.vcol + (size_t)h * a.sBt;
                wkh.pair_isdet = wk0.pair_isdet + (size_t)h * a.sSmall;
                wkh.mean_part = wk0.mean_part + (size_t)h * a.sSmall;
                const double pre_la = (t < MAX_D) ? s_pre[t] : 1.0, pre_lb = (t < MAX_D) ? s_pre[MAX_D + t] : 1.0;
                const double pre_var = s_pre[2 * MAX_D + ((t >> 8) ? 1 : 0)];
                prep_work<DT, true, PERSIST_THREADS>(a.md, wkh, a.pr, g, L, sm_all, a.glue_doubles, bx, by, a.nitems_x, wk0.NCH, pre_la, pre_lb, pre_var);
            }
            PSTAMP(2);
            stores_done();
            __syncthreads();
            if (t == 0 && bx < wk0.PL) pflag_st(a.ready + ((size_t)h * wk0.PL + bx) * wk0.NCH + by, a.epoch);
            // ---- the operands of the pairs this workgroup's waves touch are in memory (one wave polls for all twelve)
            if (t < 64) {
                const u64* rdy = a.ready + (size_t)h * wk0.PL * wk0.NCH;
                for (int j = 0; j < 3; ++j)
                    for (int k = s_geo[8 + 2 * j] + t; k < s_geo[9 + 2 * j]; k += 64) (void)pflag_wait(rdy + k, a);
            }
            after_flag();
            __syncthreads();
            PSTAMP(3);
        }
        {   // ---- pair sums of step h: this wave's range of the stream-K line
            const PersistArgs& a = kargs();
            const int wv = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63;
            if (wv < s_geo[3]) {
                const MMWork& wk0 = a.g.wk;
                // (interleaved: wave group wv / 4 sits (workgroups per XCD) virtual workgroups further along the line)
                const int w = s_geo[2] + (s_geo[7] ? (wv >> 2) * (int)(gridDim.x >> 3) * 4 + (wv & 3) : wv);
                MMWork wkh = wk0;
                wkh.At = wk0.At + (size_t)h * a.sAt;
                wkh.Bt = wk0.Bt + (size_t)h * a.sBt;
                wkh.vcol = wk0.vcol + (size_t)h * a.sBt;
                wkh.sk_part = wk0.sk_part + (size_t)h * a.sPart;
                double out0, out1;
                int p0, p1;
                sk_wave_range<KC, VSEP, true>(a.md, wkh, tab, w, lane, [](int) {}, out0, out1, p0, p1);   // (fenced MFMA results: see pair_wave)
                for (int off = 32; off > 0; off >>= 1) {
                    out0 += __shfl_down(out0, off);
                    out1 += __shfl_down(out1, off);
                }
                if (lane == 0) {   // the slot-major partial layout of k_mm_pair_sk, written through
                    const int nd_steps = wkh.sk_nd * wkh.sk_tdiag;
                    if (p0 >= 0) {
                        const long S0 = (p0 < wkh.sk_nd) ? (long)p0 * wkh.sk_tdiag : (long)wkh.sk_nd * wkh.sk_tdiag + (long)(p0 - wkh.sk_nd) * wkh.sk_toff;
                        const int wlo = sk_wave_of(S0, wkh.sk_waves, nd_steps, wkh.sk_total, wkh.sk_ud, wkh.sk_uo);
                        store_wt(&wkh.sk_part[(long)(w - wlo) * wkh.sk_pls + p0], out0);
                    }
                    if (p1 >= 0) store_wt(&wkh.sk_part[p1], out1);
                }
            }
            // ---- this workgroup has finished step h
            PSTAMP(4);
            stores_done();
            __syncthreads();
            PSTAMP(5);
            if (t == 0) pflag_st(a.done + (size_t)h * gridDim.x + blockIdx.x, a.epoch);
        }
    }
}

// instantiations: input dimension D -> (operand-kernel DT, MFMA k-steps KC = KP / 4, vsep) as launch_mm_prep /
// launch_mm_pair choose them; D <= 12 (beyond that the operand phase needs more registers than 3 waves per SIMD leave)
bool mm_persist_supported(int D, int KP, bool vsep) {
    if (D < 1 || D > 12) return false;
    return KP == mm_kp(D) && vsep == mm_vsep(D);
}

size_t mm_persist_lds_bytes(const MMModel& md, const GlueArgs& g, int reward_E) {
    const int D = md.D;
    const int DT = D <= 4 ? 4 : D <= 6 ? 6 : D <= 8 ? 8 : D <= 10 ? 10 : D == 11 ? 11 : 12;
    const size_t pair_blk = (size_t)4 * DT + 2 * (size_t)DT * DT + 4 + 256 * (size_t)(DT + 1);
    const size_t mean_blk = (size_t)2 * DT + 2 * (size_t)DT * DT + 4 + 9 * (size_t)(DT + 1) + 2 * (size_t)DT + 512 * (size_t)(DT + 2);
    const size_t rw = reward_E > 0 ? (size_t)reward_E + (size_t)reward_E * reward_E + reward_lds_doubles(reward_E) : 0;
    const size_t gd = (glue_lds_doubles_for(g) + 1) & ~(size_t)1;
    return sizeof(double) * (gd + std::max(std::max(pair_blk, mean_blk), rw));
}

int launch_rollout_persist(hipStream_t st, const PersistArgs& a, int workgroups, size_t lds) {
    int dev = 0;
    (void)hipGetDevice(&dev);
#define PERSIST1(DT_, KC_, VS_, PK_)                                                                                       \
    do {                                                                                                                   \
        static size_t configured_[64] = {};                                                                                \
        size_t& conf_ = configured_[dev & 63];                                                                             \
        if (conf_ == 0) conf_ = 48 * 1024;                                                                                 \
        if (lds > conf_) {                                                                                                 \
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_rollout_persist<DT_, KC_, VS_, PK_>),                  \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)                   \
                return -1;                                                                                                 \
            conf_ = lds;                                                                                                   \
        }                                                                                                                  \
        hipLaunchKernelGGL((k_rollout_persist<DT_, KC_, VS_, PK_>), dim3(workgroups), dim3(PERSIST_THREADS), lds, st, a);  \
        return 0;                                                                                                          \
    } while (0)
#define PERSIST(DT_, KC_, VS_)                                                  \
    do {                                                                        \
        if (a.g.pol_kind == PILCO_POLICY_LINEAR) PERSIST1(DT_, KC_, VS_, 3);    \
        else PERSIST1(DT_, KC_, VS_, 0);                                        \
    } while (0)
    switch (a.md.D) {
        case 1: case 2: PERSIST(4, 1, false);
        case 3: PERSIST(4, 1, true);
        case 4: PERSIST(4, 2, false);
        case 5: case 6: PERSIST(6, 2, false);
        case 7: PERSIST(8, 2, true);
        case 8: PERSIST(8, 3, false);
        case 9: case 10: PERSIST(10, 3, false);
        case 11: PERSIST(11, 3, true);
        case 12: PERSIST(12, 4, false);
        default: return -1;
    }
#undef PERSIST
#undef PERSIST1
}

}  // namespace pilco
