// k_mm_prep (the per-step operand kernel / fused head) and the per-DT launch helper.  The 8 instantiations per input
// dimension DT (controller kind x rank layout, see glue_body) are compiled in SEPARATE translation units (prep_dt_*.hip:
// one group of DTs each) so that the library still builds in about a minute; prep.hip holds the host-side dispatch.
#pragma once
#include "prep_device.h"

namespace pilco {

#ifndef PREP_SPARE_FIRST
#define PREP_SPARE_FIRST 1
#endif
#ifndef PREP_PRE
#define PREP_PRE 1   // the operand work's first phase is prepared BEFORE the link (0: A/B builds)
#endif
size_t prep_lds_bytes(int DT);

// ------------------------------------------------------------------ prep
// PK: the controller code compiled into the fused head's link (glue_body<PK, SR>): 0 none, 3 linear, 1 RBF from its own
// launches, 2 RBF inline; SR: single rank (no peer exchange, no gathered segments).  The plain operand kernel (FUSED = false)
// has no link and exists for <0, true> only.
// OCC2: the same kernel held to 128 registers (four waves per SIMD; it spills 17-22 registers at DT <= 6, 51 at DT = 8), so that
// TWO workgroups fit a CU.  Launched for the lanes of a batch call (MMWork::share_cu), whose launches run side by side: they
// then share the CUs instead of queueing for them -- the serial link is latency, not issue slots.  Same arithmetic, same bits.
// Measured at config-5 size (tools/restart_lanes_bench.py): three value-and-gradient lanes 1.96 -> 1.42 ms (linear) /
// 3.10 -> 2.35 ms (RBF); a SOLO call loses 3-5 % to the spills (forward pair phase 2.6 -> 3.3 us), hence two instantiations.
template <int DT, bool FUSED, int PK = 0, bool SR = true, bool OCC2 = false>
__global__ __launch_bounds__(512, OCC2 ? 4 : 2) void k_mm_prep(MMModel md, MMWork wk, PrepReward pr, GlueArgs g, int glue_doubles) {
    extern __shared__ __attribute__((aligned(16))) double sm_all[];
    kernarg_warm<(int)(sizeof(MMModel) + sizeof(MMWork) + sizeof(PrepReward) + sizeof(GlueArgs)) + 8 + 64>();   // (+ the hidden grid / group sizes behind them)
    // FUSED: the serial link of the previous step runs first, redundantly in every workgroup (see glue_device.h); it
    // leaves the joint Gaussian of THIS step (L.jm, L.js) and the current state (L.mx, L.sx) in the first glue_doubles
    // doubles of LDS.  (All LDS pointers below are derived from sm_all unconditionally: no shared/global pointer merges.)
    GlueLds L;
    glue_lds_carve(g, sm_all, L);
    // The model constants this workgroup needs (its lengthscales and signal variances) are requested BEFORE the serial
    // link, so that their memory round trip overlaps with it instead of following it.
    // Item column of this workgroup: the spare columns (mean parts, reward) are dispatched FIRST -- their workgroups run
    // 1.5-2 us longer than the pair workgroups of the same launch, and the launch ends with its last workgroup
    const int bxi = PREP_SPARE_FIRST ? (int)((blockIdx.x + (unsigned)wk.PL) % gridDim.x) : (int)blockIdx.x;
    const bool spare_wg = bxi >= wk.PL;
    const int spare_idx = (bxi - wk.PL) * (int)gridDim.y + (int)blockIdx.y;
    const bool mean_wg = spare_wg && spare_idx < wk.EL * wk.NCHM;
    int a = 0, b = 0;
    if (!spare_wg) local_pair_ab(wk, md.E, bxi, a, b);
    else if (mean_wg) a = b = (spare_idx >> __builtin_ctz(wk.NCHM)) * wk.nranks + wk.rank;   // the owner of (a,a) owns output a
    double pre_la = 1.0, pre_lb = 1.0, pre_var = 1.0;
    if (!spare_wg || mean_wg) {
        if ((int)threadIdx.x < md.D) {
            pre_la = md.ls[a * md.D + (int)threadIdx.x];
            pre_lb = md.ls[b * md.D + (int)threadIdx.x];
        }
        pre_var = mean_wg ? md.var[a] : md.lvar[(threadIdx.x >> 8) ? b : a];   // (pair workgroups want log var of their side, the mean part var_a)
    }
    if (FUSED && SR && PK != 1 && wk.fuse_pair && !spare_wg) {   // one-launch small step: the pair phase's exp table, on its way during the link
        double* tabL = sm_all + glue_doubles + prep_region_doubles(DT);
        for (int e = threadIdx.x; e < FEXP_TN; e += 512) tabL[e] = wk.exp_tab[e];
    }
    // (the link's results are stored by the first pair workgroup; handing that to an idle slot of the spare columns -- a
    // workgroup with nothing else to do -- measured 1-2 % SLOWER on every configuration: docs/dead_ends.md)
    // The first phase of the operand work -- reciprocal lengthscales, zeroed Q / T, the joint Gaussian in the region's own
    // layout -- does not wait for the link: its constants are written here, BEFORE the link, and the joint Gaussian is stored
    // there by the link's last phase (GlueLds::xm / xs), under the link's own closing barrier.  One barrier interval of the
    // step's serial path less.  (PK = 1, an RbfController's own launches: the policy head reads the state instead; it keeps the copy.)
    constexpr bool PRE = FUSED && PK != 1 && (PREP_PRE != 0);
    if constexpr (PRE) {
        double* sm = sm_all + glue_doubles;
        const int t = threadIdx.x, D = md.D;
        if (!spare_wg) {   // prep_work's layout: s_m [DT] | s_ia2 [DT] | s_ib2 [DT] | s_s [DT*DT] | s_Q [DT*DT]
            if (t < DT) {
                if (t >= D) sm[t] = 0.0;
                sm[DT + t] = (t < D) ? 1.0 / (pre_la * pre_la) : 0.0;
                sm[2 * DT + t] = (t < D) ? 1.0 / (pre_lb * pre_lb) : 0.0;
            }
            for (int e = t; e < DT * DT; e += 512) sm[3 * DT + DT * DT + e] = 0.0;
            prep_wt_constants<DT>(md, wk, bxi, sm);
            L.xm = sm;
            L.xs = sm + 3 * DT;
        } else if (mean_wg) {   // prep_mean_block's layout: s_m [DT] | s_ia [DT] | s_s [DT*DT] | s_T [DT*DT]
            if (t < DT) {
                if (t >= D) sm[t] = 0.0;
                sm[DT + t] = (t < D) ? 1.0 / pre_la : 0.0;
            }
            for (int e = t; e < DT * DT; e += 512) sm[2 * DT + DT * DT + e] = 0.0;
            L.xm = sm;
            L.xs = sm + 2 * DT;
        }
    }
    if (FUSED) glue_body<PK, SR>(g, L, bxi == 0 && blockIdx.y == 0);
    prep_work<DT, FUSED, 512, FUSED && SR && PK != 1, PRE>(md, wk, pr, g, L, sm_all, glue_doubles, bxi, (int)blockIdx.y, (int)gridDim.x, (int)gridDim.y, a, b, pre_la,
                              pre_lb, pre_var);
}

// arguments of one head launch, as launch_mm_prep has prepared them
struct PrepLaunch {
    hipStream_t st;
    dim3 grid;
    size_t lds_rw;    // LDS bytes of the reward workgroup (0: none)
    int gd;           // doubles of the link's LDS region (0: plain operand kernel)
    int dev;
    bool fused, multi;
    int pk;
    const MMModel* md;
    const MMWork* wk;
    const PrepReward* r;
    const GlueArgs* ga;
};

template <int DT>
void launch_prep_dt(const PrepLaunch& a) {
    const hipStream_t st = a.st;
    const dim3 grid = a.grid;
    const size_t lds_rw = a.lds_rw;
    const int gd = a.gd, dev_ = a.dev, pk = a.pk;
    const bool fused = a.fused, multi = a.multi;
    const MMModel& md = *a.md;
    const MMWork& wk = *a.wk;
    const PrepReward& r = *a.r;
    const GlueArgs& ga = *a.ga;
#define PREP2(DT_, F_, PK_, SR_, O_)                                                                       \
    do {                                                                                                   \
        const size_t lds_ = std::max(prep_lds_bytes(DT_), lds_rw) + sizeof(double) * (size_t)gd;           \
        static size_t configured_[64] = {};  /* beyond the default dynamic-LDS limit: opt in once PER DEVICE */ \
        size_t& conf_ = configured_[dev_ & 63];                                                            \
        if (conf_ == 0) conf_ = 48 * 1024;                                                                 \
        if (lds_ > conf_) {                                                                                \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_mm_prep<DT_, F_, PK_, SR_, O_>),     \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_);              \
            conf_ = lds_;                                                                                  \
        }                                                                                                  \
        hipLaunchKernelGGL((k_mm_prep<DT_, F_, PK_, SR_, O_>), grid, dim3(512), lds_, st, md, wk, r, ga, gd); \
    } while (0)
    /* the two-per-CU build exists for the single-rank heads with a controller at DT <= 8 (the lanes of a gradient batch) */
#define PREP1(DT_, F_, PK_, SR_)                                                                           \
    do {                                                                                                   \
        if constexpr (F_ && SR_ && (PK_ == 2 || PK_ == 3) && DT_ <= 8) {                                   \
            if (wk.share_cu) PREP2(DT_, F_, PK_, SR_, true);                                               \
            else PREP2(DT_, F_, PK_, SR_, false);                                                          \
        } else {                                                                                           \
            PREP2(DT_, F_, PK_, SR_, false);                                                               \
        }                                                                                                  \
    } while (0)
#define PREP(DT_)                                              \
    do {                                                       \
        if (!fused) PREP1(DT_, false, 0, true);                \
        else if (multi) {   /* sharded rollouts: none / linear / inline RBF */ \
            if (pk == 0) PREP1(DT_, true, 0, false);           \
            else if (pk == 3) PREP1(DT_, true, 3, false);      \
            else PREP1(DT_, true, 2, false);                   \
        } else if (pk == 0) PREP1(DT_, true, 0, true);         \
        else if (pk == 3) PREP1(DT_, true, 3, true);           \
        else if (pk == 1) PREP1(DT_, true, 1, true);           \
        else PREP1(DT_, true, 2, true);                        \
    } while (0)
    PREP(DT);
#undef PREP
#undef PREP1
#undef PREP2
}

// defined in prep_dt_*.hip
void launch_prep_4(const PrepLaunch& a);
void launch_prep_6(const PrepLaunch& a);
void launch_prep_8(const PrepLaunch& a);
void launch_prep_10(const PrepLaunch& a);
void launch_prep_11(const PrepLaunch& a);
void launch_prep_12(const PrepLaunch& a);
void launch_prep_14(const PrepLaunch& a);
void launch_prep_16(const PrepLaunch& a);
void launch_prep_32(const PrepLaunch& a);

}  // namespace pilco
