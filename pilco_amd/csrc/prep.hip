// k_mm_prep: per-step operands of the pair kernel, mean-part and reward workgroups.
#include "prep_kernel.h"

namespace pilco {

size_t prep_lds_bytes(int DT) {
    return sizeof(double) * (prep_region_doubles(DT) + PREP_TAB_DOUBLES);   // + the exp table and the wave sums of the one-launch small step
}


static int device_cus() {
    int cus = 256;
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    return cus;
}
// Row chunks per pair and per output mean part: as many as keep the whole prep grid resident in ONE round (the kernel
// needs the full register file: one workgroup per CU), each chunk a multiple of 64 rows.  One CU is left for the
// reward workgroup; the mean-part workgroups take what the pair workgroups leave (at least one chunk per output).
void mm_prep_chunks(int npad, int PL, int EL, int* nch_out, int* nchm_out) {
    const int cus = device_cus();
    const int nb = npad / 64;
    int nch = 1;
    while (nch * 2 <= nb && nb % (nch * 2) == 0 && PL * nch * 2 + EL + 1 <= cus) nch *= 2;
    int nchm = 1;
    while (nchm * 2 <= nb && nb % (nchm * 2) == 0 && nchm * 2 <= nch && PL * nch + EL * nchm * 2 + 1 <= cus) nchm *= 2;
    *nch_out = nch;
    *nchm_out = nchm;
}

static int prep_dt(int D) {   // the operand kernel's instantiation for this input dimension (see the dispatch below)
    return D <= 4 ? 4 : D <= 6 ? 6 : D <= 8 ? 8 : D <= 10 ? 10 : D == 11 ? 11 : D <= 12 ? 12 : D <= 14 ? 14 : D <= 16 ? 16 : 32;
}
int mm_prep_dt(int D) { return prep_dt(D); }
// Does the fused head (serial link + operands in one workgroup) fit the CU's LDS for this model / policy / reward set?
// Wide inputs (D > 24 with many outputs) do not: the rollout then runs the three-kernel step (same results).
bool mm_fused_head_fits(const MMModel& md, int reward_E, const GlueArgs& ga) {
    const size_t lds_rw = reward_E > 0 ? sizeof(double) * ((size_t)reward_E + (size_t)reward_E * reward_E + reward_lds_doubles(reward_E)) : 0;
    const size_t gd = (glue_lds_doubles_for(ga) + 1) & ~(size_t)1;
    const size_t lds = std::max(prep_lds_bytes(prep_dt(md.D)), lds_rw) + sizeof(double) * gd;
    int dev = 0, lim = 65536;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&lim, hipDeviceAttributeMaxSharedMemoryPerBlock, dev);
    return lds <= (size_t)lim;
}

void launch_mm_prep(hipStream_t st, const MMModel& md, const MMWork& wk, const PrepReward* pr, const GlueArgs* fused) {
    PrepReward none{};
    const PrepReward& r = pr ? *pr : none;
    const int spare = wk.EL * wk.NCHM + (r.n > 0 ? 1 : 0);   // mean-part workgroups, then the reward workgroup
    const int gy = wk.NCH * (wk.fuse_pair && wk.NCS > 1 ? wk.NCS : 1);   // (small step: column splits, see MMWork::NCS)
    dim3 grid(wk.PL + (spare + gy - 1) / gy, gy);
    const int D = md.D;
    const size_t lds_rw = r.n > 0 ? sizeof(double) * ((size_t)r.E + (size_t)r.E * r.E + reward_lds_doubles(r.E)) : 0;
    GlueArgs gnone{};
    const GlueArgs& ga = fused ? *fused : gnone;
    const int gd = fused ? (int)((glue_lds_doubles_for(ga) + 1) & ~(size_t)1) : 0;   // glue region of the fused head (even: 16-byte alignment)
    int dev_ = 0;
    (void)hipGetDevice(&dev_);
    // the fused head is instantiated per controller kind and for one rank / several (see glue_body): the serial link is a
    // chain of latencies through straight-line code, and code that is merely present in its stream costs microseconds
    const bool multi = fused && (ga.xq != nullptr || ga.xq_peers != nullptr || ga.wk.nranks != 1 ||
                                 (ga.flags & (GF_PACK | GF_ASSEMBLE | GF_PROPAGATE)) == (GF_ASSEMBLE | GF_PROPAGATE));
    const int pk = !fused ? 0 : (ga.pol_kind == PILCO_POLICY_RBF ? (ga.pol_inline ? 2 : 1) : (ga.pol_kind == PILCO_POLICY_LINEAR ? 3 : 0));
    const PrepLaunch a{st, grid, lds_rw, gd, dev_, fused != nullptr, multi, pk, &md, &wk, &r, &ga};
    // DT = D where it matters: the Gauss-Jordan costs 2 DT readlanes per pivot and DT pivots, a row DT^2 FMAs -- at
    // D = 10 the exact instantiation does 30 % less work on this latency-bound path than the padded DT = 12
    if (D <= 4) launch_prep_4(a);
    else if (D <= 6) launch_prep_6(a);
    else if (D <= 8) launch_prep_8(a);
    else if (D <= 10) launch_prep_10(a);
    else if (D == 11) launch_prep_11(a);
    else if (D <= 12) launch_prep_12(a);
    else if (D <= 14) launch_prep_14(a);
    else if (D <= 16) launch_prep_16(a);
    else launch_prep_32(a);   // (17 <= D <= 32: one instantiation; rounds 1-3 also carried DT = 20 and 24, 2 MB of code for inputs no example has)
}

}  // namespace pilco
