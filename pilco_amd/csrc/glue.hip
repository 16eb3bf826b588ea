// k_glue: the serial link of a horizon step (pack, assemble, propagate, controller, joint Gaussian).
#include "glue_device.h"

namespace pilco {

static size_t glue_lds_doubles(int E, int D, int SEG, int mp) {
    const int nm = E > D ? E : D;
    const size_t tail = (size_t)SEG + (size_t)mp;
    const size_t rew = reward_lds_doubles(E);
    return (size_t)3 * nm + 7 * (size_t)nm * nm + 256 + (tail > rew ? tail : rew);
}
size_t glue_lds_bytes(int E, int D) { return sizeof(double) * glue_lds_doubles(E, D, 0, 0); }
size_t glue_lds_doubles_for(const GlueArgs& g) {
    int mp_n = (g.flags & GF_PACK) ? g.wk.EL * g.wk.NCHM * (1 + g.D) : 0;
    int seg_n = (g.flags & (GF_PACK | GF_ASSEMBLE)) ? g.wk.SEG * ((g.flags & GF_PACK) ? 1 : g.wk.nranks) : 0;
    if (g.flags & GF_RBF_POST) {
        mp_n = g.pwk.EL * g.pwk.NCHM * (1 + g.E);
        seg_n = g.pwk.SEG;
    }
    return glue_lds_doubles(g.E, g.D, seg_n, mp_n) + (((g.flags & GF_POLICY) && g.pol_inline) ? (size_t)g.pol_lds : 0);
}

int rbf_inline_lds_doubles(int E, int U, int bf) {
    if (E < 1 || E > 16 || U < 1 || U > 4 || bf < 1 || bf > 256) return 0;
    const int P = U * (U + 1) / 2;
    if ((long)P * bf * bf > 16384) return 0;   // the O(bf^2) sums run redundantly in every workgroup of the head: keep them short
    const RbfInlineLayout lay = rbf_inline_layout(E, U, bf);
    return lay.total <= 8192 ? lay.total : 0;
}

// ones in the first n entries of row `a_row` of nA blocks and of row `b_row` of nB blocks (block stride `bs` doubles): the
// valid masks of the operand blocks (api.hip: build_work)
__global__ __launch_bounds__(256) void k_const_rows(double* a_row, int nA, double* b_row, int nB, long bs, int npad, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x, blk = blockIdx.y;
    if (i >= npad) return;
    const double v = i < n ? 1.0 : 0.0;
    if (a_row && blk < nA) a_row[(long)blk * bs + i] = v;
    if (blk < nB) b_row[(long)blk * bs + i] = v;
}
void launch_const_rows(hipStream_t st, double* a_row, int nA, double* b_row, int nB, long bs, int npad, int n) {
    const int nb = nA > nB ? nA : nB;
    if (nb <= 0) return;
    hipLaunchKernelGGL(k_const_rows, dim3((npad + 255) / 256, nb), dim3(256), 0, st, a_row, nA, b_row, nB, bs, npad, n);
}

__global__ __launch_bounds__(256) void k_glue(GlueArgs g) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    kernarg_warm<(int)sizeof(GlueArgs) + 64>();
    const int E = g.E, t = threadIdx.x;
    GlueLds L;
    glue_lds_carve(g, sm, L);
    if (blockIdx.x == 1) {
        // Workgroup 1: reward of the current (pre-propagation) state (rewards.py:19-81), evaluated
        // concurrently with workgroup 0; the state is double-buffered so there is no race.
        DBG_STAMP(g.wk, 20, t == 0);
        double* ws = L.seg;  // scratch: this workgroup uses none of the pack / assemble storage
        const LoadSeg sg[2] = {{0, g.m_x, E}, {L.o_sx, g.s_x, E * E}};
        multi_load<2, 4>(L.mx, sg);
        __syncthreads();
        double mu, var;
        reward_eval(g.n_rewards, g.rw, E, L.mx, L.sx, ws, g.rew_out != nullptr, mu, var);
        if (t == 0) {
            if (g.rew_out) {
                g.rew_out[0] = mu;   // pilco_reward_eval: mean and variance
                g.rew_out[1] = var;
            } else {
                g.reward[0] += mu;   // rollout (pilco.py:133): single writer, stream ordered
            }
        }
        DBG_STAMP(g.wk, 21, t == 0);
        return;
    }
    glue_body(g, L, true);
}

__global__ void k_stamp(unsigned long long* dbg, int slot) {
    if (threadIdx.x == 0) dbg[slot] = wall_clock64();
}
void launch_stamp(hipStream_t st, unsigned long long* dbg, int slot) {
    hipLaunchKernelGGL(k_stamp, dim3(1), dim3(64), 0, st, dbg, slot);
}

void launch_glue(hipStream_t st, const GlueArgs& g, bool with_reward_block) {
    const size_t lds = sizeof(double) * glue_lds_doubles_for(g);
    static size_t configured[64] = {};   // per DEVICE: the attribute is a property of the function on one device
    int dev = 0;
    (void)hipGetDevice(&dev);
    size_t& conf = configured[dev & 63];
    if (lds > conf) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_glue), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)lds);
        conf = lds;
    }
    hipLaunchKernelGGL(k_glue, dim3(with_reward_block ? 2 : 1), dim3(256), lds, st, g);
}

}  // namespace pilco
