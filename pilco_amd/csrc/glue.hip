// k_glue: the serial link of a horizon step (pack, assemble, propagate, controller, joint Gaussian).
#include "mm_device.h"

namespace pilco {

// ------------------------------------------------------------------ glue
// Workgroup 0 is the serial link of the step: everything it needs is pulled into LDS
// with one batch of loads, then (pack ->) assemble -> propagate -> controller -> joint.
// Workgroup 1 (rollouts with a reward) evaluates the reward of the PRE-propagation
// state concurrently (pilco.py:133); the state is double-buffered so it never races
// with workgroup 0's update.
struct GlueLds {
    double* mx;   // [nm]     current state mean
    double* sx;   // [nm*nm]  current state covariance
    double* mu;   // [nm]
    double* su;   // [nm*nm]
    double* cxu;  // [nm*nm]
    double* t1;   // [nm*nm]
    double* t2;   // [nm*nm]
    double* s1;   // [nm*nm]  s1 = [s_x, s_x c_xu] of the previous joint
    double* seg;  // [SEG]    this rank's packed results
    double* mp;   // [EL*NCH*(1+D)] mean partials
    double* misc; // [128]
};

static size_t glue_lds_doubles(int E, int D, int SEG, int mp) {
    const int nm = E > D ? E : D;
    const size_t tail = (size_t)SEG + (size_t)mp;
    const size_t rew = reward_lds_doubles(E);
    return (size_t)2 * nm + 6 * (size_t)nm * nm + 128 + (tail > rew ? tail : rew);
}
size_t glue_lds_bytes(int E, int D) { return sizeof(double) * glue_lds_doubles(E, D, 0, 0); }

// global -> LDS copy with all loads of a 1024-element chunk in flight before the first wait
__device__ __forceinline__ void bulk_load(double* dst, const double* __restrict__ src, int n) {
    for (int base = 0; base < n; base += 1024) {
        double v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int e = base + k * 256 + (int)threadIdx.x;
            v[k] = (e < n) ? src[e] : 0.0;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int e = base + k * 256 + (int)threadIdx.x;
            if (e < n) dst[e] = v[k];
        }
    }
}

// squash_sin on (mu[U], su[U][U]) in place; cdiag[u] = e_u exp(-s_uu/2) cos(m_u)   (controllers.py:13-36)
__device__ __forceinline__ void squash_inplace(const GlueLds& L, int U, const double* maxact, double* cdiag) {
    const int t = threadIdx.x;
    for (int e = t; e < U * U; e += blockDim.x) {
        const int u = e / U, v = e - u * U;
        const double du = L.su[u * U + u], dv = L.su[v * U + v];
        const double lq = -(du + dv) / 2.0;
        const double q = exp(lq);
        const double suv = L.su[e];
        const double val = (exp(lq + suv) - q) * cos(L.mu[u] - L.mu[v]) - (exp(lq - suv) - q) * cos(L.mu[u] + L.mu[v]);
        const double eu = maxact ? maxact[u] : 1.0, ev = maxact ? maxact[v] : 1.0;
        L.t2[e] = eu * ev * val / 2.0;
    }
    if (t < U) {
        const double eu = maxact ? maxact[t] : 1.0;
        const double ex = exp(-L.su[t * U + t] / 2.0);
        cdiag[t] = eu * ex * cos(L.mu[t]);
        L.misc[64 + t] = eu * ex * sin(L.mu[t]);
    }
    __syncthreads();
    for (int e = t; e < U * U; e += blockDim.x) L.su[e] = L.t2[e];
    if (t < U) L.mu[t] = L.misc[64 + t];
    __syncthreads();
}

// joint Gaussian of (x,u) from mx,sx,mu,su,cxu in LDS -> in_m, in_s, s1 (pilco.py:141-144)
__device__ __forceinline__ void write_joint(const GlueArgs& g, const GlueLds& L) {
    const int E = g.E, U = g.U, D = g.D, t = threadIdx.x;
    for (int e = t; e < E * U; e += blockDim.x) {  // sc = s_x c_xu  (E,U)
        const int r = e / U, u = e - r * U;
        double acc = 0.0;
        _Pragma("unroll 8") for (int k = 0; k < E; ++k) acc = fma(L.sx[r * E + k], L.cxu[k * U + u], acc);
        L.t1[e] = acc;
    }
    __syncthreads();
    if (t < D) g.wk.in_m[t] = (t < E) ? L.mx[t] : L.mu[t - E];
    for (int e = t; e < D * D; e += blockDim.x) {
        const int r = e / D, c = e - r * D;
        double v;
        if (r < E && c < E) v = L.sx[r * E + c];
        else if (r < E) v = L.t1[r * U + (c - E)];
        else if (c < E) v = L.t1[c * U + (r - E)];
        else v = L.su[(r - E) * U + (c - E)];
        g.wk.in_s[e] = v;
        if (r < E) g.s1[r * D + c] = v;
        if (g.tape) {
            double* rec = g.tape + (long)g.step * (D + D * D + E * D + E + E * E + D * E);
            rec[D + e] = v;
            if (r < E) rec[D + D * D + r * D + c] = v;
        }
    }
    if (g.tape && t < D) g.tape[(long)g.step * (D + D * D + E * D + E + E * E + D * E) + t] = (t < E) ? L.mx[t] : L.mu[t - E];
}

// Reduce the tile / stream-K partials of the local pairs and the row-chunk partials of the
// owned outputs into this rank's segment (LDS copy + global gather buffer).  Four lanes per
// pair sum fixed quarters of the partial list and are combined in a fixed tree.
// Round `base` of mm_pack (4 threads per pair), split in two so that the loads of the first round are ISSUED at the very
// start of the glue kernel, together with its other loads, and consumed after them (vmcnt is in order: one round trip).
// Only kernel arguments go into the addresses (closed-form wave ranges).
struct PackPre {
    double v[16];
    double isdet;
};
// stream-K partials of pair k live in sk_part[k][0 .. sk_maxw) in wave order (unused slots stay zero); lane gq of the
// pair's four lanes takes the quarter [gq * sk_maxw / 4, (gq + 1) * sk_maxw / 4): 16 contiguous doubles = one cache line
// at the usual sizes, addresses known from the thread index alone.
__device__ __forceinline__ void mm_pack_issue(const MMWork& wk, int base, PackPre& pp) {
    const int t = threadIdx.x;
    const int k = base + (t >> 2), gq = t & 3;
    pp.isdet = 0.0;
#pragma unroll
    for (int u = 0; u < 16; ++u) pp.v[u] = 0.0;
    if (k >= wk.PL) return;
    pp.isdet = wk.pair_isdet[k];
    if (wk.sk_waves > 0) {
        const int qw = wk.sk_maxw >> 2;
        const double* src = wk.sk_part + (long)k * wk.sk_maxw + gq * qw;
#pragma unroll
        for (int u = 0; u < 16; ++u)
            if (u < qw) pp.v[u] = src[u];
    }
}
__device__ __forceinline__ void mm_pack_sum(const MMWork& wk, int base, const PackPre& pp, double& s0, double& s1) {
    const int t = threadIdx.x;
    const int k = base + (t >> 2), gq = t & 3;
    s0 = 0.0;
    s1 = 0.0;
    if (k >= wk.PL) return;
    if (wk.sk_waves > 0) {
#pragma unroll
        for (int u = 0; u < 16; ++u) s0 += pp.v[u];
        const int qw = wk.sk_maxw >> 2;
        const double* src = wk.sk_part + (long)k * wk.sk_maxw + gq * qw;
        for (int u = 16; u < qw; ++u) s0 += src[u];   // few pairs spread over many waves
    } else {
        const double* part = wk.pair_part + (long)k * wk.NT * 2;
        const int q0 = (int)((long)wk.NT * gq / 4), q1 = (int)((long)wk.NT * (gq + 1) / 4);
        for (int q = q0; q < q1; ++q) {
            s0 += part[2 * q];
            s1 += part[2 * q + 1];
        }
    }
}

__device__ __forceinline__ void mm_pack(const MMWork& wk, int D, int E, const GlueLds& L, PackPre& pp) {
    const int t = threadIdx.x;
    double* seg = wk.gath + (long)wk.rank * wk.SEG;
    for (int base = 0; base < wk.PL; base += 64) {
        const int k = base + (t >> 2), gq = t & 3;
        if (base > 0) mm_pack_issue(wk, base, pp);
        double s0, s1;
        mm_pack_sum(wk, base, pp, s0, s1);
        s0 += __shfl_xor(s0, 1);
        s1 += __shfl_xor(s1, 1);
        s0 += __shfl_xor(s0, 2);
        s1 += __shfl_xor(s1, 2);
        if (k < wk.PL && gq == 0) {
            int a, b;
            local_pair_ab(wk, E, k, a, b);
            const double v = ((a == b) ? (s0 - s1) : s0) * pp.isdet;   // mgpr.py:144-145
            seg[k] = v;
            L.seg[k] = v;
        }
    }
    const int W1 = 1 + D;
    for (int e = t; e < wk.EL * W1; e += blockDim.x) {   // M_a and V_a: sums of the chunk contributions
        const int o = e / W1, idx = e - o * W1;
        double sum = 0.0;
        _Pragma("unroll 8") for (int ch = 0; ch < wk.NCHM; ++ch) sum += L.mp[(o * wk.NCHM + ch) * W1 + idx];
        seg[wk.OUTOFF + e] = sum;
        L.seg[wk.OUTOFF + e] = sum;
    }
    __syncthreads();
}

// packed results -> out_M [E], out_S [E][E], out_V [D][E]; also left in LDS (oM, oS, oV)
__device__ __forceinline__ void mm_assemble(const MMWork& wk, const double* src, const double* var, int D, int E, double* oM,
                            double* oS, double* oV) {
    const int t = threadIdx.x;
    for (int a = t; a < E; a += blockDim.x) {
        const double v = src[(a % wk.nranks) * wk.SEG + wk.OUTOFF + (a / wk.nranks) * (1 + D)];
        oM[a] = v;
        wk.out_M[a] = v;
    }
    for (int e = t; e < D * E; e += blockDim.x) {
        const int d = e / E, a = e - d * E;
        const double v = src[(a % wk.nranks) * wk.SEG + wk.OUTOFF + (a / wk.nranks) * (1 + D) + 1 + d];
        oV[e] = v;
        wk.out_V[e] = v;
    }
    __syncthreads();
    for (int e = t; e < E * E; e += blockDim.x) {
        const int a = e / E, b = e - a * E;
        const int hi = a > b ? a : b, lo = a > b ? b : a;
        const int kk = pair_order_index(E, hi, lo);
        double v = src[(kk % wk.nranks) * wk.SEG + kk / wk.nranks];
        if (a == b) v += var[a];                                   // mgpr.py:146
        v = fma(-oM[a], oM[b], v);                                 // mgpr.py:147
        oS[e] = v;
        wk.out_S[e] = v;
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void k_glue(GlueArgs g) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int E = g.E, D = g.D, U = g.U, t = threadIdx.x;
    const int nm = E > D ? E : D;
    int mp_n = (g.flags & GF_PACK) ? g.wk.EL * g.wk.NCHM * (1 + D) : 0;
    int seg_n = (g.flags & (GF_PACK | GF_ASSEMBLE)) ? g.wk.SEG * ((g.flags & GF_PACK) ? 1 : g.wk.nranks) : 0;
    if (g.flags & GF_RBF_POST) {  // this launch reduces the POLICY GP (inputs = state, outputs = controls)
        mp_n = g.pwk.EL * g.pwk.NCHM * (1 + E);
        seg_n = g.pwk.SEG;
    }
    GlueLds L;
    L.mx = sm;
    L.sx = L.mx + nm;
    L.mu = L.sx + nm * nm;
    L.su = L.mu + nm;
    L.cxu = L.su + nm * nm;
    L.t1 = L.cxu + nm * nm;
    L.t2 = L.t1 + nm * nm;
    L.s1 = L.t2 + nm * nm;
    L.misc = L.s1 + nm * nm;
    L.seg = L.misc + 128;
    L.mp = L.seg + seg_n;

    const bool dbg0 = (t == 0);
    const int dbo = (g.step == 0) ? 16 : 0;  // the initial glue of a rollout stamps slots 24..29
    if (blockIdx.x == 1) {
        // Workgroup 1: reward of the current (pre-propagation) state (rewards.py:19-81), evaluated
        // concurrently with workgroup 0; the state is double-buffered so there is no race.
        DBG_STAMP(g.wk, 20, dbg0);
        double* ws = L.seg;  // scratch: this workgroup uses none of the pack / assemble storage
        if (t < E) L.mx[t] = g.m_x[t];
        bulk_load(L.sx, g.s_x, E * E);
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
        DBG_STAMP(g.wk, 21, dbg0);
        return;
    }

    DBG_STAMP(g.wk, 8 + dbo, dbg0);
    PackPre pp;   // first round of the pack: its loads are in flight together with the batch below
    if ((g.flags & GF_PACK) && !MM_ABL(g.wk, 16)) mm_pack_issue(g.wk, 0, pp);
    // one batch of loads for everything the serial part reads
    if (g.flags & (GF_PROPAGATE | GF_TRAJ | GF_POLICY | GF_RBF_PRE)) {
        if (t < E) L.mx[t] = g.m_x[t];
        bulk_load(L.sx, g.s_x, E * E);
    }
    if (g.flags & GF_PROPAGATE) bulk_load(L.s1, g.s1, E * D);
    if (g.flags & GF_PACK) bulk_load(L.mp, g.wk.mean_part, mp_n);
    if (g.flags & GF_RBF_POST) bulk_load(L.mp, g.pwk.mean_part, mp_n);
    if ((g.flags & GF_ASSEMBLE) && !(g.flags & GF_PACK)) bulk_load(L.seg, g.wk.gath, seg_n);
    __syncthreads();

    DBG_STAMP(g.wk, 9 + dbo, dbg0);
    if ((g.flags & GF_PACK) && !MM_ABL(g.wk, 16)) mm_pack(g.wk, D, E, L, pp);
    DBG_STAMP(g.wk, 10 + dbo, dbg0);
    if (g.flags & GF_ASSEMBLE) {
        // single rank: the LDS copy of the segment is the whole gather buffer
        mm_assemble(g.wk, L.seg, g.var, D, E, L.mu, L.su, L.cxu);  // oM -> mu, oS -> su, oV -> cxu
        if (g.tape && g.step >= 1) {
            double* rec = g.tape + (long)(g.step - 1) * (D + D * D + E * D + E + E * E + D * E) + D + D * D + E * D;
            if (t < E) rec[t] = L.mu[t];
            for (int e = t; e < E * E; e += blockDim.x) rec[E + e] = L.su[e];
            for (int e = t; e < D * E; e += blockDim.x) rec[E + E * E + e] = L.cxu[e];
        }
    }
    DBG_STAMP(g.wk, 11 + dbo, dbg0);
    if (g.flags & GF_PROPAGATE) {
        // t1 = s1 V (E,E); state += increment                      (pilco.py:147-149)
        for (int e = t; e < E * E; e += blockDim.x) {
            const int r = e / E, c = e - r * E;
            double acc = 0.0;
            _Pragma("unroll 8") for (int k = 0; k < D; ++k) acc = fma(L.s1[r * D + k], L.cxu[k * E + c], acc);
            L.t1[e] = acc;
        }
        __syncthreads();
        for (int e = t; e < E * E; e += blockDim.x) {
            const int r = e / E, c = e - r * E;
            const double v = ((L.su[e] + L.sx[e]) + L.t1[e]) + L.t1[c * E + r];
            L.t2[e] = v;
            g.s_out[e] = v;
        }
        if (t < E) {
            const double v = L.mu[t] + L.mx[t];
            L.misc[96 + t] = v;
            g.m_out[t] = v;
        }
        __syncthreads();
        for (int e = t; e < E * E; e += blockDim.x) L.sx[e] = L.t2[e];
        if (t < E) L.mx[t] = L.misc[96 + t];
        __syncthreads();
    }
    DBG_STAMP(g.wk, 12 + dbo, dbg0);
    if ((g.flags & GF_TRAJ) && g.traj) {
        double* dst = g.traj + (long)g.step * (E + E * E);
        if (t < E) dst[t] = L.mx[t];
        for (int e = t; e < E * E; e += blockDim.x) dst[E + e] = L.sx[e];
    }
    if (g.flags & GF_RBF_PRE) {  // RbfController: the state is the input of the policy GP (controllers.py:115-116)
        if (t < E) g.pwk.in_m[t] = L.mx[t];
        for (int e = t; e < E * E; e += blockDim.x) g.pwk.in_s[e] = L.sx[e];
    }
    if (g.flags & GF_POLICY) {
        if (g.pol_kind == PILCO_POLICY_RBF) {
            // mean-function-only GP: iK = 0, then S -= diag(var - 1e-6)      (controllers.py:116-117)
            PackPre pq;
            mm_pack_issue(g.pwk, 0, pq);
            mm_pack(g.pwk, E, U, L, pq);
            mm_assemble(g.pwk, L.seg, g.pvar, E, U, L.mu, L.su, L.cxu);   // M (U), S (U,U), V (E,U)
            if (t < U) L.su[t * U + t] -= g.pvar[t] - 1e-6;
            __syncthreads();
            if (g.squash) {
                double* cdiag = L.misc + 1;
                squash_inplace(L, U, g.maxact, cdiag);
                for (int e = t; e < E * U; e += blockDim.x) L.cxu[e] *= cdiag[e % U];
                __syncthreads();
            }
        }
        if (g.pol_kind == PILCO_POLICY_LINEAR) {
            // M = m W^T + b, S = W s W^T, V = W^T                  (controllers.py:52-54)
            bulk_load(L.t2, g.W, U * E);
            __syncthreads();
            if (t < U) {
                double acc = g.b[t];
                _Pragma("unroll 8") for (int k = 0; k < E; ++k) acc = fma(L.t2[t * E + k], L.mx[k], acc);
                L.mu[t] = acc;
            }
            for (int e = t; e < U * E; e += blockDim.x) {
                const int u = e / E, c = e - u * E;
                double acc = 0.0;
                _Pragma("unroll 8") for (int k = 0; k < E; ++k) acc = fma(L.t2[u * E + k], L.sx[k * E + c], acc);
                L.t1[e] = acc;  // W s
                L.cxu[c * U + u] = L.t2[e];
            }
            __syncthreads();
            for (int e = t; e < U * U; e += blockDim.x) {
                const int u = e / U, v = e - u * U;
                double acc = 0.0;
                _Pragma("unroll 8") for (int k = 0; k < E; ++k) acc = fma(L.t1[u * E + k], L.t2[v * E + k], acc);
                L.su[e] = acc;
            }
            __syncthreads();
            if (g.squash) {
                double* cdiag = L.misc + 1;  // [U]
                squash_inplace(L, U, g.maxact, cdiag);
                for (int e = t; e < E * U; e += blockDim.x) L.cxu[e] *= cdiag[e % U];   // V @ C, C diagonal
                __syncthreads();
            }
        }
        if (g.act_out) {
            if (t < U) g.act_out[t] = L.mu[t];
            for (int e = t; e < U * U; e += blockDim.x) g.act_out[U + e] = L.su[e];
            for (int e = t; e < E * U; e += blockDim.x) g.act_out[U + U * U + e] = L.cxu[e];
        } else {
            write_joint(g, L);
        }
    }
    DBG_STAMP(g.wk, 13 + dbo, dbg0);
}

__global__ void k_stamp(unsigned long long* dbg, int slot) {
    if (threadIdx.x == 0) dbg[slot] = wall_clock64();
}
void launch_stamp(hipStream_t st, unsigned long long* dbg, int slot) {
    hipLaunchKernelGGL(k_stamp, dim3(1), dim3(64), 0, st, dbg, slot);
}

void launch_glue(hipStream_t st, const GlueArgs& g, bool with_reward_block) {
    int mp_n = (g.flags & GF_PACK) ? g.wk.EL * g.wk.NCHM * (1 + g.D) : 0;
    int seg_n = (g.flags & (GF_PACK | GF_ASSEMBLE)) ? g.wk.SEG * ((g.flags & GF_PACK) ? 1 : g.wk.nranks) : 0;
    if (g.flags & GF_RBF_POST) {
        mp_n = g.pwk.EL * g.pwk.NCHM * (1 + g.E);
        seg_n = g.pwk.SEG;
    }
    const size_t lds = sizeof(double) * glue_lds_doubles(g.E, g.D, seg_n, mp_n);
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
