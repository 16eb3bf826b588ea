// The analytic moment-matching step (T5-T19 of SURVEY.md section 2.2) as three
// gfx950 kernels per horizon step:
//
//   k_mm_prep  : per output pair (a,b): R_ab, det R_ab, Q_ab = R^{-1} s / 2 by a
//                register-resident Gauss-Jordan in one wave (column per lane,
//                v_readlane broadcasts, no LDS, no barriers); then the O(N D^2)
//                per-row vectors of Appendix B (u_i, p_i = 2 Q z_i | w_j, v_j)
//                written k-major so the pair kernel reads MFMA fragments with
//                128-byte segments; diagonal pairs also do the mean / input-output
//                covariance sums (mgpr.py:102-118).  One extra workgroup of the
//                same launch evaluates the reward of the current state
//                (rewards.py:32-39), off the step's critical path.
//   k_mm_pair  : the O(N^2) part (mgpr.py:120-144): exponent tile = A^T B on
//                v_mfma_f64_16x16x4_f64 with K = D+2 (u and v folded into the
//                contraction), table-driven fp64 exp, beta-weighted reduction and,
//                for a == b, the streamed iK tile.  No atomics: one partial per
//                tile, summed in a fixed order => bitwise reproducible.
//   k_glue     : one workgroup: tile-partial reduction, S assembly
//                (mgpr.py:145-147), propagate (pilco.py:147-149), controller +
//                joint Gaussian for the next step (controllers.py:13-58,
//                pilco.py:139-144).
#include "pair_device.h"

namespace pilco {

template <int KC, bool VSEP>
__global__ __launch_bounds__(256, PAIR_MINW) void k_mm_pair_tiled(MMModel md, MMWork wk, int NJB) {
    __shared__ double red[4];
    __shared__ double tab[FEXP_TN];
    kernarg_warm<(int)(sizeof(MMModel) + sizeof(MMWork)) + 8 + 64>();
    for (int e = threadIdx.x; e < FEXP_TN; e += blockDim.x) tab[e] = wk.exp_tab[e];
    __syncthreads();
    const int npad = md.npad;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int jb = blockIdx.x % NJB, ti = blockIdx.x / NJB, pl = blockIdx.y;
    int a, b;
    local_pair_ab(wk, md.E, pl, a, b);
    const bool diag = (a == b) && (md.iK != nullptr);
    const PairOps po = pair_ops(wk, md.D, npad, pl, b);
    const double* beta_a = md.beta + mm_beta_row(md, a) * npad;
    const double* beta_b = md.beta + mm_beta_row(md, b) * npad;
    const int i0 = ti * 16 * PAIR_RT;
    const int JB = npad / NJB, JW = JB / 4;
    const int jbeg = jb * JB + w * JW;
    double t1;
    if (diag)
        t1 = pair_wave<KC, true, VSEP>(po, nullptr, nullptr, nullptr, beta_a, beta_b, md.iK + mm_ik_blk(md, a) * npad * npad, tab, npad, i0, jbeg, jbeg + JW, lane);
    else
        t1 = pair_wave<KC, false, VSEP>(po, nullptr, nullptr, nullptr, beta_a, beta_b, nullptr, tab, npad, i0, jbeg, jbeg + JW, lane);
    for (int off = 32; off > 0; off >>= 1) t1 += __shfl_down(t1, off);
    if (lane == 0) red[w] = t1;
    __syncthreads();
    if (threadIdx.x == 0) {
        double* out = wk.pair_part + ((long)pl * wk.NT + ti * NJB + jb) * 2;
        out[0] = (red[0] + red[1]) + (red[2] + red[3]);
        out[1] = 0.0;  // the trace term is already folded into out[0]
    }
}

// row stride of the pair-major partial array: the largest number of waves touching one local pair, rounded up to 4
int mm_sk_maxw(const MMWork& wk) {
    int m = 4;
    for (int k = 0; k < wk.PL; ++k) {
        int wlo, fs, whi;
        sk_pair_waves(k, wk.sk_waves, wk.sk_nd, wk.sk_tdiag, wk.sk_toff, wk.sk_total, wk.sk_ud, wk.sk_uo, wlo, fs, whi);
        m = std::max(m, whi - wlo + 1);
    }
    return (m + 3) / 4 * 4;
}
int mm_sk_boundary(int w, int waves, int nd_steps, int total, int ud, int uo) {
    return sk_boundary_of(w, waves, nd_steps, total, ud, uo);
}
void mm_sk_pair_waves(int k, int waves, int nd, int tdiag, int toff, int total, int ud, int uo, int* wlo, int* fslot, int* whi) {
    sk_pair_waves(k, waves, nd, tdiag, toff, total, ud, uo, *wlo, *fslot, *whi);
}

template <int KC, bool VSEP>
__global__ __launch_bounds__(256, (KC <= 4 && PAIR_MINW < 3) ? 3 : PAIR_MINW) void k_mm_pair_sk(MMModel md, MMWork wk) {
    __shared__ double tab[FEXP_TN];
    kernarg_warm<(int)(sizeof(MMModel) + sizeof(MMWork)) + 64>();
    TabArrival ta;
    ta.request(wk.exp_tab, tab);   // requested here, stored behind the first tile's operand requests
    const int lane = threadIdx.x & 63;
    // XCD-aware placement: workgroups are dealt round-robin over the 8 XCDs (own L2 each), so workgroup b takes
    // position (b % 8) * (blocks / 8) + b / 8 of the cost line: the waves of one XCD cover one contiguous eighth of it
    // and its L2 holds the operands of ~1/8 of the pairs instead of all of them.
    int bpos = blockIdx.x;
    if ((gridDim.x & 7) == 0) bpos = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const int w = __builtin_amdgcn_readfirstlane(bpos * 4 + (threadIdx.x >> 6));
#ifdef PAIR_WAVE_TRACE   // developer build (tools/pair_trace.py): start, end and hardware id of EVERY wave
    if (wk.dbg && lane == 0) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        wk.dbg[4096 + w] = wall_clock64();
        wk.dbg[4096 + 2 * 3072 + w] = ((unsigned long long)xcc << 32) | hw;
    }
#endif
    DBG_STAMP(wk, 16, w == 0 && lane == 0);
    if (wk.dbg && w == 0 && lane == 0) wk.dbg[32] = clock64();     // shader-clock counter beside the 100 MHz wall clock: tools/ derive the engine clock under this load
    const int nd_steps = wk.sk_nd * wk.sk_tdiag;
    double out0, out1;
    int p0, p1;
    sk_wave_range<KC, VSEP>(md, wk, tab, w, lane, out0, out1, p0, p1, &ta, PAIR_FAIR != 0);
    ta.land();   // (a wave without a single step still owes the workgroup its barrier)
    for (int off = 32; off > 0; off >>= 1) {
        out0 += __shfl_down(out0, off);
        out1 += __shfl_down(out1, off);
    }
    if (lane == 0) {
        // slot-major layout sk_part[slot][pair], slot = wave - (first wave of the pair): the reader (the serial link)
        // gets coalesced loads with addresses from its thread index alone.  A wave that enters a pair from a previous one
        // IS that pair's first wave (slot 0); only the first touched pair needs the closed form.
        if (p0 >= 0) {
            const long S0 = (p0 < wk.sk_nd) ? (long)p0 * wk.sk_tdiag : (long)wk.sk_nd * wk.sk_tdiag + (long)(p0 - wk.sk_nd) * wk.sk_toff;
            const int wlo = sk_wave_of(S0, wk.sk_waves, nd_steps, wk.sk_total, wk.sk_ud, wk.sk_uo);
            wk.sk_part[(long)(w - wlo) * wk.sk_pls + p0] = out0;
        }
        if (p1 >= 0) wk.sk_part[p1] = out1;
    }
    DBG_STAMP(wk, 17, w == 0 && lane == 0);
    if (wk.dbg && w == 0 && lane == 0) wk.dbg[33] = clock64();
    DBG_STAMP(wk, 18, w == wk.sk_waves - 1 && lane == 0);
    if (wk.dbg && lane == 0 && (w & 7) == 0) wk.dbg[1024 + (w >> 3)] = wall_clock64();  // end stamp of every 8th wave
#ifdef PAIR_WAVE_TRACE
    if (wk.dbg && lane == 0) wk.dbg[4096 + 3072 + w] = wall_clock64();
#endif
}


// ------------------------------------------------------------------ pair kernel, plain VALU  (-DPILCO_DEV builds only)
// Reference implementation of the same tile sums without matrix cores: one row
// per thread (256-row tile), 64 columns staged in LDS and read by broadcast.  A cross-check of the MFMA kernels, not a
// product path: the shipped library does not contain it (pilco_set_pair_kernel(ctx, 1) fails there).
#ifdef PILCO_DEV
template <int KPT>
__global__ __launch_bounds__(256) void k_mm_pair_valu(MMModel md, MMWork wk) {
    __shared__ double Bs[KPT][64];
    __shared__ double bbs[64];
    __shared__ double vs[64];
    __shared__ double red[8];
    const int npad = md.npad;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int ncb = npad / 64;
    const int tj = blockIdx.x % ncb, ti = blockIdx.x / ncb, pl = blockIdx.y;
    int a, b;
    local_pair_ab(wk, md.E, pl, a, b);
    const bool diag = (a == b) && (md.iK != nullptr);
    const int KP = wk.KP;
    const PairOps po = pair_ops(wk, md.D, npad, pl, b);
    const int i = ti * 256 + t;
    const bool rowok = i < npad;
    double av[KPT];
#pragma unroll
    for (int k = 0; k < KPT; ++k) av[k] = (k < KP && rowok) ? po.At[(long)k * npad + i] : 0.0;
    const int j0 = tj * 64;
    for (int e = t; e < KPT * 64; e += 256) {
        const int k = e >> 6, j = e & 63;
        Bs[k][j] = (k < KP) ? (wk.vsep ? colop_row<true>(po, npad, k) : colop_row<false>(po, npad, k))[j0 + j] : 0.0;   // (row D + 1: the pair's v_j)
    }
    if (t < 64) {
        bbs[t] = md.beta[mm_beta_row(md, b) * npad + j0 + t];
        vs[t] = wk.vsep ? wk.vcol[(long)pl * npad + j0 + t] : 0.0;
    }
    __syncthreads();
    double s1 = 0.0, s2 = 0.0;
    if (rowok) {
        const double* iKrow = diag ? md.iK + (mm_ik_blk(md, a) * npad + i) * npad + j0 : nullptr;
        for (int j = 0; j < 64; ++j) {
            double e = vs[j];
#pragma unroll
            for (int k = 0; k < KPT; ++k) e = fma(av[k], Bs[k][j], e);
            const double L = exp(e);
            s1 = fma(bbs[j], L, s1);
            if (diag) s2 = fma(iKrow[j], L, s2);
        }
        s1 *= md.beta[mm_beta_row(md, a) * npad + i];
    }
    for (int off = 32; off > 0; off >>= 1) {
        s1 += __shfl_down(s1, off);
        s2 += __shfl_down(s2, off);
    }
    if (lane == 0) {
        red[2 * w] = s1;
        red[2 * w + 1] = s2;
    }
    __syncthreads();
    if (t == 0) {
        double* out = wk.pair_part + ((long)pl * wk.NT + ti * ncb + tj) * 2;
        out[0] = ((red[0] + red[2]) + red[4]) + red[6];
        out[1] = ((red[1] + red[3]) + red[5]) + red[7];
    }
}
#endif   // PILCO_DEV

static int pair_njb(int npad, int PL) {
    const int nb = npad / 64;
    int njb = 1;
    while (njb * 2 <= nb && nb % (njb * 2) == 0 && (long)PL * (npad / (16 * PAIR_RT)) * njb < 1536) njb *= 2;
    const char* env = getenv("PILCO_PAIR_NJB");
    if (env) {
        const int v = atoi(env);
        if (v >= 1 && v <= nb && nb % v == 0) njb = v;
    }
    return njb;
}

int mm_pair_nt(int npad, int variant, int PL) {
    if (variant == 1) return ((npad + 255) / 256) * (npad / 64);
    if (variant == 2) {  // depends on npad only, so the summation order is the same for every rank count
        const int nb = npad / 64;
        int njb = 1;
        while (njb * 2 <= nb && nb % (njb * 2) == 0 && njb < 4) njb *= 2;
        return (npad / (16 * PAIR_RT)) * njb;
    }
    return (npad / (16 * PAIR_RT)) * pair_njb(npad, PL);
}

void mm_pair_sk_steps(int npad, int* tdiag, int* toff) {
    const int NS = npad / 16, NTI = npad / (16 * PAIR_RT);
    *toff = NTI * NS;
    *tdiag = NTI * NS - PAIR_RT * NTI * (NTI - 1) / 2;
}

template <int KC, bool VSEP>
static int sk_capacity_of() {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_mm_pair_sk<KC, VSEP>, 256, 0) != hipSuccess || nb <= 0) nb = 2;
    int dev = 0, cus = 256;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    return nb * cus * 4;
}

int mm_pair_sk_capacity(int KP, bool vsep) {
    const char* env = getenv("PILCO_SK_WAVES");
    if (env && atoi(env) >= 4) return atoi(env) / 4 * 4;
    switch (KP / 4) {
        case 1: return vsep ? sk_capacity_of<1, true>() : sk_capacity_of<1, false>();
        case 2: return vsep ? sk_capacity_of<2, true>() : sk_capacity_of<2, false>();
        case 3: return vsep ? sk_capacity_of<3, true>() : sk_capacity_of<3, false>();
        case 4: return vsep ? sk_capacity_of<4, true>() : sk_capacity_of<4, false>();
        case 5: return vsep ? sk_capacity_of<5, true>() : sk_capacity_of<5, false>();
        case 6: return vsep ? sk_capacity_of<6, true>() : sk_capacity_of<6, false>();
        case 7: return vsep ? sk_capacity_of<7, true>() : sk_capacity_of<7, false>();
        case 8: return vsep ? sk_capacity_of<8, true>() : sk_capacity_of<8, false>();
        default: return vsep ? sk_capacity_of<9, true>() : sk_capacity_of<9, false>();
    }
}

void launch_mm_pair(hipStream_t st, const MMModel& md, const MMWork& wk, int variant) {
    const int KP = wk.KP;
#ifdef PILCO_DEV
    if (variant == 1) {
        dim3 grid(((md.npad + 255) / 256) * (md.npad / 64), wk.PL);
#define PV(K_) hipLaunchKernelGGL((k_mm_pair_valu<K_>), grid, dim3(256), 0, st, md, wk)
        if (KP <= 4) PV(4);
        else if (KP <= 8) PV(8);
        else if (KP <= 12) PV(12);
        else if (KP <= 16) PV(16);
        else if (KP <= 24) PV(24);
        else PV(36);
#undef PV
        return;
    }
#endif
    if (variant == 2) {
        const int NJB = wk.NT / (md.npad / (16 * PAIR_RT));
        dim3 grid((md.npad / (16 * PAIR_RT)) * NJB, wk.PL);
#define PM(K_)                                                                                        \
    do {                                                                                              \
        if (wk.vsep) hipLaunchKernelGGL((k_mm_pair_tiled<K_, true>), grid, dim3(256), 0, st, md, wk, NJB);  \
        else hipLaunchKernelGGL((k_mm_pair_tiled<K_, false>), grid, dim3(256), 0, st, md, wk, NJB);         \
    } while (0)
        switch (KP / 4) {
            case 1: PM(1); break;
            case 2: PM(2); break;
            case 3: PM(3); break;
            case 4: PM(4); break;
            case 5: PM(5); break;
            case 6: PM(6); break;
            case 7: PM(7); break;
            case 8: PM(8); break;
            default: PM(9); break;
        }
#undef PM
        return;
    }
    dim3 grid(wk.sk_waves / 4);
#define PS(K_)                                                                                 \
    do {                                                                                       \
        if (wk.vsep) hipLaunchKernelGGL((k_mm_pair_sk<K_, true>), grid, dim3(256), 0, st, md, wk);   \
        else hipLaunchKernelGGL((k_mm_pair_sk<K_, false>), grid, dim3(256), 0, st, md, wk);          \
    } while (0)
    switch (KP / 4) {
        case 1: PS(1); break;
        case 2: PS(2); break;
        case 3: PS(3); break;
        case 4: PS(4); break;
        case 5: PS(5); break;
        case 6: PS(6); break;
        case 7: PS(7); break;
        case 8: PS(8); break;
        default: PS(9); break;
    }
#undef PS
}

// ------------------------------------------------------------------ self test
// D = A B with A[i][k] = i + 1 + 100 k (16x4), B[k][j] = (k == 0) ? j + 1 : 0 so that
// D[i][j] = (i + 1)(j + 1): exposes both the operand and the result lane maps.
__global__ void k_selftest_mfma(double* out) {
    const int lane = threadIdx.x;
    const int i = lane & 15, k = lane >> 4;
    const double a = (double)(i + 1 + 100 * k);
    const double b = (k == 0) ? (double)((lane & 15) + 1) : 0.0;
    d4 acc = {0.0, 0.0, 0.0, 0.0};
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    MFMA_KEEP_ALIVE(a);
    MFMA_KEEP_ALIVE(b);
    for (int r = 0; r < 4; ++r) out[lane * 4 + r] = acc[r];
}

// max relative deviation of the table-driven exp from the library exp over [-720, 8]
__global__ void k_selftest_fexp(const double* tab_g, double* out) {
    __shared__ double tab[FEXP_TN];
    for (int e = threadIdx.x; e < FEXP_TN; e += blockDim.x) tab[e] = tab_g[e];
    __syncthreads();
    double worst = 0.0;
    for (int i = threadIdx.x; i < 200000; i += blockDim.x) {
        const double x = -720.0 + 728.0 * ((double)i + 0.37) / 200000.0;
        const double ref = exp(fmax(x, -700.0));
        const double got = fexp(x, tab);
        // allowed: 1 ulp of the result + the |x| eps conditioning of the single-constant reduction
        const double rel = fabs(got - ref) / ref / (2.3e-16 + 1.2e-16 * fabs(x));
        worst = fmax(worst, rel);
    }
    for (int off = 32; off > 0; off >>= 1) worst = fmax(worst, __shfl_down(worst, off));
    if ((threadIdx.x & 63) == 0) out[threadIdx.x >> 6] = worst;
}

int mm_exp_table_size() { return FEXP_TN; }

int launch_selftest_mfma(hipStream_t st, double* dbuf, double* hbuf, const double* exp_tab) {
    hipLaunchKernelGGL(k_selftest_fexp, dim3(1), dim3(256), 0, st, exp_tab, dbuf);
    if (hipMemcpyAsync(hbuf, dbuf, 4 * sizeof(double), hipMemcpyDeviceToHost, st) != hipSuccess) return -1;
    if (hipStreamSynchronize(st) != hipSuccess) return -1;
    for (int w = 0; w < 4; ++w)
        if (!(hbuf[w] < 1.0)) return 100000;
    hipLaunchKernelGGL(k_selftest_mfma, dim3(1), dim3(64), 0, st, dbuf);
    if (hipMemcpyAsync(hbuf, dbuf, 256 * sizeof(double), hipMemcpyDeviceToHost, st) != hipSuccess) return -1;
    if (hipStreamSynchronize(st) != hipSuccess) return -1;
    for (int lane = 0; lane < 64; ++lane)
        for (int r = 0; r < 4; ++r) {
            const int row = (lane >> 4) + 4 * r, col = lane & 15;
            if (hbuf[lane * 4 + r] != (double)((row + 1) * (col + 1))) return 1 + lane * 4 + r;
        }
    return 0;
}

}  // namespace pilco
