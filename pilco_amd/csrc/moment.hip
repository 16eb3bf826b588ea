// The analytic moment-matching step (T5-T19 of SURVEY.md section 2.2) as three
// gfx950 kernels per horizon step:
//
//   k_mm_prep  : per output pair (a,b): R_ab, det R_ab, Q_ab = R^{-1} s / 2 by a
//                pivoted Gauss-Jordan in LDS; then the O(N D^2) per-row vectors
//                of Appendix B (u_i, p_i = 2 Q z_i | w_j, v_j) written k-major so
//                the pair kernel reads MFMA fragments with 128-byte segments;
//                diagonal pairs also do the mean / input-output covariance sums
//                (mgpr.py:102-118).
//   k_mm_pair  : the O(N^2) part (mgpr.py:120-144): exponent tile = A^T B on
//                v_mfma_f64_16x16x4_f64 with K = D+2 (u and v folded into the
//                contraction), fp64 exp, beta-weighted reduction and, for a == b,
//                the streamed iK tile.  No atomics: one partial per tile, summed
//                in a fixed order => bitwise reproducible.
//   k_glue     : one workgroup: tile-partial reduction, S assembly
//                (mgpr.py:145-147), propagate (pilco.py:147-149), reward
//                (rewards.py:32-39), controller + joint Gaussian for the next step
//                (controllers.py:13-58, pilco.py:139-144).
#include "moment.h"

namespace pilco {

typedef double d4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void decode_pair(int p, int& a, int& b) {
    a = 0;
    while ((a + 1) * (a + 2) / 2 <= p) ++a;
    b = p - a * (a + 1) / 2;
}

// Pivoted Gauss-Jordan on an n x nc augmented matrix held in LDS (row-major,
// ld = nc), ping-ponging between two buffers: one barrier per pivot step.
// Called by the whole workgroup.  Returns the buffer holding [I | A^{-1} B];
// det = det(A) (valid in every thread).
__device__ double* gauss_jordan(double* G0, double* G1, int n, int nc, double& det) {
    double* cur = G0;
    double* nxt = G1;
    det = 1.0;
    for (int k = 0; k < n; ++k) {
        __syncthreads();
        int p = k;
        double best = fabs(cur[k * nc + k]);
        for (int r = k + 1; r < n; ++r) {
            const double v = fabs(cur[r * nc + k]);
            if (v > best) {
                best = v;
                p = r;
            }
        }
        const double piv = cur[p * nc + k];
        det *= (p == k) ? piv : -piv;
        for (int e = threadIdx.x; e < n * nc; e += blockDim.x) {
            const int r = e / nc, c = e - r * nc;
            const double pk = cur[p * nc + c] / piv;
            double val;
            if (r == k) {
                val = pk;
            } else {
                const int rs = (r == p) ? k : r;
                val = fma(-cur[rs * nc + k], pk, cur[rs * nc + c]);
            }
            nxt[e] = val;
        }
        double* tmp = cur;
        cur = nxt;
        nxt = tmp;
    }
    __syncthreads();
    return cur;
}

// ------------------------------------------------------------------ prep
template <int DT>
__global__ __launch_bounds__(256) void k_mm_prep(MMModel md, MMWork wk) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int D = md.D, npad = md.npad;
    double* s_m = sm;
    double* s_ia2 = s_m + DT;
    double* s_ib2 = s_ia2 + DT;
    double* s_ia = s_ib2 + DT;
    double* s_s = s_ia + DT;
    double* s_Q = s_s + DT * DT;
    double* s_T = s_Q + DT * DT;
    double* G0 = s_T + DT * DT;
    double* G1 = G0 + 2 * DT * DT;
    double* red = G1 + 2 * DT * DT;  // 4 * (DT + 1)
    const int t = threadIdx.x;
    const int pl = blockIdx.x, ch = blockIdx.y;
    int a, b;
    decode_pair(wk.pair_list[pl], a, b);
    const bool diag = (a == b);
    if (t < D) {
        s_m[t] = wk.in_m[t];
        const double la = md.ls[a * D + t], lb = md.ls[b * D + t];
        s_ia2[t] = 1.0 / (la * la);
        s_ib2[t] = 1.0 / (lb * lb);
        s_ia[t] = 1.0 / la;
    }
    for (int e = t; e < D * D; e += 256) s_s[e] = wk.in_s[e];
    __syncthreads();
    const int nc = 2 * D;
    // [R | s],  R = s diag(la^-2 + lb^-2) + I        (mgpr.py:121-124,129)
    for (int e = t; e < D * nc; e += 256) {
        const int r = e / nc, c = e - r * nc;
        G0[e] = (c < D) ? fma(s_s[r * D + c], s_ia2[c] + s_ib2[c], (r == c) ? 1.0 : 0.0) : s_s[r * D + c - D];
    }
    double det;
    double* res = gauss_jordan(G0, G1, D, nc, det);
    for (int e = t; e < D * D; e += 256) {
        const int r = e / D, c = e - r * D;
        s_Q[r * DT + c] = 0.5 * res[r * nc + D + c];
    }
    if (ch == 0 && t == 0) wk.pair_isdet[pl] = 1.0 / sqrt(det);
    __syncthreads();
    double cfac = 0.0;
    if (diag) {
        // [B | I],  B = Lambda^-1 s Lambda^-1 + I; T = Lambda^-1 B^-1 Lambda^-1   (mgpr.py:103-111)
        for (int e = t; e < D * nc; e += 256) {
            const int r = e / nc, c = e - r * nc;
            G0[e] = (c < D) ? fma(s_s[r * D + c], s_ia[r] * s_ia[c], (r == c) ? 1.0 : 0.0) : ((c - D == r) ? 1.0 : 0.0);
        }
        double detB;
        res = gauss_jordan(G0, G1, D, nc, detB);
        for (int e = t; e < D * D; e += 256) {
            const int r = e / D, c = e - r * D;
            const double v = res[r * nc + D + c] * s_ia[r] * s_ia[c];
            s_T[r * DT + c] = v;
            if (ch == 0) wk.T[((long)a * D + r) * D + c] = v;
        }
        cfac = md.var[a] / sqrt(detB);
        if (ch == 0 && t == 0) wk.c[a] = cfac;
        __syncthreads();
    }
    const double logva = log(md.var[a]), logvb = log(md.var[b]);
    const int KP = wk.KP;
    const int rpc = npad / wk.NCH;
    const int i_begin = ch * rpc, i_end = i_begin + rpc;
    double* At = wk.At + (long)pl * KP * npad;
    double* Bt = wk.Bt + (long)pl * KP * npad;
    const double* beta_a = md.beta + (long)a * npad;
    double g = 0.0;
    double h[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d) h[d] = 0.0;
    for (int i = i_begin + t; i < i_end; i += 256) {
        const bool valid = i < md.n;
        double zeta[DT];
#pragma unroll
        for (int d = 0; d < DT; ++d) zeta[d] = (d < D && valid) ? md.Pt[(long)d * npad + i] - s_m[d] : 0.0;
        // row side: z = zeta / la^2
        {
            double kk = logva, u = 0.0;
#pragma unroll
            for (int d = 0; d < DT; ++d)
                if (d < D) kk = fma(-0.5 * zeta[d] * zeta[d], s_ia2[d], kk);
#pragma unroll
            for (int r = 0; r < DT; ++r) {
                if (r < D) {
                    double qz = 0.0;
#pragma unroll
                    for (int c = 0; c < DT; ++c)
                        if (c < D) qz = fma(s_Q[r * DT + c], zeta[c] * s_ia2[c], qz);
                    u = fma(zeta[r] * s_ia2[r], qz, u);
                    At[(long)r * npad + i] = valid ? 2.0 * qz : 0.0;
                }
            }
            At[(long)D * npad + i] = valid ? (kk + u) : 0.0;
            At[(long)(D + 1) * npad + i] = valid ? 1.0 : 0.0;
            for (int k = D + 2; k < KP; ++k) At[(long)k * npad + i] = 0.0;
        }
        // column side: w = zeta / lb^2
        {
            double kk = logvb, v = 0.0;
#pragma unroll
            for (int d = 0; d < DT; ++d)
                if (d < D) kk = fma(-0.5 * zeta[d] * zeta[d], s_ib2[d], kk);
#pragma unroll
            for (int r = 0; r < DT; ++r) {
                if (r < D) {
                    double qw = 0.0;
#pragma unroll
                    for (int c = 0; c < DT; ++c)
                        if (c < D) qw = fma(s_Q[r * DT + c], zeta[c] * s_ib2[c], qw);
                    v = fma(zeta[r] * s_ib2[r], qw, v);
                    Bt[(long)r * npad + i] = valid ? zeta[r] * s_ib2[r] : 0.0;
                }
            }
            Bt[(long)D * npad + i] = valid ? 1.0 : 0.0;
            Bt[(long)(D + 1) * npad + i] = valid ? (kk + v) : 0.0;
            for (int k = D + 2; k < KP; ++k) Bt[(long)k * npad + i] = 0.0;
        }
        if (diag) {  // mean part: lb_i = exp(-zeta^T T zeta / 2) beta_i      (mgpr.py:113)
            double q = 0.0;
#pragma unroll
            for (int r = 0; r < DT; ++r) {
                if (r < D) {
                    double tz = 0.0;
#pragma unroll
                    for (int c = 0; c < DT; ++c)
                        if (c < D) tz = fma(s_T[r * DT + c], zeta[c], tz);
                    q = fma(zeta[r], tz, q);
                }
            }
            const double lb = exp(-0.5 * q) * beta_a[i];
            g += lb;
#pragma unroll
            for (int d = 0; d < DT; ++d) h[d] = fma(zeta[d], lb, h[d]);
        }
    }
    if (diag) {
        const int lane = t & 63, w = t >> 6;
        for (int off = 32; off > 0; off >>= 1) g += __shfl_down(g, off);
        if (lane == 0) red[w * (DT + 1)] = g;
#pragma unroll
        for (int d = 0; d < DT; ++d) {
            double v = h[d];
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
            if (lane == 0) red[w * (DT + 1) + 1 + d] = v;
        }
        __syncthreads();
        if (t < 1 + D) {
            const double v = ((red[t] + red[(DT + 1) + t]) + red[2 * (DT + 1) + t]) + red[3 * (DT + 1) + t];
            wk.mean_part[((long)a * wk.NCH + ch) * (1 + D) + t] = v;
        }
    }
}

size_t prep_lds_bytes(int DT) { return sizeof(double) * (4 * DT + 3 * DT * DT + 4 * DT * DT + 4 * (DT + 1)); }

int mm_kp(int D) { return round_up(D + 2, 4); }

int mm_prep_nch(int npad, int PL) {
    // enough row chunks to occupy the chip, each chunk a multiple of 64 rows
    const int nb = npad / 64;
    int nch = 1;
    while (nch * 2 <= nb && nb % (nch * 2) == 0 && PL * nch < 256) nch *= 2;
    return nch;
}

void launch_mm_prep(hipStream_t st, const MMModel& md, const MMWork& wk) {
    dim3 grid(wk.PL, wk.NCH);
    const int D = md.D;
#define PREP(DT_)                                                                                          \
    hipLaunchKernelGGL((k_mm_prep<DT_>), grid, dim3(256), prep_lds_bytes(DT_), st, md, wk)
    if (D <= 4) PREP(4);
    else if (D <= 8) PREP(8);
    else if (D <= 12) PREP(12);
    else if (D <= 16) PREP(16);
    else if (D <= 24) PREP(24);
    else PREP(32);
#undef PREP
}

// ------------------------------------------------------------------ pair kernel, MFMA
// Work item of a workgroup: (local pair, 64-row tile, column block); the four
// waves take consecutive column sub-ranges of JW columns.  Per 16-column step a
// wave issues 4*KC MFMAs (four 16-row tiles) and 16 exps per lane.
template <int KC>
__global__ __launch_bounds__(256) void k_mm_pair_mfma(MMModel md, MMWork wk, int NJB) {
    __shared__ double red[8];
    const int npad = md.npad;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int lr = lane >> 4, lc = lane & 15;
    const int jb = blockIdx.x % NJB, ti = blockIdx.x / NJB, pl = blockIdx.y;
    int a, b;
    decode_pair(wk.pair_list[pl], a, b);
    const bool diag = (a == b) && (md.iK != nullptr);
    const int KP = wk.KP;
    const double* At = wk.At + (long)pl * KP * npad;
    const double* Bt = wk.Bt + (long)pl * KP * npad;
    const double* beta_a = md.beta + (long)a * npad;
    const double* beta_b = md.beta + (long)b * npad;
    const int i0 = ti * 64;
    const int JB = npad / NJB, JW = JB / 4;
    const int jbeg = jb * JB + w * JW;

    double af[4][KC];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int c = 0; c < KC; ++c) af[rt][c] = At[(long)(4 * c + lr) * npad + i0 + 16 * rt + lc];
    double s1[4][4], s2[4][4];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            s1[rt][r] = 0.0;
            s2[rt][r] = 0.0;
        }
    const double* iKa = diag ? md.iK + (long)a * npad * npad : nullptr;

    for (int j0 = jbeg; j0 < jbeg + JW; j0 += 16) {
        double bf[KC];
#pragma unroll
        for (int c = 0; c < KC; ++c) bf[c] = Bt[(long)(4 * c + lr) * npad + j0 + lc];
        const double bb = beta_b[j0 + lc];
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
            d4 e = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int c = 0; c < KC; ++c) e = __builtin_amdgcn_mfma_f64_16x16x4f64(af[rt][c], bf[c], e, 0, 0, 0);
            // C/D layout of the f64 MFMA: col = lane & 15, row = (lane >> 4) + 4 * reg
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double L = exp(e[r]);
                s1[rt][r] = fma(bb, L, s1[rt][r]);
                if (diag) {
                    const int row = i0 + 16 * rt + lr + 4 * r;
                    s2[rt][r] = fma(iKa[(long)row * npad + j0 + lc], L, s2[rt][r]);
                }
            }
        }
    }
    double t1 = 0.0, t2 = 0.0;
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = i0 + 16 * rt + lr + 4 * r;
            t1 = fma(beta_a[row], s1[rt][r], t1);
            t2 += s2[rt][r];
        }
    for (int off = 32; off > 0; off >>= 1) {
        t1 += __shfl_down(t1, off);
        t2 += __shfl_down(t2, off);
    }
    if (lane == 0) {
        red[2 * w] = t1;
        red[2 * w + 1] = t2;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double* out = wk.pair_part + ((long)pl * wk.NT + ti * NJB + jb) * 2;
        out[0] = ((red[0] + red[2]) + red[4]) + red[6];
        out[1] = ((red[1] + red[3]) + red[5]) + red[7];
    }
}

// ------------------------------------------------------------------ pair kernel, plain VALU
// Reference implementation of the same tile sums without matrix cores: one row
// per thread (256-row tile), 64 columns staged in LDS and read by broadcast.
template <int KPT>
__global__ __launch_bounds__(256) void k_mm_pair_valu(MMModel md, MMWork wk) {
    __shared__ double Bs[KPT][64];
    __shared__ double bbs[64];
    __shared__ double red[8];
    const int npad = md.npad;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int ncb = npad / 64;
    const int tj = blockIdx.x % ncb, ti = blockIdx.x / ncb, pl = blockIdx.y;
    int a, b;
    decode_pair(wk.pair_list[pl], a, b);
    const bool diag = (a == b) && (md.iK != nullptr);
    const int KP = wk.KP;
    const double* At = wk.At + (long)pl * KP * npad;
    const double* Bt = wk.Bt + (long)pl * KP * npad;
    const int i = ti * 256 + t;
    const bool rowok = i < npad;
    double av[KPT];
#pragma unroll
    for (int k = 0; k < KPT; ++k) av[k] = (k < KP && rowok) ? At[(long)k * npad + i] : 0.0;
    const int j0 = tj * 64;
    for (int e = t; e < KPT * 64; e += 256) {
        const int k = e >> 6, j = e & 63;
        Bs[k][j] = (k < KP) ? Bt[(long)k * npad + j0 + j] : 0.0;
    }
    if (t < 64) bbs[t] = md.beta[(long)b * npad + j0 + t];
    __syncthreads();
    double s1 = 0.0, s2 = 0.0;
    if (rowok) {
        const double* iKrow = diag ? md.iK + ((long)a * npad + i) * npad + j0 : nullptr;
        for (int j = 0; j < 64; ++j) {
            double e = 0.0;
#pragma unroll
            for (int k = 0; k < KPT; ++k) e = fma(av[k], Bs[k][j], e);
            const double L = exp(e);
            s1 = fma(bbs[j], L, s1);
            if (diag) s2 = fma(iKrow[j], L, s2);
        }
        s1 *= md.beta[(long)a * npad + i];
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

static int pair_njb(int npad, int PL) {
    const int nb = npad / 64;
    int njb = 1;
    while (njb * 2 <= nb && nb % (njb * 2) == 0 && (long)PL * nb * njb < 1536) njb *= 2;
    const char* env = getenv("PILCO_PAIR_NJB");
    if (env) {
        const int v = atoi(env);
        if (v >= 1 && v <= nb && nb % v == 0) njb = v;
    }
    return njb;
}

int mm_pair_nt(int npad, int variant, int PL) {
    if (variant == 1) return ((npad + 255) / 256) * (npad / 64);
    return (npad / 64) * pair_njb(npad, PL);
}

void launch_mm_pair(hipStream_t st, const MMModel& md, const MMWork& wk, int variant) {
    const int KP = wk.KP;
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
    const int NJB = wk.NT / (md.npad / 64);
    dim3 grid((md.npad / 64) * NJB, wk.PL);
#define PM(K_) hipLaunchKernelGGL((k_mm_pair_mfma<K_>), grid, dim3(256), 0, st, md, wk, NJB)
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
}

// ------------------------------------------------------------------ glue (one workgroup)
struct GlueLds {
    double* mx;   // [nm]
    double* sx;   // [nm*nm]
    double* mu;   // [nm]
    double* su;   // [nm*nm]
    double* cxu;  // [nm*nm]
    double* t1;   // [nm*nm]
    double* t2;   // [nm*nm]
    double* G0;   // [2*nm*nm]
    double* G1;   // [2*nm*nm]
    double* misc; // [64 + 34*34]
};

size_t glue_lds_bytes(int E, int D) {
    const int nm = E > D ? E : D;
    return sizeof(double) * (size_t)(2 * nm + 9 * nm * nm + 64 + 34 * 34);
}

// mean (and optionally variance) of exp(-(x-t)^T W (x-t)/2), x ~ N(m, s): rewards.py:32-48
__device__ double exp_reward_mean(const GlueLds& L, int E, const double* W, const double* tg, double scale) {
    // aug = [(I + scale*SW)^T | W^T]  ->  X^T with X = W (I + scale*SW)^{-1}
    const int nc = 2 * E;
    for (int e = threadIdx.x; e < E * nc; e += blockDim.x) {
        const int r = e / nc, c = e - r * nc;
        double v;
        if (c < E) {
            double sw = 0.0;  // (S W)[c][r]
            for (int k = 0; k < E; ++k) sw = fma(L.sx[c * E + k], W[k * E + r], sw);
            v = fma(scale, sw, (r == c) ? 1.0 : 0.0);
        } else {
            v = W[(c - E) * E + r];
        }
        L.G0[e] = v;
    }
    double det;
    double* res = gauss_jordan(L.G0, L.G1, E, nc, det);
    // quad = d X d^T with X^T in res[:, E:]
    double q = 0.0;
    if (threadIdx.x == 0) {
        for (int r = 0; r < E; ++r) {
            const double dr = L.mx[r] - tg[r];
            double acc = 0.0;
            for (int c = 0; c < E; ++c) acc = fma(res[c * nc + E + r], L.mx[c] - tg[c], acc);
            q = fma(dr, acc, q);
        }
        L.misc[0] = exp(-0.5 * scale * q) / sqrt(det);
    }
    __syncthreads();
    const double out = L.misc[0];
    __syncthreads();
    return out;
}

__device__ void reward_eval(const GlueArgs& g, const GlueLds& L, double& mu_out, double& var_out, bool want_var) {
    const int E = g.E;
    double mu = 0.0, var = 0.0;
    for (int i = 0; i < g.n_rewards; ++i) {
        const RewardDev& rw = g.rw[i];
        double m_i = 0.0, v_i = 0.0;
        if (rw.kind == PILCO_REWARD_EXPONENTIAL) {
            m_i = exp_reward_mean(L, E, rw.W, rw.t, 1.0);
            if (want_var) {
                const double r2 = exp_reward_mean(L, E, rw.W, rw.t, 2.0);
                v_i = r2 - m_i * m_i;
            }
        } else {  // linear: rewards.py:58-61
            for (int k = 0; k < E; ++k) m_i = fma(L.mx[k], rw.W[k], m_i);
            if (want_var)
                for (int r = 0; r < E; ++r)
                    for (int c = 0; c < E; ++c) v_i = fma(rw.W[r] * L.sx[r * E + c], rw.W[c], v_i);
        }
        mu = fma(rw.coef, m_i, mu);
        var = fma(rw.coef * rw.coef, v_i, var);
    }
    mu_out = mu;
    var_out = var;
}

// squash_sin on (mu[U], su[U][U]) in place; cdiag[u] = e_u exp(-s_uu/2) cos(m_u)   (controllers.py:13-36)
__device__ void squash_inplace(const GlueLds& L, int U, const double* maxact, double* cdiag) {
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
        L.misc[32 + t] = eu * ex * sin(L.mu[t]);
    }
    __syncthreads();
    for (int e = t; e < U * U; e += blockDim.x) L.su[e] = L.t2[e];
    if (t < U) L.mu[t] = L.misc[32 + t];
    __syncthreads();
}

// joint Gaussian of (x,u) from mx,sx,mu,su,cxu in LDS -> in_m, in_s, s1 (pilco.py:141-144)
__device__ void write_joint(const GlueArgs& g, const GlueLds& L) {
    const int E = g.E, U = g.U, D = g.D, t = threadIdx.x;
    // sc = s_x c_xu  (E,U)
    for (int e = t; e < E * U; e += blockDim.x) {
        const int r = e / U, u = e - r * U;
        double acc = 0.0;
        for (int k = 0; k < E; ++k) acc = fma(L.sx[r * E + k], L.cxu[k * U + u], acc);
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
    }
    __syncthreads();
}


// Reduce the tile partials of the local pairs / owned outputs into this rank's
// segment of the gather buffer.  Fixed summation order (tile index, then chunk
// index): results do not depend on the number of ranks.
__device__ void mm_pack(const MMWork& wk, int D, double* scratch) {
    const int t = threadIdx.x;
    double* seg = wk.gath + (long)wk.rank * wk.SEG;
    for (int k = t; k < wk.PL; k += blockDim.x) {
        int a, b;
        decode_pair(wk.pair_list[k], a, b);
        const double* part = wk.pair_part + (long)k * wk.NT * 2;
        double s0 = 0.0, s1 = 0.0;
        for (int q = 0; q < wk.NT; ++q) {
            s0 += part[2 * q];
            s1 += part[2 * q + 1];
        }
        seg[k] = ((a == b) ? (s0 - s1) : s0) * wk.pair_isdet[k];   // mgpr.py:144-145
    }
    const int W1 = 1 + D;
    for (int e = t; e < wk.EL * W1; e += blockDim.x) {
        const int o = e / W1, idx = e - o * W1;
        const int a = wk.own_outputs[o];
        double s = 0.0;
        for (int ch = 0; ch < wk.NCH; ++ch) s += wk.mean_part[((long)a * wk.NCH + ch) * W1 + idx];
        scratch[e] = s;
    }
    __syncthreads();
    for (int e = t; e < wk.EL * W1; e += blockDim.x) {
        const int o = e / W1, idx = e - o * W1;
        const int a = wk.own_outputs[o];
        const double ca = wk.c[a];
        double v;
        if (idx == 0) {
            v = ca * scratch[o * W1];                              // M_a          (mgpr.py:117)
        } else {
            const int d = idx - 1;
            double acc = 0.0;
            for (int k = 0; k < D; ++k) acc = fma(wk.T[((long)a * D + d) * D + k], scratch[o * W1 + 1 + k], acc);
            v = ca * acc;                                          // V_a[d]       (mgpr.py:118)
        }
        seg[wk.OUTOFF + e] = v;
    }
    __syncthreads();
}

// gath -> out_M [E], out_S [E][E], out_V [D][E]; also left in LDS (oM, oS, oV)
__device__ void mm_assemble(const MMWork& wk, const double* var, int D, int E, double* oM, double* oS, double* oV) {
    const int t = threadIdx.x;
    for (int a = t; a < E; a += blockDim.x) {
        const double v = wk.gath[wk.asm_out_src[a]];
        oM[a] = v;
        wk.out_M[a] = v;
    }
    for (int e = t; e < D * E; e += blockDim.x) {
        const int d = e / E, a = e - d * E;
        const double v = wk.gath[wk.asm_out_src[a] + 1 + d];
        oV[e] = v;
        wk.out_V[e] = v;
    }
    __syncthreads();
    for (int e = t; e < E * E; e += blockDim.x) {
        const int a = e / E, b = e - a * E;
        const int hi = a > b ? a : b, lo = a > b ? b : a;
        double v = wk.gath[wk.asm_pair_src[hi * (hi + 1) / 2 + lo]];
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
    GlueLds L;
    L.mx = sm;
    L.sx = L.mx + nm;
    L.mu = L.sx + nm * nm;
    L.su = L.mu + nm;
    L.cxu = L.su + nm * nm;
    L.t1 = L.cxu + nm * nm;
    L.t2 = L.t1 + nm * nm;
    L.G0 = L.t2 + nm * nm;
    L.G1 = L.G0 + 2 * nm * nm;
    L.misc = L.G1 + 2 * nm * nm;

    if (g.flags & GF_PACK) mm_pack(g.wk, D, L.misc + 64);
    if (g.flags & GF_ASSEMBLE) {
        // oM -> mu, oS -> su, oV -> cxu (LDS scratch reused)
        mm_assemble(g.wk, g.var, D, E, L.mu, L.su, L.cxu);
    }
    if (g.flags & GF_PROPAGATE) {
        // t1 = s1 V (E,E); state += increment                      (pilco.py:147-149)
        for (int e = t; e < E * E; e += blockDim.x) {
            const int r = e / E, c = e - r * E;
            double acc = 0.0;
            for (int k = 0; k < D; ++k) acc = fma(g.s1[r * D + k], L.cxu[k * E + c], acc);
            L.t1[e] = acc;
        }
        __syncthreads();
        for (int e = t; e < E * E; e += blockDim.x) {
            const int r = e / E, c = e - r * E;
            g.s_x[e] = ((L.su[e] + g.s_x[e]) + L.t1[e]) + L.t1[c * E + r];
        }
        if (t < E) g.m_x[t] = L.mu[t] + g.m_x[t];
        __syncthreads();
    }
    if (g.flags & (GF_TRAJ | GF_REWARD | GF_POLICY | GF_RBF_PRE)) {
        if (t < E) L.mx[t] = g.m_x[t];
        for (int e = t; e < E * E; e += blockDim.x) L.sx[e] = g.s_x[e];
        __syncthreads();
    }
    if ((g.flags & GF_TRAJ) && g.traj) {
        double* dst = g.traj + (long)g.step * (E + E * E);
        if (t < E) dst[t] = L.mx[t];
        for (int e = t; e < E * E; e += blockDim.x) dst[E + e] = L.sx[e];
    }
    if (g.flags & GF_REWARD) {
        double mu, var;
        reward_eval(g, L, mu, var, g.rew_out != nullptr);
        if (t == 0) {
            if (g.rew_out) {
                g.rew_out[0] = mu;
                g.rew_out[1] = var;
            } else {
                g.reward[0] += mu;                                  // pilco.py:133
            }
        }
        __syncthreads();
    }
    if (g.flags & GF_POLICY) {
        if (g.pol_kind == PILCO_POLICY_LINEAR) {
            // M = m W^T + b, S = W s W^T, V = W^T                  (controllers.py:52-54)
            if (t < U) {
                double acc = g.b[t];
                for (int k = 0; k < E; ++k) acc = fma(g.W[t * E + k], L.mx[k], acc);
                L.mu[t] = acc;
            }
            for (int e = t; e < U * E; e += blockDim.x) {
                const int u = e / E, c = e - u * E;
                double acc = 0.0;
                for (int k = 0; k < E; ++k) acc = fma(g.W[u * E + k], L.sx[k * E + c], acc);
                L.t1[e] = acc;  // W s
                L.cxu[c * U + u] = g.W[e];
            }
            __syncthreads();
            for (int e = t; e < U * U; e += blockDim.x) {
                const int u = e / U, v = e - u * U;
                double acc = 0.0;
                for (int k = 0; k < E; ++k) acc = fma(L.t1[u * E + k], g.W[v * E + k], acc);
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
}

void launch_glue(hipStream_t st, const GlueArgs& g) {
    const size_t lds = glue_lds_bytes(g.E, g.D);
    static size_t configured = 0;
    if (lds > configured) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_glue), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)lds);
        configured = lds;
    }
    hipLaunchKernelGGL(k_glue, dim3(1), dim3(256), lds, st, g);
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
    for (int r = 0; r < 4; ++r) out[lane * 4 + r] = acc[r];
}

int launch_selftest_mfma(hipStream_t st, double* dbuf, double* hbuf) {
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
