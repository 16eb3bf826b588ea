// Dense float64 building blocks of the GP factorisation (T1-T4, T20 of SURVEY.md
// section 2.2), hand-written for gfx950:
//   * SE-ARD Gram build (coalesced reads of the transposed point set),
//   * 64x64x16 LDS-tiled batched GEMM on v_mfma_f64_16x16x4_f64,
//   * right-looking blocked Cholesky (diag block factor + inverse in LDS,
//     panel and trailing update on the MFMA GEMM),
//   * blocked triangular inverse, mat-vec, padding helpers.
// Every device matrix is padded to a multiple of 64 with an identity / zero
// padding so that no kernel needs edge handling.
#include "common.h"

namespace pilco {

typedef double d4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------ Gram
// replaces gpflow SquaredExponential.K as called from pilco/models/mgpr.py:154-157
// One workgroup: GRAM_R rows x 256 columns.  A thread owns one column: its point x_j is loaded once per dimension and serves
// the GRAM_R rows (whose points and the reciprocal lengthscales sit in LDS); every element is summed over d = 0 .. D-1 in
// that order, as before.  (Round 2: one row per workgroup, 1 / l_d divided out D times per element: 80 us for ten 1000^2
// matrices; now bound by the 10^7 double-precision exp and the write.)
constexpr int GRAM_R = 8;
__global__ __launch_bounds__(256) void k_gram(const double* __restrict__ P1t, int ld1, int n1,
                                              const double* __restrict__ P2t, int ld2, int n2, int D,
                                              const double* __restrict__ ls, const double* __restrict__ var,
                                              double* __restrict__ out, int rows_pad, int cols_pad, int diag_mode,
                                              const double* __restrict__ diag_add, double jitter, long sP1, long sP2) {
    const int a = blockIdx.z;
    P1t += (long)a * sP1;   // per-output point sets (FITC training: every output owns its inducing inputs); 0 = shared
    P2t += (long)a * sP2;
    const int i0 = blockIdx.y * GRAM_R;
    const int j = blockIdx.x * 256 + threadIdx.x;
    // diag_mode != 0: a symmetric matrix on its way to launch_potrf, which reads the diagonal tiles and the tiles below
    // them only (so do launch_trtri and the log-determinant after it): segments wholly to the right of the rows' diagonal
    // tile are left unwritten (the GRAM_R rows of a workgroup share their 64-row tile)
    if (diag_mode != 0 && (int)blockIdx.x * 256 > (i0 | 63)) return;
    __shared__ double il_s[32], xi_s[GRAM_R][33];
    for (int e = threadIdx.x; e < GRAM_R * D; e += 256) {
        const int r = e / D, d = e - r * D;
        xi_s[r][d] = (i0 + r < n1) ? P1t[(long)d * ld1 + i0 + r] : 0.0;
    }
    if ((int)threadIdx.x < D) il_s[threadIdx.x] = 1.0 / ls[a * D + threadIdx.x];
    __syncthreads();
    if (j >= cols_pad) return;
    double r2[GRAM_R];
#pragma unroll
    for (int r = 0; r < GRAM_R; ++r) r2[r] = 0.0;
    if (j < n2)
        for (int d = 0; d < D; ++d) {
            const double xj = P2t[(long)d * ld2 + j], il = il_s[d];
#pragma unroll
            for (int r = 0; r < GRAM_R; ++r) {
                const double diff = (xi_s[r][d] - xj) * il;
                r2[r] = fma(diff, diff, r2[r]);
            }
        }
    const double va = var[a];
#pragma unroll
    for (int r = 0; r < GRAM_R; ++r) {
        const int i = i0 + r;
        if (i >= rows_pad) break;
        double v;
        if (i < n1 && j < n2) {
            v = va * exp(-0.5 * r2[r]);
            if (i == j) {
                if (diag_mode == 1) v += diag_add[a];
                if (diag_mode == 2) v += jitter;
            }
        } else {
            v = (diag_mode != 0 && i == j) ? 1.0 : 0.0;
        }
        out[((long)a * rows_pad + i) * cols_pad + j] = v;
    }
}

void launch_gram(hipStream_t st, const double* P1t, int ld1, int n1, const double* P2t, int ld2, int n2, int D,
                 const double* ls, const double* var, int E, double* out, int rows_pad, int cols_pad, int diag_mode,
                 const double* diag_add, double jitter, long sP1, long sP2) {
    dim3 grid((cols_pad + 255) / 256, (rows_pad + GRAM_R - 1) / GRAM_R, E);
    hipLaunchKernelGGL(k_gram, grid, dim3(256), 0, st, P1t, ld1, n1, P2t, ld2, n2, D, ls, var, out, rows_pad, cols_pad,
                       diag_mode, diag_add, jitter, sP1, sP2);
}

// (blockIdx.y: one of `batch` point sets, sX / sXt doubles apart -- the E inducing-point sets of the FITC objective used to be E launches)
__global__ void k_transpose_points(const double* __restrict__ X, int n, int D, double* __restrict__ Xt, int ld, long sX, long sXt) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= ld) return;
    X += (long)blockIdx.y * sX;
    Xt += (long)blockIdx.y * sXt;
    for (int d = 0; d < D; ++d) Xt[(long)d * ld + i] = (i < n) ? X[(long)i * D + d] : 0.0;
}

void launch_transpose_points(hipStream_t st, const double* X, int n, int D, double* Xt, int ld, int batch, long sX, long sXt) {
    hipLaunchKernelGGL(k_transpose_points, dim3((ld + 255) / 256, batch), dim3(256), 0, st, X, n, D, Xt, ld, sX, sXt);
}

// ------------------------------------------------------------------ GEMM (f64 MFMA)
// Block tile 64x64, K step 16, 4 waves in a 2x2 arrangement, each wave a 32x32
// sub-tile = 2x2 v_mfma_f64_16x16x4_f64 accumulators.  LDS tiles are stored
// k-major ([k][i] / [k][j]) so that an MFMA operand read is 16 consecutive
// doubles per k row; the row stride of 80 doubles puts the two k rows of a
// 32-lane ds_read_b64 group on disjoint bank halves.
constexpr int LDS_LD = 80;

// (the block factorisation of the Cholesky section below, which the GEMM's fused form calls)
constexpr int POTF2_LD = 66;   // 16-byte aligned lines, lanes of a column spread over the banks
struct Potf2Lds {               // LDS of one block factorisation
    double Ls[64 * POTF2_LD];   // the block (row-major): A, then L below the diagonal, then L^-1 (potf2_block)
    double rinvs[64];           // 1 / sqrt(pivot j)
    double dump_d[64];          // where lanes past 0 put what only lane 0 has to store (branch-free)
    double line[64];            // the column being eliminated with, for broadcast reads
};
template <int NW>
__device__ __forceinline__ void potf2_block(Potf2Lds& S, double* __restrict__ A, int npad, int kb_abs, double* __restrict__ out, int out_ld,
                                            int* info_word, int lane, int w, unsigned long long* stp);

// POTF2: the workgroup of tile (0, 0) goes on to factor and invert that tile (the next diagonal block of the Cholesky chain)
template <bool TA, bool TB, bool POTF2 = false>
__global__ __launch_bounds__(256) void k_gemm64(GemmDesc g) {
    kernarg_warm<(int)sizeof(GemmDesc) + 64>();   // (the factorisation is a chain of ~50 dependent launches: every prologue is on its critical path)
    // Grid (column tiles, matrices x sub-problems, row tiles): the row tile is the SLOWEST index, so that the tile rows with
    // the longest K range (k_mode 1 / 2: the first rows; k_mode 3 / 4: the last, taken first) start on every matrix before any
    // short row does.  (Round 2 had the matrix slowest: the last matrix's longest tiles started when the chip was already
    // full of short ones and ran alone at the end -- the iK GEMM took 210 us whether all tiles or half of them were computed.)
    // ... and XCD-aware: workgroups go to the 8 XCDs (one L2 each) round-robin in launch order, so the tiles of one tile row
    // -- which share their A operand -- would land on eight different L2s.  Launch slot -> tile: the slots one XCD receives
    // are dealt whole tile rows (row `r` of the (matrix, tile row) list goes to XCD r % 8).
    int bj = blockIdx.x, byz = (int)blockIdx.y + (int)gridDim.y * (int)blockIdx.z;
    {
        const int gx = gridDim.x, nyz = gridDim.y * gridDim.z;
        if ((nyz & 7) == 0) {
            const int L = bj + gx * byz, xcd = L & 7, sq = L >> 3;   // the sq-th workgroup this XCD receives
            byz = (sq / gx) * 8 + xcd;
            bj = sq - (sq / gx) * gx;
        }
    }
    const int bz = byz % (int)gridDim.y, bzi = byz / (int)gridDim.y;
    const int bi = (g.k_mode == 3 || g.k_mode == 4) ? (int)gridDim.z - 1 - bzi : bzi;
    if (g.tile_mode != 0 && bi < bj) return;
    const int i0 = bi * 64, j0 = bj * 64;
    int kbeg = 0, kend = g.K;
    if (g.k_mode == 1) kbeg = (i0 > j0 ? i0 : j0);
    if (g.k_mode == 2 || g.k_mode == 4) kbeg = j0;
    if (g.k_mode == 3 || g.k_mode == 4) kend = (i0 + 64 < g.K) ? i0 + 64 : g.K;
    kbeg &= ~15;
    // operand chunks; tile_mode 2 reuses the space (and the rest of `sm`) to turn the finished tile around for its mirror image
    constexpr int SM_DOUBLES = POTF2 ? (int)((sizeof(Potf2Lds) + 7) / 8) : 64 * 65;
    __shared__ __attribute__((aligned(16))) double sm[SM_DOUBLES];
    double (*As)[LDS_LD] = reinterpret_cast<double (*)[LDS_LD]>(sm);
    double (*Bs)[LDS_LD] = reinterpret_cast<double (*)[LDS_LD]>(sm + 16 * LDS_LD);
    int mat = bz, sub = 0;
    if (g.nsub > 0) {
        mat = bz / g.nsub;
        sub = bz - mat * g.nsub;
        if (i0 >= g.sub_rows0 - sub * g.sub_rows_step) return;
    }
    int ldc = g.ldc;
    double beta = g.beta;
    long split_off = -1;
    if (g.ksplit > 1) {   // this workgroup's K slice and its place in the workspace
        const int nb = (int)gridDim.y / g.ksplit, sp = bz / nb;
        mat = bz - sp * nb;
        const int span = (((kend - kbeg) + g.ksplit - 1) / g.ksplit + 15) & ~15;
        kbeg += sp * span;
        kend = kbeg + span < kend ? kbeg + span : kend;
        split_off = ((long)sp * nb + mat) * g.M * g.N;
        ldc = g.N;
        beta = 0.0;
    }
    const double* A = g.A + (long)mat * g.sA + (long)sub * g.ssA;
    const double* B = g.B + (long)mat * g.sB + (long)sub * g.ssB;
    double* C = split_off >= 0 ? g.split_ws + split_off : g.C + (long)mat * g.sC + (long)sub * g.ssC;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int wi = (w >> 1) * 32, wj = (w & 1) * 32;
    const int lr = lane >> 4, lc = lane & 15;
    d4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = d4{0.0, 0.0, 0.0, 0.0};

    // beta != 0: the tile's old values are requested TOGETHER and ahead of the K loop -- read where they are used, as in round
    // 2, they cost sixteen memory round trips in a row per thread: most of a K = 64 update's 8 us
    const double scale = g.alpha * (g.alpha_vec ? g.alpha_vec[mat] : 1.0);
    double cold[2][2][4];
    if (beta != 0.0) {
#pragma unroll
        for (int ti = 0; ti < 2; ++ti)
#pragma unroll
            for (int tj = 0; tj < 2; ++tj)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    cold[ti][tj][r] = C[(long)(i0 + wi + 16 * ti + lr + 4 * r) * ldc + j0 + wj + 16 * tj + lc];
    }
    // The next K chunk's global loads are in flight while the current one is multiplied (round 2 loaded, stored, multiplied
    // in turn: the memory round trip of every chunk was hidden by other workgroups only); same sums in the same order.
    auto fetch = [&](int k0, double2& pa0, double2& pa1, double2& pb0, double2& pb1) {
        if (!TA) {  // A stored (M,K): rows contiguous along k
            const double* src = A + (long)(i0 + (t >> 2)) * g.lda + k0 + (t & 3) * 4;
            pa0 = *reinterpret_cast<const double2*>(src);
            pa1 = *reinterpret_cast<const double2*>(src + 2);
        } else {  // A stored (K,M): rows contiguous along i
            const double* src = A + (long)(k0 + (t >> 4)) * g.lda + i0 + (t & 15) * 4;
            pa0 = *reinterpret_cast<const double2*>(src);
            pa1 = *reinterpret_cast<const double2*>(src + 2);
        }
        if (!TB) {  // B stored (K,N)
            const double* src = B + (long)(k0 + (t >> 4)) * g.ldb + j0 + (t & 15) * 4;
            pb0 = *reinterpret_cast<const double2*>(src);
            pb1 = *reinterpret_cast<const double2*>(src + 2);
        } else {  // B stored (N,K)
            const double* src = B + (long)(j0 + (t >> 2)) * g.ldb + k0 + (t & 3) * 4;
            pb0 = *reinterpret_cast<const double2*>(src);
            pb1 = *reinterpret_cast<const double2*>(src + 2);
        }
    };
    auto stage = [&](const double2& pa0, const double2& pa1, const double2& pb0, const double2& pb1) {
        if (!TA) {  // transposing write
            const int i = t >> 2, kq = (t & 3) * 4;
            As[kq + 0][i] = pa0.x;
            As[kq + 1][i] = pa0.y;
            As[kq + 2][i] = pa1.x;
            As[kq + 3][i] = pa1.y;
        } else {
            const int k = t >> 4, iq = (t & 15) * 4;
            *reinterpret_cast<double2*>(&As[k][iq]) = pa0;
            *reinterpret_cast<double2*>(&As[k][iq + 2]) = pa1;
        }
        if (!TB) {
            const int k = t >> 4, jq = (t & 15) * 4;
            *reinterpret_cast<double2*>(&Bs[k][jq]) = pb0;
            *reinterpret_cast<double2*>(&Bs[k][jq + 2]) = pb1;
        } else {
            const int j = t >> 2, kq = (t & 3) * 4;
            Bs[kq + 0][j] = pb0.x;
            Bs[kq + 1][j] = pb0.y;
            Bs[kq + 2][j] = pb1.x;
            Bs[kq + 3][j] = pb1.y;
        }
    };
    auto multiply = [&]() {
#pragma unroll
        for (int kk = 0; kk < 16; kk += 4) {
            const double a0 = As[kk + lr][wi + lc];
            const double a1 = As[kk + lr][wi + 16 + lc];
            const double b0 = Bs[kk + lr][wj + lc];
            const double b1 = Bs[kk + lr][wj + 16 + lc];
            acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
        }
    };
    // (K = 64 launches of the Cholesky chain with ALL FOUR chunks requested at once: no faster -- 0.896 vs 0.904 ms --, dropped)
    double2 pa0, pa1, pb0, pb1;
    if (kbeg < kend) fetch(kbeg, pa0, pa1, pb0, pb1);
    for (int k0 = kbeg; k0 < kend; k0 += 16) {
        stage(pa0, pa1, pb0, pb1);
        __syncthreads();
        if (k0 + 16 < kend) fetch(k0 + 16, pa0, pa1, pb0, pb1);
        multiply();
        __syncthreads();
    }
    // f64 MFMA C/D layout: col = lane & 15, row = (lane >> 4) + 4 * reg
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = i0 + wi + 16 * ti + lr + 4 * r;
                const int col = j0 + wj + 16 * tj + lc;
                double* c = C + (long)row * ldc + col;
                double v = scale * acc[ti][tj][r];
                if (beta != 0.0) v = fma(beta, cold[ti][tj][r], v);
                if (POTF2 && bi == 0 && bj == 0)   // stays in LDS: the factor replaces it in memory
                    reinterpret_cast<Potf2Lds*>(sm)->Ls[(wi + 16 * ti + lr + 4 * r) * POTF2_LD + wj + 16 * tj + lc] = v;
                else
                    *c = v;
                if (g.tile_mode == 2 && bi != bj) sm[(wj + 16 * tj + lc) * 65 + wi + 16 * ti + lr + 4 * r] = v;
            }
    if (POTF2 && bi == 0 && bj == 0) {
        // Cholesky chain: this tile is the NEXT diagonal block.  Its factorisation and inverse (12 us, this workgroup's four
        // waves) run here, beside the other tiles of the trailing update, instead of in a launch of their own behind it.
        Potf2Lds& S = *reinterpret_cast<Potf2Lds*>(sm);
        __syncthreads();
        potf2_block<4>(S, C, ldc, g.potf2_kb, g.potf2_X + (long)mat * g.potf2_sX + (long)g.potf2_kb * 64 * (ldc + 1), ldc,
                       g.potf2_info + mat, lane, w, nullptr);
    }
    // symmetric result: the tile above the diagonal is this one's mirror image -- turned around in LDS and stored as full rows
    // (stored straight from the accumulators it is 4096 scattered 8-byte writes per tile: as slow as computing it)
    if (g.tile_mode == 2 && bi != bj) {
        __syncthreads();
        for (int e = t; e < 4096; e += 256) C[(long)(j0 + (e >> 6)) * ldc + i0 + (e & 63)] = sm[(e >> 6) * 65 + (e & 63)];
    }
}

// adds the K slices of a split product up in their order: C = alpha' sum_s ws[s] + beta C   (alpha is already in the slices)
__global__ __launch_bounds__(256) void k_gemm_split_reduce(GemmDesc g, int batch) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x, mn = (long)g.M * g.N;
    const int mat = blockIdx.y;
    if (e >= mn) return;
    double v = 0.0;
    for (int sp = 0; sp < g.ksplit; ++sp) v += g.split_ws[((long)sp * batch + mat) * mn + e];
    double* c = g.C + (long)mat * g.sC + (e / g.N) * g.ldc + (e % g.N);
    *c = g.beta != 0.0 ? fma(g.beta, *c, v) : v;
}

void launch_gemm(hipStream_t st, const GemmDesc& g_in, bool ta, bool tb, int batch) {
    GemmDesc g = g_in;
    if (g.M <= 0 || g.N <= 0) return;
    if (g.ksplit <= 1 || !g.split_ws || g.nsub > 0 || g.tile_mode == 1 || g.potf2_X) g.ksplit = 0;   // (tile_mode 2: every slice is mirrored inside the workspace)
    dim3 grid(g.N / 64, batch * (g.nsub > 0 ? g.nsub : 1) * (g.ksplit > 1 ? g.ksplit : 1), g.M / 64);
    if (!ta && !tb) hipLaunchKernelGGL((k_gemm64<false, false>), grid, dim3(256), 0, st, g);
    if (!ta && tb && g.potf2_X) hipLaunchKernelGGL((k_gemm64<false, true, true>), grid, dim3(256), 0, st, g);
    else if (!ta && tb) hipLaunchKernelGGL((k_gemm64<false, true>), grid, dim3(256), 0, st, g);
    if (ta && !tb) hipLaunchKernelGGL((k_gemm64<true, false>), grid, dim3(256), 0, st, g);
    if (ta && tb) hipLaunchKernelGGL((k_gemm64<true, true>), grid, dim3(256), 0, st, g);
    if (g.ksplit > 1)
        hipLaunchKernelGGL(k_gemm_split_reduce, dim3((unsigned)(((long)g.M * g.N + 255) / 256), batch), dim3(256), 0, st, g, batch);
}

// ------------------------------------------------------------------ Cholesky
// Factor the kb-th 64x64 diagonal block and invert the factor: potf2_block below, four waves per matrix.
// (History.  Round 2: one wave, factor then inverse, 38 us per block on the factorisation's critical path -- 16 blocks at
// N = 1000.  Rounds 3-4: two waves, all 64 columns eliminated in wave 0's registers -- every column published as an LDS line
// and read back as broadcasts --, wave 1 running the substitution for the inverse one column behind: 20 us.  Round 5:
// blocked in panels of 16 columns with MFMA updates and a recursive-doubling inverse: 12 us.)
// replaces tf.linalg.cholesky at pilco/models/mgpr.py:84 / smgpr.py:29,35
__device__ __forceinline__ double rsqrt_f64(double d) {
    double y = __builtin_amdgcn_rsq(d);
    y = fma(y, fma(-0.5 * d * y, y, 0.5), y);
    y = fma(y, fma(-0.5 * d * y, y, 0.5), y);
    return y;
}
__device__ __forceinline__ double lane_bcast(double v, int l) {   // v of lane l (l wave-uniform) as a scalar
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
#ifdef POTF2_STAMPS   // developer aid (tools/ubench_potf2.hip): 100 MHz stamps behind the info words
#define POTF2_STAMP(i_) do { if (stp && lane == 0) stp[i_] = wall_clock64(); } while (0)
#else
#define POTF2_STAMP(i_) do { } while (0)
#endif
// The block is in S.Ls (row-major, row stride POTF2_LD) and every thread of the workgroup (FOUR waves) has passed a barrier
// since.  Leaves L (zero above the diagonal) in the global block A and L^-1 in `out`; *info_word receives
// kb_abs * 64 + column + 1 of the first non-positive pivot.
//
// Blocked inside the block (round 5), panels of 16 columns:
//   factor: wave 0 holds row i's 16 panel entries in lane i and eliminates the panel's columns in registers (avg 7.5
//     updates per column instead of 31 over the whole block; pivot and the next column's multiplier by v_readlane, the
//     other multipliers from an LDS line); the rank-16 update of the columns behind the panel is MFMA work on the LDS copy
//     (6 / 3 / 1 tiles of 16 x 16): the next panel's column of tiles first, the others beside the next panel's elimination.
//     Rounds 3-4 eliminated all 64 columns in one wave's registers: 2016 fused multiply-adds and 1008 LDS reads issued by
//     ONE wave, 16 us of the 20 us this block takes on the chain's critical path.
//   inverse, in place behind the stored factor: the four 16 x 16 diagonal blocks by substitution in one wave (lane
//     16 d + j: column j of block d), then two levels of recursive doubling, X_BA = -X_BB (L_BA X_AA), on MFMA; the product
//     in brackets is staged in the block's unused upper-right quarter.
//     (Rounds 3-4: a second wave ran the substitution over all 64 columns one column behind the factor wave.)
__device__ __forceinline__ d4 potf2_tile_load(const double* Ls, int r0, int c0, int lr, int lc) {
    d4 v;
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = Ls[(r0 + lr + 4 * r) * POTF2_LD + c0 + lc];
    return v;
}
__device__ __forceinline__ void potf2_tile_store(double* Ls, int r0, int c0, int lr, int lc, d4 v) {
#pragma unroll
    for (int r = 0; r < 4; ++r) Ls[(r0 + lr + 4 * r) * POTF2_LD + c0 + lc] = v[r];
}
// acc += sign * P Q for 16 x 16 blocks of the LDS copy: P at (pr, pc) read as rows, Q at (qr, qc) -- QT: Q^T, i.e. the
// operand is the block at (qr, qc) read as rows too (the rank-16 update L L^T)
template <bool QT>
__device__ __forceinline__ d4 potf2_tile_mma(const double* Ls, int pr, int pc, int qr, int qc, double sign, d4 acc, int lr, int lc) {
#pragma unroll
    for (int kk = 0; kk < 16; kk += 4) {
        const double a = sign * Ls[(pr + lc) * POTF2_LD + pc + kk + lr];
        const double b = QT ? Ls[(qr + lc) * POTF2_LD + qc + kk + lr] : Ls[(qr + kk + lr) * POTF2_LD + qc + lc];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
        if (kk == 0) {   // (a first MFMA on a constant-zero accumulator: see mm_device.h, MFMA_KEEP_ALIVE)
            asm volatile("" ::"v"(a));
            asm volatile("" ::"v"(b));
        }
    }
    return acc;
}
template <int NW>
__device__ __forceinline__ void potf2_block(Potf2Lds& S, double* __restrict__ A, int npad, int kb_abs, double* __restrict__ out, int out_ld,
                                            int* info_word, int lane, int w, unsigned long long* stp) {
    static_assert(NW == 4, "four waves");
    constexpr int LD = POTF2_LD;
    (void)stp;
    const int lr = lane >> 4, lc = lane & 15;
    double* Ls = S.Ls;
    int bad = 0;
    if (w == 0) POTF2_STAMP(1);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int c0 = 16 * p;
        if (w == 0) {
            double a[16];
#pragma unroll
            for (int c = 0; c < 16; c += 2) {
                const double2 pr = *reinterpret_cast<const double2*>(&Ls[lane * LD + c0 + c]);
                a[c] = (lane >= c0) ? pr.x : 0.0;
                a[c + 1] = (lane >= c0) ? pr.y : 0.0;
            }
            // (eliminating with UNSCALED columns -- the next pivot one reciprocal and one multiply-add on scalars behind the
            // previous, square roots beside the chain -- was no faster, 10.1 against 9.2 us for the four panels: the wave is
            // bound by the ~43 instructions it issues per column, about 5 cycles each, not by the pivot chain)
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                // (few instructions per column: a failed pivot is recorded and the elimination goes on with whatever it
                // yields -- the caller discards the factor --; rows above the diagonal carry along unread values instead of
                // zeros)
                const double d = lane_bcast(a[j], c0 + j);
                bad = (!(d > 0.0) && bad == 0) ? kb_abs * 64 + c0 + j + 1 : bad;   // (NaN too)
                const double rinv = rsqrt_f64(d);
                const double l = a[j] * rinv;   // lane i >= column: L[i][column]
                a[j] = l;
                // the inverse's diagonal: 1 / sqrt(pivot) as computed here.  (Taking 1 / L_jj in the inverse instead -- one
                // division for all columns, no store per column -- is as accurate, |X L - I| 1.7e-16 against 2.2e-16, but moved
                // the cond(K) = 1e9 fixture of tests/test_gpu_parity.py::test_predictions_golden from inside its 1e-5 of the
                // 40-digit truth to 1.18e-5: the arithmetic the fixtures were pinned with stays)
                *((lane == 0) ? &S.rinvs[c0 + j] : &S.dump_d[lane]) = rinv;   // (branch-free: lanes past 0 store into a dump)
                // multipliers: the next column's from v_readlane (the next pivot waits for nothing else), the others read back
                // from an LDS line as broadcasts, two per instruction (from v_readlane all of them: two instructions each,
                // 240 of a panel's ~700)
                S.line[lane] = l;
                if (j < 15) a[j + 1] = fma(-l, lane_bcast(l, c0 + j + 1), a[j + 1]);
                if (((j + 2) & 1) && j + 2 < 16) a[j + 2] = fma(-l, S.line[c0 + j + 2], a[j + 2]);
#pragma unroll
                for (int c = (j + 3) & ~1; c < 16; c += 2) {
                    const double2 pr = *reinterpret_cast<const double2*>(&S.line[c0 + c]);
                    a[c] = fma(-l, pr.x, a[c]);
                    a[c + 1] = fma(-l, pr.y, a[c + 1]);
                }
            }
#pragma unroll
            for (int c = 0; c < 16; c += 2) *reinterpret_cast<double2*>(&Ls[lane * LD + c0 + c]) = double2{a[c], a[c + 1]};
        }
        __syncthreads();
        if (p < 3) {
            // Columns behind the panel: tile (I, J) -= L_I L_J^T.  First the next panel's column of tiles (waves 1.., one
            // each) and one of the others (wave 0); the rest of the others run on waves 1.. BESIDE the next panel's
            // elimination -- they touch neither its columns nor anything it reads.
            auto update = [&](int I, int J) {
                d4 acc = potf2_tile_load(Ls, 16 * I, 16 * J, lr, lc);
                acc = potf2_tile_mma<true>(Ls, 16 * I, c0, 16 * J, c0, -1.0, acc, lr, lc);
                potf2_tile_store(Ls, 16 * I, 16 * J, lr, lc, acc);
            };
            if (w >= 1 && p + w < 4) update(p + w, p + 1);
            if (w == 0 && p < 2) update(p + 2, p + 2);
            __syncthreads();
            if (p == 0 && w == 1) update(3, 2);
            if (p == 0 && w == 2) update(3, 3);
        }
    }
    if (w == 0) {
        if (bad && lane == 0) atomicCAS(info_word, 0, bad);
        POTF2_STAMP(2);
    }
    // The factor goes out (coalesced rows; what sits above the diagonal in LDS is the block's old upper half) from waves 1-3,
    // while wave 0 inverts the four 16 x 16 diagonal blocks in registers: X_dd = L_dd^-1 row by row, lane 16 d + j holds column
    // j of block d.  They replace L_dd in LDS once the factor's rows have been read.
    double x[16];
    if (w == 0) {
        const int base = 16 * lr * LD + 16 * lr;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            double acc0 = (i == lc) ? 1.0 : 0.0, acc1 = 0.0;
#pragma unroll
            for (int k = 0; k < i; ++k) {
                const double lik = Ls[base + i * LD + k];
                if (k & 1) acc1 = fma(-lik, x[k], acc1);
                else acc0 = fma(-lik, x[k], acc0);
            }
            x[i] = (acc0 + acc1) * S.rinvs[16 * lr + i];
        }
    } else {
        for (int r = w - 1; r < 64; r += NW - 1) A[(long)r * npad + lane] = (lane <= r) ? Ls[r * LD + lane] : 0.0;
    }
    __syncthreads();
    if (w == 0) {
        const int base = 16 * lr * LD + 16 * lr;
#pragma unroll
        for (int i = 0; i < 16; ++i) Ls[base + i * LD + lc] = x[i];   // (zero above the diagonal: the recurrence leaves x[i] = 0 for i < j)
    }
    __syncthreads();
    if (w < 2) {   // first doubling: X_10 = -X_11 (L_10 X_00) (wave 0), X_32 likewise (wave 1); one wave each, LDS operations of a wave keep their order
        const int bi = 16 * (2 * w + 1), bj = 16 * (2 * w);
        d4 t = potf2_tile_mma<false>(Ls, bi, bj, bj, bj, 1.0, d4{0.0, 0.0, 0.0, 0.0}, lr, lc);
        potf2_tile_store(Ls, bi, bj, lr, lc, t);
        d4 x = potf2_tile_mma<false>(Ls, bi, bi, bi, bj, -1.0, d4{0.0, 0.0, 0.0, 0.0}, lr, lc);
        potf2_tile_store(Ls, bi, bj, lr, lc, x);
    }
    __syncthreads();
    {   // second doubling, 32 x 32 blocks: T = L_BA X_AA into the unused upper-right quarter, then X_BA = -X_BB T over L_BA
        const int I = 2 + (w >> 1), J = w & 1;   // this wave's 16 x 16 tile of the lower-left quarter
        d4 t = d4{0.0, 0.0, 0.0, 0.0};
        for (int K = J; K < 2; ++K) t = potf2_tile_mma<false>(Ls, 16 * I, 16 * K, 16 * K, 16 * J, 1.0, t, lr, lc);   // (X_AA is lower triangular: its block (0, 1) is not a zero in LDS)
        potf2_tile_store(Ls, 16 * (I - 2), 32 + 16 * J, lr, lc, t);
        __syncthreads();
        d4 x = d4{0.0, 0.0, 0.0, 0.0};
        for (int K = 2; K <= I; ++K) x = potf2_tile_mma<false>(Ls, 16 * I, 16 * K, 16 * (K - 2), 32 + 16 * J, -1.0, x, lr, lc);
        potf2_tile_store(Ls, 16 * I, 16 * J, lr, lc, x);
    }
    __syncthreads();
    if (w == 0) POTF2_STAMP(3);
    for (int r = w; r < 64; r += NW) out[(long)r * out_ld + lane] = (lane <= r) ? Ls[r * LD + lane] : 0.0;
    if (w == 0) POTF2_STAMP(5);
}

// the first diagonal block (the others are factored by the workgroup that finishes their trailing update: k_gemm64<.., true>)
constexpr int POTF2_NW = 4;   // waves of a block factorisation
__global__ __launch_bounds__(64 * POTF2_NW) void k_potf2_inv(double* __restrict__ Aall, int npad, int kb,
                                                   double* __restrict__ Xall, int* __restrict__ info) {
    constexpr int LD = POTF2_LD;
    __shared__ __attribute__((aligned(16))) Potf2Lds S;
    const int b = blockIdx.x;
    double* A = Aall + (long)b * npad * npad + (long)kb * 64 * npad + kb * 64;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    unsigned long long* stp = nullptr;
#ifdef POTF2_STAMPS
    if (b == 0) stp = reinterpret_cast<unsigned long long*>(info + 256);
#endif
    if (w == 0) POTF2_STAMP(0);
    {   // the block, 16 rows per wave, all loads in flight together (one by one they cost a memory round trip each: 10 us)
        double v[64 / POTF2_NW];
#pragma unroll
        for (int q = 0; q < 64 / POTF2_NW; ++q) v[q] = A[(long)(POTF2_NW * q + w) * npad + lane];
#pragma unroll
        for (int q = 0; q < 64 / POTF2_NW; ++q) S.Ls[(POTF2_NW * q + w) * LD + lane] = v[q];
    }
    __syncthreads();
    potf2_block<POTF2_NW>(S, A, npad, kb, Xall + (long)b * npad * npad + (long)kb * 64 * (npad + 1), npad, &info[b], lane, w, stp);
}
#undef POTF2_STAMP

void launch_potrf(hipStream_t st, double* A, int npad, int batch, double* Linv, int* info, bool zero_linv) {
    const int nblk = npad / 64;
    const long sA = (long)npad * npad;
    if (zero_linv) (void)hipMemsetAsync(Linv, 0, sizeof(double) * sA * batch, st);
    // block 0 has a launch of its own; block kb + 1 is factored inside the trailing update of step kb (k_gemm64<.., true>)
    hipLaunchKernelGGL(k_potf2_inv, dim3(batch), dim3(64 * POTF2_NW), 0, st, A, npad, 0, Linv, info);
    for (int kb = 0; kb < nblk; ++kb) {
        const int rem = nblk - kb - 1;
        if (rem <= 0) break;
        double* panel = A + (long)(kb + 1) * 64 * npad + kb * 64;
        GemmDesc p{};  // panel <- panel * inv(L_kk)^T
        p.A = panel; p.lda = npad; p.sA = sA;
        p.B = Linv + (long)kb * 64 * (npad + 1); p.ldb = npad; p.sB = sA;
        p.C = panel; p.ldc = npad; p.sC = sA;
        p.M = rem * 64; p.N = 64; p.K = 64; p.alpha = 1.0; p.beta = 0.0; p.tile_mode = 0; p.k_mode = 0;
        launch_gemm(st, p, false, true, batch);
        GemmDesc u{};  // trailing (lower tiles) -= panel * panel^T
        u.A = panel; u.lda = npad; u.sA = sA;
        u.B = panel; u.ldb = npad; u.sB = sA;
        u.C = A + (long)(kb + 1) * 64 * npad + (kb + 1) * 64; u.ldc = npad; u.sC = sA;
        u.M = rem * 64; u.N = rem * 64; u.K = 64; u.alpha = -1.0; u.beta = 1.0; u.tile_mode = 1; u.k_mode = 0;
        u.potf2_X = Linv; u.potf2_sX = sA; u.potf2_info = info; u.potf2_kb = kb + 1;
        launch_gemm(st, u, false, true, batch);
    }
}

// ------------------------------------------------------------------ triangular inverse
// Recursive doubling: with the inverses of the 64x64 diagonal blocks in place (launch_potrf wrote them there), level h merges
// pairs of inverted h x h diagonal blocks A, B of a 2h block [[A^-1, 0], [X, B^-1]] with X = -B^-1 (C A^-1),
// C = L[lower-left].  Two batched GEMMs per level over all (matrix, block) pairs: log2(npad/64) levels instead of
// one GEMM pair per block row.  Trailing partial blocks (npad/64 not a power of two) are clipped per sub-problem.
void launch_trtri(hipStream_t st, const double* L, int npad, int batch, double* Linv, double* T, long tstride) {
    const long sA = (long)npad * npad;
    for (int h = 64; h < npad; h *= 2) {
        const int nsub = (npad - h + 2 * h - 1) / (2 * h);   // 2h blocks whose lower half is not empty
        GemmDesc a{};  // T_q = C_q * A_q^-1      (A_q^-1 lower triangular: k >= j0)
        a.A = L + (long)h * npad; a.lda = npad; a.sA = sA; a.ssA = (long)2 * h * (npad + 1);
        a.B = Linv; a.ldb = npad; a.sB = sA; a.ssB = (long)2 * h * (npad + 1);
        a.C = T; a.ldc = h; a.sC = tstride; a.ssC = (long)h * h;
        a.M = h; a.N = h; a.K = h; a.alpha = 1.0; a.beta = 0.0; a.tile_mode = 0; a.k_mode = 2;
        a.nsub = nsub; a.sub_rows0 = npad - h; a.sub_rows_step = 2 * h;
        launch_gemm(st, a, false, false, batch);
        GemmDesc c{};  // X_q = -B_q^-1 * T_q     (B_q^-1 lower triangular: k < i0 + 64)
        c.A = Linv + (long)h * (npad + 1); c.lda = npad; c.sA = sA; c.ssA = (long)2 * h * (npad + 1);
        c.B = T; c.ldb = h; c.sB = tstride; c.ssB = (long)h * h;
        c.C = Linv + (long)h * npad; c.ldc = npad; c.sC = sA; c.ssC = (long)2 * h * (npad + 1);
        c.M = h; c.N = h; c.K = h; c.alpha = -1.0; c.beta = 0.0; c.tile_mode = 0; c.k_mode = 3;
        c.nsub = nsub; c.sub_rows0 = npad - h; c.sub_rows_step = 2 * h;
        launch_gemm(st, c, false, false, batch);
    }
}

// ------------------------------------------------------------------ FITC helpers
// 64 columns per workgroup, the rows dealt over its four waves (round 2: one thread per column walking all M rows twice,
// 200 workgroups on the chip: 134 us at M = 200, N = 5000)
// G = sqrt(nu) / sn per column, V <- Vb = V / G (smgpr.py:31-33) and, in the same pass over V, the right-hand side
// r = Vb (y / G) (smgpr.py:40): the scaled element is in a register, its product with y_n / G_n is summed over the block's 64
// columns by the wave and left as a partial per (column block, row); k_fitc_rhs_sum adds the blocks in their order.  (As a
// kernel of its own the sum was a second pass over V: 40 us of a 1.4 ms FITC objective at M = 200, N = 5000.)
__global__ __launch_bounds__(256) void k_fitc_scale_rhs(double* __restrict__ V, int mpad, int npad, const double* __restrict__ var,
                                                        const double* __restrict__ noise, double* __restrict__ G,
                                                        const double* __restrict__ y, double* __restrict__ rpart) {
    __shared__ double red[4][64];
    const int b = blockIdx.y, lane = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + lane;
    const bool in = n < npad;
    double* Vb = V + (long)b * mpad * npad;
    double ss = 0.0;
    if (in)
        for (int m = q; m < mpad; m += 4) {
            const double v = Vb[(long)m * npad + n];
            ss = fma(v, v, ss);
        }
    red[q][lane] = ss;
    __syncthreads();
    ss = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
    const double g = sqrt(1.0 + (var[b] - ss) / noise[b]);   // smgpr.py:31-32
    if (q == 0 && in) G[(long)b * npad + n] = g;
    const double ig = 1.0 / g;
    const double yg = in ? y[(long)b * npad + n] * ig : 0.0;
    double* rp = rpart + ((long)b * gridDim.x + blockIdx.x) * mpad;
    for (int m = q; m < mpad; m += 4) {
        double p = 0.0;
        if (in) {
            const double vb = Vb[(long)m * npad + n] * ig;   // smgpr.py:33
            Vb[(long)m * npad + n] = vb;
            p = vb * yg;
        }
        for (int off = 32; off > 0; off >>= 1) p += __shfl_down(p, off);
        if (lane == 0) rp[m] = p;
    }
}
__global__ __launch_bounds__(256) void k_fitc_rhs_sum(const double* __restrict__ rpart, int nblk, int mpad, double* __restrict__ r) {
    const int b = blockIdx.y, m = blockIdx.x * 256 + threadIdx.x;
    if (m >= mpad) return;
    const double* rp = rpart + (long)b * nblk * mpad + m;
    double s = 0.0;
    for (int k0 = 0; k0 < nblk; k0 += 8) {   // (eight requests in flight; fixed order)
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = rp[(long)min(k0 + u, nblk - 1) * mpad];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += (k0 + u < nblk) ? v[u] : 0.0;
    }
    r[(long)b * mpad + m] = s;
}
void launch_fitc_scale_rhs(hipStream_t st, double* V, int mpad, int npad, int batch, const double* var, const double* noise, double* G,
                           const double* y, double* scratch /* [batch][ceil(npad / 64)][mpad] */, double* r) {
    const int nblk = (npad + 63) / 64;
    hipLaunchKernelGGL(k_fitc_scale_rhs, dim3(nblk, batch), dim3(256), 0, st, V, mpad, npad, var, noise, G, y, scratch);
    hipLaunchKernelGGL(k_fitc_rhs_sum, dim3((mpad + 255) / 256, batch), dim3(256), 0, st, (const double*)scratch, nblk, mpad, r);
}

__global__ void k_add_diag(double* __restrict__ A, int npad, const double* __restrict__ d) {
    const int b = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < npad) A[((long)b * npad + i) * npad + i] += d[b];
}
void launch_add_diag(hipStream_t st, double* A, int npad, int batch, const double* d) {
    hipLaunchKernelGGL(k_add_diag, dim3((npad + 255) / 256, batch), dim3(256), 0, st, A, npad, d);
}

// ------------------------------------------------------------------ GP training sums
// out[b] = sum_i log L_ii;  (y != nullptr) out[batch + b] = y_b . beta_b, the data-fit term of the same objective
__global__ __launch_bounds__(256) void k_logdet(const double* __restrict__ L, int npad, int n, double* __restrict__ out,
                                                const double* __restrict__ y, const double* __restrict__ beta) {
    __shared__ double red[2][4];
    const int b = blockIdx.x, t = threadIdx.x;
    double s = 0.0, q = 0.0;
    for (int i = t; i < n; i += 256) s += log(L[((long)b * npad + i) * npad + i]);
    if (y)
        for (int i = t; i < n; i += 256) q = fma(y[(long)b * npad + i], beta[(long)b * npad + i], q);
    for (int off = 32; off > 0; off >>= 1) {
        s += __shfl_down(s, off);
        q += __shfl_down(q, off);
    }
    if ((t & 63) == 0) {
        red[0][t >> 6] = s;
        red[1][t >> 6] = q;
    }
    __syncthreads();
    if (t == 0) {
        out[b] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        if (y) out[gridDim.x + b] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    }
}
void launch_logdet(hipStream_t st, const double* L, int npad, int n, int batch, double* out, const double* y, const double* beta) {
    hipLaunchKernelGGL(k_logdet, dim3(batch), dim3(256), 0, st, L, npad, n, out, y, beta);
}

// one workgroup per (output, 64-row tile, 64-column tile): thread t handles column j0 + (t & 63) against the 16 rows
// i0 + 16 (t >> 6) ..; all per-dimension arrays are register arrays of the compile-time width DT (the first version
// indexed [32]-arrays with a run-time D: they lived in scratch memory and one launch took 114 us at N = 225).
// partial[b][tile_i * ntiles + tile_j][NLML_MAXD + 2]; a second launch reduces the tiles in fixed order.
// Round 5: the summand w_ij dK_ij is symmetric in (i, j) -- only the tiles on and below the diagonal are computed, those
// below it count twice (an exact factor) --, a thread's sixteen iK entries are requested together (one by one each cost a
// memory round trip: sixteen in a row per workgroup), and the reduction keeps eight loads in flight (it walked its 256
// partials one dependent load after the other).  140 -> 45 us for the two launches at C2.
constexpr int NLML_MAXD = 32;
template <int DT>
__global__ __launch_bounds__(256) void k_nlml_grad_partial(const double* __restrict__ Pt, int npad, int n, int D,
                                                           const double* __restrict__ ls, const double* __restrict__ var,
                                                           const double* __restrict__ iK, const double* __restrict__ beta,
                                                           double* __restrict__ partial) {
    __shared__ double xi[DT][64];
    __shared__ double bi[64];
    __shared__ double red[4][DT + 2];
    const int b = blockIdx.z, ti = blockIdx.y, tj = blockIdx.x, t = threadIdx.x;
    if (tj > ti) return;   // (the mirror image of tile (tj, ti): counted there)
    const int i0 = ti * 64, j = tj * 64 + (t & 63), g = t >> 6;
    for (int e = t; e < DT * 64; e += 256) xi[e >> 6][e & 63] = ((e >> 6) < D) ? Pt[(long)(e >> 6) * npad + i0 + (e & 63)] : 0.0;
    if (t < 64) bi[t] = beta[(long)b * npad + i0 + t];
    __syncthreads();
    const double v = var[b];
    double il[DT], xj[DT], acc[DT + 2];
#pragma unroll
    for (int d = 0; d < DT; ++d) {
        il[d] = (d < D) ? 1.0 / ls[b * D + d] : 0.0;
        xj[d] = (d < D) ? Pt[(long)d * npad + j] : 0.0;
        acc[d] = 0.0;
    }
    acc[DT] = acc[DT + 1] = 0.0;
    const double* iKb = iK + (long)b * npad * npad;
    const double bj = beta[(long)b * npad + j];
    if (j < n) {
        double ikv[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) ikv[q] = iKb[(long)(i0 + 16 * g + q) * npad + j];   // (padding rows exist: zeros)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int ii = 16 * g + q, i = i0 + ii;
            if (i >= n) break;
            double r2 = 0.0;
            double sq[DT];
#pragma unroll
            for (int d = 0; d < DT; ++d) {
                const double df = (xi[d][ii] - xj[d]) * il[d];
                sq[d] = df * df;
                r2 += sq[d];
            }
            const double k = v * exp(-0.5 * r2);
            const double w = ikv[q] - bi[ii] * bj;
            const double wk = w * k;
#pragma unroll
            for (int d = 0; d < DT; ++d) acc[d] = fma(wk, sq[d] * il[d], acc[d]);   // (x_i-x_j)^2 / l^3
            acc[DT] += wk;
            if (i == j) acc[DT + 1] += w;
        }
    }
#pragma unroll
    for (int d = 0; d < DT + 2; ++d) {
        double s2 = acc[d];
        for (int off = 32; off > 0; off >>= 1) s2 += __shfl_down(s2, off);
        if ((t & 63) == 0) red[g][d] = s2;
    }
    __syncthreads();
    if (t < D + 2) {
        const int src = t < D ? t : DT + (t - D);
        const double sum = (red[0][src] + red[1][src]) + (red[2][src] + red[3][src]);
        partial[((long)b * gridDim.y * gridDim.x + (long)ti * gridDim.x + tj) * (NLML_MAXD + 2) + t] = (ti != tj) ? 2.0 * sum : sum;
    }
}
// one workgroup per output, four waves: lane = gradient entry, wave w takes the tiles on and below the diagonal whose running
// number is w mod 4 (eight loads in flight each); the four sums are added in their order
__global__ __launch_bounds__(256) void k_nlml_grad_reduce(const double* __restrict__ partial, int nt, int D, const double* __restrict__ var,
                                                          double* __restrict__ grad) {
    __shared__ double red[4][64];
    const int b = blockIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int ntri = nt * (nt + 1) / 2;
    double s8[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0}, s9 = 0.0;
    if (lane < D + 2) {
        // running number q of tile (ti, tj), tj <= ti: ti (ti + 1) / 2 + tj
        auto tile_of = [&](int q) {
            int ti = 0;
            while ((ti + 1) * (ti + 2) / 2 <= q) ++ti;
            return (long)ti * nt + (q - ti * (ti + 1) / 2);
        };
        const double* base = partial + (long)b * nt * nt * (NLML_MAXD + 2) + lane;
        for (int q = w; q < ntri; q += 32) {   // (the last, partial batch too: clamped requests, masked sums -- one at a time its
            double v[8];                       // up to seven tiles were seven round trips, 14 of the kernel's 20 us)
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = base[tile_of(min(q + 4 * u, ntri - 1)) * (NLML_MAXD + 2)];
#pragma unroll
            for (int u = 0; u < 8; ++u) s8[u] += (q + 4 * u < ntri) ? v[u] : 0.0;
        }
    }
    red[w][lane] = (((s8[0] + s8[1]) + (s8[2] + s8[3])) + ((s8[4] + s8[5]) + (s8[6] + s8[7]))) + s9;
    __syncthreads();
    if (w == 0 && lane < D + 2) {
        double s = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
        if (lane == D) s /= var[b];
        grad[b * (D + 2) + lane] = 0.5 * s;
    }
}
// partial: [batch][(npad / 64)^2][NLML_MAXD + 2] doubles
void launch_nlml_grad(hipStream_t st, const double* Pt, int npad, int n, int D, const double* ls, const double* var,
                      const double* iK, const double* beta, int batch, double* partial, double* grad) {
    const int ntiles = npad / 64;
    const dim3 grid(ntiles, ntiles, batch);
#define NG(DT_) hipLaunchKernelGGL(k_nlml_grad_partial<DT_>, grid, dim3(256), 0, st, Pt, npad, n, D, ls, var, iK, beta, partial)
    if (D <= 4) NG(4);
    else if (D <= 8) NG(8);
    else if (D <= 12) NG(12);
    else if (D <= 16) NG(16);
    else if (D <= 24) NG(24);
    else NG(32);
#undef NG
    hipLaunchKernelGGL(k_nlml_grad_reduce, dim3(batch), dim3(256), 0, st, partial, ntiles, D, var, grad);
}

// ------------------------------------------------------------------ mat-vec, padding
// A is LOWER TRIANGULAR (every caller passes a triangular inverse from launch_trtri, zero above the diagonal): the zero
// part is not read.  The sums keep round 2's order term by term (the skipped terms were exact zeros), so the results are
// bitwise what they were.
__global__ __launch_bounds__(256) void k_matvec(const double* __restrict__ A, int npad, const double* __restrict__ x,
                                                double* __restrict__ y, int trans) {
    const int b = blockIdx.y;
    const double* Ab = A + (long)b * npad * npad;
    const double* xb = x + (long)b * npad;
    if (trans) {  // y[j] = sum_{i >= j} A[i][j] x[i]; 64 columns per workgroup, the four waves split the rows
        __shared__ double red[4][64];
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
        const int j = blockIdx.x * 64 + lane;
        double s0 = 0.0, s1 = 0.0;
        int i = blockIdx.x * 64 + w;   // rows above the workgroup's first column are zero (a multiple of 8: s0 / s1 keep their rows)
        // eight loads in flight per wave (two, as in round 2, left this latency-bound: 45 us for ten 1024 x 1024 matrices)
        for (; i + 28 < npad; i += 32) {
            double av[8], xv[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                av[q] = Ab[(long)(i + 4 * q) * npad + j];
                xv[q] = xb[i + 4 * q];
            }
#pragma unroll
            for (int q = 0; q < 8; q += 2) {
                s0 = fma(av[q], xv[q], s0);
                s1 = fma(av[q + 1], xv[q + 1], s1);
            }
        }
        for (; i + 4 < npad; i += 8) {
            s0 = fma(Ab[(long)i * npad + j], xb[i], s0);
            s1 = fma(Ab[(long)(i + 4) * npad + j], xb[i + 4], s1);
        }
        for (; i < npad; i += 4) s0 = fma(Ab[(long)i * npad + j], xb[i], s0);
        red[w][lane] = s0 + s1;
        __syncthreads();
        if (w == 0) y[(long)b * npad + j] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
    } else {  // y[i] = sum_{j <= i} A[i][j] x[j]; one wave per row
        const int lane = threadIdx.x & 63;
        const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
        if (i >= npad) return;
        double s = 0.0;
        const int jend = (i | 63) + 1;   // the row's last non-zero entry is column i
        for (int j = lane; j < jend; j += 64) s = fma(Ab[(long)i * npad + j], xb[j], s);
        for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
        if (lane == 0) y[(long)b * npad + i] = s;
    }
}

void launch_matvec(hipStream_t st, const double* A, int npad, int batch, const double* x, double* y, bool trans) {
    dim3 grid(trans ? npad / 64 : (npad + 3) / 4, batch);
    hipLaunchKernelGGL(k_matvec, grid, dim3(256), 0, st, A, npad, x, y, trans ? 1 : 0);
}

// zero rows and columns n .. npad - 1: one 64-thread workgroup per row (the padding is at most 63 wide; round 2 launched a
// thread per ELEMENT of the matrix for it: 19 us at N = 1000)
__global__ __launch_bounds__(64) void k_clear_padding(double* __restrict__ A, int npad, int n) {
    const int b = blockIdx.y, i = blockIdx.x, t = threadIdx.x;
    double* row = A + ((long)b * npad + i) * npad;
    if (i >= n) {
        for (int j = t; j < npad; j += 64) row[j] = 0.0;
    } else if (n + t < npad) {
        row[n + t] = 0.0;
    }
}

void launch_clear_padding(hipStream_t st, double* A, int npad, int n, int batch) {
    if (n == npad) return;
    hipLaunchKernelGGL(k_clear_padding, dim3(npad, batch), dim3(64), 0, st, A, npad, n);
}

}  // namespace pilco
