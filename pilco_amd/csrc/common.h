// Internal declarations shared by the translation units of libpilco_hip.so.
// gfx950 only; no portability layer on purpose.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "pilco_hip_dev.h"   // (includes pilco_hip.h: the boundary) + the developer / measurement entry points

namespace pilco {

constexpr int NB = 64;  // factorisation block size == padding unit of every device matrix

inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

struct DevBuf {
    double* p = nullptr;
    size_t cap = 0;  // in doubles
    bool borrowed = false;  // a view of another context's buffer (rollout lanes share their parent's model): never freed here
    hipError_t ensure(size_t count) {
        if (count <= cap && p) return hipSuccess;
        if (p && !borrowed) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        borrowed = false;
        hipError_t e = hipMalloc(&p, count * sizeof(double));
        if (e == hipSuccess) cap = count;
        return e;
    }
    void release() {
        if (p && !borrowed) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        borrowed = false;
    }
    void borrow(const DevBuf& o) {
        release();
        p = o.p;
        cap = o.cap;
        borrowed = o.p != nullptr;
    }
};

// The kernel-argument segment of these kernels is 0.3-1.6 KB of by-value descriptors (MMModel, MMWork, GlueArgs, ...).  The
// compiler fetches a field where it is first needed -- an s_load, an s_waitcnt, the arithmetic that leads to the next
// field, the next s_load: the fused head's prologue is SIX such round trips in a row, each a miss of the scalar cache (the
// segment of a graph node lives in device memory and was evicted from the XCD's L2 by the 60 MB the previous pair kernel
// streamed).  kernarg_warm requests every 64-byte line of the segment at once -- one round trip -- so that the compiler's own
// loads hit the scalar cache.  (KERNARG_WARM=0: off, for A/B runs.)
#ifndef KERNARG_WARM
#define KERNARG_WARM 1
#endif
template <int BYTES>
__device__ __forceinline__ void kernarg_warm() {
#if KERNARG_WARM
    static_assert(BYTES <= 2048, "kernarg_warm covers segments of at most 2 KB");
    // ONE asm statement -- requests and wait: nothing of the compiler's may come between a request and the wait (the
    // destination register is dead for the compiler as soon as the statement ends, while a scalar load that is still in
    // flight would write it later; scalar loads return out of order).  The assembler's .if keeps the lines the segment has.
    int sink;
    asm volatile("s_load_dword %0, %1, 0\n\t"
                 ".if %2 > 64\n\ts_load_dword %0, %1, 64\n\t.endif\n\t"
                 ".if %2 > 128\n\ts_load_dword %0, %1, 128\n\t.endif\n\t"
                 ".if %2 > 192\n\ts_load_dword %0, %1, 192\n\t.endif\n\t"
                 ".if %2 > 256\n\ts_load_dword %0, %1, 256\n\t.endif\n\t"
                 ".if %2 > 320\n\ts_load_dword %0, %1, 320\n\t.endif\n\t"
                 ".if %2 > 384\n\ts_load_dword %0, %1, 384\n\t.endif\n\t"
                 ".if %2 > 448\n\ts_load_dword %0, %1, 448\n\t.endif\n\t"
                 ".if %2 > 512\n\ts_load_dword %0, %1, 512\n\t.endif\n\t"
                 ".if %2 > 576\n\ts_load_dword %0, %1, 576\n\t.endif\n\t"
                 ".if %2 > 640\n\ts_load_dword %0, %1, 640\n\t.endif\n\t"
                 ".if %2 > 704\n\ts_load_dword %0, %1, 704\n\t.endif\n\t"
                 ".if %2 > 768\n\ts_load_dword %0, %1, 768\n\t.endif\n\t"
                 ".if %2 > 832\n\ts_load_dword %0, %1, 832\n\t.endif\n\t"
                 ".if %2 > 896\n\ts_load_dword %0, %1, 896\n\t.endif\n\t"
                 ".if %2 > 960\n\ts_load_dword %0, %1, 960\n\t.endif\n\t"
                 ".if %2 > 1024\n\ts_load_dword %0, %1, 1024\n\t.endif\n\t"
                 ".if %2 > 1088\n\ts_load_dword %0, %1, 1088\n\t.endif\n\t"
                 ".if %2 > 1152\n\ts_load_dword %0, %1, 1152\n\t.endif\n\t"
                 ".if %2 > 1216\n\ts_load_dword %0, %1, 1216\n\t.endif\n\t"
                 ".if %2 > 1280\n\ts_load_dword %0, %1, 1280\n\t.endif\n\t"
                 ".if %2 > 1344\n\ts_load_dword %0, %1, 1344\n\t.endif\n\t"
                 ".if %2 > 1408\n\ts_load_dword %0, %1, 1408\n\t.endif\n\t"
                 ".if %2 > 1472\n\ts_load_dword %0, %1, 1472\n\t.endif\n\t"
                 ".if %2 > 1536\n\ts_load_dword %0, %1, 1536\n\t.endif\n\t"
                 ".if %2 > 1600\n\ts_load_dword %0, %1, 1600\n\t.endif\n\t"
                 ".if %2 > 1664\n\ts_load_dword %0, %1, 1664\n\t.endif\n\t"
                 ".if %2 > 1728\n\ts_load_dword %0, %1, 1728\n\t.endif\n\t"
                 ".if %2 > 1792\n\ts_load_dword %0, %1, 1792\n\t.endif\n\t"
                 ".if %2 > 1856\n\ts_load_dword %0, %1, 1856\n\t.endif\n\t"
                 ".if %2 > 1920\n\ts_load_dword %0, %1, 1920\n\t.endif\n\t"
                 ".if %2 > 1984\n\ts_load_dword %0, %1, 1984\n\t.endif\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&s"(sink)
                 : "s"(__builtin_amdgcn_kernarg_segment_ptr()), "n"(BYTES)
                 : "memory");
#endif
}

// ---------------------------------------------------------------- linalg.hip
struct GemmDesc {
    const double* A;
    const double* B;
    double* C;
    int M, N, K;        // all multiples of 64 / 64 / 16
    int lda, ldb, ldc;
    long sA, sB, sC;    // batch strides (doubles)
    double alpha, beta;
    const double* alpha_vec;  // optional per-batch multiplier of alpha (device), or nullptr
    int tile_mode;      // 0 all tiles, 1 only tiles with row-block >= col-block, 2 the same and every tile below the diagonal also stored transposed above it (C symmetric, beta = 0)
    int k_mode;         // 0 full K; 1: k >= max(i0,j0); 2: k >= j0; 3: k < i0+64 (A lower triangular); 4: j0 <= k < i0+64
    // optional second batch level: batch index z -> matrix z / nsub, sub-problem z % nsub (nsub = 0: off)
    int nsub;
    long ssA, ssB, ssC; // sub-problem strides (doubles)
    int sub_rows0, sub_rows_step;  // sub-problem q only has rows < sub_rows0 - q * sub_rows_step (row tiles past that are skipped)
    // Cholesky chain (launch_potrf): the workgroup of tile (0, 0) factors and inverts that tile afterwards -- diagonal block
    // potf2_kb; its inverse goes to the same diagonal block of potf2_X (the L^-1 buffer: [matrix][ldx][ldx], matrix stride
    // potf2_sX), first bad pivot to potf2_info[matrix]
    // split K (long-K products of the FITC path: K = N = 5000 against 16 tiles per matrix left every tile a chain of 316
    // chunks): K is cut in ksplit slices, each (matrix, slice) computes its partial product into split_ws
    // [ksplit][batch][M][N], and a second launch adds the slices up in their order (deterministic).  nsub must be 0.
    int ksplit;
    double* split_ws;
    double* potf2_X;
    long potf2_sX;
    int* potf2_info;
    int potf2_kb;
};
constexpr int FITC_KSPLIT = 16;   // K slices of the FITC path's M x M products over the N data points
// C = alpha * op(A) op(B) + beta * C, batched; op selected by ta/tb (0 = as stored, 1 = transposed)
void launch_gemm(hipStream_t st, const GemmDesc& g, bool ta, bool tb, int batch);

// Gram matrices: out[a][i][j] = var[a] exp(-0.5 sum_d ((P1[d][i]-P2[d][j])/ls[a][d])^2)
// P1t: [D][ld1], P2t: [D][ld2] (transposed points); out: [E][rows_pad][cols_pad].
// diag_mode 0: none; 1: += diag_add[a] on i==j<n1, padding diag = 1; 2: += jitter on diag, padding diag = 1
void launch_gram(hipStream_t st, const double* P1t, int ld1, int n1, const double* P2t, int ld2, int n2, int D,
                 const double* ls, const double* var, int E, double* out, int rows_pad, int cols_pad, int diag_mode,
                 const double* diag_add, double jitter, long sP1 = 0, long sP2 = 0);   // sP*: per-output strides of the point sets (0 = shared)

// Blocked Cholesky (lower) of batch matrices A[b] (npad x npad, ld = npad), in place; strictly-upper part zeroed.
// info[b] = 0 or 1-based index of the first non-positive pivot.  CONTRACT: with info[b] != 0 EVERYTHING this call and its
// consumers produce for matrix b is garbage (a bad pivot is not replaced: NaN / Inf run through L, the inverses of its
// diagonal blocks, every later panel and update, and from there through L^-1, iK and beta) -- the caller checks info before
// it uses a factor and leaves the slot's factor_valid false (factorize_exact, pilco_factorize_fitc, fitc_nlml_batch do;
// tests: test_a_failed_factorisation_leaves_no_usable_factor).
// The inverses of L's 64 x 64 diagonal blocks go straight into the diagonal blocks of Linv ([batch][npad][npad]), where
// launch_trtri builds on them (rounds 1-4: into a buffer of their own, copied over by a launch).  zero_linv: Linv is zeroed
// first -- needed when somebody reads Linv's tiles ABOVE the diagonal (plain GEMMs of the FITC path); the exact path's
// consumers (launch_trtri, the k_mode 1 product iK = Linv^T Linv, launch_matvec) never do, and skip 12 us at C2.
void launch_potrf(hipStream_t st, double* A, int npad, int batch, double* Linv, int* info, bool zero_linv);
// Linv = L^{-1} (lower), completing the diagonal blocks launch_potrf left there; T is scratch, batch matrices of tstride >= npad*npad/2 doubles.
void launch_trtri(hipStream_t st, const double* L, int npad, int batch, double* Linv, double* T, long tstride);
// y = op(A) x, A [batch][npad][npad], x,y [batch][npad]
void launch_matvec(hipStream_t st, const double* A, int npad, int batch, const double* x, double* y, bool trans);
// zero rows/cols >= n of batch square matrices
void launch_clear_padding(hipStream_t st, double* A, int npad, int n, int batch);
void launch_transpose_points(hipStream_t st, const double* X, int n, int D, double* Xt, int ld, int batch = 1, long sX = 0, long sXt = 0);
// FITC helpers (smgpr.py:30-43)
// G[b][n] = sqrt(1 + (var[b] - sum_m V[b][m][n]^2) / noise[b]);  V[b][m][n] /= G[b][n]
void launch_fitc_scale_rhs(hipStream_t st, double* V, int mpad, int npad, int batch, const double* var, const double* noise, double* G,
                           const double* y, double* scratch /* [batch][ceil(npad / 64)][mpad] */, double* r /* [batch][mpad] = Vb (y / G) */);
// A[b][i][i] += d[b]
void launch_add_diag(hipStream_t st, double* A, int npad, int batch, const double* d);
// r[b][m] = sum_n V[b][m][n] / G[b][n] * y[b][n]

// GP training: per output sum_i log L_ii, and the D+2 weighted sums
//   g[d] = 1/2 sum_ij W_ij K_ij (x_id - x_jd)^2 / l_d^3,  g[D] = 1/2 sum_ij W_ij K_ij / var,  g[D+1] = 1/2 tr W,
// with W = iK - beta beta^T and K the noise-free Gram matrix recomputed on the fly.
void launch_logdet(hipStream_t st, const double* L, int npad, int n, int batch, double* out, const double* y = nullptr,
                   const double* beta = nullptr);   // (y: out[batch + b] = y_b . beta_b as well)
void launch_nlml_grad(hipStream_t st, const double* Pt, int npad, int n, int D, const double* ls, const double* var,
                      const double* iK, const double* beta, int batch, double* partial, double* grad);

// ---------------------------------------------------------------- prep.hip / pair.hip / glue.hip / bwd.hip
struct MMWork;  // defined in moment.h

}  // namespace pilco
