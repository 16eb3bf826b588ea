// fp64 micro-benchmarks for gfx950: DP FMA, exp, f64 MFMA rates and their overlap.
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench.hip -o exp/ubench
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
#define ITERS 2048

__global__ void k_fma(double* out, double a, double b) {
    double x[8];
    for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 1e-3 + i;
    for (int it = 0; it < ITERS; ++it)
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = fma(x[i], a, b);
    double s = 0; for (int i = 0; i < 8; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_exp(double* out, double a, double b) {
    double x[8];
    for (int i = 0; i < 8; ++i) x[i] = -(threadIdx.x * 1e-2 + i);
    for (int it = 0; it < ITERS / 8; ++it)
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = fma(exp(x[i]), a, b);   // a=-3, b=-1 keeps x in [-4,-1]
    double s = 0; for (int i = 0; i < 8; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_mfma(double* out, double a, double b) {
    d4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = d4{0, 0, 0, 0};
    double av = a + threadIdx.x * 1e-6, bv = b;
    for (int it = 0; it < ITERS / 4; ++it)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[i], 0, 0, 0);
    double s = 0; for (int i = 0; i < 4; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// exp on MFMA results: 3 MFMAs (K=12) per 4 exps per lane, like the pair kernel
__global__ void k_mix(double* out, double a, double b) {
    double s[4] = {0, 0, 0, 0};
    double av = a * 1e-3 * (threadIdx.x & 15), bv = b * 1e-3;
    for (int it = 0; it < ITERS / 8; ++it) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            d4 e = {-1.0, -2.0, -3.0, -0.5};
            e = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, e, 0, 0, 0);
            e = __builtin_amdgcn_mfma_f64_16x16x4f64(av + t, bv, e, 0, 0, 0);
            e = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv + it * 1e-9, e, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) s[r] = fma(bv, exp(e[r]), s[r]);
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}

#define FEXP_C 92.332482616893656758
#define FEXP_MAGIC 6755399441055744.0
#define FEXP_LN2_64 0.010830424696249145
__device__ __forceinline__ double fexp(double x, const double* tab) {
    double y; const double lo = -700.0;
    asm("v_max_f64 %0, %1, %2" : "=v"(y) : "v"(x), "s"(lo));
    x = y;
    const double t = fma(x, FEXP_C, FEXP_MAGIC);
    const double r = fma(t - FEXP_MAGIC, -FEXP_LN2_64, x);
    double q = fma(r, 1.0 / 120.0, 1.0 / 24.0);
    q = fma(r, q, 1.0 / 6.0); q = fma(r, q, 0.5); q = fma(r, q, 1.0);
    const double pm1 = r * q;
    const double tv = tab[__double2loint(t) & 63];
    const double res = fma(tv, pm1, tv);
    const int l = __double2loint(t) & ~63; int hi;
    asm("v_lshl_add_u32 %0, %1, 14, %2" : "=v"(hi) : "v"(l), "v"(__double2hiint(res)));
    return __hiloint2double(hi, __double2loint(res));
}
__global__ void k_fexp(double* out, double a, double b) {
    __shared__ double tab[64];
    if (threadIdx.x < 64) tab[threadIdx.x] = exp2(threadIdx.x / 64.0);
    __syncthreads();
    double x[8];
    for (int i = 0; i < 8; ++i) x[i] = -(threadIdx.x * 1e-2 + i);
    for (int it = 0; it < ITERS / 8; ++it)
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = fma(fexp(x[i], tab), a, b);
    double s = 0; for (int i = 0; i < 8; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_mixf(double* out, double a, double b) {
    __shared__ double tab[64];
    if (threadIdx.x < 64) tab[threadIdx.x] = exp2(threadIdx.x / 64.0);
    __syncthreads();
    double s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    double av = a * 1e-3 * (threadIdx.x & 15), bv = b * 1e-3;
    for (int it = 0; it < ITERS / 8; ++it) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            d4 e0 = {-1.0, -2.0, -3.0, -0.5}, e1 = {-1.5, -2.5, -3.5, -0.25};
            e0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, e0, 0, 0, 0);
            e1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av + 1, bv, e1, 0, 0, 0);
            e0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av + t, bv, e0, 0, 0, 0);
            e1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av + t + 1, bv, e1, 0, 0, 0);
            e0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv + it * 1e-9, e0, 0, 0, 0);
            e1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av + 2, bv + it * 1e-9, e1, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) { s[r] = fma(bv, fexp(e0[r], tab), s[r]); s[4 + r] = fma(bv, fexp(e1[r], tab), s[4 + r]); }
        }
    }
    double tot = 0; for (int i = 0; i < 8; ++i) tot += s[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = tot;
}
template <typename F> float timeit(F f) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    f(); hipDeviceSynchronize();
    hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
    double* out; hipMalloc(&out, sizeof(double) * 256 * 16 * 256);
    for (int wpb : {4, 8, 16}) {   // waves per SIMD = blocks*4waves/4simd: grid = 256 CUs * k
        for (int bpc : {1, 2, 4}) {
            int grid = 256 * bpc, block = 64 * wpb; if (block > 1024) continue;
            double thr = (double)grid * block;
            float t1 = timeit([&] { hipLaunchKernelGGL(k_fma, dim3(grid), dim3(block), 0, 0, out, 0.999, 0.001); });
            float t2 = timeit([&] { hipLaunchKernelGGL(k_exp, dim3(grid), dim3(block), 0, 0, out, -3.0, -1.0); });
            float t3 = timeit([&] { hipLaunchKernelGGL(k_mfma, dim3(grid), dim3(block), 0, 0, out, 0.5, 0.25); });
            float t4 = timeit([&] { hipLaunchKernelGGL(k_mix, dim3(grid), dim3(block), 0, 0, out, 0.5, 0.25); });
            float t5 = timeit([&] { hipLaunchKernelGGL(k_fexp, dim3(grid), dim3(block), 0, 0, out, -3.0, -1.0); });
            float t6 = timeit([&] { hipLaunchKernelGGL(k_mixf, dim3(grid), dim3(block), 0, 0, out, 0.5, 0.25); });
            printf("  fexp %.1f Gexp/s (%.1f cyc) | mix-fexp %.1f Gexp/s (%.1f cyc/wave-exp)\n", thr * ITERS / (t5 * 1e-3) / 1e9, 2.4e9 * 1024 * 64 / (thr * ITERS / (t5 * 1e-3)),
                   thr * (ITERS / 8) * 16 / (t6 * 1e-3) / 1e9, 2.4e9 * 1024 * 64 / (thr * (ITERS / 8) * 16 / (t6 * 1e-3)));
            double waves_per_simd = (double)wpb * bpc / 4.0;
            printf("waves/SIMD %.1f | fma %.2f TFLOP/s | exp %.1f Gexp/s (%.1f cyc/wave-exp@2.4GHz) | mfma %.2f TFLOP/s | mix %.1f Gexp/s (%.1f cyc/wave-exp)\n",
                   waves_per_simd, thr * ITERS * 8 * 2 / (t1 * 1e-3) / 1e12, thr * ITERS / (t2 * 1e-3) / 1e9,
                   2.4e9 * 1024 * 64 / (thr * ITERS / (t2 * 1e-3)), (thr / 64) * ITERS * 2048.0 / (t3 * 1e-3) / 1e12,
                   thr * (ITERS / 8) * 16 / (t4 * 1e-3) / 1e9, 2.4e9 * 1024 * 64 / (thr * (ITERS / 8) * 16 / (t4 * 1e-3)));
        }
    }
    return 0;
}
