// Does "write-through store + flag, then FIRST-TOUCH ordinary loads" move data between workgroups of ONE resident kernel
// on gfx950 (8 XCDs, an L2 each, no hardware coherence between them) WITHOUT the L2 write-back / invalidate fences that
// tools/ubench_sync.hip shows to cost ~10 us?  The persistent-rollout design rests on it:
//   producer: data with sc1 (write-through) stores; s_waitcnt vmcnt(0); workgroup barrier; flag (sc1 store)
//   consumer: polls the flag with sc1 loads, then reads the data with ORDINARY loads from addresses nobody has loaded
//             before in this launch (every iteration marches to fresh memory), so no L1 / L2 can hold a stale copy.
// T1 bulk:  workgroup b writes an 8 KB block; every workgroup reads the blocks of 5 other workgroups (other XCDs).
// T2 false sharing: 16 workgroups of different XCDs each write ONE 8-byte word of the same 128-byte line (many lines),
//     then everybody reads whole lines -- fails if an L2 fills the whole line when it takes a partial write.
// T3 = T2 read with sc1 loads (must always pass; the cost of bypassing the L2).
// Prints mismatch counts (0 = protocol holds in this run) and us per iteration.
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench_coherence.hip -o exp/ubench_coherence
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned long long u64;
constexpr int SPIN_MAX = 4000000;

__device__ __forceinline__ void st_wt(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double ld_byp(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ u64 ld_flag(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// publish: every thread's earlier stores complete, then one flag store
__device__ __forceinline__ void publish(u64* flag, u64 epoch) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // s_waitcnt vmcnt(0): this thread's write-through stores are done
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(flag, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// flag-array barrier without cache maintenance: thread t watches workgroup t
__device__ __forceinline__ bool wait_all(const u64* flags, int nb, u64 epoch) {
    __shared__ int bad;
    if (threadIdx.x == 0) bad = 0;
    __syncthreads();
    for (int b = threadIdx.x; b < nb; b += blockDim.x) {
        int it = 0;
        while (ld_flag(flags + 8 * b) < epoch)
            if (++it > SPIN_MAX) { bad = 1; break; }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    __syncthreads();
    return bad == 0;
}

// mode 1 / 2 / 3 as in the header; words_per_wg: T1 block size in doubles
__global__ __launch_bounds__(256) void k_probe(double* buf, size_t stride_iter, u64* flags, int iters, int mode, int words_per_wg,
                                               u64* out) {
    const int nb = gridDim.x, b = blockIdx.x, t = threadIdx.x;
    u64 mism = 0;
    bool ok = true;
    const u64 t0 = wall_clock64();
    for (int i = 0; i < iters && ok; ++i) {
        double* base = buf + (size_t)i * stride_iter;
        const u64 epoch = (u64)(i + 1);
        if (mode == 1) {
            double* mine = base + (size_t)b * words_per_wg;
            for (int e = t; e < words_per_wg; e += blockDim.x) st_wt(mine + e, (double)(i * 131 + b) + 1e-4 * e);
            publish(flags + 8 * b, epoch);
            ok = wait_all(flags, nb, epoch);
            for (int k = 1; k <= 5; ++k) {
                const int src = (b + k * 37 + k) % nb;   // other workgroups, mostly other XCDs
                const double* theirs = base + (size_t)src * words_per_wg;
                for (int e = t; e < words_per_wg; e += blockDim.x)
                    if (theirs[e] != (double)(i * 131 + src) + 1e-4 * e) ++mism;
            }
        } else {
            // nlines lines of 16 words; word j of line l is written by workgroup (l * 16 + j) % nb ... every workgroup
            // writes nlines * 16 / nb words, one per line it takes part in
            const int nlines = words_per_wg;   // reuse the parameter
            for (int l = t; l < nlines; l += blockDim.x) {
                // workgroup b owns word (b + l) % 16 of line l when ((b + l) / 16) % (nb / 16) == l % (nb / 16)
                const int j = (b + l) & 15;
                if (((b + l) >> 4) % (nb >> 4) == l % (nb >> 4)) st_wt(base + (size_t)l * 16 + j, (double)(i * 7 + l) + 0.03125 * j);
            }
            publish(flags + 8 * b, epoch);
            ok = wait_all(flags, nb, epoch);
            for (int l = t; l < nlines; l += blockDim.x) {
                const double* line = base + (size_t)l * 16;
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const double v = (mode == 3) ? ld_byp(line + j) : line[j];
                    if (v != (double)(i * 7 + l) + 0.03125 * j) ++mism;
                }
            }
        }
        __syncthreads();
    }
    const u64 t1 = wall_clock64();
    // block sum of mismatches
    __shared__ u64 red[256];
    red[t] = mism;
    __syncthreads();
    if (t == 0) {
        u64 s = 0;
        for (int k = 0; k < (int)blockDim.x; ++k) s += red[k];
        out[3 * b] = t1 - t0;
        out[3 * b + 1] = ok ? 0 : 1;
        out[3 * b + 2] = s;
    }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 400;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int nb = prop.multiProcessorCount;
    printf("device: %d CUs; %d iterations per test, fresh memory every iteration\n", nb, iters);
    u64 *flags, *out;
    CK(hipMalloc(&flags, 8 * 4096 * sizeof(u64)));
    CK(hipMalloc(&out, 3 * 4096 * sizeof(u64)));
    std::vector<u64> h(3 * 4096);
    struct T { int mode; int words; const char* name; };
    const T tests[] = {{1, 1024, "T1 bulk 8 KB per workgroup, ordinary loads"},
                       {1, 5632, "T1 bulk 44 KB per workgroup (operand-sized), ordinary loads"},
                       {2, 4096, "T2 false sharing (16 writers per line, 4096 lines), ordinary loads"},
                       {3, 4096, "T3 false sharing, sc1 (L2-bypassing) loads"},
                       {2, 528, "T2 partial-sum sized (528 lines = 66 KB), ordinary loads"},
                       {3, 528, "T3 partial-sum sized, sc1 loads"}};
    for (const T& tc : tests) {
        const size_t stride = (tc.mode == 1) ? (size_t)nb * tc.words : (size_t)tc.words * 16;
        double* buf;
        CK(hipMalloc(&buf, stride * iters * sizeof(double)));
        CK(hipMemset(buf, 0xff, stride * iters * sizeof(double)));   // NaN pattern: a stale read can never look right
        CK(hipMemset(flags, 0, 8 * 4096 * sizeof(u64)));
        CK(hipDeviceSynchronize());
        hipLaunchKernelGGL(k_probe, dim3(nb), dim3(256), 0, 0, buf, stride, flags, iters, tc.mode, tc.words, out);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h.data(), out, sizeof(u64) * 3 * nb, hipMemcpyDeviceToHost));
        u64 mx = 0, bad = 0, mism = 0;
        for (int b = 0; b < nb; ++b) { mx = h[3 * b] > mx ? h[3 * b] : mx; bad += h[3 * b + 1]; mism += h[3 * b + 2]; }
        printf("%-72s mismatches %llu%s   %.2f us per iteration\n", tc.name, mism, bad ? "  TIMED OUT" : "", (double)mx * 0.01 / iters);
        CK(hipFree(buf));
    }
    return 0;
}
