// Cost of device-wide synchronisation inside ONE resident kernel on gfx950 (8 XCDs, one L2 each): what a persistent
// whole-rollout kernel pays per phase boundary instead of a kernel boundary.
//   bar_flat   : every workgroup adds to one agent-scope counter (release) and spins on it (acquire)
//   bar_hier   : workgroups of one XCD meet on their own counter first, one of them goes to the global counter
//   xchg_cnt   : the pair -> link exchange: every wave stores one partial (write-through), counter barrier, then EVERY
//                workgroup loads all partials (what the serial link of the next step needs)
//   xchg_tag   : the same with 16-byte {value, epoch} records polled directly (no counter: one round trip)
//   flag_pc    : producer -> consumer flags per "pair": 220 producer workgroups raise 55 counters, every wave waits for its one
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench_sync.hip -o exp/ubench_sync
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned long long u64;
typedef unsigned int u32;

__device__ __forceinline__ u64 ld_agent(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ u64 ld_acq(const u64* p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT); }

constexpr int SPIN_MAX = 2000000;   // bounded: a bug must not hang the GPU

// --- flat barrier: counter is monotonic, target = (iteration + 1) * nblocks
__device__ __forceinline__ bool bar_flat(u64* cnt, u64 target, int sleep) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(cnt, 1ULL, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        int it = 0;
        while (ld_acq(cnt) < target) {
            if (++it > SPIN_MAX) { ok = false; break; }
            if (sleep) __builtin_amdgcn_s_sleep(1);
        }
    }
    __syncthreads();
    return ok;
}
// relaxed polling + one acquire fence at the end (the acquire's cache invalidate is paid once, not per poll)
__device__ __forceinline__ bool bar_flat_rlx(u64* cnt, u64 target, int sleep) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(cnt, 1ULL, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        int it = 0;
        while (ld_agent(cnt) < target) {
            if (++it > SPIN_MAX) { ok = false; break; }
            if (sleep) __builtin_amdgcn_s_sleep(1);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    return ok;
}
__device__ __forceinline__ bool bar_hier(u64* xcnt /* [8] stride 16 */, u64* gcnt, u64 it1, int nb, int sleep) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        const int x = blockIdx.x & 7;
        const int per = nb >> 3;
        const u64 old = __hip_atomic_fetch_add(xcnt + 16 * x, 1ULL, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        if (old == it1 * (u64)per - 1ULL)   // last of this XCD in this round
            __hip_atomic_fetch_add(gcnt, 1ULL, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        int it = 0;
        while (ld_acq(gcnt) < it1 * 8ULL) {
            if (++it > SPIN_MAX) { ok = false; break; }
            if (sleep) __builtin_amdgcn_s_sleep(1);
        }
    }
    __syncthreads();
    return ok;
}

__global__ void k_bar(u64* cnt, int iters, int mode, int sleep, u64* out) {
    const u64 t0 = wall_clock64();
    bool ok = true;
    for (int i = 0; i < iters && ok; ++i) {
        if (mode == 0) ok = bar_flat(cnt, (u64)(i + 1) * gridDim.x, sleep);
        else if (mode == 1) ok = bar_hier(cnt + 64, cnt, (u64)(i + 1), gridDim.x, sleep);
        else ok = bar_flat_rlx(cnt, (u64)(i + 1) * gridDim.x, sleep);
    }
    const u64 t1 = wall_clock64();
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = t1 - t0;
        out[2 * blockIdx.x + 1] = ok ? 0 : 1;
    }
}

// --- exchange through a counter: nw = gridDim.x * waves per block partials
__global__ void k_xchg_cnt(u64* cnt, double* part /* [2][nw] */, int iters, u64* out, double* chk) {
    const int wpb = blockDim.x >> 6, nw = gridDim.x * wpb;
    const int w = blockIdx.x * wpb + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    extern __shared__ double lds[];
    double acc = 0.0;
    bool ok = true;
    const u64 t0 = wall_clock64();
    for (int i = 0; i < iters && ok; ++i) {
        double* p = part + (size_t)(i & 1) * nw;
        if (lane == 0) __hip_atomic_store(p + w, (double)(w + i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok = bar_flat(cnt, (u64)(i + 1) * gridDim.x, 0);
        double s = 0.0;
        for (int e = threadIdx.x; e < nw; e += blockDim.x) s += __hip_atomic_load(p + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        lds[threadIdx.x] = s;
        __syncthreads();
        if (threadIdx.x == 0) {
            double tot = 0.0;
            for (int k = 0; k < (int)blockDim.x; ++k) tot += lds[k];
            acc += tot - ((double)nw * (nw - 1) / 2.0 + (double)nw * i);   // 0 when every partial was the fresh one
        }
        __syncthreads();
    }
    const u64 t1 = wall_clock64();
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = t1 - t0;
        out[2 * blockIdx.x + 1] = ok ? 0 : 1;
        chk[blockIdx.x] = acc;
    }
}

// --- exchange through tagged 16-byte records {value, epoch}
typedef u32 u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st16(void* p, double v, u64 tag) {
    u4 x;
    const u64 vb = (u64)__double_as_longlong(v);
    x.x = (u32)vb; x.y = (u32)(vb >> 32); x.z = (u32)tag; x.w = (u32)(tag >> 32);
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(x) : "memory");
}
// all records of a thread are requested before the first is examined (buffer loads with sc0 sc1: the compiler tracks
// vmcnt, unlike an inline-asm load); records that were stale are polled again, the others are not
__global__ void k_xchg_tag(double* rec /* [2][nw][2] */, int iters, u64* out, double* chk) {
    const int wpb = blockDim.x >> 6, nw = gridDim.x * wpb;
    const int w = blockIdx.x * wpb + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    extern __shared__ double lds[];
    double acc = 0.0;
    __shared__ int bad;
    if (threadIdx.x == 0) bad = 0;
    __syncthreads();
    constexpr int MAXR = 16;
    const u64 t0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
        double* p = rec + (size_t)(i & 1) * nw * 2;
        const u64 epoch = (u64)(i + 1);
        if (lane == 0) st16(p + 2 * w, (double)(w + i), epoch);
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(p, 0, 0x7fffffff, 0x00020000);
        double val[MAXR];
        unsigned pend = 0;
#pragma unroll
        for (int k = 0; k < MAXR; ++k) {
            val[k] = 0.0;
            if ((int)threadIdx.x + k * (int)blockDim.x < nw) pend |= 1u << k;
        }
        int it = 0;
        while (pend) {
            u4 x[MAXR];
#pragma unroll
            for (int k = 0; k < MAXR; ++k)
                if (pend & (1u << k)) x[k] = __builtin_amdgcn_raw_buffer_load_b128(r, (unsigned)(threadIdx.x + k * blockDim.x) * 16u, 0, 17);
#pragma unroll
            for (int k = 0; k < MAXR; ++k)
                if (pend & (1u << k)) {
                    const u64 tag = (u64)x[k].z | ((u64)x[k].w << 32);
                    if (tag == epoch) {
                        val[k] = __longlong_as_double((long long)((u64)x[k].x | ((u64)x[k].y << 32)));
                        pend &= ~(1u << k);
                    }
                }
            if (++it > SPIN_MAX / 16) { bad = 1; break; }
        }
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < MAXR; ++k) s += val[k];
        lds[threadIdx.x] = s;
        __syncthreads();
        if (bad) break;
        if (threadIdx.x == 0) {
            double tot = 0.0;
            for (int k = 0; k < (int)blockDim.x; ++k) tot += lds[k];
            acc += tot - ((double)nw * (nw - 1) / 2.0 + (double)nw * i);
        }
        __syncthreads();   // (records of epoch i are rewritten at epoch i + 2: everybody has read epoch i before anybody can
                           // have finished epoch i + 1, because epoch i + 1 needs every wave's record)
    }
    const u64 t1 = wall_clock64();
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = t1 - t0;
        out[2 * blockIdx.x + 1] = bad;
        chk[blockIdx.x] = acc;
    }
}

// --- producer -> consumer flags: workgroup b "produces" for pair b % npairs (store a line, then raise the pair's
// counter); every wave then waits for the counter of the pair its index maps to and reads the line; a flat barrier
// closes the iteration (stands for the pair -> link exchange)
__global__ void k_flag_pc(u64* cnt, u64* pflag /* [npairs] stride 16 */, double* data /* [npairs][64] */, int npairs, int iters,
                          u64* out, double* chk) {
    const int wpb = blockDim.x >> 6;
    const int w = blockIdx.x * wpb + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int nprod = gridDim.x / npairs * npairs;   // producers: the first nprod workgroups, nprod / npairs per pair
    const int per = nprod / npairs;
    double acc = 0.0;
    bool ok = true;
    const u64 t0 = wall_clock64();
    for (int i = 0; i < iters && ok; ++i) {
        if ((int)blockIdx.x < nprod) {
            const int p = blockIdx.x % npairs, q = blockIdx.x / npairs;
            if (threadIdx.x < 64 / per) __hip_atomic_store(data + p * 64 + q * (64 / per) + threadIdx.x, (double)(i + p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            if (threadIdx.x == 0) __hip_atomic_fetch_add(pflag + 16 * p, 1ULL, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
        const int mp = w % npairs;
        if (lane == 0) {
            int it = 0;
            while (ld_acq(pflag + 16 * mp) < (u64)(i + 1) * per) {
                if (++it > SPIN_MAX) { ok = false; break; }
            }
        }
        ok = __shfl(ok ? 1 : 0, 0) != 0;
        const double v = __hip_atomic_load(data + mp * 64 + (lane / (64 / per)) * (64 / per) + lane % (64 / per), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        acc += v - (double)(i + mp);
        ok = bar_flat(cnt, (u64)(i + 1) * gridDim.x, 0) && ok;
    }
    const u64 t1 = wall_clock64();
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = t1 - t0;
        out[2 * blockIdx.x + 1] = ok ? 0 : 1;
    }
    if (lane == 0) chk[w] = acc;
}

// --- flag-array barrier: no atomics at all. Workgroup b stores the epoch into its own 8-byte flag (write-through);
// thread t of every workgroup polls flag t until it shows the epoch (one load per thread: a single round trip when
// everybody is there).  `stride` > 1 restricts the barrier to the workgroups b % stride == 0 (stride 8: one XCD).
__global__ void k_bar_flags(u64* flags, int iters, int stride, u64* out) {
    const int nb = gridDim.x;
    const bool in = (blockIdx.x % stride) == 0;
    const u64 t0 = wall_clock64();
    bool ok = true;
    if (in) {
        __shared__ int bad;
        if (threadIdx.x == 0) bad = 0;
        for (int i = 0; i < iters; ++i) {
            __syncthreads();
            const u64 epoch = (u64)(i + 1);
            if (threadIdx.x == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                __hip_atomic_store(flags + 8 * blockIdx.x, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            for (int b = threadIdx.x * stride; b < nb; b += blockDim.x * stride) {
                int it = 0;
                while (ld_agent(flags + 8 * b) < epoch)
                    if (++it > SPIN_MAX) { bad = 1; break; }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            __syncthreads();
            if (bad) { ok = false; break; }
        }
    }
    const u64 t1 = wall_clock64();
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = in ? t1 - t0 : 0;
        out[2 * blockIdx.x + 1] = ok ? 0 : 1;
    }
}
// the same without the release / acquire fences (what the flags alone cost: the L2 write-back / invalidate is separate)
__global__ void k_bar_flags_nofence(u64* flags, int iters, int stride, u64* out) {
    const int nb = gridDim.x;
    const bool in = (blockIdx.x % stride) == 0;
    const u64 t0 = wall_clock64();
    bool ok = true;
    if (in) {
        __shared__ int bad;
        if (threadIdx.x == 0) bad = 0;
        for (int i = 0; i < iters; ++i) {
            __syncthreads();
            const u64 epoch = (u64)(i + 1);
            if (threadIdx.x == 0) __hip_atomic_store(flags + 8 * blockIdx.x, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (int b = threadIdx.x * stride; b < nb; b += blockDim.x * stride) {
                int it = 0;
                while (ld_agent(flags + 8 * b) < epoch)
                    if (++it > SPIN_MAX) { bad = 1; break; }
            }
            __syncthreads();
            if (bad) { ok = false; break; }
        }
    }
    const u64 t1 = wall_clock64();
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = in ? t1 - t0 : 0;
        out[2 * blockIdx.x + 1] = ok ? 0 : 1;
    }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("device: %s, %d CUs, wall clock 100 MHz\n", prop.name, cus);
    u64 *cnt, *out;
    double *part, *chk;
    CK(hipMalloc(&cnt, 4096 * sizeof(u64)));
    CK(hipMalloc(&out, 2 * 4096 * sizeof(u64)));
    CK(hipMalloc(&part, 4 * 16384 * sizeof(double)));
    CK(hipMalloc(&chk, 16384 * sizeof(double)));
    std::vector<u64> h(2 * 4096);
    std::vector<double> hc(16384);
    auto report = [&](const char* name, int nb) -> int {
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h.data(), out, sizeof(u64) * 2 * nb, hipMemcpyDeviceToHost));
        u64 mx = 0, bad = 0;
        for (int b = 0; b < nb; ++b) { mx = h[2 * b] > mx ? h[2 * b] : mx; bad += h[2 * b + 1]; }
        printf("%-44s %7.3f us per iteration%s\n", name, (double)mx * 0.01 / iters, bad ? "   TIMED OUT" : "");
        return 0;
    };
    auto check = [&](int n) {
        hipMemcpy(hc.data(), chk, sizeof(double) * n, hipMemcpyDeviceToHost);
        double worst = 0.0;
        for (int i = 0; i < n; ++i) worst = std::max(worst, std::abs(hc[i]));
        printf("    (stale-data check: %g)\n", worst);
    };
    for (int tpb : {256, 768}) {
        const int nb = (tpb == 256) ? cus : cus;   // one workgroup per CU in both shapes
        char name[128];
        for (int mode = 0; mode < 3; ++mode)
            for (int sleep = 0; sleep < 2; ++sleep) {
                CK(hipMemset(cnt, 0, 4096 * sizeof(u64)));
                hipLaunchKernelGGL(k_bar, dim3(nb), dim3(tpb), 0, 0, cnt, iters, mode, sleep, out);
                snprintf(name, sizeof name, "barrier %s%s, %d x %d", mode == 0 ? "flat" : mode == 1 ? "hier" : "flat-relaxed-poll", sleep ? "+sleep" : "", nb, tpb);
                if (report(name, nb)) return 1;
            }
        CK(hipMemset(cnt, 0, 4096 * sizeof(u64)));
        CK(hipMemset(part, 0, 4 * 16384 * sizeof(double)));
        hipLaunchKernelGGL(k_xchg_cnt, dim3(nb), dim3(tpb), tpb * sizeof(double), 0, cnt, part, iters, out, chk);
        snprintf(name, sizeof name, "exchange via counter, %d waves", nb * tpb / 64);
        if (report(name, nb)) return 1;
        check(nb);
        CK(hipMemset(part, 0, 4 * 16384 * sizeof(double)));
        hipLaunchKernelGGL(k_xchg_tag, dim3(nb), dim3(tpb), tpb * sizeof(double), 0, part, iters, out, chk);
        snprintf(name, sizeof name, "exchange via tagged records, %d waves", nb * tpb / 64);
        if (report(name, nb)) return 1;
        check(nb);
        CK(hipMemset(cnt, 0, 4096 * sizeof(u64)));
        CK(hipMemset(part, 0, 4 * 16384 * sizeof(double)));
        hipLaunchKernelGGL(k_flag_pc, dim3(nb), dim3(tpb), 0, 0, cnt, cnt + 256, part, 55, iters, out, chk);
        snprintf(name, sizeof name, "pair flags (55) + flat barrier, %d x %d", nb, tpb);
        if (report(name, nb)) return 1;
        check(nb * tpb / 64);
    }
    // three resident workgroups of 256 per CU (the pair kernel's shape): barrier over 768 workgroups
    {
        const int nb = 3 * cus;
        CK(hipMemset(cnt, 0, 4096 * sizeof(u64)));
        hipLaunchKernelGGL(k_bar, dim3(nb), dim3(256), 0, 0, cnt, iters, 0, 0, out);
        if (report("barrier flat, 3 workgroups per CU x 256", nb)) return 1;
        CK(hipMemset(cnt, 0, 4096 * sizeof(u64)));
        hipLaunchKernelGGL(k_bar, dim3(nb), dim3(256), 0, 0, cnt, iters, 1, 0, out);
        if (report("barrier hier, 3 workgroups per CU x 256", nb)) return 1;
    }
    for (int stride : {1, 8}) {
        for (int tpb : {256, 768}) {
            char name[128];
            CK(hipMemset(cnt, 0, 4096 * sizeof(u64)));
            hipLaunchKernelGGL(k_bar_flags, dim3(cus), dim3(tpb), 0, 0, cnt, iters, stride, out);
            snprintf(name, sizeof name, "flag-array barrier, %d x %d%s", cus / stride, tpb, stride == 8 ? " (one XCD)" : "");
            if (report(name, cus)) return 1;
            CK(hipMemset(cnt, 0, 4096 * sizeof(u64)));
            hipLaunchKernelGGL(k_bar_flags_nofence, dim3(cus), dim3(tpb), 0, 0, cnt, iters, stride, out);
            snprintf(name, sizeof name, "flag-array barrier w/o fences, %d x %d%s", cus / stride, tpb, stride == 8 ? " (one XCD)" : "");
            if (report(name, cus)) return 1;
        }
    }
    return 0;
}
