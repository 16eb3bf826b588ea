// Probe: fine-grained device memory shared between two PROCESSES on one GPU through hipIpc handles, a kernel of process A
// spinning (bounded) on a flag that a kernel of process B sets with system-scope stores.   ./ipc_probe A|B <file>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <unistd.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
__global__ void k_wait(unsigned long long* area, unsigned long long want, long limit) {
    if (threadIdx.x == 0) {
        long it = 0;
        unsigned long long v;
        while ((v = __hip_atomic_load(area, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM)) < want && it < limit) { __builtin_amdgcn_s_sleep(16); ++it; }
        area[2] = it;
        area[3] = __hip_atomic_load(area + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
__global__ void k_set(unsigned long long* area, unsigned long long val) {
    if (threadIdx.x == 0) {
        __hip_atomic_store(area + 1, 0xabcdefULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __threadfence_system();
        __hip_atomic_store(area, val, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
int main(int argc, char** argv) {
    if (argc < 3) return 1;
    const bool fine = argc > 3 && atoi(argv[3]);
    CK(hipSetDevice(0));
    if (argv[1][0] == 'A') {
        unsigned long long* p = nullptr;
        if (fine) CK(hipExtMallocWithFlags((void**)&p, 1 << 16, hipDeviceMallocFinegrained));
        else CK(hipMalloc((void**)&p, 1 << 16));
        CK(hipMemset(p, 0, 1 << 16));
        hipIpcMemHandle_t h;
        CK(hipIpcGetMemHandle(&h, p));
        FILE* f = fopen(argv[2], "wb"); fwrite(&h, sizeof(h), 1, f); fclose(f);
        hipLaunchKernelGGL(k_wait, dim3(1), dim3(64), 0, 0, p, 7ULL, 4000000L);
        CK(hipDeviceSynchronize());
        unsigned long long out[4];
        CK(hipMemcpy(out, p, sizeof(out), hipMemcpyDeviceToHost));
        printf("A(fine=%d): flag %llu data %llx spins %llu seen-data %llx -> %s\n", (int)fine, out[0], out[1], out[2], out[3], (out[0] == 7 && out[3] == 0xabcdef) ? "OK" : "TIMEOUT/STALE");
    } else {
        hipIpcMemHandle_t h;
        for (int i = 0; i < 100; ++i) { FILE* f = fopen(argv[2], "rb"); if (f && fread(&h, sizeof(h), 1, f) == 1) { fclose(f); break; } if (f) fclose(f); usleep(100000); }
        unsigned long long* q = nullptr;
        CK(hipIpcOpenMemHandle((void**)&q, h, hipIpcMemLazyEnablePeerAccess));
        usleep(300000);
        hipLaunchKernelGGL(k_set, dim3(1), dim3(64), 0, 0, q, 7ULL);
        CK(hipDeviceSynchronize());
        printf("B: set\n");
        CK(hipIpcCloseMemHandle(q));
    }
    return 0;
}
