// kernel-boundary cost on one stream: N dependent launches of tiny kernels
#include <hip/hip_runtime.h>
#include <cstdio>
struct Big { double x[80]; };
__global__ void k_tiny(double* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1.0; }
__global__ void k_big(Big b, double* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += b.x[3]; }
__global__ void k_touch(double* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = p[i] * 1.0001 + 1.0; }
int main() {
    double* d; hipMalloc(&d, 64 << 20); hipMemset(d, 0, 64 << 20);
    hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    Big b{}; const int N = 2000; float ms;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0, st); for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_tiny, dim3(1), dim3(256), 0, st, d); hipEventRecord(e1, st); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1); printf("tiny 1 block: %.2f us/launch\n", ms * 1e3 / N);
        hipEventRecord(e0, st); for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_tiny, dim3(512), dim3(256), 0, st, d); hipEventRecord(e1, st); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1); printf("tiny 512 blocks: %.2f us/launch\n", ms * 1e3 / N);
        hipEventRecord(e0, st); for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_big, dim3(2), dim3(256), 0, st, b, d); hipEventRecord(e1, st); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1); printf("640-byte kernarg 2 blocks: %.2f us/launch\n", ms * 1e3 / N);
        hipEventRecord(e0, st); for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_touch, dim3(5120), dim3(256), 0, st, d, 1310720); hipEventRecord(e1, st); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1); printf("touch 10 MB (rw): %.2f us/launch\n", ms * 1e3 / N);
    }
    // graph of 300 launches
    hipGraph_t g; hipGraphExec_t ge; hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
    for (int i = 0; i < 100; ++i) { hipLaunchKernelGGL(k_tiny, dim3(1), dim3(256), 0, st, d); hipLaunchKernelGGL(k_tiny, dim3(512), dim3(256), 0, st, d); hipLaunchKernelGGL(k_big, dim3(2), dim3(256), 0, st, b, d); }
    hipStreamEndCapture(st, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphLaunch(ge, st); hipStreamSynchronize(st);
    hipEventRecord(e0, st); for (int i = 0; i < 10; ++i) hipGraphLaunch(ge, st); hipEventRecord(e1, st); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1); printf("graph of 300 tiny kernels: %.2f us/kernel\n", ms * 1e3 / 3000);
    return 0;
}
