// k_potf2_inv (linalg.hip: the 64x64 diagonal block of the blocked Cholesky, factor + inverse) on its own: time per
// launch over a dependent chain of launches, residuals |L L^T - A| and |X L - I|, and (-DPOTF2_STAMPS) its phases.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Ipilco_amd/csrc -mllvm -amdgpu-mfma-vgpr-form [-DPOTF2_STAMPS] tools/ubench_potf2.hip -o exp/ubench_potf2
#include "../pilco_amd/csrc/linalg.hip"

#include <cmath>
#include <cstdio>
#include <vector>

int main() {
    const int npad = 1024, batch = 10, reps = 200;
    std::vector<double> A((size_t)batch * npad * npad, 0.0);
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (double)(s >> 8) / 16777216.0 - 0.5; };
    for (int b = 0; b < batch; ++b) {   // block (0,0) of every matrix: G G^T + 64 I
        std::vector<double> G(64 * 64);
        for (double& g : G) g = rnd();
        for (int i = 0; i < 64; ++i)
            for (int j = 0; j < 64; ++j) {
                double acc = (i == j) ? 8.0 : 0.0;
                for (int k = 0; k < 64; ++k) acc += G[i * 64 + k] * G[j * 64 + k];
                A[(size_t)b * npad * npad + (size_t)i * npad + j] = acc;
            }
    }
    double *dA, *dA0, *dInv;
    int* dInfo;
    hipMalloc(&dA, A.size() * 8); hipMalloc(&dA0, A.size() * 8); hipMalloc(&dInv, A.size() * 8); hipMalloc(&dInfo, 4096);
    hipMemcpy(dA0, A.data(), A.size() * 8, hipMemcpyHostToDevice);
    hipMemset(dInfo, 0, 4096);
    hipStream_t st; hipStreamCreate(&st);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipMemcpy(dA, dA0, A.size() * 8, hipMemcpyDeviceToDevice);
    hipLaunchKernelGGL(pilco::k_potf2_inv, dim3(batch), dim3(64 * pilco::POTF2_NW), 0, st, dA, npad, 0, dInv, dInfo);
    hipStreamSynchronize(st);
    std::vector<double> L(A.size()), X(A.size());
    hipMemcpy(L.data(), dA, A.size() * 8, hipMemcpyDeviceToHost);
    hipMemcpy(X.data(), dInv, X.size() * 8, hipMemcpyDeviceToHost);
    double e_llt = 0, e_xl = 0;
    for (int b = 0; b < batch; ++b)
        for (int i = 0; i < 64; ++i)
            for (int j = 0; j < 64; ++j) {
                double llt = 0, xl = 0;
                for (int k = 0; k < 64; ++k) {
                    llt += L[(size_t)b * npad * npad + (size_t)i * npad + k] * L[(size_t)b * npad * npad + (size_t)j * npad + k];
                    xl += X[(size_t)b * npad * npad + (size_t)i * npad + k] * L[(size_t)b * npad * npad + (size_t)k * npad + j];
                }
                e_llt = std::fmax(e_llt, std::fabs(llt - A[(size_t)b * npad * npad + (size_t)i * npad + j]));
                e_xl = std::fmax(e_xl, std::fabs(xl - (i == j ? 1.0 : 0.0)));
            }
    // the block is overwritten by its factor: every timed launch factors the previous launch's output again (SPD as well: L + L^T
    // dominated by the diagonal is not guaranteed) -- so restore it by a copy in between and subtract the copy's own time
    auto timed = [&](bool with_kernel) {
        hipEventRecord(e0, st);
        for (int r = 0; r < reps; ++r) {
            hipMemcpy2DAsync(dA, npad * 8, dA0, npad * 8, 64 * 8, 64, hipMemcpyDeviceToDevice, st);
            if (with_kernel) hipLaunchKernelGGL(pilco::k_potf2_inv, dim3(batch), dim3(64 * pilco::POTF2_NW), 0, st, dA, npad, 0, dInv, dInfo);
        }
        hipEventRecord(e1, st);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        return ms * 1e3 / reps;
    };
    timed(true);
    const double t_with = timed(true), t_copy = timed(false);
    printf("k_potf2_inv: %.2f us per launch (chain of %d; restoring copy %.2f us subtracted)   |L L^T - A| %.2e  |X L - I| %.2e\n",
           t_with - t_copy, reps, t_copy, e_llt, e_xl);
#ifdef POTF2_STAMPS
    unsigned long long stp[8];
    hipMemcpy(stp, dInfo + 256, sizeof(stp), hipMemcpyDeviceToHost);
    auto us = [&](int a, int b) { return (double)(stp[b] - stp[a]) / 100.0; };
    printf("phases (matrix 0): load %.2f  factor (4 panels + 3 updates) %.2f  factor store + inverse %.2f  inverse store %.2f  total %.2f us\n",
           us(0, 1), us(1, 2), us(2, 3), us(3, 5), us(0, 5));
#endif
    return 0;
}
