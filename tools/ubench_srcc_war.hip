// Does the hardware order "f64 MFMA reads SrcC = R" against a following f64 MFMA that WRITES R (in-place accumulate)?
//   v_mfma_f64_16x16x4_f64 v[8:15], a, b0, v[0:7]      ; chain 0 starts from R = v[0:7]
//   <gap: 0..n independent VALU instructions>
//   v_mfma_f64_16x16x4_f64 v[0:7],  a, b1, v[0:7]      ; chain 1 accumulates in place in R
// v[8:15] must equal a*b0 + R(old).  Build: hipcc --offload-arch=gfx950 -O2 tools/ubench_srcc_war.hip -o /tmp/srcc && /tmp/srcc
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int GAP>
__global__ void k(const double* in, double* out, int reps) {
    const int lane = threadIdx.x;
    double a = in[lane], b0 = in[64 + lane], b1 = in[128 + lane];
    d4 r = {in[192 + lane], in[256 + lane], in[320 + lane], in[384 + lane]};
    d4 bad = {0, 0, 0, 0};
    for (int it = 0; it < reps; ++it) {
        d4 e0, e1 = r;
        double t0 = a, t1 = a;
        if (GAP == 0)
            asm volatile("v_mfma_f64_16x16x4_f64 %0, %2, %3, %1\n\tv_mfma_f64_16x16x4_f64 %1, %2, %4, %1\n\ts_nop 15\n\ts_nop 15"
                         : "=&v"(e0), "+v"(e1) : "v"(a), "v"(b0), "v"(b1));
        else if (GAP == 1)
            asm volatile("v_mfma_f64_16x16x4_f64 %[e0], %[a], %[b0], %[e1]\n\tv_mul_f64 %[t0], %[t0], %[t0]\n\tv_mfma_f64_16x16x4_f64 %[e1], %[a], %[b1], %[e1]\n\ts_nop 15\n\ts_nop 15"
                         : [e0] "=&v"(e0), [e1] "+v"(e1), [t0] "+v"(t0) : [a] "v"(a), [b0] "v"(b0), [b1] "v"(b1));
        else if (GAP == 2)
            asm volatile("v_mfma_f64_16x16x4_f64 %[e0], %[a], %[b0], %[e1]\n\tv_mul_f64 %[t0], %[t0], %[t0]\n\tv_mul_f64 %[t1], %[t1], %[t1]\n\tv_mfma_f64_16x16x4_f64 %[e1], %[a], %[b1], %[e1]\n\ts_nop 15\n\ts_nop 15"
                         : [e0] "=&v"(e0), [e1] "+v"(e1), [t0] "+v"(t0), [t1] "+v"(t1) : [a] "v"(a), [b0] "v"(b0), [b1] "v"(b1));
        else if (GAP == 3)
            asm volatile("v_mfma_f64_16x16x4_f64 %0, %2, %3, %1\n\ts_nop 7\n\tv_mfma_f64_16x16x4_f64 %1, %2, %4, %1\n\ts_nop 15\n\ts_nop 15"
                         : "=&v"(e0), "+v"(e1) : "v"(a), "v"(b0), "v"(b1));
        else if (GAP == 4)
            asm volatile("v_mfma_f64_16x16x4_f64 %0, %2, %3, %1\n\ts_nop 0\n\tv_mfma_f64_16x16x4_f64 %1, %2, %4, %1\n\ts_nop 15\n\ts_nop 15"
                         : "=&v"(e0), "+v"(e1) : "v"(a), "v"(b0), "v"(b1));
        else if (GAP == 5)
            asm volatile("v_mfma_f64_16x16x4_f64 %0, %2, %3, %1\n\ts_nop 2\n\tv_mfma_f64_16x16x4_f64 %1, %2, %4, %1\n\ts_nop 15\n\ts_nop 15"
                         : "=&v"(e0), "+v"(e1) : "v"(a), "v"(b0), "v"(b1));
        else if (GAP == 6)
            asm volatile("v_mfma_f64_16x16x4_f64 %[e0], %[a], %[b0], %[e1]\n\tv_mov_b64 %[t0], %[t0]\n\tv_mfma_f64_16x16x4_f64 %[e1], %[a], %[b1], %[e1]\n\ts_nop 15\n\ts_nop 15"
                         : [e0] "=&v"(e0), [e1] "+v"(e1), [t0] "+v"(t0) : [a] "v"(a), [b0] "v"(b0), [b1] "v"(b1));
        if (GAP == 7) {   // VALU writes SrcC, the MFMA reads it in the next slot
            d4 x = r;
            asm volatile("v_mov_b64 %[e1a], %[xa]\n\tv_mov_b64 %[e1b], %[xb]\n\tv_mov_b64 %[e1c], %[xc]\n\tv_mov_b64 %[e1d], %[xd]\n\t"
                         "v_mfma_f64_16x16x4_f64 %[e0], %[a], %[b0], %[e1]\n\ts_nop 15\n\ts_nop 15"
                         : [e0] "=&v"(e0), [e1] "=&v"(e1)
                         : [a] "v"(a), [b0] "v"(b0), [xa] "v"(x[0]), [xb] "v"(x[1]), [xc] "v"(x[2]), [xd] "v"(x[3]), [e1a] "v"(e1[0]), [e1b] "v"(e1[1]), [e1c] "v"(e1[2]),
                           [e1d] "v"(e1[3]));
        }
        if (GAP == 8) {   // VALU writes SrcA (the exp's exponent insertion), the MFMA reads it in the next slot
            double a2 = a;
            double half = 0.5 * a;
            asm volatile("v_add_f64 %[a2], %[h], %[h]\n\tv_mfma_f64_16x16x4_f64 %[e0], %[a2], %[b0], %[e1]\n\ts_nop 15\n\ts_nop 15"
                         : [e0] "=&v"(e0), [a2] "=&v"(a2) : [b0] "v"(b0), [e1] "v"(e1), [h] "v"(half));
        }
        if (GAP >= 10 && GAP < 30) {   // ... with GAP - 10 wait states (s_nop) or independent VALU instructions (GAP >= 20) in between
            double a2 = a, half = 0.5 * a;
            if (GAP == 10) asm volatile("v_add_f64 %[a2], %[h], %[h]\n\ts_nop 0\n\tv_mfma_f64_16x16x4_f64 %[e0], %[a2], %[b0], %[e1]\n\ts_nop 15\n\ts_nop 15" : [e0] "=&v"(e0), [a2] "=&v"(a2) : [b0] "v"(b0), [e1] "v"(e1), [h] "v"(half));
            if (GAP == 11) asm volatile("v_add_f64 %[a2], %[h], %[h]\n\ts_nop 1\n\tv_mfma_f64_16x16x4_f64 %[e0], %[a2], %[b0], %[e1]\n\ts_nop 15\n\ts_nop 15" : [e0] "=&v"(e0), [a2] "=&v"(a2) : [b0] "v"(b0), [e1] "v"(e1), [h] "v"(half));
            if (GAP == 12) asm volatile("v_add_f64 %[a2], %[h], %[h]\n\ts_nop 2\n\tv_mfma_f64_16x16x4_f64 %[e0], %[a2], %[b0], %[e1]\n\ts_nop 15\n\ts_nop 15" : [e0] "=&v"(e0), [a2] "=&v"(a2) : [b0] "v"(b0), [e1] "v"(e1), [h] "v"(half));
            if (GAP == 13) asm volatile("v_add_f64 %[a2], %[h], %[h]\n\ts_nop 3\n\tv_mfma_f64_16x16x4_f64 %[e0], %[a2], %[b0], %[e1]\n\ts_nop 15\n\ts_nop 15" : [e0] "=&v"(e0), [a2] "=&v"(a2) : [b0] "v"(b0), [e1] "v"(e1), [h] "v"(half));
            if (GAP == 15) asm volatile("v_add_f64 %[a2], %[h], %[h]\n\ts_nop 7\n\tv_mfma_f64_16x16x4_f64 %[e0], %[a2], %[b0], %[e1]\n\ts_nop 15\n\ts_nop 15" : [e0] "=&v"(e0), [a2] "=&v"(a2) : [b0] "v"(b0), [e1] "v"(e1), [h] "v"(half));
        }
        // reference: the same product with an accumulator nobody writes
        d4 ref = r;
        asm volatile("s_nop 4\n\tv_mfma_f64_16x16x4_f64 %0, %1, %2, %0\n\ts_nop 15\n\ts_nop 15" : "+v"(ref) : "v"(a), "v"(b0));
        for (int q = 0; q < 4; ++q)
            if (e0[q] != ref[q]) bad[q] += 1.0;
        a += (double)(t0 != t0) + (double)(t1 != t1);   // (keeps the fillers alive; adds 0)
    }
    for (int q = 0; q < 4; ++q) out[q * 64 + lane] = bad[q];
}
template <int GAP>
static void run(const double* din, double* dout, const char* name) {
    hipLaunchKernelGGL(k<GAP>, dim3(1), dim3(64), 0, 0, din, dout, 20000);
    std::vector<double> h(256);
    hipMemcpy(h.data(), dout, sizeof(double) * 256, hipMemcpyDeviceToHost);
    double tot = 0;
    for (double v : h) tot += v;
    printf("%-46s mismatching result entries over 20000 x 256: %.0f\n", name, tot);
}
int main() {
    std::vector<double> h(448);
    for (int i = 0; i < 448; ++i) h[i] = 0.25 + 0.001 * ((i * 7919) % 1013);
    double *din, *dout;
    hipMalloc(&din, sizeof(double) * 448);
    hipMalloc(&dout, sizeof(double) * 256);
    hipMemcpy(din, h.data(), sizeof(double) * 448, hipMemcpyHostToDevice);
    run<0>(din, dout, "MFMA(C=R) ; MFMA(D=C=R) back to back");
    run<1>(din, dout, "... one v_mul_f64 between");
    run<2>(din, dout, "... two v_mul_f64 between");
    run<3>(din, dout, "... s_nop 7 between");
    run<4>(din, dout, "... s_nop 0 between");
    run<5>(din, dout, "... s_nop 2 between");
    run<6>(din, dout, "... one v_mov_b64 between");
    run<8>(din, dout, "VALU write of SrcA ; MFMA next slot");
    run<10>(din, dout, "... s_nop 0 between");
    run<11>(din, dout, "... s_nop 1 between");
    run<12>(din, dout, "... s_nop 2 between");
    run<13>(din, dout, "... s_nop 3 between");
    run<15>(din, dout, "... s_nop 7 between");
    return 0;
}
