// Do v_mfma_f64_16x16x4_f64 and v_mfma_f64_4x4x4_4b_f64 round alike?  Both add four products to an accumulator per output
// element; if they do so in the same order with the same intermediate roundings, a network evaluated on 4x4x4 blocks can
// reproduce the bits of its 16x16x4 evaluation.  One wave: D16 = A(16x4) B(4x16) + C over random data with wide exponent
// spread, chained over several k-steps, against the same chain on 4x4x4 blocks for the columns 0..3.
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/var/ubench_mfma_equal tools/ubench_mfma_equal.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
typedef double v4d __attribute__((ext_vector_type(4)));

__global__ void k(const double* A, const double* B, const double* C, double* out16, double* out4, int steps) {
    const int lane = threadIdx.x & 63;
    const int h = lane >> 4, b = (lane >> 2) & 3, l = lane & 3;
    v4d acc16;
    for (int r = 0; r < 4; ++r) acc16[r] = C[r * 64 + lane];
    // 4x4x4 form: lane (h, b, l) accumulates H[4 b + h][env l] = lane (16 h + l), register b of the 16x16 form
    double acc4 = C[b * 64 + 16 * h + l];
    for (int s = 0; s < steps; ++s) {
        const double a = A[s * 64 + lane], x = B[s * 64 + lane];
        acc16 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, x, acc16, 0, 0, 0);
        const double x4 = B[s * 64 + 16 * h + l];   // X[k = h][env l]
        acc4 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, x4, acc4, 0, 0, 0);
    }
    for (int r = 0; r < 4; ++r) out16[r * 64 + lane] = acc16[r];
    out4[lane] = acc4;
}

int main() {
    const int steps = 16, trials = 2000;
    double *A, *B, *C, *o16, *o4;
    hipMalloc(&A, steps * 64 * 8); hipMalloc(&B, steps * 64 * 8); hipMalloc(&C, 256 * 8); hipMalloc(&o16, 256 * 8); hipMalloc(&o4, 64 * 8);
    double hA[steps * 64], hB[steps * 64], hC[256], h16[256], h4[64];
    srand(1);
    long bad = 0, total = 0;
    for (int t = 0; t < trials; ++t) {
        auto rnd = [&]() { return (rand() / (double)RAND_MAX - 0.5) * ldexp(1.0, rand() % 24 - 12); };
        for (auto& v : hA) v = rnd();
        for (auto& v : hB) v = rnd();
        for (auto& v : hC) v = rnd();
        hipMemcpy(A, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(B, hB, sizeof hB, hipMemcpyHostToDevice); hipMemcpy(C, hC, sizeof hC, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, A, B, C, o16, o4, steps);
        hipMemcpy(h16, o16, sizeof h16, hipMemcpyDeviceToHost); hipMemcpy(h4, o4, sizeof h4, hipMemcpyDeviceToHost);
        for (int lane = 0; lane < 64; ++lane) {
            const int h = lane >> 4, b = (lane >> 2) & 3, l = lane & 3;
            const double want = h16[b * 64 + 16 * h + l];
            total += 1;
            if (memcmp(&want, &h4[lane], 8) != 0) { if (bad < 5) printf("trial %d lane %d: 16x16x4 %.17g  4x4x4 %.17g\n", t, lane, want, h4[lane]); bad += 1; }
        }
    }
    printf("%ld of %ld results differ between the two MFMA shapes\n", bad, total);
    return 0;
}
