// v_mfma_f64_4x4x4_4b_f64 on gfx950: cycles per instruction (independent chains, 1 and 2 waves per SIMD) and the lane
// layout of its operands, found by probing: A = 1 in one lane, B = 1 in one lane, which lanes of D receive the product.
// Build: hipcc --offload-arch=gfx950 -O3 -o ubench_mfma4 ubench_mfma4.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(512) void k_time(double* out, long long* cyc, int iters, double seed) {
    const int lane = threadIdx.x & 63;
    double acc[8], f[16];
    for (int i = 0; i < 8; ++i) acc[i] = seed;
    for (int i = 0; i < 16; ++i) f[i] = seed * (i + 1) + lane * 1e-9;
    const double a = seed + lane * 1e-3;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 16; ++s)
#pragma unroll
            for (int t = 0; t < 8; ++t) acc[t] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, f[s], acc[t], 0, 0, 0);
    }
    const long long t1 = clock64();
    double r = 0;
    for (int i = 0; i < 8; ++i) r += acc[i];
    out[(long)blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (lane == 0) cyc[(long)blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = t1 - t0;
}

// one wave per (la, lb) pair: out[pair][lane] = D
__global__ __launch_bounds__(64) void k_probe(double* out) {
    const int lane = threadIdx.x, la = blockIdx.x >> 6, lb = blockIdx.x & 63;
    const double a = lane == la ? 1. : 0., b = lane == lb ? 1. : 0.;
    const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0., 0, 0, 0);
    out[(long)blockIdx.x * 64 + lane] = d;
}

int main() {
    double* out; long long* cyc;
    CHECK(hipMalloc(&out, 4096 * 64 * sizeof(double)));
    CHECK(hipMalloc(&cyc, 256 * 16 * sizeof(long long)));
    for (int block = 256; block <= 512; block += 256) {
        const int iters = 2000, grid = 256;
        hipLaunchKernelGGL(k_time, dim3(grid), dim3(block), 0, 0, out, cyc, 2, 1.0000001);
        CHECK(hipDeviceSynchronize());
        hipLaunchKernelGGL(k_time, dim3(grid), dim3(block), 0, 0, out, cyc, iters, 1.0000001);
        CHECK(hipDeviceSynchronize());
        const int nw = grid * block / 64;
        std::vector<long long> h(nw);
        CHECK(hipMemcpy(h.data(), cyc, nw * sizeof(long long), hipMemcpyDeviceToHost));
        double avg = 0;
        for (auto v : h) avg += (double)v;
        printf("mfma_f64_4x4x4 x128 (8 chains) block %d (waves/SIMD %d): %.1f clk/iter/wave = %.2f clk per instruction\n", block,
               block / 256, avg / nw / iters, avg / nw / iters / 128);
    }
    hipLaunchKernelGGL(k_probe, dim3(4096), dim3(64), 0, 0, out);
    CHECK(hipDeviceSynchronize());
    std::vector<double> h(4096 * 64);
    CHECK(hipMemcpy(h.data(), out, h.size() * sizeof(double), hipMemcpyDeviceToHost));
    // for every D lane: the (A lane, B lane) pairs that reach it
    for (int ld = 0; ld < 64; ++ld) {
        printf("D lane %2d <-", ld);
        for (int la = 0; la < 64; ++la)
            for (int lb = 0; lb < 64; ++lb)
                if (h[((long)la * 64 + lb) * 64 + ld] != 0.) printf(" (A%d,B%d)", la, lb);
        printf("\n");
    }
    return 0;
}
