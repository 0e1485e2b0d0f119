// Issue-rate micro-benchmark for gfx950: how many shader-clock cycles fp64 MFMA / VALU instructions cost per wave,
// alone and mixed, at 1 and 2 waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 -o ubench_issue ubench_issue.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef double v4d __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

enum { M_MFMA = 0, M_FMA, M_PHASED, M_INTERLEAVED, M_RCP, M_LDEXP, M_RNDNE, M_CVT, M_MUL, M_ADD, M_CNDMASK, M_DSREAD, M_FMA_DEP, M_SELECT64, M_DIV, M_AND, M_MIN64, M_CMP64, M_SELECT32, M_MAX3, M_IDX_FP, M_IDX_INT, M_SPLIT_HALF, M_SPLIT_PARITY, M_COUNT };
static const char* names[] = {"mfma x64 (4 chains)", "fma_f64 x1024 (16 chains)", "phased 64 mfma + 1024 fma",
                              "interleaved 1 mfma : 16 fma", "rcp_f64 x256", "ldexp_f64 x256", "rndne_f64 x256",
                              "cvt_i32_f64 x256", "mul_f64 x1024", "add_f64 x1024", "cndmask_b32 x1024", "ds_read_b64 x256",
                              "fma_f64 x1024 (1 dependent chain)", "cmp_gt_f64 + 2 cndmask x256", "IEEE div_f64 x256", "and_b32 x1024",
                              "min_f64 x1024", "cmp_gt_f64 -> ballot x256", "cmp_gt_u32 + cndmask x256", "max3_u32 x1024",
                              "tanh index, fp64 clamp x256", "tanh index, integer clamp x256",
                              "waves 0-3: 64 mfma, waves 4-7: 1024 fma", "even waves: 64 mfma, odd waves: 1024 fma"};

template <int MODE> __global__ __launch_bounds__(512) void k(double* out, long long* cyc, int iters, double seed) {
    __shared__ double lds[4096];
    for (int j = threadIdx.x; j < 4096; j += blockDim.x) lds[j] = seed * j;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    v4d acc[4];
    double f[16];
    for (int i = 0; i < 4; ++i) acc[i] = (v4d){seed, seed, seed, seed};
    for (int i = 0; i < 16; ++i) f[i] = seed * (i + 1) + lane * 1e-9;
    const double a = seed + lane * 1e-3, b = seed * 0.5;
    int ii[16];
    for (int i = 0; i < 16; ++i) ii[i] = lane + i;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if (MODE == M_MFMA || MODE == M_PHASED) {
#pragma unroll
            for (int s = 0; s < 16; ++s)
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, f[s], acc[t], 0, 0, 0);
        }
        if (MODE == M_FMA || MODE == M_PHASED) {
#pragma unroll
            for (int s = 0; s < 64; ++s)
#pragma unroll
                for (int t = 0; t < 16; ++t) f[t] = fma(f[t], a, b);
        }
        if (MODE == M_INTERLEAVED) {
#pragma unroll
            for (int s = 0; s < 16; ++s)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
#pragma unroll
                    for (int u = 0; u < 16; ++u) f[u] = fma(f[u], a, b);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 16, 0);
                }
        }
        if (MODE == M_RCP) {
#pragma unroll
            for (int s = 0; s < 16; ++s)
#pragma unroll
                for (int t = 0; t < 16; ++t) f[t] = __builtin_amdgcn_rcp(f[t]);
        }
        if (MODE == M_LDEXP) {
#pragma unroll
            for (int s = 0; s < 16; ++s)
#pragma unroll
                for (int t = 0; t < 16; ++t) { f[t] = __builtin_amdgcn_ldexp(f[t], ii[t]); asm volatile("" : "+v"(f[t])); }
        }
        if (MODE == M_RNDNE) {
#pragma unroll
            for (int s = 0; s < 16; ++s)
#pragma unroll
                for (int t = 0; t < 16; ++t) { f[t] = __builtin_rint(f[t]); asm volatile("" : "+v"(f[t])); }
        }
        if (MODE == M_CVT) {
#pragma unroll
            for (int s = 0; s < 16; ++s)
#pragma unroll
                for (int t = 0; t < 16; ++t) { ii[t] = (int)f[t]; f[t] += ii[t] == 12345 ? 1.0 : 0.0; }
        }
        if (MODE == M_MUL) {
#pragma unroll
            for (int s = 0; s < 64; ++s)
#pragma unroll
                for (int t = 0; t < 16; ++t) f[t] = f[t] * a;
        }
        if (MODE == M_ADD) {
#pragma unroll
            for (int s = 0; s < 64; ++s)
#pragma unroll
                for (int t = 0; t < 16; ++t) f[t] = f[t] + a;
        }
        if (MODE == M_CNDMASK) {
#pragma unroll
            for (int s = 0; s < 64; ++s)
#pragma unroll
                for (int t = 0; t < 16; ++t) ii[t] = (lane & (1 << (s & 3))) ? ii[t] + 1 : ii[(t + 1) & 15];
        }
        if (MODE == M_FMA_DEP) {
#pragma unroll
            for (int s = 0; s < 1024; ++s) f[0] = fma(f[0], a, b);
        }
        if (MODE == M_SELECT64) {
#pragma unroll
            for (int s = 0; s < 16; ++s)
#pragma unroll
                for (int t = 0; t < 16; ++t) { f[t] = (f[t] > a) ? b : f[t]; asm volatile("" : "+v"(f[t])); }
        }
        if (MODE == M_DIV) {
#pragma unroll
            for (int s = 0; s < 16; ++s)
#pragma unroll
                for (int t = 0; t < 16; ++t) { f[t] = a / f[t]; asm volatile("" : "+v"(f[t])); }
        }
        if (MODE == M_AND) {
#pragma unroll
            for (int s = 0; s < 64; ++s)
#pragma unroll
                for (int t = 0; t < 16; ++t) { ii[t] = ii[t] & (lane + s); asm volatile("" : "+v"(ii[t])); }
        }
        if (MODE == M_MIN64) {
#pragma unroll
            for (int s = 0; s < 64; ++s)
#pragma unroll
                for (int t = 0; t < 16; ++t) { f[t] = __builtin_fmin(f[t], a); asm volatile("" : "+v"(f[t])); }
        }
        if (MODE == M_CMP64) {
            unsigned long long m = 0;
#pragma unroll
            for (int s = 0; s < 16; ++s)
#pragma unroll
                for (int t = 0; t < 16; ++t) { m ^= __ballot(f[t] > a + s); }
            ii[0] += (int)m;
        }
        if (MODE == M_SELECT32) {
#pragma unroll
            for (int s = 0; s < 16; ++s)
#pragma unroll
                for (int t = 0; t < 16; ++t) { ii[t] = ((unsigned)ii[t] > (unsigned)(lane + s)) ? s : ii[t]; asm volatile("" : "+v"(ii[t])); }
        }
        if (MODE == M_MAX3) {
#pragma unroll
            for (int s = 0; s < 64; ++s)
#pragma unroll
                for (int t = 0; t < 16; ++t) { unsigned u = ii[t], v = ii[(t + 1) & 15], w = lane + s; u = u > v ? u : v; ii[t] = u > w ? u : w; asm volatile("" : "+v"(ii[t])); }
        }
        if (MODE == M_IDX_FP) {
#pragma unroll
            for (int s = 0; s < 16; ++s)
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    double x = __builtin_fabs(f[t]);
                    x = (x > 20.0) ? 20.0 : x;
                    const double jm = x + 26388279066624.0;
                    const double d = x - (jm - 26388279066624.0);
                    unsigned j = (unsigned)__double2loint(jm);
                    j = j > 5120u ? 5120u : j;
                    ii[t] += j << 3;
                    f[t] = d + a;
                    asm volatile("" : "+v"(f[t]), "+v"(ii[t]));
                }
        }
        if (MODE == M_IDX_INT) {
            unsigned hm = 0;
#pragma unroll
            for (int s = 0; s < 16; ++s)
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    const unsigned hi = (unsigned)__double2hiint(f[t]) & 0x7fffffffu;
                    hm = hi > hm ? hi : hm;
                    const unsigned hc = hi < 0x40340000u ? hi : 0x40340000u;
                    const double x = __hiloint2double((int)hc, __double2loint(f[t]));
                    const double jm = x + 26388279066624.0;
                    const double d = x - (jm - 26388279066624.0);
                    ii[t] += (unsigned)__double2loint(jm) << 3;
                    f[t] = d + a;
                    asm volatile("" : "+v"(f[t]), "+v"(ii[t]));
                }
            ii[0] += hm;
        }
        if (MODE == M_SPLIT_HALF || MODE == M_SPLIT_PARITY) {
            const int wv = threadIdx.x >> 6;
            const bool do_mfma = MODE == M_SPLIT_HALF ? (wv < 4) : ((wv & 1) == 0);
            if (do_mfma) {
#pragma unroll
                for (int s = 0; s < 16; ++s)
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, f[s], acc[t], 0, 0, 0);
            } else {
#pragma unroll
                for (int s = 0; s < 64; ++s)
#pragma unroll
                    for (int t = 0; t < 16; ++t) f[t] = fma(f[t], a, b);
            }
        }
        if (MODE == M_DSREAD) {
#pragma unroll
            for (int s = 0; s < 16; ++s)
#pragma unroll
                for (int t = 0; t < 16; ++t) { asm volatile("" : "+v"(ii[t])); f[t] = lds[(ii[t] + s * 64) & 4095]; }
            for (int t = 0; t < 16; ++t) ii[t] += (int)f[t];
        }
    }
    const long long t1 = clock64();
    double r = 0;
    for (int i = 0; i < 4; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 16; ++i) r += f[i] + ii[i];
    out[(long)blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (lane == 0) cyc[(long)blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = t1 - t0;
}

template <int MODE> int run(int block, int grid, int iters, double* out, long long* cyc) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(block), 0, 0, out, cyc, 2, 1.0000001);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(block), 0, 0, out, cyc, iters, 1.0000001);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const int nw = grid * block / 64;
    std::vector<long long> h(nw);
    CHECK(hipMemcpy(h.data(), cyc, nw * sizeof(long long), hipMemcpyDeviceToHost));
    double avg = 0;
    for (auto v : h) avg += (double)v;
    avg /= nw;
    printf("%-32s block %3d (waves/SIMD %d): %9.1f clk/iter/wave   %8.3f ms  (%.1f ns/iter)\n", names[MODE], block, block / 256,
           avg / iters, ms, ms * 1e6 / iters);
    return 0;
}

template <int MODE> int both(double* out, long long* cyc) {
    if (run<MODE>(256, 256, 2000, out, cyc)) return 1;
    if (run<MODE>(512, 256, 2000, out, cyc)) return 1;
    return 0;
}

int main() {
    double* out; long long* cyc;
    CHECK(hipMalloc(&out, 256 * 1024 * sizeof(double)));
    CHECK(hipMalloc(&cyc, 256 * 16 * sizeof(long long)));
    int rc = 0;
    rc |= both<M_MFMA>(out, cyc);
    rc |= both<M_FMA>(out, cyc);
    rc |= both<M_PHASED>(out, cyc);
    rc |= both<M_INTERLEAVED>(out, cyc);
    rc |= both<M_RCP>(out, cyc);
    rc |= both<M_LDEXP>(out, cyc);
    rc |= both<M_RNDNE>(out, cyc);
    rc |= both<M_CVT>(out, cyc);
    rc |= both<M_MUL>(out, cyc);
    rc |= both<M_ADD>(out, cyc);
    rc |= both<M_CNDMASK>(out, cyc);
    rc |= both<M_DSREAD>(out, cyc);
    rc |= both<M_FMA_DEP>(out, cyc);
    rc |= both<M_SELECT64>(out, cyc);
    rc |= both<M_DIV>(out, cyc);
    rc |= both<M_AND>(out, cyc);
    rc |= both<M_MIN64>(out, cyc);
    rc |= both<M_CMP64>(out, cyc);
    rc |= both<M_SELECT32>(out, cyc);
    rc |= both<M_MAX3>(out, cyc);
    rc |= both<M_IDX_FP>(out, cyc);
    rc |= both<M_IDX_INT>(out, cyc);
    rc |= both<M_SPLIT_HALF>(out, cyc);
    rc |= both<M_SPLIT_PARITY>(out, cyc);
    return rc;
}
