// Counter-based RNG for on-device domain randomization and init-state sampling: Philox4x32-10
// (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11).  The value drawn for
// (seed, global env id, row, stream) is independent of launch geometry and GPU count.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

struct Philox4 { uint32_t v[4]; };

__device__ __forceinline__ Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        // each 32 x 32 -> 64-bit product written as one: it compiles to ONE v_mad_u64_u32 (as __umulhi and a 32-bit
        // product it became a v_mul_hi_u32 and a v_mul_lo_u32; the integer multiplies run at a quarter of the vector rate
        // and are most of the cost of a draw)
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    Philox4 o;
    o.v[0] = c0; o.v[1] = c1; o.v[2] = c2; o.v[3] = c3;
    return o;
}

// 53-bit uniform in [0, 1)
__device__ __forceinline__ double psh_u01(uint32_t hi, uint32_t lo) {
    return ((double)(hi >> 5) * 67108864.0 + (double)(lo >> 6)) * (1.0 / 9007199254740992.0);
}

// Stream ids (counter word 3).  A sampler that draws several pairs per (env, row) uses stream + 16 p for pair p, so the
// ids below 16 are disjoint for every p.  Callers may hand the same seed and env offset to all samplers.
#define PSH_STREAM_RANDOMIZE 0u   // domain parameters, row = row of the table
#define PSH_STREAM_OBS_NOISE 1u   // GaussianObsNoiseWrapper, row = step
#define PSH_STREAM_ACT_NOISE 2u   // GaussianActNoiseWrapper, row = step
#define PSH_STREAM_EXPL 3u        // action-noise exploration, row = step
#define PSH_STREAM_BERNOULLI 4u   // stochastic reset, row = step
#define PSH_STREAM_INIT 5u        // initial states, row = component
#define PSH_STREAM_BOOTSTRAP 6u   // bootstrap resampling, env = replication, row = draw pair

// two uniforms for (seed, env, row, stream)
__device__ __forceinline__ void psh_draw2(uint64_t seed, uint64_t env, uint32_t row, uint32_t stream, double& ua, double& ub) {
    const Philox4 o = philox4x32_10((uint32_t)env, (uint32_t)(env >> 32), row, stream, (uint32_t)seed, (uint32_t)(seed >> 32));
    ua = psh_u01(o.v[0], o.v[1]);
    ub = psh_u01(o.v[2], o.v[3]);
}
