// libpysimhip.so — kernels and C ABI (include/pysimhip.h) of the MI355X batched rollout engine.
//
// Kernels (all gfx950, wave64):
//   k_rollout_linear<KIND, PER_ENV, WRAP>  (psh_rollout.h) fused reset + T x (LinearPolicy forward, SimPyEnv.step); one lane per env;
//                                    WRAP adds the fused environment wrappers (action/observation normalisation,
//                                    delay, noise, partial observation)
//   k_rollout_fnn<KIND, HP, FINE, WRAP, POP>  (psh_rollout.h) same with an FNNPolicy: the hidden layers run on v_mfma_f64_16x16x4_f64 with
//                                    the weights as the A operand and 16 environments as the columns of B, so a
//                                    layer's accumulator is directly the next layer's B operand
//   k_reset / k_step                 SimPyEnv.reset / SimPyEnv.step for caller-supplied actions
//   k_randomize / k_sample_init      DomainRandomizer.randomize / init_space.sample_uniform with Philox4x32-10
//   k_qbb_ik, k_segment_mean, k_pack_fnn
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>

#include "psh_rollout.h"

// instantiated in rollout_kind.hip, one translation unit per kind
extern template int launch_rollout<PSH_ENV_QQ>(psh_ctx*, const psh_rollout_desc*, RolloutArgs&, hipStream_t);
extern template int launch_rollout<PSH_ENV_QCP_SU>(psh_ctx*, const psh_rollout_desc*, RolloutArgs&, hipStream_t);
extern template int launch_rollout<PSH_ENV_QCP_ST>(psh_ctx*, const psh_rollout_desc*, RolloutArgs&, hipStream_t);
extern template int launch_rollout<PSH_ENV_QBB>(psh_ctx*, const psh_rollout_desc*, RolloutArgs&, hipStream_t);
extern template int launch_rollout<PSH_ENV_BOB>(psh_ctx*, const psh_rollout_desc*, RolloutArgs&, hipStream_t);
extern template int launch_rollout<PSH_ENV_OMO>(psh_ctx*, const psh_rollout_desc*, RolloutArgs&, hipStream_t);
extern template int launch_rollout<PSH_ENV_PEND>(psh_ctx*, const psh_rollout_desc*, RolloutArgs&, hipStream_t);
extern template int launch_rollout<PSH_ENV_QQ_ST>(psh_ctx*, const psh_rollout_desc*, RolloutArgs&, hipStream_t);

// Scratch of psh_rollout_static, the entry point that allocates nothing: work-queue counter, packed policy weights
// and QBallBalancer plate angles live in this array instead of context-owned allocations.  Every GPU the library runs
// on holds its own copy for the life of the process.  Host code never uses the array's name as an address: psh_create
// asks the runtime for the device address once per device (static_scratch_of) and everything else goes through that.
static __device__ __align__(256) unsigned char scratch[PSH_STATIC_BYTES];

static std::mutex g_scratch_lock;
static unsigned char* g_scratch_ptr[64];  // per HIP device, filled with that device current
// The array is shared by every context on a device, so the calls that use it are ordered per DEVICE: each waits for the
// event recorded behind the previous one, whatever context or stream issued that.
static hipEvent_t g_static_done[64];
static bool g_static_have_done[64];

static int static_scratch_of(psh_ctx* c) {
    if (c->device < 0 || c->device >= 64) return PSH_ERR_NODEV;
    std::lock_guard<std::mutex> guard(g_scratch_lock);
    if (!g_scratch_ptr[c->device]) {
        void* p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)) != hipSuccess || !p) return PSH_ERR_HIP;
        if (hipEventCreateWithFlags(&g_static_done[c->device], hipEventDisableTiming) != hipSuccess) return PSH_ERR_HIP;
        g_scratch_ptr[c->device] = static_cast<unsigned char*>(p);
    }
    c->static_scratch = g_scratch_ptr[c->device];
    return PSH_OK;
}

// QBallBalancerSim.reset: plate angles of every env by inverse kinematics, ahead of the rollout kernel, so that a lane
// taking over a new env only reads two numbers instead of running 200 optimiser steps while its wave waits.
__global__ __launch_bounds__(256) void k_qbb_plate(const RolloutArgs a, double* plate) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n_envs) return;
    bool bad_index;  // flagged by the rollout kernel
    const long col = dp_column(a.dp_index, a.ld_dp, i, &bad_index);
    const DpCol p{a.domain_param, a.ld_dp, col};
    const double th_x = (a.init_dim == 8) ? a.init_state[0 * a.ld_init + i] : 0.;
    const double th_y = (a.init_dim == 8) ? a.init_state[1 * a.ld_init + i] : 0.;
    plate[i] = qbb_ik(th_x + p[18], p[4], p[3]);
    plate[(long)a.n_envs + i] = qbb_ik(th_y + p[19], p[4], p[3]);
}

// --------------------------------------------------------------------------------------------- reset / step
struct StepArgs {
    int n_envs, max_steps, flags, init_dim;
    double dt;
    const double* domain_param; const int* dp_index; long ld_dp;
    const double* init_state;
    psh_task task;
    double* state; int* curr_step; int* flags_io; double* act; double* obs; double* rew; int* done; int* status;
};

template <int KIND> __global__ __launch_bounds__(256) void k_reset(const StepArgs a) {
    using E = Env<KIND>;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n_envs) return;
    const long N = a.n_envs;
    E e;
    bool bad_index;
    e.setup(DpCol{a.domain_param, a.ld_dp, dp_column(a.dp_index, a.ld_dp, i, &bad_index)});
    double init[E::S];
    for (int r = 0; r < E::S; ++r) init[r] = (r < a.init_dim) ? a.init_state[r * N + i] : 0.;
    e.reset(init, a.init_dim, (a.flags & PSH_F_SIMPLIFIED_DYN) != 0);
#pragma unroll
    for (int r = 0; r < E::S; ++r) a.state[r * N + i] = e.s[r];
    if constexpr (KIND == PSH_ENV_QBB) {
        a.state[8 * N + i] = e.plate[0];
        a.state[9 * N + i] = e.plate[1];
    }
    a.curr_step[i] = 0;
    a.flags_io[i] = 0;
    double obs[E::O];
    e.observe(obs);
    bool nan = false;
#pragma unroll
    for (int r = 0; r < E::O; ++r) { a.obs[r * N + i] = obs[r]; nan = nan || (obs[r] != obs[r]); }
    if (a.status) a.status[i] = (nan ? PSH_ST_NAN : 0) | (bad_index ? PSH_ST_BAD_INDEX : 0);
}

template <int KIND> __global__ __launch_bounds__(256) void k_step(const StepArgs a) {
    using E = Env<KIND>;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n_envs) return;
    const long N = a.n_envs;
    E e;
    bool bad_index;
    e.setup(DpCol{a.domain_param, a.ld_dp, dp_column(a.dp_index, a.ld_dp, i, &bad_index)});
    const double c_max = task_c_max<KIND>(e, a.task);
#pragma unroll
    for (int r = 0; r < E::S; ++r) e.s[r] = a.state[r * N + i];
    if constexpr (KIND == PSH_ENV_QBB) {
        e.plate[0] = a.state[8 * N + i];
        e.plate[1] = a.state[9 * N + i];
    }
    e.refresh();
    int curr_step = a.curr_step[i], yielded = a.flags_io[i] & 1, st = bad_index ? PSH_ST_BAD_INDEX : 0;
    double act[E::A];
#pragma unroll
    for (int r = 0; r < E::A; ++r) { act[r] = a.act[r * N + i]; if (act[r] != act[r]) st |= PSH_ST_NAN; }
    bool done = false;
    const double rew = env_step<KIND>(e, a.task, c_max, curr_step, yielded, act, a.dt, a.max_steps,
                                      (a.flags & PSH_F_MAX_STEPS_INF) != 0, (a.flags & PSH_F_SIMPLIFIED_DYN) != 0, done, st);
#pragma unroll
    for (int r = 0; r < E::A; ++r) a.act[r * N + i] = act[r];
#pragma unroll
    for (int r = 0; r < E::S; ++r) a.state[r * N + i] = e.s[r];
    if constexpr (KIND == PSH_ENV_QBB) {
        a.state[8 * N + i] = e.plate[0];
        a.state[9 * N + i] = e.plate[1];
    }
    a.curr_step[i] = curr_step;
    a.flags_io[i] = yielded;
    double obs[E::O];
    e.observe(obs);
#pragma unroll
    for (int r = 0; r < E::O; ++r) a.obs[r * N + i] = obs[r];
    a.rew[i] = rew;
    a.done[i] = done ? 1 : 0;
    if (a.status) a.status[i] = st;
}

// ------------------------------------------------------------------------------- randomization / init sampling
struct RandArgs {
    int n_envs, n_rows, n_distr, ld_out;
    uint64_t seed, env_offset;
    psh_distr distr[PSH_MAX_DISTR];
    double nominal[PSH_MAX_DP + 4];
    double* out;
};

__global__ __launch_bounds__(256) void k_randomize(const RandArgs a) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n_envs) return;
    for (int r = 0; r < a.n_rows; ++r) a.out[(long)r * a.ld_out + i] = a.nominal[r];
    for (int k = 0; k < a.n_distr; ++k) {
        const psh_distr& d = a.distr[k];
        double ua, ub, x = d.a;
        psh_draw2(a.seed, a.env_offset + (uint64_t)i, (uint32_t)d.row, PSH_STREAM_RANDOMIZE, ua, ub);
        if (d.kind == PSH_DISTR_NORMAL) {
            const double r = sqrt(-2.0 * log(1.0 - ua));
            x = d.a + d.b * (r * cos(2.0 * PSH_PI * ub));
        } else if (d.kind == PSH_DISTR_UNIFORM) {
            x = (d.a - d.b) + ua * ((d.a + d.b) - (d.a - d.b));
        } else if (d.kind == PSH_DISTR_BERNOULLI) {
            x = (ua < d.c) ? d.b : d.a;
        }
        if (d.kind != PSH_DISTR_CONST) {
            x = fmin(fmax(x, d.clip_lo), d.clip_up);
            if (d.roundint) x = rint(x);
        }
        a.out[(long)d.row * a.ld_out + i] = x;
    }
}

struct InitArgs {
    int env_kind, n_envs; long ld_dp, ld_out;
    uint64_t seed, env_offset;
    const double* domain_param; const int* dp_index; double* out;
};

__global__ __launch_bounds__(256) void k_sample_init(const InitArgs a) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n_envs) return;
    bool bad_index;  // no status word here: such an env gets a NaN init state, which its rollout reports
    const long col = dp_column(a.dp_index, a.ld_dp, i, &bad_index);
    double lo[4], hi[4], u[5], ub;
    const uint64_t env = a.env_offset + (uint64_t)i;
    for (int r = 0; r < 5; ++r) psh_draw2(a.seed, env, (uint32_t)r, PSH_STREAM_INIT, u[r], ub);
    int I = 4;
    switch (a.env_kind) {
    case PSH_ENV_QQ: {
        const double m[4] = {2. / 180 * PSH_PI, 1. / 180 * PSH_PI, 0.5 / 180 * PSH_PI, 0.5 / 180 * PSH_PI};
        for (int r = 0; r < 4; ++r) { lo[r] = -m[r]; hi[r] = m[r]; }
        break;
    }
    case PSH_ENV_QCP_SU: {
        const double m[4] = {0.03, 1 / 180. * PSH_PI, 0.005, 2 / 180. * PSH_PI};
        for (int r = 0; r < 4; ++r) { lo[r] = -m[r]; hi[r] = m[r]; }
        break;
    }
    case PSH_ENV_QCP_ST: {
        const double off = 8 / 180. * PSH_PI;
        lo[0] = -0.05; hi[0] = 0.05; lo[1] = PSH_PI - off; hi[1] = PSH_PI + off;
        lo[2] = -0.05; hi[2] = 0.05; lo[3] = -8 / 180. * PSH_PI; hi[3] = 8 / 180. * PSH_PI;
        break;
    }
    case PSH_ENV_QBB: {
        const double l_plate = a.domain_param[3 * a.ld_dp + col];
        lo[0] = 0.75 * l_plate / 2; hi[0] = 0.8 * l_plate / 2; lo[1] = -PSH_PI; hi[1] = PSH_PI;
        lo[2] = -0.05 * 0.5; hi[2] = 0.05 * 0.5; lo[3] = -0.05 * 0.5; hi[3] = 0.05 * 0.5;
        break;
    }
    case PSH_ENV_BOB: {
        const double l_beam = a.domain_param[4 * a.ld_dp + col];
        const bool right = u[4] >= 0.5;
        lo[0] = right ? 0.7 * l_beam / 2. : -0.8 * l_beam / 2.;
        hi[0] = right ? 0.8 * l_beam / 2. : -0.7 * l_beam / 2.;
        lo[1] = -5 / 180. * PSH_PI; hi[1] = 5 / 180. * PSH_PI;
        lo[2] = -0.02 * 10.; hi[2] = 0.02 * 10.; lo[3] = -0.02 * PSH_PI; hi[3] = 0.02 * PSH_PI;
        break;
    }
    case PSH_ENV_PEND:  // SingularStateSpace(zeros), pendulum.py:23,27
        I = 2;
        for (int r = 0; r < 4; ++r) lo[r] = hi[r] = 0.;
        break;
    case PSH_ENV_QQ_ST:  // quanser_qube.py:244-245
        lo[0] = -5. / 180 * PSH_PI; hi[0] = 5. / 180 * PSH_PI; lo[1] = 175. / 180 * PSH_PI; hi[1] = 185. / 180 * PSH_PI;
        lo[2] = hi[2] = lo[3] = hi[3] = 0.;
        break;
    default:  // OMO
        I = 2;
        lo[0] = -0.75 * 1.; hi[0] = -0.65 * 1.; lo[1] = -0.01 * 10.; hi[1] = 0.01 * 10.;
        lo[2] = hi[2] = lo[3] = hi[3] = 0.;
    }
    double v[4];
    for (int r = 0; r < 4; ++r) v[r] = lo[r] + u[r] * (hi[r] - lo[r]);
    if (a.env_kind == PSH_ENV_QBB) {
        const double rad = v[0], phi = v[1];
        double sn, cs;
        sincos(phi, &sn, &cs);
        v[0] = rad * cs; v[1] = rad * sn;
    }
    for (int r = 0; r < I; ++r) a.out[(long)r * a.ld_out + i] = bad_index ? __builtin_nan("") : v[r];
}

__global__ __launch_bounds__(256) void k_qbb_ik(int n, const double* th, const double* r_arm, const double* l_plate, double* ang) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) ang[i] = qbb_ik(th[i], r_arm[i], l_plate[i]);
}

// one wave per group: mean of group_size consecutive returns
__global__ __launch_bounds__(256) void k_segment_mean(int n_groups, int group_size, const double* ret, double* mean) {
    const int g = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (g >= n_groups) return;
    double acc = 0.;
    for (int j = lane; j < group_size; j += 64) acc += ret[(long)g * group_size + j];
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) mean[g] = acc / group_size;
}

// Descending rank of every value by counting the values ahead of it (greater, or equal with a smaller index; a NaN is
// ahead of every number).  n threads x n comparisons with the values staged through LDS in tiles, i.e. O(n^2) work: 10^7
// comparisons (tens of microseconds) for the 4 096 per-policy means of a CEM / NES generation, 10^9 (milliseconds) at
// n = 3 x 10^4, which is about where a sort would take over; the order is deterministic.
__global__ __launch_bounds__(256) void k_topk(int n, int k, const double* __restrict__ v, int* __restrict__ idx_out, double* __restrict__ val_out) {
    __shared__ double tile[256];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const double mine = i < n ? v[i] : 0.;
    const bool mine_nan = mine != mine;
    int rank = 0;
    for (int base = 0; base < n; base += 256) {
        __syncthreads();
        tile[threadIdx.x] = base + (int)threadIdx.x < n ? v[base + threadIdx.x] : 0.;
        __syncthreads();
        const int m = n - base < 256 ? n - base : 256;
        for (int j = 0; j < m; ++j) {
            const double o = tile[j];
            const bool o_nan = o != o;
            const bool ahead = o_nan ? (!mine_nan || base + j < i) : (!mine_nan && (o > mine || (o == mine && base + j < i)));
            rank += ahead ? 1 : 0;
        }
    }
    if (i < n && rank < k) {
        idx_out[rank] = i;
        if (val_out) val_out[rank] = mine;
    }
}

// one wave per bootstrap replication: n draws with replacement, summed over the lanes
__global__ __launch_bounds__(256) void k_bootstrap_means(int n, int num_reps, const double* __restrict__ data, uint64_t seed, double* __restrict__ out) {
    const int rep = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (rep >= num_reps) return;
    double acc = 0.;
    for (int p = lane; 2 * p < n; p += 64) {
        double u0, u1;
        psh_draw2(seed, (uint64_t)rep, (uint32_t)p, PSH_STREAM_BOOTSTRAP, u0, u1);
        int j0 = (int)(u0 * n), j1 = (int)(u1 * n);
        j0 = j0 < n ? j0 : n - 1; j1 = j1 < n ? j1 : n - 1;
        acc += data[j0];
        if (2 * p + 1 < n) acc += data[j1];
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) out[rep] = acc / n;
}

// element-wise evaluation of the device math functions (accuracy tests)
__global__ __launch_bounds__(256) void k_eval_math(int fn, int n, const double* x, const double* m, double* y, const double* tanh_tab) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double s, c;
    switch (fn) {
    case 0: y[i] = psh_exp(x[i]); break;
    case 1: y[i] = psh_tanh(x[i]); break;
    case 2: y[i] = psh_sigmoid(x[i]); break;
    case 3: psh_sincos(x[i], &s, &c); y[i] = s; break;
    case 4: psh_sincos(x[i], &s, &c); y[i] = c; break;
    case 6: y[i] = psh_tanh8(x[i], tanh_tab + (PSH_TANH8_N - 1)); break;
    case 7: {
        unsigned hi;
        const double r = psh_tanh_tab(x[i], tanh_tab + PSH_TANH_ODD_N, &hi);
        y[i] = hi > PSH_HI_INF ? x[i] : r;
        break;
    }
    case 8: y[i] = psh_log(x[i]); break;
    default: y[i] = psh_fmod(x[i], m[i]);
    }
}

// =================================================================================================== C ABI
static bool dims_of(int kind, int32_t d[5]) {
    switch (kind) {
    case PSH_ENV_QQ: d[0] = 4; d[1] = 6; d[2] = 1; d[3] = 10; d[4] = 4; return true;
    case PSH_ENV_QCP_SU:
    case PSH_ENV_QCP_ST: d[0] = 4; d[1] = 5; d[2] = 1; d[3] = 14; d[4] = 4; return true;
    case PSH_ENV_QBB: d[0] = 8; d[1] = 8; d[2] = 2; d[3] = 20; d[4] = 4; return true;
    case PSH_ENV_BOB: d[0] = 4; d[1] = 4; d[2] = 1; d[3] = 8; d[4] = 4; return true;
    case PSH_ENV_OMO: d[0] = 2; d[1] = 2; d[2] = 1; d[3] = 3; d[4] = 2; return true;
    case PSH_ENV_PEND: d[0] = 2; d[1] = 3; d[2] = 1; d[3] = 5; d[4] = 2; return true;
    case PSH_ENV_QQ_ST: d[0] = 4; d[1] = 4; d[2] = 1; d[3] = 10; d[4] = 4; return true;
    }
    return false;
}

extern "C" int psh_version(void) { return PSH_VERSION; }

extern "C" int psh_env_dims(int env_kind, int32_t dims[5]) { return dims_of(env_kind, dims) ? PSH_OK : PSH_ERR_ARG; }

extern "C" int psh_create(int hip_device, psh_ctx** out) {
    if (!out) return PSH_ERR_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || hip_device < 0 || hip_device >= n) return PSH_ERR_NODEV;
    if (hipSetDevice(hip_device) != hipSuccess) return PSH_ERR_NODEV;
    psh_ctx* c = new (std::nothrow) psh_ctx();
    if (!c) return PSH_ERR_ARG;
    memset(c, 0, sizeof *c);
    c->device = hip_device;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, hip_device) != hipSuccess) { delete c; return PSH_ERR_HIP; }
    c->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 1;
    // the LDS one workgroup may claim through hipFuncAttributeMaxDynamicSharedMemorySize: all of a CU's (160 KB on gfx950)
    c->lds_per_block = prop.maxSharedMemoryPerMultiProcessor > prop.sharedMemPerBlock ? prop.maxSharedMemoryPerMultiProcessor
                                                                                      : prop.sharedMemPerBlock;
    constexpr size_t tab_n = PSH_TANH_ODD_N + PSH_TANH_TAB_N;
    std::unique_ptr<double[]> tab(new (std::nothrow) double[tab_n]);
    if (!tab) { delete c; return PSH_ERR_ARG; }
    // the fine table of psh_tanh8 is the upper half of the odd one: node 0 is entry PSH_TANH8_N - 1
    for (int j = 0; j < PSH_TANH8_N; ++j) {
        const double t = (double)tanhl((long double)j / 256.0L);
        tab[PSH_TANH8_N - 1 + j] = t;
        tab[PSH_TANH8_N - 1 - j] = -t;
    }
    for (int j = 0; j < PSH_TANH_TAB_N; ++j) tab[PSH_TANH_ODD_N + j] = (double)tanhl((long double)j / 64.0L);
    const size_t tab_bytes = tab_n * sizeof(double);
    bool events_ok = true;
    for (int j = 0; j < PSH_TIMING_RING; ++j)
        events_ok = events_ok && hipEventCreate(&c->tev0[j]) == hipSuccess && hipEventCreate(&c->tev1[j]) == hipSuccess;
    if (!events_ok ||
        hipEventCreateWithFlags(&c->ev_done, hipEventDisableTiming) != hipSuccess || static_scratch_of(c) != PSH_OK ||
        hipMalloc(&c->queue, 256) != hipSuccess || hipMalloc(&c->tanh_tab, tab_bytes) != hipSuccess ||
        hipMemcpy(c->tanh_tab, tab.get(), tab_bytes, hipMemcpyHostToDevice) != hipSuccess) {
        psh_destroy(c);  // releases whatever was created before the failure
        return PSH_ERR_HIP;
    }
    *out = c;
    return PSH_OK;
}

extern "C" void psh_destroy(psh_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    if (c->wfrag) (void)hipFree(c->wfrag);
    if (c->plate) (void)hipFree(c->plate);
    if (c->delay_q) (void)hipFree(c->delay_q);
    if (c->queue) (void)hipFree(c->queue);
    if (c->tanh_tab) (void)hipFree(c->tanh_tab);
    for (int j = 0; j < PSH_TIMING_RING; ++j) {
        if (c->tev0[j]) (void)hipEventDestroy(c->tev0[j]);
        if (c->tev1[j]) (void)hipEventDestroy(c->tev1[j]);
    }
    if (c->ev_done) (void)hipEventDestroy(c->ev_done);
    delete c;
}

extern "C" const char* psh_last_error(const psh_ctx* c) { return c ? c->err : "no context"; }

static int check_task(psh_ctx* c, const psh_task& t, int S) {
    if (t.rew_kind < 0 || t.rew_kind > PSH_REW_SWINGUP) { set_err(c, "unknown reward kind"); return PSH_ERR_ARG; }
    if (t.rew_kind == PSH_REW_SWINGUP && (t.idx_x < 0 || t.idx_x >= S || t.idx_th < 0 || t.idx_th >= S)) {
        set_err(c, "swing-up reward indices outside the state");
        return PSH_ERR_ARG;
    }
    const int m = t.final_mode;
    if (m & PSH_FIN_ENABLED) {
        const bool ap = m & PSH_FIN_ALWAYS_POSITIVE, an = m & PSH_FIN_ALWAYS_NEGATIVE;
        const bool sd = m & PSH_FIN_STATE_DEPENDENT, td = m & PSH_FIN_TIME_DEPENDENT;
        // the combinations FinalRewTask.compute_final_rew implements (final_reward.py:87-183)
        const bool ok = (!ap && !an) || (ap && !an && !(sd && td)) || (!ap && an && !(sd && td));
        if (!ok) { set_err(c, "FinalRewMode combination not implemented by the reference"); return PSH_ERR_UNSUPPORTED; }
    }
    return PSH_OK;
}

static int rollout_issue(psh_ctx* c, const psh_rollout_desc* d, hipStream_t st) {
    int32_t dm[5];
    dims_of(d->env_kind, dm);
    RolloutArgs a;
    memset(&a, 0, sizeof a);
    a.n_envs = d->n_envs; a.max_steps = d->max_steps; a.flags = d->flags & PSH_F_PUBLIC_MASK; a.init_dim = d->init_dim; a.dt = d->dt;
    a.init_state = d->init_state; a.ld_init = d->ld_init;
    a.domain_param = d->domain_param; a.dp_index = d->dp_index; a.ld_dp = d->ld_dp;
    a.task = d->task; a.wrap = d->wrap; a.pol = d->policy; a.bernoulli = d->bernoulli_reset;
    a.ret = d->ret; a.length = d->length; a.status = d->status; a.final_state = d->final_state;
    a.obs = d->obs; a.act = d->act; a.rew = d->rew;
    a.queue = c->static_mode ? reinterpret_cast<unsigned int*>(c->static_scratch) : c->queue;
    PSH_HIP(c, hipMemsetAsync(a.queue, 0, PSH_STATIC_QUEUE_BYTES, st));
    if (d->env_kind == PSH_ENV_QBB && !(d->flags & PSH_F_SIMPLIFIED_DYN)) {
        double* plate = nullptr;
        const int rc = scratch_doubles(c, PSH_SCR_PLATE, 2 * (size_t)d->n_envs, &plate);
        if (rc != PSH_OK) return rc;
        hipLaunchKernelGGL(k_qbb_plate, dim3((d->n_envs + 255) / 256), dim3(256), 0, st, a, plate);
        PSH_HIP(c, hipGetLastError());
        a.plate = plate;
    }
    switch (d->env_kind) {
    case PSH_ENV_QQ: return launch_rollout<PSH_ENV_QQ>(c, d, a, st);
    case PSH_ENV_QCP_SU: return launch_rollout<PSH_ENV_QCP_SU>(c, d, a, st);
    case PSH_ENV_QCP_ST: return launch_rollout<PSH_ENV_QCP_ST>(c, d, a, st);
    case PSH_ENV_QBB: return launch_rollout<PSH_ENV_QBB>(c, d, a, st);
    case PSH_ENV_BOB: return launch_rollout<PSH_ENV_BOB>(c, d, a, st);
    case PSH_ENV_PEND: return launch_rollout<PSH_ENV_PEND>(c, d, a, st);
    case PSH_ENV_QQ_ST: return launch_rollout<PSH_ENV_QQ_ST>(c, d, a, st);
    default: return launch_rollout<PSH_ENV_OMO>(c, d, a, st);
    }
}

static int rollout_checked(psh_ctx* c, const psh_rollout_desc* d, bool use_static) {
    if (!c || !d) return PSH_ERR_ARG;
    int32_t dm[5];
    if (!dims_of(d->env_kind, dm)) { set_err(c, "unknown env_kind"); return PSH_ERR_ARG; }
    if (d->n_envs < 0 || d->max_steps < 1 || !(d->dt > 0.)) { set_err(c, "n_envs >= 0, max_steps >= 1 and dt > 0 are required"); return PSH_ERR_ARG; }
    if (d->n_envs == 0) return PSH_OK;
    if (!d->init_state || !d->domain_param || !d->ret || !d->length || !d->status || !d->policy.params) {
        set_err(c, "init_state, domain_param, policy.params, ret, length and status must be non-NULL");
        return PSH_ERR_ARG;
    }
    const bool full_state = d->init_dim == dm[0];
    if (!(d->init_dim == dm[4] || full_state)) { set_err(c, "init_dim must be the init-space or the state dimension"); return PSH_ERR_ARG; }
    if (d->ld_init < d->n_envs || d->ld_dp < 1 || (!d->dp_index && d->ld_dp < d->n_envs)) { set_err(c, "leading dimensions too small"); return PSH_ERR_ARG; }
    if (d->policy.n_policies < 1) { set_err(c, "n_policies must be >= 1"); return PSH_ERR_ARG; }
    const int dp_rows = d->dp_rows > 0 ? d->dp_rows : dm[3];
    if (dp_rows < dm[3]) { set_err(c, "dp_rows is smaller than the simulation's number of domain parameters"); return PSH_ERR_ARG; }
    for (int j = 0; j < d->wrap.n_ops && j < PSH_MAX_WRAP; ++j) {
        if (d->wrap.ops[j] == PSH_WRAP_ACT_DELAY && d->wrap.act_delay_row >= dp_rows) {
            set_err(c, "act_delay_row outside the domain-parameter table (dp_rows)");
            return PSH_ERR_ARG;
        }
    }
    int rc = check_task(c, d->task, dm[0]);
    if (rc != PSH_OK) return rc;
    PSH_HIP(c, hipSetDevice(c->device));
    hipStream_t st = (hipStream_t)d->stream;
    // A stream that is being captured into a graph takes the kernels and the memset only: events recorded outside the
    // capture cannot be waited for inside it, and events recorded inside it cannot be synchronised with afterwards.  The
    // caller then orders the replays against other rollouts that use the same scratch.
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (st && hipStreamIsCapturing(st, &cap) != hipSuccess) cap = hipStreamCaptureStatusNone;
    const bool capturing = cap != hipStreamCaptureStatusNone;
    // one rollout per scratch at a time: a launch on another stream waits for the last one that used the same scratch --
    // the context's, or (psh_rollout_static) the device's static array, which all contexts of a device share
    hipEvent_t done = c->ev_done;
    bool have = c->have_done != 0;
    if (use_static) {
        std::lock_guard<std::mutex> guard(g_scratch_lock);
        done = g_static_done[c->device];
        have = g_static_have_done[c->device];
    }
    if (have && !capturing) PSH_HIP(c, hipStreamWaitEvent(st, done, 0));
    c->static_mode = use_static ? 1 : 0;
    c->timed = capturing ? 0 : 1;
    const int slot = (int)(c->n_timed % PSH_TIMING_RING);
    c->ev0 = c->tev0[slot];
    c->ev1 = c->tev1[slot];
    rc = rollout_issue(c, d, st);
    if (rc == PSH_OK && c->timed) c->n_timed += 1;
    c->static_mode = 0;
    // recorded also after a failure: whatever was enqueued before it (memset, prologue) may still touch the scratch
    if (!capturing && hipEventRecord(done, st) == hipSuccess) {
        if (use_static) {
            std::lock_guard<std::mutex> guard(g_scratch_lock);
            g_static_have_done[c->device] = true;
        } else {
            c->have_done = 1;
        }
    }
    return rc;
}

extern "C" int psh_rollout(psh_ctx* c, const psh_rollout_desc* d) { return rollout_checked(c, d, false); }

extern "C" int psh_rollout_static(psh_ctx* c, const psh_rollout_desc* d) { return rollout_checked(c, d, true); }

extern "C" float psh_rollout_ms(psh_ctx* c, int32_t back) {
    if (!c || back < 0 || back >= PSH_TIMING_RING || (long)back >= c->n_timed) return -1.f;
    const int slot = (int)((c->n_timed - 1 - back) % PSH_TIMING_RING);
    if (hipEventSynchronize(c->tev1[slot]) != hipSuccess) return -1.f;
    float ms = -1.f;
    if (hipEventElapsedTime(&ms, c->tev0[slot], c->tev1[slot]) != hipSuccess) return -1.f;
    return ms;
}

extern "C" float psh_last_rollout_ms(psh_ctx* c) { return psh_rollout_ms(c, 0); }

extern "C" int64_t psh_alloc_count(const psh_ctx* c) { return c ? (int64_t)c->n_allocs : -1; }

static int fill_step_args(psh_ctx* c, const psh_step_desc* d, StepArgs& a, int32_t dm[5], bool is_reset) {
    if (!c || !d) return PSH_ERR_ARG;
    if (!dims_of(d->env_kind, dm)) { set_err(c, "unknown env_kind"); return PSH_ERR_ARG; }
    if (d->n_envs < 0) { set_err(c, "n_envs must be >= 0"); return PSH_ERR_ARG; }
    if (!d->domain_param || !d->state || !d->curr_step || !d->flags_io || !d->obs) { set_err(c, "domain_param, state, curr_step, flags_io and obs must be non-NULL"); return PSH_ERR_ARG; }
    if (is_reset) {
        if (!d->init_state || !(d->init_dim == dm[4] || d->init_dim == dm[0])) { set_err(c, "psh_reset needs init_state with the init-space or the state dimension"); return PSH_ERR_ARG; }
    } else {
        if (!d->act || !d->rew || !d->done || !(d->dt > 0.) || d->max_steps < 1) { set_err(c, "psh_step needs act, rew, done, dt > 0 and max_steps >= 1"); return PSH_ERR_ARG; }
    }
    if (d->ld_dp < 1 || (!d->dp_index && d->ld_dp < d->n_envs)) { set_err(c, "ld_dp too small"); return PSH_ERR_ARG; }
    int rc = check_task(c, d->task, dm[0]);
    if (rc != PSH_OK) return rc;
    memset(&a, 0, sizeof a);
    a.n_envs = d->n_envs; a.max_steps = d->max_steps; a.flags = d->flags; a.init_dim = d->init_dim; a.dt = d->dt;
    a.domain_param = d->domain_param; a.dp_index = d->dp_index; a.ld_dp = d->ld_dp; a.init_state = d->init_state;
    a.task = d->task;
    a.state = d->state; a.curr_step = d->curr_step; a.flags_io = d->flags_io; a.act = d->act; a.obs = d->obs;
    a.rew = d->rew; a.done = d->done; a.status = d->status;
    return PSH_OK;
}

#define PSH_DISPATCH_KIND(kind, KERNEL, grid, block, st, args)                                         \
    switch (kind) {                                                                                    \
    case PSH_ENV_QQ: hipLaunchKernelGGL((KERNEL<PSH_ENV_QQ>), grid, block, 0, st, args); break;       \
    case PSH_ENV_QCP_SU: hipLaunchKernelGGL((KERNEL<PSH_ENV_QCP_SU>), grid, block, 0, st, args); break; \
    case PSH_ENV_QCP_ST: hipLaunchKernelGGL((KERNEL<PSH_ENV_QCP_ST>), grid, block, 0, st, args); break; \
    case PSH_ENV_QBB: hipLaunchKernelGGL((KERNEL<PSH_ENV_QBB>), grid, block, 0, st, args); break;     \
    case PSH_ENV_BOB: hipLaunchKernelGGL((KERNEL<PSH_ENV_BOB>), grid, block, 0, st, args); break;     \
    case PSH_ENV_PEND: hipLaunchKernelGGL((KERNEL<PSH_ENV_PEND>), grid, block, 0, st, args); break;   \
    case PSH_ENV_QQ_ST: hipLaunchKernelGGL((KERNEL<PSH_ENV_QQ_ST>), grid, block, 0, st, args); break; \
    default: hipLaunchKernelGGL((KERNEL<PSH_ENV_OMO>), grid, block, 0, st, args); break;              \
    }

extern "C" int psh_reset(psh_ctx* c, const psh_step_desc* d) {
    StepArgs a;
    int32_t dm[5];
    int rc = fill_step_args(c, d, a, dm, true);
    if (rc != PSH_OK || d->n_envs == 0) return rc;
    PSH_HIP(c, hipSetDevice(c->device));
    const dim3 block(256), grid((unsigned)((d->n_envs + 255) / 256));
    hipStream_t st = (hipStream_t)d->stream;
    PSH_DISPATCH_KIND(d->env_kind, k_reset, grid, block, st, a);
    PSH_HIP(c, hipGetLastError());
    return PSH_OK;
}

extern "C" int psh_step(psh_ctx* c, const psh_step_desc* d) {
    StepArgs a;
    int32_t dm[5];
    int rc = fill_step_args(c, d, a, dm, false);
    if (rc != PSH_OK || d->n_envs == 0) return rc;
    PSH_HIP(c, hipSetDevice(c->device));
    const dim3 block(256), grid((unsigned)((d->n_envs + 255) / 256));
    hipStream_t st = (hipStream_t)d->stream;
    PSH_DISPATCH_KIND(d->env_kind, k_step, grid, block, st, a);
    PSH_HIP(c, hipGetLastError());
    return PSH_OK;
}

extern "C" int psh_randomize(psh_ctx* c, const psh_rand_desc* d) {
    if (!c || !d) return PSH_ERR_ARG;
    if (d->n_envs < 0 || d->n_rows < 1 || d->n_rows > PSH_MAX_DP + 4 || d->n_distr < 0 || d->n_distr > PSH_MAX_DISTR || !d->out || d->ld_out < d->n_envs) {
        set_err(c, "invalid randomization descriptor");
        return PSH_ERR_ARG;
    }
    for (int k = 0; k < d->n_distr; ++k) {
        const psh_distr& q = d->distr[k];
        if (q.row < 0 || q.row >= d->n_rows || q.kind < 0 || q.kind > PSH_DISTR_BERNOULLI) { set_err(c, "invalid distribution entry"); return PSH_ERR_ARG; }
    }
    if (d->n_envs == 0) return PSH_OK;
    PSH_HIP(c, hipSetDevice(c->device));
    RandArgs a;
    a.n_envs = d->n_envs; a.n_rows = d->n_rows; a.n_distr = d->n_distr; a.ld_out = d->ld_out;
    a.seed = d->seed; a.env_offset = d->env_offset;
    memcpy(a.distr, d->distr, sizeof a.distr);
    memcpy(a.nominal, d->nominal, sizeof a.nominal);
    a.out = d->out;
    hipLaunchKernelGGL(k_randomize, dim3((d->n_envs + 255) / 256), dim3(256), 0, (hipStream_t)d->stream, a);
    PSH_HIP(c, hipGetLastError());
    return PSH_OK;
}

extern "C" int psh_sample_init(psh_ctx* c, const psh_init_desc* d) {
    if (!c || !d) return PSH_ERR_ARG;
    int32_t dm[5];
    if (!dims_of(d->env_kind, dm)) { set_err(c, "unknown env_kind"); return PSH_ERR_ARG; }
    if (d->n_envs < 0 || !d->out || d->ld_out < d->n_envs) { set_err(c, "invalid init-sampling descriptor"); return PSH_ERR_ARG; }
    const bool needs_dp = d->env_kind == PSH_ENV_QBB || d->env_kind == PSH_ENV_BOB;
    if (needs_dp && (!d->domain_param || d->ld_dp < 1 || (!d->dp_index && d->ld_dp < d->n_envs))) { set_err(c, "this init space depends on the domain parameters"); return PSH_ERR_ARG; }
    if (d->n_envs == 0) return PSH_OK;
    PSH_HIP(c, hipSetDevice(c->device));
    InitArgs a;
    a.env_kind = d->env_kind; a.n_envs = d->n_envs; a.ld_dp = d->ld_dp; a.ld_out = d->ld_out;
    a.seed = d->seed; a.env_offset = d->env_offset;
    a.domain_param = d->domain_param; a.dp_index = needs_dp ? d->dp_index : nullptr; a.out = d->out;
    hipLaunchKernelGGL(k_sample_init, dim3((d->n_envs + 255) / 256), dim3(256), 0, (hipStream_t)d->stream, a);
    PSH_HIP(c, hipGetLastError());
    return PSH_OK;
}

extern "C" int psh_qbb_ik(psh_ctx* c, int32_t n, const double* th, const double* r_arm, const double* l_plate, double* ang, void* stream) {
    if (!c || n < 0 || (n > 0 && (!th || !r_arm || !l_plate || !ang))) { set_err(c, "invalid IK arguments"); return PSH_ERR_ARG; }
    if (n == 0) return PSH_OK;
    PSH_HIP(c, hipSetDevice(c->device));
    hipLaunchKernelGGL(k_qbb_ik, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, n, th, r_arm, l_plate, ang);
    PSH_HIP(c, hipGetLastError());
    return PSH_OK;
}

extern "C" int psh_segment_mean(psh_ctx* c, int32_t n_groups, int32_t group_size, const double* ret, double* mean, void* stream) {
    if (!c || n_groups < 0 || group_size < 1 || (n_groups > 0 && (!ret || !mean))) { set_err(c, "invalid segment-mean arguments"); return PSH_ERR_ARG; }
    if (n_groups == 0) return PSH_OK;
    PSH_HIP(c, hipSetDevice(c->device));
    hipLaunchKernelGGL(k_segment_mean, dim3((n_groups + 3) / 4), dim3(256), 0, (hipStream_t)stream, n_groups, group_size, ret, mean);
    PSH_HIP(c, hipGetLastError());
    return PSH_OK;
}

extern "C" int psh_topk(psh_ctx* c, int32_t n, int32_t k, const double* values, int32_t* idx_out, double* val_out, void* stream) {
    if (!c || n < 0 || k < 0 || k > n || (k > 0 && (!values || !idx_out))) { set_err(c, "invalid top-k arguments (0 <= k <= n, non-NULL pointers)"); return PSH_ERR_ARG; }
    if (k == 0) return PSH_OK;
    PSH_HIP(c, hipSetDevice(c->device));
    hipLaunchKernelGGL(k_topk, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, (int)n, (int)k, values, idx_out, val_out);
    PSH_HIP(c, hipGetLastError());
    return PSH_OK;
}

extern "C" int psh_bootstrap_means(psh_ctx* c, int32_t n, int32_t num_reps, const double* data, uint64_t seed, double* stat_bs, void* stream) {
    if (!c || n < 1 || num_reps < 0 || !data || (num_reps > 0 && !stat_bs)) { set_err(c, "invalid bootstrap arguments (n >= 1, non-NULL pointers)"); return PSH_ERR_ARG; }
    if (num_reps == 0) return PSH_OK;
    PSH_HIP(c, hipSetDevice(c->device));
    hipLaunchKernelGGL(k_bootstrap_means, dim3((num_reps + 3) / 4), dim3(256), 0, (hipStream_t)stream, (int)n, (int)num_reps, data, seed, stat_bs);
    PSH_HIP(c, hipGetLastError());
    return PSH_OK;
}

extern "C" int psh_eval_math(psh_ctx* c, int32_t fn, int32_t n, const double* x, const double* m, double* y, void* stream) {
    if (!c || fn < 0 || fn > 8 || n < 0 || (n > 0 && (!x || !y || (fn == 5 && !m)))) { set_err(c, "invalid math-evaluation arguments"); return PSH_ERR_ARG; }
    if (n == 0) return PSH_OK;
    PSH_HIP(c, hipSetDevice(c->device));
    hipLaunchKernelGGL(k_eval_math, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, (int)fn, (int)n, x, m, y, (const double*)c->tanh_tab);
    PSH_HIP(c, hipGetLastError());
    return PSH_OK;
}
