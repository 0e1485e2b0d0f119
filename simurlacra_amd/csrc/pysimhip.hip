// libpysimhip.so — kernels and C ABI (include/pysimhip.h) of the MI355X batched rollout engine.
//
// Kernels (all gfx950, wave64):
//   k_rollout_linear<KIND, PER_ENV, WRAP>  fused reset + T x (LinearPolicy forward, SimPyEnv.step); one lane per env;
//                                    WRAP adds the fused environment wrappers (action/observation normalisation,
//                                    delay, noise, partial observation)
//   k_rollout_fnn<KIND, HP, FINE, WRAP>  same with an FNNPolicy: the hidden layers run on v_mfma_f64_16x16x4_f64 with
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
#include <new>

#include "psh_env.h"
#include "psh_rng.h"

// ------------------------------------------------------------------------------------------------ context
struct psh_ctx {
    int device;
    char err[512];
    double* wfrag;       // packed FNN weights (fragment layout), grown on demand
    size_t wfrag_doubles;
    double* plate;       // QBB plate angles of the IK prologue [2, N], grown on demand
    size_t plate_doubles;
    double* delay_q;     // ActDelay ring buffers of the resident lanes, grown on demand
    size_t delay_q_doubles;
    unsigned int* queue; // work-queue counter of the rollout kernels
    double* tanh_tab;    // [tanh(j/256), j = 0..5120 (psh_tanh8)][tanh(j/64), j = 0..1280 (psh_tanh_tab)]
    int num_cus;
    size_t lds_per_block;
    hipEvent_t ev0, ev1;
    int timed;
};

// grow-only device scratch owned by the context
static int ensure_doubles(psh_ctx* c, double** buf, size_t* have, size_t need);

static void set_err(psh_ctx* c, const char* fmt, const char* a = "", const char* b = "") {
    if (c) snprintf(c->err, sizeof c->err, fmt, a, b);
}

#define PSH_HIP(ctx, call)                                                    \
    do {                                                                      \
        hipError_t e_ = (call);                                               \
        if (e_ != hipSuccess) {                                               \
            set_err(ctx, "%s failed: %s", #call, hipGetErrorString(e_));      \
            return PSH_ERR_HIP;                                               \
        }                                                                     \
    } while (0)

static int ensure_doubles(psh_ctx* c, double** buf, size_t* have, size_t need) {
    if (need <= *have) return PSH_OK;
    if (*buf) PSH_HIP(c, hipFree(*buf));
    *buf = nullptr; *have = 0;
    PSH_HIP(c, hipMalloc(buf, need * sizeof(double)));
    *have = need;
    return PSH_OK;
}

// Persistent launch: as many workgroups as the device keeps resident (lanes pull envs from the queue), fewer when the
// batch is smaller than that.
template <class K> static unsigned persistent_grid(psh_ctx* c, K kernel, int block, size_t lds_bytes, int n_envs) {
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, block, lds_bytes) != hipSuccess || per_cu < 1) per_cu = 1;
    const long resident = (long)per_cu * c->num_cus;
    const long need = ((long)n_envs + block - 1) / block;
    return (unsigned)(need < resident ? need : resident);
}

// ------------------------------------------------------------------------------------------ kernel arguments
struct RolloutArgs {
    int n_envs, max_steps, flags, init_dim;
    double dt;
    const double* init_state; long ld_init;
    const double* domain_param; const int* dp_index; long ld_dp;
    psh_task task;
    psh_wrappers wrap;     // directly behind `task`: the two are staged into LDS as one block
    psh_policy pol;
    int obs_dim;           // entries of the observation the policy sees (all, or those kept by OBS_PARTIAL)
    double* delay_q;       // ACT_DELAY: ring buffers [PSH_MAX_DELAY][A][n_lanes], one column per resident lane
    long n_lanes;
    int num_feat;          // LINEAR: total feature count F
    int lds_doubles;       // FNN: doubles of the packed weight image
    const double* wfrag;   // FNN: packed weight image (device)
    double* ret; int* length; int* status; double* final_state; double* obs; double* act; double* rew;
    unsigned int* queue;   // next unassigned env index (zeroed before the launch): lanes pull work from it
    const double* plate;   // QBB: plate angles from the IK prologue [2, N], or NULL (simplified dynamics)
    const double* tanh_tab;  // FNN: tanh table in global memory, copied to LDS by every workgroup
    int tanh_n;              // its length (fine or coarse table)
};

// The task description is read from an LDS copy inside the step loop: as kernel arguments its ~90 constants would be
// pulled into scalar registers for the whole loop and spill.
static_assert(sizeof(psh_task) % 8 == 0 && sizeof(psh_wrappers) % 8 == 0, "staged as doubles");
static_assert(offsetof(RolloutArgs, wrap) == offsetof(RolloutArgs, task) + sizeof(psh_task), "staged as one block");
constexpr int TASK_DOUBLES = (sizeof(psh_task) + sizeof(psh_wrappers)) / 8;
constexpr int TANH8_PAD = (PSH_TANH8_N + 1) / 2 * 2, TANH_TAB_PAD = (PSH_TANH_TAB_N + 1) / 2 * 2;

// The LDS copy is addressed through a base the compiler has to keep in ONE vector register (it cannot see through the
// empty asm), so every field is a ds_read with an immediate offset.  Left to itself it materialises one scalar
// address per field, runs out of scalar registers and moves each of them through v_readlane/v_mov before use.
typedef __attribute__((address_space(3))) double lds_double;
__device__ __forceinline__ const double* opaque_lds_base(double* p) {
    lds_double* q = (lds_double*)p;
    asm volatile("" : "+v"(q));
    return (const double*)q;
}

__device__ __forceinline__ void stage_task(const RolloutArgs& a, double* dst) {
    const double* src = reinterpret_cast<const double*>(&a.task);
    for (int j = threadIdx.x; j < TASK_DOUBLES; j += blockDim.x) dst[j] = src[j];
}

// ------------------------------------------------------------------------------------------------ math helpers
PSH_DEV double nl_apply(int nl, double v) {
    switch (nl) {
    case PSH_NL_TANH: return psh_tanh(v);
    case PSH_NL_RELU: return v > 0. ? v : 0.;
    case PSH_NL_SIGMOID: return psh_sigmoid(v);
    default: return v;
    }
}

PSH_DEV double feat_apply(int id, double x) {
    switch (id) {
    case PSH_FEAT_SIGN: return (double)((x > 0.) - (x < 0.));
    case PSH_FEAT_ABS: return fabs(x);
    case PSH_FEAT_SQUARED: return x * x;
    case PSH_FEAT_CUBED: return x * x * x;
    case PSH_FEAT_SIG: return psh_sigmoid(x);
    case PSH_FEAT_BELL: return psh_exp(-(x * x));
    case PSH_FEAT_SIN: { double s, c; psh_sincos(x, &s, &c); return s; }
    case PSH_FEAT_COS: { double s, c; psh_sincos(x, &s, &c); return c; }
    case PSH_FEAT_SINSIN: { double s, c; psh_sincos(x, &s, &c); return s * s; }
    case PSH_FEAT_SINCOS: { double s, c; psh_sincos(x, &s, &c); return s * c; }
    default: return x;  // identity
    }
}

// ------------------------------------------------------------------------------------------ one lane = one env
// n standard normals for (seed, env, step, stream): Box-Muller pairs, pair p from counter word 3 = stream + 16 p
template <int N> PSH_DEV void step_normals(uint64_t seed, uint64_t env, uint32_t t, uint32_t stream, double* out) {
#pragma unroll
    for (int p = 0; 2 * p < N; ++p) {
        double ua, ub, sn, cs;
        psh_draw2(seed, env, t, stream + 16u * p, ua, ub);
        const double r = sqrt(-2.0 * log(1.0 - ua));
        sincos(2.0 * PSH_PI * ub, &sn, &cs);
        out[2 * p] = r * cs;
        if (2 * p + 1 < N) out[2 * p + 1] = r * sn;
    }
}

// WRAP = true adds the fused environment wrappers (psh_wrappers) around the simulation; the bare variant carries
// none of that code or state.
template <int KIND, bool WRAP> struct Lane {
    using E = Env<KIND>;
    E e;
    double c_max, ret;
    long i;           // env this lane currently runs, -1: none
    int curr_step, yielded, st, k;
    int delay, qpos;  // WRAP: ActDelayWrapper delay of this env, write position of its ring buffer
    bool done, live;  // live: this lane still has steps to execute
    bool exhausted;   // the work queue had nothing left for this lane
    bool rec_alias;   // WRAP: the array the simulation receives is one the rollout has recorded

    PSH_DEV void init() { i = -1; live = false; exhausted = false; }

    // Lanes pull environments from a global queue: a lane whose env is finished stores its results and takes the next
    // unassigned env, so a wave keeps all 64 lanes busy although episodes end at different steps.  Which lane runs
    // an env has no influence on its results.
    PSH_DEV void turnover(const RolloutArgs& a, const psh_task& task, const psh_wrappers& wrap) {
        if (i >= 0 && !live) { store(a, wrap); i = -1; }
        if (i < 0 && !exhausted) {
            const unsigned int nxt = atomicAdd(a.queue, 1u);
            if (nxt < (unsigned int)a.n_envs) { i = (long)nxt; load(a, task, wrap); } else { exhausted = true; }
        }
    }

    PSH_DEV void load(const RolloutArgs& a, const psh_task& task, const psh_wrappers& wrap) {
        const long col = a.dp_index ? (long)a.dp_index[i] : i;
        const DpCol p{a.domain_param, a.ld_dp, col};
        e.setup(p);
        c_max = task_c_max<KIND>(e, task);
        double init[E::S];
        for (int r = 0; r < E::S; ++r) init[r] = (r < a.init_dim) ? a.init_state[r * a.ld_init + i] : 0.;
        if constexpr (KIND == PSH_ENV_QBB) {
            e.reset(init, a.init_dim, true);  // plate angles come from the IK prologue kernel
            if (a.plate) { e.plate[0] = a.plate[i]; e.plate[1] = a.plate[(long)a.n_envs + i]; }
        } else {
            e.reset(init, a.init_dim, false);
        }
        ret = 0.; curr_step = 0; yielded = 0; st = 0; k = 0; done = false;
        live = a.max_steps > 0;
        if constexpr (WRAP) {
            bool has_delay = false, new_array = false;
            for (int j = 0; j < wrap.n_ops; ++j) {
                has_delay = has_delay || wrap.ops[j] == PSH_WRAP_ACT_DELAY;
                new_array = new_array || wrap.ops[j] == PSH_WRAP_ACT_NORM || wrap.ops[j] == PSH_WRAP_ACT_NOISE;
            }
            delay = 0; qpos = 0;
            if (has_delay) {  // ActDelayWrapper.reset / _load_domain_param (action_delay.py:46-64)
                delay = wrap.act_delay_row >= 0 ? (int)a.domain_param[wrap.act_delay_row * a.ld_dp + col] : wrap.act_delay;
                delay = delay < 0 ? 0 : (delay > PSH_MAX_DELAY ? PSH_MAX_DELAY : delay);
            }
            rec_alias = !new_array;
        }
    }

    // EnvWrapperObs.reset/step (environment_wrappers/base.py:216-231): the simulation's observation through the
    // observation wrappers, innermost first; `out` receives the kept entries in order, zero padded
    PSH_DEV void wrapped_obs(const RolloutArgs& a, const psh_wrappers& wrap, uint32_t t, double* out) const {
        double o[E::O];
        e.observe(o);
        int keep = (1 << E::O) - 1;
        for (int j = wrap.n_ops - 1; j >= 0; --j) {
            const int op = wrap.ops[j];
            if (op == PSH_WRAP_OBS_NOISE) {
                double n[E::O];
                step_normals<E::O>(wrap.noise_seed, wrap.env_offset + (uint64_t)i, t, 1u, n);
#pragma unroll
                for (int r = 0; r < E::O; ++r) o[r] = o[r] + (n[r] * wrap.obs_noise_std[r] + wrap.obs_noise_mean[r]);
            } else if (op == PSH_WRAP_OBS_NORM) {
                double lo[E::O], hi[E::O];
                e.obs_bounds(lo, hi);
#pragma unroll
                for (int r = 0; r < E::O; ++r) {
                    const double l = wrap.obs_lo[r], h = wrap.obs_hi[r];
                    lo[r] = (l != l) ? lo[r] : l;
                    hi[r] = (h != h) ? hi[r] : h;
                    o[r] = (o[r] - lo[r]) / (hi[r] - lo[r]) * 2 - 1;
                }
            } else if (op == PSH_WRAP_OBS_PARTIAL) {
                keep = wrap.obs_keep_mask;
            }
        }
#pragma unroll
        for (int r = 0; r < E::O; ++r) out[r] = 0.;
        int n = 0;
#pragma unroll
        for (int r = 0; r < E::O; ++r) {
            if (keep & (1 << r)) {
#pragma unroll
                for (int q = 0; q < E::O; ++q) out[q] = (q == n) ? o[r] : out[q];  // out[n] = o[r] without dynamic indexing
                n += 1;
            }
        }
    }

    // rollout(): NaN check of the observation, record it, (policy), NaN check of the action
    PSH_DEV void pre_policy(const RolloutArgs& a, const psh_wrappers& wrap, double* obs) {
        if constexpr (WRAP) wrapped_obs(a, wrap, (uint32_t)k, obs); else e.observe(obs);
        const int Oe = WRAP ? a.obs_dim : E::O;
        bool nan = false;
#pragma unroll
        for (int r = 0; r < E::O; ++r) nan = nan || (r < Oe && obs[r] != obs[r]);
        if (a.obs) {
#pragma unroll
            for (int r = 0; r < E::O; ++r)
                if (r < Oe) a.obs[((long)k * Oe + r) * a.n_envs + i] = obs[r];
        }
        if (nan) { st |= PSH_ST_NAN; live = false; }
    }

    PSH_DEV void post_policy(const RolloutArgs& a, const psh_task& task, const psh_wrappers& wrap, double* act) {
        if constexpr (WRAP) {
            // StochasticActionExplStrat.forward (exploration/stochastic_action.py:42-68): the exploratory action
            // replaces the policy's output for everything that follows
            if (a.pol.expl_kind == PSH_EXPL_NORMAL) {
                double n[E::A];
                step_normals<E::A>(wrap.noise_seed, wrap.env_offset + (uint64_t)i, (uint32_t)k, 3u, n);
#pragma unroll
                for (int r = 0; r < E::A; ++r) act[r] = (act[r] + a.pol.expl_mean[r]) + n[r] * a.pol.expl_scale[r];
            } else if (a.pol.expl_kind == PSH_EXPL_UNIFORM) {
                double u[2];
                psh_draw2(wrap.noise_seed, wrap.env_offset + (uint64_t)i, (uint32_t)k, 3u, u[0], u[1]);
#pragma unroll
                for (int r = 0; r < E::A; ++r) {
                    const double m = act[r] + a.pol.expl_mean[r];
                    const double lo = m - a.pol.expl_scale[r], hi = m + a.pol.expl_scale[r];
                    act[r] = lo + u[r] * (hi - lo);
                }
            }
        }
        bool nan = false;
#pragma unroll
        for (int r = 0; r < E::A; ++r) nan = nan || (act[r] != act[r]);
        if (nan) { st |= PSH_ST_NAN; live = false; return; }
        const bool simplified = (a.flags & PSH_F_SIMPLIFIED_DYN) != 0;
        double rew;
        if constexpr (WRAP) {
            // EnvWrapperAct.step (environment_wrappers/base.py:168-174), outermost wrapper first
            double inner[E::A];
#pragma unroll
            for (int r = 0; r < E::A; ++r) inner[r] = act[r];
            for (int j = 0; j < wrap.n_ops; ++j) {
                const int op = wrap.ops[j];
                if (op == PSH_WRAP_ACT_DELAY && delay > 0) {  // FIFO as a ring: the slot written now is read `delay` steps on
                    const long slot = (long)blockIdx.x * blockDim.x + threadIdx.x;
#pragma unroll
                    for (int r = 0; r < E::A; ++r) {
                        double* q = a.delay_q + ((long)qpos * E::A + r) * a.n_lanes + slot;
                        const double old = (k >= delay) ? *q : 0.;
                        *q = inner[r];
                        inner[r] = old;
                    }
                    qpos = (qpos + 1 == delay) ? 0 : qpos + 1;
                } else if (op == PSH_WRAP_ACT_NORM) {
                    double lo[E::A], hi[E::A];
                    e.act_bounds(lo, hi);
#pragma unroll
                    for (int r = 0; r < E::A; ++r) inner[r] = lo[r] + (inner[r] + 1) * (hi[r] - lo[r]) / 2;
                } else if (op == PSH_WRAP_ACT_NOISE) {
                    double n[E::A];
                    step_normals<E::A>(wrap.noise_seed, wrap.env_offset + (uint64_t)i, (uint32_t)k, 2u, n);
#pragma unroll
                    for (int r = 0; r < E::A; ++r) inner[r] = inner[r] + (n[r] * wrap.act_noise_std[r] + wrap.act_noise_mean[r]);
                }
            }
            rew = env_step<KIND>(e, task, c_max, curr_step, yielded, inner, a.dt, a.max_steps, false, simplified, done, st);
            if (a.act) {
#pragma unroll
                for (int r = 0; r < E::A; ++r) a.act[((long)k * E::A + r) * a.n_envs + i] = act[r];
                // ActDelayWrapper queues the caller's array objects: without an array-creating wrapper the simulation
                // wrote its dead zone (QBB) into the action recorded `delay` steps ago
                if (rec_alias && k - delay >= 0) {
#pragma unroll
                    for (int r = 0; r < E::A; ++r) a.act[((long)(k - delay) * E::A + r) * a.n_envs + i] = inner[r];
                }
            }
        } else {
            rew = env_step<KIND>(e, task, c_max, curr_step, yielded, act, a.dt, a.max_steps, false, simplified, done, st);
            if (a.act) {
#pragma unroll
                for (int r = 0; r < E::A; ++r) a.act[((long)k * E::A + r) * a.n_envs + i] = act[r];
            }
        }
        if (a.rew) a.rew[(long)k * a.n_envs + i] = rew;
        ret += rew;
        k += 1;
        if (st & PSH_ST_NAN) live = false;
        if ((done && (a.flags & PSH_F_STOP_ON_DONE)) || curr_step >= a.max_steps) live = false;
    }

    PSH_DEV void store(const RolloutArgs& a, const psh_wrappers& wrap) {
        if (a.obs) {  // final observation
            double obs[E::O];
            if constexpr (WRAP) wrapped_obs(a, wrap, (uint32_t)k, obs); else e.observe(obs);
            const int Oe = WRAP ? a.obs_dim : E::O;
#pragma unroll
            for (int r = 0; r < E::O; ++r)
                if (r < Oe) a.obs[((long)k * Oe + r) * a.n_envs + i] = obs[r];
        }
        a.ret[i] = ret;
        a.length[i] = k;
        a.status[i] = st;
        if (a.final_state) {
#pragma unroll
            for (int r = 0; r < E::S; ++r) a.final_state[(long)r * a.n_envs + i] = e.s[r];
        }
    }
};

// ------------------------------------------------------------------------------------- LinearPolicy rollouts
// PER_ENV = false: one weight set, staged in LDS and broadcast; true: env i reads row policy_index[i] of params.
template <int KIND, bool PER_ENV, bool WRAP>
__global__ __launch_bounds__(256, KIND == PSH_ENV_QBB ? 1 : 2) void k_rollout_linear(const RolloutArgs a) {
    using E = Env<KIND>;
    constexpr int W_DOUBLES = PSH_MAX_ACT * PSH_MAX_FEATS * PSH_MAX_OBS;
    __shared__ double lds_lin[W_DOUBLES + TASK_DOUBLES];
    double* const lds_w = lds_lin;
    const double* const task_base = opaque_lds_base(lds_lin + W_DOUBLES);
    const psh_task& task = *reinterpret_cast<const psh_task*>(task_base);
    const psh_wrappers& wrap = *reinterpret_cast<const psh_wrappers*>(task_base + sizeof(psh_task) / 8);
    const int F = a.num_feat;
    const int Oe = WRAP ? a.obs_dim : E::O;  // observation entries the policy sees
    if (!PER_ENV) {
        for (int j = threadIdx.x; j < E::A * F; j += blockDim.x) lds_w[j] = a.pol.params[j];
    }
    stage_task(a, lds_lin + W_DOUBLES);
    __syncthreads();
    Lane<KIND, WRAP> L;
    L.init();
    const double* w = lds_w;
    long w_env = -1;
    for (;;) {
        asm volatile("" ::: "memory");  // keep the LDS constants out of registers across iterations
        L.turnover(a, task, wrap);
        if (!__any(L.i >= 0)) break;  // every lane of the wave is out of work
        if (L.i < 0) continue;
        if (PER_ENV && w_env != L.i) {
            w = a.pol.params + (long)(a.pol.policy_index ? a.pol.policy_index[L.i] : 0) * a.pol.num_param;
            w_env = L.i;
        }
        double obs[E::O], act[E::A];
        L.pre_policy(a, wrap, obs);
        if (!L.live) continue;
#pragma unroll
        for (int r = 0; r < E::A; ++r) act[r] = 0.;
        int n = 0;
        for (int f = 0; f < a.pol.n_feats; ++f) {
            const int id = a.pol.feats[f];
            if (id == PSH_FEAT_CONST) {
#pragma unroll
                for (int r = 0; r < E::A; ++r) act[r] += w[r * F + n];
                n += 1;
            } else {
#pragma unroll
                for (int q = 0; q < E::O; ++q) {
                    if (WRAP && q >= Oe) break;
                    const double v = feat_apply(id, obs[q]);
#pragma unroll
                    for (int r = 0; r < E::A; ++r) act[r] += w[r * F + n + q] * v;
                }
                n += Oe;
            }
        }
        L.post_policy(a, task, wrap, act);
    }
}

// ---------------------------------------------------------------------------------------- FNNPolicy rollouts
// Transposed formulation H^T[hidden, env] = W[hidden, k] . X^T[k, env] on v_mfma_f64_16x16x4_f64:
//   A operand: lane l holds W[16t + (l & 15)][4s + (l >> 4)]           (weights, read from LDS)
//   B operand: lane l holds X[k = 4s + (l >> 4)][env = l & 15]          (activations of 16 envs)
//   D result : lane l, reg r holds H[16t + 4r + (l >> 4)][env = l & 15]
// so reg r of output tile t IS the B operand of k-step 4t + r of the next layer: layers chain without any
// cross-lane movement.  A wave owns 64 envs (lane = env for the dynamics) and runs the policy as 4 tiles of 16
// envs; only the observations (through an LDS staging row per wave) and the action (two xor-shuffles) cross lanes.
//
// On gfx950 the fp64 MFMA and the fp64 vector instructions share one pipe (64 and 4 cycles per instruction, no
// overlap, also not between waves), so every vector instruction next to the MFMAs costs throughput: the activation
// function is the table-based psh_tanh_tab and the loop keeps its constants in LDS rather than in registers.
//
// Packed weight image (doubles), HT = HP/16, KS = HP/4, K0 = OP/4 with OP = observation rows padded to 4:
//   [HT*K0 frags of 64][b0: HP]  then per further hidden layer [HT*KS frags of 64][b: HP]  then [Wout: A*HP][bout: 2]
// Dynamic LDS: [weight image][8 staging areas of OP*64][tanh table][task].  FINE selects the fine tanh table (40 KB);
// the widest, deepest networks leave no room for it and run with the coarse one.
typedef double v4d __attribute__((ext_vector_type(4)));

template <int N, bool FINE> PSH_DEV void nl_apply_all(int nl, double* y, const double* tanh_tab) {
    if (nl == PSH_NL_TANH) {
        if (FINE) {
            // all table loads first, then the arithmetic: the LDS latency is paid once per layer, not per element
            constexpr int CH = N < 8 ? N : 8;  // chunked: d and T of a whole layer would not fit the register file
#pragma unroll
            for (int m0 = 0; m0 < N; m0 += CH) {
                double d[CH], T[CH];
#pragma unroll
                for (int m = 0; m < CH; ++m) T[m] = tanh_tab[psh_tanh8_index(y[m0 + m], &d[m])];
#pragma unroll
                for (int m = 0; m < CH; ++m) y[m0 + m] = psh_tanh8_finish(y[m0 + m], d[m], T[m]);
            }
        } else {
#pragma unroll
            for (int m = 0; m < N; ++m) y[m] = psh_tanh_tab(y[m], tanh_tab);
        }
    } else if (nl == PSH_NL_SIGMOID) {
#pragma unroll
        for (int m = 0; m < N; ++m) y[m] = psh_sigmoid(y[m]);
    } else if (nl == PSH_NL_RELU) {
#pragma unroll
        for (int m = 0; m < N; ++m) y[m] = y[m] > 0. ? y[m] : 0.;
    }
}

template <int KIND, int HP, bool FINE, bool WRAP>
__global__ __launch_bounds__(512) void k_rollout_fnn(const RolloutArgs a) {
    using E = Env<KIND>;
    constexpr int HT = HP / 16, KS = HP / 4, OP = (E::O + 3) / 4 * 4, K0 = OP / 4;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double* const w_lds = lds;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = lane >> 4, c = lane & 15;
    double* const stage = lds + a.lds_doubles + wave * (OP * 64);
    double* const tanh_tab = lds + a.lds_doubles + 8 * (OP * 64);
    double* const task_lds = tanh_tab + (FINE ? TANH8_PAD : TANH_TAB_PAD);
    const double* const task_base = opaque_lds_base(task_lds);
    const psh_task& task = *reinterpret_cast<const psh_task*>(task_base);
    const psh_wrappers& wrap = *reinterpret_cast<const psh_wrappers*>(task_base + sizeof(psh_task) / 8);

    for (int j = threadIdx.x; j < a.lds_doubles; j += blockDim.x) w_lds[j] = a.wfrag[j];
    for (int j = threadIdx.x; j < a.tanh_n; j += blockDim.x) tanh_tab[j] = a.tanh_tab[j];
    stage_task(a, task_lds);
    __syncthreads();

    Lane<KIND, WRAP> L;
    L.init();

    const int n_hidden = a.pol.n_hidden;
    for (;;) {
        L.turnover(a, task, wrap);
        if (!__any(L.i >= 0)) break;  // every lane of the wave is out of work
        double obs[OP], act[E::A];
#pragma unroll
        for (int r = 0; r < OP; ++r) obs[r] = 0.;
        if (L.live) L.pre_policy(a, wrap, obs);
        if (!__any(L.live)) continue;
        // stage the observations: row r of this wave's staging area holds obs[r] of its 64 envs
#pragma unroll
        for (int r = 0; r < OP; ++r) stage[r * 64 + lane] = obs[r];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        for (int tile = 0; tile < 4; ++tile) {
            double y[KS];
            // layer 0
            {
                double xin[K0];
#pragma unroll
                for (int s = 0; s < K0; ++s) xin[s] = stage[(4 * s + q) * 64 + 16 * tile + c];
                const double* wl = w_lds;
                const double* bl = w_lds + HT * K0 * 64;
#pragma unroll
                for (int t = 0; t < HT; ++t) {
                    v4d acc;
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[r] = bl[16 * t + 4 * r + q];
#pragma unroll
                    for (int s = 0; s < K0; ++s)
                        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(wl[(t * K0 + s) * 64 + lane], xin[s], acc, 0, 0, 0);
#pragma unroll
                    for (int r = 0; r < 4; ++r) y[4 * t + r] = acc[r];
                }
                nl_apply_all<KS, FINE>(a.pol.hidden_nonlin[0], y, tanh_tab);
            }
            // further hidden layers
            int off = HT * K0 * 64 + HP;
            for (int l = 1; l < n_hidden; ++l) {
                double x[KS];
#pragma unroll
                for (int m = 0; m < KS; ++m) x[m] = y[m];
                const double* wl = w_lds + off;
                const double* bl = wl + HT * KS * 64;
#pragma unroll
                for (int t = 0; t < HT; ++t) {
                    v4d acc;
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[r] = bl[16 * t + 4 * r + q];
#pragma unroll
                    for (int s = 0; s < KS; ++s)
                        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(wl[(t * KS + s) * 64 + lane], x[s], acc, 0, 0, 0);
#pragma unroll
                    for (int r = 0; r < 4; ++r) y[4 * t + r] = acc[r];
                }
                nl_apply_all<KS, FINE>(a.pol.hidden_nonlin[l], y, tanh_tab);
                off += HT * KS * 64 + HP;
            }
            // output layer on the vector ALU: each lane sums its 16 hidden rows, then the 4 lane groups are added
            const double* wo = w_lds + off;
#pragma unroll
            for (int r = 0; r < E::A; ++r) {
                double p = 0.;
#pragma unroll
                for (int m = 0; m < KS; ++m) p += wo[r * HP + (m >> 2) * 16 + (m & 3) * 4 + q] * y[m];
                p += __shfl_xor(p, 16);
                p += __shfl_xor(p, 32);
                p += wo[E::A * HP + r];
                nl_apply_all<1, FINE>(a.pol.out_nonlin, &p, tanh_tab);
                if (q == tile) act[r] = p;
            }
        }
        if (L.live) L.post_policy(a, task, wrap, act);
    }
}

// QBallBalancerSim.reset: plate angles of every env by inverse kinematics, ahead of the rollout kernel, so that a lane
// taking over a new env only reads two numbers instead of running 200 optimiser steps while its wave waits.
__global__ __launch_bounds__(256) void k_qbb_plate(const RolloutArgs a, double* plate) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n_envs) return;
    const long col = a.dp_index ? (long)a.dp_index[i] : i;
    const DpCol p{a.domain_param, a.ld_dp, col};
    const double th_x = (a.init_dim == 8) ? a.init_state[0 * a.ld_init + i] : 0.;
    const double th_y = (a.init_dim == 8) ? a.init_state[1 * a.ld_init + i] : 0.;
    plate[i] = qbb_ik(th_x + p[18], p[4], p[3]);
    plate[(long)a.n_envs + i] = qbb_ik(th_y + p[19], p[4], p[3]);
}

// Builds the packed weight image of k_rollout_fnn from the torch-ordered flat parameter vector.
__global__ void k_pack_fnn(const double* __restrict__ params, double* __restrict__ out, int total, int O, int A, int HP,
                           int OP, int n_hidden, int h0, int h1, int h2, int h3) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int hs[4] = {h0, h1, h2, h3};
    const int HT = HP / 16, KS = HP / 4, K0 = OP / 4;
    int off = 0, poff = 0, last = O;
    double v = 0.;
    bool found = false;
    for (int l = 0; l < n_hidden && !found; ++l) {
        const int ks = (l == 0) ? K0 : KS, h = hs[l];
        const int nfrag = HT * ks * 64;
        if (idx < off + nfrag) {
            const int f = (idx - off) >> 6, lane = (idx - off) & 63;
            const int t = f / ks, s = f % ks;
            const int row = 16 * t + (lane & 15), col = 4 * s + (lane >> 4);
            v = (row < h && col < last) ? params[poff + row * last + col] : 0.;
            found = true;
        } else if (idx < off + nfrag + HP) {
            const int j = idx - off - nfrag;
            v = (j < h) ? params[poff + h * last + j] : 0.;
            found = true;
        }
        off += nfrag + HP;
        poff += h * last + h;
        last = h;
    }
    if (!found) {
        const int j = idx - off;
        if (j < A * HP) {
            const int r = j / HP, m = j % HP;
            v = (m < last) ? params[poff + r * last + m] : 0.;
        } else {
            const int r = j - A * HP;
            v = (r < A) ? params[poff + A * last + r] : 0.;
        }
    }
    out[idx] = v;
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
    e.setup(DpCol{a.domain_param, a.ld_dp, a.dp_index ? (long)a.dp_index[i] : i});
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
    if (a.status) a.status[i] = nan ? PSH_ST_NAN : 0;
}

template <int KIND> __global__ __launch_bounds__(256) void k_step(const StepArgs a) {
    using E = Env<KIND>;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n_envs) return;
    const long N = a.n_envs;
    E e;
    e.setup(DpCol{a.domain_param, a.ld_dp, a.dp_index ? (long)a.dp_index[i] : i});
    const double c_max = task_c_max<KIND>(e, a.task);
#pragma unroll
    for (int r = 0; r < E::S; ++r) e.s[r] = a.state[r * N + i];
    if constexpr (KIND == PSH_ENV_QBB) {
        e.plate[0] = a.state[8 * N + i];
        e.plate[1] = a.state[9 * N + i];
    }
    e.refresh();
    int curr_step = a.curr_step[i], yielded = a.flags_io[i] & 1, st = 0;
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
        psh_draw2(a.seed, a.env_offset + (uint64_t)i, (uint32_t)d.row, 0u, ua, ub);
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
    const long col = a.dp_index ? (long)a.dp_index[i] : i;
    double lo[4], hi[4], u[5], ub;
    const uint64_t env = a.env_offset + (uint64_t)i;
    for (int r = 0; r < 5; ++r) psh_draw2(a.seed, env, (uint32_t)r, 1u, u[r], ub);
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
    for (int r = 0; r < I; ++r) a.out[(long)r * a.ld_out + i] = v[r];
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
    case 6: y[i] = psh_tanh8(x[i], tanh_tab); break;
    case 7: y[i] = psh_tanh_tab(x[i], tanh_tab + PSH_TANH8_N); break;
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
    static double tab[PSH_TANH8_N + PSH_TANH_TAB_N];
    for (int j = 0; j < PSH_TANH8_N; ++j) tab[j] = (double)tanhl((long double)j / 256.0L);
    for (int j = 0; j < PSH_TANH_TAB_N; ++j) tab[PSH_TANH8_N + j] = (double)tanhl((long double)j / 64.0L);
    if (hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess ||
        hipMalloc(&c->queue, 256) != hipSuccess || hipMalloc(&c->tanh_tab, sizeof tab) != hipSuccess ||
        hipMemcpy(c->tanh_tab, tab, sizeof tab, hipMemcpyHostToDevice) != hipSuccess) {
        delete c;
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
    (void)hipEventDestroy(c->ev0);
    (void)hipEventDestroy(c->ev1);
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

// The fused wrappers of a rollout: every op known and used once, sizes in range.  Yields the policy's input dimension.
static int check_wrap(psh_ctx* c, const psh_wrappers& w, int O, int* obs_dim, bool* has_delay) {
    *obs_dim = O;
    *has_delay = false;
    if (w.n_ops < 0 || w.n_ops > PSH_MAX_WRAP) { set_err(c, "wrap.n_ops outside 0..PSH_MAX_WRAP"); return PSH_ERR_ARG; }
    unsigned seen = 0;
    for (int j = 0; j < w.n_ops; ++j) {
        const int op = w.ops[j];
        if (op < PSH_WRAP_ACT_NORM || op > PSH_WRAP_OBS_PARTIAL) { set_err(c, "unknown wrapper op"); return PSH_ERR_ARG; }
        if (seen & (1u << op)) { set_err(c, "each wrapper kind can be fused once"); return PSH_ERR_UNSUPPORTED; }
        seen |= 1u << op;
        if (op == PSH_WRAP_ACT_DELAY) {
            *has_delay = true;
            if (w.act_delay_row < 0 && (w.act_delay < 0 || w.act_delay > PSH_MAX_DELAY)) {
                set_err(c, "act_delay outside 0..PSH_MAX_DELAY");
                return PSH_ERR_ARG;
            }
        }
        if (op == PSH_WRAP_OBS_PARTIAL) {
            const int m = w.obs_keep_mask;
            if (m <= 0 || m >= (1 << O)) {
                if (m != (1 << O) - 1) { set_err(c, "obs_keep_mask must select 1..O observations"); return PSH_ERR_ARG; }
            }
            *obs_dim = __builtin_popcount((unsigned)m);
        }
    }
    return PSH_OK;
}

// Persistent launch of a rollout kernel: grid from the occupancy query, the ActDelay ring buffers sized to it.
template <class K>
static int launch_persistent(psh_ctx* c, K kernel, int block, size_t lds_bytes, RolloutArgs& a, int act_dim, bool need_delay_q,
                             hipStream_t st) {
    if (lds_bytes)
        PSH_HIP(c, hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    const unsigned grid = persistent_grid(c, kernel, block, lds_bytes, a.n_envs);
    a.n_lanes = (long)grid * block;
    if (need_delay_q) {
        const int rc = ensure_doubles(c, &c->delay_q, &c->delay_q_doubles, (size_t)PSH_MAX_DELAY * act_dim * a.n_lanes);
        if (rc != PSH_OK) return rc;
        a.delay_q = c->delay_q;
    }
    if (c->timed) PSH_HIP(c, hipEventRecord(c->ev0, st));
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), lds_bytes, st, a);
    PSH_HIP(c, hipGetLastError());
    if (c->timed) PSH_HIP(c, hipEventRecord(c->ev1, st));
    return PSH_OK;
}

template <int KIND> static int launch_rollout(psh_ctx* c, const psh_rollout_desc* d, RolloutArgs& a, hipStream_t st) {
    using E = Env<KIND>;
    const psh_policy& p = d->policy;
    const bool per_env = p.policy_index != nullptr && p.n_policies > 1;
    int Oe = E::O;
    bool has_delay = false;
    {
        const int rc = check_wrap(c, d->wrap, E::O, &Oe, &has_delay);
        if (rc != PSH_OK) return rc;
    }
    if (p.expl_kind < PSH_EXPL_NONE || p.expl_kind > PSH_EXPL_UNIFORM) { set_err(c, "unknown exploration kind"); return PSH_ERR_ARG; }
    const bool wrapped = d->wrap.n_ops > 0 || p.expl_kind != PSH_EXPL_NONE;  // both live in the WRAP kernels
    a.obs_dim = Oe;
    if (p.kind == PSH_POL_LINEAR) {
        if (p.n_feats < 1 || p.n_feats > PSH_MAX_FEATS) { set_err(c, "LinearPolicy needs 1..8 feature functions"); return PSH_ERR_ARG; }
        int F = 0;
        for (int f = 0; f < p.n_feats; ++f) {
            if (p.feats[f] < 0 || p.feats[f] > PSH_FEAT_SINCOS) { set_err(c, "unknown feature id"); return PSH_ERR_ARG; }
            F += (p.feats[f] == PSH_FEAT_CONST) ? 1 : Oe;
        }
        if (p.num_param != E::A * F) { set_err(c, "num_param does not match the LinearPolicy layout"); return PSH_ERR_ARG; }
        a.num_feat = F;
        if (per_env)
            return wrapped ? launch_persistent(c, k_rollout_linear<KIND, true, true>, 256, 0, a, E::A, has_delay, st)
                           : launch_persistent(c, k_rollout_linear<KIND, true, false>, 256, 0, a, E::A, false, st);
        return wrapped ? launch_persistent(c, k_rollout_linear<KIND, false, true>, 256, 0, a, E::A, has_delay, st)
                       : launch_persistent(c, k_rollout_linear<KIND, false, false>, 256, 0, a, E::A, false, st);
    }
    if (p.kind != PSH_POL_FNN) { set_err(c, "unknown policy kind"); return PSH_ERR_ARG; }
    if (p.n_hidden < 1 || p.n_hidden > PSH_MAX_HIDDEN) { set_err(c, "FNNPolicy needs 1..4 hidden layers"); return PSH_ERR_ARG; }
    int hmax = 0, np = 0, last = Oe;
    for (int l = 0; l < p.n_hidden; ++l) {
        const int h = p.hidden[l];
        if (h < 1 || h > PSH_MAX_WIDTH) { set_err(c, "hidden sizes must be within 1..64"); return PSH_ERR_ARG; }
        if (p.hidden_nonlin[l] < 0 || p.hidden_nonlin[l] > PSH_NL_SIGMOID) { set_err(c, "unknown nonlinearity"); return PSH_ERR_ARG; }
        hmax = h > hmax ? h : hmax;
        np += h * last + h;
        last = h;
    }
    np += E::A * last + E::A;
    if (p.num_param != np) { set_err(c, "num_param does not match the FNNPolicy layout"); return PSH_ERR_ARG; }
    if (per_env) { set_err(c, "per-env FNN policies are not supported yet"); return PSH_ERR_UNSUPPORTED; }
    const int HP = hmax <= 16 ? 16 : (hmax <= 32 ? 32 : 64);
    const int OP = (E::O + 3) / 4 * 4, HT = HP / 16, KS = HP / 4, K0 = OP / 4;
    const int total = HT * K0 * 64 + HP + (p.n_hidden - 1) * (HT * KS * 64 + HP) + E::A * HP + 2;
    {
        const int rc = ensure_doubles(c, &c->wfrag, &c->wfrag_doubles, (size_t)total);
        if (rc != PSH_OK) return rc;
    }
    hipLaunchKernelGGL(k_pack_fnn, dim3((total + 255) / 256), dim3(256), 0, st, p.params, c->wfrag, total, Oe, (int)E::A,
                       HP, OP, p.n_hidden, p.hidden[0], p.hidden[1], p.hidden[2], p.hidden[3]);
    PSH_HIP(c, hipGetLastError());
    a.wfrag = c->wfrag;
    a.lds_doubles = total;
    constexpr int BLOCK = 512;
    const size_t base_doubles = (size_t)total + (BLOCK / 64) * OP * 64 + TASK_DOUBLES;
    const bool fine = (base_doubles + TANH8_PAD) * sizeof(double) <= c->lds_per_block;
    a.tanh_tab = fine ? c->tanh_tab : c->tanh_tab + PSH_TANH8_N;
    a.tanh_n = fine ? PSH_TANH8_N : PSH_TANH_TAB_N;
    const size_t lds_bytes = (base_doubles + (fine ? TANH8_PAD : TANH_TAB_PAD)) * sizeof(double);
    if (lds_bytes > c->lds_per_block) { set_err(c, "FNNPolicy too large for the LDS of one workgroup"); return PSH_ERR_UNSUPPORTED; }
#define PSH_LAUNCH_FNN(HP_, FINE_)                                                                                       \
    (wrapped ? launch_persistent(c, k_rollout_fnn<KIND, HP_, FINE_, true>, BLOCK, lds_bytes, a, E::A, has_delay, st)     \
             : launch_persistent(c, k_rollout_fnn<KIND, HP_, FINE_, false>, BLOCK, lds_bytes, a, E::A, false, st))
    if (HP == 16) return PSH_LAUNCH_FNN(16, true);
    if (HP == 32) return PSH_LAUNCH_FNN(32, true);
    if (fine) return PSH_LAUNCH_FNN(64, true);
    return PSH_LAUNCH_FNN(64, false);
#undef PSH_LAUNCH_FNN
}

extern "C" int psh_rollout(psh_ctx* c, const psh_rollout_desc* d) {
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
    int rc = check_task(c, d->task, dm[0]);
    if (rc != PSH_OK) return rc;
    PSH_HIP(c, hipSetDevice(c->device));
    RolloutArgs a;
    memset(&a, 0, sizeof a);
    a.n_envs = d->n_envs; a.max_steps = d->max_steps; a.flags = d->flags; a.init_dim = d->init_dim; a.dt = d->dt;
    a.init_state = d->init_state; a.ld_init = d->ld_init;
    a.domain_param = d->domain_param; a.dp_index = d->dp_index; a.ld_dp = d->ld_dp;
    a.task = d->task; a.wrap = d->wrap; a.pol = d->policy;
    a.ret = d->ret; a.length = d->length; a.status = d->status; a.final_state = d->final_state;
    a.obs = d->obs; a.act = d->act; a.rew = d->rew;
    hipStream_t st = (hipStream_t)d->stream;
    c->timed = 1;
    a.queue = c->queue;
    PSH_HIP(c, hipMemsetAsync(c->queue, 0, 256, st));
    if (d->env_kind == PSH_ENV_QBB && !(d->flags & PSH_F_SIMPLIFIED_DYN)) {
        rc = ensure_doubles(c, &c->plate, &c->plate_doubles, 2 * (size_t)d->n_envs);
        if (rc != PSH_OK) return rc;
        hipLaunchKernelGGL(k_qbb_plate, dim3((d->n_envs + 255) / 256), dim3(256), 0, st, a, c->plate);
        PSH_HIP(c, hipGetLastError());
        a.plate = c->plate;
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

extern "C" float psh_last_rollout_ms(psh_ctx* c) {
    if (!c || !c->timed) return -1.f;
    if (hipEventSynchronize(c->ev1) != hipSuccess) return -1.f;
    float ms = -1.f;
    if (hipEventElapsedTime(&ms, c->ev0, c->ev1) != hipSuccess) return -1.f;
    return ms;
}

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

extern "C" int psh_eval_math(psh_ctx* c, int32_t fn, int32_t n, const double* x, const double* m, double* y, void* stream) {
    if (!c || fn < 0 || fn > 7 || n < 0 || (n > 0 && (!x || !y || (fn == 5 && !m)))) { set_err(c, "invalid math-evaluation arguments"); return PSH_ERR_ARG; }
    if (n == 0) return PSH_OK;
    PSH_HIP(c, hipSetDevice(c->device));
    hipLaunchKernelGGL(k_eval_math, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, (int)fn, (int)n, x, m, y, (const double*)c->tanh_tab);
    PSH_HIP(c, hipGetLastError());
    return PSH_OK;
}
