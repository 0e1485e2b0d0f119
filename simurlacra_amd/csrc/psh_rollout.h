// The fused rollout kernels (reset + T x (policy forward, SimPyEnv.step) per environment lane) and their launch logic,
// as templates over the environment kind.  Each kind is instantiated in a translation unit of its own
// (rollout_kind.hip compiled with -DPSH_KIND=<k>), so the kinds build in parallel.
#pragma once

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstddef>
#include <cstring>
#include <type_traits>

#include "psh_ctx.h"
#include "psh_env.h"
#include "psh_rng.h"

// ------------------------------------------------------------------------------------------ kernel arguments
// Bit of RolloutArgs::flags set by the library alone (callers' flags are masked to the public PSH_F_* bits): the
// wrapper list is exactly [ActNormWrapper] and nothing else needs the wrapped kernels.  ActNormWrapper in front of the
// simulation is by far the most common wrapper configuration of the reference's training scripts, and it is one
// multiply-add per action: the bare kernels apply it themselves instead of leaving such launches to the wrapped ones.
#define PSH_F_INTERNAL_ACT_NORM (1 << 30)
#define PSH_F_PUBLIC_MASK (PSH_F_STOP_ON_DONE | PSH_F_SIMPLIFIED_DYN | PSH_F_MAX_STEPS_INF | PSH_F_TRAJ_ENV_MAJOR)

struct RolloutArgs {
    int n_envs, max_steps, flags, init_dim;
    double dt;
    const double* init_state; long ld_init;
    const double* domain_param; const int* dp_index; long ld_dp;
    psh_task task;
    psh_wrappers wrap;     // directly behind `task`: the two are staged into LDS as one block
    psh_policy pol;
    double bernoulli;      // rollout(bernoulli_reset): probability of ending the rollout after a step, 0 = off
    int obs_dim;           // entries of the observation the policy sees (all, or those kept by OBS_PARTIAL)
    int wrap_obs_ops;      // an observation wrapper (noise, normalisation, partial) is among wrap.ops: without one the
                           // observation path of the wrapped kernels is that of the bare ones
    double* delay_q;       // ACT_DELAY: ring buffers [n_lanes / 64][PSH_MAX_DELAY][A][64], one column per resident lane
    long n_lanes;
    int num_feat;          // LINEAR: total feature count F
    int lds_doubles;       // FNN: doubles of the packed weight image as a workgroup holds it in LDS
    int img_doubles;       // FNN: doubles of one packed weight image in global memory (>= lds_doubles)
    const double* wfrag;   // FNN: packed weight image (device)
    double* ret; int* length; int* status; double* final_state; double* obs; double* act; double* rew;
    // element (step k, row r, env i) of a trajectory array is at k * sk + r * sr + i * si: time-major [T, rows, N] by default,
    // one environment after the other [N, T, rows] with PSH_F_TRAJ_ENV_MAJOR
    // (32-bit: an offset is three 32 x 32 -> 64-bit multiply-adds; the launch rejects arrays whose strides do not fit)
    unsigned int obs_sk, obs_sr, obs_si, act_sk, act_sr, act_si, rew_sk, rew_si;
    unsigned int* queue;   // next unassigned env index (zeroed before the launch): lanes pull work from it
    const double* plate;   // QBB: plate angles from the IK prologue [2, N], or NULL (simplified dynamics)
    const double* tanh_tab;  // FNN: tanh table in global memory, copied to LDS by every workgroup
    int tanh_n;              // its length (fine or coarse table)
    int pop_tiles;           // FNN population mode: MFMA tiles (16 lanes each) per policy task: 1, 2 or 4
    int pop_groups;          // FNN population mode: groups of pop_tiles tiles a wave uses (the lanes beyond stay empty)
    int wave_lanes;          // FNN, one policy: lanes of a wave that take environments (16, 32, 48 or 64; see launch_rollout)
    int cohort;              // LINEAR, recording: this many adjacent lanes (4, 8, 16 or 32) take consecutive environments together (see Lane::turnover)
};

// The task description is read from an LDS copy inside the step loop: as kernel arguments its ~90 constants would be
// pulled into scalar registers for the whole loop and spill.
static_assert(sizeof(psh_task) % 8 == 0 && sizeof(psh_wrappers) % 8 == 0, "staged as doubles");
static_assert(offsetof(RolloutArgs, wrap) == offsetof(RolloutArgs, task) + sizeof(psh_task), "staged as one block");
constexpr int TASK_DOUBLES = (sizeof(psh_task) + sizeof(psh_wrappers)) / 8;
// Fast activations (see k_rollout_fnn): observations up to FNN_OBS_LIMIT against a first layer whose largest absolute
// row sum (+ |bias|) stays below FNN_GAIN_LIMIT keep every pre-activation finite and below 2^40, where the table split of
// the tanh is exact without any clamp.
constexpr double FNN_OBS_LIMIT = 1048576.0, FNN_GAIN_LIMIT = 524288.0;   // 2^20, 2^19
// Behind the packed weight image: [0..3] the per-layer bounds of k_fnn_bounds, [4..11] the largest |weight| of each input
// column of the first layer, [12] its largest |bias| (see k_fnn_bounds), the rest unused.
constexpr int FNN_TAIL = 16;
constexpr double FNN_BOUND_LIMIT = 19.9;   // k_fnn_bounds: the tanh table ends at 20; the margin covers the rounding of the sums
constexpr int TANH8_PAD = (PSH_TANH8_N + 1) / 2 * 2, TANH_TAB_PAD = (PSH_TANH_TAB_N + 1) / 2 * 2, TANH_ODD_PAD = (PSH_TANH_ODD_N + 1) / 2 * 2;

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
    case PSH_NL_RELU: return !(v <= 0.) ? v : 0.;   // a NaN stays a NaN, as in torch.relu
    case PSH_NL_SIGMOID: return psh_sigmoid(v);
    default: return v;
    }
}

// p * v[idx] for a register-resident array.  Written as a chain of products so that the array is never indexed
// dynamically (a select between its elements gets turned into a select between addresses of a scratch copy).
template <int N> PSH_DEV double times_element(double p, const double* v, int idx) {
#pragma unroll
    for (int q = 0; q < N; ++q) p = (idx == q) ? p * v[q] : p;
    return p;
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

// One value of a trajectory.  A plain store: nontemporal stores (measured at 2^21 envs) change nothing for launches
// whose waves write whole 512-byte rows and cost 25-40 % where refilled lanes write pieces of rows, which L2 no longer
// merges.
PSH_DEV void traj_store(double* p, double v) { *p = v; }
// offset (in elements) of (step k, row r, env i) in a trajectory array with the strides sk, sr, si
PSH_DEV unsigned long traj_off(int k, int r, long i, unsigned int sk, unsigned int sr, unsigned int si) {
    return (unsigned long)(unsigned int)k * sk + (unsigned long)(unsigned int)r * sr + (unsigned long)(unsigned int)i * si;
}

// ------------------------------------------------------------------------------------------ one lane = one env
// n standard normals for (seed, env, step, stream): Box-Muller pairs, pair p from counter word 3 = stream + 16 p
template <int N> PSH_DEV void step_normals(uint64_t seed, uint64_t env, uint32_t t, uint32_t stream, double* out) {
#pragma unroll
    for (int p = 0; 2 * p < N; ++p) {
        double ua, ub, sn, cs;
        psh_draw2(seed, env, t, stream + 16u * p, ua, ub);
        const double r = sqrt(-2.0 * psh_log_normal(1.0 - ua));   // (1 - ua lies in [2^-53, 1])
        psh_sincos_moderate(2.0 * PSH_PI * ub, &sn, &cs);  // (the argument lies in [0, 2 pi))
        out[2 * p] = r * cs;
        if (2 * p + 1 < N) out[2 * p + 1] = r * sn;
    }
}

// WRAP = true adds the fused environment wrappers (psh_wrappers) around the simulation; the bare variant carries
// none of that code or state.
// ANORM: the bare lane can apply a lone ActNormWrapper itself (PSH_F_INTERNAL_ACT_NORM)
template <int KIND, bool WRAP, bool FIXT = false, bool ANORM = false> struct Lane {
    using E = Env<KIND>;
    E e;
    double c_max, ret;
    long i;           // env this lane currently runs, -1: none
    int yielded, st, k;   // k: steps taken = SimPyEnv._curr_step (the two always agree here)
    int delay, qpos;  // WRAP: ActDelayWrapper delay of this env, write position of its ring buffer
    double delay_pre[E::A];  // WRAP: the ring-buffer entry the next step will take out, requested a step ahead
    bool done, live;  // live: this lane still has steps to execute
    // the work queue had nothing left for this lane.  An integer on purpose: as a bool the flag lives as a 64-bit lane mask
    // in scalar registers, and in the register-weight population kernels (90 spilled scalar registers) a build was seen in
    // which single lanes of an idle group read it as set: they stayed out of the group's next task (their rollouts never
    // ran) or, when the group's first lane was among them, the others took the queue position of an inactive lane (0) again
    // and again and the launch never ended.  With the flag in a vector register the same build runs to the end.
    int exhausted;
    unsigned int chunk_next, chunk_end;  // wave-uniform: the envs of this wave's current chunk not handed out yet
    int pad_row;      // cohort mode: the next trajectory row this finished lane pads while its neighbours still run, -1: none yet
    bool rec_alias;   // WRAP: the array the simulation receives is one the rollout has recorded

    static constexpr int QUEUE_CHUNK = 8;
    PSH_DEV void init() { i = -1; live = false; exhausted = 0; chunk_next = chunk_end = 0; pad_row = -1; }

    // Lanes pull environments from a global queue: a lane whose env is finished stores its results and takes the next
    // unassigned env, so a wave keeps all 64 lanes busy although episodes end at different steps.  Which lane runs
    // an env has no influence on its results.
    // BATCH > 1: the exchange is deferred until BATCH lanes of the wave wait for it (or none is running any more).
    // Storing and loading an env costs about as much as one step of a LinearPolicy rollout and the whole wave pays
    // for it whenever a single lane needs it; with short episodes that is most steps.
    // COHORT (a.cohort = 16, launches that record time-major trajectories): the lanes of an aligned group of 16 take 16
    // consecutive environments together and give them back together, when the last of them has ended.  The group then
    // stays in lock step, so every trajectory store of it is one whole 128-byte line of a row (a lane whose environment has
    // ended pads its rows with NaN until the group is done, see pre_policy): environments that end early otherwise leave
    // their neighbours writing 8- to 120-byte parts of a line, which the memory system turns into read-modify-write
    // cycles.  Lanes idle for max - own length of their group; what that costs is less than what the partial lines did.
    template <int BATCH = 1> PSH_DEV void turnover(const RolloutArgs& a, const psh_task& task, const psh_wrappers& wrap) {
        const bool cohort = BATCH > 1 && a.cohort != 0;
        bool octet_done = true;
        if (cohort) {
            const int cs = a.cohort;   // 4, 8 or 16 lanes
            octet_done = ((__ballot(live) >> ((threadIdx.x & 63) & ~(cs - 1))) & ((1ull << cs) - 1ull)) == 0;
        }
        if (BATCH > 1) {
            const bool want = (i >= 0 && !live && octet_done) || (i < 0 && !exhausted && octet_done);
            if (__popcll(__ballot(want)) < BATCH && __any(live)) return;
        }
        if (i >= 0 && !live && octet_done) { store(a, wrap, 0, cohort && pad_row >= 0); i = -1; }
        // The queue is drawn from in chunks of 8 consecutive envs per wave: row r of the inputs of such a chunk is one
        // 64-byte line, fetched once and still in L2 when the wave's next free lane takes the neighbouring env (single
        // envs handed out across all waves cost a line per 8-byte value: 1.8 GB of HBM reads per launch for 0.24 GB of
        // inputs); the 8-byte results of a chunk likewise merge into whole lines.  All lanes of the wave get here
        // together, so the chunk bounds are wave-uniform.
        bool want = i < 0 && !exhausted && octet_done, fresh = false;
        unsigned long long m = __ballot(want);
        while (m) {
            if (chunk_next == chunk_end) {
                // one draw covers every lane that is waiting, in whole lines (the first fill of a wave takes 64 at once:
                // eight draws each would queue eight times as many atomics of all waves on one address)
                // (BATCH > 1, the LinearPolicy kernels with their short steps: at least 4 BATCH envs per draw, handed out over
                // the wave's next exchanges.  With episodes of 70 steps a million envs drawn 8 at a time are 131 072 atomics
                // on one address, which L2 serialises at about 10 ns each: more than the rollouts themselves take)
                unsigned int size = ((unsigned int)__popcll(m) + QUEUE_CHUNK - 1) / QUEUE_CHUNK * QUEUE_CHUNK;
                if (BATCH > 1 && size < 4u * BATCH) size = 4u * BATCH;
                unsigned int base = 0;
                if ((threadIdx.x & 63) == 0) base = atomicAdd(a.queue, size);
                base = __builtin_amdgcn_readfirstlane(base);
                const unsigned int n = (unsigned int)a.n_envs;
                chunk_next = base < n ? base : n;
                chunk_end = base + size < n ? base + size : n;
                if (chunk_next == chunk_end) {  // the queue is empty
                    if (want) exhausted = 1;
                    break;
                }
            }
            const unsigned int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
            const unsigned int avail = chunk_end - chunk_next;
            if (want && rank < avail) {
                i = (long)(chunk_next + rank);
                want = false;
                fresh = true;
            }
            const unsigned int asked = (unsigned int)__popcll(m);
            chunk_next += asked < avail ? asked : avail;
            m = __ballot(want);
        }
        if (fresh) load(a, task, wrap);  // once for all lanes that got an env, whatever chunks they came from
    }

    // Population mode of the FNN kernel: the 16 lanes of an MFMA tile share the A operand, i.e. the policy.  The lanes
    // of a wave form groups of G = a.pop_tiles tiles (16, 32 or 64 lanes); a group runs one policy at a time, whose
    // weight image the wave keeps in LDS.  Task q = (policy q / chunks, rollouts 16 G (q % chunks) ... of that policy);
    // a group takes the next task when none of its lanes has an env.  Returns the policy of this lane's group;
    // *fresh has the bit of a group's first lane set if the group took a new task (the caller loads its weight image).
    //
    // Must be called with all 64 lanes of the wave active.  Everything a group decides is computed from ballots over the
    // whole wave, so the lanes of a group cannot come to different conclusions whatever their own flags hold: which
    // groups are idle, which of them still draw from the queue, and how many queue positions the wave takes (ONE atomic
    // per wave, issued by lane 0 under a wave-uniform condition; group g of the taking groups gets position base + its
    // rank).  An earlier form tested the per-lane `exhausted` flag inside the divergent region of each group and broadcast
    // the queue position with a shuffle from the group's first lane: a single lane with a wrong flag then stayed out of
    // its group's tasks, and a wrong flag in the first lane made the others read position 0 from an inactive lane for
    // ever.  Here a group counts as exhausted only if ALL its lanes say so (a draw from the empty queue costs nothing and
    // sets the flag in every lane again), and a disagreement is remembered in `fault` (kept by the caller, per lane) and
    // goes into the status word of every rollout the lane stores from then on (PSH_ST_INTERNAL).
    PSH_DEV int turnover_group(const RolloutArgs& a, const psh_task& task, const psh_wrappers& wrap, int pol,
                               unsigned long long* fresh, int& fault) {
        if (i >= 0 && !live) { store(a, wrap, fault); i = -1; }
        const int lane = threadIdx.x & 63, GL = 16 * a.pop_tiles;
        const int first = lane & ~(GL - 1);
        const unsigned long long group = (GL == 64 ? ~0ull : ((1ull << GL) - 1ull)) << first;
        const unsigned long long busy = __ballot(i >= 0), exh = __ballot(exhausted != 0);
        const unsigned long long exh_g = exh & group;
        if (exh_g != 0 && exh_g != group) fault |= PSH_ST_INTERNAL;
        const bool take = (busy & group) == 0 && exh_g != group && first < GL * a.pop_groups;
        const unsigned long long heads = __ballot(take && lane == first);
        *fresh = 0;
        if (heads == 0) return pol;  // (wave-uniform)
        unsigned int base = 0;
        if (lane == 0) base = atomicAdd(a.queue, (unsigned int)__popcll(heads));
        base = __builtin_amdgcn_readfirstlane(base);
        bool got = false;
        if (take) {
            const int R = a.pol.policy_group, chunks = (R + GL - 1) / GL;
            const unsigned int q = base + (unsigned int)__popcll(heads & ((1ull << first) - 1ull));
            if (q < (unsigned int)(a.pol.n_policies * chunks)) {
                pol = (int)(q / chunks);
                const int r = GL * (int)(q % chunks) + (lane - first);
                if (r < R && (long)pol * R + r < (long)a.n_envs) { i = (long)pol * R + r; load(a, task, wrap); }
                got = lane == first;
            } else {
                exhausted = 1;
            }
        }
        *fresh = __ballot(got);
        return pol;
    }

    PSH_DEV void load(const RolloutArgs& a, const psh_task& task, const psh_wrappers& wrap) {
        bool bad_index;
        const long col = dp_column(a.dp_index, a.ld_dp, i, &bad_index);
        if (a.pol.policy_index) bad_index = bad_index || (unsigned)a.pol.policy_index[i] >= (unsigned)a.pol.n_policies;
        const DpCol p{a.domain_param, a.ld_dp, col};
        e.setup(p);
        c_max = task_c_max<KIND, FIXT>(e, task);
        double init[E::S];
        for (int r = 0; r < E::S; ++r) init[r] = (r < a.init_dim) ? a.init_state[r * a.ld_init + i] : 0.;
        if constexpr (KIND == PSH_ENV_QBB) {
            e.reset(init, a.init_dim, true);  // plate angles come from the IK prologue kernel
            if (a.plate) { e.plate[0] = a.plate[i]; e.plate[1] = a.plate[(long)a.n_envs + i]; }
        } else {
            e.reset(init, a.init_dim, false);
        }
        ret = 0.; yielded = 0; k = 0; done = false; pad_row = -1;
        st = bad_index ? PSH_ST_BAD_INDEX : 0;  // not run: stored with length 0 at the next turnover
        live = a.max_steps > 0 && !bad_index;
        if constexpr (WRAP) {
            bool has_delay = false, new_array = false;
            for (int j = 0; j < wrap.n_ops; ++j) {
                has_delay = has_delay || wrap.ops[j] == PSH_WRAP_ACT_DELAY;
                new_array = new_array || wrap.ops[j] == PSH_WRAP_ACT_NORM || wrap.ops[j] == PSH_WRAP_ACT_NOISE;
            }
            delay = 0; qpos = 0;
#pragma unroll
            for (int r = 0; r < E::A; ++r) delay_pre[r] = 0.;
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
                // the step made opaque inside the branch: the draws do not depend on j, and as loop-invariant, speculatable
                // arithmetic they would be hoisted in front of the loop over the wrappers and run on every step of every launch
                uint32_t tt = t;
                asm volatile("" : "+v"(tt));
                double n[E::O];
                step_normals<E::O>(wrap.noise_seed, wrap.env_offset + (uint64_t)i, tt, PSH_STREAM_OBS_NOISE, n);
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
    // NAN_CHECK = false is for a policy whose action is a NaN whenever an observation is one (a weighted sum of ALL
    // observations): post_policy's test of the action then ends the rollout at the same step with the same status and
    // the same rows recorded, and the comparisons per observation are saved.
    // pad (cohort mode): this lane's environment has ended while others of its group still run.  It goes through the same
    // stores as they do and writes its share of the rows they write -- its final observation first (row k, pad_row == k),
    // NaN afterwards -- so that the group's stores stay whole lines and the wave runs ONE instruction stream per step
    // (as a pass of its own for the ended lanes the padding cost BallOnBeam's recording launch a sixth of its time).
    template <bool NAN_CHECK = true>
    PSH_DEV void pre_policy(const RolloutArgs& a, const psh_wrappers& wrap, double* obs, bool pad = false) {
        if (WRAP && a.wrap_obs_ops) wrapped_obs(a, wrap, (uint32_t)k, obs); else e.observe(obs);
        const int Oe = WRAP ? a.obs_dim : E::O;
        bool nan = false;
        if constexpr (NAN_CHECK) {
#pragma unroll
            for (int r = 0; r < E::O; ++r) nan = nan || (r < Oe && obs[r] != obs[r]);
        }
        if (a.obs) {
            const int row = pad ? pad_row : k;
            const bool blank = pad && pad_row != k;
            if (!pad || row <= a.max_steps) {   // (a stepping lane's row is k < max_steps)
#pragma unroll
                for (int r = 0; r < E::O; ++r)
                    if (r < Oe) traj_store(a.obs + traj_off(row, r, i, a.obs_sk, a.obs_sr, a.obs_si), blank ? __builtin_nan("") : obs[r]);
            }
        }
        if (nan && !pad) { st |= PSH_ST_NAN; live = false; }
    }

    // PADK: the caller's lanes may be padding (see pre_policy); the rows of the action and the reward are then written
    // here, by stepping and padding lanes alike.  Without it step_env records them itself, as it goes.
    template <bool PADK = false>
    PSH_DEV void post_policy(const RolloutArgs& a, const psh_task& task, const psh_wrappers& wrap, double* act, bool pad = false) {
        double rew;
        int cs = k;   // the environment's step counter: k + 1 after env_step
        bool stepped = false;
        if constexpr (PADK) {
            double act_rec[E::A];   // what this step's row of the action receives
            rew = __builtin_nan("");
#pragma unroll
            for (int r = 0; r < E::A; ++r) act_rec[r] = __builtin_nan("");
            if (!pad) step_env<false>(a, task, wrap, act, act_rec, rew, cs, stepped);
            const int row = pad ? pad_row : k;
            if (pad ? row < a.max_steps : stepped) {
                if (a.act) {
#pragma unroll
                    for (int r = 0; r < E::A; ++r) traj_store(a.act + traj_off(row, r, i, a.act_sk, a.act_sr, a.act_si), act_rec[r]);
                }
                if (a.rew) traj_store(a.rew + traj_off(row, 0, i, a.rew_sk, 0u, a.rew_si), rew);
            }
            if (pad) { pad_row += 1; return; }
        } else {
            step_env<true>(a, task, wrap, act, nullptr, rew, cs, stepped);
        }
        if (!stepped) return;
        ret += rew;
        k = cs;
        // a NaN state has failed its bounds and the last step has timed out: both arrive with `done` set
        if (done && ((a.flags & PSH_F_STOP_ON_DONE) || (st & (PSH_ST_NAN | PSH_ST_TIMEOUT)))) live = false;
        if constexpr (WRAP) {
            if (live && a.bernoulli > 0.) {  // stochastic reset (sampling/rollout.py:248-254)
                double u0, u1;
                psh_draw2(wrap.noise_seed, wrap.env_offset + (uint64_t)i, (uint32_t)(k - 1), PSH_STREAM_BERNOULLI, u0, u1);
                if (u0 < a.bernoulli) { st |= PSH_ST_BERNOULLI; live = false; }
            }
        }
    }

    // The part of a step between the policy's output and the recording of its rows: exploration noise, NaN test of the
    // action, action wrappers, SimPyEnv.step.  stepped stays false if the action was a NaN (nothing recorded for the step).
    // STORE: the rows of the action and the reward are written here; otherwise the action's comes back in act_rec.
    template <bool STORE>
    PSH_DEV void step_env(const RolloutArgs& a, const psh_task& task, const psh_wrappers& wrap, double* act, double* act_rec,
                          double& rew, int& cs, bool& stepped) {
        if constexpr (WRAP) {
            // StochasticActionExplStrat.forward (exploration/stochastic_action.py:42-68): the exploratory action
            // replaces the policy's output for everything that follows
            if (a.pol.expl_kind == PSH_EXPL_NORMAL) {
                double n[E::A];
                step_normals<E::A>(wrap.noise_seed, wrap.env_offset + (uint64_t)i, (uint32_t)k, PSH_STREAM_EXPL, n);
#pragma unroll
                for (int r = 0; r < E::A; ++r) act[r] = (act[r] + a.pol.expl_mean[r]) + n[r] * a.pol.expl_scale[r];
            } else if (a.pol.expl_kind == PSH_EXPL_UNIFORM) {
                double u[2];
                psh_draw2(wrap.noise_seed, wrap.env_offset + (uint64_t)i, (uint32_t)k, PSH_STREAM_EXPL, u[0], u[1]);
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
        stepped = true;
        if constexpr (WRAP) {
            // EnvWrapperAct.step (environment_wrappers/base.py:168-174), outermost wrapper first
            double inner[E::A];
#pragma unroll
            for (int r = 0; r < E::A; ++r) inner[r] = act[r];
            for (int j = 0; j < wrap.n_ops; ++j) {
                const int op = wrap.ops[j];
                if (op == PSH_WRAP_ACT_DELAY && delay > 0) {  // FIFO as a ring: the slot written now is read `delay` steps on
                    // The entry taken out at step k was requested at the end of step k - 1 (delay_pre): the ring lives in
                    // global memory, and a launch with a wave or two per SIMD would wait a memory round trip per step for it.
                    // (a wave's rings are one contiguous block [PSH_MAX_DELAY][A][64]: the rows a wave walks through stay
                    // within a page and a few dozen lines, whatever the size of the grid)
                    const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
                    double* const ring = a.delay_q + wave * (long)(PSH_MAX_DELAY * E::A * 64) + (threadIdx.x & 63);
                    double now[E::A];
#pragma unroll
                    for (int r = 0; r < E::A; ++r) {
                        ring[(qpos * E::A + r) * 64] = inner[r];
                        now[r] = inner[r];
                        inner[r] = (k >= delay) ? delay_pre[r] : 0.;
                    }
                    qpos = (qpos + 1 == delay) ? 0 : qpos + 1;
                    if (k + 1 >= delay) {  // what step k + 1 takes out was written at step k + 1 - delay
#pragma unroll
                        for (int r = 0; r < E::A; ++r)
                            delay_pre[r] = (delay == 1) ? now[r] : ring[(qpos * E::A + r) * 64];
                    }
                } else if (op == PSH_WRAP_ACT_NORM) {
                    double lo[E::A], hi[E::A];
                    e.act_bounds(lo, hi);
#pragma unroll
                    for (int r = 0; r < E::A; ++r) inner[r] = lo[r] + (inner[r] + 1) * (hi[r] - lo[r]) / 2;
                } else if (op == PSH_WRAP_ACT_NOISE) {
                    uint32_t tt = (uint32_t)k;  // opaque inside the branch, as in wrapped_obs
                    asm volatile("" : "+v"(tt));
                    double n[E::A];
                    step_normals<E::A>(wrap.noise_seed, wrap.env_offset + (uint64_t)i, tt, PSH_STREAM_ACT_NOISE, n);
#pragma unroll
                    for (int r = 0; r < E::A; ++r) inner[r] = inner[r] + (n[r] * wrap.act_noise_std[r] + wrap.act_noise_mean[r]);
                }
            }
            rew = env_step<KIND, FIXT>(e, task, c_max, cs, yielded, inner, a.dt, a.max_steps, false, simplified, done, st);
            if constexpr (!STORE) {
#pragma unroll
                for (int r = 0; r < E::A; ++r) act_rec[r] = act[r];
            }
            if (a.act) {
                if constexpr (STORE) {
#pragma unroll
                    for (int r = 0; r < E::A; ++r) traj_store(a.act + traj_off(k, r, i, a.act_sk, a.act_sr, a.act_si), act[r]);
                }
                // ActDelayWrapper queues the caller's array objects: without an array-creating wrapper the simulation
                // wrote its dead zone (QBB) into the action recorded `delay` steps ago
                // (with no delay that is this step's own row: after the store above, or into what the caller stores)
                if (!STORE && rec_alias && delay == 0) {
#pragma unroll
                    for (int r = 0; r < E::A; ++r) act_rec[r] = inner[r];
                } else if (rec_alias && k - delay >= 0) {
#pragma unroll
                    for (int r = 0; r < E::A; ++r) traj_store(a.act + traj_off(k - delay, r, i, a.act_sk, a.act_sr, a.act_si), inner[r]);
                }
            }
        } else {
            // (a lone ActNormWrapper, environment_wrappers/action_normalization.py:10-22: the simulation receives a new
            // array, so what it writes into its action -- QBallBalancer's dead zone -- does not reach the recorded one)
            const bool act_norm = ANORM && (a.flags & PSH_F_INTERNAL_ACT_NORM) != 0;
            double inner[E::A];
#pragma unroll
            for (int r = 0; r < E::A; ++r) inner[r] = act[r];
            if (act_norm) {
                asm volatile("");   // a wave-uniform branch, not four instructions and two selects on every step
                double lo[E::A], hi[E::A];
                e.act_bounds(lo, hi);
#pragma unroll
                for (int r = 0; r < E::A; ++r) inner[r] = lo[r] + (inner[r] + 1) * (hi[r] - lo[r]) / 2;
            }
            rew = env_step<KIND, FIXT>(e, task, c_max, cs, yielded, inner, a.dt, a.max_steps, false, simplified, done, st);
            if constexpr (STORE) {
                if (a.act) {
#pragma unroll
                    for (int r = 0; r < E::A; ++r)
                        traj_store(a.act + traj_off(k, r, i, a.act_sk, a.act_sr, a.act_si), act_norm ? act[r] : inner[r]);
                }
            } else {
#pragma unroll
                for (int r = 0; r < E::A; ++r) act_rec[r] = act_norm ? act[r] : inner[r];
            }
        }
        if constexpr (STORE) {
            if (a.rew) traj_store(a.rew + traj_off(k, 0, i, a.rew_sk, 0u, a.rew_si), rew);
        }
    }

    PSH_DEV void store(const RolloutArgs& a, const psh_wrappers& wrap, int extra_status = 0, bool final_obs_written = false) {
        if (a.obs && !final_obs_written) {  // final observation
            double obs[E::O];
            if (WRAP && a.wrap_obs_ops) wrapped_obs(a, wrap, (uint32_t)k, obs); else e.observe(obs);
            const int Oe = WRAP ? a.obs_dim : E::O;
#pragma unroll
            for (int r = 0; r < E::O; ++r)
                if (r < Oe) traj_store(a.obs + traj_off(k, r, i, a.obs_sk, a.obs_sr, a.obs_si), obs[r]);
        }
        a.ret[i] = ret;
        a.length[i] = k;
        a.status[i] = st | extra_status;
        if (a.final_state) {
#pragma unroll
            for (int r = 0; r < E::S; ++r) a.final_state[(long)r * a.n_envs + i] = e.s[r];
        }
    }
};

// ------------------------------------------------------------------------------------- LinearPolicy rollouts
// PER_ENV = false: one weight set, staged in LDS and broadcast; true: env i reads row policy_index[i] (or, in
// population mode, row i / policy_group) of params.
// SIMPLE: the FeatureStack is [identity_feat] or [const_feat, identity_feat] (the usual LinearPolicy): a dot product on
// the observation with the weights in registers; the variant carries none of the feature functions' code, whose
// dispatch costs scalar instructions on every step (the scalar unit of a CU, shared by its four SIMDs, is what bounds
// the cheap simulations).
// FIXT: the task has the default shape of this simulation (TaskShape), decided at compile time.
// PAD: a launch that records time-major trajectories, whose lanes run in cohorts (Lane::turnover) and pad the rows of
// environments that have ended (Lane::pre_policy); a kernel of its own, so that the others carry none of it.
template <int KIND, bool PER_ENV, bool WRAP, bool SIMPLE = false, bool FIXT = false, bool PAD = false>
// (waves per SIMD the register allocation aims at: the wrapped kernels land a few registers above the 168 that three
// waves per SIMD allow unless told to stay below, at the price of a few spills -- QQube behind ActNormWrapper, 2^20
// rollouts: 8.5 -> 7.5 ms; the per-environment-policy variants mostly run populations of a wave or two per SIMD, which
// gain nothing from a third wave and lose 12 % to the spills)
__global__ __launch_bounds__(256, KIND == PSH_ENV_QBB ? 1 : (WRAP && !PER_ENV ? 3 : 2)) void k_rollout_linear(const RolloutArgs a) {
    using E = Env<KIND>;
    constexpr int W_DOUBLES = PSH_MAX_ACT * PSH_MAX_LIN_FEAT;
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
    Lane<KIND, WRAP, FIXT, !WRAP> L;
    L.init();
    const double* w = lds_w;
    long w_env = -1;
    // SIMPLE: the weights live in registers, [r][0] = what the const feature contributes (0 without one), [r][1 + q] =
    // the weight of observation q
    double wreg[SIMPLE ? E::A : 1][SIMPLE ? E::O + 1 : 1];
    auto load_wreg = [&](const double* src) {
        if constexpr (SIMPLE) {
            const int off = a.pol.feats[0] == PSH_FEAT_CONST ? 1 : 0;
#pragma unroll
            for (int r = 0; r < E::A; ++r) {
                wreg[r][0] = off ? 0. + src[r * F] : 0.;
#pragma unroll
                for (int q = 0; q < E::O; ++q) wreg[r][1 + q] = (!WRAP || q < Oe) ? src[r * F + off + q] : 0.;
            }
        }
    };
    if (SIMPLE && !PER_ENV) load_wreg(lds_w);
    for (;;) {
        asm volatile("" ::: "memory");  // keep the LDS constants out of registers across iterations
        // as long as every lane of the wave is in the middle of an episode there is nothing to exchange
        if (!__all(L.live)) {
            L.template turnover<8>(a, task, wrap);
            if (!__any(L.i >= 0)) break;  // every lane of the wave is out of work
            // (a lane without an env is not live; one whose env has ended while its cohort still runs pads its rows)
            if (!L.live && !(PAD && L.i >= 0)) continue;
            if (PER_ENV && w_env != L.i) {
                const long row = a.pol.policy_index ? (long)a.pol.policy_index[L.i]
                                                    : (a.pol.policy_group > 0 ? L.i / a.pol.policy_group : 0);
                w = a.pol.params + row * a.pol.num_param;
                w_env = L.i;
                load_wreg(w);
            }
        }
        double obs[E::O], act[E::A];
        const bool pad = PAD && !L.live;
        if (pad && L.pad_row < 0) L.pad_row = L.k;
        L.template pre_policy<!SIMPLE>(a, wrap, obs, pad);
        if (!SIMPLE && !L.live && !pad) continue;
        if constexpr (SIMPLE) {
#pragma unroll
            for (int r = 0; r < E::A; ++r) act[r] = wreg[r][0];
#pragma unroll
            for (int q = 0; q < E::O; ++q) {
                if (WRAP && q >= Oe) break;
#pragma unroll
                for (int r = 0; r < E::A; ++r) act[r] += wreg[r][1 + q] * obs[q];
            }
            L.template post_policy<PAD>(a, task, wrap, act, pad);
            continue;
        }
#pragma unroll
        for (int r = 0; r < E::A; ++r) act[r] = 0.;
        int n = 0;
        for (int f = 0; f < a.pol.n_feats; ++f) {
            const int id = a.pol.feats[f];
            if (id == PSH_FEAT_CONST) {
#pragma unroll
                for (int r = 0; r < E::A; ++r) act[r] += w[r * F + n];
                n += 1;
            } else if (id == PSH_FEAT_MULT) {  // MultFeat (policies/features.py:143-167)
                const int arg = a.pol.feat_arg[f];
                double v = 1.0;  // 1 * x is exact
                for (int j = 0; j < a.pol.feat_size[f]; ++j) v = times_element<E::O>(v, obs, (arg >> (4 * j)) & 15);
#pragma unroll
                for (int r = 0; r < E::A; ++r) act[r] += w[r * F + n] * v;
                n += 1;
            } else if (id == PSH_FEAT_RFF) {  // RandFourierFeat (policies/features.py:203-219)
                const int D = a.pol.feat_size[f];
                const double* data = a.pol.feat_data + a.pol.feat_off[f];  // wave-uniform: scalar loads
                const double scale = data[0];
                const double* shift = data + 1;
                const double* freq = data + 1 + D;
                for (int j = 0; j < D; ++j) {
                    double z = 0.;
#pragma unroll
                    for (int q = 0; q < E::O; ++q)
                        if (!WRAP || q < Oe) z = fma(obs[q], freq[j * Oe + q], z);
                    double sn, cs;
                    psh_sincos(z + shift[j], &sn, &cs);
                    const double v = scale * cs;
#pragma unroll
                    for (int r = 0; r < E::A; ++r) act[r] += w[r * F + n + j] * v;
                }
                n += D;
            } else if (id == PSH_FEAT_RBF) {  // RBFFeat (policies/features.py:278-303)
                // sum_k w_k e_k / norm instead of sum_k w_k (e_k / norm): the activations need not be kept
                const int K = a.pol.feat_size[f];
                const bool state_wise = a.pol.feat_arg[f] != 0;
                const double* scale = a.pol.feat_data + a.pol.feat_off[f];
                const double* cen = scale + Oe;
                double total = 0., joint[E::A];
#pragma unroll
                for (int r = 0; r < E::A; ++r) joint[r] = 0.;
#pragma unroll
                for (int q = 0; q < E::O; ++q) {
                    if (WRAP && q >= Oe) break;
                    double norm = 0., acc[E::A];
#pragma unroll
                    for (int r = 0; r < E::A; ++r) acc[r] = 0.;
                    for (int j = 0; j < K; ++j) {
                        const double dl = obs[q] - cen[j * Oe + q];
                        const double e = psh_exp(-scale[q] * (dl * dl));
                        norm += e;
#pragma unroll
                        for (int r = 0; r < E::A; ++r) acc[r] += w[r * F + n + q * K + j] * e;
                    }
                    if (state_wise) {
                        norm = (norm > 1e-8) ? norm : 1.;  // normalize() of utils/normalizing.py:87-105
#pragma unroll
                        for (int r = 0; r < E::A; ++r) act[r] += acc[r] / norm;
                    } else {
                        total += norm;
#pragma unroll
                        for (int r = 0; r < E::A; ++r) joint[r] += acc[r];
                    }
                }
                if (!state_wise) {
                    total = (total > 1e-8) ? total : 1.;
#pragma unroll
                    for (int r = 0; r < E::A; ++r) act[r] += joint[r] / total;
                }
                n += K * Oe;
            } else {
#pragma unroll
                for (int q = 0; q < E::O; ++q) {
                    if (WRAP && q >= Oe) break;
                    // opaque per iteration: otherwise every feature function of every observation is loop-invariant
                    // in f and gets hoisted in front of the loop, i.e. evaluated on every step whatever the stack holds
                    double x = obs[q];
                    asm volatile("" : "+v"(x));
                    const double v = feat_apply(id, x);
#pragma unroll
                    for (int r = 0; r < E::A; ++r) act[r] += w[r * F + n + q] * v;
                }
                n += Oe;
            }
        }
        L.template post_policy<PAD>(a, task, wrap, act, pad);
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
// Packed weight image (doubles), HT = HP/16, KS = HP/4, K0 = O/4 whole groups of 4 inputs, REM = O%4 inputs left over:
//   [HT*K0 frags of 64][Wrem: REM*HP][b0: HP]  then per further hidden layer [HT*KS frags of 64][b: HP]  then
//   [Wout: A*HP][bout: 2]
// Dynamic LDS: [weight image (POP: one per wave)][a staging area of OP*64 per wave, OP = O padded to 4][tanh table][task].  FINE selects the fine tanh table (40 KB);
// the widest, deepest networks leave no room for it and run with the coarse one.
typedef double v4d __attribute__((ext_vector_type(4)));

// x(lane) + x(lane ^ 16) + x(lane ^ 32) + x(lane ^ 48) in every lane, in the order (x + x^16) + (that of lane ^ 32):
// v_permlane16_swap exchanges the odd 16-lane rows of its first operand with the even rows of its second, so swapping a
// value with a copy of itself leaves [r0 r0 r2 r2] and [r1 r1 r3 r3]; v_permlane32_swap does the same with the halves.
PSH_DEV double sum_over_lane_groups(double x) {
    unsigned lo = (unsigned)__double2loint(x), hi = (unsigned)__double2hiint(x);
    auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    const double s = __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
    lo = (unsigned)__double2loint(s); hi = (unsigned)__double2hiint(s);
    a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
}

// x of the lane N places further down in this lane's row of 16 (cyclically): two v_mov_b32 with a row_ror modifier
template <int N> PSH_DEV double rotate_in_row(double x) {
    static_assert(N > 0 && N < 16, "row_ror:1 .. row_ror:15");
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(x), 0x120 + N, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(x), 0x120 + N, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

// all N values in registers at this point of the program (so that the loads that produce them are all in flight before
// the first of them is used)
template <int N> PSH_DEV void hold_together(double* v) {
    static_assert(N == 2 || N == 4 || N == 8 || N == 16, "one statement per size");
    if constexpr (N == 2) asm volatile("" : "+v"(v[0]), "+v"(v[1]));
    if constexpr (N == 4) asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));
    if constexpr (N == 8)
        asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
    if constexpr (N == 16)
        asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]),
                          "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]));
}

// y[0..N) are the pre-activations this lane holds for ONE environment; all of them feed every unit of the next
// layer, so a NaN in any of them has to end up in the action.  The table tanh loses NaNs (its clamp works on
// the high word), so the largest |y| high word of the layer is tracked (one 32-bit max per element) and a NaN is put
// back into y[0] once per layer.
// BOUNDED = 1: every y is known to be finite and below 2^40 in magnitude (the caller has checked what feeds the layer):
// no clamp of |y| and no NaN bookkeeping, only the table index is limited.  BOUNDED = 2: every |y| is known to stay
// below 20 as well (see FNN_BOUND_LIMIT): the index needs no limit either.  BOUNDED = 3, 4: as 1, 2 on a table that
// continues to negative nodes (`tanh_tab` points at node 0): no |y| and no sign transfer (psh_tanh8_index_signed).
template <int N, bool FINE, int BOUNDED = 0> PSH_DEV void nl_apply_all(int nl, double* y, const double* tanh_tab) {
    if (nl == PSH_NL_TANH) {
        unsigned hi_max = 0;
        if (FINE && BOUNDED >= 3) {
            constexpr int CH = N < 8 ? N : 8;
#pragma unroll
            for (int m0 = 0; m0 < N; m0 += CH) {
                double d[CH], T[CH];
#pragma unroll
                for (int m = 0; m < CH; ++m) {
                    int j = psh_tanh8_index_signed(y[m0 + m], &d[m]);
                    if (BOUNDED == 3) j = j < -(PSH_TANH8_N - 1) ? -(PSH_TANH8_N - 1) : (j > PSH_TANH8_N - 1 ? PSH_TANH8_N - 1 : j);
                    T[m] = tanh_tab[j];
                }
#pragma unroll
                for (int m = 0; m < CH; ++m) y[m0 + m] = psh_tanh8_finish_signed(d[m], T[m]);
            }
            return;
        }
        if (FINE && BOUNDED) {
            constexpr int CH = N < 8 ? N : 8;
#pragma unroll
            for (int m0 = 0; m0 < N; m0 += CH) {
                double d[CH], T[CH];
#pragma unroll
                for (int m = 0; m < CH; ++m)
                    T[m] = tanh_tab[BOUNDED == 2 ? psh_tanh8_index_free(y[m0 + m], &d[m]) : psh_tanh8_index_bounded(y[m0 + m], &d[m])];
#pragma unroll
                for (int m = 0; m < CH; ++m) y[m0 + m] = psh_tanh8_finish(y[m0 + m], d[m], T[m]);
            }
            return;
        }
        if (FINE) {
            // all table loads first, then the arithmetic: the LDS latency is paid once per layer, not per element
            constexpr int CH = N < 8 ? N : 8;  // chunked: d and T of a whole layer would not fit the register file
#pragma unroll
            for (int m0 = 0; m0 < N; m0 += CH) {
                double d[CH], T[CH];
#pragma unroll
                for (int m = 0; m < CH; ++m) {
                    unsigned hi;
                    T[m] = tanh_tab[psh_tanh8_index(y[m0 + m], &d[m], &hi)];
                    hi_max = hi > hi_max ? hi : hi_max;
                }
#pragma unroll
                for (int m = 0; m < CH; ++m) y[m0 + m] = psh_tanh8_finish(y[m0 + m], d[m], T[m]);
            }
        } else {
#pragma unroll
            for (int m = 0; m < N; ++m) {
                unsigned hi;
                y[m] = psh_tanh_tab(y[m], tanh_tab, &hi);
                hi_max = hi > hi_max ? hi : hi_max;
            }
        }
        y[0] = hi_max > PSH_HI_INF ? __builtin_nan("") : y[0];
    } else if (nl == PSH_NL_SIGMOID) {
#pragma unroll
        for (int m = 0; m < N; ++m) y[m] = psh_sigmoid(y[m]);
    } else if (nl == PSH_NL_RELU) {
#pragma unroll
        for (int m = 0; m < N; ++m) y[m] = !(y[m] <= 0.) ? y[m] : 0.;   // a NaN stays a NaN, as in torch.relu
    }
}

// Sends the whole state of this lane (a trivially copyable struct) to lane `dst`; `dst` must be a permutation of the
// wave's lanes.  Which lane runs an environment has no influence on its results.
template <class T> PSH_DEV void move_lane_state(T& v, int dst) {
    static_assert(std::is_trivially_copyable<T>::value, "moved as raw words");
    constexpr int W = (sizeof(T) + 3) / 4;
    int w[W];
#pragma unroll
    for (int j = 0; j < W; ++j) w[j] = 0;
    __builtin_memcpy(w, &v, sizeof(T));
#pragma unroll
    for (int j = 0; j < W; ++j) w[j] = __builtin_amdgcn_ds_permute(dst << 2, w[j]);
    __builtin_memcpy(&v, w, sizeof(T));
}

// TANH: every hidden layer uses tanh, known at compile time.  With the nonlinearity chosen at run time the arms of
// the switch meet again behind each layer and the register allocator moves all 16 activations of a tile into a common
// place on every arm (32 v_mov_b64 per tile and layer, 3 % of the step); the usual all-tanh network does not pay that.
//
// WREG (population mode, two hidden layers of up to 64 units): the 64 x 64 weights between the hidden layers, 64 MFMA
// fragments = 128 registers per lane, stay in the REGISTERS of the wave that runs the policy (the A operand of an MFMA
// is a register anyway) instead of LDS.  Four images of 39 KB do not fit into a CU's LDS next to a tanh table, so the
// LDS version keeps three waves per CU, one SIMD idle; without that block an image is 6 KB, one wave per SIMD runs
// its policies (256 threads per workgroup leave a wave all 512 registers).  WREG = 1: the whole wave runs one policy
// (more than 32 rollouts per policy); WREG = 2: tiles 0 and 1 of the wave run one policy each, two register sets, the
// other 32 lanes stay empty -- eight policies resident per CU, and the environment step of a wave serves two of them.
// WREG = 3: as WREG = 2 for up to 8 rollouts per policy, where half of the 16 columns of a 16x16x4 tile would be empty.
// The layers run on v_mfma_f64_4x4x4_4b_f64 instead: four independent 4 x 4 x 4 blocks per instruction at the same
// rate per multiply-add (16 cycles), here four groups of 4 hidden units against the same 4 rollouts, so a layer costs
// two column groups of 4 instead of one of 16 -- half the MFMA time and half the activations.  WREG = 4: one column
// group, for up to 4 rollouts per policy.  WREG = 5: as WREG = 3 with ONE policy per wave and two waves per SIMD (workgroups
// of 8 waves, 256 registers each, 128 of them the policy's): the same number of policies per SIMD, but two instruction
// streams -- a wave alone on its SIMD runs at the latency of its dependent chains (environment step, activations) and
// leaves the pipe half idle.  WREG = 6: the same with one column group (up to 4 rollouts), WREG = 7: the same with a 16-column
// tile (up to 16 rollouts per task).  The 32-wide networks (16 fragments, 32 registers per policy) run WREG = 2 and 3
// with two waves per SIMD: their LDS images of 10 KB each kept only two or three waves of four policies on a CU.
constexpr int NSET_OF(int wreg) { return wreg >= 2 && wreg <= 4 ? 2 : 1; }
template <int KIND, int HP, bool FINE, bool WRAP, bool POP, bool TANH, int WREG = 0, bool FIXT = false, bool SGN = false>
__global__ __launch_bounds__(WREG ? (HP == 64 && WREG < 5 ? 256 : 512) : ((WRAP || POP) ? 512 : 768)) void k_rollout_fnn(const RolloutArgs a) {
    using E = Env<KIND>;
    static_assert(!WREG || (POP && TANH && FINE && HP >= 32), "WREG is the population kernel of the 32- and 64-wide tanh networks");
    // SGN: the tanh table in LDS is the odd one, tanh(j/256) for j = -5120..5120 (twice the fine table's LDS): the fast
    // activations then need neither |y| nor the sign transfer (one vector instruction less each)
    static_assert(!SGN || (!POP && TANH && FINE), "SGN is a variant of the single-policy all-tanh kernels");
    constexpr int LV1 = SGN ? 3 : 1, LV2 = SGN ? 4 : 2;   // nl_apply_all levels of the fast activations
    // Layer 0: whole groups of 4 inputs go through the MFMA (K0 k-steps); the REM inputs left over are added on the
    // vector ALU (16 FMAs each per tile) instead of through a k-step that is half zeros (QQube's 6 inputs: 4 + 2)
    constexpr int HT = HP / 16, KS = HP / 4, OP = (E::O + 3) / 4 * 4, K0 = E::O / 4, REM = E::O % 4;
    // Wave priorities by phase of the step.  The MFMA blocks of the hidden layers are long runs of independent
    // instructions, the environment step and the activations are dependent chains with table and LDS latencies in
    // them: when the chains of one wave go first and the MFMAs of another fill what they leave open, the pipe idles
    // less than under the default oldest-first issue (1.2-1.6 % of the launch; any other assignment of the three
    // levels gains less, a fixed priority per wave nothing).
    constexpr int PRIO_MFMA = 0, PRIO_ACT = 2, PRIO_ENV = 3;
    // TAIL: the single-policy all-tanh kernels run a tile that holds at most 8 environments on 4x4x4 MFMAs (run_tail)
    constexpr bool TAIL = !POP && TANH && FINE && WREG == 0;
    // NARROW: the register-weight kernels whose policies run in column groups of 4 rollouts (run_narrow)
    constexpr bool NARROW = WREG >= 3 && WREG != 7;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = lane >> 4, c = lane & 15;
    // POP (population mode): every wave has weight images of its own in LDS, those of the policies whose rollouts its
    // groups of tiles run (streamed from global memory per MFMA fragment the weights cost several microseconds of
    // latency per layer: 100 us per step against 4)
    const int n_waves = (int)(blockDim.x >> 6);
    const int pop_shift = a.pop_tiles >> 1;     // log2 of 1, 2, 4
    const int imgs = WREG ? NSET_OF(WREG) : (POP ? a.pop_groups : 1);  // images per wave
    const int w_doubles = POP ? n_waves * imgs * a.lds_doubles : a.lds_doubles;
    double* const w_lds = lds + (POP ? wave * imgs * a.lds_doubles : 0);
    // (a wave's staging area: one row of 64 per observation entry; the rows that pad the observation to a multiple of 4 are never read)
    double* const stage = lds + w_doubles + wave * (E::O * 64);
    double* const tanh_lds = lds + w_doubles + n_waves * (E::O * 64);
    double* const task_lds = tanh_lds + (SGN ? TANH_ODD_PAD : (FINE ? TANH8_PAD : TANH_TAB_PAD));
    const double* const tanh_tab = tanh_lds + (SGN ? PSH_TANH8_N - 1 : 0);   // node 0
    const double* const task_base = opaque_lds_base(task_lds);
    const psh_task& task = *reinterpret_cast<const psh_task*>(task_base);
    const psh_wrappers& wrap = *reinterpret_cast<const psh_wrappers*>(task_base + sizeof(psh_task) / 8);

    if constexpr (!POP) {
        for (int j = threadIdx.x; j < w_doubles; j += blockDim.x) w_lds[j] = a.wfrag[j];
    }
    for (int j = threadIdx.x; j < a.tanh_n; j += blockDim.x) tanh_lds[j] = a.tanh_tab[j];
    stage_task(a, task_lds);
    __syncthreads();

    Lane<KIND, WRAP, FIXT, !WRAP && !POP> L;   // FIXT: the simulation's default task shape compiled in (see k_rollout_linear)
    L.init();
    if (!POP && lane >= a.wave_lanes) L.exhausted = 1;  // a small launch is spread over more waves, fewer tiles each

    const int n_hidden = a.pol.n_hidden;
    int pol = 0;  // POP: policy of this lane's group of tiles
    int fault = 0;  // POP: PSH_ST_INTERNAL once the lanes of this lane's group were seen to disagree (see turnover_group)
    // WREG: fragment (t, s) of the hidden-to-hidden weights of this wave's policy / of the policies of its tiles 0 and 1
    constexpr int NSET = NSET_OF(WREG);
    double wreg[NSET][WREG ? HT * KS : 1];
    if constexpr (WREG != 0) {
#pragma unroll
        for (int g = 0; g < NSET; ++g)
#pragma unroll
            for (int f = 0; f < HT * KS; ++f) wreg[g][f] = 0.;
    }
    // WREG = 3: the 4-lane block whose value a rotation by 4 rot lanes within the row of 16 delivers to this lane
    int b_from[4] = {0, 0, 0, 0};
    if constexpr (NARROW) {
        b_from[0] = (lane >> 2) & 3;
        b_from[1] = (__builtin_amdgcn_update_dpp(0, lane, 0x120 + 4, 0xf, 0xf, false) >> 2) & 3;
        b_from[2] = (__builtin_amdgcn_update_dpp(0, lane, 0x120 + 8, 0xf, 0xf, false) >> 2) & 3;
        b_from[3] = (__builtin_amdgcn_update_dpp(0, lane, 0x120 + 12, 0xf, 0xf, false) >> 2) & 3;
    }
    // one policy: which hidden layers may run the activation without clamp (read once; per tile it would be an LDS round
    // trip in front of every layer's activations)
    int bounded_layers = 0;
    if constexpr (!POP) {
        for (int l = 1; l < n_hidden; ++l) bounded_layers |= (w_lds[a.lds_doubles - FNN_TAIL + l] <= FNN_BOUND_LIMIT ? 1 : 0) << l;
        // bit 0: every weight of the hidden layers is finite and the first layer's gain is moderate (k_fnn_bounds)
        bounded_layers |= w_lds[a.lds_doubles - FNN_TAIL] <= FNN_GAIN_LIMIT ? 1 : 0;
        bounded_layers = __builtin_amdgcn_readfirstlane(bounded_layers);
    }
    for (;;) {
        if constexpr (POP) {
            unsigned long long fresh;
            pol = L.turnover_group(a, task, wrap, pol, &fresh, fault);
            if (fresh) {  // only this wave reads or writes its images, in program order
                while (fresh) {
                    const int first = __ffsll((long long)fresh) - 1;
                    fresh &= fresh - 1;
                    const double* src = a.wfrag + (long)__shfl(pol, first) * a.img_doubles;
                    double* dst = w_lds + (first >> (4 + pop_shift)) * a.lds_doubles;
                    if constexpr (WREG != 0) {
                        // the block of hidden-layer fragments goes to registers, what lies before and behind it to LDS
                        constexpr int OFF1 = HT * K0 * 64 + REM * HP + HP, BLOCK = HT * KS * 64;
                        for (int j = lane; j < OFF1; j += 64) dst[j] = src[j];
                        for (int j = OFF1 + lane; j < a.lds_doubles; j += 64) dst[j] = src[j + BLOCK];
                        // WREG = 3: register (rs, rp, rot) holds the lane's value of fragment (rs, 4 rp + block the rotation
                        // `rot` of the activations reads from), see run_narrow.  One base per rotation, constant offsets behind it.
                        if constexpr (NARROW) {
                            const double* base[4];
#pragma unroll
                            for (int rot = 0; rot < 4; ++rot) base[rot] = src + OFF1 + b_from[rot] * 64 + lane;
                            if (first < 16) {
#pragma unroll
                                for (int f = 0; f < HT * KS; ++f) wreg[0][f] = base[f & 3][((f / (4 * HT)) * KS + 4 * ((f >> 2) % HT)) * 64];
                            } else {
#pragma unroll
                                for (int f = 0; f < HT * KS; ++f) wreg[NSET - 1][f] = base[f & 3][((f / (4 * HT)) * KS + 4 * ((f >> 2) % HT)) * 64];
                            }
                        } else if (NSET == 1 || first < 16) {
#pragma unroll
                            for (int f = 0; f < HT * KS; ++f) wreg[0][f] = src[OFF1 + f * 64 + lane];
                        } else {
#pragma unroll
                            for (int f = 0; f < HT * KS; ++f) wreg[NSET - 1][f] = src[OFF1 + f * 64 + lane];
                        }
                    } else {
                        for (int j = lane; j < a.lds_doubles; j += 64) dst[j] = src[j];
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
        } else {
            L.turnover(a, task, wrap);
        }
        if (!__any(L.i >= 0)) break;  // every lane of the wave is out of work
        if constexpr (!POP) {
            // The drain: once the queue is empty, finished lanes leave holes, and a tile costs its MFMAs as long as one
            // of its 16 lanes is busy.  Whenever the busy lanes would fit into fewer tiles than they occupy, the
            // environments of the wave are moved together (at most three times per wave).
            // (not with an ActDelayWrapper: its ring buffer belongs to the lane, not to the environment)
            if ((!WRAP || a.delay_q == nullptr) && __any(L.exhausted)) {
                const unsigned long long busy = __ballot(L.i >= 0);
                const int n_busy = __popcll(busy);
                int tiles = 0;
#pragma unroll
                for (int t = 0; t < 4; ++t) tiles += ((busy >> (16 * t)) & 0xFFFFull) != 0;
                // (TAIL, below: a last tile with at most 8 environments runs them in column groups of 4, which have to be its
                // first lanes -- the wave is then kept packed whenever a lane has left)
                const int rest = n_busy & 15;
                const bool pack = TAIL && rest >= 1 && rest <= 8 && busy != (n_busy == 64 ? ~0ull : (1ull << n_busy) - 1ull);
                if ((n_busy + 15) / 16 < tiles || pack) {
                    const unsigned long long idle = ~busy;
                    const int rank_busy = __builtin_amdgcn_mbcnt_hi((unsigned)(busy >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)busy, 0u));
                    const int rank_idle = __builtin_amdgcn_mbcnt_hi((unsigned)(idle >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)idle, 0u));
                    move_lane_state(L, L.i >= 0 ? rank_busy : n_busy + rank_idle);
                }
            }
        }
        double obs[OP], act[E::A];
#pragma unroll
        for (int r = 0; r < OP; ++r) obs[r] = 0.;
#pragma unroll
        for (int r = 0; r < E::A; ++r) act[r] = 0.;  // lanes of skipped tiles get no action
        if (L.live) L.pre_policy(a, wrap, obs);
        if (!__any(L.live)) continue;
        // Fast activations: a NaN or an infinity can enter a pre-activation only through an observation or a weight.  While
        // every observation of the wave is finite and moderate (lanes without an environment hold zeros) and the policy's
        // weights are (checked once, when they were packed), the activations run without clamp and NaN bookkeeping; any
        // other step takes the careful forms.
        // One policy: with the largest |weight| per input column of the first layer (k_fnn_bounds) the wave bounds this
        // step's first-layer pre-activations from the observations it holds.  Inside the tanh table's range they need no
        // limit on the table index either (the common case: 16 -> 15 instructions per activation of the first layer), and
        // the test covers the one below (a NaN or an infinity anywhere fails it).
        bool obs_ok = true, tight0 = false;
        if constexpr (TANH && FINE && !POP) {
            const double* const tail = w_lds + (a.lds_doubles - FNN_TAIL);
            double b0 = tail[12];
#pragma unroll
            for (int r = 0; r < E::O; ++r) b0 = fma(fabs(obs[r]), tail[4 + r], b0);
            tight0 = (bounded_layers & 1) != 0 && !__any(!(b0 <= FNN_BOUND_LIMIT));
        }
        if constexpr (TANH && FINE) {
            if (!tight0) {
                bool big = false;
#pragma unroll
                for (int r = 0; r < E::O; ++r) big = big || !(fabs(obs[r]) < FNN_OBS_LIMIT);
                obs_ok = !__any(big);
            }
        }
        // stage the observations: row r of this wave's staging area holds obs[r] of its 64 envs
#pragma unroll
        for (int r = 0; r < E::O; ++r) stage[r * 64 + lane] = obs[r];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        const unsigned long long live_mask = __ballot(L.live);
        __builtin_amdgcn_s_setprio(PRIO_ACT);
        // one MFMA tile: the policy forward of 16 environments; `wr` holds the register-resident fragments of the tile's policy
        auto run_tile = [&](const int tile, const double* wr) {
            // nobody in this tile needs an action: skip it (the drain at the end of a launch, when few long episodes
            // are left, then costs a quarter per step; population tasks with fewer than 64 rollouts)
            if (((live_mask >> (16 * tile)) & 0xFFFFull) == 0) return;
            const double* const wimg = POP ? w_lds + (tile >> pop_shift) * a.lds_doubles : w_lds;  // of this tile's policy
            const bool fast = TANH && FINE && obs_ok && (POP ? wimg[a.lds_doubles - FNN_TAIL] <= FNN_GAIN_LIMIT : (bounded_layers & 1) != 0);
            double y[KS];
            // layer 0
            {
                double xin[K0 > 0 ? K0 : 1], xr[REM > 0 ? REM : 1];
#pragma unroll
                for (int s = 0; s < K0; ++s) xin[s] = stage[(4 * s + q) * 64 + 16 * tile + c];
#pragma unroll
                for (int j = 0; j < REM; ++j) xr[j] = stage[(4 * K0 + j) * 64 + 16 * tile + c];  // of this lane's env
                const double* wl = wimg;
                const double* wr = wimg + HT * K0 * 64;  // [REM][HP]: weights of the left-over inputs
                const double* bl = wr + REM * HP;
#pragma unroll
                for (int t = 0; t < HT; ++t) {
                    v4d acc;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = 16 * t + 4 * r + q;
                        acc[r] = bl[row];
#pragma unroll
                        for (int j = 0; j < REM; ++j) acc[r] = fma(wr[j * HP + row], xr[j], acc[r]);
                    }
#pragma unroll
                    for (int s = 0; s < K0; ++s)
                        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(wl[(t * K0 + s) * 64 + lane], xin[s], acc, 0, 0, 0);
#pragma unroll
                    for (int r = 0; r < 4; ++r) y[4 * t + r] = acc[r];
                }
                if (!POP && tight0)
                    nl_apply_all<KS, FINE, LV2>((int)PSH_NL_TANH, y, tanh_tab);
                else if (fast)
                    nl_apply_all<KS, FINE, LV1>((int)PSH_NL_TANH, y, tanh_tab);
                else
                    nl_apply_all<KS, FINE>(TANH ? (int)PSH_NL_TANH : a.pol.hidden_nonlin[0], y, tanh_tab);
            }
            // further hidden layers
            int off = HT * K0 * 64 + REM * HP + HP;
            if constexpr (WREG) {
                // the second hidden layer from registers; the four output tiles advance together, so that consecutive
                // MFMAs do not depend on each other (this wave has its SIMD to itself)
                double x[KS];
#pragma unroll
                for (int m = 0; m < KS; ++m) x[m] = y[m];
                const double* bl = wimg + off;
                v4d acc[HT];
#pragma unroll
                for (int t = 0; t < HT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[t][r] = bl[16 * t + 4 * r + q];
#pragma unroll
                for (int s = 0; s < KS; ++s)
#pragma unroll
                    for (int t = 0; t < HT; ++t)
                        acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(wr[t * KS + s], x[s], acc[t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < HT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) y[4 * t + r] = acc[t][r];
                if (wimg[a.lds_doubles - FNN_TAIL + 1] <= FNN_BOUND_LIMIT) {
                    if (fast) nl_apply_all<KS, FINE, LV2>((int)PSH_NL_TANH, y, tanh_tab);
                    else nl_apply_all<KS, FINE, 1>((int)PSH_NL_TANH, y, tanh_tab);
                } else {
                    nl_apply_all<KS, FINE>((int)PSH_NL_TANH, y, tanh_tab);
                }
                off += HP;
            }
            for (int l = 1; !WREG && l < n_hidden; ++l) {
                double x[KS];
#pragma unroll
                for (int m = 0; m < KS; ++m) x[m] = y[m];
                const double* wl = wimg + off;
                const double* bl = wl + HT * KS * 64;
                __builtin_amdgcn_s_setprio(PRIO_MFMA);
#pragma unroll
                for (int t = 0; t < HT; ++t) {
                    v4d acc;
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[r] = bl[16 * t + 4 * r + q];
#pragma unroll
                    for (int s = 0; s < KS; ++s)
                        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(wl[(t * KS + s) * 64 + lane], x[s], acc, 0, 0, 0);
                    // the fragment reads run one pair of MFMAs ahead (by itself the scheduler issues each read right before
                    // its MFMAs and waits for it: no loss with three waves per SIMD, a bubble per pair for a wave alone)
                    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
                    for (int s = 0; s < KS; s += 2) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) y[4 * t + r] = acc[r];
                }
                __builtin_amdgcn_s_setprio(PRIO_ACT);
                // behind a tanh layer the inputs lie in [-1, 1]: if the weights of this layer keep every pre-activation
                // below the table's end (row sums of |W| plus |b|, computed by k_fnn_bounds), the clamp and the NaN
                // bookkeeping of the activation are not needed
                if (TANH && FINE && (POP ? wimg[a.lds_doubles - FNN_TAIL + l] <= FNN_BOUND_LIMIT : ((bounded_layers >> l) & 1) != 0)) {
                    if (fast) nl_apply_all<KS, FINE, LV2>((int)PSH_NL_TANH, y, tanh_tab);
                    else nl_apply_all<KS, FINE, 1>((int)PSH_NL_TANH, y, tanh_tab);
                } else {
                    nl_apply_all<KS, FINE>(TANH ? (int)PSH_NL_TANH : a.pol.hidden_nonlin[l], y, tanh_tab);
                }
                off += HT * KS * 64 + HP;
            }
            // output layer on the vector ALU: each lane sums its 16 hidden rows, then the 4 lane groups are added.  The
            // weights are all requested before the first product (left alone, the compiler reuses one register pair and
            // waits for every LDS read in turn), and the two exchange steps run on the vector ALU (row swaps) instead of
            // going through LDS: both are latency a wave alone on its SIMD cannot hide.
            const double* wo = wimg + off;
#pragma unroll
            for (int r = 0; r < E::A; ++r) {
                double wv[KS];
#pragma unroll
                for (int m = 0; m < KS; ++m) wv[m] = wo[r * HP + (m >> 2) * 16 + (m & 3) * 4 + q];
                hold_together<KS>(wv);
                double p = 0.;
#pragma unroll
                for (int m = 0; m < KS; ++m) p += wv[m] * y[m];
                // bias + the partial sums of the 4 lane groups: one 4x4x4 MFMA with an all-ones A operand.  Block b of it adds
                // the B values of the lanes 16 k + 4 b + j over k and hands the sum to all four of them (16 cycles, against
                // four row swaps, their register copies and two additions).
                p = __builtin_amdgcn_mfma_f64_4x4x4f64(1.0, p, wo[E::A * HP + r], 0, 0, 0);
                if (q == tile) act[r] = p;
            }
        };
        // The end of a launch: the queue is empty, the wave's environments are packed into its first lanes, and the last
        // tile that holds any has 1..8 of them left.  Such a tile runs on v_mfma_f64_4x4x4_4b_f64 in NG = 1 or 2 column groups of 4
        // environments -- a quarter or half of the MFMA time and of the activations of a 16-column tile -- from the SAME
        // packed image in LDS: lane 16 h + 4 b + l of a 16x16x4 fragment (t, s) holds W[16 t + 4 b + l][4 s + h], which is
        // A_b[i = l][k = h] of block b for the hidden units 16 t + 4 b + (0..3).  All four blocks take the same 4
        // environments 16 p + 4 g + (0..3) as their columns; lane (h, b, l) receives unit 16 t + 4 b + h of environment l.
        // BIT FOR BIT the numbers of run_tile: the two MFMA shapes round alike (tools/ubench_mfma_equal.hip), every block
        // goes through the k-steps in run_tile's order -- as the B operand of k-step ks every block needs the units
        // 4 ks + h, which block ks % 4 holds: a broadcast of that block's lanes through ds_bpermute -- and the output layer
        // adds its 16 products per lane group in run_tile's order, the running sum handed from block to block.  So which
        // path evaluates a step has no influence on an environment's results, and those do not depend on the schedule.
        // The whole launch ends 1-2 % earlier with it (tools/tail_model.py gran=4 on the benchmark's episode lengths).
        auto run_tail = [&](const int p, auto ngc) {
            if constexpr (TAIL) {
                constexpr int NG = decltype(ngc)::value;
                const double* const wimg = w_lds;
                const bool fast = obs_ok && (bounded_layers & 1) != 0;
                const int h = lane >> 4, b = (lane >> 2) & 3, l = lane & 3;
                const int unit = 4 * b + h;          // + 16 rs
                const int env_lane = 16 * p + l;     // + 4 g
                double y[NG][HT];
                {
                    double xin[K0 > 0 ? K0 : 1][NG], xr[REM > 0 ? REM : 1][NG];
#pragma unroll
                    for (int g = 0; g < NG; ++g) {
#pragma unroll
                        for (int s = 0; s < K0; ++s) xin[s][g] = stage[(4 * s + h) * 64 + env_lane + 4 * g];
#pragma unroll
                        for (int j = 0; j < REM; ++j) xr[j][g] = stage[(4 * K0 + j) * 64 + env_lane + 4 * g];
                    }
                    const double* wl = wimg;
                    const double* wrem = wimg + HT * K0 * 64;
                    const double* bl = wrem + REM * HP;
#pragma unroll
                    for (int rs = 0; rs < HT; ++rs) {
                        const double bias = bl[16 * rs + unit];
                        double wj[REM > 0 ? REM : 1], wa[K0 > 0 ? K0 : 1];
#pragma unroll
                        for (int j = 0; j < REM; ++j) wj[j] = wrem[j * HP + 16 * rs + unit];
#pragma unroll
                        for (int s = 0; s < K0; ++s) wa[s] = wl[(rs * K0 + s) * 64 + lane];
#pragma unroll
                        for (int g = 0; g < NG; ++g) {
                            double acc = bias;
#pragma unroll
                            for (int j = 0; j < REM; ++j) acc = fma(wj[j], xr[j][g], acc);
#pragma unroll
                            for (int s = 0; s < K0; ++s) acc = __builtin_amdgcn_mfma_f64_4x4x4f64(wa[s], xin[s][g], acc, 0, 0, 0);
                            y[g][rs] = acc;
                        }
                    }
#pragma unroll
                    for (int g = 0; g < NG; ++g) {
                        if (fast) nl_apply_all<HT, FINE, LV1>((int)PSH_NL_TANH, y[g], tanh_tab);
                        else nl_apply_all<HT, FINE>((int)PSH_NL_TANH, y[g], tanh_tab);
                    }
                }
                int off = HT * K0 * 64 + REM * HP + HP;
                const int row_lane = (lane & 0x33) << 2;   // byte address (ds_bpermute) of lane (h, block 0, l)
                for (int ly = 1; ly < n_hidden; ++ly) {
                    const double* wl = wimg + off;
                    const double* bl = wl + HT * KS * 64;
                    double acc[HT][NG];
#pragma unroll
                    for (int rt = 0; rt < HT; ++rt) {
                        const double bias = bl[16 * rt + unit];
#pragma unroll
                        for (int g = 0; g < NG; ++g) acc[rt][g] = bias;
                    }
#pragma unroll
                    for (int rs = 0; rs < HT; ++rs) {
                        // the units 16 rs + 4 c + h of block c, for every block: the B operands of the k-steps 4 rs + c
                        double xb[4][NG];
#pragma unroll
                        for (int c4 = 0; c4 < 4; ++c4)
#pragma unroll
                            for (int g = 0; g < NG; ++g) {
                                const int lo = __builtin_amdgcn_ds_bpermute(row_lane + 16 * c4, __double2loint(y[g][rs]));
                                const int hi = __builtin_amdgcn_ds_bpermute(row_lane + 16 * c4, __double2hiint(y[g][rs]));
                                xb[c4][g] = __hiloint2double(hi, lo);
                            }
#pragma unroll
                        for (int c4 = 0; c4 < 4; ++c4)
#pragma unroll
                            for (int rt = 0; rt < HT; ++rt) {
                                const double wf = wl[(rt * KS + 4 * rs + c4) * 64 + lane];
#pragma unroll
                                for (int g = 0; g < NG; ++g)
                                    acc[rt][g] = __builtin_amdgcn_mfma_f64_4x4x4f64(wf, xb[c4][g], acc[rt][g], 0, 0, 0);
                            }
                    }
#pragma unroll
                    for (int g = 0; g < NG; ++g)
#pragma unroll
                        for (int rt = 0; rt < HT; ++rt) y[g][rt] = acc[rt][g];
                    const bool bounded = ((bounded_layers >> ly) & 1) != 0;
#pragma unroll
                    for (int g = 0; g < NG; ++g) {
                        if (bounded && fast) nl_apply_all<HT, FINE, LV2>((int)PSH_NL_TANH, y[g], tanh_tab);
                        else if (bounded) nl_apply_all<HT, FINE, 1>((int)PSH_NL_TANH, y[g], tanh_tab);
                        else nl_apply_all<HT, FINE>((int)PSH_NL_TANH, y[g], tanh_tab);
                    }
                    off += HT * KS * 64 + HP;
                }
                // Output layer.  run_tile's lane group q adds the products of its 16 units 16 t + 4 r + q in the order (t, r);
                // here that is group h, and unit (t, r) sits in block r: the running sum goes round the blocks, one rotation
                // of the row per term (a lane that is not the term's block computes something nobody reads), and ends in
                // block 0.  Then the four groups as in run_tile.
                const double* wo = wimg + off;
#pragma unroll
                for (int r = 0; r < E::A; ++r) {
                    double wv[HT];
#pragma unroll
                    for (int rs = 0; rs < HT; ++rs) wv[rs] = wo[r * HP + 16 * rs + unit];
                    const double bo = wo[E::A * HP + r];
                    double pg[NG];
#pragma unroll
                    for (int g = 0; g < NG; ++g) {
                        double sm = 0.;
#pragma unroll
                        for (int rs = 0; rs < HT; ++rs)
#pragma unroll
                            for (int c4 = 0; c4 < 4; ++c4) {
                                sm = fma(wv[rs], y[g][rs], sm);     // valid in block c4
                                sm = rotate_in_row<4>(sm);           // block c4 + 1 takes it over
                            }
                        pg[g] = __builtin_amdgcn_mfma_f64_4x4x4f64(1.0, sm, bo, 0, 0, 0);   // (block 0 holds the sums)
                    }
                    if (NG == 2) pg[NG - 1] = rotate_in_row<4>(pg[NG - 1]);   // to block 1, whose lanes run column group 1
                    if (h == p && b < NG) act[r] = b == 0 ? pg[0] : pg[NG - 1];
                }
            }
        };
        // WREG = 3: the policy forward of the up to 8 environments in the first 8 lanes of tile p, on 4x4x4 MFMAs.
        // Lane (h, b, l) = 16 h + 4 b + l supplies A_b[i = l][k = h] and B_b[k = h][j = l] of block b and receives
        // D_b[i = h][j = l].  Block b takes the hidden units 16 rs + 4 b + (0..3) of row step rs, all blocks the same 4
        // environments 16 p + 4 g + (0..3) of column group g: the lane holds unit 16 rs + 4 b + h of environment 4 g + l,
        // y[g][rs].  As the B operand of the next layer block b needs the units 4 ks + h of k-step ks: its own for
        // ks = 4 rp + b, those of the other blocks by rotating the row of 16 lanes by 4, 8 and 12 -- every block goes
        // through the 16 k-steps in an order of its own, which its weight registers follow (frag_of above).
        auto run_narrow = [&](const int p, const double* wr) {
            if constexpr (NARROW) {
                if (((live_mask >> (16 * p)) & 0xFFFFull) == 0) return;
                constexpr int NG = (WREG == 4 || WREG == 6) ? 1 : 2;   // column groups of 4 rollouts
                const double* const wimg = w_lds + p * a.lds_doubles;
                const bool fast = obs_ok && wimg[a.lds_doubles - FNN_TAIL] <= FNN_GAIN_LIMIT;
                const int h = lane >> 4, b = (lane >> 2) & 3, l = lane & 3;
                const int unit = 4 * b + h;          // + 16 rs
                const int env_lane = 16 * p + l;     // + 4 g
                double y[NG][HT];
                {
                    double xin[K0 > 0 ? K0 : 1][NG], xr[REM > 0 ? REM : 1][NG];
#pragma unroll
                    for (int g = 0; g < NG; ++g) {
#pragma unroll
                        for (int s = 0; s < K0; ++s) xin[s][g] = stage[(4 * s + h) * 64 + env_lane + 4 * g];
#pragma unroll
                        for (int j = 0; j < REM; ++j) xr[j][g] = stage[(4 * K0 + j) * 64 + env_lane + 4 * g];
                    }
                    const double* wl = wimg;
                    const double* wrem = wimg + HT * K0 * 64;
                    const double* bl = wrem + REM * HP;
#pragma unroll
                    for (int rs = 0; rs < HT; ++rs) {
                        const double bias = bl[16 * rs + unit];
                        double wj[REM > 0 ? REM : 1], wa[K0 > 0 ? K0 : 1];
#pragma unroll
                        for (int j = 0; j < REM; ++j) wj[j] = wrem[j * HP + 16 * rs + unit];
#pragma unroll
                        for (int s = 0; s < K0; ++s) wa[s] = wl[(rs * K0 + s) * 64 + lane];
#pragma unroll
                        for (int g = 0; g < NG; ++g) {
                            double acc = bias;
#pragma unroll
                            for (int j = 0; j < REM; ++j) acc = fma(wj[j], xr[j][g], acc);
#pragma unroll
                            for (int s = 0; s < K0; ++s) acc = __builtin_amdgcn_mfma_f64_4x4x4f64(wa[s], xin[s][g], acc, 0, 0, 0);
                            y[g][rs] = acc;
                        }
                    }
                    if (fast) {
#pragma unroll
                        for (int g = 0; g < NG; ++g) nl_apply_all<HT, FINE, 1>((int)PSH_NL_TANH, y[g], tanh_tab);
                    } else {
#pragma unroll
                        for (int g = 0; g < NG; ++g) nl_apply_all<HT, FINE>((int)PSH_NL_TANH, y[g], tanh_tab);
                    }
                }
                int off = HT * K0 * 64 + REM * HP + HP;
                {
                    const double* bl = wimg + off;
                    double acc[HT][NG];
#pragma unroll
                    for (int rs = 0; rs < HT; ++rs) {
                        const double bias = bl[16 * rs + unit];
#pragma unroll
                        for (int g = 0; g < NG; ++g) acc[rs][g] = bias;
                    }
#pragma unroll
                    for (int rp = 0; rp < HT; ++rp) {
#pragma unroll
                        for (int rot = 0; rot < 4; ++rot) {
                            double xb[NG];
#pragma unroll
                            for (int g = 0; g < NG; ++g) {
                                if (rot == 0) xb[g] = y[g][rp];
                                if (rot == 1) xb[g] = rotate_in_row<4>(y[g][rp]);
                                if (rot == 2) xb[g] = rotate_in_row<8>(y[g][rp]);
                                if (rot == 3) xb[g] = rotate_in_row<12>(y[g][rp]);
                            }
#pragma unroll
                            for (int rs = 0; rs < HT; ++rs)
#pragma unroll
                                for (int g = 0; g < NG; ++g)
                                    acc[rs][g] = __builtin_amdgcn_mfma_f64_4x4x4f64(wr[(rs * HT + rp) * 4 + rot], xb[g], acc[rs][g], 0, 0, 0);
                        }
                    }
#pragma unroll
                    for (int g = 0; g < NG; ++g)
#pragma unroll
                        for (int rs = 0; rs < HT; ++rs) y[g][rs] = acc[rs][g];
                    if (wimg[a.lds_doubles - FNN_TAIL + 1] <= FNN_BOUND_LIMIT) {
                        if (fast) {
#pragma unroll
                            for (int g = 0; g < NG; ++g) nl_apply_all<HT, FINE, 2>((int)PSH_NL_TANH, y[g], tanh_tab);
                        } else {
#pragma unroll
                            for (int g = 0; g < NG; ++g) nl_apply_all<HT, FINE, 1>((int)PSH_NL_TANH, y[g], tanh_tab);
                        }
                    } else {
#pragma unroll
                        for (int g = 0; g < NG; ++g) nl_apply_all<HT, FINE>((int)PSH_NL_TANH, y[g], tanh_tab);
                    }
                    off += HP;
                }
                // output layer: the lane's 4 units per environment, then the sum over the 16 lanes (b, h) with the same l
                const double* wo = wimg + off;
#pragma unroll
                for (int r = 0; r < E::A; ++r) {
                    double wv[HT];
#pragma unroll
                    for (int rs = 0; rs < HT; ++rs) wv[rs] = wo[r * HP + 16 * rs + unit];
                    hold_together<HT>(wv);
                    const double bo = wo[E::A * HP + r];
                    double pg[NG];
#pragma unroll
                    for (int g = 0; g < NG; ++g) {
                        double s = 0.;
#pragma unroll
                        for (int rs = 0; rs < HT; ++rs) s += wv[rs] * y[g][rs];
                        s += rotate_in_row<4>(s);
                        s += rotate_in_row<8>(s);
                        pg[g] = __builtin_amdgcn_mfma_f64_4x4x4f64(1.0, s, bo, 0, 0, 0);   // + the other three rows of 16 lanes
                    }
                    if (h == p && b < NG) act[r] = b == 0 ? pg[0] : pg[NG - 1];
                }
            }
        };
        if constexpr (NARROW) {   // policies of up to 8 (WREG = 4, 6: up to 4) rollouts, each with its own register set
            run_narrow(0, wreg[0]);
            if constexpr (NSET == 2) run_narrow(1, wreg[NSET - 1]);
        } else if constexpr (WREG == 2 || WREG == 7) {   // one or two tiles, each with the register set of its own policy
            run_tile(0, wreg[0]);
            if constexpr (NSET == 2) run_tile(1, wreg[NSET - 1]);
        } else {
            for (int tile = 0; tile < 4; ++tile) {
                if constexpr (TAIL) {
                    // (only a draining wave has its environments packed, so only then can a tile hold few of them in its first lanes)
                    const unsigned in_tile = (unsigned)(live_mask >> (16 * tile)) & 0xFFFFu;
                    if (in_tile != 0 && (in_tile & 0xFFF0u) == 0) { run_tail(tile, std::integral_constant<int, 1>{}); continue; }
                    if (in_tile != 0 && (in_tile & 0xFF00u) == 0) { run_tail(tile, std::integral_constant<int, 2>{}); continue; }
                }
                run_tile(tile, wreg[0]);
            }
        }
        __builtin_amdgcn_s_setprio(PRIO_ENV);
        // output nonlinearity, once per lane on its own action.  A branch the compiler must keep: without side
        // effects in it, all nonlinearities were evaluated speculatively and selected from (as they were per tile)
        if (a.pol.out_nonlin != PSH_NL_NONE) {
#pragma unroll
            for (int r = 0; r < E::A; ++r) {
                asm volatile("" : "+v"(act[r]));
                nl_apply_all<1, FINE>(a.pol.out_nonlin, &act[r], tanh_tab);
            }
        }
        if (L.live) L.post_policy(a, task, wrap, act);
    }
}

// Largest pre-activation a hidden layer behind a bounded layer can produce: max over its units of sum_i |W[u][i]| + |b[u]|
// (inputs in [-1, 1]).  Written behind the packed image, one double per layer.  A NaN or infinite weight yields an
// infinite bound, which fails the kernel's comparison and keeps the clamping activation.  Entry 0 is the same sum for the
// input layer (its bound per unit of the largest observation), made infinite when any hidden layer holds a weight that is
// not finite: the kernels take it as "a NaN or infinity can only come in through an observation".  One wave per policy.
static __global__ __launch_bounds__(64) void k_fnn_bounds(const double* __restrict__ params, double* __restrict__ out, int total, int O,
                                                   int A, int n_hidden, int h0, int h1, int h2, int h3, int num_param) {
    params += (long)blockIdx.x * num_param;
    out += (long)blockIdx.x * total + (total - FNN_TAIL);
    const int hs[4] = {h0, h1, h2, h3};
    const int u = threadIdx.x;
    int poff = 0, last = O;
    double first = 0.;
    bool all_finite = true;
    for (int l = 0; l < 4; ++l) {
        double bound = 0.;
        if (l < n_hidden) {
            const int h = hs[l];
            if (u < h) {
                for (int i = 0; i < last; ++i) bound += fabs(params[poff + u * last + i]);
                bound += fabs(params[poff + h * last + u]);
            }
            bool bad = !(bound < __builtin_inf());
            for (int o = 32; o > 0; o >>= 1) {
                const double other = __shfl_xor(bound, o);
                bound = other > bound ? other : bound;
            }
            bad = __any(bad);
            if (bad) bound = __builtin_inf();
            all_finite = all_finite && !bad;
            if (l == 0) first = bound;
            poff += h * last + h;
            last = h;
        } else {
            bound = __builtin_inf();
        }
        if (u == 0 && l > 0) out[l] = bound;
    }
    // entry 0: the first layer's bound per unit of observation, infinite unless EVERY hidden layer's weights are finite
    if (u == 0) out[0] = all_finite ? first : __builtin_inf();
    // entries 4.., 12: per input column the largest |weight| of the first layer, and its largest |bias|: with them a wave
    // bounds its first-layer pre-activations from the observations it holds, |y| <= sum_r colmax[r] |obs[r]| + bmax
    // (a NaN or infinite weight gives a NaN or infinite entry, which fails the kernel's comparison)
    const int h = hs[0];
    for (int r = 0; r <= O && r <= 8; ++r) {
        double m = 0.;
        if (u < h) m = fabs(r < O ? params[u * O + r] : params[h * O + u]);
        bool bad = m != m;
        for (int o = 32; o > 0; o >>= 1) {
            const double other = __shfl_xor(m, o);
            m = other > m ? other : m;
        }
        if (__any(bad)) m = __builtin_nan("");
        if (u == 0) out[r < O ? 4 + r : 12] = m;
    }
}

// Builds the packed weight image of k_rollout_fnn from the torch-ordered flat parameter vector.
// The rows that pad a hidden layer to the MFMA tile (units h .. HP - 1) are copies of the layer's unit 0, not zeros: the
// next layer's weights for them are zero either way, so a padding unit contributes 0 x (a value unit 0 also takes) --
// whereas a zero ROW would turn an infinite input into 0 x inf = NaN, which the zero weight behind it keeps a NaN, in a
// network whose real units saturate at +-1 as they do in the reference.
template <int KIND>
__global__ void k_pack_fnn(const double* __restrict__ params, double* __restrict__ out, int total, int O, int A, int HP,
                           int K0, int REM, int n_hidden, int h0, int h1, int h2, int h3, int num_param) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    params += (long)blockIdx.y * num_param;  // one image per policy (blockIdx.y)
    out += (long)blockIdx.y * total;
    const int hs[4] = {h0, h1, h2, h3};
    const int HT = HP / 16, KS = HP / 4;
    int off = 0, poff = 0, last = O;
    double v = 0.;
    bool found = false;
    for (int l = 0; l < n_hidden && !found; ++l) {
        const int ks = (l == 0) ? K0 : KS, h = hs[l];
        const int nfrag = HT * ks * 64;
        const int nrem = (l == 0) ? REM * HP : 0;  // layer 0: weights of the inputs beyond the last whole group of 4
        if (idx < off + nfrag) {
            const int f = (idx - off) >> 6, lane = (idx - off) & 63;
            const int t = f / ks, s = f % ks;
            const int row = 16 * t + (lane & 15), col = 4 * s + (lane >> 4);
            v = (col < last) ? params[poff + (row < h ? row : 0) * last + col] : 0.;
            found = true;
        } else if (idx < off + nfrag + nrem) {
            const int j = idx - off - nfrag;
            const int row = j % HP, col = 4 * K0 + j / HP;
            v = (col < last) ? params[poff + (row < h ? row : 0) * last + col] : 0.;
            found = true;
        } else if (idx < off + nfrag + nrem + HP) {
            const int j = idx - off - nfrag - nrem;
            v = params[poff + h * last + (j < h ? j : 0)];
            found = true;
        }
        off += nfrag + nrem + HP;
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

// The fused wrappers of a rollout: every op known and used once, sizes in range.  Yields the policy's input dimension.
static inline int check_wrap(psh_ctx* c, const psh_wrappers& w, int O, int* obs_dim, bool* has_delay) {
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
static inline int launch_persistent(psh_ctx* c, K kernel, int block, size_t lds_bytes, RolloutArgs& a, int act_dim, bool need_delay_q,
                                    hipStream_t st, long work_lanes = -1, int max_per_cu = 0) {
    if (lds_bytes)
        PSH_HIP(c, hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    const unsigned grid = persistent_grid(c, kernel, block, lds_bytes, work_lanes < 0 ? (long)a.n_envs : work_lanes, max_per_cu);
    a.n_lanes = (long)grid * block;
    if (need_delay_q) {
        const int rc = scratch_doubles(c, PSH_SCR_DELAY, (size_t)PSH_MAX_DELAY * act_dim * a.n_lanes, &a.delay_q);
        if (rc != PSH_OK) return rc;
    }
    if (c->timed) PSH_HIP(c, hipEventRecord(c->ev0, st));
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), lds_bytes, st, a);
    PSH_HIP(c, hipGetLastError());
    if (c->timed) PSH_HIP(c, hipEventRecord(c->ev1, st));
    return PSH_OK;
}

template <int KIND> int launch_rollout(psh_ctx* c, const psh_rollout_desc* d, RolloutArgs& a, hipStream_t st) {
    using E = Env<KIND>;
    const psh_policy& p = d->policy;
    const bool pop = p.policy_group > 0;
    if (pop && (p.policy_index != nullptr || (long)p.n_policies * p.policy_group != (long)d->n_envs)) {
        set_err(c, "population mode needs policy_index == NULL and n_envs == n_policies * policy_group");
        return PSH_ERR_ARG;
    }
    const bool per_env = (p.policy_index != nullptr || pop) && p.n_policies > 1;
    int Oe = E::O;
    bool has_delay = false;
    {
        const int rc = check_wrap(c, d->wrap, E::O, &Oe, &has_delay);
        if (rc != PSH_OK) return rc;
    }
    if (p.expl_kind < PSH_EXPL_NONE || p.expl_kind > PSH_EXPL_UNIFORM) { set_err(c, "unknown exploration kind"); return PSH_ERR_ARG; }
    if (!(d->bernoulli_reset >= 0. && d->bernoulli_reset <= 1.)) { set_err(c, "bernoulli_reset must be within [0, 1]"); return PSH_ERR_ARG; }
    // wrappers, exploration noise and the stochastic reset live in the WRAP kernels
    bool wrapped = d->wrap.n_ops > 0 || p.expl_kind != PSH_EXPL_NONE || d->bernoulli_reset > 0.;
    // (a lone ActNormWrapper runs on the bare kernels)
    // (LinearPolicy kernels and the single-policy FNN kernels; FNN populations keep the wrapped kernels)
    if (d->wrap.n_ops == 1 && d->wrap.ops[0] == PSH_WRAP_ACT_NORM && p.expl_kind == PSH_EXPL_NONE && !(d->bernoulli_reset > 0.) &&
        !(p.kind == PSH_POL_FNN && per_env)) {
        wrapped = false;
        a.flags |= PSH_F_INTERNAL_ACT_NORM;
    }
    a.obs_dim = Oe;
    a.wrap_obs_ops = 0;
    for (int j = 0; j < d->wrap.n_ops; ++j)
        a.wrap_obs_ops |= d->wrap.ops[j] == PSH_WRAP_OBS_NOISE || d->wrap.ops[j] == PSH_WRAP_OBS_NORM || d->wrap.ops[j] == PSH_WRAP_OBS_PARTIAL;
    {
        const unsigned long N = (unsigned long)a.n_envs, T = (unsigned long)a.max_steps, O_ = (unsigned long)Oe, A_ = E::A;
        if ((d->obs || d->act || d->rew) && (O_ * N > 0xffffffffull || A_ * N > 0xffffffffull || (T + 1) * O_ > 0xffffffffull || T * A_ > 0xffffffffull)) {
            set_err(c, "trajectory arrays whose strides exceed 2^32 elements are not supported");
            return PSH_ERR_UNSUPPORTED;
        }
        if (a.flags & PSH_F_TRAJ_ENV_MAJOR) {
            a.obs_sk = (unsigned)O_; a.obs_sr = 1; a.obs_si = (unsigned)((T + 1) * O_);
            a.act_sk = (unsigned)A_; a.act_sr = 1; a.act_si = (unsigned)(T * A_);
            a.rew_sk = 1; a.rew_si = (unsigned)T;
        } else {
            a.obs_sk = (unsigned)(O_ * N); a.obs_sr = (unsigned)N; a.obs_si = 1;
            a.act_sk = (unsigned)(A_ * N); a.act_sr = (unsigned)N; a.act_si = 1;
            a.rew_sk = (unsigned)N; a.rew_si = 1;
        }
    }
    if (p.kind == PSH_POL_LINEAR) {
        if (p.n_feats < 1 || p.n_feats > PSH_MAX_FEATS) { set_err(c, "LinearPolicy needs 1..8 feature functions"); return PSH_ERR_ARG; }
        int F = 0;
        for (int f = 0; f < p.n_feats; ++f) {
            const int id = p.feats[f], sz = p.feat_size[f];
            if (id < 0 || id > PSH_FEAT_RBF) { set_err(c, "unknown feature id"); return PSH_ERR_ARG; }
            long data = 0;  // doubles the feature reads from feat_data
            if (id == PSH_FEAT_MULT) {
                if (sz < 2 || sz > 8) { set_err(c, "MultFeat takes 2..8 indices"); return PSH_ERR_ARG; }
                for (int j = 0; j < sz; ++j)
                    if (((p.feat_arg[f] >> (4 * j)) & 15) >= Oe) { set_err(c, "MultFeat index outside the observation"); return PSH_ERR_ARG; }
                F += 1;
            } else if (id == PSH_FEAT_RFF) {
                if (sz < 1 || sz > PSH_MAX_LIN_FEAT) { set_err(c, "RandFourierFeat needs 1..512 features"); return PSH_ERR_ARG; }
                data = 1 + (long)sz + (long)sz * Oe;
                F += sz;
            } else if (id == PSH_FEAT_RBF) {
                if (sz < 2 || sz > PSH_MAX_LIN_FEAT) { set_err(c, "RBFFeat needs 2..512 centers per dimension"); return PSH_ERR_ARG; }
                data = (long)Oe + (long)sz * Oe;
                F += sz * Oe;
            } else {
                F += (id == PSH_FEAT_CONST) ? 1 : Oe;
            }
            if (data > 0 && (!p.feat_data || p.feat_off[f] < 0 || (long)p.feat_off[f] + data > (long)p.feat_data_len)) {
                set_err(c, "feat_data does not hold the data of a RandFourierFeat / RBFFeat");
                return PSH_ERR_ARG;
            }
            if (F > PSH_MAX_LIN_FEAT) { set_err(c, "LinearPolicy with more than 512 feature values"); return PSH_ERR_ARG; }
        }
        if (p.num_param != E::A * F) { set_err(c, "num_param does not match the LinearPolicy layout"); return PSH_ERR_ARG; }
        a.num_feat = F;
        // time-major trajectories: octets of lanes in lock step, whole 64-byte pieces per store (Lane::turnover)
        // (16 lanes = one 128-byte line per row: BallOnBeam, 2^21 rollouts of 69 of 500 steps on average, records in 7.6 ms
        // with lanes refilled one by one, 6.1 / 4.2 / 3.8 / 3.6 ms in cohorts of 4 / 8 / 16 / 32)
        a.cohort = (d->obs || d->act || d->rew) && !(a.flags & PSH_F_TRAJ_ENV_MAJOR) ? 16 : 0;
        const bool simple = (p.n_feats == 1 && p.feats[0] == PSH_FEAT_IDENTITY) ||
                            (p.n_feats == 2 && p.feats[0] == PSH_FEAT_CONST && p.feats[1] == PSH_FEAT_IDENTITY);
        // the usual case gets a kernel of its own: bare simulation, const / identity features, the simulation's default task
        const bool fixt = simple && task_has_default_shape<KIND>(d->task);
        // At most four waves per SIMD for the simulations with a step of a few hundred instructions: they are bound by vector
        // instruction issue from four waves on, and a fifth or sixth (the bare kernels of BallOnBeam and QQubeStab have the
        // registers for them) only slows every wave's step down -- which is what a launch of short episodes waits for at its
        // end, when the few environments that run to the horizon are left (2^20 rollouts, BallOnBeam 75 -> 80 G, QQubeStab 47.5
        // -> 50.7 G env-steps/s).  The two-state simulations have steps short enough to be latency-bound and gain from every
        // wave (OneMassOscillator 663 -> 727 G, Pendulum 216 -> 243 G with six instead of four).
        const int lin_per_cu = E::S <= 2 ? 0 : 4;
#define PSH_LIN_KERNEL(PER_ENV_, WRAP_, SIMPLE_, FIXT_) \
    (a.cohort ? launch_persistent(c, k_rollout_linear<KIND, PER_ENV_, WRAP_, SIMPLE_, FIXT_, true>, 256, 0, a, E::A, WRAP_ && has_delay, st, -1, lin_per_cu) \
              : launch_persistent(c, k_rollout_linear<KIND, PER_ENV_, WRAP_, SIMPLE_, FIXT_, false>, 256, 0, a, E::A, WRAP_ && has_delay, st, -1, lin_per_cu))
#define PSH_LAUNCH_LIN(PER_ENV_, WRAP_) (simple ? PSH_LIN_KERNEL(PER_ENV_, WRAP_, true, false) : PSH_LIN_KERNEL(PER_ENV_, WRAP_, false, false))
        if (fixt && wrapped) return per_env ? PSH_LIN_KERNEL(true, true, true, true) : PSH_LIN_KERNEL(false, true, true, true);
        if (fixt) return per_env ? PSH_LIN_KERNEL(true, false, true, true) : PSH_LIN_KERNEL(false, false, true, true);
        if (per_env) return wrapped ? PSH_LAUNCH_LIN(true, true) : PSH_LAUNCH_LIN(true, false);
        return wrapped ? PSH_LAUNCH_LIN(false, true) : PSH_LAUNCH_LIN(false, false);
#undef PSH_LIN_KERNEL
#undef PSH_LAUNCH_LIN
    }
    if (p.kind != PSH_POL_FNN) { set_err(c, "unknown policy kind"); return PSH_ERR_ARG; }
    if (p.n_hidden < 1 || p.n_hidden > PSH_MAX_HIDDEN) { set_err(c, "FNNPolicy needs 1..4 hidden layers"); return PSH_ERR_ARG; }
    int hmax = 0, np = 0, last = Oe;
    bool all_tanh = true;
    for (int l = 0; l < p.n_hidden; ++l) {
        all_tanh = all_tanh && p.hidden_nonlin[l] == PSH_NL_TANH;
        const int h = p.hidden[l];
        if (h < 1 || h > PSH_MAX_WIDTH) { set_err(c, "hidden sizes must be within 1..64"); return PSH_ERR_ARG; }
        if (p.hidden_nonlin[l] < 0 || p.hidden_nonlin[l] > PSH_NL_SIGMOID) { set_err(c, "unknown nonlinearity"); return PSH_ERR_ARG; }
        hmax = h > hmax ? h : hmax;
        np += h * last + h;
        last = h;
    }
    np += E::A * last + E::A;
    if (p.num_param != np) { set_err(c, "num_param does not match the FNNPolicy layout"); return PSH_ERR_ARG; }
    if (per_env && !pop) {
        set_err(c, "FNN policies per environment need population mode (policy_group): the 16 environments of an MFMA tile share one policy");
        return PSH_ERR_UNSUPPORTED;
    }
    const int HP = hmax <= 16 ? 16 : (hmax <= 32 ? 32 : 64);
    const int OP = (E::O + 3) / 4 * 4, HT = HP / 16, KS = HP / 4, K0 = E::O / 4, REM = E::O % 4;
    // the image: weight fragments, biases, output layer (2 doubles of bias), FNN_TAIL doubles of bounds
    const int total = HT * K0 * 64 + REM * HP + HP + (p.n_hidden - 1) * (HT * KS * 64 + HP) + E::A * HP + 2 + FNN_TAIL;
    double* wfrag = nullptr;
    {
        const int rc = scratch_doubles(c, PSH_SCR_WFRAG, (size_t)total * (per_env ? p.n_policies : 1), &wfrag);
        if (rc != PSH_OK) return rc;
    }
    hipLaunchKernelGGL(k_pack_fnn<KIND>, dim3((total + 255) / 256, per_env ? p.n_policies : 1), dim3(256), 0, st, p.params,
                       wfrag, total, Oe, (int)E::A, HP, K0, REM, p.n_hidden, p.hidden[0], p.hidden[1], p.hidden[2], p.hidden[3],
                       p.num_param);
    PSH_HIP(c, hipGetLastError());
    hipLaunchKernelGGL(k_fnn_bounds, dim3(per_env ? p.n_policies : 1), dim3(64), 0, st, p.params, wfrag, total, Oe, (int)E::A,
                       p.n_hidden, p.hidden[0], p.hidden[1], p.hidden[2], p.hidden[3], p.num_param);
    PSH_HIP(c, hipGetLastError());
    a.wfrag = wfrag;
    a.lds_doubles = total;
    a.img_doubles = total;
    // 8 waves per workgroup (2 per SIMD) or 12 (3 per SIMD, which the 168 registers of the bare kernels allow).  The third
    // wave hides more latency (+2..9 % throughput while the queue lasts), but every wave then steps 1.5 times slower,
    // and so does the drain at the end of a launch: taken when the queue fills the larger grid at least twice
    // (measured at 2 and 5 envs per lane: ahead for 6 of 8 environment kinds, within 4 % for the others).
    const long lanes12 = (long)c->num_cus * 768;
    // (and only if the four extra staging areas do not push the fine tanh table out of LDS)
    const bool fits12 = ((size_t)total + 12 * OP * 64 + TASK_DOUBLES + TANH8_PAD) * sizeof(double) <= c->lds_per_block;
    int BLOCK = (!per_env && !wrapped && fits12 && (long)d->n_envs >= 2 * lanes12) ? 768 : 512;
    // A launch that does not fill two waves per SIMD is spread out: a wave-step costs its per-wave part plus one share
    // per MFMA tile (16 lanes) in use, so 1 000 environments run three to four times faster as 63 waves of one tile on
    // 63 CUs than as 16 full waves on 2.  Tiles per wave = what two waves per SIMD need to hold the launch; one wave per
    // workgroup while there are more CUs than waves, then one per SIMD, then two.
    long work_lanes = -1;
    a.wave_lanes = 64;
    const long tiles16 = ((long)d->n_envs + 15) / 16, simds = 4L * c->num_cus;
    if (!per_env && tiles16 <= 8 * simds) {
        const int tpw = (int)((tiles16 + 2 * simds - 1) / (2 * simds));   // 1..4
        const long waves = (tiles16 + tpw - 1) / tpw;
        BLOCK = waves <= c->num_cus ? 64 : (waves <= simds ? 256 : 512);
        work_lanes = waves * 64;
        a.wave_lanes = 16 * tpw;
    }
    // One such network on so few environments that every 8 of them get a SIMD: run as a population of copies of the policy
    // (image stride 0) with 8 rollouts each -- a wave-step is then the environment step plus a forward pass of 8 columns
    // with the weights in registers, against one of 16 columns from LDS (QQube, T = 600: 1 000 rollouts 2.24 -> 1.31 ms,
    // 4 096 rollouts 3.07 -> 1.83 ms).
    const bool copies = !per_env && HP == 64 && p.n_hidden == 2 && all_tanh && (long)d->n_envs <= 8 * simds;
    const bool copies32_ = !per_env && HP == 32 && p.n_hidden == 2 && all_tanh && (long)d->n_envs <= 8 * simds;
    if ((copies || copies32_) && (a.flags & PSH_F_INTERNAL_ACT_NORM)) {  // (population kernels: the lone ActNormWrapper stays a wrapper)
        a.flags &= ~PSH_F_INTERNAL_ACT_NORM;
        wrapped = true;
    }
    // (groups of 4 while every 4 of them get a SIMD: one column group)
    const int grp = copies ? ((long)d->n_envs <= 4 * simds ? 4 : 8) : p.policy_group;
    const int npol = copies ? (int)(((long)d->n_envs + grp - 1) / grp) : p.n_policies;
    if ((per_env || copies) && HP == 64 && p.n_hidden == 2 && all_tanh) {
        // population of two-hidden-layer tanh networks wider than 32: hidden-to-hidden weights in registers, one wave
        // per SIMD (see WREG at the kernel).  Up to 32 rollouts per policy: two policies per wave, 16 rollouts per task;
        // more: the whole wave on one policy, 64 rollouts per task.
        const int block_l1 = HT * KS * 64;
        const bool two = grp <= 32;
        const int sets = two ? 2 : 1;
        // one policy per wave: a task takes 64 lanes, or 32 / 16 as long as that still leaves every task a SIMD of its own
        // (512 policies x 64 rollouts: 1 024 waves of two tiles instead of 512 waves of four on half of the chip)
        int task_lanes = two ? 16 : 64;
        if (!two) {
            const long simds = 4L * c->num_cus;
            while (task_lanes > 16 && (long)npol * ((grp + task_lanes / 2 - 1) / (task_lanes / 2)) <= simds) task_lanes /= 2;
        }
        a.lds_doubles = total - block_l1;
        a.pop_tiles = task_lanes / 16;
        a.pop_groups = two ? 2 : 1;
        // (a population with a SIMD for every task: one policy per wave, whose step then carries one forward pass, not two)
        const long tasks16 = (long)npol * ((grp + 15) / 16);
        const int per_wave = (two && tasks16 > 4L * c->num_cus) ? 2 : 1;
        if (two) a.pop_groups = per_wave;
        a.tanh_tab = c->tanh_tab + (PSH_TANH8_N - 1);
        a.tanh_n = PSH_TANH8_N;
        const size_t lds_w = (4 * ((size_t)sets * a.lds_doubles + OP * 64) + TASK_DOUBLES + TANH8_PAD) * sizeof(double);
        if (lds_w <= c->lds_per_block) {
            if (copies) { a.pol.policy_group = grp; a.pol.n_policies = npol; a.img_doubles = 0; }
            const long tasks = (long)npol * ((grp + task_lanes - 1) / task_lanes);
            const long lanes = (tasks + per_wave - 1) / per_wave * 64;   // a wave takes `per_wave` tasks at a time
#define PSH_LAUNCH_WREG(WRAP_, SETS_, FIXT_) launch_persistent(c, k_rollout_fnn<KIND, 64, true, WRAP_, true, true, SETS_, FIXT_>, 256, lds_w, a, E::A, WRAP_ && has_delay, st, lanes)
            const bool fixt = task_has_default_shape<KIND>(d->task);   // the default task shape compiled in
            // (up to 8 rollouts per policy: the layers on 4x4x4 MFMAs, two column groups of 4 instead of a tile of 16)
            if (two) {
                // one policy per wave, two waves per SIMD once there are more tasks than SIMDs (WREG = 5, 6, 7):
                // up to 8 rollouts in two column groups of 4, up to 4 in one, more in tiles of 16
                const int block5 = tasks16 > 4L * c->num_cus ? 512 : 256;
                const size_t lds5 = ((block5 / 64) * ((size_t)a.lds_doubles + OP * 64) + TASK_DOUBLES + TANH8_PAD) * sizeof(double);
                const long lanes5 = tasks16 * 64;
                a.pop_groups = 1;
#define PSH_LAUNCH_WREG5(WRAP_, SETS_, FIXT_) launch_persistent(c, k_rollout_fnn<KIND, 64, true, WRAP_, true, true, SETS_, FIXT_>, block5, lds5, a, E::A, WRAP_ && has_delay, st, lanes5)
#define PSH_LAUNCH_WREG5_ALL(SETS_) (wrapped ? (fixt ? PSH_LAUNCH_WREG5(true, SETS_, true) : PSH_LAUNCH_WREG5(true, SETS_, false)) : (fixt ? PSH_LAUNCH_WREG5(false, SETS_, true) : PSH_LAUNCH_WREG5(false, SETS_, false)))
                if (grp <= 4) return PSH_LAUNCH_WREG5_ALL(6);
                if (grp <= 8) return PSH_LAUNCH_WREG5_ALL(5);
                if (block5 == 512) return PSH_LAUNCH_WREG5_ALL(7);
#undef PSH_LAUNCH_WREG5_ALL
#undef PSH_LAUNCH_WREG5
                // (tiles of 16 with a SIMD for every task: the kernel with all 512 registers of a wave alone on its SIMD is
                // ahead -- 256 policies x 16 rollouts 3.1 against 3.5 ms)
                a.pop_groups = per_wave;
                return wrapped ? (fixt ? PSH_LAUNCH_WREG(true, 2, true) : PSH_LAUNCH_WREG(true, 2, false)) : (fixt ? PSH_LAUNCH_WREG(false, 2, true) : PSH_LAUNCH_WREG(false, 2, false));
            }
            return wrapped ? (fixt ? PSH_LAUNCH_WREG(true, 1, true) : PSH_LAUNCH_WREG(true, 1, false)) : (fixt ? PSH_LAUNCH_WREG(false, 1, true) : PSH_LAUNCH_WREG(false, 1, false));
#undef PSH_LAUNCH_WREG
        }
        a.lds_doubles = total;
    }
    // (one 32-wide policy on up to 8 environments per SIMD: as copies of the policy with 8 rollouts each, as above)
    const bool copies32 = !per_env && HP == 32 && p.n_hidden == 2 && all_tanh && (long)d->n_envs <= 8 * simds;
    const int grp32 = copies32 ? 8 : p.policy_group, npol32 = copies32 ? (int)(((long)d->n_envs + 7) / 8) : p.n_policies;
    if ((per_env || copies32) && HP == 32 && p.n_hidden == 2 && all_tanh && grp32 <= 32) {
        // the same for 32-wide networks with up to 32 rollouts per policy: two policies per wave, workgroups of 8 waves (two per
        // SIMD with 256 registers each) -- 16 policies resident per CU on all four SIMDs, where their LDS images allowed 8 to 12
        // on two or three
        const int block_l1 = HT * KS * 64, waves = 8;
        a.lds_doubles = total - block_l1;
        a.pop_tiles = 1;
        const long tasks = (long)npol32 * ((grp32 + 15) / 16);
        const int per_wave = tasks > (long)waves * c->num_cus ? 2 : 1;
        a.pop_groups = per_wave;
        a.tanh_tab = c->tanh_tab + (PSH_TANH8_N - 1);
        a.tanh_n = PSH_TANH8_N;
        const long lanes = (tasks + per_wave - 1) / per_wave * 64;
        // no more waves per workgroup than keep every CU busy
        int W = waves;
        while (W > 1 && lanes / 64 < (long)c->num_cus * W) W -= 1;
        const size_t lds_w = (W * (2 * (size_t)a.lds_doubles + OP * 64) + TASK_DOUBLES + TANH8_PAD) * sizeof(double);
        if (lds_w <= c->lds_per_block) {
            if (copies32) { a.pol.policy_group = grp32; a.pol.n_policies = npol32; a.img_doubles = 0; }
#define PSH_LAUNCH_WREG32(WRAP_, SETS_, FIXT_) launch_persistent(c, k_rollout_fnn<KIND, 32, true, WRAP_, true, true, SETS_, FIXT_>, 64 * W, lds_w, a, E::A, WRAP_ && has_delay, st, lanes)
            const bool fixt = task_has_default_shape<KIND>(d->task);
            if (grp32 <= 8) return wrapped ? (fixt ? PSH_LAUNCH_WREG32(true, 3, true) : PSH_LAUNCH_WREG32(true, 3, false)) : (fixt ? PSH_LAUNCH_WREG32(false, 3, true) : PSH_LAUNCH_WREG32(false, 3, false));
            return wrapped ? (fixt ? PSH_LAUNCH_WREG32(true, 2, true) : PSH_LAUNCH_WREG32(true, 2, false)) : (fixt ? PSH_LAUNCH_WREG32(false, 2, true) : PSH_LAUNCH_WREG32(false, 2, false));
#undef PSH_LAUNCH_WREG32
        }
        a.lds_doubles = total;
    }
    if (per_env) {
        // population: the tiles of a wave in groups of G, one policy per group, the weight images of a wave's groups in
        // LDS next to its staging area.  G = 1, 2, 4 from the rollouts per policy (a group holds up to 16 G of them),
        // waves per workgroup W and tanh table chosen for the most policies resident per CU; among equals fewer,
        // fuller waves (the environment step costs a wave the same with 16 or 64 busy lanes), then the fine table.
        const size_t fixed_fine = ((size_t)TASK_DOUBLES + TANH8_PAD) * sizeof(double);
        const size_t fixed_coarse = ((size_t)TASK_DOUBLES + TANH_TAB_PAD) * sizeof(double);
        const int g_min = p.policy_group > 32 ? 4 : (p.policy_group > 16 ? 2 : 1);
        long W = 0, best = 0;
        int G = 4;
        bool fine = true;
        for (int g = g_min; g <= 4; g *= 2) {
            const size_t per_wave = ((size_t)(4 / g) * total + OP * 64) * sizeof(double);
            for (int f = 1; f >= (HP >= 32 ? 0 : 1); --f) {  // the 16-wide kernels exist with the fine table only
                const size_t fixed = f ? fixed_fine : fixed_coarse;
                long w = c->lds_per_block > fixed ? (long)((c->lds_per_block - fixed) / per_wave) : 0;
                w = w > 8 ? 8 : w;
                if (w * (4 / g) > best) { best = w * (4 / g); W = w; G = g; fine = f != 0; }
            }
        }
        if (W < 1) { set_err(c, "FNNPolicy too large for the LDS of one workgroup"); return PSH_ERR_UNSUPPORTED; }
        // (a small population: the rollouts of a policy in tasks of fewer tiles while every task still gets a SIMD of its own)
        while (G > 1 && (long)p.n_policies * ((p.policy_group + 8 * G - 1) / (8 * G)) <= 4L * c->num_cus) G /= 2;
        const long GL = 16 * G, tasks = (long)p.n_policies * ((p.policy_group + GL - 1) / GL);
        // fewer groups per wave while that still gives every task a group at once (a wave's step carries the forward
        // passes of all its groups, one after the other), and no more waves per workgroup than keep every CU busy
        int groups = 4 / G;
        while (groups > 1 && tasks <= 4L * c->num_cus * (groups / 2)) groups /= 2;
        while (W > 1 && tasks < (long)c->num_cus * W * groups) W -= 1;
        const size_t waves_bytes = (size_t)W * ((size_t)groups * total + OP * 64) * sizeof(double);
        fine = fine || waves_bytes + fixed_fine <= c->lds_per_block;  // fewer waves may leave room for the fine table
        const int BLOCK_POP = 64 * (int)W;
        a.pop_tiles = G;
        a.pop_groups = groups;
        a.tanh_tab = fine ? c->tanh_tab + (PSH_TANH8_N - 1) : c->tanh_tab + PSH_TANH_ODD_N;
        a.tanh_n = fine ? PSH_TANH8_N : PSH_TANH_TAB_N;
        const size_t lds_pop = waves_bytes + (fine ? fixed_fine : fixed_coarse);
        const long lanes = (tasks + a.pop_groups - 1) / a.pop_groups * 64;
#define PSH_LAUNCH_POP(HP_, FINE_, TANH_)                                                                                \
    (wrapped ? launch_persistent(c, k_rollout_fnn<KIND, HP_, FINE_, true, true, TANH_>, BLOCK_POP, lds_pop, a, E::A, has_delay, st, lanes) \
             : launch_persistent(c, k_rollout_fnn<KIND, HP_, FINE_, false, true, TANH_>, BLOCK_POP, lds_pop, a, E::A, false, st, lanes))
        // (the usual case again with kernels of its own: bare simulation, default task shape, all-tanh, fine table)
        if (all_tanh && fine && task_has_default_shape<KIND>(d->task)) {
#define PSH_LAUNCH_POP_FIXT(HP_)                                                                                                           \
    (wrapped ? launch_persistent(c, k_rollout_fnn<KIND, HP_, true, true, true, true, 0, true>, BLOCK_POP, lds_pop, a, E::A, has_delay, st, lanes) \
             : launch_persistent(c, k_rollout_fnn<KIND, HP_, true, false, true, true, 0, true>, BLOCK_POP, lds_pop, a, E::A, false, st, lanes))
            if (HP == 16) return PSH_LAUNCH_POP_FIXT(16);
            if (HP == 32) return PSH_LAUNCH_POP_FIXT(32);
            return PSH_LAUNCH_POP_FIXT(64);
#undef PSH_LAUNCH_POP_FIXT
        }
        if (HP == 16) return all_tanh ? PSH_LAUNCH_POP(16, true, true) : PSH_LAUNCH_POP(16, true, false);
        if (HP == 32 && fine) return all_tanh ? PSH_LAUNCH_POP(32, true, true) : PSH_LAUNCH_POP(32, true, false);
        if (HP == 32) return all_tanh ? PSH_LAUNCH_POP(32, false, true) : PSH_LAUNCH_POP(32, false, false);
        if (fine) return all_tanh ? PSH_LAUNCH_POP(64, true, true) : PSH_LAUNCH_POP(64, true, false);
        return all_tanh ? PSH_LAUNCH_POP(64, false, true) : PSH_LAUNCH_POP(64, false, false);
#undef PSH_LAUNCH_POP
    }
    const size_t base_doubles = (size_t)total + (BLOCK / 64) * OP * 64 + TASK_DOUBLES;
    const bool fine = (base_doubles + TANH8_PAD) * sizeof(double) <= c->lds_per_block;
    a.tanh_tab = fine ? c->tanh_tab + (PSH_TANH8_N - 1) : c->tanh_tab + PSH_TANH_ODD_N;
    a.tanh_n = fine ? PSH_TANH8_N : PSH_TANH_TAB_N;
    const size_t lds_bytes = (base_doubles + (fine ? TANH8_PAD : TANH_TAB_PAD)) * sizeof(double);
    if (lds_bytes > c->lds_per_block) { set_err(c, "FNNPolicy too large for the LDS of one workgroup"); return PSH_ERR_UNSUPPORTED; }
#define PSH_LAUNCH_FNN(HP_, FINE_, TANH_)                                                                                \
    (wrapped ? launch_persistent(c, k_rollout_fnn<KIND, HP_, FINE_, true, false, TANH_>, BLOCK, lds_bytes, a, E::A, has_delay, st, work_lanes) \
             : launch_persistent(c, k_rollout_fnn<KIND, HP_, FINE_, false, false, TANH_>, BLOCK, lds_bytes, a, E::A, false, st, work_lanes))
    // the narrower kernels exist with the fine table only (on gfx950 their weight images always leave room for it)
    if (HP <= 32 && !fine) { set_err(c, "no LDS for the tanh table of this FNNPolicy"); return PSH_ERR_UNSUPPORTED; }
    // the usual case has kernels of its own: bare simulation with its default task shape, all-tanh network, fine table
    if (all_tanh && fine && task_has_default_shape<KIND>(d->task)) {
        // (the bare kernel with the odd tanh table when that table fits next to the staging areas of this block size:
        // twice the LDS of the fine table for one vector instruction less per activation, ~1 % of a launch)
        const size_t lds_sgn = ((size_t)total + (BLOCK / 64) * E::O * 64 + TASK_DOUBLES + TANH_ODD_PAD) * sizeof(double);
        if (!wrapped && HP == 64 && lds_sgn <= c->lds_per_block) {
            a.tanh_tab = c->tanh_tab;
            a.tanh_n = PSH_TANH_ODD_N;
            return launch_persistent(c, k_rollout_fnn<KIND, 64, true, false, false, true, 0, true, true>, BLOCK, lds_sgn, a, E::A, false, st, work_lanes);
        }
#define PSH_LAUNCH_FNN_FIXT(HP_)                                                                                                           \
    (wrapped ? launch_persistent(c, k_rollout_fnn<KIND, HP_, true, true, false, true, 0, true>, BLOCK, lds_bytes, a, E::A, has_delay, st, work_lanes) \
             : launch_persistent(c, k_rollout_fnn<KIND, HP_, true, false, false, true, 0, true>, BLOCK, lds_bytes, a, E::A, false, st, work_lanes))
        if (HP == 16) return PSH_LAUNCH_FNN_FIXT(16);
        if (HP == 32) return PSH_LAUNCH_FNN_FIXT(32);
        return PSH_LAUNCH_FNN_FIXT(64);
#undef PSH_LAUNCH_FNN_FIXT
    }
    if (HP == 16) return all_tanh ? PSH_LAUNCH_FNN(16, true, true) : PSH_LAUNCH_FNN(16, true, false);
    if (HP == 32) return all_tanh ? PSH_LAUNCH_FNN(32, true, true) : PSH_LAUNCH_FNN(32, true, false);
    if (fine) return all_tanh ? PSH_LAUNCH_FNN(64, true, true) : PSH_LAUNCH_FNN(64, true, false);
    return all_tanh ? PSH_LAUNCH_FNN(64, false, true) : PSH_LAUNCH_FNN(64, false, false);
#undef PSH_LAUNCH_FNN
}
