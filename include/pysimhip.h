/*
 * pysimhip.h — C ABI of libpysimhip.so, the MI355X (gfx950) batched rollout engine that sits behind
 * the Pyrado object protocol for the standalone-Python simulations (SimPyEnv subclasses).
 *
 * The reference has no FFI for this path: its boundary is the Python object protocol
 *   Env.reset / Env.step            Pyrado/pyrado/environments/pysim/base.py:119-201
 *   rollout()                       Pyrado/pyrado/sampling/rollout.py:27-291
 *   DomainRandomizer.randomize()    Pyrado/pyrado/domain_randomization/domain_randomizer.py:76-94
 * Every entry point below names the reference function(s) it replaces.
 *
 * Conventions
 *  - plain C, no exceptions, no torch types.  Return 0 on success, a negative psh_status otherwise;
 *    psh_last_error(ctx) gives the message of the last failure on that context.
 *  - every `double*` / `int32_t*` marked [dev] is a caller-owned DEVICE pointer (hipMalloc or a
 *    PyTorch-ROCm tensor), fp64 / int32, contiguous, environment-minor: element (row r, env i) of an
 *    [R, N] array lives at base[r * ld + i] with ld = the stated leading dimension (normally N).
 *  - struct members marked [host] are read on the host while the call is being issued.
 *  - calls are asynchronous on `stream` (a hipStream_t passed as void*; NULL = the null stream);
 *    the library never synchronises the device except in psh_destroy.
 *  - one psh_ctx per (device, host thread).  A ctx is not thread-safe; distinct ctxs are independent.
 *    The rollouts of one ctx share its scratch buffers: the library orders them on the device (a rollout issued on
 *    another stream waits for the previous rollout of the same ctx), so one rollout per ctx runs at a time.
 *  - NaN never aborts a kernel: the per-env `status` word carries PSH_ST_NAN and the host layer raises
 *    the reference's ValueErr (Pyrado/pyrado/sampling/rollout.py:141-147,187-193).
 *  - Infinite values are not an error in the reference and none here: they go through the dynamics, the rewards and
 *    the policies as IEEE arithmetic takes them through the reference's NumPy / torch code (x / inf = 0, 0 * inf = NaN
 *    in a matrix product, tanh(inf) = 1, relu(NaN) = NaN), so an environment whose state has left the finite range ends
 *    with the status word, length and values the reference computes for it (a finite state beyond 1e154, whose square
 *    overflows in the reward, excepted: -inf or NaN there depends on the order of the sums).
 */
#ifndef PYSIMHIP_H
#define PYSIMHIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PSH_VERSION 111

#define PSH_MAX_STATE 8   /* largest state vector (QBallBalancer)                 */
#define PSH_MAX_OBS 8     /* largest observation vector                            */
#define PSH_MAX_ACT 2     /* largest action vector (QBallBalancer)                 */
#define PSH_MAX_DP 20     /* largest domain-parameter set (QBallBalancer)          */
#define PSH_MAX_HIDDEN 4  /* hidden layers of an FNN policy                        */
#define PSH_MAX_WIDTH 64  /* widest hidden layer of an FNN policy                  */
#define PSH_MAX_FEATS 8   /* feature functions in a LinearPolicy's FeatureStack    */
#define PSH_MAX_LIN_FEAT 512 /* feature values of a LinearPolicy in total          */
#define PSH_MAX_DISTR 32  /* distributions in one randomizer                       */
#define PSH_MAX_DELAY 32  /* longest action delay (steps) of the fused ActDelay    */
#define PSH_MAX_WRAP 6    /* fused environment wrappers around one simulation      */

typedef enum psh_status {
    PSH_OK = 0,
    PSH_ERR_ARG = -1,     /* invalid descriptor (NULL pointer, bad size, unknown enum)   */
    PSH_ERR_HIP = -2,     /* a HIP runtime call failed                                   */
    PSH_ERR_NODEV = -3,   /* no usable gfx950 device                                     */
    PSH_ERR_UNSUPPORTED = -4
} psh_status;

/* Environment kinds.  Row order of the domain-parameter table = key order of the reference's
 * get_nominal_domain_param() dicts:
 *  QQ   (10) g Rm km Mr Lr Dr Mp Lp Dp max_volt        pysim/quanser_qube.py:61-71
 *  QCP  (14) g m_cart l_rail eta_m eta_g K_g J_m r_mp R_m k_m B_pole B_eq m_pole l_pole
 *                                                       pysim/quanser_cartpole.py:66-81
 *  QBB  (20) g m_ball r_ball l_plate r_arm K_g eta_g J_l J_m k_m R_m eta_m B_eq c_frict
 *            V_thold_x_pos V_thold_x_neg V_thold_y_pos V_thold_y_neg offset_th_x offset_th_y
 *                                                       pysim/quanser_ball_balancer.py:131-152
 *  BOB  (8)  g m_ball r_ball m_beam l_beam d_beam c_frict ang_offset   pysim/ball_on_beam.py:49-56
 *  OMO  (3)  m k d                                      pysim/one_mass_oscillator.py:55-57
 *  PEND (5)  g m_pole l_pole d_pole tau_max             pysim/pendulum.py:41-47
 *  QQ_ST     as QQ                                      pysim/quanser_qube.py:222-271        */
typedef enum psh_env_kind {
    PSH_ENV_QQ = 0,      /* QQubeSim (QQubeSwingUpSim)   S=4 O=6 A=1 P=10  init 4 */
    PSH_ENV_QCP_SU = 1,  /* QCartPoleSwingUpSim          S=4 O=5 A=1 P=14  init 4 */
    PSH_ENV_QCP_ST = 2,  /* QCartPoleStabSim             S=4 O=5 A=1 P=14  init 4 */
    PSH_ENV_QBB = 3,     /* QBallBalancerSim             S=8 O=8 A=2 P=20  init 4 (polar-sampled x,y,xd,yd) or 8 */
    PSH_ENV_BOB = 4,     /* BallOnBeamSim                S=4 O=4 A=1 P=8   init 4 */
    PSH_ENV_OMO = 5,     /* OneMassOscillatorSim         S=2 O=2 A=1 P=3   init 2 */
    PSH_ENV_PEND = 6,    /* PendulumSim                  S=2 O=3 A=1 P=5   init 2 (always the zero state) */
    PSH_ENV_QQ_ST = 7,   /* QQubeStabSim                 S=4 O=4 A=1 P=10  init 4 (observes des - state) */
    PSH_ENV_COUNT = 8
} psh_env_kind;

/* Reward functions, Pyrado/pyrado/tasks/reward_functions.py */
typedef enum psh_rew_kind {
    PSH_REW_QUADR = 0,            /* QuadrErrRewFcn            :167-186  r = -cost                */
    PSH_REW_EXP_QUADR = 1,        /* ExpQuadrErrRewFcn         :202-208  r = exp(-cost)           */
    PSH_REW_SCALED_EXP_QUADR = 2, /* ScaledExpQuadrErrRewFcn   :241-262  r = exp(-c_max*cost)     */
    PSH_REW_SWINGUP = 3           /* UnderActuatedSwingUpRewFcn :299-303                          */
} psh_rew_kind;

/* FinalRewMode bits, Pyrado/pyrado/tasks/final_reward.py:12-18,70-187.  0 = no FinalRewTask. */
#define PSH_FIN_ENABLED 1
#define PSH_FIN_ALWAYS_POSITIVE 2
#define PSH_FIN_ALWAYS_NEGATIVE 4
#define PSH_FIN_STATE_DEPENDENT 8
#define PSH_FIN_TIME_DEPENDENT 16

/* Task = desired state + reward function + final reward.
 * DesStateTask.step_rew            Pyrado/pyrado/tasks/desired_state.py:82-85
 * RadiallySymmDesStateTask.step_rew                       :119-128 (fmod on `radial_mask` dims, then
 *                                  the +-pi fold on EVERY dim, as the reference does)
 * FinalRewTask.compute_final_rew   Pyrado/pyrado/tasks/final_reward.py:70-187 */
typedef struct psh_task {
    int32_t rew_kind;     /* psh_rew_kind */
    int32_t radial;       /* 1: RadiallySymmDesStateTask, 0: DesStateTask */
    int32_t radial_mask;  /* bit i set: state dim i is taken fmod(err, mod[i]) */
    int32_t final_mode;   /* PSH_FIN_* bits */
    int32_t idx_x;        /* swing-up reward: index of the cart position  */
    int32_t idx_th;       /* swing-up reward: index of the pole angle     */
    int32_t pad_[2];
    double final_factor;  /* FinalRewTask.factor */
    double min_rew;       /* ScaledExpQuadrErrRewFcn min_rew (c_max is derived per env from its bounds) */
    double mod[PSH_MAX_STATE];        /* modulation per state dim (2*pi by default) */
    double state_des[PSH_MAX_STATE];
    double Q[PSH_MAX_STATE * PSH_MAX_STATE]; /* row-major [S,S], dense */
    double R[PSH_MAX_ACT * PSH_MAX_ACT];     /* row-major [A,A], dense */
    double swingup[5];    /* c_pole, c_cart, c_act, c_theta_sq, c_theta_log */
} psh_task;

/* Policies.  LinearPolicy.forward Pyrado/pyrado/policies/linear.py:65-80 with the element-wise
 * feature functions of Pyrado/pyrado/policies/features.py:90-140; FNN.forward
 * Pyrado/pyrado/policies/fnn.py:129-151.  `params` uses the order of
 * torch.nn.utils.parameters_to_vector(policy.parameters()) (Pyrado/pyrado/policies/base.py:56-70):
 * Linear: W[A,F] row-major; FNN: W0[h0,O], b0[h0], W1[h1,h0], b1[h1], ..., Wout[A,h], bout[A]. */
typedef enum psh_policy_kind { PSH_POL_LINEAR = 0, PSH_POL_FNN = 1 } psh_policy_kind;

typedef enum psh_feat {
    PSH_FEAT_CONST = 0, PSH_FEAT_IDENTITY = 1, PSH_FEAT_SIGN = 2, PSH_FEAT_ABS = 3,
    PSH_FEAT_SQUARED = 4, PSH_FEAT_CUBED = 5, PSH_FEAT_SIG = 6, PSH_FEAT_BELL = 7,
    PSH_FEAT_SIN = 8, PSH_FEAT_COS = 9, PSH_FEAT_SINSIN = 10, PSH_FEAT_SINCOS = 11,
    /* Feature objects with data of their own (Pyrado/pyrado/policies/features.py:143-303); O is the dimension of the
     * observation the policy sees.
     *  MULT  MultFeat(idcs) (:143-167): one value, the product of obs[idcs[..]].  feat_size = number of indices
     *        (2..8), feat_arg = the indices, 4 bits each, first index in the lowest bits.
     *  RFF   RandFourierFeat (:170-219): feat_size = D values scale * cos(obs . freq[j] + shift[j]).
     *        feat_data at feat_off: scale, shift[D], freq[D, O] row-major.
     *  RBF   RBFFeat (:222-303): feat_size = K centers per input dimension, K * O values ordered dimension by
     *        dimension, exp(-scale[q] (obs[q] - centers[k, q])^2) divided by its L1 norm over k (feat_arg = 1,
     *        state_wise_norm) or over all K * O values (feat_arg = 0); a norm <= 1e-8 divides by 1
     *        (Pyrado/pyrado/utils/normalizing.py:87-105).  feat_data at feat_off: scale[O], centers[K, O] row-major. */
    PSH_FEAT_MULT = 12, PSH_FEAT_RFF = 13, PSH_FEAT_RBF = 14
} psh_feat;

typedef enum psh_nonlin { PSH_NL_NONE = 0, PSH_NL_TANH = 1, PSH_NL_RELU = 2, PSH_NL_SIGMOID = 3 } psh_nonlin;

typedef enum psh_expl_kind { PSH_EXPL_NONE = 0, PSH_EXPL_NORMAL = 1, PSH_EXPL_UNIFORM = 2 } psh_expl_kind;

typedef struct psh_policy {
    int32_t kind;                          /* psh_policy_kind */
    int32_t n_feats;                       /* LINEAR: length of feats[] */
    int32_t feats[PSH_MAX_FEATS];          /* LINEAR: psh_feat ids in FeatureStack order */
    int32_t n_hidden;                      /* FNN: number of hidden layers (1..PSH_MAX_HIDDEN) */
    int32_t hidden[PSH_MAX_HIDDEN];        /* FNN: hidden sizes (each <= PSH_MAX_WIDTH) */
    int32_t hidden_nonlin[PSH_MAX_HIDDEN]; /* FNN: psh_nonlin per hidden layer */
    int32_t out_nonlin;                    /* FNN: psh_nonlin of the output layer */
    int32_t n_policies;                    /* rows of params (1 unless population mode) */
    int32_t num_param;                     /* parameters per policy (checked against the layout) */
    int32_t pad_;
    const double* params;                  /* [dev] [n_policies, num_param] row-major */
    const int32_t* policy_index;           /* [dev] [N] row of params used by env i, or NULL (row 0); LINEAR only */
    /* Action-noise exploration around the policy, StochasticActionExplStrat.forward
     * (Pyrado/pyrado/exploration/stochastic_action.py:42-68): the exploratory action is what the rollout records
     * and what the environment chain receives.
     *  NORMAL   NormalActNoiseExplStrat  (:96-136, exploration/normal_noise.py:100-108)
     *           a + expl_mean + expl_scale * eps,  eps ~ N(0, 1);   expl_scale = std
     *  UNIFORM  UniformActNoiseExplStrat (:139-176, exploration/uniform_noise.py)
     *           lo + u (hi - lo) with lo/hi = a + expl_mean -/+ expl_scale, u ~ U[0, 1);   expl_scale = halfspan
     * Drawn from Philox4x32-10 keyed like the wrapper noise (psh_wrappers.noise_seed / env_offset, step index),
     * on a stream of its own. */
    int32_t expl_kind;                     /* psh_expl_kind */
    int32_t policy_group;                  /* > 0: population mode of ParameterExplorationSampler.sample
                                              (Pyrado/pyrado/sampling/parameter_exploration_sampler.py:174-201): N =
                                              n_policies * policy_group and env i uses row i / policy_group of params
                                              (policy_index must be NULL).  LINEAR and FNN. */
    double expl_mean[PSH_MAX_ACT];
    double expl_scale[PSH_MAX_ACT];
    int32_t feat_size[PSH_MAX_FEATS];      /* LINEAR: MULT / RFF / RBF, see psh_feat; 0 for the other features */
    int32_t feat_arg[PSH_MAX_FEATS];
    int32_t feat_off[PSH_MAX_FEATS];       /* offset (doubles) of the feature's data within feat_data */
    int32_t feat_data_len;                 /* doubles in feat_data */
    int32_t pad2_;
    const double* feat_data;               /* [dev] data of the RFF / RBF features, NULL without any */
} psh_policy;

/* rollout flags */
#define PSH_F_STOP_ON_DONE 1      /* rollout(stop_on_done=True), the default of the reference          */
#define PSH_F_SIMPLIFIED_DYN 2    /* QBallBalancerSim(simplified_dyn=True)                              */
#define PSH_F_MAX_STEPS_INF 4     /* env.max_steps == inf inside step(): remaining_steps = 0, no time-out done
                                     (pysim/base.py:179,194); only meaningful for psh_step               */
#define PSH_F_TRAJ_ENV_MAJOR 8    /* psh_rollout: trajectories one environment after the other, obs [N, T+1, O'],
                                     act [N, T, A], rew [N, T]: every rollout is a contiguous block, as the
                                     StepSequence of the reference holds it (sampling/step_sequence.py).  The
                                     same numbers at other addresses, and slower to write: every lane stores
                                     8-byte pieces into a block of its own (full-length OneMassOscillator
                                     rollouts: 6 x the time and 5.7 x the bytes written of the default
                                     layout; episodes of 70 steps: about equal) */

/* per-env status bits */
#define PSH_ST_NAN 1      /* NaN in observation, action or state: the reference raises ValueErr */
#define PSH_ST_FAILED 2   /* state left the (inclusive) state-space box: task.has_failed        */
#define PSH_ST_TIMEOUT 4  /* curr_step reached max_steps                                        */
#define PSH_ST_BERNOULLI 8 /* rollout ended by the stochastic reset (bernoulli_reset)           */
#define PSH_ST_BAD_INDEX 16 /* dp_index[i] or policy_index[i] points outside its table: the env is
                              not run (length 0, return 0); nothing outside the tables is read  */
#define PSH_ST_INTERNAL 32 /* the lanes of a wave disagreed about a decision that is theirs in common (population mode:
                             whether their group still draws from the work queue).  Never set by a correct run; the
                             kernel carries on from the wave-wide view, the host layer raises on it.               */

/* Environment wrappers fused into the rollout (Pyrado/pyrado/environment_wrappers/): what EnvWrapperAct.step
 * and EnvWrapperObs.reset/step (base.py:145-231) do around the simulation each step.  `ops` lists the wrappers
 * OUTERMOST FIRST (domain-randomization wrappers are not part of it); each kind may appear once.  The action of the
 * policy passes through the action wrappers in that order, the observation of the simulation through the
 * observation wrappers in the reverse order, as the nesting of the Python objects implies.
 *  ACT_NORM     ActNormWrapper._process_act           action_normalization.py:10-22
 *               a = lo + (a + 1) (hi - lo) / 2 with the simulation's own (parameter-dependent) action bounds
 *  ACT_DELAY    ActDelayWrapper.reset/_process_act    action_delay.py:55-81
 *               FIFO pre-filled with `delay` zero actions; delay = `act_delay`, or per env the integer in row
 *               `act_delay_row` of the domain-parameter table (the randomized 'act_delay' parameter, :46-53).
 *               The queue holds the caller's array OBJECTS: with no array-creating wrapper (ACT_NORM, ACT_NOISE)
 *               between it and QBallBalancerSim, the simulation's in-place dead-zone write lands in the action
 *               recorded `delay` steps earlier.
 *  ACT_NOISE    GaussianActNoiseWrapper._process_act  action_noise.py:42-47     a + randn * std + mean
 *  OBS_NOISE    GaussianObsNoiseWrapper._process_obs  observation_noise.py:34-39  o + randn * std + mean
 *  OBS_NORM     ObsNormWrapper._process_obs           observation_normalization.py:73-86
 *               (o - lo) / (hi - lo) * 2 - 1; obs_lo/obs_hi hold the bounds after the explicit overrides, NaN
 *               entries mean "the simulation's own bound for this env" (it may depend on its parameters)
 *  OBS_PARTIAL  ObsPartialWrapper._process_obs        observation_partial.py:43-44   o[keep_mask]
 * Vectors are indexed by the SIMULATION's observation / action index whatever the position of OBS_PARTIAL; the
 * policy sees, and `obs` records, the kept entries in order (their count is the policy's input dimension).
 * The reference draws the noise from NumPy's global generator; here it is Philox4x32-10 keyed by
 * (noise_seed, global env index = env_offset + i, step), so only the distribution can agree with it. */
typedef enum psh_wrap_op {
    PSH_WRAP_ACT_NORM = 1, PSH_WRAP_ACT_DELAY = 2, PSH_WRAP_ACT_NOISE = 3,
    PSH_WRAP_OBS_NOISE = 4, PSH_WRAP_OBS_NORM = 5, PSH_WRAP_OBS_PARTIAL = 6
} psh_wrap_op;

typedef struct psh_wrappers {
    int32_t n_ops;                /* 0: bare simulation */
    int32_t ops[PSH_MAX_WRAP];    /* psh_wrap_op, outermost first */
    int32_t act_delay;            /* ACT_DELAY: delay in steps (0..PSH_MAX_DELAY) when act_delay_row < 0 */
    int32_t act_delay_row;        /* ACT_DELAY: row of domain_param with the per-env delay, or -1 */
    int32_t obs_keep_mask;        /* OBS_PARTIAL: bit r set = observation r of the simulation is kept */
    int32_t pad_[2];
    uint64_t noise_seed;
    uint64_t env_offset;
    double act_noise_mean[PSH_MAX_ACT], act_noise_std[PSH_MAX_ACT];
    double obs_noise_mean[PSH_MAX_OBS], obs_noise_std[PSH_MAX_OBS];
    double obs_lo[PSH_MAX_OBS], obs_hi[PSH_MAX_OBS];
} psh_wrappers;

typedef struct psh_ctx psh_ctx;

/* ---- context ------------------------------------------------------------------------------- */
int psh_version(void);
/* Bind a context to HIP device `hip_device`.  Fails with PSH_ERR_NODEV when there is no GPU. */
int psh_create(int hip_device, psh_ctx** out);
void psh_destroy(psh_ctx* ctx);
const char* psh_last_error(const psh_ctx* ctx);
/* Static facts about an environment kind (no device needed): dims[0..4] = S, O, A, P, init_dim. */
int psh_env_dims(int env_kind, int32_t dims[5]);

/* ---- fused rollouts --------------------------------------------------------------------------
 * Replaces, for N environments at once, one call of rollout(env, policy, reset_kwargs={init_state,
 * domain_param}) each (Pyrado/pyrado/sampling/rollout.py:88-275), i.e. SimPyEnv.reset
 * (pysim/base.py:119-156, incl. the domain_param setter :75-87 -> _calc_constants/_create_spaces/
 * task.reset), the policy forward, SimPyEnv.step (:177-201) and the per-env _step_dynamics:
 *   QQubeSim              pysim/quanser_qube.py:73-142
 *   QCartPole*Sim         pysim/quanser_cartpole.py:83-124
 *   QBallBalancerSim      pysim/quanser_ball_balancer.py:154-271 (+ IK at reset :181-195,413-480)
 *   BallOnBeamSim         pysim/ball_on_beam.py:58-98
 *   OneMassOscillatorSim  pysim/one_mass_oscillator.py:59-84
 *   PendulumSim           pysim/pendulum.py:49-61
 *   QQubeStabSim          pysim/quanser_qube.py:222-271 (dynamics of QQubeSim)
 * One GPU lane owns one environment; state, domain parameters and derived constants stay in
 * registers for the whole horizon. */
typedef struct psh_rollout_desc {
    int32_t env_kind;    /* psh_env_kind */
    int32_t n_envs;      /* N >= 0 (0 is a no-op) */
    int32_t max_steps;   /* T >= 1 */
    int32_t flags;       /* PSH_F_* */
    double dt;
    /* initial states [init_dim, N], ld = ld_init.  init_dim = the env's init-space dim, or S to set the
     * full state (QBB: 4 -> (x,y,xd,yd) expanded by _state_from_init, 8 -> full state). */
    const double* init_state; /* [dev] */
    int32_t init_dim;
    int32_t ld_init;
    /* domain parameters: [P, M] table, ld = ld_dp.  dp_index == NULL: M == N and env i reads column i;
     * else env i reads column dp_index[i] (buffer / sweep mode: DomainRandWrapperBuffer,
     * Pyrado/pyrado/environment_wrappers/domain_randomization.py:179-199).  The kernels follow no entry outside
     * [0, ld_dp): such an env (likewise one whose policy_index entry is outside [0, n_policies)) is not run and
     * reports PSH_ST_BAD_INDEX; psh_sample_init, which has no status word, gives it a NaN init state. */
    const double* domain_param; /* [dev] */
    const int32_t* dp_index;    /* [dev] or NULL */
    int32_t ld_dp;
    int32_t dp_rows;            /* rows of domain_param; 0 = exactly the simulation's P.  wrap.act_delay_row must be below it */
    psh_task task;
    psh_policy policy;
    psh_wrappers wrap;          /* [host] fused environment wrappers; all-zero = none */
    double bernoulli_reset;     /* rollout(bernoulli_reset=p), sampling/rollout.py:248-254: after every step the rollout
                                   ends with probability p (0 = off); drawn from the Philox stream of wrap.noise_seed */
    /* outputs, all [dev]; ret/length/status are mandatory, the rest may be NULL */
    double* ret;          /* [N]   undiscounted return = sum of rewards (step_sequence.py:626-635) */
    int32_t* length;      /* [N]   number of executed steps (rollout length) */
    int32_t* status;      /* [N]   PSH_ST_* bits */
    double* final_state;  /* [S, N] ld = N, state after the last executed step */
    double* obs;          /* [T+1, O', N] observations incl. the final one (O' = O, or the kept entries of
                             OBS_PARTIAL); the rows of a rollout past its end (all three arrays) are either left as the
                             caller passed them or set to NaN (the LinearPolicy kernels pad whole 128-byte lines).  With PSH_F_TRAJ_ENV_MAJOR the three
                             trajectory arrays are [N, T+1, O'], [N, T, A], [N, T] instead */
    double* act;          /* [T, A, N]   recorded actions (raw policy output; QBB: dead-zoned, see ACT_DELAY) */
    double* rew;          /* [T, N]      step rewards (final reward folded into the last step) */
    void* stream;
} psh_rollout_desc;
int psh_rollout(psh_ctx* ctx, const psh_rollout_desc* desc);

/* psh_rollout for callers that cannot allocate device memory.  No counterpart in the reference: it exists for code
 * under HIP graph capture, tight serving loops and harnesses that forbid hipMalloc.  Same descriptor, same kernels and
 * bit-identical results, but the scratch of the launch (work-queue counter, packed FNN weights, QBallBalancer plate
 * angles) lives in a static device array of the library (about 8.1 MiB per GPU, held until the process ends) instead
 * of context-owned allocations, and the call itself performs no hipMalloc / hipFree.  Limits, rejected with
 * PSH_ERR_UNSUPPORTED and a message: the packed weights of all policies of the call at most
 * PSH_STATIC_MAX_WEIGHT_DOUBLES doubles (any single FNN the kernels support fits; a population of 64x64 networks up to
 * about 3), QBallBalancerSim with the full dynamics at most PSH_STATIC_MAX_QBB_ENVS environments, no fused
 * ActDelayWrapper (its ring buffers scale with the launch).  All calls on one GPU share the array, whatever their
 * context: the library orders them per GPU (each waits for an event recorded behind the previous one, on whatever
 * stream or context that was issued), so they run one after the other.  On a stream that is being captured into a graph
 * the call enqueues its memset and kernels only -- no event is waited for or recorded, no timing is taken -- and the
 * caller orders the replays of that graph against every other call that uses the array: replays that overlap with one
 * another or with another static call on the same GPU race. */
#define PSH_STATIC_MAX_WEIGHT_DOUBLES 16384
#define PSH_STATIC_MAX_QBB_ENVS 524288
int psh_rollout_static(psh_ctx* ctx, const psh_rollout_desc* desc);

/* ---- single steps for arbitrary (non-fusable) policies ---------------------------------------
 * psh_reset replaces SimPyEnv.reset (pysim/base.py:119-156) for N envs: state from init_state,
 * QBB plate angles by IK, step counter 0, first observation.
 * psh_step  replaces SimPyEnv.step  (pysim/base.py:177-201) for N envs with caller-supplied actions.
 * The environment memory is caller-owned: `state` has S rows (+2 hidden plate-angle rows for QBB,
 * pysim/quanser_ball_balancer.py:240-271), `curr_step` the step counters, `flags_io` bit0 =
 * "final reward already yielded" (tasks/final_reward.py:78-83). */
typedef struct psh_step_desc {
    int32_t env_kind;
    int32_t n_envs;
    int32_t max_steps;
    int32_t flags;
    double dt;
    const double* domain_param; /* [dev] [P, M] */
    const int32_t* dp_index;    /* [dev] or NULL */
    int32_t ld_dp;
    int32_t init_dim;           /* psh_reset only */
    const double* init_state;   /* [dev] psh_reset only: [init_dim, N] ld = N */
    psh_task task;
    double* state;       /* [dev] [S (+2 for QBB), N] in/out */
    int32_t* curr_step;  /* [dev] [N] in/out */
    int32_t* flags_io;   /* [dev] [N] in/out */
    double* act;         /* [dev] psh_step: [A, N] in; QBB writes the dead-zoned action back (in place,
                            as the reference does when the action was inside its bounds) */
    double* obs;         /* [dev] [O, N] out */
    double* rew;         /* [dev] psh_step: [N] out */
    int32_t* done;       /* [dev] psh_step: [N] out (0/1) */
    int32_t* status;     /* [dev] [N] out, PSH_ST_* bits */
    void* stream;
} psh_step_desc;
int psh_reset(psh_ctx* ctx, const psh_step_desc* desc);
int psh_step(psh_ctx* ctx, const psh_step_desc* desc);

/* ---- on-device domain randomization ----------------------------------------------------------
 * Replaces DomainRandomizer.randomize(n) + get_params
 * (Pyrado/pyrado/domain_randomization/domain_randomizer.py:76-140) and DomainParam.sample
 * (domain_parameter.py:67-91, Bernoulli :209-236): x ~ distr, clamp(x, clip_lo, clip_up), optional
 * round-to-int.  Counter-based Philox4x32-10 keyed by (seed, env_offset + i, row), so the value of
 * env i does not depend on N, the launch geometry or the number of GPUs.  Rows without a
 * distribution are filled with `nominal`. */
typedef enum psh_distr_kind { PSH_DISTR_CONST = 0, PSH_DISTR_NORMAL = 1, PSH_DISTR_UNIFORM = 2, PSH_DISTR_BERNOULLI = 3 } psh_distr_kind;

typedef struct psh_distr {
    int32_t kind;     /* psh_distr_kind */
    int32_t row;      /* row of the output table this distribution writes */
    int32_t roundint; /* 1: round to nearest integer after clipping */
    int32_t pad_;
    double a;         /* NORMAL/UNIFORM: mean;  BERNOULLI: val_0;  CONST: value */
    double b;         /* NORMAL: std;  UNIFORM: halfspan;  BERNOULLI: val_1 */
    double c;         /* BERNOULLI: prob_1 */
    double clip_lo;
    double clip_up;
} psh_distr;

typedef struct psh_rand_desc {
    int32_t n_envs;
    int32_t n_rows;                  /* rows of `out` (P, or P+1 with an act_delay row) */
    int32_t n_distr;
    int32_t ld_out;
    uint64_t seed;
    uint64_t env_offset;             /* global index of env 0 of this call */
    psh_distr distr[PSH_MAX_DISTR];  /* [host] */
    double nominal[PSH_MAX_DP + 4];  /* [host] per-row fill value */
    double* out;                     /* [dev] [n_rows, N] */
    void* stream;
} psh_rand_desc;
int psh_randomize(psh_ctx* ctx, const psh_rand_desc* desc);

/* Replaces init_space.sample_uniform() of SimPyEnv.reset (pysim/base.py:127-129) for N envs:
 * BoxSpace (Pyrado/pyrado/spaces/box.py:133-142), Polar2DPosVelSpace (spaces/polar.py:64-69, QBB),
 * CompoundSpace of two boxes (spaces/compound.py:51-54, BOB).  The init spaces of BOB and QBB depend
 * on the env's own domain parameters (l_beam, l_plate), read from `domain_param`.
 * Philox keyed by (seed, env_offset + i) on a stream of its own: the same seed and offset may be handed to
 * psh_randomize, psh_sample_init and the noise streams of psh_rollout without any two of them sharing a draw. */
typedef struct psh_init_desc {
    int32_t env_kind;
    int32_t n_envs;
    int32_t ld_dp;
    int32_t ld_out;
    uint64_t seed;
    uint64_t env_offset;
    const double* domain_param; /* [dev] [P, M] */
    const int32_t* dp_index;    /* [dev] or NULL */
    double* out;                /* [dev] [init_dim, N] */
    void* stream;
} psh_init_desc;
int psh_sample_init(psh_ctx* ctx, const psh_init_desc* desc);

/* QBallBalancerKin.__call__ (pysim/quanser_ball_balancer.py:413-480) for N servo angles: 100 steps of
 * SGD(lr 0.01, momentum 0.9) on the two-link loss from tip0 = [r_arm, l_plate/2], then
 * pi/2 - atan2(r + l - tip0, tip1 - d), d = 0.10. */
int psh_qbb_ik(psh_ctx* ctx, int32_t n, const double* th /*[dev][n]*/, const double* r_arm /*[dev][n]*/,
               const double* l_plate /*[dev][n]*/, double* ang /*[dev][n]*/, void* stream);

/* Per-policy mean of returns: ParameterSample.mean_undiscounted_return
 * (Pyrado/pyrado/sampling/parameter_exploration_sampler.py:28-30,69-72) for n_groups consecutive
 * groups of group_size returns each. */
int psh_segment_mean(psh_ctx* ctx, int32_t n_groups, int32_t group_size, const double* ret /*[dev]*/,
                     double* mean /*[dev][n_groups]*/, void* stream);

/* Elite selection of the population-based algorithms: indices (and values) of the k largest of n values in descending
 * order, i.e. to.argsort(rets_avg_ros, descending=True)[:num_is_samples] of CEM.update
 * (Pyrado/pyrado/algorithms/cem.py:116-119), argmax of HC (Pyrado/pyrado/algorithms/hc.py:66-82) with k = 1.  Ties are
 * ordered by index; a NaN ranks above every number (as torch sorts it).  O(n^2) comparisons: sized for the up to a few
 * 10^4 per-policy means of a generation, not for per-rollout arrays.  Applied to the all-gathered per-policy means
 * it keeps the returns on the device between psh_segment_mean and the parameter update. */
int psh_topk(psh_ctx* ctx, int32_t n, int32_t k, const double* values /*[dev][n]*/, int32_t* idx_out /*[dev][k]*/,
             double* val_out /*[dev][k] or NULL*/, void* stream);

/* Bootstrap replications of the mean, the resampling loop of bootstrap_ci
 * (Pyrado/pyrado/sampling/bootstrapping.py:47-61 with stat_fcn = np.mean, as BayRn and SPOTA call it,
 * Pyrado/pyrado/algorithms/bayrn.py:300-304): replication r is the mean of n draws with replacement from data[0..n).
 * Draw j of replication r is data[floor(u n)] with u the (j & 1)-th uniform of Philox4x32-10 at counter
 * (r, 0, j >> 1, stream 6) under key `seed`; the reference draws from NumPy's global generator, so only the
 * distribution of the replications can agree with it. */
int psh_bootstrap_means(psh_ctx* ctx, int32_t n, int32_t num_reps, const double* data /*[dev][n]*/, uint64_t seed,
                        double* stat_bs /*[dev][num_reps]*/, void* stream);

/* Diagnostic, no counterpart in the reference: evaluates one of the device's fp64 elementary functions
 * (the ones the step loop uses in place of NumPy's np.exp / np.tanh / np.sin / np.cos / np.fmod) element-wise, so that
 * their accuracy can be tested directly.  fn: 0 exp, 1 tanh (exponential form), 2 sigmoid, 3 sin, 4 cos,
 * 5 fmod(x[i], m[i]), 6 / 7 tanh from the fine / coarse lookup table (the forms the FNNPolicy kernels use), 8 log (the
 * swing-up reward's and the Box-Muller draws'). */
int psh_eval_math(psh_ctx* ctx, int32_t fn, int32_t n, const double* x /*[dev][n]*/, const double* m /*[dev][n] or NULL*/,
                  double* y /*[dev][n]*/, void* stream);

/* Elapsed time of the most recent psh_rollout kernel of this ctx in milliseconds, measured with HIP
 * events on the launch stream; synchronises on the stop event.  Returns < 0 if none was recorded. */
float psh_last_rollout_ms(psh_ctx* ctx);

/* The same for the rollout launched `back` launches before the most recent one of this ctx (0 = the most recent; the
 * library keeps the event pairs of the last 64 launches), so that a caller can issue a series of rollouts without
 * waiting in between and read every launch's kernel time afterwards -- what the reference's rollout(record_dts=True)
 * collects per step with time.time() (Pyrado/pyrado/sampling/rollout.py:125-138,196-216), here per launch.
 * Launches issued while their stream was being captured into a graph record no events and are not counted. */
float psh_rollout_ms(psh_ctx* ctx, int32_t back);

/* Diagnostic, no counterpart in the reference: the number of device allocations (hipMalloc) the library has made for
 * this ctx's scratch buffers since psh_create.  psh_rollout grows them on demand; psh_rollout_static never does. */
int64_t psh_alloc_count(const psh_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* PYSIMHIP_H */
