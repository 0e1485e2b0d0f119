"""
ctypes mirror of include/pysimhip.h: the structs and constants of the C ABI of libpysimhip.so.
Field order and types must match the header exactly.
"""
import ctypes as C

PSH_VERSION = 111
PSH_MAX_STATE = 8
PSH_MAX_OBS = 8
PSH_MAX_ACT = 2
PSH_MAX_DP = 20
PSH_MAX_HIDDEN = 4
PSH_MAX_WIDTH = 64
PSH_MAX_FEATS = 8
PSH_MAX_LIN_FEAT = 512
PSH_MAX_DISTR = 32
PSH_MAX_DELAY = 32
PSH_MAX_WRAP = 6
PSH_STATIC_MAX_WEIGHT_DOUBLES = 16384
PSH_STATIC_MAX_QBB_ENVS = 524288

PSH_OK, PSH_ERR_ARG, PSH_ERR_HIP, PSH_ERR_NODEV, PSH_ERR_UNSUPPORTED = 0, -1, -2, -3, -4

ENV_QQ, ENV_QCP_SU, ENV_QCP_ST, ENV_QBB, ENV_BOB, ENV_OMO, ENV_PEND, ENV_QQ_ST = range(8)
REW_QUADR, REW_EXP_QUADR, REW_SCALED_EXP_QUADR, REW_SWINGUP = range(4)
FIN_ENABLED, FIN_ALWAYS_POSITIVE, FIN_ALWAYS_NEGATIVE, FIN_STATE_DEPENDENT, FIN_TIME_DEPENDENT = 1, 2, 4, 8, 16
POL_LINEAR, POL_FNN = 0, 1
EXPL_NONE, EXPL_NORMAL, EXPL_UNIFORM = 0, 1, 2
FEAT_IDS = dict(const=0, identity=1, sign=2, abs=3, squared=4, cubed=5, sig=6, bell=7, sin=8, cos=9, sinsin=10,
                sincos=11, mult=12, rff=13, rbf=14)
NL_IDS = {None: 0, 'none': 0, 'tanh': 1, 'relu': 2, 'sigmoid': 3}
F_STOP_ON_DONE, F_SIMPLIFIED_DYN, F_MAX_STEPS_INF, F_TRAJ_ENV_MAJOR = 1, 2, 4, 8
ST_NAN, ST_FAILED, ST_TIMEOUT, ST_BERNOULLI, ST_BAD_INDEX, ST_INTERNAL = 1, 2, 4, 8, 16, 32
DISTR_CONST, DISTR_NORMAL, DISTR_UNIFORM, DISTR_BERNOULLI = range(4)
WRAP_ACT_NORM, WRAP_ACT_DELAY, WRAP_ACT_NOISE, WRAP_OBS_NOISE, WRAP_OBS_NORM, WRAP_OBS_PARTIAL = range(1, 7)

# (S, O, A, P, init_dim) per env kind
ENV_DIMS = {ENV_QQ: (4, 6, 1, 10, 4), ENV_QCP_SU: (4, 5, 1, 14, 4), ENV_QCP_ST: (4, 5, 1, 14, 4),
            ENV_QBB: (8, 8, 2, 20, 4), ENV_BOB: (4, 4, 1, 8, 4), ENV_OMO: (2, 2, 1, 3, 2),
            ENV_PEND: (2, 3, 1, 5, 2), ENV_QQ_ST: (4, 4, 1, 10, 4)}

c_double_p = C.POINTER(C.c_double)
c_int32_p = C.POINTER(C.c_int32)


class PshTask(C.Structure):
    _fields_ = [
        ('rew_kind', C.c_int32), ('radial', C.c_int32), ('radial_mask', C.c_int32), ('final_mode', C.c_int32),
        ('idx_x', C.c_int32), ('idx_th', C.c_int32), ('pad_', C.c_int32*2),
        ('final_factor', C.c_double), ('min_rew', C.c_double),
        ('mod', C.c_double*PSH_MAX_STATE), ('state_des', C.c_double*PSH_MAX_STATE),
        ('Q', C.c_double*(PSH_MAX_STATE*PSH_MAX_STATE)), ('R', C.c_double*(PSH_MAX_ACT*PSH_MAX_ACT)),
        ('swingup', C.c_double*5),
    ]


class PshPolicy(C.Structure):
    _fields_ = [
        ('kind', C.c_int32), ('n_feats', C.c_int32), ('feats', C.c_int32*PSH_MAX_FEATS),
        ('n_hidden', C.c_int32), ('hidden', C.c_int32*PSH_MAX_HIDDEN), ('hidden_nonlin', C.c_int32*PSH_MAX_HIDDEN),
        ('out_nonlin', C.c_int32), ('n_policies', C.c_int32), ('num_param', C.c_int32), ('pad_', C.c_int32),
        ('params', C.c_void_p), ('policy_index', C.c_void_p),
        ('expl_kind', C.c_int32), ('policy_group', C.c_int32), ('expl_mean', C.c_double*PSH_MAX_ACT), ('expl_scale', C.c_double*PSH_MAX_ACT),
        ('feat_size', C.c_int32*PSH_MAX_FEATS), ('feat_arg', C.c_int32*PSH_MAX_FEATS), ('feat_off', C.c_int32*PSH_MAX_FEATS),
        ('feat_data_len', C.c_int32), ('pad2_', C.c_int32), ('feat_data', C.c_void_p),
    ]


class PshWrappers(C.Structure):
    _fields_ = [
        ('n_ops', C.c_int32), ('ops', C.c_int32*PSH_MAX_WRAP), ('act_delay', C.c_int32), ('act_delay_row', C.c_int32),
        ('obs_keep_mask', C.c_int32), ('pad_', C.c_int32*2),
        ('noise_seed', C.c_uint64), ('env_offset', C.c_uint64),
        ('act_noise_mean', C.c_double*PSH_MAX_ACT), ('act_noise_std', C.c_double*PSH_MAX_ACT),
        ('obs_noise_mean', C.c_double*PSH_MAX_OBS), ('obs_noise_std', C.c_double*PSH_MAX_OBS),
        ('obs_lo', C.c_double*PSH_MAX_OBS), ('obs_hi', C.c_double*PSH_MAX_OBS),
    ]


class PshRolloutDesc(C.Structure):
    _fields_ = [
        ('env_kind', C.c_int32), ('n_envs', C.c_int32), ('max_steps', C.c_int32), ('flags', C.c_int32),
        ('dt', C.c_double),
        ('init_state', C.c_void_p), ('init_dim', C.c_int32), ('ld_init', C.c_int32),
        ('domain_param', C.c_void_p), ('dp_index', C.c_void_p), ('ld_dp', C.c_int32), ('dp_rows', C.c_int32),
        ('task', PshTask), ('policy', PshPolicy), ('wrap', PshWrappers), ('bernoulli_reset', C.c_double),
        ('ret', C.c_void_p), ('length', C.c_void_p), ('status', C.c_void_p), ('final_state', C.c_void_p),
        ('obs', C.c_void_p), ('act', C.c_void_p), ('rew', C.c_void_p),
        ('stream', C.c_void_p),
    ]


class PshStepDesc(C.Structure):
    _fields_ = [
        ('env_kind', C.c_int32), ('n_envs', C.c_int32), ('max_steps', C.c_int32), ('flags', C.c_int32),
        ('dt', C.c_double),
        ('domain_param', C.c_void_p), ('dp_index', C.c_void_p), ('ld_dp', C.c_int32), ('init_dim', C.c_int32),
        ('init_state', C.c_void_p),
        ('task', PshTask),
        ('state', C.c_void_p), ('curr_step', C.c_void_p), ('flags_io', C.c_void_p), ('act', C.c_void_p),
        ('obs', C.c_void_p), ('rew', C.c_void_p), ('done', C.c_void_p), ('status', C.c_void_p),
        ('stream', C.c_void_p),
    ]


class PshDistr(C.Structure):
    _fields_ = [
        ('kind', C.c_int32), ('row', C.c_int32), ('roundint', C.c_int32), ('pad_', C.c_int32),
        ('a', C.c_double), ('b', C.c_double), ('c', C.c_double), ('clip_lo', C.c_double), ('clip_up', C.c_double),
    ]


class PshRandDesc(C.Structure):
    _fields_ = [
        ('n_envs', C.c_int32), ('n_rows', C.c_int32), ('n_distr', C.c_int32), ('ld_out', C.c_int32),
        ('seed', C.c_uint64), ('env_offset', C.c_uint64),
        ('distr', PshDistr*PSH_MAX_DISTR), ('nominal', C.c_double*(PSH_MAX_DP + 4)),
        ('out', C.c_void_p), ('stream', C.c_void_p),
    ]


class PshInitDesc(C.Structure):
    _fields_ = [
        ('env_kind', C.c_int32), ('n_envs', C.c_int32), ('ld_dp', C.c_int32), ('ld_out', C.c_int32),
        ('seed', C.c_uint64), ('env_offset', C.c_uint64),
        ('domain_param', C.c_void_p), ('dp_index', C.c_void_p), ('out', C.c_void_p), ('stream', C.c_void_p),
    ]


# every symbol include/pysimhip.h declares: name -> (restype, argtypes)
PSH_SYMBOLS = {
    'psh_version': (C.c_int, []),
    'psh_create': (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    'psh_destroy': (None, [C.c_void_p]),
    'psh_last_error': (C.c_char_p, [C.c_void_p]),
    'psh_env_dims': (C.c_int, [C.c_int, C.POINTER(C.c_int32*5)]),
    'psh_rollout': (C.c_int, [C.c_void_p, C.POINTER(PshRolloutDesc)]),
    'psh_rollout_static': (C.c_int, [C.c_void_p, C.POINTER(PshRolloutDesc)]),
    'psh_reset': (C.c_int, [C.c_void_p, C.POINTER(PshStepDesc)]),
    'psh_step': (C.c_int, [C.c_void_p, C.POINTER(PshStepDesc)]),
    'psh_randomize': (C.c_int, [C.c_void_p, C.POINTER(PshRandDesc)]),
    'psh_sample_init': (C.c_int, [C.c_void_p, C.POINTER(PshInitDesc)]),
    'psh_qbb_ik': (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'psh_segment_mean': (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    'psh_topk': (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'psh_bootstrap_means': (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]),
    'psh_last_rollout_ms': (C.c_float, [C.c_void_p]),
    'psh_rollout_ms': (C.c_float, [C.c_void_p, C.c_int32]),
    'psh_alloc_count': (C.c_int64, [C.c_void_p]),
    'psh_eval_math': (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
}
