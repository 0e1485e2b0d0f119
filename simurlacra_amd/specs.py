"""
Host-side descriptions of a task and a policy, and their conversion into the C structs of the ABI.
These carry exactly the information that Pyrado's task / policy objects hold for the hot path:
tasks -> Pyrado/pyrado/tasks/{desired_state,final_reward,reward_functions}.py,
policies -> Pyrado/pyrado/policies/{linear,features,fnn}.py.
"""
from dataclasses import dataclass, field
from typing import Optional, Sequence

import numpy as np

from . import _abi as abi

REW_KINDS = dict(quadr=abi.REW_QUADR, exp=abi.REW_EXP_QUADR, scaledexp=abi.REW_SCALED_EXP_QUADR,
                 swingup=abi.REW_SWINGUP)


@dataclass
class TaskSpec:
    state_des: np.ndarray
    rew_kind: str                       # 'quadr' | 'exp' | 'scaledexp' | 'swingup'
    Q: Optional[np.ndarray] = None      # [S,S]
    R: Optional[np.ndarray] = None      # [A,A]
    radial_idcs: Optional[Sequence[int]] = None  # None -> DesStateTask
    mod: Optional[Sequence[float]] = None        # modulation per radial index (default 2*pi)
    final_mode: int = 0                 # FIN_* bits, 0 = no FinalRewTask
    final_factor: float = 1e3
    min_rew: float = 1e-4
    swingup: Sequence[float] = (1.0, 0.2, 1e-3, 1.0, 0.1)
    idx_x: int = 0
    idx_th: int = 1

    def to_c(self, S: int, A: int) -> abi.PshTask:
        t = abi.PshTask()
        t.rew_kind = REW_KINDS[self.rew_kind]
        t.radial = 0 if self.radial_idcs is None else 1
        mask = 0
        for i in range(abi.PSH_MAX_STATE):
            t.mod[i] = 2*np.pi
        if self.radial_idcs is not None:
            mods = self.mod if self.mod is not None else [2*np.pi]*len(self.radial_idcs)
            for i, m in zip(self.radial_idcs, mods):
                if not 0 <= int(i) < S:
                    raise ValueError(f'radial index {i} outside the state of dimension {S}')
                mask |= 1 << int(i)
                t.mod[int(i)] = float(m)
        t.radial_mask = mask
        t.final_mode = int(self.final_mode)
        t.idx_x, t.idx_th = int(self.idx_x), int(self.idx_th)
        t.final_factor = float(self.final_factor)
        t.min_rew = float(self.min_rew)
        sd = np.asarray(self.state_des, dtype=np.float64).reshape(-1)
        if sd.size != S:
            raise ValueError(f'state_des has {sd.size} entries, the state has {S}')
        for i in range(S):
            t.state_des[i] = sd[i]
        if self.rew_kind != 'swingup':
            Q = np.asarray(self.Q, dtype=np.float64)
            R = np.asarray(self.R, dtype=np.float64)
            if Q.shape != (S, S) or R.shape != (A, A):
                raise ValueError(f'Q must be {(S, S)} and R {(A, A)}, got {Q.shape} and {R.shape}')
            for i, v in enumerate(Q.reshape(-1)):
                t.Q[i] = v
            for i, v in enumerate(R.reshape(-1)):
                t.R[i] = v
        for i, v in enumerate(self.swingup):
            t.swingup[i] = float(v)
        return t


@dataclass
class FeatSpec:
    """ A feature object of the LinearPolicy with data of its own (Pyrado/pyrado/policies/features.py:143-303).
    kind 'mult': idcs.  kind 'rff': scale, freq [D, O], shift [D].  kind 'rbf': centers [K, O], scale (scalar or [O]),
    state_wise. """
    kind: str
    idcs: Optional[Sequence[int]] = None
    scale: Optional[object] = None
    freq: Optional[np.ndarray] = None
    shift: Optional[np.ndarray] = None
    centers: Optional[np.ndarray] = None
    state_wise: bool = True

    def num_feat(self, obs_dim: int) -> int:
        return {'mult': 1, 'rff': self.size(), 'rbf': self.size()*obs_dim}[self.kind]

    def size(self) -> int:
        if self.kind == 'mult':
            return len(self.idcs)
        return int(np.asarray(self.freq if self.kind == 'rff' else self.centers).shape[0])

    def arg(self) -> int:
        if self.kind == 'mult':
            if not 2 <= len(self.idcs) <= 8 or any(not 0 <= int(i) < abi.PSH_MAX_OBS for i in self.idcs):
                raise ValueError(f'a fused MultFeat takes 2..8 indices below {abi.PSH_MAX_OBS}')
            return sum(int(i) << (4*j) for j, i in enumerate(self.idcs))
        return int(bool(self.state_wise)) if self.kind == 'rbf' else 0

    def data(self, obs_dim: int) -> np.ndarray:
        if self.kind == 'mult':
            return np.zeros(0)
        if self.kind == 'rff':
            freq = np.asarray(self.freq, dtype=np.float64)
            shift = np.asarray(self.shift, dtype=np.float64).reshape(-1)
            if freq.ndim != 2 or freq.shape[1] != obs_dim or shift.size != freq.shape[0]:
                raise ValueError(f'RandFourierFeat: freq must be [D, {obs_dim}] and shift [D]')
            return np.concatenate([[float(self.scale)], shift, freq.reshape(-1)])
        centers = np.asarray(self.centers, dtype=np.float64)
        if centers.ndim != 2 or centers.shape[1] != obs_dim:
            raise ValueError(f'RBFFeat: centers must be [K, {obs_dim}]')
        scale = np.broadcast_to(np.asarray(self.scale, dtype=np.float64).reshape(-1), (obs_dim,))
        return np.concatenate([scale, centers.reshape(-1)])


@dataclass
class PolicySpec:
    kind: str                                   # 'linear' | 'fnn'
    obs_dim: int
    act_dim: int
    # linear: the FeatureStack in order.  A name for const / element-wise features, or a FeatSpec for MultFeat,
    # RandFourierFeat, RBFFeat
    feats: Sequence = field(default_factory=list)
    hidden: Sequence[int] = field(default_factory=list)    # fnn: hidden sizes
    hidden_nonlin: Sequence[Optional[str]] = field(default_factory=list)
    out_nonlin: Optional[str] = None
    # action-noise exploration around the policy (Pyrado/pyrado/exploration/stochastic_action.py:96-176)
    expl_kind: Optional[str] = None                        # None | 'normal' | 'uniform'
    expl_mean: Optional[Sequence[float]] = None
    expl_scale: Optional[Sequence[float]] = None           # std / halfspan per action dimension

    @property
    def num_feat(self) -> int:
        return sum(f.num_feat(self.obs_dim) if isinstance(f, FeatSpec) else (1 if f == 'const' else self.obs_dim)
                   for f in self.feats)

    def feat_data(self) -> np.ndarray:
        """ The data of the RFF / RBF features, concatenated in stack order (psh_policy.feat_data). """
        parts = [f.data(self.obs_dim) for f in self.feats if isinstance(f, FeatSpec)]
        return np.concatenate(parts) if parts else np.zeros(0)

    @property
    def num_param(self) -> int:
        if self.kind == 'linear':
            return self.act_dim*self.num_feat
        n, last = 0, self.obs_dim
        for h in self.hidden:
            n += h*last + h
            last = h
        return n + self.act_dim*last + self.act_dim

    def to_c(self, n_policies: int = 1, params_ptr: int = 0, index_ptr: int = 0, policy_group: int = 0,
             feat_data_ptr: int = 0) -> abi.PshPolicy:
        p = abi.PshPolicy()
        if self.kind == 'linear':
            p.kind = abi.POL_LINEAR
            if not 1 <= len(self.feats) <= abi.PSH_MAX_FEATS:
                raise ValueError(f'a fused LinearPolicy takes 1..{abi.PSH_MAX_FEATS} feature functions')
            if self.num_feat > abi.PSH_MAX_LIN_FEAT:
                raise ValueError(f'a fused LinearPolicy takes at most {abi.PSH_MAX_LIN_FEAT} feature values')
            p.n_feats = len(self.feats)
            off = 0
            for i, f in enumerate(self.feats):
                if isinstance(f, FeatSpec):
                    p.feats[i] = abi.FEAT_IDS[f.kind]
                    p.feat_size[i], p.feat_arg[i], p.feat_off[i] = f.size(), f.arg(), off
                    off += f.data(self.obs_dim).size
                else:
                    p.feats[i] = abi.FEAT_IDS[f]
            p.feat_data_len = off
            p.feat_data = feat_data_ptr if off else 0
        elif self.kind == 'fnn':
            p.kind = abi.POL_FNN
            if not 1 <= len(self.hidden) <= abi.PSH_MAX_HIDDEN:
                raise ValueError(f'a fused FNNPolicy takes 1..{abi.PSH_MAX_HIDDEN} hidden layers')
            p.n_hidden = len(self.hidden)
            nls = list(self.hidden_nonlin)
            if len(nls) != len(self.hidden):
                raise ValueError('one nonlinearity per hidden layer is required')
            for i, (h, nl) in enumerate(zip(self.hidden, nls)):
                if not 1 <= h <= abi.PSH_MAX_WIDTH:
                    raise ValueError(f'hidden sizes must be within 1..{abi.PSH_MAX_WIDTH}')
                p.hidden[i] = int(h)
                p.hidden_nonlin[i] = abi.NL_IDS[nl]
            p.out_nonlin = abi.NL_IDS[self.out_nonlin]
        else:
            raise ValueError(self.kind)
        p.expl_kind = {None: abi.EXPL_NONE, 'normal': abi.EXPL_NORMAL, 'uniform': abi.EXPL_UNIFORM}[self.expl_kind]
        if self.expl_kind is not None:
            mean = np.zeros(self.act_dim) if self.expl_mean is None else np.asarray(self.expl_mean, dtype=np.float64).reshape(-1)
            scale = np.asarray(self.expl_scale, dtype=np.float64).reshape(-1)
            if mean.size != self.act_dim or scale.size != self.act_dim:
                raise ValueError('exploration mean / scale need one entry per action dimension')
            for i in range(self.act_dim):
                p.expl_mean[i], p.expl_scale[i] = mean[i], scale[i]
        p.n_policies = int(n_policies)
        p.policy_group = int(policy_group)
        p.num_param = self.num_param
        p.params = params_ptr
        p.policy_index = index_ptr
        return p


WRAP_IDS = dict(act_norm=abi.WRAP_ACT_NORM, act_delay=abi.WRAP_ACT_DELAY, act_noise=abi.WRAP_ACT_NOISE,
                obs_noise=abi.WRAP_OBS_NOISE, obs_norm=abi.WRAP_OBS_NORM, obs_partial=abi.WRAP_OBS_PARTIAL)


@dataclass
class WrapSpec:
    """ The environment wrappers fused into a rollout (Pyrado/pyrado/environment_wrappers/{action_normalization,
    action_delay,action_noise,observation_noise,observation_normalization,observation_partial}.py), OUTERMOST first.
    Vectors are indexed by the simulation's own observation / action index. """
    ops: Sequence[str] = field(default_factory=list)
    act_delay: int = 0
    act_delay_row: int = -1                 # row of the domain-parameter table holding per-env delays
    obs_keep: Optional[Sequence[bool]] = None
    noise_seed: int = 0
    env_offset: int = 0
    act_noise_mean: Optional[Sequence[float]] = None
    act_noise_std: Optional[Sequence[float]] = None
    obs_noise_mean: Optional[Sequence[float]] = None
    obs_noise_std: Optional[Sequence[float]] = None
    obs_lo: Optional[Sequence[float]] = None   # NaN entries: the simulation's own bound
    obs_hi: Optional[Sequence[float]] = None

    def obs_dim(self, O: int) -> int:
        if 'obs_partial' in self.ops:
            return int(np.count_nonzero(np.asarray(self.obs_keep, dtype=bool)))
        return O

    def to_c(self, O: int, A: int) -> abi.PshWrappers:
        w = abi.PshWrappers()
        ops = list(self.ops)
        if len(ops) > abi.PSH_MAX_WRAP or len(set(ops)) != len(ops):
            raise ValueError('each wrapper kind can be fused once')
        w.n_ops = len(ops)
        for i, name in enumerate(ops):
            w.ops[i] = WRAP_IDS[name]
        if not 0 <= int(self.act_delay) <= abi.PSH_MAX_DELAY:
            raise ValueError(f'act_delay must be within 0..{abi.PSH_MAX_DELAY}')
        w.act_delay, w.act_delay_row = int(self.act_delay), int(self.act_delay_row)
        keep = np.ones(O, dtype=bool) if self.obs_keep is None else np.asarray(self.obs_keep, dtype=bool).reshape(-1)
        if keep.size != O:
            raise ValueError(f'obs_keep has {keep.size} entries, the simulation observes {O}')
        w.obs_keep_mask = int(sum(1 << r for r in range(O) if keep[r]))
        w.noise_seed, w.env_offset = int(self.noise_seed) & (2**64 - 1), int(self.env_offset)

        def fill(dst, src, n, default):
            v = np.full(n, default, dtype=np.float64) if src is None else np.asarray(src, dtype=np.float64).reshape(-1)
            if v.size != n:
                raise ValueError(f'expected {n} entries, got {v.size}')
            for i in range(n):
                dst[i] = v[i]
        fill(w.act_noise_mean, self.act_noise_mean, A, 0.)
        fill(w.act_noise_std, self.act_noise_std, A, 0.)
        fill(w.obs_noise_mean, self.obs_noise_mean, O, 0.)
        fill(w.obs_noise_std, self.obs_noise_std, O, 0.)
        fill(w.obs_lo, self.obs_lo, O, np.nan)
        fill(w.obs_hi, self.obs_hi, O, np.nan)
        return w
