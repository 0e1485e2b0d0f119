"""
Translation of a task object into the flat TaskSpec that the C ABI carries.  Tasks outside the hot path's set raise.
"""
from .. import _abi as abi
from ..specs import TaskSpec
from .base import TaskWrapper, never_succeeded
from .desired_state import DesStateTask, RadiallySymmDesStateTask
from .final_reward import FinalRewTask
from .reward_functions import UnderActuatedSwingUpRewFcn, QuadrErrRewFcn


def task_to_spec(task) -> TaskSpec:
    final_mode, factor = 0, 1e3
    if isinstance(task, FinalRewTask):
        m = task.mode
        if m.user_input:
            raise NotImplementedError('FinalRewMode(user_input=True) cannot run on the device')
        final_mode = (abi.FIN_ENABLED | (abi.FIN_ALWAYS_POSITIVE if m.always_positive else 0) |
                      (abi.FIN_ALWAYS_NEGATIVE if m.always_negative else 0) |
                      (abi.FIN_STATE_DEPENDENT if m.state_dependent else 0) |
                      (abi.FIN_TIME_DEPENDENT if m.time_dependent else 0))
        factor = task.factor
        task = task.wrapped_task
    if isinstance(task, TaskWrapper) or not isinstance(task, DesStateTask):
        raise NotImplementedError(f'{type(task).__name__} is not supported by the rollout kernels (supported: '
                                  f'DesStateTask, RadiallySymmDesStateTask, optionally inside one FinalRewTask)')
    if task.success_fcn is not never_succeeded:
        raise NotImplementedError('custom success functions are not supported by the rollout kernels')
    rf = task.rew_fcn
    radial = isinstance(task, RadiallySymmDesStateTask)
    kw = dict(radial_idcs=list(task.idcs) if radial else None, mod=list(task.mod) if radial else None,
              final_mode=final_mode, final_factor=factor)
    if isinstance(rf, UnderActuatedSwingUpRewFcn):
        return TaskSpec(task.state_des, 'swingup', swingup=(*rf.c_costs, rf.c_theta_sq, rf.c_theta_log),
                        idx_x=rf.idx_x, idx_th=rf.idx_th, **kw)
    if isinstance(rf, QuadrErrRewFcn):
        return TaskSpec(task.state_des, rf.kind, rf.Q, rf.R, min_rew=getattr(rf, 'min_rew', 1e-4), **kw)
    raise NotImplementedError(f'{type(rf).__name__} is not supported by the rollout kernels')
