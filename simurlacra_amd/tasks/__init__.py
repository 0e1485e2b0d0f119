from .base import Task, TaskWrapper, all_tasks, never_succeeded
from .desired_state import DesStateTask, RadiallySymmDesStateTask
from .final_reward import FinalRewMode, FinalRewTask
from .reward_functions import (ExpQuadrErrRewFcn, QuadrErrRewFcn, RewFcn, ScaledExpQuadrErrRewFcn,
                               UnderActuatedSwingUpRewFcn)
from .to_spec import task_to_spec
