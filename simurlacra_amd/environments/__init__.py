from .base import Env
from .sim_base import SimEnv
from .pysim import (BallOnBeamSim, OneMassOscillatorSim, QBallBalancerKin, QBallBalancerSim, QCartPoleSim,
                    QCartPoleStabSim, QCartPoleSwingUpSim, PendulumSim, QQubeSim, QQubeStabSim, QQubeSwingUpSim, SimPyEnv)
