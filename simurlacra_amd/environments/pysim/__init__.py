from .base import SimPyEnv
from .ball_on_beam import BallOnBeamSim
from .one_mass_oscillator import OneMassOscillatorSim
from .quanser_ball_balancer import QBallBalancerKin, QBallBalancerSim
from .quanser_cartpole import QCartPoleSim, QCartPoleStabSim, QCartPoleSwingUpSim
from .pendulum import PendulumSim
from .quanser_qube import QQubeSim, QQubeStabSim, QQubeSwingUpSim
