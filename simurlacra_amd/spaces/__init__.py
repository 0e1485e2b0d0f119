from .base import Space
from .box import BoxSpace
from .polar import Polar2DPosVelSpace
from .compound import CompoundSpace
from .singular import SingularStateSpace

__all__ = ['Space', 'BoxSpace', 'Polar2DPosVelSpace', 'CompoundSpace', 'SingularStateSpace']
