"""
EnvSpec and RenderMode (Pyrado/pyrado/utils/data_types.py:11-21,24-37), plus small helpers.
"""
from typing import NamedTuple

from ..spaces.base import Space


class EnvSpec(NamedTuple):
    """ The specification of an environment's input and output space """
    obs_space: Space
    act_space: Space
    state_space: Space = None


class RenderMode(NamedTuple):
    """ What to render: console text, an animation, or both (nothing by default) """
    text: bool = False
    video: bool = False
    render: bool = False
