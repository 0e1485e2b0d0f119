"""
Polar2DPosVelSpace (Pyrado/pyrado/spaces/polar.py:43-77): a ring (or ring section) of 2D positions, sampled in polar
coordinates and returned in Cartesian ones, with Cartesian velocities appended.
"""
import numpy as np

from .box import BoxSpace


class Polar2DPosVelSpace(BoxSpace):
    def __init__(self, bound_lo, bound_up, labels=None):
        bound_lo, bound_up = np.asarray(bound_lo, dtype=np.float64), np.asarray(bound_up, dtype=np.float64)
        assert bound_lo.size == bound_up.size == 4
        super().__init__(bound_lo, bound_up, labels=labels)

    def sample_uniform(self, concrete_inf: float = 1e6) -> np.ndarray:
        sample = super().sample_uniform()
        r, phi = sample[0], sample[1]
        sample[:2] = np.array([r*np.cos(phi), r*np.sin(phi)])
        return sample

    def contains(self, cand: np.ndarray, verbose: bool = False) -> bool:
        assert cand.size == 4
        x, y = cand[0], cand[1]
        polar = np.array([np.sqrt(x**2 + y**2), np.arctan2(y, x)])
        return super().contains(np.r_[polar, cand[2:]], verbose=verbose)
