"""
simurlacra_amd — MI355X-native batched rollout engine behind the Pyrado API for the standalone-Python
simulations (SimPyEnv subclasses).  The numerics run in hand-written HIP kernels (libpysimhip.so); this
package is the host-side mirror of the Pyrado interface for that path (Pyrado/pyrado/__init__.py for the
package-level names).
"""
import random as _random

inf = float('inf')
nan = float('nan')

from .utils.exceptions import PathErr, ShapeErr, TypeErr, ValueErr  # noqa: E402


def set_seed(seed: int):
    """ Seed Python's, NumPy's and torch's global generators (Pyrado/pyrado/__init__.py:83-93). """
    import numpy as np
    import torch
    _random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


__all__ = ['inf', 'nan', 'set_seed', 'TypeErr', 'ValueErr', 'ShapeErr', 'PathErr']
