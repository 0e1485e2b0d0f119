"""
FeatureStack and the element-wise feature functions of the LinearPolicy
(Pyrado/pyrado/policies/features.py:14-140).
"""
from typing import Callable, Sequence

import torch as to


def const_feat(inp: to.Tensor):
    if inp.dim() == 2:
        return to.ones(inp.shape[0], 1).type_as(inp)
    return to.tensor([1.]).type_as(inp)


def identity_feat(inp: to.Tensor):
    return inp.clone()


def sign_feat(inp: to.Tensor):
    return to.sign(inp)


def abs_feat(inp: to.Tensor):
    return to.abs(inp)


def squared_feat(inp: to.Tensor):
    return to.pow(inp, 2)


def qubic_feat(inp: to.Tensor):
    return to.pow(inp, 3)


def sig_feat(inp: to.Tensor, scale: float = 1.):
    return to.sigmoid(scale*inp)


def bell_feat(inp: to.Tensor, scale: float = 1.):
    return to.exp(-scale*to.pow(inp, 2))


def sin_feat(inp: to.Tensor):
    return to.sin(inp)


def cos_feat(inp: to.Tensor):
    return to.cos(inp)


def sinsin_feat(inp: to.Tensor):
    return to.sin(inp)*to.sin(inp)


def sincos_feat(inp: to.Tensor):
    return to.sin(inp)*to.cos(inp)


# names of the feature functions that the rollout kernels evaluate in-register
FUSABLE_FEATS = {const_feat: 'const', identity_feat: 'identity', sign_feat: 'sign', abs_feat: 'abs',
                 squared_feat: 'squared', qubic_feat: 'cubed', sig_feat: 'sig', bell_feat: 'bell', sin_feat: 'sin',
                 cos_feat: 'cos', sinsin_feat: 'sinsin', sincos_feat: 'sincos'}


class FeatureStack:
    def __init__(self, feat_fcns: Sequence[Callable]):
        self.feat_fcns = list(feat_fcns)

    def __call__(self, inp: to.Tensor) -> to.Tensor:
        return to.cat([f(inp) for f in self.feat_fcns], dim=-1)

    def get_num_feat(self, inp_flat_dim: int) -> int:
        return sum(1 if f is const_feat else getattr(f, 'num_feat', inp_flat_dim) for f in self.feat_fcns)

    def fused_names(self):
        """ Kernel names of the features in stack order, or None if one of them is not fusable. """
        names = [FUSABLE_FEATS.get(f) for f in self.feat_fcns]
        return None if any(n is None for n in names) else names
