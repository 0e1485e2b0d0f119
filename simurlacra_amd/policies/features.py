"""
FeatureStack and the element-wise feature functions of the LinearPolicy
(Pyrado/pyrado/policies/features.py:14-140).
"""
from typing import Callable, Sequence

import numpy as np
import torch as to

from ..utils.exceptions import ShapeErr, TypeErr, ValueErr


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


class MultFeat:
    """ Product of selected input dimensions (Pyrado/pyrado/policies/features.py:143-166): one feature value. """
    num_feat = 1

    def __init__(self, idcs):
        if not len(idcs) >= 2:
            raise ShapeErr(msg='Provide at least provide two indices.')
        self._idcs = list(idcs)

    def __str__(self):
        return f'MultFeat (indices ' + ' '.join(str(i) for i in self._idcs) + ')'

    def __call__(self, inp: to.Tensor) -> to.Tensor:
        out = inp[..., self._idcs[0]]
        for i in self._idcs[1:]:
            out = out*inp[..., i]
        return out.unsqueeze(-1)

    def fused_spec(self):
        from ..specs import FeatSpec
        ok = 2 <= len(self._idcs) <= 8 and all(0 <= int(i) < 8 for i in self._idcs)
        return FeatSpec('mult', idcs=[int(i) for i in self._idcs]) if ok else None


class RandFourierFeat:
    """ Random Fourier features of a Gaussian kernel, scale*cos(inp @ freq^T + shift)
    (Pyrado/pyrado/policies/features.py:169-218): num_feat_per_dim feature values in total. """

    def __init__(self, inp_dim: int, num_feat_per_dim: int, bandwidth):
        self.num_feat_per_dim = self.num_feat = num_feat_per_dim
        self.scale = to.sqrt(to.tensor(2./num_feat_per_dim, dtype=to.float64))
        self.freq = to.randn(num_feat_per_dim, inp_dim, dtype=to.float64)
        if not isinstance(bandwidth, to.Tensor):
            bandwidth = to.from_numpy(np.asanyarray(bandwidth, dtype=np.float64))
        self.freq *= to.sqrt(to.tensor(2., dtype=to.float64)/to.atleast_2d(bandwidth.to(to.float64)))
        self.shift = 2*np.pi*to.rand(num_feat_per_dim, dtype=to.float64)

    def __call__(self, inp: to.Tensor) -> to.Tensor:
        if inp.ndimension() > 2:
            raise ShapeErr(msg='RBF class can only handle 1-dim or 2-dim input!')
        batched = inp.ndimension() == 2
        out = self.scale*to.cos(to.atleast_2d(inp).to(to.float64)@self.freq.t() + self.shift)
        return out if batched else out.squeeze(0)   # 1-dim in, 1-dim out, so that it stacks with the other features

    def fused_spec(self):
        from ..specs import FeatSpec
        return FeatSpec('rff', scale=float(self.scale), freq=self.freq.detach().cpu().numpy(),
                        shift=self.shift.detach().cpu().numpy())


class RBFFeat:
    """ Normalised Gaussian radial basis functions on a grid of centers between the given bounds
    (Pyrado/pyrado/policies/features.py:221-303): num_feat_per_dim values per input dimension. """

    def __init__(self, num_feat_per_dim: int, bounds, scale: float = None, state_wise_norm: bool = True):
        if not num_feat_per_dim > 1:
            raise ValueErr(given=num_feat_per_dim, g_constraint='1')
        if not len(bounds) == 2:
            raise ShapeErr(given=bounds, expected_match=np.empty(2))
        bnd = []
        for b in bounds:
            if isinstance(b, np.ndarray):
                bnd.append(to.from_numpy(b).to(to.float64))
            elif isinstance(b, to.Tensor):
                bnd.append(b.clone().to(to.float64))
            elif isinstance(b, (int, float)):
                bnd.append(to.tensor(b, dtype=to.float64).view(1,))
            else:
                raise TypeErr(given=b, expected_type=(np.ndarray, to.Tensor, int, float))
        # infinite bounds are clipped to +-1e6 as the reference does
        bound_lo, bound_up = [to.clamp(b, min=-1e6, max=1e6) for b in bnd]
        num_dim = len(bound_lo)
        self.num_feat = num_feat_per_dim*num_dim
        self.centers = to.empty(num_feat_per_dim, num_dim, dtype=to.float64)
        for i in range(num_dim):
            self.centers[:, i] = to.linspace(float(bound_lo[i]), float(bound_up[i]), num_feat_per_dim, dtype=to.float64)
        if scale is None:
            # neighbouring basis functions have the value 0.2 at each other's center
            delta_center = self.centers[1, :] - self.centers[0, :]
            self.scale = -to.log(to.tensor(0.2, dtype=to.float64))/to.pow(delta_center, 2)
        else:
            self.scale = scale
        self._state_wise_norm = state_wise_norm

    def _exp_sq_dist(self, inp: to.Tensor) -> to.Tensor:
        inp = to.atleast_2d(inp).to(to.float64)                     # [batch, dim]
        return to.exp(-self.scale*to.pow(inp.unsqueeze(1) - self.centers, 2))   # [batch, num_feat_per_dim, dim]

    def __call__(self, inp: to.Tensor) -> to.Tensor:
        if inp.ndimension() > 2:
            raise ShapeErr(msg='RBF class can only handle 1-dim or 2-dim input!')
        batched = inp.ndimension() == 2
        out = self._normalized(self._exp_sq_dist(inp))
        return out if batched else out.squeeze(0)

    def fused_spec(self):
        from ..specs import FeatSpec
        scale = self.scale.detach().cpu().numpy() if isinstance(self.scale, to.Tensor) else float(self.scale)
        return FeatSpec('rbf', centers=self.centers.detach().cpu().numpy(), scale=scale, state_wise=self._state_wise_norm)

    def _normalized(self, e: to.Tensor) -> to.Tensor:
        if self._state_wise_norm:
            # every input dimension's activations sum to one; values ordered dimension by dimension
            n = e.sum(dim=1, keepdim=True)
            n = to.where(n > 1e-8, n, to.ones_like(n))
            return (e/n).transpose(1, 2).reshape(e.shape[0], -1)
        flat = e.transpose(1, 2).reshape(e.shape[0], -1)
        n = flat.abs().sum(dim=1, keepdim=True)
        n = to.where(n > 1e-8, n, to.ones_like(n))
        return flat/n


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
        """ The features in stack order as the kernels take them (names of the element-wise functions, FeatSpecs of
        the feature objects), or None if one of them cannot be fused. """
        names = [f.fused_spec() if hasattr(f, 'fused_spec') else FUSABLE_FEATS.get(f) for f in self.feat_fcns]
        return None if any(n is None for n in names) else names
