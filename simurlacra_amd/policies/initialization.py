"""
Parameter initialisation of the policies (Pyrado/pyrado/policies/initialization.py:11-40 for nn.Linear):
orthogonal weights, standard-normal biases (or uniform +-1/sqrt(n) with uniform_bias=True).
"""
from math import sqrt

import torch.nn as nn
import torch.nn.init as init


def init_param(m, **kwargs):
    if isinstance(m, nn.Linear):
        if m.weight.data.ndimension() >= 2:
            init.orthogonal_(m.weight.data)
        else:
            init.normal_(m.weight.data)
        if m.bias is not None:
            if kwargs.get('uniform_bias', False):
                n = m.bias.data.nelement()
                init.uniform_(m.bias.data, a=-1./sqrt(n), b=1./sqrt(n))
            else:
                init.normal_(m.bias.data)
