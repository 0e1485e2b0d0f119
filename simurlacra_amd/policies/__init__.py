from .base import Policy
from .dummy import DummyPolicy, IdlePolicy
from .features import (FeatureStack, MultFeat, RandFourierFeat, RBFFeat, abs_feat, bell_feat, const_feat, cos_feat, identity_feat, qubic_feat, sig_feat,
                       sign_feat, sin_feat, sincos_feat, sinsin_feat, squared_feat)
from .fnn import FNN, FNNPolicy
from .linear import LinearPolicy
from .initialization import init_param
from .time import TimePolicy
