"""Parameter initialisation used by the conv blocks (reference nn/init.py:4-30)."""
from torch import nn

_BN_TYPES = (nn.BatchNorm1d, nn.BatchNorm2d, nn.BatchNorm3d)


def init_uniform(module):
    """Xavier-uniform kernel, zero bias."""
    if getattr(module, "weight", None) is not None:
        nn.init.xavier_uniform_(module.weight)
    if getattr(module, "bias", None) is not None:
        nn.init.zeros_(module.bias)


def init_bn(module):
    """gamma = 1, beta = 0."""
    if module.weight is not None:
        nn.init.ones_(module.weight)
    if module.bias is not None:
        nn.init.zeros_(module.bias)


def set_bn(model, momentum):
    for m in model.modules():
        if isinstance(m, _BN_TYPES):
            m.momentum = momentum


def set_eps(model, eps):
    for m in model.modules():
        if isinstance(m, _BN_TYPES):
            m.eps = eps
