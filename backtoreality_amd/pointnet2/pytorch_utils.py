"""Layer helpers with the reference's names, constructor keywords and parameter names
(/root/reference/detection/Votenet/pointnet2/pytorch_utils.py) so that reference checkpoints
load unchanged: `SharedMLP` children are `layer{i}`, each a conv block with children
`conv` / `bn` / `activation`, and `bn` is itself a one-child Sequential named `bn`
(hence `...layer0.bn.bn.weight`, pytorch_utils.py:39-46).
"""
from typing import List, Tuple

import torch.nn as nn


def _default_act():
    return nn.ReLU(inplace=True)


class _BNBase(nn.Sequential):
    """One batch-norm child called `<name>bn`, weight=1 / bias=0 (pytorch_utils.py:39-46)."""

    def __init__(self, in_size, batch_norm=None, name=""):
        super().__init__()
        norm = batch_norm(in_size)
        nn.init.constant_(norm.weight, 1.0)
        nn.init.constant_(norm.bias, 0)
        self.add_module(name + "bn", norm)


class BatchNorm1d(_BNBase):
    def __init__(self, in_size: int, *, name: str = ""):
        super().__init__(in_size, batch_norm=nn.BatchNorm1d, name=name)


class BatchNorm2d(_BNBase):
    def __init__(self, in_size: int, name: str = ""):
        super().__init__(in_size, batch_norm=nn.BatchNorm2d, name=name)


class BatchNorm3d(_BNBase):
    def __init__(self, in_size: int, name: str = ""):
        super().__init__(in_size, batch_norm=nn.BatchNorm3d, name=name)


class _ConvBase(nn.Sequential):
    """conv -> [bn] -> [activation], or with preact=True: [bn] -> [activation] -> conv.
    The conv carries a bias only when there is no batch norm (pytorch_utils.py:87)."""

    def __init__(self, in_size, out_size, kernel_size, stride, padding, activation, bn, init,
                 conv=None, batch_norm=None, bias=True, preact=False, name=""):
        super().__init__()
        use_bias = bool(bias) and not bn
        conv_unit = conv(in_size, out_size, kernel_size=kernel_size, stride=stride,
                         padding=padding, bias=use_bias)
        init(conv_unit.weight)
        if use_bias:
            nn.init.constant_(conv_unit.bias, 0)

        tail = []
        if bn:
            tail.append((name + "bn", batch_norm(in_size if preact else out_size)))
        if activation is not None:
            tail.append((name + "activation", activation))
        head = [(name + "conv", conv_unit)]
        for key, mod in (tail + head if preact else head + tail):
            self.add_module(key, mod)


def _conv_block(conv_cls, bn_cls, ndim):
    ones, zeros = (1,) * ndim, (0,) * ndim
    if ndim == 1:
        ones, zeros = 1, 0

    class _Block(_ConvBase):
        def __init__(self, in_size: int, out_size: int, *, kernel_size=ones, stride=ones,
                     padding=zeros, activation="__default__", bn: bool = False,
                     init=nn.init.kaiming_normal_, bias: bool = True, preact: bool = False,
                     name: str = ""):
            if isinstance(activation, str):
                activation = _default_act()
            super().__init__(in_size, out_size, kernel_size, stride, padding, activation, bn,
                             init, conv=conv_cls, batch_norm=bn_cls, bias=bias, preact=preact,
                             name=name)

    return _Block


Conv1d = _conv_block(nn.Conv1d, BatchNorm1d, 1)
Conv1d.__name__ = Conv1d.__qualname__ = "Conv1d"
Conv2d = _conv_block(nn.Conv2d, BatchNorm2d, 2)
Conv2d.__name__ = Conv2d.__qualname__ = "Conv2d"
Conv3d = _conv_block(nn.Conv3d, BatchNorm3d, 3)
Conv3d.__name__ = Conv3d.__qualname__ = "Conv3d"


class SharedMLP(nn.Sequential):
    """Stack of 1x1 Conv2d blocks `layer0..` over channel sizes `args` (pytorch_utils.py:11-36).
    With `first and preact`, layer 0 gets neither norm nor activation."""

    def __init__(self, args: List[int], *, bn: bool = False, activation="__default__",
                 preact: bool = False, first: bool = False, name: str = ""):
        super().__init__()
        if isinstance(activation, str):
            activation = _default_act()
        for i, (cin, cout) in enumerate(zip(args[:-1], args[1:])):
            plain = first and preact and i == 0
            self.add_module(
                name + "layer{}".format(i),
                Conv2d(cin, cout, bn=bn and not plain,
                       activation=None if plain else activation, preact=preact))


class FC(nn.Sequential):
    """Linear -> [bn] -> [activation] (or pre-activation order) (pytorch_utils.py:229-268)."""

    def __init__(self, in_size: int, out_size: int, *, activation="__default__",
                 bn: bool = False, init=None, preact: bool = False, name: str = ""):
        super().__init__()
        if isinstance(activation, str):
            activation = _default_act()
        fc = nn.Linear(in_size, out_size, bias=not bn)
        if init is not None:
            init(fc.weight)
        if not bn:
            nn.init.constant_(fc.bias, 0)
        tail = []
        if bn:
            tail.append((name + "bn", BatchNorm1d(in_size if preact else out_size)))
        if activation is not None:
            tail.append((name + "activation", activation))
        head = [(name + "fc", fc)]
        for key, mod in (tail + head if preact else head + tail):
            self.add_module(key, mod)


def set_bn_momentum_default(bn_momentum):
    def fn(m):
        if isinstance(m, (nn.BatchNorm1d, nn.BatchNorm2d, nn.BatchNorm3d)):
            m.momentum = bn_momentum

    return fn


class BNMomentumScheduler(object):
    """Sets every BatchNorm's momentum to bn_lambda(epoch) on step() (pytorch_utils.py:271-296)."""

    def __init__(self, model, bn_lambda, last_epoch=-1, setter=set_bn_momentum_default):
        if not isinstance(model, nn.Module):
            raise RuntimeError(
                "Class '{}' is not a PyTorch nn Module".format(type(model).__name__))
        self.model = model
        self.setter = setter
        self.lmbd = bn_lambda
        self.step(last_epoch + 1)
        self.last_epoch = last_epoch

    def step(self, epoch=None):
        if epoch is None:
            epoch = self.last_epoch + 1
        self.last_epoch = epoch
        self.model.apply(self.setter(self.lmbd(epoch)))
