"""Minimal stand-in for `torchvision.models.inception` (torchvision is neither vendored in /root/reference nor installed in this
image; the reference pins torch 1.13 / torchvision 0.14 in its docker file). TEST INFRASTRUCTURE ONLY.

It restates the PUBLISHED torchvision structure the reference subclasses and instantiates (reference
src/metrics/inception_net.py:1,117-127,135,159,186,218): `BasicConv2d` = Conv2d(bias=False) -> BatchNorm2d(eps=0.001) -> ReLU,
the constructors of InceptionA / B / C / D / E with torchvision's attribute names, the unpatched forwards of B and D, and the
`inception_v3(num_classes, aux_logits, pretrained)` factory. With this module registered as torchvision.models[.inception], the
reference's OWN code -- FIDInceptionA / C / E_1 / E_2.forward, fid_inception_v3(), InceptionV3.__init__ / forward -- runs unmodified on CPU,
which is what tests/test_oracle_cpu.py uses to pin oracle/inception.py block by block and end to end (random weights: the FID
checkpoint itself is not obtainable offline, so FID *values* stay unpinned).
"""
import types

import torch
import torch.nn as nn
import torch.nn.functional as F


class BasicConv2d(nn.Module):
    def __init__(self, in_channels, out_channels, **kwargs):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, bias=False, **kwargs)
        self.bn = nn.BatchNorm2d(out_channels, eps=0.001)

    def forward(self, x):
        return F.relu(self.bn(self.conv(x)), inplace=True)


class InceptionA(nn.Module):
    def __init__(self, in_channels, pool_features, conv_block=None):
        super().__init__()
        cb = conv_block or BasicConv2d
        self.branch1x1 = cb(in_channels, 64, kernel_size=1)
        self.branch5x5_1 = cb(in_channels, 48, kernel_size=1)
        self.branch5x5_2 = cb(48, 64, kernel_size=5, padding=2)
        self.branch3x3dbl_1 = cb(in_channels, 64, kernel_size=1)
        self.branch3x3dbl_2 = cb(64, 96, kernel_size=3, padding=1)
        self.branch3x3dbl_3 = cb(96, 96, kernel_size=3, padding=1)
        self.branch_pool = cb(in_channels, pool_features, kernel_size=1)


class InceptionB(nn.Module):
    def __init__(self, in_channels, conv_block=None):
        super().__init__()
        cb = conv_block or BasicConv2d
        self.branch3x3 = cb(in_channels, 384, kernel_size=3, stride=2)
        self.branch3x3dbl_1 = cb(in_channels, 64, kernel_size=1)
        self.branch3x3dbl_2 = cb(64, 96, kernel_size=3, padding=1)
        self.branch3x3dbl_3 = cb(96, 96, kernel_size=3, stride=2)

    def forward(self, x):
        b3 = self.branch3x3(x)
        bd = self.branch3x3dbl_3(self.branch3x3dbl_2(self.branch3x3dbl_1(x)))
        return torch.cat([b3, bd, F.max_pool2d(x, kernel_size=3, stride=2)], 1)


class InceptionC(nn.Module):
    def __init__(self, in_channels, channels_7x7, conv_block=None):
        super().__init__()
        cb = conv_block or BasicConv2d
        c7 = channels_7x7
        self.branch1x1 = cb(in_channels, 192, kernel_size=1)
        self.branch7x7_1 = cb(in_channels, c7, kernel_size=1)
        self.branch7x7_2 = cb(c7, c7, kernel_size=(1, 7), padding=(0, 3))
        self.branch7x7_3 = cb(c7, 192, kernel_size=(7, 1), padding=(3, 0))
        self.branch7x7dbl_1 = cb(in_channels, c7, kernel_size=1)
        self.branch7x7dbl_2 = cb(c7, c7, kernel_size=(7, 1), padding=(3, 0))
        self.branch7x7dbl_3 = cb(c7, c7, kernel_size=(1, 7), padding=(0, 3))
        self.branch7x7dbl_4 = cb(c7, c7, kernel_size=(7, 1), padding=(3, 0))
        self.branch7x7dbl_5 = cb(c7, 192, kernel_size=(1, 7), padding=(0, 3))
        self.branch_pool = cb(in_channels, 192, kernel_size=1)


class InceptionD(nn.Module):
    def __init__(self, in_channels, conv_block=None):
        super().__init__()
        cb = conv_block or BasicConv2d
        self.branch3x3_1 = cb(in_channels, 192, kernel_size=1)
        self.branch3x3_2 = cb(192, 320, kernel_size=3, stride=2)
        self.branch7x7x3_1 = cb(in_channels, 192, kernel_size=1)
        self.branch7x7x3_2 = cb(192, 192, kernel_size=(1, 7), padding=(0, 3))
        self.branch7x7x3_3 = cb(192, 192, kernel_size=(7, 1), padding=(3, 0))
        self.branch7x7x3_4 = cb(192, 192, kernel_size=3, stride=2)

    def forward(self, x):
        b3 = self.branch3x3_2(self.branch3x3_1(x))
        b7 = self.branch7x7x3_4(self.branch7x7x3_3(self.branch7x7x3_2(self.branch7x7x3_1(x))))
        return torch.cat([b3, b7, F.max_pool2d(x, kernel_size=3, stride=2)], 1)


class InceptionE(nn.Module):
    def __init__(self, in_channels, conv_block=None):
        super().__init__()
        cb = conv_block or BasicConv2d
        self.branch1x1 = cb(in_channels, 320, kernel_size=1)
        self.branch3x3_1 = cb(in_channels, 384, kernel_size=1)
        self.branch3x3_2a = cb(384, 384, kernel_size=(1, 3), padding=(0, 1))
        self.branch3x3_2b = cb(384, 384, kernel_size=(3, 1), padding=(1, 0))
        self.branch3x3dbl_1 = cb(in_channels, 448, kernel_size=1)
        self.branch3x3dbl_2 = cb(448, 384, kernel_size=3, padding=1)
        self.branch3x3dbl_3a = cb(384, 384, kernel_size=(1, 3), padding=(0, 1))
        self.branch3x3dbl_3b = cb(384, 384, kernel_size=(3, 1), padding=(1, 0))
        self.branch_pool = cb(in_channels, 192, kernel_size=1)


class Inception3(nn.Module):
    """Attribute names of torchvision.models.Inception3 with aux_logits=False (the reference only picks sub-modules off it,
    inception_net.py:46-70, and calls load_state_dict on it, :131)."""

    def __init__(self, num_classes=1000, aux_logits=False):
        super().__init__()
        assert not aux_logits
        self.Conv2d_1a_3x3 = BasicConv2d(3, 32, kernel_size=3, stride=2)
        self.Conv2d_2a_3x3 = BasicConv2d(32, 32, kernel_size=3)
        self.Conv2d_2b_3x3 = BasicConv2d(32, 64, kernel_size=3, padding=1)
        self.Conv2d_3b_1x1 = BasicConv2d(64, 80, kernel_size=1)
        self.Conv2d_4a_3x3 = BasicConv2d(80, 192, kernel_size=3)
        self.Mixed_5b = InceptionA(192, pool_features=32)
        self.Mixed_5c = InceptionA(256, pool_features=64)
        self.Mixed_5d = InceptionA(288, pool_features=64)
        self.Mixed_6a = InceptionB(288)
        self.Mixed_6b = InceptionC(768, channels_7x7=128)
        self.Mixed_6c = InceptionC(768, channels_7x7=160)
        self.Mixed_6d = InceptionC(768, channels_7x7=160)
        self.Mixed_6e = InceptionC(768, channels_7x7=192)
        self.Mixed_7a = InceptionD(768)
        self.Mixed_7b = InceptionE(1280)
        self.Mixed_7c = InceptionE(2048)
        self.fc = nn.Linear(2048, num_classes)


def inception_v3(num_classes=1000, aux_logits=False, pretrained=False, **kwargs):
    assert not pretrained
    return Inception3(num_classes=num_classes, aux_logits=aux_logits)


def as_modules():
    """(torchvision, torchvision.models, torchvision.models.inception) module objects carrying the definitions above."""
    inc = types.ModuleType("torchvision.models.inception")
    for k in ("BasicConv2d", "InceptionA", "InceptionB", "InceptionC", "InceptionD", "InceptionE", "Inception3", "inception_v3"):
        setattr(inc, k, globals()[k])
    models = types.ModuleType("torchvision.models")
    models.inception = inc
    models.inception_v3 = inception_v3
    tv = types.ModuleType("torchvision")
    tv.models = models
    return tv, models, inc
