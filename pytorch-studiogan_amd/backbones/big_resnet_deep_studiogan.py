"""StudioGAN's own BigGAN-deep variant with the reference's constructor and forward contracts (reference
src/models/big_resnet_deep_studiogan.py:15-177 Generator, :180-400 Discriminator). Same bottleneck stages, tables, heads and
state_dict layout as `big_resnet_deep_legacy` (whose Generator / Discriminator classes are reused here); the blocks differ:

  GenBlock   skip = conv1x1(nearest-up(x)) (learned, :31-35,58,74-77) instead of the channel slice: one launch with
             upsample-on-load, the main branch rides in as the residual operand
  DiscBlock  (:193-250) the 2x2 average pool sits BEFORE the last ReLU + conv1x1 (:238-240): it is the pooling epilogue of conv2d3;
             skip = conv1x1 over all output channels (:223-229) followed by the pool (fused epilogue), or -- first block, `optblock` --
             pool first, then the conv (:242-245); identity when neither the width nor the resolution changes.
             nn.ReLU(inplace=True) on the block input also rewrites the skip tensor (same storage): x0 = relu(x).
  32x32      the stem is d_conv_dim wide (:259) instead of 4 * d_conv_dim.
"""
import torch.nn as nn

from .. import functional as F
from . import big_resnet_deep_legacy as legacy


class GenBlock(nn.Module):
    def __init__(self, in_channels, out_channels, g_cond_mtd, affine_input_dim, upsample, MODULES, channel_ratio=4):
        super().__init__()
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.g_cond_mtd = g_cond_mtd
        self.upsample = upsample
        self.hidden_channels = self.in_channels // channel_ratio
        self.bn1 = MODULES.g_bn(affine_input_dim, self.in_channels, MODULES)
        self.bn2 = MODULES.g_bn(affine_input_dim, self.hidden_channels, MODULES)
        self.bn3 = MODULES.g_bn(affine_input_dim, self.hidden_channels, MODULES)
        self.bn4 = MODULES.g_bn(affine_input_dim, self.hidden_channels, MODULES)
        self.activation = MODULES.g_act_fn
        self.conv2d0 = MODULES.g_conv2d(in_channels=self.in_channels, out_channels=self.out_channels, kernel_size=1, stride=1, padding=0)
        self.conv2d1 = MODULES.g_conv2d(in_channels=self.in_channels, out_channels=self.hidden_channels, kernel_size=1, stride=1, padding=0)
        self.conv2d2 = MODULES.g_conv2d(in_channels=self.hidden_channels, out_channels=self.hidden_channels, kernel_size=3, stride=1, padding=1)
        self.conv2d3 = MODULES.g_conv2d(in_channels=self.hidden_channels, out_channels=self.hidden_channels, kernel_size=3, stride=1, padding=1)
        self.conv2d4 = MODULES.g_conv2d(in_channels=self.hidden_channels, out_channels=self.out_channels, kernel_size=1, stride=1, padding=0)

    def forward_nhwc(self, x, affine, slot):
        h = self.bn1.forward_nhwc(x, affine, slot, relu=True)
        h = self.conv2d1.forward_nhwc(h, slot)
        h = self.bn2.forward_nhwc(h, affine, slot, relu=True)
        h = self.conv2d2.forward_nhwc(h, slot, in_upsample=self.upsample)
        h = self.bn3.forward_nhwc(h, affine, slot, relu=True)
        h = self.conv2d3.forward_nhwc(h, slot)
        h = self.bn4.forward_nhwc(h, affine, slot, relu=True)
        h = self.conv2d4.forward_nhwc(h, slot)
        return self.conv2d0.forward_nhwc(x, slot, in_upsample=self.upsample, res=h)


class Generator(legacy.Generator):
    BLOCK = GenBlock


class DiscBlock(nn.Module):
    def __init__(self, in_channels, out_channels, MODULES, optblock, downsample=True, channel_ratio=4):
        super().__init__()
        self.optblock = optblock
        self.downsample = downsample
        hidden_channels = out_channels // channel_ratio
        self.ch_mismatch = in_channels != out_channels
        if self.optblock:
            assert self.downsample and self.ch_mismatch, "downsample and ch_mismatch should be True."
        self.activation = MODULES.d_act_fn
        self.conv2d1 = MODULES.d_conv2d(in_channels=in_channels, out_channels=hidden_channels, kernel_size=1, stride=1, padding=0)
        self.conv2d2 = MODULES.d_conv2d(in_channels=hidden_channels, out_channels=hidden_channels, kernel_size=3, stride=1, padding=1)
        self.conv2d3 = MODULES.d_conv2d(in_channels=hidden_channels, out_channels=hidden_channels, kernel_size=3, stride=1, padding=1)
        self.conv2d4 = MODULES.d_conv2d(in_channels=hidden_channels, out_channels=out_channels, kernel_size=1, stride=1, padding=0)
        if self.ch_mismatch or self.downsample:
            self.conv2d0 = MODULES.d_conv2d(in_channels=in_channels, out_channels=out_channels, kernel_size=1, stride=1, padding=0)
        if self.downsample:
            self.average_pooling = nn.AvgPool2d(2)

    def forward_nhwc(self, x, slot):
        h = self.conv2d1.forward_nhwc(x, slot, in_relu=True)
        h = self.conv2d2.forward_nhwc(h, slot, in_relu=True)
        h = self.conv2d3.forward_nhwc(h, slot, in_relu=True, out_pool=self.downsample)
        if self.optblock:
            x0 = self.conv2d0.forward_nhwc(F.AvgPool2Fn.apply(F.ReluFn.apply(x)), slot)
        elif self.downsample or self.ch_mismatch:
            x0 = self.conv2d0.forward_nhwc(x, slot, in_relu=True, out_pool=self.downsample)
        else:
            x0 = F.ReluFn.apply(x)
        return self.conv2d4.forward_nhwc(h, slot, in_relu=True, res=x0)


class Discriminator(legacy.Discriminator):
    IN32_FIRST = 1

    @staticmethod
    def make_block(index, d_index, in_channels, out_channels, MODULES, downsample):
        return DiscBlock(in_channels=in_channels, out_channels=out_channels, MODULES=MODULES, optblock=(index == 0 and d_index == 0),
                         downsample=downsample)
