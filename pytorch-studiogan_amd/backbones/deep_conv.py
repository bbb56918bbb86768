"""DCGAN backbone with the reference's constructor and forward contracts (reference src/models/deep_conv.py:42-121
Generator, :152-299 Discriminator; the CIFAR10 DCGAN / GGAN / LSGAN / WGAN-WC configs). Widths are
fixed by the reference (512-256-128 -> 256-128-64 in G, 3-64-128 -> 64-128-256 then 512 in D) for 32x32 images.

  GenBlock   [ConvTranspose 4x4 s2 as a transposed gather on the MFMA engine] -> (c)BN + ReLU       (2 launches)
  DiscBlock  conv3x3 -> BN+ReLU -> conv4x4 s2 -> BN+ReLU        (with SN in D: ReLU is fused into the next conv's load)
"""
import torch
import torch.nn as nn
import torch.nn.functional as TF

from .. import functional as F
from .. import ops
from ..bank import get_bank
from .heads import build_heads, apply_heads
from .big_resnet import _dtype, _need_graph


class GenBlock(nn.Module):
    def __init__(self, in_channels, out_channels, g_cond_mtd, g_info_injection, affine_input_dim, MODULES):
        super().__init__()
        self.g_cond_mtd = g_cond_mtd
        self.g_info_injection = g_info_injection
        self.conditional = g_cond_mtd == "cBN" or g_info_injection == "cBN"
        self.deconv0 = MODULES.g_deconv2d(in_channels=in_channels, out_channels=out_channels, kernel_size=4, stride=2, padding=1)
        if g_cond_mtd == "W/O" and g_info_injection in ["N/A", "concat"]:
            self.bn0 = MODULES.g_bn(in_features=out_channels)
        elif self.conditional:
            self.bn0 = MODULES.g_bn(affine_input_dim, out_channels, MODULES)
        else:
            raise NotImplementedError
        self.activation = MODULES.g_act_fn

    def forward_nhwc(self, x, affine, slot):
        x = self.deconv0.forward_nhwc(x, slot)
        if self.conditional:
            return self.bn0.forward_nhwc(x, affine, slot, relu=True)
        return self.bn0.forward_nhwc(x, relu=True)


class Generator(nn.Module):
    def __init__(self, z_dim, g_shared_dim, img_size, g_conv_dim, apply_attn, attn_g_loc, g_cond_mtd, num_classes, g_init, g_depth,
                 mixed_precision, MODULES, MODEL):
        super().__init__()
        self.in_dims = [512, 256, 128]
        self.out_dims = [256, 128, 64]
        self.z_dim = z_dim
        self.num_classes = num_classes
        self.g_cond_mtd = g_cond_mtd
        self.mixed_precision = mixed_precision
        self.MODEL = MODEL
        self.affine_input_dim = 0
        # InfoGAN (reference src/models/deep_conv.py:56-68,101-106; configs/CIFAR10/DCGAN-Info.yaml)
        self.info_type = getattr(MODEL, "info_type", "N/A")
        self.g_info_injection = getattr(MODEL, "g_info_injection", "N/A")
        info_dim = 0
        if self.info_type in ("discrete", "both"):
            info_dim += MODEL.info_num_discrete_c * MODEL.info_dim_discrete_c
        if self.info_type in ("continuous", "both"):
            info_dim += MODEL.info_num_conti_c
        if self.info_type != "N/A":
            if self.g_info_injection == "concat":
                self.info_mix_linear = MODULES.g_linear(in_features=self.z_dim + info_dim, out_features=self.z_dim, bias=True)
            elif self.g_info_injection == "cBN":
                self.affine_input_dim += self.z_dim
                self.info_proj_linear = MODULES.g_linear(in_features=info_dim, out_features=self.z_dim, bias=True)
            else:
                raise NotImplementedError(f"g_info_injection = {self.g_info_injection}")
        if self.g_cond_mtd != "W/O" and self.g_cond_mtd == "cBN":
            self.affine_input_dim += self.num_classes
        self.linear0 = MODULES.g_linear(in_features=self.z_dim, out_features=self.in_dims[0] * 4 * 4, bias=True)
        blocks = []
        for index in range(len(self.in_dims)):
            blocks += [[GenBlock(in_channels=self.in_dims[index], out_channels=self.out_dims[index], g_cond_mtd=self.g_cond_mtd,
                                 g_info_injection=self.g_info_injection, affine_input_dim=self.affine_input_dim, MODULES=MODULES)]]
            if index + 1 in attn_g_loc and apply_attn:
                blocks += [[ops.SelfAttention(self.out_dims[index], is_generator=True, MODULES=MODULES)]]
        self.blocks = nn.ModuleList([nn.ModuleList(block) for block in blocks])
        self.conv4 = MODULES.g_conv2d(in_channels=self.out_dims[-1], out_channels=3, kernel_size=3, stride=1, padding=1)
        self.tanh = nn.Tanh()
        self.conv4._sg_rows_pad = 8
        ops.init_weights(self.modules, g_init)
        ops.adopt(self, _dtype(mixed_precision))

    def forward(self, z, label, shared_label=None, eval=False):
        dtype = self.compute_dtype
        bank = get_bank(self, dtype)
        slot = bank.begin_forward(_need_graph(self, z))
        affine_list = []
        if self.info_type != "N/A":
            if self.g_info_injection == "concat":
                z = self.info_mix_linear.forward_rt(z, slot)
            else:
                z, z_info = z[:, :self.z_dim], z[:, self.z_dim:]
                affine_list.append(self.info_proj_linear.forward_rt(z_info, slot))
        if self.g_cond_mtd != "W/O":
            affine_list.append(TF.one_hot(label, num_classes=self.num_classes).to(torch.float32))
        affines = torch.cat(affine_list, 1) if len(affine_list) > 0 else None
        act = self.linear0.forward_rt(z, slot)
        act = F.NchwToNhwcFn.apply(act.view(-1, self.in_dims[0], 4, 4), dtype)
        for blocklist in self.blocks:
            for block in blocklist:
                if isinstance(block, ops.SelfAttention):
                    act = block.forward_nhwc(act, slot)
                else:
                    act = block.forward_nhwc(act, affines, slot)
        act = self.conv4.forward_nhwc(act, slot)
        return F.NhwcToNchwFn.apply(act, True, 3)


class DiscBlock(nn.Module):
    def __init__(self, in_channels, out_channels, apply_d_sn, MODULES):
        super().__init__()
        self.apply_d_sn = apply_d_sn
        self.conv0 = MODULES.d_conv2d(in_channels=in_channels, out_channels=out_channels, kernel_size=3, stride=1, padding=1)
        self.conv1 = MODULES.d_conv2d(in_channels=out_channels, out_channels=out_channels, kernel_size=4, stride=2, padding=1)
        if not apply_d_sn:
            self.bn0 = MODULES.d_bn(in_features=out_channels)
            self.bn1 = MODULES.d_bn(in_features=out_channels)
        self.activation = MODULES.d_act_fn

    def forward_nhwc(self, x, slot, in_relu=False):
        """returns the block output BEFORE its last activation when spectral norm is on (the consumer fuses the ReLU
        into its load), after it otherwise."""
        if self.apply_d_sn:
            h = self.conv0.forward_nhwc(x, slot, in_relu=in_relu)
            return self.conv1.forward_nhwc(h, slot, in_relu=True)
        h = self.conv0.forward_nhwc(x, slot)
        h = self.bn0.forward_nhwc(h, relu=True)
        h = self.conv1.forward_nhwc(h, slot)
        return self.bn1.forward_nhwc(h, relu=True)


class Discriminator(nn.Module):
    def __init__(self, img_size, d_conv_dim, apply_d_sn, apply_attn, attn_d_loc, d_cond_mtd, aux_cls_type, d_embed_dim, normalize_d_embed,
                 num_classes, d_init, d_depth, mixed_precision, MODULES, MODEL):
        super().__init__()
        self.in_dims = [3] + [64, 128]
        self.out_dims = [64, 128, 256]
        self.apply_d_sn = apply_d_sn
        self.d_cond_mtd = d_cond_mtd
        self.aux_cls_type = aux_cls_type
        self.normalize_d_embed = normalize_d_embed
        self.num_classes = num_classes
        self.mixed_precision = mixed_precision
        self.MODEL = MODEL
        blocks = []
        for index in range(len(self.in_dims)):
            blocks += [[DiscBlock(in_channels=self.in_dims[index], out_channels=self.out_dims[index], apply_d_sn=self.apply_d_sn, MODULES=MODULES)]]
            if index + 1 in attn_d_loc and apply_attn:
                blocks += [[ops.SelfAttention(self.out_dims[index], is_generator=False, MODULES=MODULES)]]
        self.blocks = nn.ModuleList([nn.ModuleList(block) for block in blocks])
        self.activation = MODULES.d_act_fn
        self.conv1 = MODULES.d_conv2d(in_channels=256, out_channels=512, kernel_size=3, stride=1, padding=1)
        if not self.apply_d_sn:
            self.bn1 = MODULES.d_bn(in_features=512)
        build_heads(self, MODULES, 512, d_cond_mtd, aux_cls_type, d_embed_dim, num_classes, MODEL)
        self.blocks[0][0].conv0._sg_cin_pad = 8     # RGB image as an 8-channel NHWC tensor (zero-filled)
        if d_init:
            ops.init_weights(self.modules, d_init)
        ops.adopt(self, _dtype(mixed_precision))

    def forward(self, x, label, eval=False, adc_fake=False):
        dtype = self.compute_dtype
        bank = get_bank(self, dtype)
        slot = bank.begin_forward(_need_graph(self, x))
        h = ops.to_nhwc(x, dtype, 8)
        pending_relu = False       # with SN the block's trailing ReLU rides on the next stride-1 conv's load
        for blocklist in self.blocks:
            for block in blocklist:
                if isinstance(block, ops.SelfAttention):
                    if pending_relu:
                        h, pending_relu = F.ReluFn.apply(h), False
                    h = block.forward_nhwc(h, slot)
                else:
                    h = block.forward_nhwc(h, slot, in_relu=pending_relu)
                    pending_relu = self.apply_d_sn
        if self.apply_d_sn:
            h = self.conv1.forward_nhwc(h, slot, in_relu=pending_relu)
        else:
            h = self.conv1.forward_nhwc(h, slot)
            h = self.bn1.forward_nhwc(h)
        hw = h.shape[1] * h.shape[2]
        h = F.ReluSumFn.apply(h)
        return apply_heads(self, h, label, slot, adc_fake, hw=hw)
