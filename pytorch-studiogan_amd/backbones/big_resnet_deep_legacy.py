"""BigGAN-deep generator / discriminator with the reference's constructor and forward contracts (reference
src/models/big_resnet_deep_legacy.py:15-197 Generator, :200-413 Discriminator; configs/ImageNet/BigGAN-Deep-*.yaml, SURVEY.md §8
C4). Bottleneck blocks (hidden = channels / 4): 1x1 -> 3x3 -> 3x3 -> 1x1, `depth` blocks per stage.

  GenBlock   cBN+ReLU -> conv1x1 -> cBN+ReLU -> [up x2 +] conv3x3 -> cBN+ReLU -> conv3x3 -> cBN+ReLU ->
             [conv1x1 + residual], residual = nearest-up(x[:, :out]) (channel-slice skip, one small launch)
  DiscBlock  [ReLU + conv1x1] -> [ReLU + conv3x3] -> [ReLU + conv3x3] -> [ReLU + conv1x1 (+ avg-pool) + residual],
             residual = cat([x0, conv1x1(x0)]) with x0 = avgpool(relu(x)): the 1x1 writes into its channel slice.
             (pool(relu(.)) in front of the last 1x1 is evaluated as the 1x1 followed by the fused pooling epilogue: the 1x1 is
             pointwise-linear, so both orders are the same function.)
"""
import torch
import torch.nn as nn

from .. import functional as F
from .. import ops
from ..bank import get_bank
from .heads import build_heads, apply_heads
from .big_resnet import _dtype, _need_graph


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
        self.conv2d1 = MODULES.g_conv2d(in_channels=self.in_channels, out_channels=self.hidden_channels, kernel_size=1, stride=1, padding=0)
        self.conv2d2 = MODULES.g_conv2d(in_channels=self.hidden_channels, out_channels=self.hidden_channels, kernel_size=3, stride=1, padding=1)
        self.conv2d3 = MODULES.g_conv2d(in_channels=self.hidden_channels, out_channels=self.hidden_channels, kernel_size=3, stride=1, padding=1)
        self.conv2d4 = MODULES.g_conv2d(in_channels=self.hidden_channels, out_channels=self.out_channels, kernel_size=1, stride=1, padding=0)

    def forward_nhwc(self, x, affine, slot):
        # x feeds bn1 and the channel-slice skip: bn1's backward adds the skip's gradient in its own launch (functional.GradLink)
        link = F.GradLink() if (F._GRAD_LINK[0] and torch.is_grad_enabled() and x.requires_grad) else None
        h = self.bn1.forward_nhwc(x, affine, slot, relu=True, link=link)
        h = self.conv2d1.forward_nhwc(h, slot)
        h = self.bn2.forward_nhwc(h, affine, slot, relu=True)
        h = self.conv2d2.forward_nhwc(h, slot, in_upsample=self.upsample, stats=self.upsample)      # (bn3's statistics from the quad launch's epilogue)
        h = self.bn3.forward_nhwc(h, affine, slot, relu=True)
        h = self.conv2d3.forward_nhwc(h, slot)
        h = self.bn4.forward_nhwc(h, affine, slot, relu=True)
        if self.upsample or self.in_channels != self.out_channels or link is not None:
            x0 = F.SliceUpFn.apply(x, self.out_channels, 2 if self.upsample else 1, link)
        else:
            x0 = x
        return self.conv2d4.forward_nhwc(h, slot, res=x0)


class Generator(nn.Module):
    BLOCK = GenBlock       # big_resnet_deep_studiogan.py swaps in its own block (learned 1x1 skip)

    def __init__(self, z_dim, g_shared_dim, img_size, g_conv_dim, apply_attn, attn_g_loc, g_cond_mtd, num_classes, g_init, g_depth,
                 mixed_precision, MODULES, MODEL):
        super().__init__()
        g_in_dims_collection = {
            "32": [g_conv_dim * 4, g_conv_dim * 4, g_conv_dim * 4],
            "64": [g_conv_dim * 16, g_conv_dim * 8, g_conv_dim * 4, g_conv_dim * 2],
            "128": [g_conv_dim * 16, g_conv_dim * 16, g_conv_dim * 8, g_conv_dim * 4, g_conv_dim * 2],
            "256": [g_conv_dim * 16, g_conv_dim * 16, g_conv_dim * 8, g_conv_dim * 8, g_conv_dim * 4, g_conv_dim * 2],
            "512": [g_conv_dim * 16, g_conv_dim * 16, g_conv_dim * 8, g_conv_dim * 8, g_conv_dim * 4, g_conv_dim * 2, g_conv_dim]
        }
        g_out_dims_collection = {
            "32": [g_conv_dim * 4, g_conv_dim * 4, g_conv_dim * 4],
            "64": [g_conv_dim * 8, g_conv_dim * 4, g_conv_dim * 2, g_conv_dim],
            "128": [g_conv_dim * 16, g_conv_dim * 8, g_conv_dim * 4, g_conv_dim * 2, g_conv_dim],
            "256": [g_conv_dim * 16, g_conv_dim * 8, g_conv_dim * 8, g_conv_dim * 4, g_conv_dim * 2, g_conv_dim],
            "512": [g_conv_dim * 16, g_conv_dim * 8, g_conv_dim * 8, g_conv_dim * 4, g_conv_dim * 2, g_conv_dim, g_conv_dim]
        }
        self.z_dim = z_dim
        self.g_shared_dim = g_shared_dim
        self.g_cond_mtd = g_cond_mtd
        self.num_classes = num_classes
        self.mixed_precision = mixed_precision
        self.MODEL = MODEL
        self.in_dims = g_in_dims_collection[str(img_size)]
        self.out_dims = g_out_dims_collection[str(img_size)]
        self.bottom = 4
        self.num_blocks = len(self.in_dims)
        self.affine_input_dim = self.z_dim
        # InfoGAN (reference src/models/big_resnet_deep_legacy.py:110-121,156-161)
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
                self.affine_input_dim += self.g_shared_dim
                self.info_proj_linear = MODULES.g_linear(in_features=info_dim, out_features=self.g_shared_dim, bias=True)
            else:
                raise NotImplementedError(f"g_info_injection = {self.g_info_injection}")
        if self.g_cond_mtd != "W/O":
            self.affine_input_dim += self.g_shared_dim
            self.shared = ops.embedding(num_embeddings=self.num_classes, embedding_dim=self.g_shared_dim)
        self.linear0 = MODULES.g_linear(in_features=self.affine_input_dim, out_features=self.in_dims[0] * self.bottom * self.bottom, bias=True)
        blocks = []
        for index in range(self.num_blocks):
            blocks += [[self.BLOCK(in_channels=self.in_dims[index], out_channels=self.in_dims[index] if g_index == 0 else self.out_dims[index],
                                   g_cond_mtd=g_cond_mtd, affine_input_dim=self.affine_input_dim, upsample=True if g_index == (g_depth - 1) else False,
                                   MODULES=MODULES)] for g_index in range(g_depth)]
            if index + 1 in attn_g_loc and apply_attn:
                blocks += [[ops.SelfAttention(self.out_dims[index], is_generator=True, MODULES=MODULES)]]
        self.blocks = nn.ModuleList([nn.ModuleList(block) for block in blocks])
        self.bn4 = ops.batchnorm_2d(in_features=self.out_dims[-1])
        self.activation = MODULES.g_act_fn
        self.conv2d5 = MODULES.g_conv2d(in_channels=self.out_dims[-1], out_channels=3, kernel_size=3, stride=1, padding=1)
        self.tanh = nn.Tanh()
        self.conv2d5._sg_rows_pad = 8
        ops.init_weights(self.modules, g_init)
        ops.adopt(self, _dtype(mixed_precision))

    def forward(self, z, label, shared_label=None, eval=False):
        dtype = self.compute_dtype
        bank = get_bank(self, dtype)
        slot = bank.begin_forward(_need_graph(self, z, shared_label))
        affine_list = []
        if self.info_type != "N/A":
            if self.g_info_injection == "concat":
                z = self.info_mix_linear.forward_rt(z, slot)
            else:
                z, z_info = z[:, :self.z_dim], z[:, self.z_dim:]
                affine_list.append(self.info_proj_linear.forward_rt(z_info, slot))
        if self.g_cond_mtd != "W/O":
            if shared_label is None:
                shared_label = self.shared(label)
            affine_list.append(shared_label)
        if len(affine_list) > 0:
            z = torch.cat(affine_list + [z], 1)
        affine = z
        # every conditional batch norm's [1 + gain(y) | bias(y)] rows in ONE launch (functional.cbn_prefetch): BigGAN-deep conditions all of them -- four per block --
        # on the same vector
        pairs = []
        for blocklist in self.blocks:
            for block in blocklist:
                if not isinstance(block, ops.SelfAttention):
                    for bn in (block.bn1, block.bn2, block.bn3, block.bn4):
                        if isinstance(bn, ops.ConditionalBatchNorm2d):
                            pairs.append((bn, affine))
        F.cbn_prefetch(slot, pairs)
        act = self.linear0.forward_rt(z, slot)
        act = F.NchwToNhwcFn.apply(act.view(-1, self.in_dims[0], self.bottom, self.bottom), dtype)
        act = ops.block_boundary(self, -1, act)
        nxt = bank.boundaries(self.blocks) if bank.exchange is not None else None
        with ops.bump_batches_tracked(self):      # (every batch norm of the network runs once below: their counters move in one launch)
            for bi, blocklist in enumerate(self.blocks):
                for block in blocklist:
                    if isinstance(block, ops.SelfAttention):
                        act = block.forward_nhwc(act, slot)
                    else:
                        act = block.forward_nhwc(act, affine, slot)
                if nxt is not None:
                    act = bank.mark(act, nxt[bi])      # data parallelism: the backward's return to this point releases the gradients behind it
                act = ops.block_boundary(self, bi, act)
            act = self.bn4.forward_nhwc(act, relu=True)
        act = self.conv2d5.forward_nhwc(act, slot)
        return F.NhwcToNchwFn.apply(act, True, 3)


class DiscBlock(nn.Module):
    def __init__(self, in_channels, out_channels, MODULES, downsample=True, channel_ratio=4):
        super().__init__()
        self.downsample = downsample
        hidden_channels = out_channels // channel_ratio
        self.activation = MODULES.d_act_fn
        self.conv2d1 = MODULES.d_conv2d(in_channels=in_channels, out_channels=hidden_channels, kernel_size=1, stride=1, padding=0)
        self.conv2d2 = MODULES.d_conv2d(in_channels=hidden_channels, out_channels=hidden_channels, kernel_size=3, stride=1, padding=1)
        self.conv2d3 = MODULES.d_conv2d(in_channels=hidden_channels, out_channels=hidden_channels, kernel_size=3, stride=1, padding=1)
        self.conv2d4 = MODULES.d_conv2d(in_channels=hidden_channels, out_channels=out_channels, kernel_size=1, stride=1, padding=0)
        self.learnable_sc = True if (in_channels != out_channels) else False
        if self.learnable_sc:
            self.conv2d0 = MODULES.d_conv2d(in_channels=in_channels, out_channels=out_channels - in_channels, kernel_size=1, stride=1, padding=0)
        if self.downsample:
            self.average_pooling = nn.AvgPool2d(2)

    def forward_nhwc(self, x, slot):
        # nn.ReLU(inplace=True) on the block input also rewrites the skip tensor (same storage): x0 = relu(x)
        # x feeds conv2d1 and the skip: conv2d1's data-gradient launch takes the skip's (masked) gradient as its residual (functional.GradLink)
        link = F.GradLink() if (F._GRAD_LINK[0] and torch.is_grad_enabled() and x.requires_grad) else None
        h = self.conv2d1.forward_nhwc(x, slot, in_relu=True, link=link)
        h = self.conv2d2.forward_nhwc(h, slot, in_relu=True)
        h = self.conv2d3.forward_nhwc(h, slot, in_relu=True)
        x0 = F.ReluFn.apply(x, link)
        if self.downsample:
            x0 = F.AvgPool2Fn.apply(x0)
        if self.learnable_sc:
            rt = self.conv2d0._sg_rt
            x0 = F.CatConvFn.apply(x0, self.conv2d0.master_weight, self.conv2d0.bias, rt, slot)
        return self.conv2d4.forward_nhwc(h, slot, in_relu=True, out_pool=self.downsample, res=x0)


class Discriminator(nn.Module):
    IN32_FIRST = 4         # width multiple of the 32x32 stem (big_resnet_deep_studiogan.py: 1)

    @staticmethod
    def make_block(index, d_index, in_channels, out_channels, MODULES, downsample):
        return DiscBlock(in_channels=in_channels, out_channels=out_channels, MODULES=MODULES, downsample=downsample)

    def __init__(self, img_size, d_conv_dim, apply_d_sn, apply_attn, attn_d_loc, d_cond_mtd, aux_cls_type, d_embed_dim, normalize_d_embed,
                 num_classes, d_init, d_depth, mixed_precision, MODULES, MODEL):
        super().__init__()
        d_in_dims_collection = {
            "32": [d_conv_dim * self.IN32_FIRST, d_conv_dim * 4, d_conv_dim * 4],
            "64": [d_conv_dim, d_conv_dim * 2, d_conv_dim * 4, d_conv_dim * 8],
            "128": [d_conv_dim, d_conv_dim * 2, d_conv_dim * 4, d_conv_dim * 8, d_conv_dim * 16],
            "256": [d_conv_dim, d_conv_dim * 2, d_conv_dim * 4, d_conv_dim * 8, d_conv_dim * 8, d_conv_dim * 16],
            "512": [d_conv_dim, d_conv_dim, d_conv_dim * 2, d_conv_dim * 4, d_conv_dim * 8, d_conv_dim * 8, d_conv_dim * 16]
        }
        d_out_dims_collection = {
            "32": [d_conv_dim * 4, d_conv_dim * 4, d_conv_dim * 4],
            "64": [d_conv_dim * 2, d_conv_dim * 4, d_conv_dim * 8, d_conv_dim * 16],
            "128": [d_conv_dim * 2, d_conv_dim * 4, d_conv_dim * 8, d_conv_dim * 16, d_conv_dim * 16],
            "256": [d_conv_dim * 2, d_conv_dim * 4, d_conv_dim * 8, d_conv_dim * 8, d_conv_dim * 16, d_conv_dim * 16],
            "512": [d_conv_dim, d_conv_dim * 2, d_conv_dim * 4, d_conv_dim * 8, d_conv_dim * 8, d_conv_dim * 16, d_conv_dim * 16]
        }
        d_down = {
            "32": [True, True, False, False],
            "64": [True, True, True, True, False],
            "128": [True, True, True, True, True, False],
            "256": [True, True, True, True, True, True, False],
            "512": [True, True, True, True, True, True, True, False]
        }
        self.d_cond_mtd = d_cond_mtd
        self.aux_cls_type = aux_cls_type
        self.normalize_d_embed = normalize_d_embed
        self.num_classes = num_classes
        self.mixed_precision = mixed_precision
        self.in_dims = d_in_dims_collection[str(img_size)]
        self.out_dims = d_out_dims_collection[str(img_size)]
        self.MODEL = MODEL
        down = d_down[str(img_size)]
        if not apply_d_sn:
            pass    # the reference's deep blocks have no batch norm either way (big_resnet_deep_legacy.py:200-240)
        self.input_conv = MODULES.d_conv2d(in_channels=3, out_channels=self.in_dims[0], kernel_size=3, stride=1, padding=1)
        self.input_conv._sg_cin_pad = 8      # RGB image as an 8-channel NHWC tensor (zero-filled)
        blocks = []
        for index in range(len(self.in_dims)):
            blocks += [[self.make_block(index, d_index, self.in_dims[index] if d_index == 0 else self.out_dims[index], self.out_dims[index],
                                        MODULES, True if down[index] and d_index == 0 else False)] for d_index in range(d_depth)]
            if (index + 1) in attn_d_loc and apply_attn:
                blocks += [[ops.SelfAttention(self.out_dims[index], is_generator=False, MODULES=MODULES)]]
        self.blocks = nn.ModuleList([nn.ModuleList(block) for block in blocks])
        self.activation = MODULES.d_act_fn
        build_heads(self, MODULES, self.out_dims[-1], d_cond_mtd, aux_cls_type, d_embed_dim, num_classes, MODEL)
        if d_init:
            ops.init_weights(self.modules, d_init)
        ops.adopt(self, _dtype(mixed_precision))

    def forward(self, x, label, eval=False, adc_fake=False):
        dtype = self.compute_dtype
        bank = get_bank(self, dtype)
        slot = bank.begin_forward(_need_graph(self, x))
        h = ops.to_nhwc(x, dtype, 8)
        h = self.input_conv.forward_nhwc(h, slot)
        h = ops.block_boundary(self, -1, h)
        nxt = bank.boundaries(self.blocks) if bank.exchange is not None else None
        for bi, blocklist in enumerate(self.blocks):
            for block in blocklist:
                h = block.forward_nhwc(h, slot)
            if nxt is not None:
                h = bank.mark(h, nxt[bi])          # data parallelism: the backward's return to this point releases the gradients behind it
            h = ops.block_boundary(self, bi, h)
        hw = h.shape[1] * h.shape[2]
        h = F.ReluSumFn.apply(h)
        return apply_heads(self, h, label, slot, adc_fake, hw=hw)
