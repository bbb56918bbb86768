"""BigGAN generator / discriminator with the reference's constructor and forward contracts
(reference src/models/big_resnet.py:45-158, 245-428; imported by name through src/models/model.py:22), built from
studiogan_amd.ops modules. Module / parameter / buffer names are identical to the reference's, so state_dicts
interchange. Each block is a short chain of fused launches:

  GenBlock   cBN+ReLU -> [up x2 + conv3x3] -> cBN+ReLU -> [conv3x3 + (up x2 + conv1x1 skip) fused]              (4 launches + 4 tiny GEMMs)
  DiscBlock  [ReLU + conv3x3] -> [ReLU + conv3x3 + (ReLU + conv1x1 skip) + avgpool, fused]                 (2 launches)
  (the 1x1 skip convolution rides in the block's last 3x3 launch as extra K-slices: functional.ConvSkipFn / csrc/conv_v4.h SKIP;
   shapes the fused kernel does not take run as two chained launches)

compute dtype: bf16 activations / weight images with fp32 accumulation, statistics, master weights and gradients when
`mixed_precision=True` (the reference's fp16 autocast + GradScaler, big_resnet.py:124,350, becomes scaler-free bf16),
fp32 otherwise.
"""
import torch
import torch.nn as nn

from .. import functional as F
from .. import ops
from ..bank import get_bank
from .heads import build_heads, apply_heads


def _dtype(mixed_precision):
    return torch.bfloat16 if mixed_precision else torch.float32


def _need_graph(module, *inputs):
    if not torch.is_grad_enabled():
        return False
    for t in inputs:
        if torch.is_tensor(t) and t.requires_grad:
            return True
    # any trainable parameter (freezeD, reference src/utils/misc.py:199-216, freezes only the FIRST blocks)
    return any(p.requires_grad for p in module.parameters())


class GenBlock(nn.Module):
    def __init__(self, in_channels, out_channels, g_cond_mtd, affine_input_dim, MODULES):
        super().__init__()
        self.g_cond_mtd = g_cond_mtd
        self.bn1 = MODULES.g_bn(affine_input_dim, in_channels, MODULES)
        self.bn2 = MODULES.g_bn(affine_input_dim, out_channels, MODULES)
        self.activation = MODULES.g_act_fn
        self.conv2d0 = MODULES.g_conv2d(in_channels=in_channels, out_channels=out_channels, kernel_size=1, stride=1, padding=0)
        self.conv2d1 = MODULES.g_conv2d(in_channels=in_channels, out_channels=out_channels, kernel_size=3, stride=1, padding=1)
        self.conv2d2 = MODULES.g_conv2d(in_channels=out_channels, out_channels=out_channels, kernel_size=3, stride=1, padding=1)

    def forward_nhwc(self, x, affine, slot):
        link = F.GradLink()      # x feeds bn1 and the skip: bn1's backward adds the skip's gradient in its own launch (no autograd add)
        h = self.bn1.forward_nhwc(x, affine, slot, relu=True, link=link)
        # (stats=True: a batch norm reads the result next -- bn2 / the next block's bn1 / bn4: its statistics come out of the convolution's epilogue)
        h = self.conv2d1.forward_nhwc(h, slot, in_upsample=True, stats=True)
        h = self.bn2.forward_nhwc(h, affine, slot, relu=True)
        # conv2d2(h) + conv2d0(up(x)): one launch, the skip as extra K-slices (functional.ConvSkipFn)
        return ops.conv_skip_nhwc(self.conv2d2, self.conv2d0, h, x, slot, skip_upsample=True, link=link, stats=True)


class Generator(nn.Module):
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
        self.chunk_size = z_dim // (self.num_blocks + 1)
        self.affine_input_dim = self.chunk_size
        assert self.z_dim % (self.num_blocks + 1) == 0, "z_dim should be divided by the number of blocks"
        # InfoGAN: the codes ride behind z (reference src/utils/sample.py:113-118) and enter either through a mixing linear layer in front of the z chunks
        # ("concat") or as one more conditioning vector of every conditional batch norm ("cBN") -- src/models/big_resnet.py:81-92,125-130
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

        self.linear0 = MODULES.g_linear(in_features=self.chunk_size, out_features=self.in_dims[0] * self.bottom * self.bottom, bias=True)
        if self.g_cond_mtd != "W/O":
            self.affine_input_dim += self.g_shared_dim
            self.shared = ops.embedding(num_embeddings=self.num_classes, embedding_dim=self.g_shared_dim)

        blocks = []
        for index in range(self.num_blocks):
            blocks += [[GenBlock(in_channels=self.in_dims[index], out_channels=self.out_dims[index], g_cond_mtd=self.g_cond_mtd,
                                 affine_input_dim=self.affine_input_dim, MODULES=MODULES)]]
            if index + 1 in attn_g_loc and apply_attn:
                blocks += [[ops.SelfAttention(self.out_dims[index], is_generator=True, MODULES=MODULES)]]
        self.blocks = nn.ModuleList([nn.ModuleList(block) for block in blocks])

        self.bn4 = ops.batchnorm_2d(in_features=self.out_dims[-1])
        self.activation = MODULES.g_act_fn
        self.conv2d5 = MODULES.g_conv2d(in_channels=self.out_dims[-1], out_channels=3, kernel_size=3, stride=1, padding=1)
        self.tanh = nn.Tanh()
        self.conv2d5._sg_rows_pad = 8     # RGB rows are written as 8-channel pixels (16-byte stores / loads around the image boundary)
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
        zs = torch.split(z, self.chunk_size, 1)
        z0 = zs[0]
        if self.g_cond_mtd != "W/O":
            if shared_label is None:
                shared_label = self.shared(label)
            affine_list.append(shared_label)
        if len(affine_list) == 0:
            affines = [item for item in zs[1:]]
        else:
            affines = [torch.cat(affine_list + [item], 1) for item in zs[1:]]

        # every conditional batch norm's [1 + gain(y) | bias(y)] rows in ONE launch: all conditioning vectors exist before the first block runs
        pairs, counter = [], 0
        for blocklist in self.blocks:
            for block in blocklist:
                if not isinstance(block, ops.SelfAttention):
                    if isinstance(block.bn1, ops.ConditionalBatchNorm2d):
                        pairs += [(block.bn1, affines[counter]), (block.bn2, affines[counter])]
                    counter += 1
        F.cbn_prefetch(slot, pairs)

        act = self.linear0.forward_rt(z0, slot)
        act = F.NchwToNhwcFn.apply(act.view(-1, self.in_dims[0], self.bottom, self.bottom), dtype)
        act = ops.block_boundary(self, -1, act)
        counter = 0
        nxt = bank.boundaries(self.blocks) if bank.exchange is not None else None
        with ops.bump_batches_tracked(self):      # (every batch norm of the network runs once below: their counters move in one launch)
            for bi, blocklist in enumerate(self.blocks):
                for block in blocklist:
                    if isinstance(block, ops.SelfAttention):
                        act = block.forward_nhwc(act, slot)
                    else:
                        act = block.forward_nhwc(act, affines[counter], slot)
                        counter += 1
                if nxt is not None:
                    act = bank.mark(act, nxt[bi])      # data parallelism: the backward's return to this point releases the gradients behind it
                act = ops.block_boundary(self, bi, act)
            act = self.bn4.forward_nhwc(act, relu=True)
        act = self.conv2d5.forward_nhwc(act, slot)
        return F.NhwcToNchwFn.apply(act, True, 3)


class DiscOptBlock(nn.Module):
    def __init__(self, in_channels, out_channels, apply_d_sn, MODULES):
        super().__init__()
        self.apply_d_sn = apply_d_sn
        self.conv2d0 = MODULES.d_conv2d(in_channels=in_channels, out_channels=out_channels, kernel_size=1, stride=1, padding=0)
        self.conv2d1 = MODULES.d_conv2d(in_channels=in_channels, out_channels=out_channels, kernel_size=3, stride=1, padding=1)
        self.conv2d2 = MODULES.d_conv2d(in_channels=out_channels, out_channels=out_channels, kernel_size=3, stride=1, padding=1)
        if not apply_d_sn:
            self.bn0 = MODULES.d_bn(in_features=in_channels)
            self.bn1 = MODULES.d_bn(in_features=out_channels)
        self.activation = MODULES.d_act_fn
        self.average_pooling = nn.AvgPool2d(2)
        # the RGB image travels as an 8-channel NHWC tensor (zero-filled) when no batch norm touches it: 16-byte loaders everywhere
        self.cpad = 8 if (apply_d_sn and in_channels < 8) else 0
        if self.cpad:
            self.conv2d1._sg_cin_pad = self.conv2d0._sg_cin_pad = self.cpad

    def forward_nhwc(self, x, slot):
        h = self.conv2d1.forward_nhwc(x, slot)
        if not self.apply_d_sn:
            h = self.bn1.forward_nhwc(h, relu=True)
            h = self.conv2d2.forward_nhwc(h, slot, out_pool=True)
        else:
            if self.cpad == 8 and h.dtype == torch.bfloat16 and not x.requires_grad:      # (an image that wants its gradient -- gradient penalty, R1 -- keeps the chain below)
                # pool(conv2d2(relu h)) + conv2d0(pool x) = pool(conv2d2(relu h) + conv2d0(x)): pooling is linear, so the skip on the image rides in
                # the block tail's launch like the other blocks' (functional.ConvSkipFn; the skip input takes no ReLU here)
                return ops.conv_skip_nhwc(self.conv2d2, self.conv2d0, h, x, slot, in_relu=True, out_pool=True, skip_relu=False)
            h = self.conv2d2.forward_nhwc(h, slot, in_relu=True, out_pool=True)
        x0 = F.AvgPool2Fn.apply(x)
        if not self.apply_d_sn:
            x0 = self.bn0.forward_nhwc(x0)
        return self.conv2d0.forward_nhwc(x0, slot, res=h)


class DiscBlock(nn.Module):
    def __init__(self, in_channels, out_channels, apply_d_sn, MODULES, downsample=True):
        super().__init__()
        self.apply_d_sn = apply_d_sn
        self.downsample = downsample
        self.activation = MODULES.d_act_fn
        self.ch_mismatch = in_channels != out_channels
        if self.ch_mismatch or downsample:
            self.conv2d0 = MODULES.d_conv2d(in_channels=in_channels, out_channels=out_channels, kernel_size=1, stride=1, padding=0)
            if not apply_d_sn:
                self.bn0 = MODULES.d_bn(in_features=in_channels)
        self.conv2d1 = MODULES.d_conv2d(in_channels=in_channels, out_channels=out_channels, kernel_size=3, stride=1, padding=1)
        self.conv2d2 = MODULES.d_conv2d(in_channels=out_channels, out_channels=out_channels, kernel_size=3, stride=1, padding=1)
        if not apply_d_sn:
            self.bn1 = MODULES.d_bn(in_features=in_channels)
            self.bn2 = MODULES.d_bn(in_features=out_channels)
        self.average_pooling = nn.AvgPool2d(2)

    def forward_nhwc(self, x, slot):
        if self.apply_d_sn:
            # nn.ReLU(inplace=True) on x also rewrites the skip tensor x0 (same storage) in the reference:
            # both paths see relu(x)  (reference big_resnet.py:221-242, config.py:476)
            if self.downsample or self.ch_mismatch:
                link = F.GradLink()      # x feeds conv2d1 and the skip: conv2d1's data gradient takes the skip's gradient as its residual
                h = self.conv2d1.forward_nhwc(x, slot, in_relu=True, link=link)
                # pool(conv2d2(relu h)) + pool(conv2d0(relu x)) = pool(conv2d2(relu h) + conv2d0(relu x)): one launch (functional.ConvSkipFn)
                return ops.conv_skip_nhwc(self.conv2d2, self.conv2d0, h, x, slot, in_relu=True, out_pool=self.downsample, link=link)
            h = self.conv2d1.forward_nhwc(x, slot, in_relu=True)
            h = self.conv2d2.forward_nhwc(h, slot, in_relu=True, out_pool=False)
            return F.AddReluFn.apply(h, x)
        h = self.bn1.forward_nhwc(x, relu=True)
        h = self.conv2d1.forward_nhwc(h, slot)
        h = self.bn2.forward_nhwc(h, relu=True)
        h = self.conv2d2.forward_nhwc(h, slot, out_pool=self.downsample)
        if self.downsample or self.ch_mismatch:
            x0 = self.bn0.forward_nhwc(x)
            return self.conv2d0.forward_nhwc(x0, slot, out_pool=self.downsample, res=h)
        return F.AddFn.apply(h, x)


class Discriminator(nn.Module):
    def __init__(self, img_size, d_conv_dim, apply_d_sn, apply_attn, attn_d_loc, d_cond_mtd, aux_cls_type, d_embed_dim, normalize_d_embed,
                 num_classes, d_init, d_depth, mixed_precision, MODULES, MODEL):
        super().__init__()
        d_in_dims_collection = {
            "32": [3] + [d_conv_dim * 2, d_conv_dim * 2, d_conv_dim * 2],
            "64": [3] + [d_conv_dim, d_conv_dim * 2, d_conv_dim * 4, d_conv_dim * 8],
            "128": [3] + [d_conv_dim, d_conv_dim * 2, d_conv_dim * 4, d_conv_dim * 8, d_conv_dim * 16],
            "256": [3] + [d_conv_dim, d_conv_dim * 2, d_conv_dim * 4, d_conv_dim * 8, d_conv_dim * 8, d_conv_dim * 16],
            "512": [3] + [d_conv_dim, d_conv_dim, d_conv_dim * 2, d_conv_dim * 4, d_conv_dim * 8, d_conv_dim * 8, d_conv_dim * 16]
        }
        d_out_dims_collection = {
            "32": [d_conv_dim * 2, d_conv_dim * 2, d_conv_dim * 2, d_conv_dim * 2],
            "64": [d_conv_dim, d_conv_dim * 2, d_conv_dim * 4, d_conv_dim * 8, d_conv_dim * 16],
            "128": [d_conv_dim, d_conv_dim * 2, d_conv_dim * 4, d_conv_dim * 8, d_conv_dim * 16, d_conv_dim * 16],
            "256": [d_conv_dim, d_conv_dim * 2, d_conv_dim * 4, d_conv_dim * 8, d_conv_dim * 8, d_conv_dim * 16, d_conv_dim * 16],
            "512": [d_conv_dim, d_conv_dim, d_conv_dim * 2, d_conv_dim * 4, d_conv_dim * 8, d_conv_dim * 8, d_conv_dim * 16, d_conv_dim * 16]
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

        blocks = []
        for index in range(len(self.in_dims)):
            if index == 0:
                blocks += [[DiscOptBlock(in_channels=self.in_dims[index], out_channels=self.out_dims[index], apply_d_sn=apply_d_sn, MODULES=MODULES)]]
            else:
                blocks += [[DiscBlock(in_channels=self.in_dims[index], out_channels=self.out_dims[index], apply_d_sn=apply_d_sn,
                                      MODULES=MODULES, downsample=down[index])]]
            if index + 1 in attn_d_loc and apply_attn:
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
        h = ops.to_nhwc(x, dtype, self.blocks[0][0].cpad)
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
