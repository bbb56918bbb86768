"""ResNet GAN backbone (SNGAN / SAGAN / WGAN-GP family) with the reference's constructor and forward contracts
(reference src/models/resnet.py:62-169 Generator, :172-442 Discriminator). The discriminator is block-for-block the
BigGAN one (reference resnet.py:172-260 == big_resnet.py:161-242), so it is shared; the generator differs in its
conditioning: the whole z feeds linear0 and the cBN affine input is the one-hot label (resnet.py:137-152)."""
import torch
import torch.nn as nn
import torch.nn.functional as TF

from .. import functional as F
from .. import ops
from ..bank import get_bank
from .big_resnet import DiscOptBlock, DiscBlock, Discriminator, _dtype, _need_graph  # noqa: F401


class GenBlock(nn.Module):
    def __init__(self, in_channels, out_channels, g_cond_mtd, g_info_injection, affine_input_dim, MODULES):
        super().__init__()
        self.g_cond_mtd = g_cond_mtd
        self.g_info_injection = g_info_injection
        self.conditional = g_cond_mtd == "cBN" or g_info_injection == "cBN"
        if g_cond_mtd == "W/O" and g_info_injection in ["N/A", "concat"]:
            self.bn1 = MODULES.g_bn(in_features=in_channels)
            self.bn2 = MODULES.g_bn(in_features=out_channels)
        elif self.conditional:
            self.bn1 = MODULES.g_bn(affine_input_dim, in_channels, MODULES)
            self.bn2 = MODULES.g_bn(affine_input_dim, out_channels, MODULES)
        else:
            raise NotImplementedError
        self.activation = MODULES.g_act_fn
        self.conv2d0 = MODULES.g_conv2d(in_channels=in_channels, out_channels=out_channels, kernel_size=1, stride=1, padding=0)
        self.conv2d1 = MODULES.g_conv2d(in_channels=in_channels, out_channels=out_channels, kernel_size=3, stride=1, padding=1)
        self.conv2d2 = MODULES.g_conv2d(in_channels=out_channels, out_channels=out_channels, kernel_size=3, stride=1, padding=1)

    def forward_nhwc(self, x, affine, slot):
        link = F.GradLink()      # x feeds bn1 and the skip: bn1's backward adds the skip's gradient in its own launch
        h = self.bn1.forward_nhwc(x, affine, slot, relu=True, link=link) if self.conditional else self.bn1.forward_nhwc(x, relu=True, link=link)
        # (stats=True: a batch norm reads the result next; where the kernel can, its statistics come out of the convolution's epilogue)
        h = self.conv2d1.forward_nhwc(h, slot, in_upsample=True, stats=True)
        h = self.bn2.forward_nhwc(h, affine, slot, relu=True) if self.conditional else self.bn2.forward_nhwc(h, relu=True)
        # conv2d2(h) + conv2d0(up(x)): one launch where the fused kernel takes the shape (bf16; functional.ConvSkipFn), two otherwise
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
        self.num_classes = num_classes
        self.g_cond_mtd = g_cond_mtd
        self.mixed_precision = mixed_precision
        self.MODEL = MODEL
        self.in_dims = g_in_dims_collection[str(img_size)]
        self.out_dims = g_out_dims_collection[str(img_size)]
        self.bottom = 4
        self.num_blocks = len(self.in_dims)
        self.affine_input_dim = 0
        # InfoGAN (reference src/models/resnet.py:95-107,142-147): codes behind z, through a mixing layer ("concat") or as conditioning of the block norms ("cBN")
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
        self.linear0 = MODULES.g_linear(in_features=self.z_dim, out_features=self.in_dims[0] * self.bottom * self.bottom, bias=True)
        if self.g_cond_mtd != "W/O" and self.g_cond_mtd == "cBN":
            self.affine_input_dim += self.num_classes
        blocks = []
        for index in range(self.num_blocks):
            blocks += [[GenBlock(in_channels=self.in_dims[index], out_channels=self.out_dims[index], g_cond_mtd=self.g_cond_mtd,
                                 g_info_injection=self.g_info_injection, affine_input_dim=self.affine_input_dim, MODULES=MODULES)]]
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
        act = F.NchwToNhwcFn.apply(act.view(-1, self.in_dims[0], self.bottom, self.bottom), dtype)
        act = ops.block_boundary(self, -1, act)
        nxt = bank.boundaries(self.blocks) if bank.exchange is not None else None
        for bi, blocklist in enumerate(self.blocks):
            for block in blocklist:
                if isinstance(block, ops.SelfAttention):
                    act = block.forward_nhwc(act, slot)
                else:
                    act = block.forward_nhwc(act, affines, slot)
            act = ops.block_boundary(self, bi, act)
            if nxt is not None:
                act = bank.mark(act, nxt[bi])      # data parallelism: the backward's return to this point releases the gradients behind it
        act = self.bn4.forward_nhwc(act, relu=True)
        act = self.conv2d5.forward_nhwc(act, slot)
        return F.NhwcToNchwFn.apply(act, True, 3)
