"""Host-side mirror of the reference's operator module (reference src/utils/ops.py): same factory names, same ctor
arguments, same parameter / buffer names -- so it plugs into `cfgs.MODULES` (reference src/config.py:435-495) -- but every
forward/backward runs on libsgamd.so's gfx950 kernels.

Module-level tensors are logical NCHW like the reference's; internally they are NHWC (zero-copy when the incoming
tensor is already channels_last). The backbones in `backbones/` call the `*_nhwc` methods directly.
"""
import torch
import torch.nn as nn
from torch.nn import init
import torch.nn.functional as F_

from . import functional as F
from .bank import get_bank

COMPUTE_DTYPE = torch.float32  # default compute dtype of standalone modules (backbones set it per network)


# ---------------------------------------------------------------------------------------------------------
# helpers
# ---------------------------------------------------------------------------------------------------------
def to_nhwc(x, dtype, cpad=0):
    """logical NCHW tensor -> internal NHWC tensor of the compute dtype (cpad: zero-pad the channels, see NchwToNhwcFn)."""
    if x.dim() != 4:
        raise RuntimeError("expected a 4-D NCHW tensor")
    if x.dtype == torch.uint8:      # the dataset's own format: uint8 [N,H,W,3] (reference src/data_util.py:102-142), normalised on the device
        return F.u8_to_nhwc(x, dtype, max(cpad, 3))
    if cpad > x.shape[1]:
        return F.NchwToNhwcFn.apply(x.float(), dtype, cpad)
    if x.dtype == dtype and x.is_contiguous(memory_format=torch.channels_last) and not (x.shape[1] > 1 and x.is_contiguous() and x.shape[2] * x.shape[3] > 1):
        return x.permute(0, 2, 3, 1)
    if x.dtype == torch.float32 and x.is_contiguous():
        return F.NchwToNhwcFn.apply(x, dtype)
    y = x.permute(0, 2, 3, 1).contiguous()
    if y.dtype != dtype:
        y = F.ConvertFn.apply(y, dtype)
    return y


def to_nchw(y):
    return y.permute(0, 3, 1, 2)


BLOCK_HOOKS = [False]      # flipped by tests/test_blocks_gpu.py while a teacher is installed: the forward of a network carries no per-block lookup otherwise


def block_boundary(net, bi, act):
    """Block boundary `bi` of a backbone's forward (-1: in front of the first block). The identity unless a test has switched BLOCK_HOOKS on AND installed a
    teacher on the network (`net._sg_teacher`, tests/test_blocks_gpu.py TeacherForcing): then the activation is handed to it -- it records the block's output
    and may return the tensor the NEXT block is to read instead (teacher forcing: every block sees the emulating oracle's input and upstream gradient, nothing
    compounds across blocks)."""
    if not BLOCK_HOOKS[0]:
        return act
    tf = net.__dict__.get("_sg_teacher")
    return act if tf is None else tf(bi, act)


def _root_and_bank(m):
    """Bank that serves module m: the enclosing network's (set by the backbone) or a private single-layer one."""
    root = m.__dict__.get("_sg_root")
    root = root() if root is not None else None
    if root is None:
        root = m
    dtype = getattr(root, "compute_dtype", None) or getattr(m, "compute_dtype", None) or COMPUTE_DTYPE
    return root, get_bank(root, dtype)


def _standalone_slot(m, x):
    """Standalone use of an op module (operator-factory seam): run the spectral-norm step for this module's own
    bank right here, exactly like torch's forward pre-hook does for the reference."""
    root, bank = _root_and_bank(m)
    if root is m:
        need_graph = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in m.parameters()))
        return bank.begin_forward(need_graph)
    return bank.current


class _WeightLayerMixin:
    """Bookkeeping shared by Conv2d / Linear / Embedding: optional spectral norm state with torch's names
    (weight_orig, weight_u, weight_v) and the matrix view the bank normalises."""
    _sg_weight_layer = True

    def _sg_setup(self, kind, rows, cols, cin, rs, sn, eps=1e-6):
        self._sg_kind, self._sg_rows, self._sg_cols, self._sg_cin, self._sg_rs = kind, rows, cols, cin, rs
        self._sg_sn = bool(sn)
        self._sg_eps = eps
        if sn:
            w = self.weight
            del self._parameters["weight"]
            self.register_parameter("weight_orig", w)
            with torch.no_grad():
                mat = w.reshape(rows, cols)
                u = F_.normalize(w.new_empty(rows).normal_(0, 1), dim=0, eps=eps)
                v = F_.normalize(w.new_empty(cols).normal_(0, 1), dim=0, eps=eps)
            self.register_buffer("weight_u", u)
            self.register_buffer("weight_v", v)

    @property
    def master_weight(self):
        return self.weight_orig if self._sg_sn else self.weight

    def __getattr__(self, name):
        # `module.weight` on a spectral-norm layer: the reference exposes the (last) normalised weight as a plain
        # attribute sharing storage with weight_orig right after construction -- init_weights relies on that.
        if name == "weight" and "weight_orig" in self.__dict__.get("_parameters", {}):
            return self._parameters["weight_orig"].data
        return super().__getattr__(name)


# ---------------------------------------------------------------------------------------------------------
# layers
# ---------------------------------------------------------------------------------------------------------
class Conv2d(_WeightLayerMixin, nn.Conv2d):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True, sn=False,
                 cout_pad=0):
        nn.Conv2d.__init__(self, in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias)
        if self.dilation != (1, 1) or self.groups != 1:
            raise NotImplementedError("dilation/groups are not on the StudioGAN hot path")
        if self.kernel_size[0] != self.kernel_size[1] and self.stride != (1, 1):
            raise NotImplementedError
        kh, kw = self.kernel_size
        self._sg_rows_pad = cout_pad
        self._sg_dgrad_noflip = self.stride[0] != 1     # strided conv: data gradient is a transposed gather (unflipped image)
        self._sg_setup("conv", out_channels, in_channels * kh * kw, in_channels, kh * kw, sn)

    def forward_nhwc(self, x, slot=None, in_relu=False, in_upsample=False, out_pool=False, res=None, link=None, stats=False):
        """link: a functional.GradLink shared with the block's tail (conv_skip_nhwc): this convolution's data gradient then also adds the
        gradient the same input receives through the skip path"""
        rt = self._sg_rt
        slot = slot if slot is not None else rt.bank().current
        kh, kw = self.kernel_size
        # (stats: only a batch norm in batch-statistics mode takes the offer; a frozen network -- eval mode, the FID loop's generator -- reads running statistics)
        cfg = F.ConvCfg(kh, kw, self.stride[0], self.padding[0], self.padding[1], in_relu, in_upsample, out_pool, stats and self.training)
        return F.ConvFn.apply(x, self.master_weight, self.bias, res, rt, slot, cfg, link)

    def forward(self, x):
        slot = _standalone_slot(self, x)
        _, bank = _root_and_bank(self)
        y = self.forward_nhwc(to_nhwc(x, bank.dtype, getattr(self, "_sg_cin_pad", 0)), slot)
        if self._sg_rows_pad and self._sg_rows_pad != self.out_channels:
            y = y[..., :self.out_channels]
        return to_nchw(y)


def conv_skip_nhwc(conv_main, conv_skip, h, x, slot=None, in_relu=False, out_pool=False, skip_upsample=False, link=None, stats=False, skip_relu=None):
    """Tail of a residual block: [pool](conv_main(relu?(h))) + [pool](conv_skip(up?(relu?(x)))) -- ONE fused launch when the kernel takes the
    shape (functional.ConvSkipFn), the two chained launches otherwise. conv_main: 3x3 / pad 1, conv_skip: 1x1."""
    rt2, rt0 = conv_main._sg_rt, conv_skip._sg_rt
    slot = slot if slot is not None else rt2.bank().current
    cfg2 = F.ConvCfg(3, 3, 1, 1, 1, in_relu, False, out_pool, stats)
    cfg0 = F.ConvCfg(1, 1, 1, 0, 0, in_relu if skip_relu is None else skip_relu, skip_upsample, out_pool)
    return F.ConvSkipFn.apply(h, x, conv_main.master_weight, conv_main.bias, conv_skip.master_weight, conv_skip.bias, rt2, rt0, slot, cfg2, cfg0, link)


class ConvTranspose2d(_WeightLayerMixin, nn.ConvTranspose2d):
    """nn.ConvTranspose2d mirror (weight [Cin, Cout, kh, kw]; spectral norm over dim 1 like torch does)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=2, padding=0, dilation=1, groups=1, bias=True, sn=False):
        nn.ConvTranspose2d.__init__(self, in_channels, out_channels, kernel_size, stride, padding, 0, groups, bias, dilation)
        if self.dilation != (1, 1) or self.groups != 1 or self.output_padding != (0, 0):
            raise NotImplementedError
        kh, kw = self.kernel_size
        self._sg_trans = True
        self._sg_dgrad_noflip = True
        self._sg_setup("conv", out_channels, in_channels * kh * kw, in_channels, kh * kw, sn)

    def forward_nhwc(self, x, slot=None):
        rt = self._sg_rt
        slot = slot if slot is not None else rt.bank().current
        kh, kw = self.kernel_size
        cfg = F.ConvCfg(kh, kw, self.stride[0], self.padding[0], self.padding[1])
        return F.ConvTransposeFn.apply(x, self.master_weight, self.bias, rt, slot, cfg)

    def forward(self, x, output_size=None):
        slot = _standalone_slot(self, x)
        _, bank = _root_and_bank(self)
        return to_nchw(self.forward_nhwc(to_nhwc(x, bank.dtype), slot))


class Linear(_WeightLayerMixin, nn.Linear):
    def __init__(self, in_features, out_features, bias=True, sn=False):
        nn.Linear.__init__(self, in_features, out_features, bias)
        self._sg_setup("linear", out_features, in_features, in_features, 1, sn)

    def forward_rt(self, x, slot=None, const_bias=None):
        rt = self._sg_rt
        slot = slot if slot is not None else rt.bank().current
        return F.LinearFn.apply(x, self.master_weight, self.bias, rt, slot, const_bias)

    def forward(self, x):
        slot = _standalone_slot(self, x)
        shp = x.shape
        y = self.forward_rt(x.reshape(-1, shp[-1]), slot)
        return y.reshape(*shp[:-1], self.out_features)


class Embedding(_WeightLayerMixin, nn.Embedding):
    def __init__(self, num_embeddings, embedding_dim, sn=False):
        nn.Embedding.__init__(self, num_embeddings, embedding_dim)
        self._sg_setup("embedding", num_embeddings, embedding_dim, embedding_dim, 1, sn)

    def forward(self, idx):
        if not self._sg_sn:
            return F.EmbeddingFn.apply(self.weight, idx.reshape(-1)).reshape(*idx.shape, self.embedding_dim)
        slot = _standalone_slot(self, idx)
        rt = self._sg_rt
        return F.SNEmbeddingFn.apply(self.weight_orig, idx.reshape(-1), rt, slot).reshape(*idx.shape, self.embedding_dim)


_NBT_BULK = [False]      # True while a network forward runs whose batch norms' num_batches_tracked were bumped in one launch (bump_batches_tracked)


class bump_batches_tracked:
    """with bump_batches_tracked(network): every BatchNorm2d of `network` is about to run exactly once in training mode with tracked statistics (a generator
    forward): their num_batches_tracked buffers -- adjacent int64 words of the network's buffer arena -- take their + 1 in ONE launch, and the modules skip their
    own. Anything else (a frozen or non-tracking batch norm, an arena that .to() / deepcopy replaced, foreign int64 buffers) leaves the per-module increments on."""

    def __init__(self, root):
        self.on = False
        bns = root.__dict__.get("_sg_bn_list")
        if bns is None:
            bns = [m for m in root.modules() if isinstance(m, BatchNorm2d)]
            root.__dict__["_sg_bn_list"] = bns
        if not bns or _NBT_BULK[0]:
            return
        for m in bns:
            if not (m.training and m.track_running_stats and m.running_mean is not None and m.num_batches_tracked is not None):
                return
        from .bank import get_buffer_arena
        ar = get_buffer_arena(root)
        if ar.inumel != len(bns) or any(b is not m.num_batches_tracked for b, m in zip(ar.ibufs, bns)):
            return
        with torch.no_grad():
            ar.idata.add_(1)
        self.on = True

    def __enter__(self):
        if self.on:
            _NBT_BULK[0] = True
        return self

    def __exit__(self, *a):
        if self.on:
            _NBT_BULK[0] = False


class BatchNorm2d(nn.BatchNorm2d):
    """nn.BatchNorm2d(eps=1e-4, momentum=0.1) semantics (reference src/utils/ops.py:227-228) incl. the
    track_running_stats toggling the driver does (reference src/utils/misc.py:239-267)."""
    sync_group = None  # set to a process group (or True for WORLD) to get synchronised statistics

    def _cfg(self, relu):
        # torch: bn_training = self.training or (running_mean is None and running_var is None)
        batch_stats = self.training or (self.running_mean is None)
        # F.batch_norm gets running stats only `if not self.training or self.track_running_stats`
        track = self.training and self.track_running_stats and self.running_mean is not None
        mom = 0.0 if self.momentum is None else self.momentum
        return F.BNCfg(batch_stats, track, self.eps, mom, relu, self.sync_group)

    def forward_nhwc(self, x, gain=None, bias=None, relu=False, link=None, packed=False):
        cfg = self._cfg(relu)
        cfg.packed = packed
        if cfg.track and self.num_batches_tracked is not None and not _NBT_BULK[0]:
            self.num_batches_tracked.add_(1)
        if gain is None and self.affine:
            gain, bias = self.weight, self.bias
        return F.BNFn.apply(x, gain, bias, self.running_mean, self.running_var, cfg, link)

    def forward(self, x):
        dtype = getattr(self, "compute_dtype", None) or (x.dtype if x.dtype in (torch.float32, torch.bfloat16) else COMPUTE_DTYPE)
        return to_nchw(self.forward_nhwc(to_nhwc(x, dtype)))


class ConditionalBatchNorm2d(nn.Module):
    """reference src/utils/ops.py:14-28:  bn(x) * (1 + gain(y)) + bias(y)"""

    def __init__(self, in_features, out_features, MODULES):
        super().__init__()
        self.in_features = in_features
        self.bn = batchnorm_2d(out_features, eps=1e-4, momentum=0.1, affine=False)
        self.gain = MODULES.g_linear(in_features=in_features, out_features=out_features, bias=False)
        self.bias = MODULES.g_linear(in_features=in_features, out_features=out_features, bias=False)
        self.register_buffer("_ones", torch.ones(out_features), persistent=False)
        self.register_buffer("_ones2", torch.cat([torch.ones(out_features), torch.zeros(out_features)]), persistent=False)

    def forward_nhwc(self, x, y, slot=None, relu=False, link=None):
        pre = slot.__dict__.get("cbn_rows") if slot is not None else None
        if pre:
            gb = pre.pop(id(self), None)      # computed with every other conditional batch norm of this forward (functional.cbn_prefetch)
            if gb is not None:
                return self.bn.forward_nhwc(x, gb, None, relu, link, packed=True)
        if F._CBN_MERGED[0] and self.gain.bias is None and self.bias.bias is None:
            # [1 + gain(y) | bias(y)] as one GEMM over the two adjacent weight images (functional.CbnAffineFn); the '1 +' rides in its epilogue bias
            rt_g, rt_b = self.gain._sg_rt, self.bias._sg_rt
            slot = slot if slot is not None else rt_g.bank().current
            gb = F.CbnAffineFn.apply(y, self.gain.master_weight, self.bias.master_weight, rt_g, rt_b, slot, self._ones2)
            return self.bn.forward_nhwc(x, gb, None, relu, link, packed=True)
        gain = self.gain.forward_rt(y, slot, const_bias=self._ones)  # 1 + gain(y) through the GEMM epilogue bias
        bias = self.bias.forward_rt(y, slot)
        return self.bn.forward_nhwc(x, gain, bias, relu, link)

    def forward(self, x, y):
        _, bank = _root_and_bank(self.gain)
        if self.gain.__dict__.get("_sg_root") is None:
            raise RuntimeError("ConditionalBatchNorm2d must live inside a studiogan_amd backbone (its two linears share the network's bank)")
        return to_nchw(self.forward_nhwc(to_nhwc(x, bank.dtype), y))


class SelfAttention(nn.Module):
    """reference src/utils/ops.py:31-103. theta/phi outputs are padded to a multiple of 8 channels internally."""

    def __init__(self, in_channels, is_generator, MODULES):
        super().__init__()
        self.in_channels = in_channels
        conv = MODULES.g_conv2d if is_generator else MODULES.d_conv2d
        c8, c2 = in_channels // 8, in_channels // 2
        pad8 = (c8 + 7) // 8 * 8
        self.conv1x1_theta = conv(in_channels=in_channels, out_channels=c8, kernel_size=1, stride=1, padding=0, bias=False)
        self.conv1x1_phi = conv(in_channels=in_channels, out_channels=c8, kernel_size=1, stride=1, padding=0, bias=False)
        self.conv1x1_g = conv(in_channels=in_channels, out_channels=c2, kernel_size=1, stride=1, padding=0, bias=False)
        self.conv1x1_attn = conv(in_channels=c2, out_channels=in_channels, kernel_size=1, stride=1, padding=0, bias=False)
        for m in (self.conv1x1_theta, self.conv1x1_phi):
            m._sg_rows_pad = pad8
        self.sigma = nn.Parameter(torch.zeros(1), requires_grad=True)

    def forward_nhwc(self, x, slot=None):
        # x has four readers (theta, phi, g, the residual): their gradients are summed inside the three data-gradient launches
        link = F.GradLink(chain=True) if (F._GRAD_LINK[0] and torch.is_grad_enabled() and x.requires_grad) else None
        theta = self.conv1x1_theta.forward_nhwc(x, slot, link=link)
        phi = self.conv1x1_phi.forward_nhwc(x, slot, link=link)
        g = self.conv1x1_g.forward_nhwc(x, slot, link=link)
        o = F.AttnCoreFn.apply(theta, phi, g)
        rt = self.conv1x1_attn._sg_rt
        slot = slot if slot is not None else rt.bank().current
        return F.AttnOutFn.apply(x, o, self.conv1x1_attn.master_weight, self.sigma, rt, slot, link)

    def forward(self, x):
        _, bank = _root_and_bank(self.conv1x1_theta)
        return to_nchw(self.forward_nhwc(to_nhwc(x, bank.dtype)))


class LeCamEMA(object):
    """reference src/utils/ops.py:106-133 (host-side scalar EMAs of the discriminator's mean logits / losses)."""

    def __init__(self, init=7777, decay=0.9, start_iter=0):
        self.G_loss = init
        self.D_loss_real = init
        self.D_loss_fake = init
        self.D_real = init
        self.D_fake = init
        self.decay = decay
        self.start_itr = start_iter

    def update(self, cur, mode, itr):
        decay = 0.0 if itr < self.start_itr else self.decay
        if mode not in ("G_loss", "D_loss_real", "D_loss_fake", "D_real", "D_fake"):
            raise ValueError(mode)
        setattr(self, mode, getattr(self, mode) * decay + cur * (1 - decay))


# ---------------------------------------------------------------------------------------------------------
# factory functions with the reference's names/signatures (reference src/utils/ops.py:165-228)
# ---------------------------------------------------------------------------------------------------------
def conv2d(in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True):
    return Conv2d(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias, sn=False)


def snconv2d(in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True):
    return Conv2d(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias, sn=True)


def deconv2d(in_channels, out_channels, kernel_size, stride=2, padding=0, dilation=1, groups=1, bias=True):
    return ConvTranspose2d(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias, sn=False)


def sndeconv2d(in_channels, out_channels, kernel_size, stride=2, padding=0, dilation=1, groups=1, bias=True):
    return ConvTranspose2d(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias, sn=True)


def linear(in_features, out_features, bias=True):
    return Linear(in_features, out_features, bias, sn=False)


def snlinear(in_features, out_features, bias=True):
    return Linear(in_features, out_features, bias, sn=True)


def embedding(num_embeddings, embedding_dim):
    return Embedding(num_embeddings, embedding_dim, sn=False)


def sn_embedding(num_embeddings, embedding_dim):
    return Embedding(num_embeddings, embedding_dim, sn=True)


def batchnorm_2d(in_features, eps=1e-4, momentum=0.1, affine=True):
    return BatchNorm2d(in_features, eps=eps, momentum=momentum, affine=affine, track_running_stats=True)


def init_weights(modules, initialize):
    """reference src/utils/ops.py:135-162 (orthogonal / N(0,0.02) / xavier on conv, linear, embedding weights)."""
    for module in modules():
        if isinstance(module, (nn.Conv2d, nn.ConvTranspose2d, nn.Linear)):
            w = module.master_weight if hasattr(module, "master_weight") else module.weight
            if initialize == "ortho":
                init.orthogonal_(w)
            elif initialize == "N02":
                init.normal_(w, 0, 0.02)
            elif initialize in ["glorot", "xavier"]:
                init.xavier_uniform_(w)
            else:
                continue
            if module.bias is not None:
                module.bias.data.fill_(0.)
        elif isinstance(module, nn.Embedding):
            w = module.master_weight if hasattr(module, "master_weight") else module.weight
            if initialize == "ortho":
                init.orthogonal_(w)
            elif initialize == "N02":
                init.normal_(w, 0, 0.02)
            elif initialize in ["glorot", "xavier"]:
                init.xavier_uniform_(w)


def adopt(root, compute_dtype):
    """Mark every op module under `root` as belonging to root's weight bank."""
    import weakref
    root.compute_dtype = compute_dtype
    ref = weakref.ref(root)
    for m in root.modules():
        if m is not root:
            m.__dict__["_sg_root"] = ref


class Modules:
    """Drop-in for `cfgs.MODULES` (reference src/config.py:435-495) built from the same MODEL flags."""

    def __init__(self, apply_g_sn=False, apply_d_sn=False, g_cond_mtd="W/O", backbone="big_resnet", g_act_fn="ReLU", d_act_fn="ReLU", g_info_injection="N/A"):
        self.g_conv2d = snconv2d if apply_g_sn else conv2d
        self.g_deconv2d = sndeconv2d if apply_g_sn else deconv2d
        self.d_deconv2d = sndeconv2d if apply_d_sn else deconv2d
        self.g_linear = snlinear if apply_g_sn else linear
        self.g_embedding = sn_embedding if apply_g_sn else embedding
        self.d_conv2d = snconv2d if apply_d_sn else conv2d
        self.d_linear = snlinear if apply_d_sn else linear
        self.d_embedding = sn_embedding if apply_d_sn else embedding
        if g_cond_mtd == "cBN" or g_info_injection == "cBN" or backbone == "big_resnet":      # src/config.py:458
            self.g_bn = ConditionalBatchNorm2d
        else:
            self.g_bn = batchnorm_2d
        if not apply_d_sn:
            self.d_bn = batchnorm_2d
        if g_act_fn != "ReLU" or d_act_fn != "ReLU":
            raise NotImplementedError("only ReLU is on the benchmarked hot path")
        self.g_act_fn = nn.ReLU(inplace=True)
        self.d_act_fn = nn.ReLU(inplace=True)
