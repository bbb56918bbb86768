"""FID / IS feature-extraction path on the GPU, mirroring the reference's evaluation interfaces:

  LoadEvalModel(...).get_outputs(x, quantize)           reference src/metrics/preparation.py:43-122
  generate_images_and_stack_features(...)               reference src/metrics/features.py:17-65
  calculate_moments / frechet_inception_distance        reference src/metrics/fid.py:34-98
  calculate_kl_div / top-k accuracy                     reference src/metrics/ins.py:28-79

What changes underneath: the per-batch GPU->CPU->per-image-resize->GPU round trip (reference src/utils/ops.py:251-263) is
ONE kernel (`sg_quantize_resize_normalize`: bit-exact uint8 quantisation, bilinear align_corners=False, normalise,
NHWC); InceptionV3 runs as 94 fused conv+foldedBN+ReLU launches writing straight into the concat buffers (no cat
copies), pools and the fc on libsgamd.so; FID moments are accumulated on the device in fp64 instead of gathering
50k x 2048 features to the host. The matrix square root stays on the host in fp64 like the reference.
"""
import math

import numpy as np
import torch

from . import _lib as L
from . import functional as F

BN_EPS = 1e-3


def _spec():
    def A(p, c, pf):
        return [(p + ".branch1x1", c, 64, 1, 1, 1, 0, 0), (p + ".branch5x5_1", c, 48, 1, 1, 1, 0, 0), (p + ".branch5x5_2", 48, 64, 5, 5, 1, 2, 2),
                (p + ".branch3x3dbl_1", c, 64, 1, 1, 1, 0, 0), (p + ".branch3x3dbl_2", 64, 96, 3, 3, 1, 1, 1), (p + ".branch3x3dbl_3", 96, 96, 3, 3, 1, 1, 1),
                (p + ".branch_pool", c, pf, 1, 1, 1, 0, 0)]

    def B(p, c):
        return [(p + ".branch3x3", c, 384, 3, 3, 2, 0, 0), (p + ".branch3x3dbl_1", c, 64, 1, 1, 1, 0, 0), (p + ".branch3x3dbl_2", 64, 96, 3, 3, 1, 1, 1),
                (p + ".branch3x3dbl_3", 96, 96, 3, 3, 2, 0, 0)]

    def Cb(p, c, c7):
        return [(p + ".branch1x1", c, 192, 1, 1, 1, 0, 0), (p + ".branch7x7_1", c, c7, 1, 1, 1, 0, 0), (p + ".branch7x7_2", c7, c7, 1, 7, 1, 0, 3),
                (p + ".branch7x7_3", c7, 192, 7, 1, 1, 3, 0), (p + ".branch7x7dbl_1", c, c7, 1, 1, 1, 0, 0), (p + ".branch7x7dbl_2", c7, c7, 7, 1, 1, 3, 0),
                (p + ".branch7x7dbl_3", c7, c7, 1, 7, 1, 0, 3), (p + ".branch7x7dbl_4", c7, c7, 7, 1, 1, 3, 0), (p + ".branch7x7dbl_5", c7, 192, 1, 7, 1, 0, 3),
                (p + ".branch_pool", c, 192, 1, 1, 1, 0, 0)]

    def D(p, c):
        return [(p + ".branch3x3_1", c, 192, 1, 1, 1, 0, 0), (p + ".branch3x3_2", 192, 320, 3, 3, 2, 0, 0), (p + ".branch7x7x3_1", c, 192, 1, 1, 1, 0, 0),
                (p + ".branch7x7x3_2", 192, 192, 1, 7, 1, 0, 3), (p + ".branch7x7x3_3", 192, 192, 7, 1, 1, 3, 0), (p + ".branch7x7x3_4", 192, 192, 3, 3, 2, 0, 0)]

    def E(p, c):
        return [(p + ".branch1x1", c, 320, 1, 1, 1, 0, 0), (p + ".branch3x3_1", c, 384, 1, 1, 1, 0, 0), (p + ".branch3x3_2a", 384, 384, 1, 3, 1, 0, 1),
                (p + ".branch3x3_2b", 384, 384, 3, 1, 1, 1, 0), (p + ".branch3x3dbl_1", c, 448, 1, 1, 1, 0, 0), (p + ".branch3x3dbl_2", 448, 384, 3, 3, 1, 1, 1),
                (p + ".branch3x3dbl_3a", 384, 384, 1, 3, 1, 0, 1), (p + ".branch3x3dbl_3b", 384, 384, 3, 1, 1, 1, 0), (p + ".branch_pool", c, 192, 1, 1, 1, 0, 0)]
    stem = [("Conv2d_1a_3x3", 3, 32, 3, 3, 2, 0, 0), ("Conv2d_2a_3x3", 32, 32, 3, 3, 1, 0, 0), ("Conv2d_2b_3x3", 32, 64, 3, 3, 1, 1, 1),
            ("Conv2d_3b_1x1", 64, 80, 1, 1, 1, 0, 0), ("Conv2d_4a_3x3", 80, 192, 3, 3, 1, 0, 0)]
    layers = (stem + A("Mixed_5b", 192, 32) + A("Mixed_5c", 256, 64) + A("Mixed_5d", 288, 64) + B("Mixed_6a", 288) + Cb("Mixed_6b", 768, 128) +
              Cb("Mixed_6c", 768, 160) + Cb("Mixed_6d", 768, 160) + Cb("Mixed_6e", 768, 192) + D("Mixed_7a", 768) + E("Mixed_7b", 1280) + E("Mixed_7c", 2048))
    return {l[0]: l for l in layers}


SPEC = _spec()


def synthetic_state_dict(seed=0):
    """Seeded random weights under torchvision's inception_v3 state_dict names (94 BasicConv2d + fc), for throughput measurements
    when the pretrained FID weights (reference src/metrics/inception_net.py:13,130, downloaded at run time) are not available.
    Same values as the test oracle's generator for the same seed (same draw order), without importing it."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for name, (_, cin, cout, kh, kw, _, _, _) in SPEC.items():
        fan = cin * kh * kw
        sd[name + ".conv.weight"] = torch.randn(cout, cin, kh, kw, generator=g) * math.sqrt(2.0 / fan)
        sd[name + ".bn.weight"] = 1.0 + 0.1 * torch.randn(cout, generator=g)
        sd[name + ".bn.bias"] = 0.1 * torch.randn(cout, generator=g)
        sd[name + ".bn.running_mean"] = 0.1 * torch.randn(cout, generator=g)
        sd[name + ".bn.running_var"] = 0.5 + torch.rand(cout, generator=g)
        sd[name + ".bn.num_batches_tracked"] = torch.tensor(0)
    sd["fc.weight"] = torch.randn(1008, 2048, generator=g) * math.sqrt(1.0 / 2048)
    sd["fc.bias"] = 0.01 * torch.randn(1008, generator=g)
    return sd


# ---- the pin of the published FID network -------------------------------------------------------------------------
# reference src/metrics/inception_net.py:13,117-130: torchvision inception_v3(num_classes=1008, aux_logits=False) patched with the FID blocks,
# then `load_state_dict(load_state_dict_from_url(FID_WEIGHTS_URL))`. The file cannot be fetched here, but everything that identifies it can be
# enforced: its name carries the first 8 hex digits of its sha256 (the torch.hub convention `<name>-<sha256 prefix>.pth`, which
# load_state_dict_from_url(check_hash=True) verifies), and a strict load needs exactly these tensors with exactly these shapes.
FID_WEIGHTS_URL = "https://github.com/mseitzer/pytorch-fid/releases/download/fid_weights/pt_inception-2015-12-05-6726825d.pth"
FID_WEIGHTS_FILE = "pt_inception-2015-12-05-6726825d.pth"
FID_WEIGHTS_SHA256_PREFIX = "6726825d"


def inception_manifest():
    """{state_dict key: shape} of the FID InceptionV3: 94 BasicConv2d (conv weight without bias + BatchNorm2d(eps 1e-3) weight / bias / running
    statistics) and the 1008-way fc. `num_batches_tracked` entries are optional (a checkpoint converted from TensorFlow has none and torch's
    strict load tolerates their absence for such BatchNorm modules)."""
    m = {}
    for name, (_, cin, cout, kh, kw, _, _, _) in SPEC.items():
        m[name + ".conv.weight"] = (cout, cin, kh, kw)
        for t in ("weight", "bias", "running_mean", "running_var"):
            m[f"{name}.bn.{t}"] = (cout,)
    m["fc.weight"] = (1008, 2048)
    m["fc.bias"] = (1008,)
    return m


def validate_inception_state_dict(sd):
    """What `inception.load_state_dict(state_dict)` (strict) enforces in the reference: every expected tensor present with its shape, floating
    point, nothing else in the file. Raises RuntimeError naming the offending keys."""
    man = inception_manifest()
    keys = {k for k in sd if not k.endswith("num_batches_tracked")}
    missing = sorted(set(man) - keys)
    unexpected = sorted(keys - set(man))
    wrong = sorted(k for k in man if k in sd and (tuple(sd[k].shape) != man[k] or not torch.is_floating_point(sd[k])))
    if missing or unexpected or wrong:
        def head(v):
            return ", ".join(v[:4]) + (f" (+{len(v) - 4} more)" if len(v) > 4 else "")
        raise RuntimeError("not an FID InceptionV3 state_dict (" + FID_WEIGHTS_FILE + "): "
                           + (f"missing {len(missing)}: {head(missing)}; " if missing else "")
                           + (f"unexpected {len(unexpected)}: {head(unexpected)}; " if unexpected else "")
                           + (f"wrong shape / dtype {len(wrong)}: {head(wrong)}" if wrong else ""))
    return True


def load_fid_weights(path, check_hash=True):
    """torch.load of the published FID weights with the hash check torch.hub applies to `<name>-<sha256 prefix>.pth` files and the strict
    structural check; returns (state_dict, sha256 hex)."""
    import hashlib
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for chunk in iter(lambda: f.read(1 << 20), b""):
            h.update(chunk)
    digest = h.hexdigest()
    if check_hash and not digest.startswith(FID_WEIGHTS_SHA256_PREFIX):
        raise RuntimeError(f"{path}: sha256 {digest[:16]}... does not start with {FID_WEIGHTS_SHA256_PREFIX} -- not {FID_WEIGHTS_FILE}")
    sd = torch.load(path, map_location="cpu")
    validate_inception_state_dict(sd)
    return sd, digest


class InceptionV3:
    """FID InceptionV3 (torchvision structure + the reference's FID patches, src/metrics/inception_net.py:135-249),
    inference only, BN folded into the convolutions at load time. `state_dict` uses torchvision's key names, i.e. the
    reference's `pt_inception-2015-12-05-6726825d.pth` loads as is."""

    def __init__(self, state_dict, device, dtype=torch.float32, f32_mode="exact"):
        """f32_mode (dtype fp32 only): "exact" = fp32 MFMA; "bf16x3" = fp32 tensors, every convolution's operands split into two bf16 terms in registers and
        contracted with three bf16 MFMAs per k-tile (functional.f32_mode): ~2^-16 relative per product, measured 3e-6 on the pool3 features against the oracle."""
        self.device, self.dtype = device, dtype
        if f32_mode not in F.f32_mode.MODES:
            raise ValueError(f"InceptionV3: f32_mode {f32_mode!r}")
        self.f32_mode = f32_mode if dtype == torch.float32 else "exact"
        self.w, self.b = {}, {}
        with torch.no_grad():
            for name, (_, cin, cout, kh, kw, stride, ph, pw) in SPEC.items():
                w = state_dict[name + ".conv.weight"].to(device=device, dtype=torch.float32)
                assert tuple(w.shape) == (cout, cin, kh, kw), f"{name}: weight shape {tuple(w.shape)}"
                g, be = state_dict[name + ".bn.weight"].to(device).float(), state_dict[name + ".bn.bias"].to(device).float()
                mu, var = state_dict[name + ".bn.running_mean"].to(device).float(), state_dict[name + ".bn.running_var"].to(device).float()
                scale = g / torch.sqrt(var + BN_EPS)
                self.w[name] = (w * scale.view(-1, 1, 1, 1)).permute(0, 2, 3, 1).contiguous().to(dtype)   # [Cout][R][S][Cin]
                self.b[name] = (be - mu * scale).contiguous()
            self.fc_w = state_dict["fc.weight"].to(device=device, dtype=torch.float32).contiguous()
            self.fc_b = state_dict["fc.bias"].to(device=device, dtype=torch.float32).contiguous()

    # -- building blocks ---------------------------------------------------------------------------------------
    def _bc(self, x, name, out=None, coff=0):
        _, cin, cout, kh, kw, stride, ph, pw = SPEC[name]
        return F.conv2d_raw(x, self.w[name].data_ptr(), cin, cout, kh, kw, stride, ph, pw, 0, L.EPI_RELU, bias=self.b[name], out=out, out_coff=coff)

    def _pool(self, x, k, stride, pad, mode, out=None, coff=0):
        N, H, W, Cc = x.shape
        OH, OW = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
        if out is None:
            out = torch.empty((N, OH, OW, Cc), dtype=x.dtype, device=x.device)
        L.call("sg_pool2d", L.dt(x), L.ptr(x), L.ptr(out), N, H, W, Cc, k, stride, pad, mode, out.shape[-1], coff, L.stream())
        return out

    def _cat(self, x, ch, stride=1):
        N, H, W, _ = x.shape
        if stride == 2:
            H, W = (H - 3) // 2 + 1, (W - 3) // 2 + 1
        return torch.empty((N, H, W, ch), dtype=x.dtype, device=x.device)

    def _A(self, x, p, pf):
        o = self._cat(x, 64 + 64 + 96 + pf)
        self._bc(x, p + ".branch1x1", o, 0)
        self._bc(self._bc(x, p + ".branch5x5_1"), p + ".branch5x5_2", o, 64)
        self._bc(self._bc(self._bc(x, p + ".branch3x3dbl_1"), p + ".branch3x3dbl_2"), p + ".branch3x3dbl_3", o, 128)
        self._bc(self._pool(x, 3, 1, 1, 2), p + ".branch_pool", o, 224)
        return o

    def _B(self, x, p):
        cin = x.shape[3]
        o = self._cat(x, 384 + 96 + cin, stride=2)
        self._bc(x, p + ".branch3x3", o, 0)
        self._bc(self._bc(self._bc(x, p + ".branch3x3dbl_1"), p + ".branch3x3dbl_2"), p + ".branch3x3dbl_3", o, 384)
        self._pool(x, 3, 2, 0, 0, o, 480)
        return o

    def _C(self, x, p):
        o = self._cat(x, 768)
        self._bc(x, p + ".branch1x1", o, 0)
        self._bc(self._bc(self._bc(x, p + ".branch7x7_1"), p + ".branch7x7_2"), p + ".branch7x7_3", o, 192)
        t = x
        for i in range(1, 5):
            t = self._bc(t, p + f".branch7x7dbl_{i}")
        self._bc(t, p + ".branch7x7dbl_5", o, 384)
        self._bc(self._pool(x, 3, 1, 1, 2), p + ".branch_pool", o, 576)
        return o

    def _D(self, x, p):
        cin = x.shape[3]
        o = self._cat(x, 320 + 192 + cin, stride=2)
        self._bc(self._bc(x, p + ".branch3x3_1"), p + ".branch3x3_2", o, 0)
        t = x
        for i in range(1, 4):
            t = self._bc(t, p + f".branch7x7x3_{i}")
        self._bc(t, p + ".branch7x7x3_4", o, 320)
        self._pool(x, 3, 2, 0, 0, o, 512)
        return o

    def _E(self, x, p, pool_mode):
        o = self._cat(x, 2048)
        self._bc(x, p + ".branch1x1", o, 0)
        t = self._bc(x, p + ".branch3x3_1")
        self._bc(t, p + ".branch3x3_2a", o, 320)
        self._bc(t, p + ".branch3x3_2b", o, 704)
        t = self._bc(self._bc(x, p + ".branch3x3dbl_1"), p + ".branch3x3dbl_2")
        self._bc(t, p + ".branch3x3dbl_3a", o, 1088)
        self._bc(t, p + ".branch3x3dbl_3b", o, 1472)
        self._bc(self._pool(x, 3, 1, 1, pool_mode), p + ".branch_pool", o, 1856)
        return o

    @torch.no_grad()
    def forward_nhwc(self, x):
        """x: [B,299,299,3] NHWC in the compute dtype, values in [-1,1] -> (pool3 features [B,2048], logits [B,1008]), fp32."""
        if self.f32_mode != "exact":
            with F.f32_mode(self.f32_mode):
                return self._forward_nhwc(x)
        return self._forward_nhwc(x)

    def _forward_nhwc(self, x):
        x = self._bc(self._bc(self._bc(x, "Conv2d_1a_3x3"), "Conv2d_2a_3x3"), "Conv2d_2b_3x3")
        x = self._pool(x, 3, 2, 0, 0)
        x = self._bc(self._bc(x, "Conv2d_3b_1x1"), "Conv2d_4a_3x3")
        x = self._pool(x, 3, 2, 0, 0)
        x = self._A(x, "Mixed_5b", 32); x = self._A(x, "Mixed_5c", 64); x = self._A(x, "Mixed_5d", 64)
        x = self._B(x, "Mixed_6a")
        for n in ("Mixed_6b", "Mixed_6c", "Mixed_6d", "Mixed_6e"):
            x = self._C(x, n)
        x = self._D(x, "Mixed_7a")
        x = self._E(x, "Mixed_7b", 2)   # average pool without padded zeros (count_include_pad=False)
        x = self._E(x, "Mixed_7c", 0)   # max pool
        N, H, W, Cc = x.shape
        feat = torch.empty((N, Cc), dtype=torch.float32, device=x.device)
        L.call("sg_global_avgpool", L.dt(x), L.ptr(x), L.ptr(feat), N, H * W, Cc, L.stream())
        logits = torch.empty((N, 1008), dtype=torch.float32, device=x.device)
        F.gemm_raw(L.F32, self.fc_w, 0, 2048, feat, 0, 2048, logits, 1008, 1008, N, 2048, bias=self.fc_b)
        return feat, logits


def preprocess(x, dtype, quantize=True, size=299, want_uint8=False):
    """ops.quantize_images + resize_images('legacy') + normalise on the device (reference utils/ops.py:251-263)."""
    x = x.float().contiguous()
    N, Cc, H, W = x.shape
    out = torch.empty((N, size, size, Cc), dtype=dtype, device=x.device)
    q = torch.empty((N, Cc, H, W), dtype=torch.uint8, device=x.device) if want_uint8 else None
    L.call("sg_quantize_resize_normalize", L.dt(dtype), L.ptr(x), L.ptr(out), L.ptr(q), N, Cc, H, W, size, size, 1 if quantize else 0, L.stream())
    return (out, q) if want_uint8 else out


_PIL_FILTERS = {"bilinear": (1.0, lambda x: np.where(np.abs(x) < 1.0, 1.0 - np.abs(x), 0.0)),
                "bicubic": (2.0, None)}


def _bicubic(x, a=-0.5):
    x = np.abs(x)
    return np.where(x < 1.0, ((a + 2.0) * x - (a + 3.0)) * x * x + 1.0, np.where(x < 2.0, (((x - 5.0) * x + 8.0) * x - 4.0) * a, 0.0))


def pil_coeffs(in_size, out_size, filt):
    """Pillow's precompute_coeffs (src/libImaging/Resample.c) for a full-image box: per output index the first source index, the number of
    taps and the normalised double coefficients; the support is the filter's, scaled by in / out when reducing (antialiasing)."""
    support, fn = _PIL_FILTERS[filt]
    fn = fn or _bicubic
    scale = float(in_size) / float(out_size)
    filterscale = max(scale, 1.0)
    support = support * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.float64)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = fn((np.arange(xmax) + xmin - center + 0.5) * ss)
        ww = w.sum()
        if ww != 0.0:
            w = w / ww
        kk[xx, :xmax] = w
        bounds[xx] = (xmin, xmax)
    return bounds, kk, ksize


_PIL_CACHE = {}


def preprocess_pil(x, dtype, filt, quantize=True, size=299):
    """ops.quantize_images + resize_images with a PIL resizer (reference utils/resize.py:39-78: 'clean' = bicubic, 'friendly' = bilinear for
    InceptionV3_tf) + normalise, on the device."""
    x = x.float().contiguous()
    N, Cc, H, W = x.shape
    key = (H, W, size, filt, str(x.device))
    if key not in _PIL_CACHE:
        bh, kh, nh = pil_coeffs(W, size, filt)
        bv, kv, nv = pil_coeffs(H, size, filt)
        to = lambda a: torch.from_numpy(a).to(x.device).contiguous()
        _PIL_CACHE[key] = (to(bh), to(kh), nh, to(bv), to(kv), nv)
    bh, kh, nh, bv, kv, nv = _PIL_CACHE[key]
    tmp = torch.empty((N, Cc, H, size), dtype=torch.float32, device=x.device)
    out = torch.empty((N, size, size, Cc), dtype=dtype, device=x.device)
    L.call("sg_pil_resize_normalize", L.dt(dtype), L.ptr(x), L.ptr(out), L.ptr(tmp), N, Cc, H, W, size, size, L.ptr(bh), L.ptr(kh), nh, L.ptr(bv), L.ptr(kv), nv,
           1 if quantize else 0, L.stream())
    return out


class LoadEvalModel:
    """reference src/metrics/preparation.py:43-122 for eval_backbone == "InceptionV3_tf" with the post-resizers "legacy" (torch bilinear),
    "clean" (PIL bicubic) and "friendly" (PIL bilinear), reference src/utils/resize.py:49-69."""

    def __init__(self, eval_backbone="InceptionV3_tf", post_resizer="legacy", world_size=1, distributed_data_parallel=False, device="cuda",
                 state_dict=None, dtype=torch.float32, weights_path=None, f32_mode="exact"):
        """state_dict: tensors under torchvision's inception_v3 names, checked against inception_manifest() like the reference's strict load
        (anything else raises); weights_path: the published file itself, additionally checked against the sha256 prefix in its name.
        `self.weights_pinned` says which: True only for a hash-verified file -- FID / IS values from any other weights (the seeded random
        ones of bench.py and the tests) are NOT comparable with published numbers."""
        if eval_backbone != "InceptionV3_tf" or post_resizer not in ("legacy", "clean", "friendly"):
            raise NotImplementedError("InceptionV3_tf with the legacy / clean / friendly resizers is on the hot path (SURVEY §2)")
        self.weights_pinned, self.weights_sha256 = False, None
        if weights_path is not None:
            state_dict, self.weights_sha256 = load_fid_weights(weights_path)
            self.weights_pinned = True
        if state_dict is None:
            raise RuntimeError("pass the FID Inception weights (weights_path=.../" + FID_WEIGHTS_FILE + ", from " + FID_WEIGHTS_URL +
                               ", or its state_dict); there is no network access here")
        validate_inception_state_dict(state_dict)
        self.eval_backbone, self.post_resizer, self.device = eval_backbone, post_resizer, torch.device(device)
        self.res = 299
        self.model = InceptionV3(state_dict, self.device, dtype, f32_mode=f32_mode)      # f32_mode: InceptionV3.__init__
        self.dtype = dtype

    def eval(self):
        pass

    @torch.no_grad()
    def get_outputs(self, x, quantize=False):
        if self.post_resizer == "legacy":
            return self.model.forward_nhwc(preprocess(x, self.dtype, quantize, self.res))
        return self.model.forward_nhwc(preprocess_pil(x, self.dtype, "bicubic" if self.post_resizer == "clean" else "bilinear", quantize, self.res))


def softmax_rows(logits):
    p = torch.empty_like(logits)
    L.call("sg_softmax_rows", L.F32, L.ptr(logits), L.ptr(p), logits.shape[0], logits.shape[1], L.stream())
    return p


@torch.no_grad()
def generate_images_and_stack_features(generator, eval_model, num_generate, batch_size, z_dim, num_classes, quantize=True, world_size=1,
                                       DDP=False, device="cuda", moments=None, z_prior="gaussian", truncation_factor=-1.0, MODEL=None, latent_opt=None, langevin=None):
    """reference src/metrics/features.py:17-65. Returns (features [n,2048], probs [n,1008], labels list).
    moments: optional `FeatureMoments` accumulator fed on the device. It receives exactly the rows the reference keeps
    (`fake_feats[:num_generate]` of the rank-major gathered stack, src/metrics/fid.py:68-69): the over-generated tail --
    ceil(num_generate / batch) batches, `num_batches // world_size + 1` per rank under DDP -- is NOT accumulated.
    z_prior / truncation_factor / MODEL (InfoGAN codes behind z): the reference's evaluation-time sampling (src/utils/sample.py:90-118 with is_train=False).
    latent_opt: LOGAN at evaluation time (src/utils/sample.py:96,123-135 with LOSS.lo_steps4eval): dict(discriminator=, lo_rate=, lo_steps=, lo_alpha=, lo_beta=) -- the
    latents take their step along d D(G(z)) / dz (the one place of this function that runs with gradients enabled) before the images are generated.
    langevin: RUN.langevin_sampling (src/utils/sample.py:136-148,195-216): dict(discriminator=, langevin_rate=, langevin_noise_std=, langevin_decay=, langevin_decay_steps=,
    langevin_steps=) -- losses.langevin_sampling on the drawn latents."""
    from .worker import sample_zy, sample_latents
    plain = z_prior == "gaussian" and truncation_factor == -1.0 and getattr(MODEL, "info_type", "N/A") == "N/A" and latent_opt is None and langevin is None
    num_batches = int(math.ceil(float(num_generate) / float(batch_size)))
    rank = 0
    if DDP:
        num_batches = num_batches // world_size + 1
        if world_size > 1:
            import torch.distributed as dist
            rank = dist.get_rank()
    # rows of this rank sit at [rank * per_rank, (rank + 1) * per_rank) of the gathered stack; the first num_generate survive
    per_rank = num_batches * batch_size
    keep = max(0, min(per_rank, num_generate - rank * per_rank))
    feats, probs, labels = [], [], []
    for b in range(num_batches):
        if plain:
            zs, ys = sample_zy(batch_size, z_dim, num_classes, device)
        else:
            zs, ys = sample_latents(batch_size, z_dim, num_classes, device, z_prior=z_prior, truncation_factor=truncation_factor, MODEL=MODEL)
        if latent_opt is not None:
            from .losses import latent_optimise
            with torch.enable_grad():
                zs, _ = latent_optimise(zs=zs, fake_labels=ys, generator=generator, discriminator=latent_opt["discriminator"], batch_size=zs.shape[0],
                                        lo_rate=latent_opt["lo_rate"], lo_steps=latent_opt["lo_steps"], lo_alpha=latent_opt["lo_alpha"], lo_beta=latent_opt["lo_beta"],
                                        eval=True, cal_trsp_cost=False, device=device)
            zs = zs.detach()
        if langevin is not None:          # src/utils/sample.py:136-148 (RUN.langevin_sampling; exclusive with latent optimisation, src/config.py:650-651)
            from .losses import langevin_sampling
            with torch.enable_grad():
                zs = langevin_sampling(zs=zs, z_dim=z_dim, fake_labels=ys, generator=generator, discriminator=langevin["discriminator"], batch_size=zs.shape[0],
                                       langevin_rate=langevin["langevin_rate"], langevin_noise_std=langevin["langevin_noise_std"], langevin_decay=langevin["langevin_decay"],
                                       langevin_decay_steps=langevin["langevin_decay_steps"], langevin_steps=langevin["langevin_steps"], device=device)
            zs = zs.detach()
        fake = generator(zs, ys, eval=True)
        f, logit = eval_model.get_outputs(fake, quantize=quantize)
        if moments is not None:
            take = max(0, min(batch_size, keep - b * batch_size))
            if take:
                moments.add(f[:take])
        feats.append(f)
        probs.append(softmax_rows(logit))
        labels.append(ys)
    feats, probs, labels = torch.cat(feats, 0), torch.cat(probs, 0), torch.cat(labels, 0)
    if DDP and world_size > 1:
        import torch.distributed as dist
        def gather(t):
            out = [torch.zeros_like(t) for _ in range(world_size)]
            dist.all_gather(out, t)
            return torch.cat(out, 0)
        feats, probs, labels = gather(feats), gather(probs), gather(labels)
    return feats, probs, list(labels.detach().cpu().numpy())


class FeatureMoments:
    """sum f and sum f f^T in fp64 on the device (replaces np.mean / np.cov over gathered features, fid.py:96-97); in
    data-parallel runs the two accumulators are all-reduced instead of gathering 50k x 2048 floats per rank."""

    def __init__(self, dim, device):
        self.dim, self.n = dim, 0
        self.s1 = torch.zeros(dim, dtype=torch.float64, device=device)
        self.s2 = torch.zeros(dim, dim, dtype=torch.float64, device=device)

    def add(self, f):
        f = f.float().contiguous()
        L.call("sg_feat_moments_accumulate", L.ptr(f), f.shape[0], self.dim, L.ptr(self.s1), L.ptr(self.s2), L.stream())
        self.n += f.shape[0]

    def finalize(self, group=None):
        n = self.n
        if group is not None:
            import torch.distributed as dist
            dist.all_reduce(self.s1, group=group); dist.all_reduce(self.s2, group=group)
            t = torch.tensor([n], dtype=torch.float64, device=self.s1.device); dist.all_reduce(t, group=group); n = int(t.item())
        mu = (self.s1 / n).cpu().numpy()
        s2 = self.s2.cpu().numpy()
        sigma = (s2 - n * np.outer(mu, mu)) / (n - 1)      # np.cov(rowvar=False): unbiased
        return mu, sigma


def calculate_moments(feats, num_generate=None):
    f = feats[:num_generate] if num_generate else feats
    m = FeatureMoments(f.shape[1], f.device)
    for i in range(0, f.shape[0], 4096):
        m.add(f[i:i + 4096])
    return m.finalize()


def frechet_inception_distance(mu1, sigma1, mu2, sigma2, eps=1e-6):
    """reference src/metrics/fid.py:34-62 (host fp64, scipy.linalg.sqrtm)."""
    from scipy import linalg
    mu1, mu2 = np.atleast_1d(mu1), np.atleast_1d(mu2)
    sigma1, sigma2 = np.atleast_2d(sigma1), np.atleast_2d(sigma2)
    diff = mu1 - mu2
    covmean, _ = linalg.sqrtm(sigma1.dot(sigma2), disp=False)
    if not np.isfinite(covmean).all():
        offset = np.eye(sigma1.shape[0]) * eps
        covmean = linalg.sqrtm((sigma1 + offset).dot(sigma2 + offset))
    if np.iscomplexobj(covmean):
        covmean = covmean.real
    return diff.dot(diff) + np.trace(sigma1) + np.trace(sigma2) - 2 * np.trace(covmean)


def frechet_inception_distance_device(mu1, sigma1, mu2, sigma2, device=None, max_sweeps=40, tol=1e-12):
    """The same distance with tr sqrtm(S1 S2) computed on the GPU in fp64 (csrc/linalg.hip): Cholesky factors, B = L2^T L1, one-sided
    Jacobi sweeps until the rows of B are orthogonal to `tol`, sum of their norms (= sum of sqrt of the eigenvalues of S1 S2).
    When a covariance is not positive definite (fewer samples than dimensions) the host formula above -- the reference's own route,
    eps branch included -- is used instead. Returns a python float."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    mu1, mu2 = np.atleast_1d(np.asarray(mu1, dtype=np.float64)), np.atleast_1d(np.asarray(mu2, dtype=np.float64))
    s1, s2 = np.atleast_2d(np.asarray(sigma1, dtype=np.float64)), np.atleast_2d(np.asarray(sigma2, dtype=np.float64))
    n = s1.shape[0]
    ne = n + (n & 1)                                    # the tournament pairs rows: pad an odd dimension with one zero row / column of B
    L1 = torch.from_numpy(s1).to(dev).contiguous().clone()
    L2 = torch.from_numpy(s2).to(dev).contiguous().clone()
    flags = torch.zeros(2, dtype=torch.int32, device=dev)
    L.call("sg_chol_lower", L.ptr(L1), n, L.ptr(flags), L.stream())
    L.call("sg_chol_lower", L.ptr(L2), n, L.ptr(flags) + 4, L.stream())
    if int(flags.abs().sum().item()) != 0:
        return float(frechet_inception_distance(mu1, s1, mu2, s2))
    B = torch.empty((n, n), dtype=torch.float64, device=dev)
    L.call("sg_dgemm_tn", L.ptr(L2), L.ptr(L1), L.ptr(B), n, L.stream())
    if ne != n:
        Bp = torch.zeros((ne, ne), dtype=torch.float64, device=dev)
        Bp[:n, :n] = B
        B = Bp
    offd = torch.zeros(1, dtype=torch.float64, device=dev)
    for _ in range(max_sweeps):
        L.call("sg_jacobi_sweep", L.ptr(B), ne, L.ptr(offd), L.stream())
        if float(offd.item()) < tol:
            break
    else:
        if float(offd.item()) > 1e-8:       # (the sum of the singular values is second-order accurate in this measure)
            raise RuntimeError(f"one-sided Jacobi did not converge in {max_sweeps} sweeps (off-diagonal measure {float(offd.item()):.2e})")
    nuc = torch.zeros(1, dtype=torch.float64, device=dev)
    L.call("sg_row_norm_sum", L.ptr(B), ne, L.ptr(nuc), L.stream())
    diff = mu1 - mu2
    return float(diff.dot(diff) + np.trace(s1) + np.trace(s2) - 2.0 * float(nuc.item()))


def calculate_kl_div(ps, splits):
    """reference src/metrics/ins.py:28-42 (device tensors in, numpy scalars out)."""
    scores = []
    n = ps.shape[0]
    with torch.no_grad():
        for j in range(splits):
            part = ps[(j * n // splits):((j + 1) * n // splits), :]
            kl = part * (torch.log(part) - torch.log(torch.unsqueeze(torch.mean(part, 0), 0)))
            scores.append(torch.exp(torch.mean(torch.sum(kl, 1))).unsqueeze(0))
        scores = torch.cat(scores, 0)
        return torch.mean(scores).detach().cpu().numpy(), torch.std(scores).detach().cpu().numpy()


def top_k_accuracy(probs, labels, k, c0=1, ncls=1000):
    """Top-k accuracy over probs[:, c0:c0+ncls] with sklearn's tie rule, decided on the device (bit-exact integer
    logic; reference ins.py:75-76 slices classes 1..1000 and offsets the TF label by +1 -> pass 0-based class here)."""
    probs = probs.float().contiguous()
    lab = torch.as_tensor(labels, device=probs.device).long().contiguous()
    hits = torch.empty(probs.shape[0], dtype=torch.uint8, device=probs.device)
    L.call("sg_topk_hits", L.ptr(probs) + 4 * c0, probs.shape[1], ncls, L.ptr(lab), k, probs.shape[0], L.ptr(hits), L.stream())
    return float(hits.float().mean().item())


def load_ImageNet_label_dict(label_table_path):
    """reference src/utils/misc.py:582-595 (TF-Inception branch): `folder -> line index` of the reference's
    `src/utils/tf_imagenet_folder_label_pairs.txt` (1000 lines `n02119789 1 kit_fox`; the file stays in the StudioGAN checkout -- pass
    its path)."""
    d, label = {}, 0
    with open(label_table_path, "r") as f:
        for line in f:
            if not line.strip():
                continue
            d[line.split(" ")[0]] = label
            label += 1
    return d


def convert_labels(labels, class_to_idx, folder_label_dict):
    """loader label -> folder name -> TF-Inception class index (reference src/metrics/ins.py:48-49,68-70)."""
    loader_label_folder_dict = {v: k for k, v in class_to_idx.items()}
    return [folder_label_dict[loader_label_folder_dict[int(l)]] for l in labels]


def eval_features(probs, labels, num_features, split, is_acc, class_to_idx=None, folder_label_dict=None, topk_fn=None):
    """reference src/metrics/ins.py:45-79 for the TF-Inception backbone on ImageNet: IS over the first num_features rows; top-1 / top-5 of
    the remapped labels against probs[:, 1:1001] (the reference passes `[i + 1 for i in converted]` with that slice to sklearn, which
    maps label i + 1 to column i: the same decision as testing 0-based class i here). The accuracy itself runs on the device
    (`top_k_accuracy`: sklearn's tie rule, bit-exact)."""
    topk_fn = topk_fn or top_k_accuracy
    probs, labels = probs[:num_features], labels[:num_features]
    m_scores, m_std = calculate_kl_div(probs, splits=split)
    top1, top5 = "N/A", "N/A"
    if is_acc:
        converted = convert_labels(labels, class_to_idx, folder_label_dict)
        top1 = topk_fn(probs, converted, 1, 1, 1000)
        top5 = topk_fn(probs, converted, 5, 1, 1000)
    return m_scores, m_std, top1, top5


# ---------------------------------------------------------------------------------------------------------
# precision / recall / density / coverage on the device (reference src/metrics/prdc.py:87-168)
# ---------------------------------------------------------------------------------------------------------
def _sq_dist_block(xb, y, y_sq, out):
    """out[r][c] = |y_c|^2 - 2 x_r . y_c for a row block xb of X (exact-fp32 MFMA GEMM; |x_r|^2 is added by the consumer kernels)."""
    rows, C = xb.shape
    F.gemm_raw(L.F32, y, 0, C, xb, 0, C, out, y.shape[0], y.shape[0], rows, C, bias=y_sq, alpha=-2.0)
    return out


def _kth_nn_sq(feats, k, block=4096):
    """squared distance of every row to its k-th nearest neighbour, the row itself included (prdc.py:128-140 with k = nearest_k + 1)."""
    n, C = feats.shape
    sq = torch.empty(n, dtype=torch.float32, device=feats.device)
    L.call("sg_row_sqnorm", L.ptr(feats), n, C, L.ptr(sq), L.stream())
    out = torch.empty(n, dtype=torch.float32, device=feats.device)
    buf = torch.empty((min(block, n), n), dtype=torch.float32, device=feats.device)
    for r0 in range(0, n, block):
        xb = feats[r0:r0 + block]
        d = _sq_dist_block(xb, feats, sq, buf[:xb.shape[0]])
        L.call("sg_kth_smallest_rows", L.ptr(d), n, xb.shape[0], n, k, L.ptr(sq[r0:r0 + block]), L.ptr(out[r0:r0 + block]), L.stream())
    return out, sq


@torch.no_grad()
def compute_prdc(real_features, fake_features, nearest_k, block=4096):
    """reference src/metrics/prdc.py:143-168 `compute_prdc`: dict(precision, recall, density, coverage). Features: [N, dim] tensors
    (any float dtype / device; computed in fp32 on the GPU). Three blocked fp32 GEMMs + streaming row reductions replace
    sklearn.metrics.pairwise_distances(n_jobs=8) + np.argpartition over N x N float64 matrices; decisions are taken on squared
    distances (monotone), the manifold radii are the (nearest_k + 1)-th smallest self-inclusive distances exactly as in the reference."""
    assert 1 <= nearest_k < 16, "nearest_k + 1 <= 16 neighbours are kept per lane"
    dev = torch.device("cuda", torch.cuda.current_device()) if not (torch.is_tensor(real_features) and real_features.is_cuda) else real_features.device
    real = torch.as_tensor(real_features).to(device=dev, dtype=torch.float32).contiguous()
    fake = torch.as_tensor(fake_features).to(device=dev, dtype=torch.float32).contiguous()
    nr, nf = real.shape[0], fake.shape[0]
    r2_real, sq_real = _kth_nn_sq(real, nearest_k + 1, block)
    r2_fake, sq_fake = _kth_nn_sq(fake, nearest_k + 1, block)
    col_cnt = torch.zeros(nf, dtype=torch.int32, device=dev)
    row_any = torch.empty(nr, dtype=torch.uint8, device=dev)
    row_min = torch.empty(nr, dtype=torch.float32, device=dev)
    buf = torch.empty((min(block, nr), nf), dtype=torch.float32, device=dev)
    for r0 in range(0, nr, block):
        xb = real[r0:r0 + block]
        d = _sq_dist_block(xb, fake, sq_fake, buf[:xb.shape[0]])
        L.call("sg_prdc_rows", L.ptr(d), nf, xb.shape[0], nf, L.ptr(sq_real[r0:r0 + block]), L.ptr(r2_real[r0:r0 + block]), L.ptr(r2_fake), L.ptr(col_cnt),
               L.ptr(row_any[r0:r0 + block]), L.ptr(row_min[r0:r0 + block]), L.stream())
    cnt = col_cnt.double()
    precision = float((cnt > 0).double().mean())
    recall = float(row_any.double().mean())
    density = float(cnt.mean() / float(nearest_k))
    coverage = float((row_min < r2_real).double().mean())
    return dict(precision=precision, recall=recall, density=density, coverage=coverage)


def calculate_pr_dc(real_feats, fake_feats, num_generate, nearest_k):
    """reference src/metrics/prdc.py:66-84 with the features already extracted (the fake stack is truncated to num_generate first)."""
    m = compute_prdc(real_feats, fake_feats[:num_generate], nearest_k)
    return m["precision"], m["recall"], m["density"], m["coverage"]
