"""Kernel-level spectral-norm checks (test infrastructure; shared by tests/test_sn_gpu.py and the interpreter-backed CPU test): sg_sn_forward / sg_sn_backward called
through the C ABI on hand-filled descriptor tables, against torch.nn.utils.spectral_norm evaluated in fp64 -- what reference src/utils/ops.py:195-224 wraps around
Conv2d / ConvTranspose2d / Linear / Embedding with eps = 1e-6."""
import ctypes

import torch
import torch.nn as nn

SPLITS, SNB_BLOCKS = 16, 512      # csrc/sn.hip SN_SPLITS / SNB_BLOCKS (bank.WeightBank sizes the workspace with the same numbers)

# (kind, rows(Cout), Cin, R, options)
LAYERS = [
    ("conv", 96, 48, 3, {}),                                 # a block convolution: vector paths of every kernel
    ("conv", 40, 20, 3, {}),                                 # cols % 4 == 0, Cin % 8 != 0: scalar image path, tile backward
    ("conv", 96, 3, 3, {"cin_pad": 8}),                      # the discriminator's RGB stem: 8-channel operand image, 3 real channels
    ("conv", 3, 96, 3, {"rows_pad": 8}),                     # the generator's RGB layer: 8 output rows, 3 real
    ("conv", 64, 96, 1, {}),                                 # 1x1 (attention theta / phi / g, skip connections)
    ("conv", 33, 7, 4, {"noflip": True}),                    # DCGAN discriminator 4x4 stride 2: odd sizes, un-flipped data-gradient image, RS = 16
    ("deconv", 24, 16, 4, {"noflip": True}),                 # ConvTranspose2d: weight [Cin][Cout][4][4], spectral norm over dim 1
    ("linear", 1000, 20, 1, {}),                             # linear0-like [rows x 20]
    ("linear", 1, 1536, 1, {}),                              # linear1: one row
    ("embedding", 1000, 64, 1, {}),                          # projection table
    ("conv", 32, 16, 3, {"apply_sn": False}),                # plain layer: images with sigma = 1
    ("conv", 48, 32, 3, {"power_iter": False}),              # eval mode: sigma from the stored u, v, no iteration
    ("conv", 8, 1536, 3, {}),                                # 13824 columns: the k_sn_wtu column tiling and the 8-deep v copy
]


def _align(n, a=4):
    return (n + a - 1) // a * a


def reference(kind, w64, u64, v64, training, eps=1e-6):
    """torch.nn.utils.spectral_norm on a fp64 module carrying (w, u, v) -> (W_sn with graph to weight_orig, module)"""
    rows = u64.numel()
    if kind == "conv":
        m = nn.Conv2d(w64.shape[1], w64.shape[0], w64.shape[2], bias=False)
    elif kind == "deconv":
        m = nn.ConvTranspose2d(w64.shape[0], w64.shape[1], w64.shape[2], bias=False)
    elif kind == "linear":
        m = nn.Linear(w64.shape[1], w64.shape[0], bias=False)
    else:
        m = nn.Embedding(w64.shape[0], w64.shape[1])
    m = m.double()
    with torch.no_grad():
        m.weight.copy_(w64)
    m = nn.utils.spectral_norm(m, eps=eps)
    with torch.no_grad():
        m.weight_u.copy_(u64)
        m.weight_v.copy_(v64)
    m.train(training)
    # (the hook recomputes `weight` in the forward pre-hook: call it directly instead of running a convolution)
    hook = next(h for h in m._forward_pre_hooks.values() if type(h).__name__ == "SpectralNorm")
    hook(m, None)
    assert m.weight_u.numel() == rows
    return m.weight, m


def run(dev, dtype, L, call, ptr, stream, seed=0):
    """-> list of (name, error, bound) rows; L = studiogan_amd._lib"""
    g = torch.Generator().manual_seed(seed)
    es = 2 if dtype == torch.bfloat16 else 4
    n = len(LAYERS)
    recs, work_floats = [], 0
    for kind, rows, cin, R, opt in LAYERS:
        RS = R * R
        cols = cin * RS
        if kind == "deconv":
            w = torch.randn(cin, rows, R, R, generator=g) * 0.1
        elif kind == "conv":
            w = torch.randn(rows, cin, R, R, generator=g) * 0.1
        else:
            w = torch.randn(rows, cols, generator=g) * 0.1
        u = nn.functional.normalize(torch.randn(rows, generator=g), dim=0, eps=1e-6)
        v = nn.functional.normalize(torch.randn(cols, generator=g), dim=0, eps=1e-6)
        r = dict(kind=kind, rows=rows, cin=cin, RS=RS, cols=cols, w=w, u=u, v=v, opt=opt, cin_pad=opt.get("cin_pad", cin), rows_pad=opt.get("rows_pad", rows),
                 apply_sn=opt.get("apply_sn", True), pi=opt.get("power_iter", True), trans=kind == "deconv", noflip=opt.get("noflip", False),
                 conv=kind in ("conv", "deconv"), work_off=work_floats)
        work_floats += SPLITS * cols + rows
        recs.append(r)
    work = torch.zeros(max(work_floats, SNB_BLOCKS * n) + 64, device=dev)
    arr = (L.SnLayer * n)()
    keep = []
    for i, r in enumerate(recs):
        d = arr[i]
        for k in ("w", "u", "v"):
            r[k + "_d"] = r[k].clone().to(dev).contiguous()          # (a copy on the CPU interpreter too: the kernel updates u, v in place)
        r["sigma_d"] = torch.full((1,), -7.0, device=dev)
        r["us_d"], r["vs_d"] = torch.zeros(r["rows"], device=dev), torch.zeros(r["cols"], device=dev)
        ro = max(r["rows"], r["rows_pad"])
        img = ro * r["RS"] * r["cin_pad"]
        r["fwd_d"] = torch.full((_align(img, 16),), 3.0, device=dev, dtype=dtype) if r["conv"] else None
        r["dg_d"] = torch.zeros(_align(img, 16), device=dev, dtype=dtype) if r["conv"] else None       # (the bank zero-fills its slots once: padding rows / channels stay zero)
        r["f32_d"] = torch.full((r["rows"] * r["cols"],), 3.0, device=dev) if not r["conv"] else None
        d.w, d.u, d.v, d.sigma = ptr(r["w_d"]), ptr(r["u_d"]), ptr(r["v_d"]), ptr(r["sigma_d"])
        d.u_snap, d.v_snap = ptr(r["us_d"]), ptr(r["vs_d"])
        d.w_fwd, d.w_dgrad, d.w_f32 = ptr(r["fwd_d"]), ptr(r["dg_d"]), ptr(r["f32_d"])
        d.rows, d.cols, d.Cin, d.RS = r["rows"], r["cols"], r["cin"], r["RS"]
        d.do_power_iter, d.apply_sn, d.rows_pad, d.work_off = int(r["pi"]), int(r["apply_sn"]), r["rows_pad"], r["work_off"]
        d.trans, d.dgrad_noflip, d.Cin_pad = int(r["trans"]), int(r["noflip"]), r["cin_pad"]
    tab = torch.frombuffer(bytearray(arr), dtype=torch.uint8).to(dev)
    call("sg_sn_forward", L.dt(dtype), ptr(tab), arr, n, 1e-6, ptr(work), work.numel(), stream())
    rows_out = []
    img_tol = 4e-3 if dtype == torch.bfloat16 else 2e-6          # one bf16 rounding of W / sigma (2^-9 relative) resp. fp32 rounding

    def rel(a, b):
        a, b = a.detach().double().cpu().reshape(-1), b.detach().double().reshape(-1)
        return float((a - b).abs().max() / max(float(b.abs().max()), 1e-30))
    for i, r in enumerate(recs):
        tag = f"{i}:{r['kind']} {r['rows']}x{r['cin']}x{r['RS']}" + "".join(f" {k}" for k in r["opt"])
        if r["apply_sn"]:
            wsn, m = reference(r["kind"], r["w"].double(), r["u"].double(), r["v"].double(), r["pi"])
            sigma_ref = float((r["w"].double().reshape(-1)[0] / wsn.detach().reshape(-1)[0]))
            rows_out.append((tag + " u", rel(r["u_d"], m.weight_u), 2e-6))
            rows_out.append((tag + " v", rel(r["v_d"], m.weight_v), 2e-6))
            rows_out.append((tag + " sigma", abs(float(r["sigma_d"]) - sigma_ref) / sigma_ref, 2e-6))
            rows_out.append((tag + " u_snap", rel(r["us_d"], m.weight_u), 2e-6))
            rows_out.append((tag + " v_snap", rel(r["vs_d"], m.weight_v), 2e-6))
        else:
            wsn, m = r["w"].double().clone().requires_grad_(True), None
            rows_out.append((tag + " sigma", abs(float(r["sigma_d"]) - 1.0), 0.0))
        r["wsn"], r["m"] = wsn, m
        W = wsn.detach()
        if r["conv"]:
            Wm = W.permute(1, 0, 2, 3) if r["trans"] else W                  # -> [Cout][Cin][R][S]
            R_ = Wm.shape[2]
            ro = max(r["rows"], r["rows_pad"])
            fwd = torch.zeros(ro, R_, R_, r["cin_pad"], dtype=torch.float64)
            fwd[:r["rows"], :, :, :r["cin"]] = Wm.permute(0, 2, 3, 1)
            Wd = Wm if r["noflip"] else Wm.flip(2, 3)
            dgr = torch.zeros(r["cin"], R_, R_, ro, dtype=torch.float64)
            dgr[:, :, :, :r["rows"]] = Wd.permute(1, 2, 3, 0)
            rows_out.append((tag + " w_fwd image", rel(r["fwd_d"][:fwd.numel()].float(), fwd), img_tol))
            rows_out.append((tag + " w_dgrad image", rel(r["dg_d"][:dgr.numel()].float(), dgr), img_tol))
        else:
            rows_out.append((tag + " w_f32", rel(r["f32_d"], W), 2e-6))
    # ---- backward: dW_orig += d <G, W_sn> / d weight_orig, G handed over in the layout the weight-gradient kernels write --------------------------
    barr = (L.SnBwdLayer * n)()
    for i, r in enumerate(recs):
        G = torch.randn(r["wsn"].shape, generator=g, dtype=torch.float64)
        leaf = r["m"].weight_orig if r["m"] is not None else r["wsn"]
        (gref,) = torch.autograd.grad((r["wsn"] * G).sum(), leaf)
        prior = torch.randn(leaf.shape, generator=g)                          # the kernel accumulates (+=)
        r["gref"] = gref + prior.double()
        if r["conv"]:
            Gm = G.permute(1, 0, 2, 3) if r["trans"] else G                   # [Cout][Cin][R][S]
            if r["trans"]:
                lay, natural = Gm.permute(1, 2, 3, 0).contiguous(), 2         # [Cin][R][S][Cout]
            else:
                lay = torch.zeros(r["rows"], Gm.shape[2], Gm.shape[3], r["cin_pad"], dtype=torch.float64)
                lay[..., :r["cin"]] = Gm.permute(0, 2, 3, 1)
                lay[..., r["cin"]:] = 123.0                                   # padding channels carry garbage the kernel must not read into the result
                natural = 0
        else:
            lay, natural = G, 1
        r["dwt_d"], r["dw_d"] = lay.float().to(dev).contiguous(), prior.to(dev).contiguous()
        b = barr[i]
        b.dwt, b.w, b.u, b.v, b.sigma, b.dw = ptr(r["dwt_d"]), ptr(r["w_d"]), ptr(r["us_d"]), ptr(r["vs_d"]), ptr(r["sigma_d"]), ptr(r["dw_d"])
        b.rows, b.cols, b.Cin, b.RS, b.natural, b.apply_sn, b.trans, b.Cin_pad = r["rows"], r["cols"], r["cin"], r["RS"], natural, int(r["apply_sn"]), int(r["trans"]), r["cin_pad"]
        keep.append(lay)
    btab = torch.frombuffer(bytearray(barr), dtype=torch.uint8).to(dev)
    call("sg_sn_backward", ptr(btab), barr, n, ptr(work), work.numel(), stream())
    if dev.type == "cuda":
        torch.cuda.synchronize()
    for i, r in enumerate(recs):
        tag = f"{i}:{r['kind']} {r['rows']}x{r['cin']}x{r['RS']}" + "".join(f" {k}" for k in r["opt"])
        rows_out.append((tag + " dW_orig", rel(r["dw_d"], r["gref"]), 1e-5))
    return rows_out


def forward_table(dev, dtype, L, call, ptr, stream, shapes, seed, split_at=None):
    """sg_sn_forward on a table of plain convolution layers `shapes` = [(rows, cin, R)]; split_at = None: ONE call; an index: two calls ([0, split_at), [split_at, n)) on the
    same table. -> per layer (u, v, sigma, forward image, data-gradient image) as CPU tensors."""
    g = torch.Generator().manual_seed(seed)
    n = len(shapes)
    arr = (L.SnLayer * n)()
    recs, work_floats = [], 0
    for i, (rows, cin, R) in enumerate(shapes):
        RS, cols = R * R, cin * R * R
        r = dict(w=(torch.randn(rows, cin, R, R, generator=g) * 0.1).to(dev), u=nn.functional.normalize(torch.randn(rows, generator=g), dim=0).to(dev),
                 v=nn.functional.normalize(torch.randn(cols, generator=g), dim=0).to(dev), sigma=torch.zeros(1, device=dev),
                 us=torch.zeros(rows, device=dev), vs=torch.zeros(cols, device=dev),
                 fwd=torch.zeros(_align(rows * cols, 16), device=dev, dtype=dtype), dg=torch.zeros(_align(rows * cols, 16), device=dev, dtype=dtype))
        d = arr[i]
        d.w, d.u, d.v, d.sigma, d.u_snap, d.v_snap, d.w_fwd, d.w_dgrad = ptr(r["w"]), ptr(r["u"]), ptr(r["v"]), ptr(r["sigma"]), ptr(r["us"]), ptr(r["vs"]), ptr(r["fwd"]), ptr(r["dg"])
        d.rows, d.cols, d.Cin, d.RS = rows, cols, cin, RS
        d.do_power_iter, d.apply_sn, d.rows_pad, d.work_off, d.trans, d.dgrad_noflip, d.Cin_pad = 1, 1, rows, work_floats, 0, 0, cin
        work_floats += SPLITS * cols + rows
        recs.append(r)
    work = torch.zeros(work_floats + 64, device=dev)
    tab = torch.frombuffer(bytearray(arr), dtype=torch.uint8).to(dev)
    esz = ctypes.sizeof(L.SnLayer)
    for first, count in ([(0, n)] if split_at is None else [(0, split_at), (split_at, n - split_at)]):
        call("sg_sn_forward", L.dt(dtype), tab.data_ptr() + first * esz, ctypes.cast(ctypes.addressof(arr) + first * esz, ctypes.POINTER(L.SnLayer)), count, 1e-6, ptr(work), work.numel(), stream())
    return [tuple(r[k].detach().float().cpu().clone() for k in ("u", "v", "sigma", "fwd", "dg")) for r in recs]
