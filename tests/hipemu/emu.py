"""hipemu driver (TEST INFRASTRUCTURE): builds a translation unit of pytorch-studiogan_amd/csrc for the host against tests/hipemu/include and loads it with
ctypes. The library exports the SAME C entry points as libsgamd.so for that translation unit (e.g. sg_conv2d_wgrad from conv_wgrad.hip), so a test
fills the same descriptor the product fills and gets the result the kernel's own index arithmetic produces -- on numpy arrays, without a GPU."""
import ctypes as C
import hashlib
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
BUILD = os.path.join(HERE, "_build")
SRC = os.path.join(BUILD, "src")
CXX = "/opt/rocm/lib/llvm/bin/clang++"
sys.path.insert(0, HERE)
import translate  # noqa: E402

_vp, _i, _f, _ll = C.c_void_p, C.c_int, C.c_float, C.c_longlong

# field lists: include/sgamd.h (tests/test_host_cpu.py checks the product's copies of these against the header; test_hipemu_cpu.py checks these against the product's)
WGRAD_FIELDS = [("dtype", _i), ("N", _i), ("xHs", _i), ("xWs", _i), ("C", _i), ("ldx", _i), ("x_flags", _i), ("gHs", _i), ("gWs", _i),
                ("Cout", _i), ("ldg", _i), ("g_flags", _i), ("Ho", _i), ("Wo", _i), ("R", _i), ("S", _i), ("stride", _i),
                ("pad_h", _i), ("pad_w", _i), ("alpha", _f), ("x", _vp), ("dy", _vp), ("dw", _vp), ("alpha_ptr", _vp),
                ("splits", _i), ("no_tr", _i), ("work", _vp), ("work_floats", _ll), ("dbias", _vp)]


class ConvWgradDesc(C.Structure):
    _fields_ = WGRAD_FIELDS


BF16, PIX_RELU, PIX_UPSAMPLE = 1, 1, 2


def available():
    return os.path.exists(CXX)


def build(unit, opt="-O1"):
    """ONE translation unit + stubs.cpp as its own small library (used where a test needs a private, modified copy of a unit); returns the path"""
    translate.translate_tree(SRC)
    deps = [os.path.join(SRC, f) for f in sorted(os.listdir(SRC)) if f.endswith(".h")] + [os.path.join(SRC, unit + ".hip"),
            os.path.join(HERE, "stubs.cpp"), os.path.join(HERE, "rt.cpp"), os.path.join(HERE, "include", "hip", "hip_runtime.h")]
    h = hashlib.sha256()
    for d in deps:
        with open(d, "rb") as f:
            h.update(f.read())
    h.update(opt.encode())
    lib = os.path.join(BUILD, "libemu_%s_%s.so" % (unit, h.hexdigest()[:12]))
    if not os.path.exists(lib):
        for old in os.listdir(BUILD):
            if old.startswith("libemu_%s_" % unit):
                os.remove(os.path.join(BUILD, old))
        common = [CXX, "-x", "c++", "-std=c++17", "-fPIC", "-ffp-contract=off", "-I" + os.path.join(HERE, "include"), "-I" + SRC,
                  "-Wno-unknown-attributes", "-Wno-unused-value"]
        rt = lib + ".rt.o"            # the interpreter's own loops (MFMA arithmetic): -O3 for the machine this runs on
        for cmd in (common + ["-O3", "-march=native", "-c", os.path.join(HERE, "rt.cpp"), "-o", rt],
                    common + [opt, "-c", os.path.join(SRC, unit + ".hip"), "-o", lib + ".o"],
                    common + [opt, "-c", os.path.join(HERE, "stubs.cpp"), "-o", lib + ".st.o"],
                    [CXX, "-shared", "-o", lib + ".tmp", lib + ".o", lib + ".st.o", rt]):
            r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
            if r.returncode != 0:
                raise RuntimeError("hipemu build of %s failed:\n%s" % (unit, r.stdout[-4000:]))
        for f in (rt, lib + ".o", lib + ".st.o"):
            os.remove(f)
        os.rename(lib + ".tmp", lib)
    return lib


_libs = {}


def load(unit=None):
    """the emulated library. One build serves every test: the WHOLE library (fullemu.build: all translation units in parallel, cached per unit), whose
    C ABI contains every unit's entry points; `unit` only documents which translation unit a test exercises."""
    if "full" not in _libs:
        import fullemu
        _libs["full"] = C.CDLL(fullemu.build())
        _libs["full"].sg_last_error.restype = C.c_char_p
    return _libs["full"]


def config(lib, dma_late=0, greedy=0, seed=0):
    lib.hipemu_config(int(dma_late), int(greedy), C.c_uint(seed))


def counters(lib):
    out = (C.c_long * 5)()
    lib.hipemu_counters(out)
    return dict(zip(("launches", "blocks", "mfma", "dma_ops", "tr_reads"), list(out)))


# ---- bf16 helpers (numpy has no bfloat16: uint16 bit patterns, round to nearest even like torch) -------------------------------------------------
def to_bf16(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    u = a.view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)
    return r


def from_bf16(b):
    return (b.astype(np.uint32) << 16).view(np.float32)


def ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def aligned(shape, dtype, align=64):
    n = int(np.prod(shape)) * np.dtype(dtype).itemsize
    raw = np.zeros(n + align, dtype=np.uint8)
    off = (-raw.ctypes.data) % align
    return raw[off:off + n].view(dtype).reshape(shape)


def conv_wgrad(lib, x, dy, Cout, x_flags=0, g_flags=0, alpha=1.0, bias=False, splits=0, env=None, R=3, pad=1):
    """x: uint16 bf16 [N][xHs][xWs][C], dy: uint16 bf16 [N][gHs][gWs][Cout]; 3x3 / stride 1 / pad 1. Returns (dw fp32 [Cout][3][3][C], dbias or None)
    through the translation unit's own sg_conv2d_wgrad_plan + sg_conv2d_wgrad (workspace form: the deterministic two-stage reduction)."""
    N, xHs, xWs, Cin = x.shape
    _, gHs, gWs, _ = dy.shape
    up = 2 if (x_flags & PIX_UPSAMPLE) else 1
    Ho, Wo = xHs * up, xWs * up
    gup = 2 if (g_flags & PIX_UPSAMPLE) else 1
    assert (gHs * gup, gWs * gup) == (Ho, Wo)
    xa = aligned(x.shape, np.uint16); xa[...] = x
    ga = aligned(dy.shape, np.uint16); ga[...] = dy
    dw = aligned((Cout, R, R, Cin), np.float32)
    db = aligned((Cout,), np.float32) if bias else None
    d = ConvWgradDesc(dtype=BF16, N=N, xHs=xHs, xWs=xWs, C=Cin, ldx=Cin, x_flags=x_flags, gHs=gHs, gWs=gWs, Cout=Cout, ldg=Cout, g_flags=g_flags,
                      Ho=Ho, Wo=Wo, R=R, S=R, stride=1, pad_h=pad, pad_w=pad, alpha=alpha, x=ptr(xa), dy=ptr(ga), dw=ptr(dw), alpha_ptr=None,
                      splits=splits, no_tr=0, work=None, work_floats=0, dbias=ptr(db) if bias else None)
    old = {}
    for k, v in (env or {}).items():
        old[k] = os.environ.get(k)
        os.environ[k] = v
    try:
        sp, wf = C.c_int(0), C.c_longlong(0)
        if lib.sg_conv2d_wgrad_plan(C.byref(d), C.byref(sp), C.byref(wf)) != 0:
            raise RuntimeError(lib.sg_last_error())
        work = aligned((max(int(wf.value), 1),), np.float32)
        d.work, d.work_floats, d.splits = ptr(work), wf.value, sp.value
        fused = bias and lib.sg_conv2d_wgrad_fuses_bias(C.byref(d)) == 1
        if bias and not fused:
            d.dbias = None
        if lib.sg_conv2d_wgrad(C.byref(d), None) != 0:
            raise RuntimeError(lib.sg_last_error())
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v
    return dw, (db if bias and fused else None), sp.value


def wgrad_ref(x, dy, x_flags=0, g_flags=0, alpha=1.0, R=3, pad=1):
    """fp64 restatement of include/sgamd.h's formula for the 3x3 / pad-1 weight gradient on bf16 inputs"""
    xf = from_bf16(x).astype(np.float64)
    gf = from_bf16(dy).astype(np.float64)
    if x_flags & PIX_UPSAMPLE:
        xf = xf.repeat(2, axis=1).repeat(2, axis=2)
    if x_flags & PIX_RELU:
        xf = np.maximum(xf, 0)
    if g_flags & PIX_UPSAMPLE:
        gf = gf.repeat(2, axis=1).repeat(2, axis=2)
    N, H, W, Cin = xf.shape
    xp = np.zeros((N, H + 2 * pad, W + 2 * pad, Cin))
    xp[:, pad:pad + H, pad:pad + W] = xf
    dw = np.zeros((gf.shape[3], R, R, Cin))
    for r in range(R):
        for s in range(R):
            dw[:, r, s, :] = np.einsum("nhwk,nhwc->kc", gf, xp[:, r:r + H, s:s + W], optimize=True)
    db = from_bf16(dy).astype(np.float64).sum(axis=(0, 1, 2))      # over the STORED dy pixels (sgamd.h)
    return alpha * dw, db


# ---- quad convolutions (conv_q.hip: sg_quad_pack, sg_conv2d_q, sg_conv2d_q_wgrad) ---------------------------------------------------------------
CONVQ_FIELDS = [("dtype", _i), ("form", _i), ("N", _i), ("Hl", _i), ("Wl", _i), ("C", _i), ("ldx", _i), ("Cout", _i),
                ("pix_flags", _i), ("epi_flags", _i), ("alpha", _f), ("beta", _f), ("x", _vp), ("wq", _vp), ("bias", _vp), ("res", _vp),
                ("mask", _vp), ("out", _vp), ("alpha_ptr", _vp), ("ldo", _i), ("ldr", _i), ("ldm", _i),
                ("x2", _vp), ("w2q", _vp), ("bias2", _vp), ("C2", _i), ("ldx2", _i), ("stats", _vp), ("x2_norelu", _i)]
CONVQ_WGRAD_FIELDS = [("dtype", _i), ("form", _i), ("N", _i), ("Hl", _i), ("Wl", _i), ("C", _i), ("ldx", _i), ("x_flags", _i), ("Cout", _i),
                      ("ldg", _i), ("alpha", _f), ("alpha_ptr", _vp), ("x", _vp), ("dy", _vp), ("dw", _vp), ("dbias", _vp), ("work", _vp),
                      ("work_floats", _ll), ("splits", _i)]


class ConvQDesc(C.Structure):
    _fields_ = CONVQ_FIELDS


class ConvQWgradDesc(C.Structure):
    _fields_ = CONVQ_WGRAD_FIELDS


Q_POOL, Q_UP, EPI_RELU = 0, 1, 8


class _Env:
    def __init__(self, env):
        self.env, self.old = env or {}, {}

    def __enter__(self):
        for k, v in self.env.items():
            self.old[k] = os.environ.get(k)
            os.environ[k] = v

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v


def copy_aligned(a):
    b = aligned(a.shape, a.dtype)
    b[...] = a
    return b


def quad_pack(lib, w9, mode):
    """w9: uint16 bf16 [M][3][3][Cs] -> [M][4][4][Cs] (sg_quad_pack)"""
    M, _, _, Cs = w9.shape
    src, dst = copy_aligned(w9), aligned((M, 4, 4, Cs), np.uint16)
    if lib.sg_quad_pack(BF16, mode, ptr(src), ptr(dst), M, Cs, None) != 0:
        raise RuntimeError(lib.sg_last_error())
    return dst


def conv_q(lib, form, x, wq, Cout, relu_in=False, bias=None, mask=None, res=None, beta=1.0, alpha=1.0, relu_out=False, x2=None, w2q=None, bias2=None,
           x2_norelu=False, env=None):
    """x: uint16 bf16 NHWC (POOL: the fine tensor, UP: the low one); returns uint16 bf16 NHWC (POOL: low, UP: fine)"""
    N, H, W, Cin = x.shape
    Hl, Wl = (H // 2, W // 2) if form == Q_POOL else (H, W)
    oshape = (N, Hl, Wl, Cout) if form == Q_POOL else (N, 2 * Hl, 2 * Wl, Cout)
    xa, wa, out = copy_aligned(x), copy_aligned(wq), aligned(oshape, np.uint16)
    keep = [xa, wa, out]
    d = ConvQDesc(dtype=BF16, form=form, N=N, Hl=Hl, Wl=Wl, C=Cin, ldx=Cin, Cout=Cout, pix_flags=PIX_RELU if relu_in else 0,
                  epi_flags=EPI_RELU if relu_out else 0, alpha=alpha, beta=beta, x=ptr(xa), wq=ptr(wa), out=ptr(out), ldo=Cout, ldr=Cout, ldm=Cout)
    for name, arr, dt in (("bias", bias, np.float32), ("mask", mask, np.uint16), ("res", res, np.uint16), ("x2", x2, np.uint16), ("w2q", w2q, np.uint16),
                          ("bias2", bias2, np.float32)):
        if arr is not None:
            a = copy_aligned(np.ascontiguousarray(arr, dtype=dt))
            keep.append(a)
            setattr(d, name, ptr(a))
    if x2 is not None:
        d.C2, d.ldx2, d.x2_norelu = x2.shape[3], x2.shape[3], int(x2_norelu)
    with _Env(env):
        if lib.sg_conv2d_q_ok(C.byref(d)) != 1:
            raise RuntimeError("sg_conv2d_q_ok == 0")
        if lib.sg_conv2d_q(C.byref(d), None) != 0:
            raise RuntimeError(lib.sg_last_error())
    return out


def conv_q_wgrad(lib, form, x, dy, relu_in=False, alpha=1.0, bias=False, splits=0, env=None):
    """returns (dw fp32 [Cout][3][3][C] -- the folded 3x3 gradient, dbias or None, splits)"""
    N, H, W, Cin = x.shape
    Cout = dy.shape[3]
    Hl, Wl = (H // 2, W // 2) if form == Q_POOL else (H, W)
    xa, ga = copy_aligned(x), copy_aligned(dy)
    dw = aligned((Cout, 3, 3, Cin), np.float32)
    db = aligned((Cout,), np.float32) if bias else None
    d = ConvQWgradDesc(dtype=BF16, form=form, N=N, Hl=Hl, Wl=Wl, C=Cin, ldx=Cin, x_flags=PIX_RELU if relu_in else 0, Cout=Cout, ldg=Cout, alpha=alpha,
                       x=ptr(xa), dy=ptr(ga), dw=ptr(dw), dbias=ptr(db) if bias else None, splits=splits)
    with _Env(env):
        sp, wf = C.c_int(0), C.c_longlong(0)
        if lib.sg_conv2d_q_wgrad_plan(C.byref(d), C.byref(sp), C.byref(wf)) != 0 or sp.value == 0:
            raise RuntimeError("sg_conv2d_q_wgrad_plan: not eligible")
        work = aligned((max(int(wf.value), 1),), np.float32)
        d.work, d.work_floats, d.splits = ptr(work), wf.value, sp.value
        if lib.sg_conv2d_q_wgrad(C.byref(d), None) != 0:
            raise RuntimeError(lib.sg_last_error())
    return dw, db, sp.value


# ---- plain convolution forward / data gradient (conv.hip sg_conv2d_fwd and the engines behind it) --------------------------------------------------------
CONV_FWD_FIELDS = [("dtype", _i), ("N", _i), ("Hs", _i), ("Ws", _i), ("C", _i), ("ldx", _i), ("Ho", _i), ("Wo", _i), ("Cout", _i),
                   ("R", _i), ("S", _i), ("stride", _i), ("pad_h", _i), ("pad_w", _i), ("pix_flags", _i), ("epi_flags", _i),
                   ("alpha", _f), ("beta", _f), ("x", _vp), ("w", _vp), ("bias", _vp), ("res", _vp), ("mask", _vp), ("out", _vp),
                   ("alpha_ptr", _vp), ("ldo", _i), ("ldr", _i), ("ldm", _i)]


class ConvFwdDesc(C.Structure):
    _fields_ = CONV_FWD_FIELDS


PIX_QUAD, EPI_POOL = 4, 4


def conv_fwd(lib, x, w, R, S, pad, relu_in=False, up=False, pool=False, bias=None, mask=None, res=None, alpha=1.0, beta=1.0, relu_out=False, env=None):
    """x: uint16 bf16 [N][Hs][Ws][C]; w: uint16 bf16 [Cout][R][S][C]; stride 1. pool: 2x2 average pooling in the epilogue (the kernel then wants its
    pixels in quad order: SG_PIX_QUAD). Returns uint16 bf16 NHWC through sg_conv2d_fwd -- whichever engine takes the problem (see the SG_CONV_* switches)."""
    N, Hs, Ws, Cin = x.shape
    Cout = w.shape[0]
    u = 2 if up else 1
    Ho, Wo = Hs * u + 2 * pad - R + 1, Ws * u + 2 * pad - S + 1
    oshape = (N, Ho // 2, Wo // 2, Cout) if pool else (N, Ho, Wo, Cout)
    xa, wa, out = copy_aligned(x), copy_aligned(w), aligned(oshape, np.uint16)
    keep = [xa, wa, out]
    pf = (PIX_RELU if relu_in else 0) | (PIX_UPSAMPLE if up else 0) | (PIX_QUAD if pool else 0)
    ef = (EPI_POOL if pool else 0) | (EPI_RELU if relu_out else 0)
    d = ConvFwdDesc(dtype=BF16, N=N, Hs=Hs, Ws=Ws, C=Cin, ldx=Cin, Ho=Ho, Wo=Wo, Cout=Cout, R=R, S=S, stride=1, pad_h=pad, pad_w=pad, pix_flags=pf, epi_flags=ef,
                    alpha=alpha, beta=beta, x=ptr(xa), w=ptr(wa), out=ptr(out), ldo=Cout, ldr=Cout, ldm=Cout)
    for name, arr, dt in (("bias", bias, np.float32), ("mask", mask, np.uint16), ("res", res, np.uint16)):
        if arr is not None:
            a = copy_aligned(np.ascontiguousarray(arr, dtype=dt))
            keep.append(a)
            setattr(d, name, ptr(a))
    with _Env(env):
        if lib.sg_conv2d_fwd(C.byref(d), None) != 0:
            raise RuntimeError(lib.sg_last_error())
    return out
