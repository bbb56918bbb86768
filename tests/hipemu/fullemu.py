"""Whole-library hipemu build (TEST INFRASTRUCTURE): every translation unit of pytorch-studiogan_amd/csrc compiled for the host against the interpreter, linked
into one library with the SAME C ABI as libsgamd.so, and put behind the product's ctypes binding FROM THE TEST PROCESS (nothing in the package
knows about it: `install()` swaps the handle `_lib._lib`, and the three places that insist on a GPU -- `_lib.ptr`, `_lib.stream`, `_lib.require_gpu` -- are
monkeypatched for the duration). With it the package's Python layer (autograd functions, weight bank, optimiser, worker) runs its real launch
sequence on CPU tensors through the real kernel sources, lane by lane: network-level checks of host-side changes without GPU time. It is slow
(~10^5 wave-level operations per second): width-8 fixtures only."""
import concurrent.futures
import ctypes as C
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import emu  # noqa: E402
import translate  # noqa: E402

OBJ = os.path.join(emu.BUILD, "obj")


def _sha(paths, extra=""):
    h = hashlib.sha256()
    for p in paths:
        with open(p, "rb") as f:
            h.update(f.read())
    h.update(extra.encode())
    return h.hexdigest()[:16]


def build(opt="-O1", jobs=None):
    translate.translate_tree(emu.SRC)
    os.makedirs(OBJ, exist_ok=True)
    hdrs = [os.path.join(emu.SRC, f) for f in sorted(os.listdir(emu.SRC)) if f.endswith(".h")] + [os.path.join(HERE, "include", "hip", "hip_runtime.h"),
                                                                                                   os.path.join(emu.REPO if hasattr(emu, "REPO") else translate.REPO, "include", "sgamd.h")]
    hh = _sha(hdrs, opt)
    units = [f[:-4] for f in sorted(os.listdir(emu.SRC)) if f.endswith(".hip") and f != "p2p.hip"]      # (p2p.hip: IPC peer memory, not emulated -- p2p_stub.cpp keeps the ABI complete)
    common = [emu.CXX, "-x", "c++", "-std=c++17", "-fPIC", "-ffp-contract=off", "-I" + os.path.join(HERE, "include"), "-I" + emu.SRC,
              "-Wno-unknown-attributes", "-Wno-unused-value"]
    todo, objs = [], []
    for u in units + ["rt", "p2p_stub"]:
        src = os.path.join(HERE, u + ".cpp") if u in ("rt", "p2p_stub") else os.path.join(emu.SRC, u + ".hip")
        o = os.path.join(OBJ, "%s_%s.o" % (u, _sha([src], hh)))
        objs.append(o)
        if not os.path.exists(o):
            for old in os.listdir(OBJ):
                if old.startswith(u + "_") and old.endswith(".o") and len(old) == len(u) + 1 + 16 + 2:
                    os.remove(os.path.join(OBJ, old))
            todo.append(common + (["-O3", "-march=native"] if u == "rt" else [opt]) + ["-c", src, "-o", o])

    def run(cmd):
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipemu build failed: %s\n%s" % (" ".join(cmd[-3:]), r.stdout[-4000:]))
    with concurrent.futures.ThreadPoolExecutor(max_workers=jobs or max(2, os.cpu_count() or 2)) as ex:
        list(ex.map(run, todo))
    lib = os.path.join(emu.BUILD, "libsgamd_emu_%s.so" % _sha(objs))
    if not os.path.exists(lib):
        for old in os.listdir(emu.BUILD):
            if old.startswith("libsgamd_emu_"):
                os.remove(os.path.join(emu.BUILD, old))
        run([emu.CXX, "-shared", "-o", lib + ".tmp"] + objs + ["-ldl"])
        os.rename(lib + ".tmp", lib)
    return lib


class Installed:
    """context manager: the product's `_lib` module bound to the emulated library"""

    def __init__(self, dma_late=1, greedy=1, seed=1):
        self.cfg = (dma_late, greedy, seed)

    def __enter__(self):
        import studiogan_amd
        L = studiogan_amd._lib
        self.L = L
        lib = C.CDLL(build())
        lib.sg_last_error.restype = C.c_char_p
        lib.sg_version.restype = C.c_int
        for name, args in L._PROTOS.items():
            fn = getattr(lib, name)
            fn.argtypes = args
            fn.restype = C.c_int
        lib.sg_conv_rs_launches.restype = C.c_longlong
        lib.hipemu_config(*[int(v) for v in self.cfg[:2]], C.c_uint(self.cfg[2]))
        self.saved = (L._lib, L.ptr, L.stream, L.require_gpu)
        L._lib = lib
        L.ptr = lambda t: None if t is None else t.data_ptr()
        L.stream = lambda: None
        L.require_gpu = lambda dev: None
        self.lib = lib
        return self

    def counters(self):
        return emu.counters(self.lib)

    def __exit__(self, *a):
        L = self.L
        L._lib, L.ptr, L.stream, L.require_gpu = self.saved
