"""Does an HBM-bound spectral-norm pass hide behind an MFMA-bound convolution stack when it is issued on a second HIP stream? (round 6 design probe)
Times, with HIP events on the main stream: a D forward alone, the D table's sg_sn_forward alone, both back to back on one stream, and the sg_sn_forward on a side stream
while the D forward runs on the main one (the main stream waits for the side stream at the end)."""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import studiogan_amd
from studiogan_amd import _lib as L, bank as BK

dev = torch.device("cuda:0")
wl = bench.WORKLOADS["biggan128"]
G, D = bench.build(wl, True, dev)
B = int(os.environ.get("B", "256"))
x = torch.randn(B, 3, 128, 128, device=dev)
y = torch.randint(0, 1000, (B,), device=dev)
with torch.no_grad():
    for _ in range(2):
        D(x, y)
bank = BK.get_bank(D, torch.bfloat16)
slot = bank.slots[2]
flags = tuple(True for _ in bank.layers)
arr, tab, groups = bank._desc(slot, flags)
esz = ctypes.sizeof(L.SnLayer)
side = torch.cuda.Stream()


def sn(stream_handle):
    for first, count in groups:
        L.call("sg_sn_forward", bank.sgdt, tab.data_ptr() + first * esz, ctypes.cast(ctypes.addressof(arr) + first * esz, ctypes.POINTER(L.SnLayer)), count, bank.eps,
               bank.work2.data_ptr(), bank.work2.numel(), stream_handle)


bank.work2 = torch.zeros_like(bank.work)      # (its own scratch: the D forward's own spectral-norm pass uses bank.work)


def timed(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def fwd():
    with torch.no_grad():
        D(x, y)


def serial():
    sn(L.stream())
    fwd()


def overlapped():
    side.wait_stream(torch.cuda.current_stream())
    sn(side.cuda_stream)
    fwd()
    torch.cuda.current_stream().wait_stream(side)


t_f, t_s = timed(fwd), timed(lambda: sn(L.stream()))
t_ser, t_ov = timed(serial), timed(overlapped)
print(f"batch {B}: D forward (incl. its own SN pass) {t_f:.3f} ms | extra SN pass alone {t_s:.3f} ms | serial {t_ser:.3f} ms | side stream {t_ov:.3f} ms "
      f"-> hidden {100 * (t_ser - t_ov) / max(t_s, 1e-9):.0f} % of the pass")
# the same against a G forward (cBN + upsampling convolutions)
z = torch.randn(B, wl["z_dim"], device=dev)


def gfwd():
    with torch.no_grad():
        G(z, y)


def g_serial():
    sn(L.stream())
    gfwd()


def g_over():
    side.wait_stream(torch.cuda.current_stream())
    sn(side.cuda_stream)
    gfwd()
    torch.cuda.current_stream().wait_stream(side)


t_g = timed(gfwd)
t_gs, t_go = timed(g_serial), timed(g_over)
print(f"batch {B}: G forward {t_g:.3f} ms | serial with D's SN pass {t_gs:.3f} ms | side stream {t_go:.3f} ms -> hidden {100 * (t_gs - t_go) / max(t_s, 1e-9):.0f} % of the pass")
