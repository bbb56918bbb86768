"""Noise floor of bf16 parity at full depth, measured on the CPU ORACLE ALONE (no GPU): how far the bf16-emulating oracle
(oracle/restate.py Bf16Emu: bf16 rounding at the storage points of the HIP path) moves when every weight is perturbed by a relative
1e-5 -- far below one bf16 ulp (3.9e-3) -- next to the same perturbation on the fp32 oracle, and the distance emulated-bf16 vs fp32.
If the emulated network moves by X under such a perturbation, no two bf16 evaluations that differ in a single rounding decision can
be expected to agree better than X: that is the bound the full-width bf16 tests use (tests/test_blocks_gpu.py).

    python tools/bf16_noise_floor.py biggan128w bigdeep128w sngan32w wgangp128w > profiles/r02_bf16_noise_floor.txt
TEST INFRASTRUCTURE (imports oracle/)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from util import load_golden, sub  # noqa: E402
from oracle import make_golden as MG, restate as O  # noqa: E402

EPS = 1e-5


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / max(float(b.norm()), 1e-30)), float((a - b).abs().max() / max(float(b.abs().max()), 1e-30))


def run(name):
    fix, meta = load_golden(name)
    y = meta["yaml"]
    isb = lambda k: any(s in k for s in ("weight_u", "weight_v", "running_", "num_batches"))
    S = y["DATA"]["img_size"]
    for which in ("G", "D"):
        init = sub(fix, which + "_init/")
        res = {}
        for emu in (False, True):
            ocfg = dict(MG.oracle_cfg(y), **({"emu": O.Bf16Emu} if emu else {}))
            for rep in range(2):
                g = torch.Generator().manual_seed(5)
                P = {k: (v.clone() if rep == 0 else v * (1 + EPS * torch.randn(v.shape, generator=g))).requires_grad_(True) for k, v in init.items() if not isb(k)}
                B = {k: v.clone() for k, v in init.items() if isb(k)}
                if which == "G":
                    z, lab = fix["in/z0"], fix["in/fl0"]
                    out = O.model_fns(ocfg)[0](z, lab, P, B, bn_mode="track")
                    (out * torch.randn(z.shape[0], 3, S, S, generator=torch.Generator().manual_seed(11))).sum().backward()
                else:
                    x, lab = fix["in/real0"], fix["in/rl0"]
                    out, _ = O.model_fns(ocfg)[1](x, lab, P, B)
                    out.sum().backward()
                gm = max(float(v.grad.abs().max()) for v in P.values() if v.grad is not None)
                res[(emu, rep)] = (out.detach(), {k: v.grad.detach() for k, v in P.items() if v.grad is not None and float(v.grad.abs().max()) > 1e-2 * gm})
        for emu in (False, True):
            (o0, g0), (o1, g1) = res[(emu, 0)], res[(emu, 1)]
            l2, mx = rel(o1, o0)
            gl = sorted(rel(g1[k], g0[k])[0] for k in g0 if k in g1)
            print(f"{name:12s} {which} {'emulated bf16' if emu else 'fp32         '} oracle, weights * (1 + {EPS:g} N(0,1)): output L2 {l2:.2e} max {mx:.2e} | "
                  f"weight-gradient L2 median {gl[len(gl) // 2]:.2e} worst {gl[-1]:.2e}")
        l2, mx = rel(res[(True, 0)][0], res[(False, 0)][0])
        gl = sorted(rel(res[(True, 0)][1][k], res[(False, 0)][1][k])[0] for k in res[(False, 0)][1] if k in res[(True, 0)][1])
        print(f"{name:12s} {which} emulated bf16 vs fp32 oracle:                         output L2 {l2:.2e} max {mx:.2e} | "
              f"weight-gradient L2 median {gl[len(gl) // 2]:.2e} worst {gl[-1]:.2e}")


if __name__ == "__main__":
    for n in sys.argv[1:]:
        run(n)
