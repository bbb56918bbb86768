"""Shared helpers for the parity tests."""
import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def rel_err(a, b):
    """max |a-b| relative to the range of the expected tensor b (the tolerance unit of SURVEY.md §8c)."""
    a = a.detach().double().cpu().reshape(-1)
    b = b.detach().double().cpu().reshape(-1)
    assert a.shape == b.shape, f"shape mismatch {tuple(a.shape)} vs {tuple(b.shape)}"
    assert torch.isfinite(a).all(), "non-finite values in result"
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def check(name, a, b, tol, report=None):
    e = rel_err(a, b)
    line = f"{name:48s} err={e:.3e} tol={tol:.1e} {'ok' if e <= tol else 'FAIL'}"
    print(line)
    if report is not None:
        report.append((name, e, tol))
    assert e <= tol, line
    return e


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    meta = json.load(open(os.path.join(GOLDEN, name + ".json")))
    fix = {k: torch.from_numpy(z[k]) for k in z.files}
    return fix, meta


def sub(fix, prefix):
    return {k[len(prefix):]: v for k, v in fix.items() if k.startswith(prefix)}


def tol_for(dtype):
    # fp32 path: exact-fp32 MFMA, only summation order differs; bf16 path vs the fp32 oracle
    return 2e-4 if dtype == torch.float32 else 3e-2


class Collector:
    """Record every comparison, fail once at the end (so one GPU run shows all mismatches)."""

    def __init__(self):
        self.rows = []

    def check(self, name, a, b, tol, floor=0.0, l2=False):
        """err = max|a-b| / max(max|b|, floor): `floor` keeps tensors that are analytically ~0 (e.g. the bias of a
        convolution feeding a batch norm) from being judged against their own rounding noise.
        l2=True: err = ||a-b||_2 / max(||b||_2, floor*sqrt(n)) -- the robust metric for bf16 gradients, where a few
        ReLU units flipping under 2^-9 relative rounding produce sparse O(1) element errors."""
        a = a.detach().double().cpu().reshape(-1)
        b = b.detach().double().cpu().reshape(-1)
        assert a.shape == b.shape, f"{name}: shape mismatch {tuple(a.shape)} vs {tuple(b.shape)}"
        finite = bool(torch.isfinite(a).all())
        if not finite:
            e = float("inf")
        elif l2:
            e = float((a - b).norm() / max(float(b.norm()), floor * (a.numel() ** 0.5), 1e-30))
        else:
            e = float((a - b).abs().max() / max(float(b.abs().max()), floor, 1e-30))
        self.rows.append((name, e, tol))
        print(f"{name:52s} {'l2' if l2 else 'mx'}err={e:.3e} tol={tol:.1e} {'ok' if e <= tol else 'FAIL'}")
        return e

    def finish(self):
        bad = [(n, e, t) for n, e, t in self.rows if not e <= t]
        assert not bad, "mismatches: " + "; ".join(f"{n} err={e:.2e} tol={t:.0e}" for n, e, t in bad[:12]) + (f" (+{len(bad) - 12} more)" if len(bad) > 12 else "")
