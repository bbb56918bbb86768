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
