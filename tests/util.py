"""Shared helpers for the parity tests."""
import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class Sampled:
    """Expected value of a LARGE tensor in a compact fixture (oracle/make_golden.py compact_entries): evenly spaced
    samples plus [sum, l2] of the whole tensor."""

    def __init__(self, values, norms):
        self.values, self.norms = values, norms

    def pick(self, a):
        from oracle.make_golden import sample_index
        flat = a.detach().reshape(-1)
        return flat[sample_index(flat.numel(), self.values.numel()).to(flat.device)]

    def absmax(self):
        return float(self.values.abs().max())


def absmax(v):
    return v.absmax() if isinstance(v, Sampled) else float(v.abs().max())


def _pair(a, b):
    if isinstance(b, Sampled):
        return b.pick(a), b.values
    return a, b


def rel_err(a, b):
    """max |a-b| relative to the range of the expected tensor b (the tolerance unit of SURVEY.md §8c)."""
    a, b = _pair(a, b)
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
    fix = {k: torch.from_numpy(z[k]) for k in z.files if not k.startswith(("exps/", "expn/"))}
    if meta.get("compact"):
        # compact fixture: the initial state is a formula (numpy RandomState keyed by tensor name), large expected tensors are samples
        from oracle.make_golden import formula_state
        for k, v in formula_state(meta["G_spec"], meta["seed"]).items():
            fix["G_init/" + k] = v
        for k, v in formula_state(meta["D_spec"], meta["seed"] + 1).items():
            fix["D_init/" + k] = v
        for k in z.files:
            if k.startswith("exps/"):
                fix["exp/" + k[5:]] = Sampled(torch.from_numpy(z[k]), torch.from_numpy(z["expn/" + k[5:]]))
    return fix, meta


def load_cond(name):
    """Measured conditioning of a fixture (tests/golden/<name>.cond.npz) as {key: tensor([rms, max])}; empty when the fixture has none."""
    path = os.path.join(GOLDEN, name + ".cond.npz")
    if not os.path.exists(path):
        return {}
    z = np.load(path)
    return {k: torch.from_numpy(z[k]) for k in z.files}


def hyper(y):
    """hyper-parameters of a fixture's yaml with the reference's defaults filled in (reference src/config.py:128,232-247)."""
    M, O, Ls = y.get("MODEL", {}), y.get("OPTIMIZATION", {}), y.get("LOSS", {})
    return dict(z_dim=M.get("z_dim", 128), adv_loss=Ls.get("adv_loss", "vanilla"), g_lr=O.get("g_lr", 0.0002), d_lr=O.get("d_lr", 0.0002),
                beta1=O.get("beta1", 0.5), beta2=O.get("beta2", 0.999), apply_gp=Ls.get("apply_gp", False), gp_lambda=Ls.get("gp_lambda", 10.0))


def sub(fix, prefix):
    return {k[len(prefix):]: v for k, v in fix.items() if k.startswith(prefix)}


def tol_for(dtype):
    # fp32 path: exact-fp32 MFMA, only summation order differs; bf16 path vs the fp32 oracle
    return 2e-4 if dtype == torch.float32 else 3e-2


class Collector:
    """Record every comparison, fail once at the end (so one GPU run shows all mismatches)."""

    def __init__(self):
        self.rows = []

    NOISE_FACTOR = 4.0

    def check(self, name, a, b, tol, floor=0.0, l2=False, abs_tol=None, noise=None, abs_ok=None):
        """err = max|a-b| / max(max|b|, floor): `floor` keeps tensors that are analytically ~0 (e.g. the bias of a
        convolution feeding a batch norm) from being judged against their own rounding noise.
        l2=True: err = ||a-b||_2 / max(||b||_2, floor*sqrt(n)) -- the robust metric where a few ReLU units whose
        pre-activation sits within rounding distance of 0 flip and produce sparse O(1) element errors (bf16 gradients;
        fp32 gradients of the full-width networks).  abs_tol: pass on max|a-b| <= abs_tol instead.
        abs_ok: the comparison also passes when max|a-b| <= abs_ok whatever the relative error -- for parameters after Adam steps, where
        an element whose gradient is rounding noise moves by +-lr with a sign that two correct implementations need not share.
        noise: [rms, max] of the ORACLE's own movement of this tensor under a 2e-6 relative weight perturbation
        (tests/golden/<name>.cond.npz, oracle/make_golden.py conditioning()): the tolerance becomes tol + NOISE_FACTOR * that movement
        expressed in the metric used -- a measured bound for ill-conditioned quantities instead of a hand-picked one.
        Both metrics are always printed."""
        n_rms = float(noise[0]) if noise is not None else 0.0
        n_mx = float(noise[1]) if noise is not None else 0.0
        if isinstance(b, Sampled):
            # whole-tensor l2 norm first (catches errors between the sample points), then the samples
            na, nb = float(a.detach().double().norm()), float(b.norms[1])
            den = max(nb, floor * (a.numel() ** 0.5), 1e-30)
            en = abs(na - nb) / den
            tn = tol + self.NOISE_FACTOR * n_rms * (a.numel() ** 0.5) / den
            self.rows.append((name + "#norm", en, tn))
            print(f"{name + '#norm':52s} nrerr={en:.3e} tol={tn:.1e} {'ok' if en <= tn else 'FAIL'}")
        a, b = _pair(a, b)
        a = a.detach().double().cpu().reshape(-1)
        b = b.detach().double().cpu().reshape(-1)
        assert a.shape == b.shape, f"{name}: shape mismatch {tuple(a.shape)} vs {tuple(b.shape)}"
        if not bool(torch.isfinite(a).all()):
            e = e2 = em = float("inf")
        else:
            d2 = max(float(b.norm()), floor * (a.numel() ** 0.5), 1e-30)
            dm = max(float(b.abs().max()), floor, 1e-30)
            e2 = float((a - b).norm() / d2)
            em = float((a - b).abs().max() / dm)
            e = e2 if l2 else em
            if noise is not None and abs_tol is None:
                tol = tol + self.NOISE_FACTOR * (n_rms * (a.numel() ** 0.5) / d2 if l2 else n_mx / dm)
            if abs_tol is not None:
                e, tol = float((a - b).abs().max()), abs_tol + self.NOISE_FACTOR * n_mx
            elif abs_ok is not None and float((a - b).abs().max()) <= abs_ok:
                e = min(e, tol)
        self.rows.append((name, e, tol))
        print(f"{name:52s} mx={em:.3e} l2={e2:.3e} [{'abs' if abs_tol is not None else 'l2' if l2 else 'mx'}] tol={tol:.1e} {'ok' if e <= tol else 'FAIL'}")
        return e

    def finish(self):
        bad = [(n, e, t) for n, e, t in self.rows if not e <= t]
        assert not bad, "mismatches: " + "; ".join(f"{n} err={e:.2e} tol={t:.0e}" for n, e, t in bad[:12]) + (f" (+{len(bad) - 12} more)" if len(bad) > 12 else "")
