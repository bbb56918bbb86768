"""Index-level CPU model of csrc/conv_rs96.h (the EXPERIMENTAL 96 -> 96 channel row-streaming kernel, not yet run on a GPU).

It replays, lane by lane and with the kernel's own formulas, what the kernel does to memory: the LDS-DMA piece -> (pixel, 16-byte chunk)
mapping, the ring-slot arithmetic, the B-fragment addresses of every (tap, sub-step, pixel tile), the A-fragment (weight) addresses, the
32x32x16 MFMA operand / accumulator layout, and the register epilogue (bias, scale, 2x2 pooling by row-pair accumulation + quad_perm add,
output addresses) -- and compares the result with a direct convolution. What it can NOT see: barriers, waits, anything about timing.

    python tools/rs96_model.py            # prints max |difference| for the four (relu, pool) variants; exits non-zero on a mismatch
Run by tests/test_host_cpu.py at a small size."""
import sys

import numpy as np

W, C, NKT, PITCH = 128, 96, 6, 2 * 96 + 16
ROWB, NRING, NPIECE = (W + 2) * PITCH, 5, W * (2 * 96 + 16) // 1024


def run(N=1, H=8, SH=4, relu=False, pool=False, seed=0, ldx=96, ldo=96):
    rng = np.random.RandomState(seed)
    x = rng.randint(-3, 4, size=(N, H, W, ldx)).astype(np.float64)          # small integers: every sum is exact
    w = rng.randint(-2, 3, size=(96, 9, C)).astype(np.float64)              # [cout][tap][cin], K index = tap * C + c
    bias = rng.randint(-4, 5, size=96).astype(np.float64)
    al = 0.25 if pool else 1.0
    K = 9 * C
    wflat = w.reshape(96, K)
    Ho, Wo = (H // 2, W // 2) if pool else (H, W)
    out = np.full((N * Ho * Wo * ldo,), np.nan)
    spi = H // SH
    for blk in range(N * spi):
        n, r0 = blk // spi, (blk % spi) * SH
        lds = np.zeros(NRING * ROWB // 2 + 64)                               # 2-byte elements; pad pixels zero (the kernel's first loop)
        lds[:] = np.nan
        for row in range(NRING):
            for side in range(2):
                o = (row * ROWB + side * (W + 1) * PITCH) // 2
                lds[o:o + PITCH // 2] = 0.0

        def issue_row(rho):
            r = r0 - 1 + rho
            rv = (0 <= r < H) and (rho <= SH + 1)
            slot = (rho % NRING) * ROWB + PITCH
            for q in range(NPIECE):
                for lane in range(64):
                    b = q * 1024 + lane * 16
                    pix, ch = b // PITCH, (b % PITCH) >> 4
                    dst = (slot + q * 1024 + lane * 16) // 2
                    if rv and ch < C // 8:
                        lds[dst:dst + 8] = x[n, r, pix, ch * 8:ch * 8 + 8]
                    else:
                        lds[dst:dst + 8] = 0.0                               # out-of-range offset: the hardware writes zeros
        for rho in range(4):
            issue_row(rho)
        acc = np.zeros((3, 4, 64, 16))                                       # [consumer wave][pixel tile][lane][register]
        s0 = 0
        for j in range(SH):
            issue_row(j + 4)                                                 # (the producer, after the barrier of step j)
            s1 = (s0 + 1) % NRING
            s2 = (s1 + 1) % NRING
            if (not pool) or (j & 1) == 0:
                acc[:] = 0.0
            for wave in range(3):
                co0 = 32 * wave
                for t in range(9):
                    for ks in range(NKT):
                        # operands of all 64 lanes
                        A = np.zeros((32, 16)); Bm = np.zeros((4, 32, 16))
                        for lane in range(64):
                            frow, fhi = lane & 31, lane >> 5
                            a0 = (co0 + frow) * K + fhi * 8 + t * C + ks * 16
                            A[frow, fhi * 8:fhi * 8 + 8] = wflat.reshape(-1)[a0:a0 + 8]
                            lb = frow * PITCH + fhi * 16
                            base = ((s0, s1, s2)[t // 3]) * ROWB + lb
                            for pt in range(4):
                                ad = (base + (t % 3) * PITCH + ks * 32 + pt * 32 * PITCH) // 2
                                v = lds[ad:ad + 8]
                                if relu:
                                    v = np.maximum(v, 0.0)
                                Bm[pt, frow, fhi * 8:fhi * 8 + 8] = v
                        for pt in range(4):
                            Dm = A @ Bm[pt].T                                # [cout row][pixel col]
                            for lane in range(64):
                                frow, fhi = lane & 31, lane >> 5
                                for r in range(16):
                                    acc[wave, pt, lane, r] += Dm[(r // 4) * 8 + fhi * 4 + (r % 4), frow]
            if (not pool) or (j & 1) == 1:
                for wave in range(3):
                    co0 = 32 * wave
                    for lane in range(64):
                        frow, fhi = lane & 31, lane >> 5
                        obase = ((n * Ho + (r0 // 2 if pool else r0)) * Wo + (frow // 2 if pool else frow)) * ldo + co0 + 4 * fhi
                        o = obase + ((j >> 1) if pool else j) * (Wo * ldo)
                        for pt in range(4):
                            for g in range(4):
                                for e in range(4):
                                    a = acc[wave, pt, lane, 4 * g + e]
                                    if pool:
                                        a = a + acc[wave, pt, lane ^ 1, 4 * g + e]      # quad_perm [1,0,3,2]
                                    v = a * al + bias[co0 + 8 * g + 4 * fhi + e]
                                    if (not pool) or (lane & 1) == 0:
                                        out[o + (pt * 16 if pool else pt * 32) * ldo + 8 * g + e] = v
            s0 = s1
    # reference
    xr = np.maximum(x, 0.0) if relu else x
    xp = np.zeros((N, H + 2, W + 2, C)); xp[:, 1:-1, 1:-1, :] = xr[..., :C]
    ref = np.zeros((N, H, W, 96))
    for dr in range(3):
        for dc in range(3):
            ref += np.einsum("nhwc,oc->nhwo", xp[:, dr:dr + H, dc:dc + W, :], w[:, dr * 3 + dc, :])
    if pool:
        ref = ref.reshape(N, H // 2, 2, W // 2, 2, 96).sum(axis=(2, 4))
    ref = ref * al + bias
    got = out.reshape(N, Ho, Wo, ldo)[..., :96]
    assert not np.isnan(got).any(), "outputs never written"
    return float(np.abs(got - ref).max())


if __name__ == "__main__":
    bad = 0
    for relu in (False, True):
        for pool in (False, True):
            e = run(N=1, H=8, SH=4, relu=relu, pool=pool)
            print(f"conv_rs96 model  relu={relu!s:5} pool={pool!s:5}  max |diff| = {e}")
            bad += e != 0.0
    sys.exit(1 if bad else 0)
