"""CPU, world_size 2 over gloo: the host-side data-parallel logic of the N>1 path (the kernels themselves need a GPU):
 * chunking + pipelined all-reduce of the flat gradient arena used by FusedAdam (studiogan_amd/optim.py),
 * the sync-BN contract: all-reducing the per-rank fp64 partial sums {sum x, sum x^2} (count * world) and the backward
   channel terms reproduces full-batch batch-norm forward/backward exactly (what sg_bn_partial_stats / sg_bn_finalize /
   sg_bn_bwd_finalize compute around the all-reduce in studiogan_amd/functional/norm.py:BNFn),
 * gradient averaging across ranks == full-batch gradient for a mean-reduced loss."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn.functional as TF


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(rank, world, port, fn, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ret[rank] = fn(rank, world)
    finally:
        dist.destroy_process_group()


def _spawn(fn, world=2):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_run, args=(world, _free_port(), fn, ret), nprocs=world, join=True)
    return dict(ret)


def test_chunk_ranges_cover_and_align():
    import studiogan_amd  # noqa: F401
    from studiogan_amd.optim import chunk_ranges
    for n in (5, 1 << 20, (1 << 22) + 12, 88_000_004):
        for nch in (1, 4, 7):
            r = chunk_ranges(n, nch)
            assert r[0][0] == 0 and r[-1][1] == n
            for (a, b), (c, d) in zip(r[:-1], r[1:]):
                assert b == c and a % 4 == 0 and c % 4 == 0 and b > a
            assert len(r) <= nch


def _pipeline_job(rank, world):
    import studiogan_amd  # noqa: F401
    from studiogan_amd.optim import chunk_ranges, pipelined_allreduce
    n = (1 << 21) + 36
    g = torch.Generator().manual_seed(100 + rank)
    grad = torch.randn(n, generator=g)
    p = torch.zeros(n)
    seen = []
    for lo, hi in pipelined_allreduce(grad, chunk_ranges(n, 4, min_chunk=1 << 18)):
        p[lo:hi] -= 0.1 * grad[lo:hi] / world      # the "optimizer" consumes each slice as it lands
        seen.append((lo, hi))
    return p, seen


def test_pipelined_allreduce_gloo():
    out = _spawn(_pipeline_job)
    n = (1 << 21) + 36
    full = sum(torch.randn(n, generator=torch.Generator().manual_seed(100 + r)) for r in range(2)) / 2
    for r in range(2):
        p, seen = out[r]
        assert seen[0][0] == 0 and seen[-1][1] == n and len(seen) > 1
        assert torch.allclose(p, -0.1 * full, atol=1e-6)
    assert torch.equal(out[0][0], out[1][0]), "ranks must end bit-identical"


def _syncbn_job(rank, world):
    """Each rank holds half of the batch; statistics and backward channel terms are exchanged exactly the way BNFn does."""
    g = torch.Generator().manual_seed(7)
    N, C, H = 8, 6, 5
    x_full = torch.randn(N, C, H, H, generator=g, dtype=torch.float64) * 2 + 0.3
    gy_full = torch.randn(N, C, H, H, generator=g, dtype=torch.float64)
    gain_full = 1 + 0.2 * torch.randn(N, C, generator=g, dtype=torch.float64)
    sl = slice(rank * N // world, (rank + 1) * N // world)
    x, gy, gain = x_full[sl], gy_full[sl], gain_full[sl]
    eps = 1e-4
    # forward: partial sums -> all-reduce -> finalize (sg_bn_partial_stats / sg_bn_finalize)
    partial = torch.stack([x.sum(dim=(0, 2, 3)), (x * x).sum(dim=(0, 2, 3))], dim=1).reshape(-1).clone()
    dist.all_reduce(partial)
    count = x[:, 0].numel() * world
    p = partial.view(C, 2)
    mean = p[:, 0] / count
    var = p[:, 1] / count - mean * mean
    invstd = 1.0 / torch.sqrt(var + eps)
    xh = (x - mean.view(1, C, 1, 1)) * invstd.view(1, C, 1, 1)
    y = xh * gain.view(-1, C, 1, 1)
    # backward: per-(n,c) sums -> per-channel terms -> all-reduce -> apply (sg_bn_bwd_reduce/finalize/apply)
    s1 = gy.sum(dim=(2, 3))
    s2 = (gy * xh).sum(dim=(2, 3))
    chan = torch.stack([(gain * s1).sum(0), (gain * s2).sum(0)], dim=1).reshape(-1).clone()
    dist.all_reduce(chan)
    ch = chan.view(C, 2)
    dx = invstd.view(1, C, 1, 1) * (gain.view(-1, C, 1, 1) * gy - (ch[:, 0].view(1, C, 1, 1) + xh * ch[:, 1].view(1, C, 1, 1)) / count)
    # full-batch reference through autograd
    xr = x_full.clone().requires_grad_(True)
    yr = TF.batch_norm(xr, None, None, None, None, True, 0.1, eps) * gain_full.view(N, C, 1, 1)
    yr.backward(gy_full)
    return float((y - yr.detach()[sl]).abs().max()), float((dx - xr.grad[sl]).abs().max())


def test_sync_bn_contract_gloo():
    out = _spawn(_syncbn_job)
    for r in range(2):
        ef, eb = out[r]
        assert ef < 1e-10 and eb < 1e-10, (ef, eb)


def _grad_avg_job(rank, world):
    torch.manual_seed(0)
    lin = torch.nn.Linear(6, 3)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(8, 6, generator=g)
    sl = slice(rank * 4, rank * 4 + 4)
    loss = TF.relu(1.0 - lin(x[sl]).sum(1)).mean()          # hinge-style, mean over the LOCAL batch
    loss.backward()
    flat = torch.cat([p.grad.reshape(-1) for p in lin.parameters()])
    dist.all_reduce(flat)
    flat /= world                                            # == grad_scale = 1/world in sg_adam_ema
    lin2 = torch.nn.Linear(6, 3)
    lin2.load_state_dict(lin.state_dict())
    TF.relu(1.0 - lin2(x).sum(1)).mean().backward()
    ref = torch.cat([p.grad.reshape(-1) for p in lin2.parameters()])
    return float((flat - ref).abs().max())


def test_gradient_average_equals_full_batch_gloo():
    out = _spawn(_grad_avg_job)
    assert max(out.values()) < 1e-6


def _gather_job(rank, world):
    import studiogan_amd  # noqa: F401
    from studiogan_amd import losses as SL
    g = torch.Generator().manual_seed(7)
    full = torch.randn(world * 3, 5, generator=g)
    w = torch.randn(world * 3, 5, generator=g)
    x = full[rank * 3:(rank + 1) * 3].clone().requires_grad_(True)
    out = SL.gather_logits(x, dist.group.WORLD)
    (out * w).sum().backward()
    return out.detach().clone(), x.grad.clone(), full, w


def test_gather_layer_forward_concat_and_local_gradient():
    """GatherLayer (reference src/utils/losses.py:19-37): every rank sees the rank-ordered concatenation; the gradient of a rank's own
    tensor is its slice of the upstream gradient (no cross-rank reduction), and a single process is the identity."""
    res = _spawn(_gather_job, world=2)
    for rank in (0, 1):
        out, gx, full, w = res[rank]
        assert torch.equal(out, full)
        assert torch.equal(gx, w[rank * 3:(rank + 1) * 3])
    import studiogan_amd  # noqa: F401
    from studiogan_amd import losses as SL
    t = torch.randn(4)
    assert SL.gather_logits(t, None) is t


class _FakeArena:
    def __init__(self, n):
        self.numel = n
        self.grad = torch.zeros(n)


class _FakeBank:
    """stands in for bank.WeightBank: records which arena ranges the plan asked to fold"""
    def __init__(self):
        self.folded = []

    def flush(self, lo=None, hi=None):
        self.folded.append((lo, hi))


def _exchange_plan_job(rank, world):
    """optim.ExchangePlan's range logic on CPU tensors over gloo: a network of five 'blocks' whose parameters sit at arena offsets
    0 / 1000 / 3000 / 3200 / 6000 (total 8000), two forwards per update (a discriminator update), min range 1500 elements."""
    import studiogan_amd  # noqa: F401
    from studiogan_amd.optim import ExchangePlan
    a = _FakeArena(8000)
    a.grad[:] = torch.arange(8000, dtype=torch.float32) * (rank + 1)      # rank r holds (r + 1) * i: the all-reduced sum is 3 i at world 2
    plan = ExchangePlan(a, min_elems=1500)
    bank = _FakeBank()
    bounds = [1000, 3000, 3200, 6000]            # offset of the first parameter BEHIND each block boundary
    for _ in range(2):                            # two forwards register the same boundaries
        for off in bounds:
            plan.expect(off)
    log = []
    # an unarmed plain backward (an earlier accumulation micro-step would look like this) only counts
    for off in reversed(bounds):
        plan.crossed(off, bank)
    assert plan.inflight == [] and bank.folded == []
    plan.reset()
    for _ in range(2):
        for off in bounds:
            plan.expect(off)
    plan.arm(dist.group.WORLD)
    # first forward's backward (the later-created graph runs first): nothing may be sent yet
    for off in reversed(bounds):
        plan.crossed(off, bank)
        log.append(len(plan.inflight))
    assert log == [0, 0, 0, 0] and bank.folded == []
    # second forward's backward: 6000 closes [6000, 8000) = 2000 >= 1500 -> sent; 3200 closes only [3200, 6000) = 2800 -> sent;
    # 3000 closes [3000, 3200) = 200 < 1500 -> waits; 1000 closes [1000, 3200) = 2200 -> sent (takes the waiting range along)
    sent = []
    for off in reversed(bounds):
        plan.crossed(off, bank)
        sent.append([(lo, hi) for lo, hi, _ in plan.inflight])
    assert sent[0] == [(6000, 8000)] and sent[1] == [(6000, 8000), (3200, 6000)] and sent[2] == sent[1]
    assert sent[3] == [(6000, 8000), (3200, 6000), (1000, 3200)]
    assert bank.folded == [(6000, 8000), (3200, 6000), (1000, 3200)], "each sent range is folded first, exactly once"
    early, rest = plan.take()
    assert rest == 1000                                       # the head of the arena is left for step()
    for lo, hi, work in early:
        work.wait()
    ref = torch.arange(8000, dtype=torch.float32)
    s = sum(r + 1 for r in range(world))
    ok_early = bool(torch.equal(a.grad[1000:], ref[1000:] * s))
    ok_head = bool(torch.equal(a.grad[:1000], ref[:1000] * (rank + 1)))     # untouched
    # a boundary at or behind what was already sent is ignored; reset re-arms cleanly
    plan.crossed(6000, bank)
    n_after = len(plan.inflight)
    plan.reset()
    return ok_early, ok_head, n_after, plan.done_lo, plan.armed


def test_exchange_plan_ranges_gloo():
    res = _spawn(_exchange_plan_job)
    for r in range(2):
        ok_early, ok_head, n_after, done_lo, armed = res[r]
        assert ok_early, "ranges sent from the backward must hold the all-reduced sum"
        assert ok_head, "the head of the arena must be left to step()"
        assert n_after == 3 and done_lo == 8000 and armed is False


def _emulated_step_job(rank, world):
    """one D + one G update (and two more with the EMA twin) of the width-8 ResNet fixture on `world` ranks, the kernels on the CPU interpreter"""
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path[:0] = [here, os.path.join(here, "hipemu")]
    import fullemu
    import test_dist_gpu as TD
    torch.set_num_threads(1)
    os.environ["SG_TEST_DEVICE"] = "cpu"
    with fullemu.Installed(dma_late=1, greedy=1, seed=5):
        out = TD._job(rank, world, "resgan32", False, steps=2)
    return {"state": {k: v.clone() for k, v in out["state"].items()}}


@pytest.mark.parametrize("sharded", ["0", "1"])
def test_two_ranks_emulated_training_step_equals_full_batch(sharded, monkeypatch):
    """The world > 1 PRODUCT path on the CPU: two gloo ranks run worker.Worker's updates with the kernel SOURCES on the interpreter (tests/hipemu) -- sync-BN statistics
    and backward terms all-reduced between the kernels, the gradient arena exchanged by FusedAdam (chunked all-reduce, or SG_SHARDED_ADAM=1: reduce-scatter emulation -> Adam on
    this rank's shard -> all-gather), the EMA twin -- against the single-rank run over the full batch: both ranks bit-identical, equal to the full batch up to the +-lr
    kicks of noise-gradient elements (the bounds of tests/test_dist_gpu.py, which runs the same job on the GPU)."""
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path[:0] = [here, os.path.join(here, "hipemu")]
    import emu
    if not emu.available():
        pytest.skip("host clang++ of the ROCm toolchain not found")
    import fullemu
    fullemu.build()                      # (once, in the parent: the ranks find the cached library)
    monkeypatch.setenv("SG_TEST_EMA", "1")
    monkeypatch.setenv("SG_SHARDED_ADAM", sharded)
    two = _spawn(_emulated_step_job, 2)
    monkeypatch.setenv("SG_SHARDED_ADAM", "0")
    full = _spawn(_emulated_step_job, 1)[0]
    a, b = two[0]["state"], two[1]["state"]
    assert any(k.startswith("G_ema/") for k in a)
    for k in a:
        assert torch.equal(a[k], b[k]), f"replicas diverged: {k}"
        if a[k].dtype.is_floating_point:
            err = float((a[k] - full["state"][k]).abs().max())
            assert err <= 2e-3 * max(float(full["state"][k].abs().max()), 0.05) + 2 * 2.2 * 2e-4, (k, err)
        else:
            assert torch.equal(a[k], full["state"][k]), k
