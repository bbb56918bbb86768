"""GPU: the world > 1 PRODUCT path on one device -- two ranks on cuda:0 over gloo (the collectives are torch.distributed's, the
kernels around them are the real ones): G / D forward + backward with synchronised batch norm (functional.BNFn's all-reduces of the
statistics and of the backward channel terms), FusedAdam.step(group=...) (replica check, pipelined all-reduce of the flat gradient
arena, 1/world folded into the Adam launch), sync_replicas -- compared with the single-rank run over the full batch:
  * after one D update and one G update both ranks hold BIT-IDENTICAL parameters and buffers,
  * they equal the full-batch single-rank result (gradient average == full-batch gradient of the mean-reduced loss, sync-BN
    statistics == full-batch statistics),
  * replicas that start from different weights are caught (assert_replicas_identical) and repaired (sync_replicas).
Plus the native RCCL entry points of the C ABI (sg_comm_*, sg_allreduce_flat, sg_bn_stats_sync) on a one-rank communicator."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(rank, world, port, name, desync, ret, steps=1, p2p=False):
    sys.path.insert(0, os.path.dirname(HERE))
    sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ret[rank] = _job(rank, world, name, desync, steps, p2p)
    finally:
        if world > 1:
            dist.destroy_process_group()


def _job(rank, world, name, desync, steps=1, p2p=False):
    import studiogan_amd  # noqa: F401
    from studiogan_amd import ops
    from studiogan_amd.worker import Worker
    from studiogan_amd.optim import sync_replicas
    from util import load_golden, sub, hyper
    from test_model_gpu import build_from_yaml
    dev = torch.device(os.environ.get("SG_TEST_DEVICE", "cuda:0"))      # ("cpu": the package bound to the interpreted library, tests/test_dist_cpu.py)
    sync = torch.cuda.synchronize if dev.type == "cuda" else (lambda: None)
    fix, meta = load_golden(name)
    y = meta["yaml"]
    G, D = build_from_yaml(y, False, dev)
    G.load_state_dict({k: v.to(dev) for k, v in sub(fix, "G_init/").items()}, strict=True)
    D.load_state_dict({k: v.to(dev) for k, v in sub(fix, "D_init/").items()}, strict=True)
    group = dist.group.WORLD if world > 1 else None
    box = None
    if p2p and world > 1:
        # sync-BN through the peer-store mailboxes (csrc/p2p.hip): the exchange runs INSIDE the finalize kernel, over IPC-mapped device memory -- the two
        # processes share one GPU here, the kernels and the protocol are the ones N GPUs run over xGMI. (gloo only hands the 64-byte handles around.)
        from studiogan_amd import comm
        box = comm.enable_p2p(group)
        v = torch.arange(1, 1001, dtype=torch.float64, device=dev) * (rank + 1) + 0.125 * rank
        for _ in range(5):                     # back-to-back calls: the two-slot epoch protocol
            box.allreduce_f64_(v)
        sync()
        p2p_vec = v.cpu()
    caught = None
    if desync and rank == 1:                       # what per-rank seeding does to an un-broadcast model (reference src/loader.py:99)
        with torch.no_grad():
            for p in D.parameters():
                p.add_(0.01 * torch.randn_like(p))
    if desync and world > 1:
        from studiogan_amd.optim import assert_replicas_identical, _arena_for
        try:
            assert_replicas_identical(_arena_for(list(D.parameters())).data, group, "D")
            caught = False
        except RuntimeError:
            caught = True
    if world > 1:
        for m in list(G.modules()) + list(D.modules()):
            if isinstance(m, ops.BatchNorm2d):
                m.sync_group = True
    opt = hyper(y)
    w = Worker(G, D, opt["z_dim"], y["DATA"]["num_classes"], meta["batch"] // world, opt["adv_loss"], opt["g_lr"], opt["d_lr"], opt["beta1"],
               opt["beta2"], d_updates_per_step=1, apply_g_ema=os.environ.get("SG_TEST_EMA") == "1", g_ema_decay=0.9, g_ema_start=0,
               group=group)      # Worker.__init__ runs sync_replicas(G / D, group)
    ins = sub(fix, "in/")
    h = meta["batch"] // world
    sl = slice(rank * h, (rank + 1) * h)
    put = lambda k: ins[k][sl].to(dev)
    w.train_discriminator(0, [(put("real0"), put("rl0"))], [(put("z0"), put("fl0"))])
    d_grad = {k: p.grad.detach().cpu().clone() / world for k, p in D.named_parameters()}      # the arena holds the all-reduced SUM
    w.train_generator(0, [(put("z1"), put("fl1"))])
    sync()
    g_grad = {k: p.grad.detach().cpu().clone() / world for k, p in G.named_parameters()}
    early = None
    if steps > 1:
        # further updates: from the second one on the block boundaries put finished gradient ranges on the wire DURING the backward
        # (optim.ExchangePlan; the first exchange stays in step() because it also verifies the replicas)
        for it in range(1, steps):
            w.train_discriminator(it, [(put("real0"), put("rl0"))], [(put(f"z{it % 2}"), put(f"fl{it % 2}"))])
            w.train_generator(it, [(put("z1"), put("fl1"))])
        sync()
        early = {"D": dict(w.d_optimizer.exchange_stats), "G": dict(w.g_optimizer.exchange_stats)}
    state = {"D/" + k: v.detach().cpu().clone() for k, v in D.state_dict().items()}
    state.update({"G/" + k: v.detach().cpu().clone() for k, v in G.state_dict().items()})
    if w.Gen_ema is not None:
        state.update({"G_ema/" + k: v.detach().cpu().clone() for k, v in w.Gen_ema.state_dict().items()})
    out = {"state": state, "d_grad": d_grad, "g_grad": g_grad, "caught": caught, "early": early}
    if box is not None:
        out["p2p_vec"], out["p2p_timeouts"] = p2p_vec, box.timeouts()
        from studiogan_amd import comm
        comm.disable_all()
    return out


def _spawn(world, name, desync=False, steps=1, p2p=False):
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_run, args=(world, _free_port(), name, desync, ret, steps, p2p), nprocs=world, join=True)
    return dict(ret)


@pytest.mark.parametrize("name", ["resgan32", "biggan32"])     # BN in G and D (no SN) / cBN in G + SN + attention + projection D
def test_two_ranks_on_one_gpu_equal_the_full_batch_step(sg, name):
    from util import Collector
    full = _spawn(1, name)[0]
    two = _spawn(2, name)
    a, b = two[0], two[1]
    for k in a["state"]:
        assert torch.equal(a["state"][k], b["state"][k]), f"replicas diverged: {k}"
    C = Collector()
    gm = max(float(v.abs().max()) for v in full["d_grad"].values())
    for k, v in full["d_grad"].items():
        C.check("D grad " + k, a["d_grad"][k], v, 2e-4, floor=1e-2 * gm)
    gm = max(float(v.abs().max()) for v in full["g_grad"].values())
    for k, v in full["g_grad"].items():
        C.check("G grad " + k, a["g_grad"][k], v, 5e-3, floor=1e-2 * gm)      # through every ReLU of D after one D update (see DESIGN "conditioning")
    from util import load_golden, hyper
    lr = max(hyper(load_golden(name)[1]["yaml"])[k] for k in ("g_lr", "d_lr"))
    for k, v in full["state"].items():
        if v.dtype.is_floating_point:
            # (an element whose gradient is rounding noise -- e.g. a conv bias in front of a BN -- moves by +-lr with either sign)
            C.check("state " + k, a["state"][k], v, 2e-3, floor=0.05, abs_ok=2.2 * lr)
        else:
            assert torch.equal(a["state"][k], v), k
    C.finish()


@pytest.mark.parametrize("name", ["resgan32", "biggan32"])
def test_two_ranks_sync_bn_fused_peer_store_exchange(sg, name):
    """The north star's "cross-rank mean/var reduction fused into the BN kernel" (csrc/p2p.hip): two PROCESSES on this GPU map each other's mailbox through IPC handles;
    functional.BNFn's statistics go out as peer stores from inside sg_bn_finalize_p2p's kernel, the backward's channel terms through sg_p2p_allreduce_f64 -- no collective
    library in the sync-BN path. Checked: five back-to-back tiny all-reduces are exact (epoch / two-slot protocol), no granule wait ran into its limit, both ranks end
    BIT-identical, and the two-rank step equals the full-batch single-rank step (the bounds of test_two_ranks_on_one_gpu_equal_the_full_batch_step)."""
    from util import Collector, load_golden, hyper
    full = _spawn(1, name)[0]
    two = _spawn(2, name, p2p=True)
    a, b = two[0], two[1]
    base = torch.arange(1, 1001, dtype=torch.float64)
    want = (base * 1 + 0.0) + (base * 2 + 0.125)
    for _ in range(4):
        want = want * 2
    assert torch.equal(a["p2p_vec"], want) and torch.equal(b["p2p_vec"], want)
    assert a["p2p_timeouts"] == 0 and b["p2p_timeouts"] == 0
    for k in a["state"]:
        assert torch.equal(a["state"][k], b["state"][k]), f"replicas diverged: {k}"
    C = Collector()
    gm = max(float(v.abs().max()) for v in full["d_grad"].values())
    for k, v in full["d_grad"].items():
        C.check("D grad " + k, a["d_grad"][k], v, 2e-4, floor=1e-2 * gm)
    gm = max(float(v.abs().max()) for v in full["g_grad"].values())
    for k, v in full["g_grad"].items():
        C.check("G grad " + k, a["g_grad"][k], v, 5e-3, floor=1e-2 * gm)
    lr = max(hyper(load_golden(name)[1]["yaml"])[k] for k in ("g_lr", "d_lr"))
    for k, v in full["state"].items():
        if v.dtype.is_floating_point:
            C.check("state " + k, a["state"][k], v, 2e-3, floor=0.05, abs_ok=2.2 * lr)
        else:
            assert torch.equal(a["state"][k], v), k
    C.finish()


@pytest.mark.parametrize("name", ["biggan32", "resgan32"])
def test_sharded_optimizer_step_equals_the_allreduce_step(sg, name, monkeypatch):
    """FusedAdam's sharded step (SG_SHARDED_ADAM=1: reduce-scatter -> Adam on this rank's 1/world of the arena -> all-gather of the updated parameters, EMA lerped
    over the gathered arena) against the all-reduce step, two ranks, three updates of both networks with the EMA twin on: the two RANKS bit-identical in every
    parameter, buffer and EMA parameter; the two SCHEMES equal up to what two separate runs of one scheme differ by (fp64 atomics in the BN statistics make a run
    not bit-reproducible; an element whose gradient is rounding noise then moves by +-lr per update with either sign -- the bound of
    test_early_gradient_exchange_matches_the_exchange_in_step). Adam is element-wise, so the shard boundaries themselves change nothing."""
    monkeypatch.setenv("SG_TEST_EMA", "1")
    monkeypatch.setenv("SG_SHARDED_ADAM", "1")
    on = _spawn(2, name, steps=3)
    monkeypatch.setenv("SG_SHARDED_ADAM", "0")
    off = _spawn(2, name, steps=3)
    assert any(k.startswith("G_ema/") for k in on[0]["state"])
    n_diff = 0
    for k in on[0]["state"]:
        assert torch.equal(on[0]["state"][k], on[1]["state"][k]), f"replicas diverged under the sharded step: {k}"
        a, b = on[0]["state"][k], off[0]["state"][k]
        if not torch.equal(a, b):
            # two separate runs are not bit-identical run to run (fp64 atomics in the BN statistics): bounded like test_early_gradient_exchange
            n_diff += 1
            if a.dtype.is_floating_point:
                assert float((a - b).abs().max()) <= 2e-3 * max(float(b.abs().max()), 0.05) + 3 * 2.2 * 2e-4, k
            else:
                raise AssertionError(k)
    print("tensors not bit-identical between the two schemes:", n_diff, "of", len(on[0]["state"]))


def test_desynchronised_replicas_are_caught_and_repaired(sg):
    two = _spawn(2, "sngan32", desync=True)
    assert two[0]["caught"] is True and two[1]["caught"] is True, "assert_replicas_identical must flag replicas that differ"
    for k in two[0]["state"]:
        assert torch.equal(two[0]["state"][k], two[1]["state"][k]), f"sync_replicas (Worker.__init__) must leave identical replicas: {k}"


def test_native_rccl_entry_points_single_rank(sg):
    """sg_comm_unique_id / sg_comm_init_rank / sg_allreduce_flat / sg_bn_stats_sync on a communicator of ONE rank (the 1-GPU box cannot
    host two RCCL ranks on one device): the all-reduce is the identity, sg_bn_stats_sync equals partial_stats + finalize."""
    from studiogan_amd import comm, _lib as L
    dev = torch.device("cuda:0")
    nc = comm.NativeComm()
    assert nc.world == 1 and nc.handle
    t = torch.randn(100003, device=dev)
    ref = t.clone()
    nc.allreduce_(t)
    d = torch.randn(77, device=dev, dtype=torch.float64)
    dref = d.clone()
    nc.allreduce_(d)
    torch.cuda.synchronize()
    assert torch.equal(t, ref) and torch.equal(d, dref)
    x = (torch.randn(6, 5, 7, 24, device=dev) * 1.5 + 0.2).to(torch.bfloat16)
    Cc, rows = 24, 6 * 5 * 7
    out = []
    for native in (True, False):
        mean, invstd = torch.empty(Cc, device=dev), torch.empty(Cc, device=dev)
        rm, rv = torch.zeros(Cc, device=dev), torch.ones(Cc, device=dev)
        if native:
            partial = torch.empty(2 * Cc, dtype=torch.float64, device=dev)
            L.call("sg_bn_stats_sync", L.dt(x), x.data_ptr(), Cc, rows, Cc, partial.data_ptr(), nc.handle, 1e-4, 0.1, mean.data_ptr(), invstd.data_ptr(),
                   rm.data_ptr(), rv.data_ptr(), L.stream())
        else:
            partial = torch.zeros(2 * Cc, dtype=torch.float64, device=dev)
            L.call("sg_bn_partial_stats", L.dt(x), x.data_ptr(), Cc, rows, Cc, partial.data_ptr(), L.stream())
            L.call("sg_bn_finalize", partial.data_ptr(), float(rows), Cc, 1e-4, 0.1, mean.data_ptr(), invstd.data_ptr(), rm.data_ptr(), rv.data_ptr(), L.stream())
        torch.cuda.synchronize()
        out.append((mean.cpu(), invstd.cpu(), rm.cpu(), rv.cpu()))
    for u, v in zip(*out):
        assert torch.equal(u, v)
    xf = x.float().reshape(-1, Cc).double().cpu()
    assert torch.allclose(out[0][0].double(), xf.mean(0), atol=1e-5)
    nc.close()


@pytest.mark.parametrize("name", ["biggan32", "resgan32"])
def test_early_gradient_exchange_matches_the_exchange_in_step(sg, name, monkeypatch):
    """optim.ExchangePlan: three D + G updates on two ranks with the all-reduce of finished arena ranges issued from inside the backward
    pass (block-boundary marks, bank.GradReadyFn) against the same updates with the whole exchange in FusedAdam.step (SG_EARLY_EXCHANGE=0):
    the two ranks bit-identical to each other, the result equal to the in-step exchange up to the +-lr kicks of noise-gradient elements
    (tests/test_dist_gpu.py::test_exchange_ranges_are_final_when_sent checks tensor by tensor that a sent range never changes afterwards),
    and the early path really ran (ranges sent from the backward of the second and
    third update of both networks)."""
    monkeypatch.setenv("SG_EXCHANGE_MIN_ELEMS", "256")        # width-8 networks: let every boundary that closes 256 gradients send
    monkeypatch.setenv("SG_EARLY_EXCHANGE", "1")
    on = _spawn(2, name, steps=3)
    monkeypatch.setenv("SG_EARLY_EXCHANGE", "0")
    off = _spawn(2, name, steps=3)
    from util import Collector, load_golden, hyper
    lr = max(hyper(load_golden(name)[1]["yaml"])[k] for k in ("g_lr", "d_lr"))
    C = Collector()
    for k in on[0]["state"]:
        assert torch.equal(on[0]["state"][k], on[1]["state"][k]), f"replicas diverged with the early exchange: {k}"
        a, b = on[0]["state"][k], off[0]["state"][k]
        if a.dtype.is_floating_point:
            # two separate runs are not bit-identical (fp64 atomics in the BN statistics): an element whose gradient is rounding noise -- a conv
            # bias in front of a batch norm -- moves by +-lr per update with either sign, everything else agrees to rounding
            C.check("early vs in-step " + k, a, b, 2e-3, floor=0.05, abs_ok=3 * 2.2 * lr)
        else:
            assert torch.equal(a, b), k
    C.finish()
    for net in ("D", "G"):
        st = on[0]["early"][net]
        assert st["early_ranges"] >= 2 and st["early_elems"] > 0, (net, st)
        assert off[0]["early"][net]["early_ranges"] == 0


@pytest.mark.parametrize("name", ["biggan32", "resgan32", "bigdeep32", "sngan32"])
def test_exchange_ranges_are_final_when_sent(sg, name, monkeypatch):
    """optim.ExchangePlan in its single-process self-test mode (SG_EXCHANGE_SELFTEST=1): every arena range that a block boundary would put on the
    wire during the backward pass is snapshotted instead, and FusedAdam.step raises if any gradient in it changed afterwards -- i.e. the
    'everything behind this boundary is final' claim is checked tensor by tensor, for D (two forwards per update) and G."""
    from studiogan_amd.worker import Worker
    from util import load_golden, sub, hyper
    from test_model_gpu import build_from_yaml
    monkeypatch.setenv("SG_EXCHANGE_SELFTEST", "1")
    monkeypatch.setenv("SG_EXCHANGE_MIN_ELEMS", "64")
    dev = torch.device("cuda:0")
    fix, meta = load_golden(name)
    y = meta["yaml"]
    G, D = build_from_yaml(y, False, dev)
    G.load_state_dict({k: v.to(dev) for k, v in sub(fix, "G_init/").items()}, strict=True)
    D.load_state_dict({k: v.to(dev) for k, v in sub(fix, "D_init/").items()}, strict=True)
    opt = hyper(y)
    w = Worker(G, D, opt["z_dim"], y["DATA"]["num_classes"], meta["batch"], opt["adv_loss"], opt["g_lr"], opt["d_lr"], opt["beta1"], opt["beta2"],
               d_updates_per_step=1, apply_g_ema=False)
    ins = {k: v.to(dev) for k, v in sub(fix, "in/").items()}
    for it in range(3):
        w.train_discriminator(it, [(ins["real0"], ins["rl0"])], [(ins["z0"], ins["fl0"])])
        w.train_generator(it, [(ins["z1"], ins["fl1"])])
    torch.cuda.synchronize()
    for net, o in (("D", w.d_optimizer), ("G", w.g_optimizer)):
        assert o.exchange_stats["early_ranges"] >= 2, (net, o.exchange_stats)
