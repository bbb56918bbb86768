"""Fused Adam (+EMA) over the flat parameter arena and the explicit data-parallel gradient exchange.

Replaces torch.optim.Adam(eps=1e-6) (reference src/config.py:541-563), utils/ema.py:27-40 and DistributedDataParallel's
bucketed all-reduce (reference src/models/model.py:171-180): gradients of a network already sit in ONE contiguous fp32
buffer, so the exchange is a few large RCCL all-reduces issued once per update (not per accumulation micro-step),
pipelined against the optimizer kernel chunk by chunk.
"""
import os

import torch
import torch.distributed as dist

from . import _lib as L
from . import comm as _comm
from .bank import ParamArena, arena_of, get_buffer_arena


def chunk_ranges(n, nch, min_chunk=1 << 20):
    """Split [0, n) into at most nch contiguous 16-byte-aligned ranges of >= min_chunk elements."""
    nch = max(1, min(nch, n // min_chunk or 1))
    per = ((n + nch - 1) // nch + 3) // 4 * 4
    return [(lo, min(n, lo + per)) for lo in range(0, n, per)]


def pipelined_allreduce(flat, ranges, group=None):
    """Issue one async all-reduce(sum) per range up front (they queue on the RCCL stream back to back) and yield each
    range as soon as ITS reduction is complete, so the consumer (the fused Adam launch of that slice) overlaps with the
    ranges still on the wire."""
    works = [(lo, hi, dist.all_reduce(flat[lo:hi], group=group, async_op=True)) for lo, hi in ranges]
    for lo, hi, w in works:
        with _comm.exposed():
            w.wait()
        yield lo, hi


def _fingerprint(flat):
    """[sum, sum of squares, weighted sum] of a flat fp32 buffer in fp64: equal on two ranks iff (for all practical purposes) the buffers are."""
    d = flat.double()
    w = torch.arange(1, d.numel() + 1, dtype=torch.float64, device=d.device) / d.numel()
    return torch.stack([d.sum(), (d * d).sum(), (d * w).sum()])


def assert_replicas_identical(flat, group=None, what="buffer"):
    """Raise if `flat` differs across the ranks of `group` (one tiny MIN/MAX all-reduce pair)."""
    fp = _fingerprint(flat)
    lo, hi = fp.clone(), fp.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=group)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=group)
    if not bool(torch.equal(lo, hi)):
        raise RuntimeError(f"data-parallel replicas disagree: {what} differ across ranks. Without DistributedDataParallel nothing "
                           "broadcasts the initial weights -- call studiogan_amd.optim.sync_replicas(module, group) after building "
                           "each network (INTEGRATION.md §5), or seed every rank identically before construction.")


@torch.no_grad()
def sync_replicas(module, group=None, src=0):
    """What DistributedDataParallel's constructor does for the reference (src/models/model.py:171-180): every rank takes rank
    `src`'s parameters AND buffers (spectral-norm u / v, BN running statistics). Two broadcasts: the flat parameter arena and the
    flat buffer arena; integer buffers (num_batches_tracked) one by one."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return module
    a = _arena_for(list(module.parameters()))
    root = src if group is None else dist.get_global_rank(group, src)
    dist.broadcast(a.data, src=root, group=group)
    b = get_buffer_arena(module)
    if b.numel:
        dist.broadcast(b.data, src=root, group=group)
    for _, buf in module.named_buffers():
        if buf is not None and buf.dtype != torch.float32:
            dist.broadcast(buf, src=root, group=group)
    return module


class ExchangePlan:
    """Early data-parallel gradient exchange of ONE network: the all-reduce of a finished tail of the flat gradient arena is issued from
    inside the backward pass, as soon as the pass has come back across a block boundary that every forward of this update has passed
    (bank.GradReadyFn), and overlaps the backward of the blocks in front of it; FusedAdam.step then only waits for those reductions and
    exchanges what is left. Replaces DistributedDataParallel's bucketed all-reduce during backward (reference src/models/model.py:171-180).

    Parameters sit in the arena in module order = forward order (block granularity), the backward visits them tail first, so the finished
    part is always [offset of the boundary, start of what was already exchanged). A range is sent once it holds >= min_elems gradients
    (few, large collectives: xGMI is per-link bound). Gradient penalties (create_graph passes), gradient accumulation micro-steps and anything
    else that does not arm the plan fall back to the exchange inside step()."""

    def __init__(self, arena, min_elems=8 << 20):
        self.arena, self.min_elems = arena, int(os.environ.get("SG_EXCHANGE_MIN_ELEMS", min_elems))
        self.group = None
        self.reset()

    def reset(self):
        # an all-reduce still in flight (a backward that sent ranges but whose step() never ran: exception, skipped step, custom loop) must
        # not race with the zeroing / next accumulation of the gradients it reads
        for _, _, handle in getattr(self, "inflight", ()):
            if isinstance(handle, torch.cuda.Event):
                torch.cuda.current_stream().wait_event(handle)
            else:
                handle.wait()
        self.armed = False
        self.expected, self.seen = {}, {}
        self.done_lo = self.arena.numel
        self.inflight = []          # (lo, hi, handle): torch.distributed Work or a torch.cuda.Event on the native communicator's stream
        self.issued_in_backward = 0
        self.selftest = os.environ.get("SG_EXCHANGE_SELFTEST") == "1"      # one process: no collective, snapshot the range instead; step() then
        self.snapshots = []                                                # checks that nothing was written into it afterwards

    def expect(self, off):
        self.expected[off] = self.expected.get(off, 0) + 1

    def arm(self, group):
        self.armed, self.group = True, group

    def crossed(self, off, bank):
        self.seen[off] = self.seen.get(off, 0) + 1          # (counted for every plain backward: accumulation micro-steps before the armed one)
        if not self.armed:
            return
        if self.seen[off] < self.expected.get(off, 0) or off >= self.done_lo:
            return                                          # another forward of this update still has to come back across this boundary
        lo, hi = off, self.done_lo
        if hi - lo < self.min_elems and lo > 0:
            return                                          # too small to be worth a collective of its own: rides with the next boundary
        bank.flush(lo, hi)                                  # spectral-norm backward of the finished layers -> gradient arena
        g = self.arena.grad[lo:hi]
        nc = _comm.native_for(self.group)
        if self.selftest:
            self.snapshots.append((lo, hi, g.clone()))
        elif nc is not None:
            main = torch.cuda.current_stream()
            ready = torch.cuda.Event()
            ready.record(main)
            nc.stream.wait_event(ready)
            nc.allreduce_(g, stream=nc.stream.cuda_stream, grad=True)
            done = torch.cuda.Event()
            done.record(nc.stream)
            self.inflight.append((lo, hi, done))
        else:
            self.inflight.append((lo, hi, dist.all_reduce(g, group=self.group, async_op=True)))
        self.done_lo = lo
        self.issued_in_backward += 1

    def take(self):
        """-> (ranges already on the wire [(lo, hi, handle)], end of the part step() still has to exchange)"""
        out, rest = self.inflight, self.done_lo
        return out, rest


def _arena_for(params):
    ents = [arena_of(p) for p in params]
    if any(e is None for e in ents) or len({id(e[0]) for e in ents}) != 1 or not ents[0][0].intact():
        return ParamArena(params)
    return ents[0][0]


class FusedAdam(torch.optim.Optimizer):
    """torch.optim.Adam semantics (no amsgrad) for ALL parameters of one network in one launch."""

    def __init__(self, params, lr=2e-4, betas=(0.5, 0.999), eps=1e-6, weight_decay=0.0, comm_chunks=4, sharded=None):
        defaults = dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        assert len(self.param_groups) == 1, "one parameter group per network"
        self._arena = None
        self._m = self._v = None
        self._t = 0
        self.comm_chunks = comm_chunks
        # sharded step (SG_SHARDED_ADAM=1 or sharded=True; world > 1 only): reduce-scatter of the gradient arena -> Adam on this rank's 1/world of it -> all-gather
        # of the updated parameters, the gather (and the EMA lerp behind it) left in flight behind the next forward of the OTHER network. Same bytes on the wire as
        # the all-reduce, half of them off the critical path, 1/world of the optimizer's HBM traffic per rank.
        self.sharded = (os.environ.get("SG_SHARDED_ADAM") == "1") if sharded is None else bool(sharded)
        self._replicas_checked = False
        self._module = None
        self._plan = None
        self.exchange_stats = {"early_ranges": 0, "early_elems": 0, "late_elems": 0}

    def attach(self, module):
        """Tell the optimizer which network its parameters belong to: under data parallelism the network's block boundaries then start the
        gradient all-reduce of finished arena ranges during the backward pass (ExchangePlan). Optional -- without it the whole exchange
        happens inside step()."""
        self._module = module
        return self

    def arm_exchange(self, group):
        """Call right before the backward of the LAST accumulation micro-step of an update (Worker does): from now on a block boundary the
        backward comes back to may send its finished gradients.
        OPT-IN (SG_EARLY_EXCHANGE=1): the path has only ever run with two gloo ranks on one device. On the native path it puts gradient
        all-reduces on a second RCCL communicator while sync-BN all-reduces run on the first -- two communicators in flight with no cross-rank
        ordering is a documented hang hazard -- so until a multi-GPU run has shown it green the default is the exchange inside step()
        (chunked all-reduce overlapped with the optimizer launches), which has no collective in flight next to another one."""
        selftest = os.environ.get("SG_EXCHANGE_SELFTEST") == "1"
        if self._module is None or (group is None and not selftest) or (os.environ.get("SG_EARLY_EXCHANGE", "0") != "1" and not selftest):
            return False
        if not selftest and (not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) < 2):
            return False
        h = self._module.__dict__.get("_sg_bank_holder")
        bank = h.obj if h is not None else None
        a = self._state()
        if bank is None or bank.params is not a:
            return False
        if self._plan is None or self._plan.arena is not a:
            self._plan = ExchangePlan(a)
        if bank.exchange is not self._plan:
            bank.exchange = self._plan        # boundaries of forwards run from now on register with the plan
            self._plan.reset()
            return False                      # (this update's forwards ran without marks: plain exchange in step())
        if not self._replicas_checked and not selftest:
            return False                      # the first exchange also verifies the replicas: keep it in step()
        self._plan.arm(group)
        return True

    def _selftest_check(self, a):
        """SG_EXCHANGE_SELFTEST=1 (single process): every range the backward would have put on the wire must still hold exactly the values it
        held at that moment -- i.e. its gradients really were final. Raises with the names of the parameters that changed."""
        plan = self._plan
        if plan is None or not plan.snapshots:
            return
        bad = []
        for lo, hi, snap in plan.snapshots:
            cur = a.grad[lo:hi]
            if not torch.equal(cur, snap):
                for p, o in zip(a.params, a.offsets):
                    if lo <= o < hi and not torch.equal(a.grad[o:o + p.numel()], snap[o - lo:o - lo + p.numel()]):
                        name = next((n for n, q in self._module.named_parameters() if q is p), "?") if self._module is not None else "?"
                        bad.append(name)
        self.exchange_stats["early_ranges"] += len(plan.snapshots)
        self.exchange_stats["early_elems"] += sum(h - l for l, h, _ in plan.snapshots)
        plan.reset()
        if bad:
            raise RuntimeError("early gradient exchange: these parameters received gradient AFTER their arena range was sent: " + ", ".join(bad[:12]))

    def _state(self):
        params = self.param_groups[0]["params"]
        a = _arena_for(params)
        if a is not self._arena:
            m = torch.zeros_like(a.data)
            v = torch.zeros_like(a.data)
            if self._arena is not None:  # arena was rebuilt (module.to()/deepcopy): carry the moments over
                old = self._arena
                for p, o_new in zip(a.params, a.offsets):
                    for q, o_old in zip(old.params, old.offsets):
                        if q is p:
                            m[o_new:o_new + p.numel()] = self._m[o_old:o_old + p.numel()]
                            v[o_new:o_new + p.numel()] = self._v[o_old:o_old + p.numel()]
            self._arena, self._m, self._v = a, m, v
        return a

    # -- checkpoint layout of torch.optim.Adam (reference src/utils/ckpt.py saves optimizer.state_dict() and restores it with
    #    load_state_dict): state[i] = {step, exp_avg, exp_avg_sq} per parameter. The moments live in two flat arenas here, so
    #    state_dict() emits per-parameter copies of the arena slices and load_state_dict() copies them back.
    def state_dict(self):
        a = self._state()
        a.wait_ready()
        step = torch.tensor(float(self._t), dtype=torch.float32)
        for p, o in zip(a.params, a.offsets):
            n = p.numel()
            self.state[p] = {"step": step.clone(), "exp_avg": self._m[o:o + n].view(p.shape).clone(),
                             "exp_avg_sq": self._v[o:o + n].view(p.shape).clone()}
        try:
            return super().state_dict()
        finally:
            self.state.clear()          # the arenas stay the single source of truth

    def load_state_dict(self, state_dict):
        a = self._state()
        super().load_state_dict(state_dict)      # torch's own validation / device + dtype casting of the per-parameter tensors
        steps = set()
        with torch.no_grad():
            for p, o in zip(a.params, a.offsets):
                st = self.state.get(p)
                n = p.numel()
                if not st:                       # a parameter torch.optim.Adam never stepped (no gradient yet): zero moments
                    self._m[o:o + n].zero_()
                    self._v[o:o + n].zero_()
                    continue
                self._m[o:o + n].copy_(st["exp_avg"].reshape(-1))
                self._v[o:o + n].copy_(st["exp_avg_sq"].reshape(-1))
                steps.add(int(float(st["step"])))
        if len(steps) > 1:
            raise RuntimeError(f"FusedAdam keeps one step count per network; the checkpoint holds {sorted(steps)}")
        self._t = steps.pop() if steps else 0
        self.state.clear()

    @torch.no_grad()
    def clamp_(self, bound):
        """Weight clipping of every parameter of the network to [-bound, bound] (reference src/worker.py:489-492) as one pass over the parameter arena."""
        a = self._state()
        a.wait_ready()
        L.call("sg_clamp_flat", a.data.data_ptr(), a.numel, -float(bound), float(bound), L.stream())

    def zero_grad(self, set_to_none=False):
        a = self._state()
        if self._plan is not None:
            self._plan.reset()
        a.grad.zero_()
        for p, o in zip(a.params, a.offsets):
            p.grad = a.grad[o:o + p.numel()].view(p.shape)

    @torch.no_grad()
    def step(self, closure=None, ema=None, iteration=None, group=None):
        """ema: optional `Ema` whose target shares this network's arena layout -> fused into the same launch."""
        a = self._state()
        a.wait_ready()
        g = self.param_groups[0]
        for p, o in zip(a.params, a.offsets):
            if p.grad is None or p.grad.data_ptr() != a.grad.data_ptr() + 4 * o:
                # a foreign gradient tensor (e.g. produced by plain autograd): fold it into the arena
                if p.grad is not None:
                    a.grad[o:o + p.numel()].copy_(p.grad.reshape(-1))
                p.grad = a.grad[o:o + p.numel()].view(p.shape)
        self._t += 1
        ema_ptr, decay = None, 0.0
        if ema is not None:
            ema_ptr = ema.target_arena().data.data_ptr()
            decay = ema.decay_at(iteration)
        world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
        n = a.numel
        st = L.stream()
        if world > 1 and not self._replicas_checked:
            # DDP's constructor broadcast is gone with DDP (reference src/models/model.py:171-180): replicas that start from
            # different weights would silently never converge to one model. Checked once, on the first exchange.
            assert_replicas_identical(a.data, group, "parameters of the network handed to FusedAdam")
            self._replicas_checked = True
        nc = _comm.native_for(group) if world > 1 else None
        if world == 1 and self._plan is not None and self._plan.selftest:
            self._selftest_check(a)
        if world > 1 and self._plan is not None and self._plan.inflight:
            # ranges whose all-reduce was issued from inside the backward pass: wait for each, apply Adam to it, then fall through with n
            # shortened to what is still to be exchanged
            early, rest = self._plan.take()
            main = torch.cuda.current_stream()
            for lo, hi, handle in early:
                with _comm.exposed():
                    if isinstance(handle, torch.cuda.Event):
                        main.wait_event(handle)
                    else:
                        handle.wait()
                L.call("sg_adam_ema", a.data.data_ptr() + 4 * lo, a.grad.data_ptr() + 4 * lo, self._m.data_ptr() + 4 * lo,
                       self._v.data_ptr() + 4 * lo, (ema_ptr + 4 * lo) if ema_ptr else None, hi - lo, g["lr"], g["betas"][0], g["betas"][1],
                       g["eps"], g["weight_decay"], self._t, decay, 1.0 / world, st)
                self.exchange_stats["early_ranges"] += 1
                self.exchange_stats["early_elems"] += hi - lo
            n = rest
        if self._plan is not None and (self._plan.armed or self._plan.expected or self._plan.seen):
            self._plan.inflight = []         # (waited for above)
            self._plan.reset()               # a plan that sent nothing must not stay armed into the other network's update either
        if world > 1:
            self.exchange_stats["late_elems"] += n
        if n == 0:
            pass
        elif world > 1 and self.sharded and n == a.numel:
            self._step_sharded(a, g, world, group, nc, ema, decay, st)
        elif nc is not None:
            # the same pipeline through the C ABI (sg_allreduce_flat): the reductions queue on the communicator's side stream behind
            # an event that marks "gradients complete"; the Adam launch of chunk i waits for ITS reduction only
            ranges = chunk_ranges(n, self.comm_chunks)
            main = torch.cuda.current_stream()
            ready = torch.cuda.Event()
            ready.record(main)
            nc.stream.wait_event(ready)
            done = []
            for lo, hi in ranges:
                nc.allreduce_(a.grad[lo:hi], stream=nc.stream.cuda_stream, grad=True)
                ev = torch.cuda.Event()
                ev.record(nc.stream)
                done.append(ev)
            for (lo, hi), ev in zip(ranges, done):
                with _comm.exposed():
                    main.wait_event(ev)
                L.call("sg_adam_ema", a.data.data_ptr() + 4 * lo, a.grad.data_ptr() + 4 * lo, self._m.data_ptr() + 4 * lo,
                       self._v.data_ptr() + 4 * lo, (ema_ptr + 4 * lo) if ema_ptr else None, hi - lo, g["lr"], g["betas"][0], g["betas"][1],
                       g["eps"], g["weight_decay"], self._t, decay, 1.0 / world, st)
        elif world > 1:
            # pipelined all-reduce(sum) -> Adam(grad/world): chunk i+1 is on the wire while chunk i is being applied
            for lo, hi in pipelined_allreduce(a.grad, chunk_ranges(n, self.comm_chunks), group):
                L.call("sg_adam_ema", a.data.data_ptr() + 4 * lo, a.grad.data_ptr() + 4 * lo, self._m.data_ptr() + 4 * lo,
                       self._v.data_ptr() + 4 * lo, (ema_ptr + 4 * lo) if ema_ptr else None, hi - lo, g["lr"], g["betas"][0], g["betas"][1],
                       g["eps"], g["weight_decay"], self._t, decay, 1.0 / world, st)
        else:
            L.call("sg_adam_ema", a.data.data_ptr(), a.grad.data_ptr(), self._m.data_ptr(), self._v.data_ptr(), ema_ptr, n, g["lr"],
                   g["betas"][0], g["betas"][1], g["eps"], g["weight_decay"], self._t, decay, 1.0, st)
        if ema is not None:
            ema.update_buffers(decay)


    def _step_sharded(self, a, g, world, group, nc, ema, decay, st):
        """reduce-scatter -> Adam on this rank's shard -> all-gather (see __init__). The arena is cut into world ranges of `per` elements (16-byte aligned) plus a
        tail of < 4 * world elements that every rank reduces and updates redundantly. Native communicator: both collectives on its side stream, the gather and the
        EMA lerp stay in flight (ParamArena.defer); torch.distributed: reduce_scatter_tensor / all_gather_into_tensor where the backend has them (nccl = RCCL), an
        all-reduce + all_gather otherwise (gloo: the CPU / one-device tests)."""
        n = a.numel
        rank = dist.get_rank(group)
        per = (n // (4 * world)) * 4
        body = per * world
        lo, hi = rank * per, (rank + 1) * per
        main = torch.cuda.current_stream() if a.data.is_cuda else None
        hyper = (g["lr"], g["betas"][0], g["betas"][1], g["eps"], g["weight_decay"], self._t)

        def adam(lo_, hi_):
            if hi_ > lo_:
                L.call("sg_adam_ema", a.data.data_ptr() + 4 * lo_, a.grad.data_ptr() + 4 * lo_, self._m.data_ptr() + 4 * lo_, self._v.data_ptr() + 4 * lo_, None,
                       hi_ - lo_, *hyper, 0.0, 1.0 / world, st)
        ema_args = None
        if ema is not None:
            ema_args = (a.data.data_ptr(), ema.target_arena().data.data_ptr(), n, decay)
        if nc is not None:
            ready = torch.cuda.Event()
            ready.record(main)
            nc.stream.wait_event(ready)
            if per:
                nc.reduce_scatter_(a.grad, per, stream=nc.stream.cuda_stream)
            if n > body:
                nc.allreduce_(a.grad[body:n], stream=nc.stream.cuda_stream, grad=True)
            ev = torch.cuda.Event()
            ev.record(nc.stream)
            with _comm.exposed():
                main.wait_event(ev)                      # exposed: the reduce-scatter only
            adam(lo, hi)
            adam(body, n)
            upd = torch.cuda.Event()
            upd.record(main)
            nc.stream.wait_event(upd)
            if per:
                nc.allgather_(a.data, per, stream=nc.stream.cuda_stream)
            if ema_args is not None:                     # p_ema = lerp(p, p_ema, decay) over the gathered parameters, behind the gather on the same stream
                L.call("sg_ema_lerp", *ema_args, nc.stream.cuda_stream)
            done = torch.cuda.Event()
            done.record(nc.stream)
            a.defer(done)
            if ema is not None:
                ema.target_arena().defer(done)
            return
        gshard = a.grad[lo:hi]
        has_rs = dist.get_backend(group) == "nccl"
        with _comm.exposed():
            if per and has_rs:
                dist.reduce_scatter_tensor(gshard, a.grad[:body], group=group)
            elif per:
                dist.all_reduce(a.grad[:body], group=group)
            if n > body:
                dist.all_reduce(a.grad[body:n], group=group)
        adam(lo, hi)
        adam(body, n)
        if per:
            if has_rs:
                work = dist.all_gather_into_tensor(a.data[:body], a.data[lo:hi].clone(), group=group, async_op=ema_args is None)
                if ema_args is None and work is not None:
                    a.defer(work)
            else:
                parts = [torch.empty_like(gshard) for _ in range(world)]
                dist.all_gather(parts, a.data[lo:hi].clone(), group=group)
                for r, t in enumerate(parts):
                    if r != rank:
                        a.data[r * per:(r + 1) * per].copy_(t)
        if ema_args is not None:
            L.call("sg_ema_lerp", *ema_args, st)


class Ema:
    """reference src/utils/ema.py:11-40 on the flat arenas: parameters are lerped inside the fused Adam launch (or by
    `update()` when driven like the reference does), buffers by one more launch."""

    def __init__(self, source, target, decay=0.9999, start_iter=0):
        self.source, self.target = source, target
        self.decay, self.start_iter = decay, start_iter
        with torch.no_grad():
            for p_ema, p in zip(target.parameters(), source.parameters()):
                p_ema.copy_(p)
            for b_ema, b in zip(target.buffers(), source.buffers()):
                b_ema.copy_(b)

    def decay_at(self, iteration):
        if iteration is not None and 0 <= iteration < self.start_iter:
            return 0.0
        return self.decay

    def source_arena(self):
        return _arena_for(list(self.source.parameters()))

    def target_arena(self):
        return _arena_for(list(self.target.parameters()))

    def _buffers(self):
        return get_buffer_arena(self.source), get_buffer_arena(self.target)

    @torch.no_grad()
    def update_buffers(self, decay):
        sb, tb = self._buffers()
        if sb.numel:
            L.call("sg_ema_lerp", sb.data.data_ptr(), tb.data.data_ptr(), sb.numel, decay, L.stream())
        if sb.inames == tb.inames and sb.inumel == tb.inumel:
            if sb.inumel:
                tb.idata.copy_(sb.idata)  # every num_batches_tracked at once
        else:
            for (n_t, b_t), (n_s, b_s) in zip(self.target.named_buffers(), self.source.named_buffers()):
                if b_t.dtype == torch.int64:
                    b_t.copy_(b_s)  # num_batches_tracked
        for (n_t, b_t), (n_s, b_s) in zip(self.target.named_buffers(), self.source.named_buffers()):
            if b_t.dtype not in (torch.float32, torch.int64):
                b_t.copy_(b_s)

    @torch.no_grad()
    def update(self, iter=None):
        decay = self.decay_at(iter)
        sa, ta = self.source_arena(), self.target_arena()
        L.call("sg_ema_lerp", sa.data.data_ptr(), ta.data.data_ptr(), sa.numel, decay, L.stream())
        self.update_buffers(decay)


class SmallAdam:
    """torch.optim.Adam semantics for a HANDFUL of parameters that live inside another network's arena (InfoGAN's Q heads: they sit in the discriminator module and
    are trained in the generator update with the generator's optimiser settings, reference src/config.py:501-512): one sg_adam_ema launch per parameter on its own
    data / gradient views, moments of its own, one step count. The gradients are whatever the kernels accumulated into p.grad (views of the owning arena)."""

    def __init__(self, params, lr=2e-4, betas=(0.5, 0.999), eps=1e-6, weight_decay=0.0):
        self.params = [p for p in params]
        assert self.params, "SmallAdam: no parameters"
        self.lr, self.betas, self.eps, self.weight_decay = lr, tuple(betas), eps, weight_decay
        self._m = [torch.zeros_like(p, dtype=torch.float32).reshape(-1) for p in self.params]
        self._v = [torch.zeros_like(p, dtype=torch.float32).reshape(-1) for p in self.params]
        self._t = 0

    def zero_grad(self):
        for p in self.params:
            if p.grad is not None:
                p.grad.zero_()

    @torch.no_grad()
    def step(self, group=None):
        world = dist.get_world_size(group) if (group is not None and dist.is_available() and dist.is_initialized()) else 1
        self._t += 1
        for p, m, v in zip(self.params, self._m, self._v):
            if p.grad is None:
                continue
            assert p.is_contiguous() and p.grad.is_contiguous() and p.dtype == torch.float32
            if world > 1:
                dist.all_reduce(p.grad, op=dist.ReduceOp.SUM, group=group)
            L.call("sg_adam_ema", p.data_ptr(), p.grad.data_ptr(), m.data_ptr(), v.data_ptr(), None, p.numel(), self.lr, self.betas[0], self.betas[1], self.eps,
                   self.weight_decay, self._t, 0.0, 1.0 / world, L.stream())
