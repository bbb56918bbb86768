"""Flat parameter arenas and the per-network weight bank.

MI355X-first layout of the *weight side* of the hot path:

* every fp32 master parameter of a network lives in ONE contiguous arena (and its gradient in a twin arena), so the
  optimizer step (+EMA) is one launch and the data-parallel gradient exchange is one RCCL all-reduce of one buffer
  (replaces DDP's 25 MB buckets, reference src/models/model.py:171-180);
* spectral norm (reference src/utils/ops.py:195-224) runs for ALL layers of the network in 4-5 batched launches at the
  start of each forward (`WeightBank.begin_forward`), emitting the normalised weight directly in the compute dtype and
  in the two operand layouts the convolution engine reads;
* the backward of the normalisation is likewise batched: convolution backward kernels drop dL/dW_sn into a per-forward
  fp32 arena and one `sg_sn_backward` per forward (queued as an autograd engine callback) folds them into the
  gradient arena.

Parameter names/shapes stay exactly the reference's (`weight_orig`, `weight_u`, `weight_v`, `bias`, ...), so
`state_dict()` / `load_state_dict(strict=True)` interchange with StudioGAN checkpoints (reference src/utils/ckpt.py:38).
"""
import ctypes
import os
import weakref

import torch

from . import _lib as L

_EVAL_CACHE = [os.environ.get("SG_EVAL_CACHE", "0") == "1"]   # frozen-network weight-image cache (begin_forward), opt-in; a list so that tests can flip it
_ARENA_BY_ID = {}  # id(Parameter) -> (weakref to the Parameter, ParamArena, offset); keyed by identity, never by ==


def _register(p, arena, off):
    key = id(p)

    def _gone(_ref, key=key, table=_ARENA_BY_ID):      # `table` bound here: module globals are None during interpreter shutdown
        ent = table.get(key)
        if ent is not None and ent[0] is _ref:
            del table[key]
    _ARENA_BY_ID[key] = (weakref.ref(p, _gone), arena, off)


def arena_of(p):
    ent = _ARENA_BY_ID.get(id(p))
    if ent is None or ent[0]() is not p:
        return None
    return ent[1], ent[2]


def _align(n, a=4):
    return (n + a - 1) // a * a


class ParamArena:
    """Contiguous fp32 storage for a list of parameters plus a twin gradient arena."""

    def __init__(self, params):
        params = [p for p in params]
        assert len(params) > 0
        dev = params[0].device
        offs, total = [], 0
        for p in params:
            assert p.dtype == torch.float32 and p.device == dev
            offs.append(total)
            total += _align(p.numel())
        self.params = params
        self.offsets = offs
        self.numel = total
        self.data = torch.zeros(total, device=dev, dtype=torch.float32)
        self.grad = torch.zeros(total, device=dev, dtype=torch.float32)
        with torch.no_grad():
            for p, o in zip(params, offs):
                view = self.data[o:o + p.numel()].view(p.shape)
                view.copy_(p.data)
                p.data = view
                p.grad = None
                _register(p, self, o)

    def grad_view(self, p):
        a, o = arena_of(p)
        return a.grad[o:o + p.numel()].view(p.shape)

    # -- a write to `data` still in flight on another stream (optim.FusedAdam's sharded step leaves the all-gather of the updated parameters, and the EMA lerp
    #    behind it, on the communicator's side stream so that they overlap the next forward of the OTHER network): everything that reads the parameters through
    #    this package -- the weight bank's spectral-norm passes, the optimizer, checkpointing helpers -- calls wait_ready() first.
    pending = None

    def defer(self, waiter):
        """waiter: a torch.cuda.Event recorded behind the write, or any object with .wait() (a torch.distributed Work)"""
        self.wait_ready()
        self.pending = waiter

    def wait_ready(self):
        w = self.pending
        if w is not None:
            self.pending = None
            if isinstance(w, torch.cuda.Event):
                torch.cuda.current_stream().wait_event(w)
            else:
                w.wait()

    def intact(self):
        p, o = self.params[0], self.offsets[0]
        q, oq = self.params[-1], self.offsets[-1]
        return p.data_ptr() == self.data.data_ptr() + 4 * o and q.data_ptr() == self.data.data_ptr() + 4 * oq


def ensure_grad(p):
    """Gradient tensor of a parameter that our kernels accumulate into (a view of the gradient arena when the
    parameter lives in one). Honour optimizer.zero_grad(set_to_none=True): a None grad is re-created zeroed."""
    g = p.grad
    if g is not None:
        return g
    ent = arena_of(p)
    if ent is not None:
        a, o = ent
        g = a.grad[o:o + p.numel()].view(p.shape)
        g.zero_()
    else:
        g = torch.zeros_like(p.data)
    p.grad = g
    return g


class BufferArena:
    """Contiguous storage for the float buffers of a network (BN running stats, SN u/v) -> one fused EMA launch."""

    def __init__(self, named_buffers):
        bufs = [(n, b) for n, b in named_buffers if b is not None and b.dtype == torch.float32]
        self.names = [n for n, _ in bufs]
        self.offsets, total = [], 0
        for _, b in bufs:
            self.offsets.append(total)
            total += _align(b.numel())
        dev = bufs[0][1].device if bufs else torch.device("cpu")
        self.data = torch.zeros(max(total, 4), device=dev, dtype=torch.float32)
        self.numel = total
        self.bufs = [b for _, b in bufs]
        # the int64 buffers (BatchNorm's num_batches_tracked) side by side as well: the EMA twin copies them in one launch and a generator forward bumps all of
        # them in one launch (ops.bump_batches_tracked) instead of one tiny kernel per batch norm -- 44 launches per BigGAN-128 step
        ib = [(n, b) for n, b in named_buffers if b is not None and b.dtype == torch.int64]
        self.inames = [n for n, _ in ib]
        self.ibufs = [b for _, b in ib]
        self.ioffsets, itotal = [], 0
        for b in self.ibufs:
            self.ioffsets.append(itotal)
            itotal += b.numel()
        self.idata = torch.zeros(max(itotal, 1), device=dev, dtype=torch.int64)
        self.inumel = itotal
        with torch.no_grad():
            for b, o in zip(self.bufs, self.offsets):
                view = self.data[o:o + b.numel()].view(b.shape)
                view.copy_(b)
                b.set_(view)  # in place: the module's registered buffer object now aliases the arena
            for b, o in zip(self.ibufs, self.ioffsets):
                view = self.idata[o:o + b.numel()].view(b.shape)
                view.copy_(b)
                b.set_(view)

    def intact(self, root):
        bufs = [b for _, b in root.named_buffers() if b is not None and b.dtype == torch.float32]
        if len(bufs) != len(self.bufs):
            return False
        base = self.data.data_ptr()
        if not all(b.data_ptr() == base + 4 * o for b, o in zip(bufs, self.offsets)):
            return False
        ib = [b for _, b in root.named_buffers() if b is not None and b.dtype == torch.int64]
        ibase = self.idata.data_ptr()
        return len(ib) == len(self.ibufs) and all(b.data_ptr() == ibase + 8 * o for b, o in zip(ib, self.ioffsets))


def get_buffer_arena(root):
    """The (single) buffer arena of a network; rebuilt when .to()/deepcopy replaced the buffer tensors."""
    h = root.__dict__.get("_sg_buf_holder")
    if h is None:
        h = _Holder()
        root.__dict__["_sg_buf_holder"] = h
    if h.obj is None or not h.obj.intact(root):
        h.obj = BufferArena(list(root.named_buffers()))
    return h.obj


class _Holder:
    """Keeps runtime state (ctypes tables, arenas) out of copy.deepcopy / pickling of the owning nn.Module."""

    def __init__(self):
        self.obj = None

    def __deepcopy__(self, memo):
        return _Holder()

    def __getstate__(self):
        return {}

    def __setstate__(self, st):
        self.obj = None


class GradReadyFn(torch.autograd.Function):
    """Identity on an activation at a block boundary of a network. Its backward tells the network's exchange plan (optim.ExchangePlan) that the
    backward pass has come back to this point: every parameter used BEHIND it in the forward now has its final gradient, so the all-reduce of
    that part of the flat gradient arena can start while the rest of the backward still runs (what DistributedDataParallel's buckets do for
    the reference, src/models/model.py:171-180). create_graph passes (gradient penalty) are ignored."""

    @staticmethod
    def forward(ctx, act, plan, bank_ref, offset):
        ctx.plan, ctx.bank_ref, ctx.offset = plan, bank_ref, offset
        return act.view_as(act)

    @staticmethod
    def backward(ctx, dy):
        if not torch.is_grad_enabled():
            bank = ctx.bank_ref()
            if bank is not None:
                ctx.plan.crossed(ctx.offset, bank)
        return dy, None, None, None


class LayerRT:
    """Runtime record of one weight-bearing layer inside a bank."""
    __slots__ = ("module", "index", "param", "kind", "rows", "cols", "Cin", "RS", "apply_sn", "rows_pad", "want_fwd", "want_dgrad",
                 "want_f32", "fwd_off", "dgrad_off", "f32_off", "dwt_off", "uv_off", "work_off", "natural_dwt", "bank", "trans", "noflip", "cin_pad",
                 "param_off")


class _Slot:
    """One set of per-forward weight-side buffers (operand images, u / v / sigma snapshots, dL/dW_sn scratch). `begin_forward` hands
    out a HANDLE per forward: a second _Slot object that shares this one's attribute dict. Every autograd node of that forward keeps
    the handle alive through ctx.slot, so "the handle is gone" == "no backward of that forward can come any more"."""
    pass


class WeightBank:
    SN_SPLITS = 16          # row groups of csrc/sn.hip k_sn_wtu (same constant there)
    SNB_BLOCKS = 512        # block partials of csrc/sn.hip k_snb_dot

    def __init__(self, root, compute_dtype, nslots=4, eps=1e-6):
        self.root_ref = weakref.ref(root)
        self.dtype = compute_dtype
        self.sgdt = L.dt(compute_dtype)
        self.eps = eps
        self.nslots = nslots
        layers = []
        for m in root.modules():
            if getattr(m, "_sg_weight_layer", False):
                layers.append(m)
        assert layers, "no weight layers found"
        dev = next(root.parameters()).device
        L.require_gpu(dev)
        self.device = dev
        # 1. flatten parameters / buffers of the whole network (not only weight layers)
        params = list(root.parameters())
        if not all((arena_of(p) is not None) for p in params) or len({id(arena_of(p)[0]) for p in params}) != 1 \
                or not arena_of(params[0])[0].intact():
            self.params = ParamArena(params)
        else:
            self.params = arena_of(params[0])[0]
        self.buffers = get_buffer_arena(root)
        # 2. per-layer records and arena offsets
        es = 2 if compute_dtype == torch.bfloat16 else 4
        self.layers = []
        img_elems = f32_elems = dwt_elems = uv_elems = work = 0
        for i, m in enumerate(layers):
            r = LayerRT()
            r.module, r.index, r.bank = weakref.ref(m), i, weakref.ref(self)
            r.param = m.weight_orig if m._sg_sn else m.weight
            r.kind = m._sg_kind
            r.rows, r.cols, r.Cin, r.RS = m._sg_rows, m._sg_cols, m._sg_cin, m._sg_rs
            r.apply_sn = 1 if m._sg_sn else 0
            r.rows_pad = getattr(m, "_sg_rows_pad", 0) or r.rows
            r.trans = 1 if getattr(m, "_sg_trans", False) else 0
            r.noflip = 1 if getattr(m, "_sg_dgrad_noflip", False) else 0
            r.cin_pad = getattr(m, "_sg_cin_pad", 0) or r.Cin
            assert r.cin_pad == r.Cin or (r.kind == "conv" and not r.trans), "input-channel padding is for plain convolutions"
            r.want_fwd = r.kind == "conv"
            r.want_dgrad = r.kind == "conv"
            r.want_f32 = r.kind in ("linear", "embedding")
            r.natural_dwt = (2 if getattr(m, "_sg_trans", False) else 0) if r.kind == "conv" else 1
            r.fwd_off = r.dgrad_off = r.f32_off = -1
            if r.want_fwd:
                r.fwd_off = img_elems
                img_elems += _align(r.rows_pad * r.RS * r.cin_pad, 16)
            if r.want_dgrad:
                r.dgrad_off = img_elems
                img_elems += _align(r.rows_pad * r.RS * r.cin_pad, 16)   # [cin_pad][RS][rows_pad], zero outside [Cin][RS][rows]
            if r.want_f32:
                r.f32_off = f32_elems
                f32_elems += _align(r.rows * r.cols)
            r.dwt_off = dwt_elems
            dwt_elems += _align(r.rows_pad * r.RS * r.cin_pad)
            r.uv_off = uv_elems
            uv_elems += _align(r.rows) + _align(r.cols)
            r.work_off = work
            work += self.SN_SPLITS * r.cols + r.rows
            r.param_off = arena_of(r.param)[1]        # position of the master weight in the flat arena (exchange ranges)
            m._sg_rt = r
            self.layers.append(r)
        self.work = torch.zeros(max(work, self.SNB_BLOCKS * len(layers)) + 64, device=dev, dtype=torch.float32)
        self._sizes = (img_elems, f32_elems, dwt_elems, uv_elems)
        self.slots = []
        for s in range(nslots):
            self.slots.append(self._new_slot(s))
        self._ring = 0
        self._fwd_train_epoch = 0           # forwards of this network that ran the power iteration (they move u / v)
        self.current = self.slots[0]
        self._cb_queued = False
        self.es = es
        self.exchange = None        # optim.ExchangePlan of the optimizer that owns this network's parameters (data parallelism only)

    def boundaries(self, blocks):
        """For each entry of `blocks` (the network's top-level block list, in forward order): the first parameter, in arena order, of what
        runs behind it -- the argument of mark(). Cached. None where nothing follows."""
        b = self.__dict__.get("_boundaries")
        if b is None:
            root = self.root_ref()
            plist = list(root.parameters())
            order = {id(p): k for k, p in enumerate(plist)}
            b, last = [], -1
            for blk in blocks:
                ids = [order[id(p)] for p in blk.parameters()]
                if ids and min(ids) <= last:      # registration order is not forward order: no early exchange for this network
                    b = [None] * len(blocks)
                    break
                nxt = (max(ids) + 1) if ids else None
                last = max(ids) if ids else last
                b.append(plist[nxt] if (nxt is not None and nxt < len(plist)) else None)
            self._boundaries = b
        return b

    def mark(self, act, next_param):
        """Block boundary in a backbone's forward: `next_param` is the first parameter (in arena order) of what runs behind this point.
        No-op unless a data-parallel optimizer attached an exchange plan and a graph is being built."""
        plan = self.exchange
        if plan is None or next_param is None or not torch.is_grad_enabled() or not act.requires_grad:
            return act
        ent = arena_of(next_param)
        if ent is None or ent[0] is not self.params:
            return act
        plan.expect(ent[1])
        return GradReadyFn.apply(act, plan, weakref.ref(self), ent[1])

    MAX_SLOTS = 12

    def _new_slot(self, s):
        img_elems, f32_elems, dwt_elems, uv_elems = self._sizes
        dev = self.device
        sl = _Slot()
        sl.index = s
        sl.img = torch.zeros(max(img_elems, 16), device=dev, dtype=self.dtype)
        sl.f32 = torch.zeros(max(f32_elems, 4), device=dev, dtype=torch.float32)
        sl.dwt = torch.zeros(max(dwt_elems, 4), device=dev, dtype=torch.float32) if s > 0 else None
        sl.uv = torch.zeros(max(uv_elems, 4), device=dev, dtype=torch.float32)
        sl.sigma = torch.ones(len(self.layers), device=dev, dtype=torch.float32)
        sl.dwt_zeroed = False
        sl.pending = []
        sl.desc_cache = {}
        sl.bwd_cache = {}
        sl.quad = {}            # (layer index, mode) -> [quad filter image, id of the forward it was packed for]  (w_quad)
        sl.quad_tab = {}        # tuple of (layer index, mode) -> (host table, device table) of one sg_quad_pack_batch launch
        sl.fwd_id = 0
        sl.live = None          # weakref to the handle of the forward that currently owns the slot
        return sl

    def _free_graph_slot(self):
        """Next graph slot (ring order) whose previous forward can no longer run a backward. A D step with several penalties
        (apply_gp + apply_dra / apply_maxgp, bCR / zCR) keeps more forwards waiting for ONE backward than the initial ring holds:
        the ring grows (288 GB of HBM) instead of silently recycling a slot whose sigma / u / v / operand images are still needed."""
        # `self.current` is only a convenience pointer for standalone modules: it must not be what keeps the previous forward's handle
        # (and with it a whole slot: img + f32 + dwt + uv copies of the network, ~1 GB for BigGAN's D) alive. Autograd nodes and the
        # caller's output tensors hold their own references; hold graph outputs across iterations (e.g. log an undetached loss) and the
        # ring grows by one slot per forward until MAX_SLOTS -- detach what you keep.
        self.current = self.slots[0]
        n = len(self.slots) - 1
        for k in range(1, n + 1):
            idx = (self._ring + k - 1) % n + 1
            sl = self.slots[idx]
            if sl.live is None or sl.live() is None:
                self._ring = idx
                return sl
        if len(self.slots) >= self.MAX_SLOTS:
            raise RuntimeError(f"{len(self.slots) - 1} forwards of this network are waiting for their backward; refusing to overwrite the "
                               "spectral-norm state of the oldest one (raise WeightBank.MAX_SLOTS if this is intended)")
        sl = self._new_slot(len(self.slots))
        self.slots.append(sl)
        self.nslots = len(self.slots)
        self._ring = sl.index
        import sys
        sys.stderr.write(f"[studiogan_amd] weight bank grew to {self.nslots - 1} graph slots: {self.nslots - 2} earlier forwards of this network still "
                         "hold their graph (several penalties in one D step is expected; otherwise detach the outputs you keep across iterations)\n")
        return sl

    # -- forward ------------------------------------------------------------------------------------------
    def _desc(self, slot, flags):
        """-> (host table, device table, groups): the layers in table order = convolutions first, then linear / embedding layers; groups = [(first, count)] = one run.
        (Until round 6 sg_sn_forward sized its grids by the largest layer of the table, and a generator's [24576 x 20] linear0 next to [1536 x 13824] convolutions had to go
        through a call of its own; the launches are flat tile tables now -- csrc/sn.hip sn_flat -- and every layer kind shares one call.)"""
        ent = slot.desc_cache.get(flags)
        if ent is not None:
            return ent
        n = len(self.layers)
        arr = (L.SnLayer * n)()
        es = self.es
        order = [r for r in self.layers if r.kind == "conv"] + [r for r in self.layers if r.kind != "conv"]
        pi_of = {r.index: pi for r, pi in zip(self.layers, flags)}
        for pos, r in enumerate(order):
            pi = pi_of[r.index]
            m = r.module()
            d = arr[pos]
            d.w = r.param.data_ptr()
            if r.apply_sn:
                d.u, d.v = m.weight_u.data_ptr(), m.weight_v.data_ptr()
            d.sigma = slot.sigma.data_ptr() + 4 * r.index
            d.u_snap = slot.uv.data_ptr() + 4 * r.uv_off
            d.v_snap = slot.uv.data_ptr() + 4 * (r.uv_off + _align(r.rows))
            d.w_fwd = slot.img.data_ptr() + es * r.fwd_off if r.want_fwd else None
            d.w_dgrad = slot.img.data_ptr() + es * r.dgrad_off if (r.want_dgrad and slot.dwt is not None) else None
            d.w_f32 = slot.f32.data_ptr() + 4 * r.f32_off if r.want_f32 else None
            d.rows, d.cols, d.Cin, d.RS = r.rows, r.cols, r.Cin, r.RS
            d.do_power_iter = 1 if pi else 0
            d.apply_sn = r.apply_sn
            d.rows_pad = r.rows_pad
            d.work_off = r.work_off
            d.trans, d.dgrad_noflip = r.trans, r.noflip
            d.Cin_pad = r.cin_pad
        dev_tab = L.upload_bytes(arr, self.device)
        groups = [(0, n)]
        ent = (arr, dev_tab, groups)
        slot.desc_cache[flags] = ent
        return ent

    def _versioned(self):
        """the tensors whose torch-side writes invalidate the emitted weight images: every weight-layer parameter and its u / v vectors"""
        v = self.__dict__.get("_versioned_cache")
        if v is None:
            v = []
            for r in self.layers:
                m = r.module()
                v.append(r.param)
                if r.apply_sn:
                    v += [m.weight_u, m.weight_v]
            self._versioned_cache = v
        return v

    def intact(self):
        root = self.root_ref()
        return self.params.intact() and root is not None and self.buffers.intact(root) and get_buffer_arena(root) is self.buffers

    def begin_forward(self, need_graph):
        """One spectral-norm power iteration + weight image emission for every layer; returns the slot."""
        self.params.wait_ready()          # (a deferred all-gather of this network's parameters: optim.FusedAdam sharded step)
        if need_graph:
            phys = self._free_graph_slot()
            phys.dwt_zeroed = False
            phys.pending = []
            slot = _Slot()
            slot.__dict__ = phys.__dict__          # handle: same attributes, its lifetime = the lifetime of this forward's graph
            phys.live = weakref.ref(slot)
        else:
            slot = self.slots[0]
        slot.cbn_rows = None              # (functional.cbn_prefetch: the conditional batch norms' affine rows of THIS forward)
        flags = tuple(bool(r.module().training) for r in self.layers)
        # A frozen network run without a graph (the evaluation generator of the FID / IS loop, reference src/metrics/features.py:17-65: one forward per
        # batch of 50 k samples) gets the SAME weight images every time: no power iteration (eval mode), same weights. Slot 0 keeps them until
        # something writes the parameters or the spectral-norm vectors: our own raw-pointer writers move _lib.write_epoch, torch-side writes
        # (load_state_dict, an in-place op on a parameter) move the tensors' version counters. What NEITHER sees is an in-place write through
        # `param.data` (a detached alias with its own version counter): the cache is therefore OPT-IN (SG_EVAL_CACHE=1) for loops that own their network.
        key = None
        if not need_graph and not any(flags) and _EVAL_CACHE[0]:
            key = (L.write_epoch[0], sum(p._version for p in self._versioned()), self.params.data._version, self.buffers.data._version, self._fwd_train_epoch)
            if slot.__dict__.get("emit_key") == key:
                self.current = slot
                self.eval_cache_hits = self.__dict__.get("eval_cache_hits", 0) + 1
                return slot
        if any(flags):
            self._fwd_train_epoch += 1          # the power iteration of this forward moves u / v
        if not need_graph:
            slot.emit_key = key
        arr, dev_tab, groups = self._desc(slot, flags)
        esz = ctypes.sizeof(L.SnLayer)
        for first, count in groups:
            L.call("sg_sn_forward", self.sgdt, dev_tab.data_ptr() + first * esz, ctypes.cast(ctypes.addressof(arr) + first * esz, ctypes.POINTER(L.SnLayer)), count, self.eps, self.work.data_ptr(),
                   self.work.numel(), L.stream())
        self._fwd_counter = self.__dict__.get("_fwd_counter", 0) + 1
        slot.fwd_id = self._fwd_counter      # (handle and physical slot share one attribute dict)
        self._pack_quad(slot, (0, 1, 4, 5))  # the forward quad images this network is known to use (csrc/conv_q.h), one launch
        self.current = slot
        return slot

    # -- pointers -----------------------------------------------------------------------------------------
    def w_fwd(self, slot, r):
        return slot.img.data_ptr() + self.es * r.fwd_off

    def w_dgrad(self, slot, r):
        return slot.img.data_ptr() + self.es * r.dgrad_off

    def w_f32(self, slot, r):
        return slot.f32.data_ptr() + 4 * r.f32_off

    def w_quad(self, slot, r, mode):
        """Quad filter image of a 3x3 / pad-1 layer that sits next to a 2x resampling (csrc/conv_q.h) for this forward: mode 0 / 1 = forward image
        of the POOL (conv + avg-pool) / UP (upsample + conv) form, from the forward image sg_sn_forward has just written; mode 2 / 3 = their
        data-gradient images from the flipped transposed image. Which (layer, mode) pairs a network uses is learnt on first use; from then on
        begin_forward packs all forward images in ONE launch and the first data-gradient request of a backward packs all of those."""
        key = (r.index, mode)
        ent = slot.quad.get(key)
        if ent is not None and ent[1] == slot.fwd_id:
            return ent[0].data_ptr()
        known = self.__dict__.setdefault("_quad_known", set())
        known.add(key)
        self._pack_quad(slot, (2, 3) if mode in (2, 3) else None, only=None if mode in (2, 3) else key)
        return slot.quad[key][0].data_ptr()

    def _pack_quad(self, slot, modes, only=None):
        """one sg_quad_pack_batch launch for the known (layer, mode) pairs of `modes` that this forward has not packed yet (only: just that pair)"""
        known = self.__dict__.get("_quad_known")
        if not known:
            return
        keys = [only] if only is not None else sorted(k for k in known if k[1] in modes)
        keys = tuple(k for k in keys if not (k in slot.quad and slot.quad[k][1] == slot.fwd_id))
        if not keys or (keys[0][1] in (2, 3) and slot.dwt is None):
            return
        if len(keys) > 64:          # sg_quad_pack_batch takes at most 64 items per launch (a deep / high-resolution backbone: ADVICE r4)
            for i in range(0, len(keys), 64):
                self._pack_quad_keys(slot, keys[i:i + 64])
            return
        self._pack_quad_keys(slot, keys)

    def _pack_quad_keys(self, slot, keys):
        tab = slot.quad_tab.get(keys)
        if tab is None:
            arr = (L.QuadItem * len(keys))()
            for j, (idx, mode) in enumerate(keys):
                r = self.layers[idx]
                ent = slot.quad.get((idx, mode))
                if ent is None:
                    ent = [torch.empty(r.rows_pad * (r.RS if mode == 4 else (4 if mode == 5 else 16)) * r.cin_pad, dtype=self.dtype, device=self.device), -1]
                    slot.quad[(idx, mode)] = ent
                it = arr[j]
                if mode == 5:        # the 8-channel (image) skip filter x 1/4, once per parity view
                    it.src, it.M, it.Cs = self.w_fwd(slot, r), r.rows_pad, 8
                elif mode == 4:      # the 1x1 skip filter of a pooled block tail x 1/4
                    it.src, it.M, it.Cs = self.w_fwd(slot, r), r.rows_pad, r.RS * r.cin_pad
                elif mode < 2:
                    it.src, it.M, it.Cs = self.w_fwd(slot, r), r.rows_pad, r.cin_pad
                else:
                    it.src, it.M, it.Cs = self.w_dgrad(slot, r), r.cin_pad, r.rows_pad
                it.dst, it.mode = ent[0].data_ptr(), mode
            tab = (arr, L.upload_bytes(arr, self.device))
            slot.quad_tab[keys] = tab
        L.call("sg_quad_pack_batch", self.sgdt, tab[1].data_ptr(), tab[0], len(keys), L.stream())
        for k in keys:
            slot.quad[k][1] = slot.fwd_id

    def w_f32_tensor(self, slot, r):
        return slot.f32[r.f32_off:r.f32_off + r.rows * r.cols].view(r.rows, r.cols)

    def dwt(self, slot, r):
        """fp32 scratch for dL/dW_sn of this forward (zeroed once per backward pass), registers the layer for the
        batched normalisation backward."""
        if slot.dwt is None:
            raise RuntimeError("weight gradient requested for a forward that was run without a graph")
        if not slot.dwt_zeroed:
            slot.dwt.zero_()
            slot.dwt_zeroed = True
        if r.index not in slot.pending:
            slot.pending.append(r.index)
        if not self._cb_queued:
            self._cb_queued = True
            torch.autograd.Variable._execution_engine.queue_callback(self.flush)
        return slot.dwt.data_ptr() + 4 * r.dwt_off

    def dwt_tensor(self, slot, r):
        self.dwt(slot, r)
        return slot.dwt[r.dwt_off:r.dwt_off + r.rows_pad * r.RS * r.cin_pad].view(r.rows_pad, r.RS * r.cin_pad)

    # -- backward of the normalisation, batched ----------------------------------------------------------------
    def flush(self, lo=None, hi=None):
        """Fold the pending dL/dW_sn scratch of every forward into the gradient arena. lo / hi (arena offsets): only the layers whose master
        weight lies in [lo, hi) -- the early gradient exchange folds a finished range while the backward is still running; the engine
        callback at the end of the pass (no arguments) takes whatever is left."""
        if lo is None:
            self._cb_queued = False
        self.params.wait_ready()
        for slot in self.slots[1:]:
            if not slot.pending:
                continue
            if lo is None:
                key = tuple(sorted(slot.pending))
                slot.pending = []
            else:
                key = tuple(sorted(i for i in slot.pending if lo <= self.layers[i].param_off < hi))
                if not key:
                    continue
                slot.pending = [i for i in slot.pending if i not in key]
            ent = slot.bwd_cache.get(key)
            if ent is None:
                arr = (L.SnBwdLayer * len(key))()
                for j, idx in enumerate(key):
                    r = self.layers[idx]
                    d = arr[j]
                    d.dwt = slot.dwt.data_ptr() + 4 * r.dwt_off
                    d.w = r.param.data_ptr()
                    d.u = slot.uv.data_ptr() + 4 * r.uv_off
                    d.v = slot.uv.data_ptr() + 4 * (r.uv_off + _align(r.rows))
                    d.sigma = slot.sigma.data_ptr() + 4 * r.index
                    d.dw = 0
                    d.rows, d.cols, d.Cin, d.RS = r.rows, r.cols, r.Cin, r.RS
                    d.natural = r.natural_dwt
                    d.apply_sn = r.apply_sn
                    d.trans = r.trans
                    d.Cin_pad = r.cin_pad
                ent = [arr, None, None]
                slot.bwd_cache[key] = ent
            arr = ent[0]
            grads = tuple(ensure_grad(self.layers[idx].param).data_ptr() for idx in key)
            if ent[2] != grads:  # gradient tensors may be re-created by zero_grad(set_to_none=True)
                for j, gp in enumerate(grads):
                    arr[j].dw = gp
                ent[1] = L.upload_bytes(arr, self.device)
                ent[2] = grads
            L.call("sg_sn_backward", ent[1].data_ptr(), arr, len(key), self.work.data_ptr(), self.work.numel(), L.stream())


def get_bank(root, compute_dtype):
    """Bank of a network (built lazily at the first forward, rebuilt if .to()/deepcopy invalidated the arenas)."""
    h = root.__dict__.get("_sg_bank_holder")
    if h is None:
        h = _Holder()
        root.__dict__["_sg_bank_holder"] = h
    b = h.obj
    if b is None or b.dtype != compute_dtype or not b.intact():
        b = WeightBank(root, compute_dtype)
        h.obj = b
    return b
