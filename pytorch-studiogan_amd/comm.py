"""Native communicators of libsgamd.so (include/sgamd.h: sg_comm_*, sg_allreduce_flat, sg_reduce_scatter_flat / sg_allgather_flat, sg_bn_stats_sync over RCCL;
sg_p2p_* peer-store mailboxes with the sync-BN exchange fused into the statistics kernel): the data-parallel
exchanges of the training step -- gradient all-reduce of the flat arena, sync-BN statistics -- issued through the C ABI on HIP
streams instead of torch.distributed collectives. Replaces DistributedDataParallel / SyncBatchNorm of reference
src/models/model.py:157-180.

torch.distributed is only used ONCE, to hand rank 0's 128-byte RCCL unique id to the other ranks (any rendezvous would do).
Opt-in: `studiogan_amd.comm.enable(group)` (bench.py: --native-comm or SG_NATIVE_COMM=1). Without it the same exchanges run through
torch.distributed (backend "nccl" == RCCL on ROCm), which is also what the gloo-based CPU / one-GPU tests drive.
"""
import ctypes as C

import torch
import torch.distributed as dist

from . import _lib as L

_REGISTRY = {}     # id(process group) or "world" -> NativeComm
_TIMING = None     # comm.timing(True): [(start, end)] hipEvent pairs on the compute stream around every collective it has to wait for


def timing(on):
    """Measure the EXPOSED communication of the data-parallel step (bench.py `exposed_comm_ms_per_step`): with it on, every point where the
    compute stream waits for a collective -- the sync-BN all-reduces, which run on it, and the waits on the gradient reductions in
    FusedAdam.step -- is bracketed by a pair of events recorded on that stream; what elapses between them is time the stream could not compute."""
    global _TIMING
    _TIMING = [] if on else None


class exposed:
    """with comm.exposed(): <issue / wait for a collective on the current stream>"""

    def __enter__(self):
        if _TIMING is not None:
            self.a = torch.cuda.Event(enable_timing=True)
            self.a.record(torch.cuda.current_stream())
        return self

    def __exit__(self, *exc):
        if _TIMING is not None:
            b = torch.cuda.Event(enable_timing=True)
            b.record(torch.cuda.current_stream())
            _TIMING.append((self.a, b))
        return False


def timing_ms():
    """total exposed time since the last call (synchronises the device)"""
    if _TIMING is None:
        return None
    torch.cuda.synchronize()
    ms = sum(a.elapsed_time(b) for a, b in _TIMING)
    _TIMING.clear()
    return ms


def _key(group):
    return "world" if group is None or group is True or group is dist.group.WORLD else id(group)


class NativeComm:
    def __init__(self, group=None, device=None):
        ddp = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if ddp else 1
        self.rank = dist.get_rank(group) if ddp else 0
        uid = (C.c_char * 128)()
        if self.rank == 0:
            L.call("sg_comm_unique_id", uid)
        src = 0 if (group is None or not ddp) else dist.get_global_rank(group, 0)
        if self.world > 1:
            box = [bytes(uid.raw)]
            dist.broadcast_object_list(box, src=src, group=group, device=device)
            uid = (C.c_char * 128).from_buffer_copy(box[0])
        self.handle = C.c_void_p()
        L.call("sg_comm_init_rank", uid, self.world, self.rank, C.byref(self.handle))
        # A SECOND communicator for the gradient exchange: it runs on the side stream, possibly while sync-BN all-reduces of the same
        # backward pass run on the compute stream -- two streams must not share one RCCL communicator (operations of a communicator have to
        # be issued in the same order on every rank, and stream interleaving does not guarantee that).
        uid2 = (C.c_char * 128)()
        if self.rank == 0:
            L.call("sg_comm_unique_id", uid2)
        if self.world > 1:
            box = [bytes(uid2.raw)]
            dist.broadcast_object_list(box, src=src, group=group, device=device)
            uid2 = (C.c_char * 128).from_buffer_copy(box[0])
        self.grad_handle = C.c_void_p()
        L.call("sg_comm_init_rank", uid2, self.world, self.rank, C.byref(self.grad_handle))
        self.stream = torch.cuda.Stream()      # side stream of the pipelined gradient exchange

    def allreduce_(self, t, stream=None, grad=False):
        """in-place sum of a contiguous fp32 / fp64 CUDA tensor over the ranks, enqueued on `stream` (default: the current stream).
        grad=True: on the gradient-exchange communicator (side stream), else on the one the sync-BN statistics use (compute stream)."""
        assert t.is_cuda and t.is_contiguous() and t.dtype in (torch.float32, torch.float64)
        L.call("sg_allreduce_flat", self.grad_handle if grad else self.handle, t.data_ptr(), t.numel(), L.F32 if t.dtype == torch.float32 else L.F64,
               stream if stream is not None else L.stream())
        return t

    def reduce_scatter_(self, flat, per_rank, stream=None):
        """in place on the gradient communicator: afterwards flat[rank * per_rank : (rank + 1) * per_rank] holds the sum over the ranks of that range"""
        assert flat.is_cuda and flat.is_contiguous() and flat.dtype == torch.float32 and flat.numel() >= per_rank * self.world
        L.call("sg_reduce_scatter_flat", self.grad_handle, flat.data_ptr(), per_rank, stream if stream is not None else L.stream())

    def allgather_(self, flat, per_rank, stream=None):
        """in place on the gradient communicator: every rank's range [rank * per_rank, (rank + 1) * per_rank) of flat reaches every rank"""
        assert flat.is_cuda and flat.is_contiguous() and flat.dtype == torch.float32 and flat.numel() >= per_rank * self.world
        L.call("sg_allgather_flat", self.grad_handle, flat.data_ptr(), per_rank, stream if stream is not None else L.stream())

    def close(self):
        for name in ("handle", "grad_handle"):
            h = getattr(self, name, None)
            if h:
                L.call("sg_comm_destroy", h)
                setattr(self, name, C.c_void_p())


class P2PMailbox:
    """Peer-store mailboxes of csrc/p2p.hip for `group`: sync-BN's exchange fused into the statistics kernel (sg_bn_finalize_p2p) and the one-launch tiny
    all-reduce of the backward pass's channel terms (sg_p2p_allreduce_f64). torch.distributed is used ONCE, to hand the 64-byte IPC handles around (any
    backend: the data path never touches it) -- so two processes on ONE GPU exercise the very kernels eight GPUs run over xGMI."""

    MAX_DOUBLES = 8192          # 2 x C of the widest batch norm (BigGAN-deep at ch 128: C = 2048) with room to spare; 5 x C of the second-order terms up to C = 1638

    def __init__(self, group=None):
        ddp = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if ddp else 1
        self.rank = dist.get_rank(group) if ddp else 0
        self.handle = C.c_void_p()
        mine = (C.c_char * 64)()
        L.call("sg_p2p_create", self.world, self.rank, self.MAX_DOUBLES, C.byref(self.handle), mine)
        if self.world > 1:
            box = [None] * self.world
            dist.all_gather_object(box, bytes(mine.raw), group=group)
            L.call("sg_p2p_connect", self.handle, (C.c_char * (64 * self.world)).from_buffer_copy(b"".join(box)))

    def allreduce_f64_(self, t):
        """in-place sum of a small fp64 CUDA tensor over the ranks (rank order on every rank: bit-identical replicas), one launch on the current stream"""
        assert t.is_cuda and t.is_contiguous() and t.dtype == torch.float64 and t.numel() <= self.MAX_DOUBLES
        L.call("sg_p2p_allreduce_f64", self.handle, t.data_ptr(), t.numel(), L.stream())
        return t

    def timeouts(self):
        n = C.c_int(0)
        L.call("sg_p2p_timeouts", self.handle, C.byref(n))
        return n.value

    def close(self):
        if self.handle:
            L.call("sg_p2p_destroy", self.handle)
            self.handle = C.c_void_p()


_P2P = {}          # group key -> P2PMailbox


def enable_p2p(group=None):
    """Create (once) the peer-store mailboxes serving `group`: from then on functional.BNFn exchanges sync-BN statistics inside its finalize kernel and the
    backward's channel terms with one sg_p2p_allreduce_f64 launch instead of a collective call (RCCL or torch.distributed). Independent of enable():
    the gradient exchange keeps its RCCL communicator."""
    k = _key(group)
    if k not in _P2P:
        _P2P[k] = P2PMailbox(None if k == "world" else group)
    return _P2P[k]


def p2p_for(group):
    return _P2P.get(_key(group)) if _P2P else None


def enable(group=None, device=None):
    """Create (once) and register the native communicator serving `group`; returns it."""
    k = _key(group)
    if k not in _REGISTRY:
        _REGISTRY[k] = NativeComm(None if k == "world" else group, device)
    return _REGISTRY[k]


def native_for(group):
    """The native communicator registered for `group`, or None (-> torch.distributed path)."""
    return _REGISTRY.get(_key(group)) if _REGISTRY else None


def disable_all():
    for c in list(_REGISTRY.values()) + list(_P2P.values()):
        c.close()
    _REGISTRY.clear()
    _P2P.clear()
