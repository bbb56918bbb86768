"""Native RCCL communicator of libsgamd.so (include/sgamd.h: sg_comm_*, sg_allreduce_flat, sg_bn_stats_sync): the data-parallel
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

    def close(self):
        for name in ("handle", "grad_handle"):
            h = getattr(self, name, None)
            if h:
                L.call("sg_comm_destroy", h)
                setattr(self, name, C.c_void_p())


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
    for c in _REGISTRY.values():
        c.close()
    _REGISTRY.clear()
