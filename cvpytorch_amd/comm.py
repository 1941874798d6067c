"""Data-parallel collectives for the train step: ONE small interface (`allreduce_`, `broadcast_`, `barrier`) with two
transports.

  RcclComm       the product path: libcvhip's RCCL communicator (include/cvhip.h cvhip_comm_*, csrc/comm.hip) — librccl bound
                 directly, collectives enqueued on the caller's HIP stream, capturable in a hipGraph. One process per GPU.
                 Replaces reference trainer.py:312-313 (DistributedDataParallel) + src/utils/distributed.py:82-98
                 (init_process_group('nccl')): no torch process group is created.
  TorchDistComm  TEST transport only: wraps an existing torch.distributed group (gloo) so the multi-rank bookkeeping
                 (bucketing, readiness, SyncBN totals) can be exercised on CPU / with several ranks sharing one GPU, where
                 RCCL cannot run (it refuses two ranks on one device).

Rendezvous of RcclComm: the 128-byte RCCL unique id travels through a torch.distributed.TCPStore at MASTER_ADDR:MASTER_PORT —
a plain key-value store (the one torchrun's agent already serves when it launched the ranks), not a process group.
"""
import ctypes as C
import datetime
import os

import torch

from . import lib as L

_OPS = {"sum": L.RED_SUM, "max": L.RED_MAX, "min": L.RED_MIN}
_DTYPES = {torch.float32: L.DTYPE_F32, torch.float64: L.DTYPE_F64, torch.int32: L.DTYPE_I32, torch.bfloat16: L.DTYPE_BF16,
           torch.uint8: L.DTYPE_U8}


class Comm:
    """Interface. `capturable`: collectives may be issued while a hipGraph is being captured (they become graph nodes)."""
    world = 1
    rank = 0
    capturable = False

    def allreduce_(self, t, op="sum", stream=None):
        raise NotImplementedError

    def broadcast_(self, t, root=0, stream=None):
        raise NotImplementedError

    def reduce_scatter_allgather_(self, t, stream=None):
        """the same sum as allreduce_ as its two ring phases (reduce-scatter, then all-gather of the owned chunks): what SURVEY.md
        §8(d)/(e) recommends for 29-165 MB gradient payloads over the 7 point-to-point xGMI links. Transports without the split
        form (and element counts that do not divide by the world size) run the plain all-reduce."""
        return self.allreduce_(t, "sum", stream)

    def barrier(self):
        raise NotImplementedError

    def wait(self):
        """Block the CURRENT stream / host until collectives issued so far are complete enough to read their results from the
        current stream (async work handles of the test transport; stream ordering does it for RCCL)."""

    def close(self):
        pass


class RcclComm(Comm):
    capturable = True

    def __init__(self, world, rank, unique_id, device=None):
        if device is not None:
            torch.cuda.set_device(device)
        self.world, self.rank = int(world), int(rank)
        self._h = C.c_void_p()
        buf = (C.c_char * len(unique_id)).from_buffer_copy(unique_id)
        L.call("cvhip_comm_init_rank", C.byref(self._h), self.world, self.rank, C.cast(buf, C.c_void_p))
        self.device = torch.device("cuda", torch.cuda.current_device())

    @staticmethod
    def unique_id():
        lib = L.load()
        n = lib.cvhip_comm_unique_id_bytes()
        buf = (C.c_char * n)()
        L.call("cvhip_comm_get_unique_id", C.cast(buf, C.c_void_p))
        return bytes(buf)

    def _check(self, t):
        if not t.is_cuda or not t.is_contiguous():
            raise L.CvhipError("RcclComm collectives need contiguous device tensors")

    def allreduce_(self, t, op="sum", stream=None):
        self._check(t)
        st = stream.cuda_stream if stream is not None else torch.cuda.current_stream().cuda_stream
        if t.dtype == torch.float32 and op == "sum":
            L.call("cvhip_allreduce_bucket", self._h, t.data_ptr(), t.numel(), st)
        else:
            L.call("cvhip_comm_allreduce", self._h, t.data_ptr(), t.numel(), _DTYPES[t.dtype], _OPS[op], st)
        return t

    def broadcast_(self, t, root=0, stream=None):
        self._check(t)
        st = stream.cuda_stream if stream is not None else torch.cuda.current_stream().cuda_stream
        L.call("cvhip_comm_broadcast", self._h, t.data_ptr(), t.numel() * t.element_size(), int(root), st)
        return t

    def reduce_scatter_allgather_(self, t, stream=None):
        self._check(t)
        if t.dtype != torch.float32 or t.numel() % self.world != 0 or t.numel() == 0:
            return self.allreduce_(t, "sum", stream)
        st = stream.cuda_stream if stream is not None else torch.cuda.current_stream().cuda_stream
        L.call("cvhip_comm_reduce_scatter_f32", self._h, t.data_ptr(), t.numel(), st)
        L.call("cvhip_comm_all_gather_f32", self._h, t.data_ptr(), t.numel(), st)
        return t

    def barrier(self):
        flag = torch.ones(1, dtype=torch.float32, device=self.device)
        self.allreduce_(flag)
        torch.cuda.synchronize()

    def close(self):
        if self._h:
            L.load().cvhip_comm_destroy(self._h)
            self._h = C.c_void_p()


class TorchDistComm(Comm):
    """Test transport over an initialised torch.distributed group (gloo)."""
    capturable = False

    def __init__(self, group=None):
        import torch.distributed as dist
        self._dist = dist
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self._works = []

    def allreduce_(self, t, op="sum", stream=None):
        dist = self._dist
        rop = {"sum": dist.ReduceOp.SUM, "max": dist.ReduceOp.MAX, "min": dist.ReduceOp.MIN}[op]
        if stream is not None and t.is_cuda:
            with torch.cuda.stream(stream):
                self._works.append(dist.all_reduce(t, op=rop, group=self.group, async_op=True))
        else:
            dist.all_reduce(t, op=rop, group=self.group)
        return t

    def broadcast_(self, t, root=0, stream=None):
        self._dist.broadcast(t, root, group=self.group)
        return t

    def barrier(self):
        self._dist.barrier(group=self.group)
        if torch.cuda.is_available():
            torch.cuda.synchronize()

    def wait(self):
        for w in self._works:
            w.wait()
        self._works = []


def _store(world, rank, timeout_s=600):
    import torch.distributed as dist
    addr = os.environ.get("MASTER_ADDR", "127.0.0.1")
    port = int(os.environ.get("MASTER_PORT", "29500"))
    agent = os.environ.get("TORCHELASTIC_USE_AGENT_STORE") == "True"  # torchrun's agent already serves the store on MASTER_PORT
    store = dist.TCPStore(addr, port, world, is_master=(rank == 0 and not agent), timeout=datetime.timedelta(seconds=timeout_s),
                          wait_for_workers=False)
    return dist.PrefixStore("cvhip_comm/%s" % os.environ.get("TORCHELASTIC_RESTART_COUNT", "0"), store)


def init_from_env(device=None):
    """RcclComm for this process from RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (what torch.distributed.run exports), or
    None for a single-process run. The caller must have selected its HIP device (or pass `device`)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world <= 1:
        return None
    if not L.load().cvhip_comm_available():
        raise L.CvhipError("librccl could not be bound: " + (L.load().cvhip_last_error() or b"").decode())
    store = _store(world, rank)
    if rank == 0:
        uid = RcclComm.unique_id()
        store.set("uid", uid)
    else:
        uid = store.get("uid")
    return RcclComm(world, rank, bytes(uid), device)


def default_comm(process_group=None):
    """The transport a FlatTrainState / SyncBN layer uses when none is given: the process-wide RcclComm if one was installed with
    `set_default`, else a TorchDistComm over the (default) torch.distributed group if that is initialised, else None."""
    if _DEFAULT[0] is not None and process_group is None:
        return _DEFAULT[0]
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            key = id(process_group) if process_group is not None else None
            c = _TD_CACHE.get(key)
            # a cached transport freezes world / rank: after destroy_process_group() + re-init (test-suite, elastic restart) the
            # entry of the default group would be stale — rebuild it whenever it no longer describes the live group
            if c is None or c.group is not process_group or c.world != dist.get_world_size(process_group) or c.rank != dist.get_rank(process_group):
                c = _TD_CACHE[key] = TorchDistComm(process_group)   # one transport object per group (shared async work list)
            return c
    except Exception:
        pass
    return None


_DEFAULT = [None]
_TD_CACHE = {}


def set_default(comm):
    _DEFAULT[0] = comm
    return comm
