"""Communicators for the multi-GPU propagation loops (csrc/sharded.hip through cleora_amd/sharded.py; the model in tests/sharded_model.py).

The product path is RcclComm: the C-ABI communicator of libcleora_hip.so (csrc/comm.hip, include/cleora_hip.h
"multi-GPU exchange steps"), which binds RCCL directly — the same entry points a Rust host would call
(INTEGRATION.md).  torch.distributed is only the LAUNCHER there: it hands the 128-byte RCCL unique id from
rank 0 to the other ranks.  TorchComm runs the same interface over a torch.distributed process group: the
CPU tests use it with gloo (no GPU in the build container), and a one-GPU box can exercise the N > 1 code
path with it (RCCL refuses two ranks on one device).  LocalComm is world 1.

Semantics shared by the three (mirroring stream-ordered NCCL calls):
  * every collective is in place;
  * `allgather_rows` and `allreduce_async` are ASYNCHRONOUS with respect to the compute stream: they start
    after the work enqueued on it so far, run beside whatever is enqueued next, and are complete (for the
    compute stream) only after `join()`;
  * `allreduce`, `broadcast`, `alltoall` are ordered on the compute stream like a kernel launch.
The reference has no counterpart: pycleora is single-process (src/embedding.rs:59-63).
"""
import ctypes

import numpy as np

from . import _hip


class LocalComm:
    """World of one: every collective is the identity."""
    rank, world = 0, 1

    def allgather_rows(self, buf, row_bounds):
        pass

    def allreduce_async(self, t):
        pass

    def allreduce(self, t):
        pass

    def broadcast(self, t, root=0):
        pass

    def alltoall(self, send, recv):
        recv.copy_(send)

    def join(self):
        pass

    def close(self):
        pass


class TorchComm:
    """The interface over a torch.distributed group (gloo in the CPU tests)."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist, self.group = dist, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self._works = []

    def allgather_rows(self, buf, row_bounds):
        dist = self.dist
        sizes = {row_bounds[r + 1] - row_bounds[r] for r in range(self.world)}
        if len(sizes) == 1:
            mine = buf[row_bounds[self.rank]:row_bounds[self.rank + 1]]
            self._works.append(dist.all_gather_into_tensor(buf[row_bounds[0]:row_bounds[-1]], mine,
                                                           group=self.group, async_op=True))
            return
        for r in range(self.world):                      # all-gather-v: one broadcast per owner
            if row_bounds[r + 1] > row_bounds[r]:
                self._works.append(dist.broadcast(buf[row_bounds[r]:row_bounds[r + 1]],
                                                  src=dist.get_global_rank(self.group, r) if self.group else r,
                                                  group=self.group, async_op=True))

    def allreduce_async(self, t):
        self._works.append(self.dist.all_reduce(t, group=self.group, async_op=True))

    def allreduce(self, t):
        self.dist.all_reduce(t, group=self.group)

    def broadcast(self, t, root=0):
        self.dist.broadcast(t, src=self.dist.get_global_rank(self.group, root) if self.group else root,
                            group=self.group)

    def alltoall(self, send, recv):
        self.dist.all_to_all_single(recv.view(-1), send.reshape(-1), group=self.group)

    def join(self):
        for w in self._works:
            w.wait()
        self._works = []

    def close(self):
        self.join()


class RcclComm:
    """The C-ABI communicator, collectives on a communication stream of its own.  Two transports under the same calls
    (csrc/comm.hip, csrc/peer.hip):
      * RCCL over xGMI (the default; `enable_peer()` adds the peer-direct all-gather as a third algorithm);
      * local=True: no RCCL at all — hipIpc mappings between the ranks of one node: peer-direct all-gather, all-reduce summed
        in rank order on every rank, broadcast.  Unlike RCCL it accepts several ranks on ONE device, which is how the multi-rank
        loops run through the C ABI on a one-GPU box (tests, bench.py --share-gpu).

    stream_fn() returns the compute stream (a hipStream_t as int, None = the default stream); with torch
    present it defaults to torch's current stream on `device`."""

    def __init__(self, unique_id, rank, world, device=0, stream_fn=None, local=False):
        self.L = _hip.lib()
        self.rank, self.world, self.device = int(rank), int(world), int(device)
        self.local = bool(local)
        if len(unique_id) != _hip.COMM_ID_BYTES:
            raise ValueError(f"unique_id must be {_hip.COMM_ID_BYTES} bytes")
        idbuf = (ctypes.c_char * _hip.COMM_ID_BYTES).from_buffer_copy(bytes(unique_id))
        _hip.check(self.L.cleora_set_device(self.device))
        h = _hip.vp()
        create = self.L.cleora_comm_create_local if self.local else self.L.cleora_comm_create
        _hip.check(create(ctypes.cast(idbuf, _hip.vp), self.rank, self.world, self.device, ctypes.byref(h)))
        self.handle = h
        s = _hip.vp()
        _hip.check(self.L.cleora_stream_create(ctypes.byref(s)))
        self.comm_stream = s
        self._stream_fn = stream_fn or self._torch_stream
        self._pending = False

    @staticmethod
    def unique_id(local=False):
        buf = (ctypes.c_char * _hip.COMM_ID_BYTES)()
        L = _hip.lib()
        _hip.check((L.cleora_comm_local_id if local else L.cleora_comm_unique_id)(ctypes.cast(buf, _hip.vp)))
        return bytes(buf)

    @classmethod
    def from_torch_distributed(cls, device, group=None, local=False):
        """torch.distributed (any backend) as the launcher: rank 0 draws the id, the group broadcasts it."""
        import torch.distributed as dist
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        box = [cls.unique_id(local) if rank == 0 else None]
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group else 0, group=group)
        return cls(box[0], rank, world, device, local=local)

    def enable_peer(self):
        """Adds the peer-direct transport to an RCCL communicator (collective): ALLGATHER_PEER becomes selectable."""
        _hip.check(self.L.cleora_comm_enable_peer(self.handle))

    def register(self, t):
        """Every rank passes its copy of a buffer the peer-direct all-gather will work in (collective; a no-op without the peer
        transport).  `t`: a torch tensor / DevArray-like with data_ptr() or .ptr, and its byte size."""
        p = t.data_ptr() if hasattr(t, "data_ptr") else t.ptr
        nbytes = t.numel() * t.element_size() if hasattr(t, "numel") else t.nbytes
        _hip.check(self.L.cleora_comm_register(self.handle, p, nbytes))

    def unregister(self, t):
        p = t.data_ptr() if hasattr(t, "data_ptr") else t.ptr
        _hip.check(self.L.cleora_comm_unregister(self.handle, p))

    def check(self):
        """Raises if a device-side wait of this rank ever timed out (a peer died)."""
        _hip.check(self.L.cleora_comm_check(self.handle))

    def _torch_stream(self):
        try:
            import torch
            return torch.cuda.current_stream(self.device).cuda_stream
        except Exception:
            return None

    def _compute(self):
        s = self._stream_fn()
        return _hip.vp(s) if s else None

    def set_allgather(self, algo):
        _hip.check(self.L.cleora_comm_set_allgather(self.handle, int(algo)))

    def _fork(self):
        # the collective starts after everything enqueued on the compute stream so far
        _hip.check(self.L.cleora_stream_wait_stream(self.comm_stream, self._compute()))
        self._pending = True

    def allgather_rows(self, buf, row_bounds):
        assert buf.is_contiguous() and buf.dim() == 2
        d = buf.shape[1]
        off = np.asarray([b * d for b in row_bounds], dtype=np.uint64)
        self._fork()
        _hip.check(self.L.cleora_allgatherv_f32_dev(self.handle, buf.data_ptr(), _hip.ptr(off), self.comm_stream))

    def _allreduce(self, t, stream):
        import torch
        assert t.is_contiguous()
        if t.dtype == torch.float32:
            _hip.check(self.L.cleora_allreduce_f32_dev(self.handle, t.data_ptr(), t.numel(), stream))
        elif t.dtype == torch.float64:
            _hip.check(self.L.cleora_allreduce_f64_dev(self.handle, t.data_ptr(), t.numel(), stream))
        else:
            raise TypeError(f"allreduce of {t.dtype} is not part of the path")

    def allreduce_async(self, t):
        self._fork()
        self._allreduce(t, self.comm_stream)

    def allreduce(self, t):
        self._allreduce(t, self._compute())

    def broadcast(self, t, root=0):
        assert t.is_contiguous()
        _hip.check(self.L.cleora_broadcast_dev(self.handle, t.data_ptr(), t.numel() * t.element_size(), int(root),
                                               self._compute()))

    def alltoall(self, send, recv):
        assert send.is_contiguous() and recv.is_contiguous() and send.numel() == recv.numel()
        _hip.check(self.L.cleora_alltoall_f32_dev(self.handle, send.data_ptr(), recv.data_ptr(),
                                                  send.numel() // self.world, self._compute()))

    def join(self):
        if self._pending:
            _hip.check(self.L.cleora_stream_wait_stream(self._compute(), self.comm_stream))
            self._pending = False

    def close(self):
        if getattr(self, "handle", None):
            self.L.cleora_stream_sync(self.comm_stream)
            self.L.cleora_comm_destroy(self.handle)
            self.L.cleora_stream_destroy(self.comm_stream)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def default_comm(group=None):
    """LocalComm for one process; otherwise the torch.distributed group as a TorchComm (tests / dev runs —
    the product path passes an RcclComm explicitly)."""
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            return TorchComm(group)
    except ImportError:
        pass
    return LocalComm()
