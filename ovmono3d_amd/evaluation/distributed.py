"""One gather of fixed-width detection records to a destination rank.

Replaces ``comm.synchronize(); comm.gather(inference_json, dst=0)`` of the reference
(cubercnn/evaluation/omni3d_evaluation.py:717-720; Detectron2 pickles Python lists over a Gloo group).
Here the payload is the [n, 48] float32 record tensor: an all-gather of the per-rank counts followed by
point-to-point sends into the destination (a gatherv). Backend-agnostic over ``torch.distributed``:
``nccl`` (= RCCL over xGMI) on GPU tensors, ``gloo`` on CPU tensors (the CPU rehearsal test).
``libovm3d``'s ``ovm_gather_records`` is the same exchange directly on an ncclComm_t for non-Python hosts.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def get_world_size() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def get_rank() -> int:
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


class NativeComm:
    """RCCL communicator owned by libovm3d (``ovm_comm_init``): the gather a non-Python host of the C ABI would use, driven
    from Python. The 128-byte unique id travels over the host's own rendezvous; ``from_torch_distributed`` broadcasts it
    through the already initialised ``torch.distributed`` group."""

    def __init__(self, unique_id: bytes, rank: int, world: int, device: torch.device):
        import ctypes as C
        from .. import lib as _lib
        self.L, self.rank, self.world, self.device = _lib.load(), int(rank), int(world), torch.device(device)
        self._comm = C.c_void_p()
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        _lib.check(self.L.ovm_comm_init(buf, self.rank, self.world, self.device.index or 0, C.byref(self._comm)), what="ovm_comm_init")

    @staticmethod
    def new_unique_id() -> bytes:
        import ctypes as C
        from .. import lib as _lib
        buf = (C.c_uint8 * 128)()
        _lib.check(_lib.load().ovm_comm_unique_id(buf), what="ovm_comm_unique_id")
        return bytes(buf)

    @classmethod
    def from_torch_distributed(cls, device: torch.device) -> "NativeComm":
        obj = [cls.new_unique_id() if get_rank() == 0 else None]
        if get_world_size() > 1:
            dist.broadcast_object_list(obj, src=0)
        return cls(obj[0], get_rank(), get_world_size(), device)

    def gather(self, rec: torch.Tensor) -> Tuple[torch.Tensor, List[int]]:
        """rec [n, 48] float32 on the comm's device -> (all records in rank order on rank 0, counts)."""
        import ctypes as C
        from .. import lib as _lib
        assert rec.is_cuda and rec.dtype == torch.float32 and rec.dim() == 2 and rec.shape[1] == _lib.OVM_REC_FLOATS
        rec = rec.contiguous()
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        counts = (C.c_int32 * self.world)()
        n = int(rec.shape[0])
        _lib.check(self.L.ovm_gather_counts(self._comm, self.rank, self.world, n, counts, stream), what="ovm_gather_counts")
        total = sum(counts) if self.rank == 0 else 0
        out = torch.empty((total, rec.shape[1]), dtype=torch.float32, device=self.device)
        _lib.check(self.L.ovm_gather_records(self._comm, self.rank, self.world, rec.data_ptr() if n else None, n,
                                             out.data_ptr() if total else None, counts, stream), what="ovm_gather_records")
        return out, [int(c) for c in counts]

    def close(self):
        if self._comm:
            self.L.ovm_comm_destroy(self._comm)
            self._comm = None


_native_comm: Optional[NativeComm] = None


def set_native_comm(comm: Optional[NativeComm]) -> None:
    """Route ``gather_records`` of device tensors through libovm3d's own RCCL gather (``ovm_gather_records``)."""
    global _native_comm
    _native_comm = comm


def gather_records(rec: torch.Tensor, dst: int = 0) -> Tuple[torch.Tensor, List[int]]:
    """rec [n_local, W] (same W on all ranks). Returns (all records in rank order, counts) on ``dst``;
    (empty, counts) elsewhere."""
    world, rank = get_world_size(), get_rank()
    if _native_comm is not None and rec.is_cuda and dst == 0 and _native_comm.world == world:
        return _native_comm.gather(rec)
    if world == 1:
        return rec, [int(rec.shape[0])]
    dev = rec.device
    cnt = torch.tensor([rec.shape[0]], dtype=torch.int64, device=dev)
    allc = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(allc, cnt)
    counts = [int(c.item()) for c in allc]
    W = rec.shape[1]
    if rank == dst:
        out = torch.empty((sum(counts), W), dtype=rec.dtype, device=dev)
        ofs, reqs = 0, []
        for r in range(world):
            sl = out[ofs: ofs + counts[r]]
            if r == dst:
                sl.copy_(rec)
            elif counts[r] > 0:
                reqs.append(dist.irecv(sl, src=r))
            ofs += counts[r]
        for q in reqs:
            q.wait()
        return out, counts
    if counts[rank] > 0:
        dist.send(rec.contiguous(), dst=dst)
    return rec.new_zeros((0, W)), counts
