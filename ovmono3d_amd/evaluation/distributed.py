"""One gather of fixed-width detection records to a destination rank.

Replaces ``comm.synchronize(); comm.gather(inference_json, dst=0)`` of the reference
(cubercnn/evaluation/omni3d_evaluation.py:717-720; Detectron2 pickles Python lists over a Gloo group).
Here the payload is the [n, 48] float32 record tensor: an all-gather of the per-rank counts followed by
point-to-point sends into the destination (a gatherv). Backend-agnostic over ``torch.distributed``:
``nccl`` (= RCCL over xGMI) on GPU tensors, ``gloo`` on CPU tensors (the CPU rehearsal test).
``libovm3d``'s ``ovm_gather_records`` is the same exchange directly on an ncclComm_t for non-Python hosts.
"""
from __future__ import annotations

from typing import List, Tuple

import torch
import torch.distributed as dist


def get_world_size() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def get_rank() -> int:
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def gather_records(rec: torch.Tensor, dst: int = 0) -> Tuple[torch.Tensor, List[int]]:
    """rec [n_local, W] (same W on all ranks). Returns (all records in rank order, counts) on ``dst``;
    (empty, counts) elsewhere."""
    world, rank = get_world_size(), get_rank()
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
