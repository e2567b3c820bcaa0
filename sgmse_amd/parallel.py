"""Multi-GPU: utterances shard embarrassingly (reference enhancement.py:57 loops files one by one; the contiguous
per-rank split is the one its validation loop uses, model.py:212-223).  One process per GPU, no collective on the data
path; the only collective is one broadcast of the flat weight blob from rank 0 (RCCL over xGMI with backend 'nccl')."""
from __future__ import annotations

from typing import Dict, List, Tuple

import torch


def shard_range(n_items: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous [start, stop) of rank's items; the last rank takes the remainder (model.py:212-223)."""
    per = n_items // world_size
    start = rank * per
    stop = n_items if rank == world_size - 1 else start + per
    return start, stop


def flatten_state(state: Dict[str, torch.Tensor], device) -> Tuple[torch.Tensor, List[Tuple[str, torch.Size, int]]]:
    meta, off = [], 0
    for k, v in state.items():
        meta.append((k, v.shape, off))
        off += v.numel()
    flat = torch.empty(off, dtype=torch.float32, device=device)
    for (k, shp, o) in meta:
        flat[o:o + state[k].numel()].copy_(state[k].reshape(-1))
    return flat, meta


def broadcast_backbone_weights(dnn: torch.nn.Module, src: int = 0, force: bool = False):
    """One broadcast of all backbone parameters (65.6 M fp32 = 262 MB for ncsnpp) from ``src``; afterwards every
    rank's HIP engine is loaded from the received device buffer.  No-op without an initialised process group or with a
    single rank, unless ``force`` (the single-GPU test of the RCCL path runs the collective at world size 1)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not force):
        return
    params = dict(dnn.state_dict())
    dev = next(iter(params.values())).device
    flat, meta = flatten_state(params, dev)
    dist.broadcast(flat, src=src)
    with torch.no_grad():
        for (k, shp, o) in meta:
            params[k].copy_(flat[o:o + params[k].numel()].reshape(shp))
    dnn.mark_weights_changed()
