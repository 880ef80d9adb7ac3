"""Data-parallel plumbing for the one-process-per-GPU launch (torch.distributed; backend "nccl" is
RCCL on ROCm, "gloo" in the CPU tests). The hot path itself has no collective: images are
independent units (reference image_normalize is per image, src/visp/image.cpp:537-576), so each
rank owns a contiguous shard of the batch. RCCL is used once, at load, to replicate the packed
weight arena that rank 0 built from the GGUF file, and optionally to gather outputs."""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_range(n_items: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous [begin, end) shard of n_items for `rank`; sizes differ by at most one."""
    base, rem = divmod(n_items, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def broadcast_bytes(buf: torch.Tensor, src: int = 0) -> torch.Tensor:
    """Replicates a uint8 buffer (the packed weight arena) from `src` to every rank, in place."""
    assert buf.dtype == torch.uint8 and buf.is_contiguous()
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(buf, src=src)
    return buf


def max_over_ranks(value: float, device) -> float:
    """Whole-job time of a step loop = the slowest rank's time."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_outputs(local: torch.Tensor, dst: int = 0):
    """Optional: collects every rank's [b_i, h, w] outputs on `dst` (shards may differ in size by one).
    Returns the concatenated tensor on dst, None elsewhere."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return local
    world, rank = dist.get_world_size(), dist.get_rank()
    sizes = [torch.zeros(1, dtype=torch.int64, device=local.device) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device))
    n_max = int(max(s.item() for s in sizes))
    pad = torch.zeros((n_max,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    bufs = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
    dist.gather(pad, bufs, dst=dst)
    if rank != dst:
        return None
    return torch.cat([b[: int(s.item())] for b, s in zip(bufs, sizes)])
