"""Multi-GPU frame sharding: one process per GPU, interleaved row strips, global per-pixel
seeds (so the assembled frame is bit-identical to a single-GPU render), and ONE collective per output:
a reduce(SUM) of the HDR accumulation buffers onto rank 0 -- strips are disjoint and every rank's buffer
is zero outside its own strips, so the sum is exact.  Backend: torch.distributed ("nccl" = RCCL over xGMI
on the GPU box, "gloo" in the CPU tests)."""
from __future__ import annotations

import numpy as np


def owned_rows(height: int, strip_rows: int, rank: int, world: int) -> np.ndarray:
    """Rows y with (y // strip_rows) % world == rank -- the same rule as svr_set_row_shard."""
    y = np.arange(height)
    if world <= 1:
        return y
    return y[(y // strip_rows) % world == rank]


def owned_row_count(height: int, strip_rows: int, rank: int, world: int) -> int:
    return int(len(owned_rows(height, strip_rows, rank, world)))


def shard(dev, strip_rows: int, rank: int, world: int):
    """Restrict this process's renderer to its strips (no-op for world == 1)."""
    dev.check(dev.lib.svr_set_row_shard(int(strip_rows), int(rank), int(world)))


def reduce_hdr(hdr_tensor, dst: int = 0):
    """Sum the per-rank HDR buffers onto `dst` (torch tensor on the rank's device; in place)."""
    import torch.distributed as dist

    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.reduce(hdr_tensor, dst=dst, op=dist.ReduceOp.SUM)
    return hdr_tensor


def allgather_strips(hdr_tensor, height: int, width: int, strip_rows: int):
    """Alternative to reduce_hdr when every rank needs the frame: all-reduce(SUM) of the disjoint strips."""
    import torch.distributed as dist

    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(hdr_tensor, op=dist.ReduceOp.SUM)
    return hdr_tensor
