"""Multi-GPU frame sharding: one process per GPU, interleaved row strips, global per-pixel seeds
(so the assembled frame is bit-identical to a single-GPU render), and ONE collective per output frame.

Each rank's renderer only touches the rows it owns (svr_set_row_shard); its HDR accumulator is zero elsewhere.
`FrameAssembler` builds the whole frame on the destination rank WITHOUT touching the ranks' live accumulators, so a
progressive render can be assembled again and again (reduce -> more frames -> reduce):

* mode "gather" (default): every rank packs the rows it owns (H/world rows) and sends only those -- one
  `dist.gather`, i.e. with RCCL one point-to-point transfer per peer over its own xGMI link (6 MiB per peer for a
  2048^2 frame on 8 GPUs); rank `dst` scatters the strips into the frame.
* mode "reduce": `dist.reduce(SUM)` of full-size copies (strips are disjoint and the rest is zero, so the sum is
  exact) -- what BASELINE.json's north star names; it moves `world` times the bytes of the gather.

Backend: torch.distributed ("nccl" = RCCL on the GPU box, "gloo" in the CPU tests and one-GPU rehearsals).  Tensors
live wherever the process group wants them (CUDA for nccl, CPU for gloo).
"""
from __future__ import annotations

from typing import List

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


class FrameAssembler:
    """Assembles the W x H x 3 float32 HDR frame on rank `dst` from the ranks' strip-sharded accumulators.

    All staging tensors are allocated once (device = that of the first tensor passed to `assemble`), so the timed
    region of a benchmark contains the pack, the collective and the scatter, and no allocation."""

    def __init__(self, height: int, width: int, strip_rows: int, rank: int, world: int, dst: int = 0, mode: str = "gather"):
        if mode not in ("gather", "reduce"):
            raise ValueError(mode)
        self.H, self.W, self.strip, self.rank, self.world, self.dst, self.mode = height, width, strip_rows, rank, world, dst, mode
        self.rows: List[np.ndarray] = [owned_rows(height, strip_rows, r, world) for r in range(world)]
        self.max_rows = max(len(r) for r in self.rows)
        self._dev = None
        self._idx = None          # row-index tensors (this rank's; on dst: every rank's)
        self._mine = None         # [max_rows, W*3] packed strips of this rank (zero padded)
        self._parts = None        # dst: one [max_rows, W*3] buffer per rank
        self.frame = None         # dst: the assembled [H, W, 3] frame

    def _prepare(self, like):
        import torch

        if self._dev == like.device:
            return
        self._dev = like.device
        to_idx = lambda r: torch.from_numpy(np.ascontiguousarray(r, dtype=np.int64)).to(like.device)
        self._idx = [to_idx(r) if (q == self.rank or self.rank == self.dst) else None for q, r in enumerate(self.rows)]
        self._mine = torch.zeros((self.max_rows, self.W * 3), dtype=torch.float32, device=like.device)
        if self.rank == self.dst:
            self.frame = torch.zeros((self.H, self.W, 3), dtype=torch.float32, device=like.device)
            if self.mode == "gather":
                self._parts = [torch.zeros_like(self._mine) for _ in range(self.world)]
        elif self.mode == "reduce":
            self.frame = torch.zeros((self.H, self.W, 3), dtype=torch.float32, device=like.device)

    def assemble(self, hdr):
        """hdr: this rank's accumulator, a float32 tensor of H*W*3 elements (any shape).  Returns the assembled
        [H, W, 3] frame on rank `dst`, None on the others.  `hdr` is not modified."""
        import torch
        import torch.distributed as dist

        self._prepare(hdr)
        rows2d = hdr.view(self.H, self.W * 3)
        if self.world <= 1 or not (dist.is_available() and dist.is_initialized()):
            self.frame.view(self.H, self.W * 3).copy_(rows2d)
            return self.frame
        if self.mode == "reduce":
            self.frame.view(self.H, self.W * 3).copy_(rows2d)
            dist.reduce(self.frame, dst=self.dst, op=dist.ReduceOp.SUM)
            return self.frame if self.rank == self.dst else None
        n = len(self.rows[self.rank])
        if n:
            torch.index_select(rows2d, 0, self._idx[self.rank], out=self._mine[:n])
        dist.gather(self._mine, gather_list=self._parts if self.rank == self.dst else None, dst=self.dst)
        if self.rank != self.dst:
            return None
        out2d = self.frame.view(self.H, self.W * 3)
        for r in range(self.world):
            k = len(self.rows[r])
            if k:
                out2d.index_copy_(0, self._idx[r], self._parts[r][:k])
        return self.frame

    def bytes_sent_per_rank(self) -> int:
        """Payload one non-destination rank puts on the wire per assembled frame."""
        rows = self.max_rows if self.mode == "gather" else self.H
        return rows * self.W * 3 * 4


def assemble_hdr(hdr, height: int, width: int, strip_rows: int, dst: int = 0, mode: str = "gather"):
    """One-shot convenience wrapper around FrameAssembler for the current process group."""
    import torch.distributed as dist

    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    return FrameAssembler(height, width, strip_rows, rank, world, dst, mode).assemble(hdr)
