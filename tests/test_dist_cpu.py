"""N>1 path on CPU: two gloo ranks render their interleaved strips with the oracle (standing in for the GPU
kernel, which is bit-identical to it), reduce(SUM) the HDR buffers onto rank 0, and rank 0 compares the
assembled frame with the single-process render."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import torch
import torch.distributed as tdist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parents[1]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_path):
    sys.path.insert(0, str(ROOT))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    tdist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import binding
    from sunvolumerender_amd import dist, scenes

    sc = scenes.make_scene("tiny_head", trace_depth=2)
    o = binding.OracleScene(sc)
    hdr = o.new_hdr()
    strip = 8
    rows = dist.owned_rows(sc.height, strip, rank, world)
    for f in range(2):
        # contiguous runs of owned rows -> windows
        start = None
        prev = None
        for y in list(rows) + [None]:
            if start is None:
                start, prev = y, y
            elif y is not None and y == prev + 1:
                prev = y
            else:
                o.render_pathtracer(hdr, f, window=(0, int(start), sc.width, int(prev) + 1), nthreads=2)
                start, prev = y, y
    t = torch.from_numpy(hdr)
    dist.reduce_hdr(t, dst=0)
    if rank == 0:
        np.save(out_path, t.numpy())
    tdist.barrier()
    tdist.destroy_process_group()


def test_two_rank_strip_sharding_matches_single_process(tmp_path):
    sys.path.insert(0, str(ROOT))
    from oracle import binding
    from sunvolumerender_amd import scenes
    from tests.util import assert_bit_exact

    out = str(tmp_path / "assembled.npy")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    sc = scenes.make_scene("tiny_head", trace_depth=2)
    o = binding.OracleScene(sc)
    ref = o.new_hdr()
    for f in range(2):
        o.render_pathtracer(ref, f)
    assert_bit_exact(np.load(out), ref, "2-rank assembled frame")
