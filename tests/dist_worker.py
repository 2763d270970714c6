"""Worker of the multi-process sharding tests (started by torch.distributed.run, one process per rank).

    python -m torch.distributed.run --nproc-per-node N ... tests/dist_worker.py <renderer> <mode> <out.npz>

renderer "oracle": the ranks render their strips with the CPU oracle (CPU-only test of the collective plumbing);
renderer "hip": every rank drives libsvr_hip.so on GPU 0 through svr_set_row_shard (the product path; several ranks
share the one GPU of the test box, backend gloo).  Protocol, in every case: 3 progressive frames -> assemble ->
3 more frames -> assemble again (the ranks' accumulators must survive the first assembly untouched) -> rank 0 tone-maps
the assembled frame and saves both assemblies."""
import ctypes as C
import os
import sys
from pathlib import Path

import numpy as np
import torch
import torch.distributed as tdist

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

STRIP = 8
SCENE = ("tiny_head", 2)      # name, trace depth


def main():
    renderer, mode, out_path = sys.argv[1:4]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    tdist.init_process_group("gloo", rank=rank, world_size=world)
    from sunvolumerender_amd import dist, scenes

    sc = scenes.make_scene(SCENE[0], trace_depth=SCENE[1])
    H, W = sc.height, sc.width
    asm = dist.FrameAssembler(H, W, STRIP, rank, world, dst=0, mode=mode)
    frames = []

    if renderer == "oracle":
        from oracle import binding

        o = binding.OracleScene(sc)
        hdr = o.new_hdr()
        rows = dist.owned_rows(H, STRIP, rank, world)
        runs = np.split(rows, np.where(np.diff(rows) != 1)[0] + 1) if len(rows) else []

        def render(f):
            for r in runs:
                o.render_pathtracer(hdr, f, window=(0, int(r[0]), W, int(r[-1]) + 1), nthreads=2)

        def accumulator():
            return torch.from_numpy(hdr)

        def tonemap(frame):
            return o.hdr_to_ldr(frame.numpy())
    else:
        from sunvolumerender_amd import abi, host

        dev = host.Device(0, fatal_errors=False)
        canvas = host.Canvas(dev, W, H)
        scenes.apply_to_canvas(sc, canvas)
        dist.shard(dev, STRIP, rank, world)

        def render(f):
            assert canvas.renderParams.frameNo == f
            canvas.paint()

        def accumulator():
            dev.synchronize()
            return torch.from_numpy(canvas.read_hdr())

        def tonemap(frame):
            # the full-frame tone map of the assembled frame: svr_hdr_to_ldr_frame ignores the shard
            a = np.ascontiguousarray(frame.numpy())
            d_hdr, d_img = dev.malloc(a.nbytes), dev.malloc(H * W * 4)
            dev.to_device(d_hdr, a)
            dev.check(dev.lib.svr_hdr_to_ldr_frame(C.c_void_p(d_img), C.c_void_p(d_hdr), W, H))
            dev.synchronize()
            img = dev.to_host(d_img, (H, W, 4), np.uint8)
            dev.free(d_hdr); dev.free(d_img)
            return img

    for f in range(3):
        render(f)
    mine = accumulator()
    before = mine.clone()
    a3 = asm.assemble(mine)
    assert torch.equal(mine, before), "assembling modified the rank's accumulator"
    if rank == 0:
        frames.append(a3.clone())
    for f in range(3, 6):
        render(f)
    a6 = asm.assemble(accumulator())
    if rank == 0:
        np.savez(out_path, hdr3=frames[0].numpy(), hdr6=a6.numpy(), img6=tonemap(a6))
    tdist.barrier()
    tdist.destroy_process_group()
    if renderer == "hip":
        canvas.close()


if __name__ == "__main__":
    main()
