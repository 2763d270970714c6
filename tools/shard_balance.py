#!/usr/bin/env python3
"""Load balance of the row-strip sharding: renders every rank's strips of c3 on ONE GPU, one rank after the other,
and prints the per-rank times (the slowest rank bounds the multi-GPU step)."""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from sunvolumerender_amd import host, scenes  # noqa: E402

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
sc = scenes.make_scene("c3")
dev = host.Device(0)
c = host.Canvas(dev, sc.width, sc.height)
scenes.apply_to_canvas(sc, c, 0)
for strip in (8, 16, 32, 64):
    times = []
    for r in range(world):
        dev.check(dev.lib.svr_set_row_shard(strip, r, world))
        c.ReStartRender(); c.paint_frames(32); dev.synchronize()
        best = 1e9
        for rep in range(3):
            c.ReStartRender()
            dev.synchronize()
            t0 = time.perf_counter()
            for _ in range(4):
                c.paint_frames(32)
            dev.synchronize()
            best = min(best, (time.perf_counter() - t0) / 4 * 1e3)
        times.append(best)
    dev.lib.svr_set_row_shard(0, 0, 1)
    mean = sum(times) / len(times)
    print(f"world {world} strip {strip:3d}: per-rank ms per 32-spp step " + " ".join(f"{t:.3f}" for t in times) +
          f" | max/mean {max(times) / mean:.3f}  sum {sum(times):.3f}")
c.close()
