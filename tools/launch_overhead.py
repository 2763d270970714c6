#!/usr/bin/env python3
"""Fixed cost of one trace launch: kernel time (HIP events) for windows of growing size on c3."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from sunvolumerender_amd import abi, host, scenes
sc = scenes.make_scene("c3")
dev = host.Device(0)
c = host.Canvas(dev, sc.width, sc.height)
scenes.apply_to_canvas(sc, c, 0)
for (x0, y0, x1, y1) in [(0, 0, 8, 8), (508, 508, 516, 516), (384, 508, 640, 516), (0, 504, 1024, 520), (0, 448, 1024, 576), (0, 0, 1024, 1024)]:
    dev.check(dev.lib.svr_set_render_window(x0, y0, x1, y1))
    c.ReStartRender(); c.paint_frames(32); dev.synchronize()
    dev.set_option(abi.OPT_TIMING, 1); dev.check(dev.lib.svr_reset_kernel_time())
    c.ReStartRender(); dev.synchronize()
    for _ in range(8): c.paint_frames(32)
    dev.synchronize()
    k_ms, k_n = dev.kernel_time()
    dev.set_option(abi.OPT_TIMING, 0)
    print(f"window {x1-x0:4d} x {y1-y0:4d} at ({x0},{y0}): trace kernel {k_ms / k_n * 1e3:8.1f} us")
dev.lib.svr_set_render_window(0, 0, -1, -1)
c.close()
