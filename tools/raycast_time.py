#!/usr/bin/env python3
"""Time render_raycasting on a scene (default c3)."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from sunvolumerender_amd import abi, host, scenes
name = sys.argv[1] if len(sys.argv) > 1 else "c3"
sc = scenes.make_scene(name)
dev = host.Device(0)
c = host.Canvas(dev, sc.width, sc.height)
scenes.apply_to_canvas(sc, c, 0)
c.SetRenderMode(host.Canvas.RENDER_MODE_RAYCASTING)
import os
if "RCL" in os.environ: dev.set_option(abi.OPT_RAYCAST_LANES_LOG2, int(os.environ["RCL"]))
c.paint(sync=True)
t0 = time.perf_counter(); n = 5
for _ in range(n):
    c.paint()
dev.synchronize()
dt = (time.perf_counter() - t0) / n
dev.set_option(abi.OPT_COUNT, 1); dev.reset_counters(); c.paint(sync=True); cnt = dev.counters()
print(f"raycast {name}: {dt*1e3:.3f} ms/frame  {sc.width*sc.height/dt/1e6:.1f} Mpix/s  steps={cnt['raycast_steps']} taps_executed={cnt['vol_taps_executed']} of {cnt['vol_taps']}")
c.close()
