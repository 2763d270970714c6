#!/usr/bin/env python3
"""Timing sweep over kernel options for one library build (select with SVR_HIP_LIB).
usage: tools/exp.py [--scene c3] [--frames 16] combos...   combo = kernel,layout,bpc,refill,spp,pipeline,skip"""
import argparse
import ctypes as C
import os
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from sunvolumerender_amd import abi, host, scenes  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--scene", default="c3")
ap.add_argument("--frames", type=int, default=16)
ap.add_argument("--depth", type=int, default=1)
ap.add_argument("--fl2", type=int, default=-1, help="log2 frames per wave (-1 auto)")
ap.add_argument("--unit", type=int, default=0)
ap.add_argument("--stop", type=int, default=0, help="timing ablation: 1 set-up only, 2 +whole-ray test, 3 +primary walk")
ap.add_argument("combos", nargs="+")
a = ap.parse_args()

sc = scenes.make_scene(a.scene, trace_depth=a.depth)
dev = host.Device(0)
if a.stop or a.unit:
    # options 100 / 101 exist only in experiment builds:
    #   SVR_EXTRA_HIPCC_FLAGS=-DSVR_TEST_HOOKS python -m sunvolumerender_amd._build --force
    dev.set_option(100, a.stop)
    dev.set_option(101, a.unit)
dev.set_option(abi.OPT_FRAMES_PER_WAVE_LOG2, a.fl2)
print("stop:", a.stop, "lib:", abi.library_path().name, "|", dev.info(), flush=True)
canv = {}
for combo in a.combos:
    k, lay, bpc, refill, spp, pipe, skip = [int(v) for v in combo.split(",")]
    if lay not in canv:
        for c in canv.values():
            c.close()
        canv.clear()
        c = host.Canvas(dev, sc.width, sc.height)
        scenes.apply_to_canvas(sc, c, lay)
        canv[lay] = c
    c = canv[lay]
    dev.set_option(abi.OPT_KERNEL, k)
    dev.set_option(abi.OPT_BLOCKS_PER_CU, bpc)
    dev.set_option(abi.OPT_REFILL_MIN_IDLE, refill)
    dev.set_option(abi.OPT_PIPELINE, pipe)
    dev.set_option(abi.OPT_EMPTY_SKIP, skip)
    best = None
    for rep in range(3):
        c.ReStartRender()
        dev.synchronize()
        t0 = time.perf_counter()
        n = 0
        while n < a.frames:
            if spp == 1:
                c.paint()
            else:
                c.paint_frames(spp)
            n += spp
        dev.synchronize()
        dt = (time.perf_counter() - t0) / n * 1e3
        best = dt if best is None else min(best, dt)
    print(f"kernel={k} layout={lay} bpc={bpc} refill={refill:2d} spp/call={spp:2d} pipeline={pipe} skip={skip}:  {best:7.3f} ms/frame  "
          f"{sc.width * sc.height / best / 1e3:8.1f} Msamples/s", flush=True)
for c in canv.values():
    c.close()
