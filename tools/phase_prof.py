#!/usr/bin/env python3
"""Phase profile of the lane machine (experiment build: SVR_EXTRA_HIPCC_FLAGS=-DSVR_TEST_HOOKS SVR_HIP_LIB=<path> python -m sunvolumerender_amd._build).
usage: SVR_HIP_LIB=... tools/phase_prof.py [--scene c3] [--depth 4] [--frames 32] setting..."""
import argparse, ctypes as C, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from sunvolumerender_amd import abi, host, scenes  # noqa: E402
import numpy as np

ap = argparse.ArgumentParser()
ap.add_argument("--scene", default="c3"); ap.add_argument("--depth", type=int, default=4); ap.add_argument("--frames", type=int, default=32)
ap.add_argument("settings", nargs="*", default=["queue=2"])
a = ap.parse_args()
sc = scenes.make_scene(a.scene, trace_depth=a.depth)
dev = host.Device(0, fatal_errors=False)
c = host.Canvas(dev, sc.width, sc.height)
scenes.apply_to_canvas(sc, c)
names = ["primary", "refill", "shade", "cheap", "fetch", "march", "end", "fold"]
fn = dev.lib.svr_debug_phase_profile
fn.restype, fn.argtypes = C.c_int, [C.c_void_p, C.c_int]
for setting in a.settings:
    for kv in filter(None, setting.split(",")):
        k, v = kv.split("=")
        dev.set_option(getattr(abi, "OPT_" + k.upper()), int(v))
    c.ReStartRender(); c.paint_frames(a.frames); dev.synchronize()
    dev.reset_counters()
    c.ReStartRender()
    t0 = time.perf_counter(); c.paint_frames(a.frames); dev.synchronize(); dt = time.perf_counter() - t0
    out = np.zeros(32, dtype=np.uint64)
    fn(out.ctypes.data_as(C.c_void_p), 32)
    tot = float(sum(out[2 * i] for i in range(8)))
    print(f"== {a.scene} depth {a.depth} {setting}: {dt / a.frames * 1e3:.4f} ms/frame; wave-cycles by phase (share, lane utilisation)")
    for i, n in enumerate(names):
        cyc, lc = float(out[2 * i]), float(out[2 * i + 1])
        if cyc:
            print(f"   {n:8s} {cyc / tot * 100:6.2f} %   util {lc / cyc / 64 * 100:5.1f} %")
    if out[16]:
        if a.depth > 1:
            print(f"   cheap-loop iterations {int(out[16])}, mean walking lanes {float(out[17]) / float(out[16]):.1f}")
        else:
            print(f"   primary walks (count=1 only): executed lane-iterations / (64 x longest walk of the wave) = {float(out[17]) / float(out[16]):.3f}")
c.close()
