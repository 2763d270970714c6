#!/usr/bin/env python3
"""Phase profile of the local-majorant pool kernel (experiment build: SVR_EXTRA_HIPCC_FLAGS=-DSVR_TEST_HOOKS SVR_HIP_LIB=<path> python -m sunvolumerender_amd._build).
Wave cycles (s_memtime) per phase of k_trace_lm_pool and the lanes that had work in it.
usage: SVR_HIP_LIB=... tools/lm_phase_prof.py [--scene c3] [--frames 64] [name=value ...]"""
import argparse, ctypes as C, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from sunvolumerender_amd import abi, host, scenes  # noqa: E402
import numpy as np

ap = argparse.ArgumentParser()
ap.add_argument("--scene", default="c3"); ap.add_argument("--depth", type=int, default=1); ap.add_argument("--frames", type=int, default=64)
ap.add_argument("settings", nargs="*", default=[])
a = ap.parse_args()
sc = scenes.make_scene(a.scene, trace_depth=a.depth)
dev = host.Device(0, fatal_errors=False)
c = host.Canvas(dev, sc.width, sc.height)
scenes.apply_to_canvas(sc, c)
names = ["gen", "refill", "shade", "dda", "tentative", "march", "settle", "fold"]
fn = dev.lib.svr_debug_phase_profile
fn.restype, fn.argtypes = C.c_int, [C.c_void_p, C.c_int]
dev.set_option(abi.OPT_LOCAL_MAJORANT, 1)
for kv in a.settings:
    k, v = kv.split("=")
    dev.set_option(getattr(abi, "OPT_" + k.upper()), int(v, 0))
c.ReStartRender(); c.paint_frames(a.frames); dev.synchronize()
dev.reset_counters()
c.ReStartRender()
t0 = time.perf_counter(); c.paint_frames(a.frames); dev.synchronize(); dt = time.perf_counter() - t0
out = np.zeros(32, dtype=np.uint64)
fn(out.ctypes.data_as(C.c_void_p), 32)
tot = float(sum(out[2 * i] for i in range(8)))
print(f"== {a.scene} depth {a.depth} local majorants {' '.join(a.settings)}: {dt / a.frames * 1e3:.4f} ms/frame; wave time by phase (share of the profiled time, lanes with work)")
for i, n in enumerate(names):
    cyc, lc = float(out[2 * i]), float(out[2 * i + 1])
    if cyc:
        print(f"   {n:10s} {cyc / tot * 100:6.2f} %   lanes {lc / cyc:5.1f}")
c.close()
