#!/usr/bin/env python3
"""Timing sweep over library options on one scene (one process, one canvas).
usage: tools/sweep.py [--scene c3] [--depth 1] [--frames 64] [--spp 32] [--count] setting...
       setting = comma-separated name=value pairs; names: see KEYS below
       e.g.  tools/sweep.py --scene c3n defaults bound_cull=0 queue=2,park_end=16"""
import argparse
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from sunvolumerender_amd import abi, host, scenes  # noqa: E402

KEYS = {"bound_cull": abi.OPT_BOUND_CULL, "empty_skip": abi.OPT_EMPTY_SKIP, "ray_skip": abi.OPT_RAY_SKIP,
        "fl2": abi.OPT_FRAMES_PER_WAVE_LOG2, "kernel": abi.OPT_KERNEL, "frame_ahead": abi.OPT_FRAME_AHEAD, "fast_math": abi.OPT_FAST_MATH, "fold": abi.OPT_FOLD, "queue": abi.OPT_QUEUE, "park_end": abi.OPT_PARK_END, "fine_mask": abi.OPT_FINE_MASK, "row_order": abi.OPT_ROW_ORDER, "group": abi.OPT_GROUP_FRAMES, "lm": abi.OPT_LOCAL_MAJORANT, "lm_tune": abi.OPT_LM_TUNE, "light_cull": abi.OPT_LIGHT_CULL, "lm_sub": abi.OPT_LM_SUBCELLS, "park_cheap": abi.OPT_PARK_CHEAP, "pinhole_fast": abi.OPT_PINHOLE_FAST, "pool": abi.OPT_POOL, "trips": abi.OPT_TRIPS, "split": abi.OPT_SPLIT, "nan_guard": abi.OPT_NAN_GUARD, "env_nee": abi.OPT_ENV_NEE, "fast_bound": abi.OPT_FAST_BOUND}
ap = argparse.ArgumentParser()
ap.add_argument("--scene", default="c3")
ap.add_argument("--depth", type=int, default=1)
ap.add_argument("--frames", type=int, default=64)
ap.add_argument("--spp", type=int, default=32)
ap.add_argument("--count", action="store_true")
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--layout", type=int, default=0)
ap.add_argument("--shard", default="", help="strip_rows,rank,world")
ap.add_argument("settings", nargs="*", default=[""])
a = ap.parse_args()

sc = scenes.make_scene(a.scene, trace_depth=a.depth)
dev = host.Device(0, fatal_errors=False)
defaults = {k: dev.lib.svr_get_option(v) for k, v in KEYS.items()}
c = host.Canvas(dev, sc.width, sc.height)
scenes.apply_to_canvas(sc, c, a.layout)
if a.shard:
    dev.check(dev.lib.svr_set_row_shard(*[int(v) for v in a.shard.split(',')]))
print(f"# {a.scene} depth {a.depth}, {a.frames} frames per measurement, {a.spp} per call | {dev.info()}", flush=True)


def render():
    c.ReStartRender()
    n = 0
    while n < a.frames:
        if a.spp == 1:
            c.paint()
        else:
            c.paint_frames(a.spp)
        n += a.spp
    dev.synchronize()
    return n


for setting in a.settings:
    for k, v in defaults.items():
        if v >= -1:
            dev.lib.svr_set_option(KEYS[k], v)
    dev.lib.svr_clear_error()
    for kv in filter(None, ("" if setting == "defaults" else setting).split(",")):
        k, v = kv.split("=")
        dev.set_option(KEYS[k], int(v, 0))
    render()
    best = None
    for _ in range(a.reps):
        t0 = time.perf_counter()
        n = render()
        dt = (time.perf_counter() - t0) / n * 1e3
        best = dt if best is None else min(best, dt)
    extra = ""
    if a.count:
        dev.set_option(abi.OPT_COUNT, 1)
        dev.reset_counters()
        c.ReStartRender()
        c.paint_frames(min(a.spp, 8)) if a.spp > 1 else c.paint()
        dev.synchronize()
        k = dev.counters()
        dev.set_option(abi.OPT_COUNT, 0)
        p = max(1, k["paths"])
        extra = f"  taps/path {k['vol_taps'] / p:.1f} fetched {k['vol_taps_executed'] / p:.2f} culled {k['taps_bound_culled'] / p:.2f} iters {k['woodcock_iters'] / p:.1f} prefix/dda {k.get('iters_prefix_skipped', 0) / p:.1f}"
    print(f"{setting or 'defaults':40s} {best:8.4f} ms/frame {sc.width * sc.height / best / 1e3:9.1f} Msamples/s{extra}", flush=True)
c.close()
