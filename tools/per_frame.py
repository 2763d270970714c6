#!/usr/bin/env python3
"""The reference's own call protocol (gui/canvas.cpp:96-116): one render_pathtracer call + one device synchronisation per
frame, frameNo++.  Times the ramp (frames 0..63: frames are traced ahead in batches of 1, 2, 4 ...) and the steady state
(frames 64..N-1) separately.  usage: tools/per_frame.py [--scene c3] [--frames 1088] [--no-sync] [name=value ...]"""
import argparse
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from sunvolumerender_amd import abi, host, scenes  # noqa: E402


def measure(dev, canvas, frames, sync=True, ramp=64):
    canvas.ReStartRender()
    dev.synchronize()
    t0 = time.perf_counter()
    t_ramp = None
    for f in range(frames):
        canvas.paint(sync=sync)
        if f + 1 == ramp:
            dev.synchronize()
            t_ramp = time.perf_counter()
    dev.synchronize()
    t1 = time.perf_counter()
    px = canvas.W * canvas.H
    return px * ramp / (t_ramp - t0) / 1e6, px * (frames - ramp) / (t1 - t_ramp) / 1e6


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="c3")
    ap.add_argument("--depth", type=int, default=1)
    ap.add_argument("--frames", type=int, default=1088)
    ap.add_argument("--no-sync", action="store_true")
    ap.add_argument("settings", nargs="*")
    a = ap.parse_args()
    sc = scenes.make_scene(a.scene, trace_depth=a.depth)
    dev = host.Device(0, fatal_errors=False)
    c = host.Canvas(dev, sc.width, sc.height)
    scenes.apply_to_canvas(sc, c)
    for kv in a.settings:
        k, v = kv.split("=")
        dev.set_option(getattr(abi, "OPT_" + k.upper()), int(v))
    measure(dev, c, 200, sync=not a.no_sync)
    for rep in range(2):
        r, s = measure(dev, c, a.frames, sync=not a.no_sync)
        print(f"{a.scene} depth {a.depth} {' '.join(a.settings) or 'defaults'}: one call{'' if a.no_sync else ' + one sync'} per frame: frames 0..63 {r:8.1f} Msamples/s, frames 64..{a.frames - 1} {s:8.1f} Msamples/s", flush=True)
    c.close()
