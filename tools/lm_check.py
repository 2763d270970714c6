#!/usr/bin/env python3
"""Sanity of a local-majorant render: non-finite / negative pixels, and which frames produce them.
usage: tools/lm_check.py scene [frames]"""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from sunvolumerender_amd import abi, host, scenes  # noqa: E402

name, n = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 64
dev = host.Device(0, fatal_errors=False)
sc = scenes.make_scene(name, trace_depth=1)
c = host.Canvas(dev, sc.width, sc.height)
scenes.apply_to_canvas(sc, c)


def bad_pixels(a):
    bad = ~np.isfinite(a).all(axis=2) | (a < 0).any(axis=2)
    ys, xs = np.nonzero(bad)
    return list(zip(ys.tolist(), xs.tolist())), a[bad]


found = None
for lm, sub in ((1, 1), (1, 0), (2, 1), (0, 1)):
    dev.set_option(abi.OPT_LOCAL_MAJORANT, lm)
    dev.set_option(abi.OPT_LM_SUBCELLS, sub)
    c.ReStartRender()
    c.paint_frames(n, sync=True)
    a = c.read_hdr()
    px, vals = bad_pixels(a)
    print(f"lm={lm} sub={sub}: bad pixels {len(px)} {px[:6]} values {vals[:3].tolist()}", flush=True)
    if px and found is None:
        found = (lm, sub, px[0])
if found:
    lm, sub, (y, x) = found
    dev.set_option(abi.OPT_LOCAL_MAJORANT, lm)
    dev.set_option(abi.OPT_LM_SUBCELLS, sub)
    dev.check(dev.lib.svr_set_render_window(x, y, x + 1, y + 1))
    for f0 in range(0, n, 64):
        for f in range(f0, min(n, f0 + 64)):
            c.renderParams.frameNo = f
            dev.check(dev.lib.svr_memset_device(c.renderParams.hdrBuffer, 0, sc.width * sc.height * 12))
            c.paint_frames(1, sync=True)                      # one frame, straight-line kernel (non-folding launch)
            v = c.read_hdr()[y, x]
            if not np.isfinite(v).all() or (v < 0).any():
                print(f"pixel ({y},{x}) frame {f}: {v.tolist()} [single-frame launch]", flush=True)
    dev.lib.svr_set_render_window(0, 0, -1, -1)
dev.set_option(abi.OPT_LOCAL_MAJORANT, 0)
c.close()
