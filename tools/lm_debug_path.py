#!/usr/bin/env python3
"""Experiment build with -DSVR_LM_DEBUG...: render the one debugged frame of one pixel in the local-majorant mode (prints the walk).
usage: SVR_HIP_LIB=.../libsvr_dbg.so tools/lm_debug_path.py scene x y frame"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from sunvolumerender_amd import abi, host, scenes  # noqa: E402

name, x, y, f = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
dev = host.Device(0, fatal_errors=False)
sc = scenes.make_scene(name, trace_depth=1)
c = host.Canvas(dev, sc.width, sc.height)
scenes.apply_to_canvas(sc, c)
dev.set_option(abi.OPT_LOCAL_MAJORANT, 1)
dev.check(dev.lib.svr_set_render_window(x, y, x + 1, y + 1))
c.renderParams.frameNo = f
dev.check(dev.lib.svr_memset_device(c.renderParams.hdrBuffer, 0, sc.width * sc.height * 12))
c.paint_frames(1, sync=True)
print("result", c.read_hdr()[y, x].tolist())
c.close()
