#!/usr/bin/env python3
"""Bias check of the local-majorant mode: per-channel mean(LM) / mean(default) - 1 with its standard error, high sample counts.
usage: tools/lm_bias.py scene[:depth[:mode]] ...   (mode 1 = pool where it applies, 2 = straight-line)"""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from sunvolumerender_amd import abi, host, scenes  # noqa: E402

dev = host.Device(0, fatal_errors=False)
N = 4096
for spec in sys.argv[1:]:
    parts = spec.split(":")
    name, depth, mode = parts[0], int(parts[1]) if len(parts) > 1 else 1, int(parts[2]) if len(parts) > 2 else 1
    sc = scenes.make_scene(name, trace_depth=depth)
    c = host.Canvas(dev, sc.width, sc.height)
    scenes.apply_to_canvas(sc, c)

    def render(lm, n):
        dev.set_option(abi.OPT_LOCAL_MAJORANT, lm)
        c.ReStartRender()
        c.paint_frames(n, sync=True)
        a = c.read_hdr().astype(np.float64)
        c.paint_frames(n, sync=True)
        b = c.read_hdr().astype(np.float64)
        dev.set_option(abi.OPT_LOCAL_MAJORANT, 0)
        return a, b

    A, A2 = render(0, N)
    B = 2 * A2 - A
    F, F2 = render(mode, N)
    bad = ~(np.isfinite(A2).all(axis=2) & np.isfinite(A).all(axis=2) & np.isfinite(F2).all(axis=2))      # (the reference's own 0/0 pixels)
    A, A2, B, F2 = (np.where(bad[..., None], 0.0, v) for v in (A, A2, B, F2))
    npx = A.shape[0] * A.shape[1]
    se = np.sqrt(2.0 * np.mean((A - B) ** 2, axis=(0, 1)) / 4.0 / npx)
    m = A2.mean(axis=(0, 1))
    print(f"{spec:28s} mean {m.round(4)}  LM/default - 1 = {((F2.mean(axis=(0, 1)) - m) / m * 100).round(3)} %  (4 se = {(4 * se / m * 100).round(3)} %)", flush=True)
    c.close()
