#!/usr/bin/env python3
import dataclasses, sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from sunvolumerender_amd import abi, host, scenes
from tests.test_env_nee_gpu import _scene, _render
dev = host.Device(0, fatal_errors=False)
for name, depth, off in [("tiny_head_noisy", 3, (0.55, 0.1)), ("tiny_head_noisy", 3, (0.0, 0.0)), ("tiny_head_noisy", 3, (0.0, 0.1)), ("tiny_head", 3, (0.0, 0.1)), ("tiny_head_noisy", 2, (0.55, 0.1))]:
    sc = _scene(name, depth, env_offset=off)
    c = host.Canvas(dev, sc.width, sc.height)
    scenes.apply_to_canvas(sc, c)
    N = 4096
    A, A2 = _render(dev, c, False, (N, N))
    B = 2 * A2 - A
    (E,) = _render(dev, c, True, (2 * N,))
    (E2,) = _render(dev, c, True, (N,))
    H, W = A.shape[:2]
    print(f"== {name} depth {depth} offset {off}: NaN px default {int((~np.isfinite(A2)).any(axis=2).sum())} mode {int((~np.isfinite(E)).any(axis=2).sum())}")
    bad = (~np.isfinite(A2)).any(axis=2) | (~np.isfinite(E)).any(axis=2) | (~np.isfinite(A)).any(axis=2)
    A, A2, B, E = [np.where(bad[..., None], 0, x) for x in (A, A2, B, E)]
    for (y0, y1, x0, x1) in [(0, H, 0, W), (0, H // 2, 0, W // 2), (0, H // 2, W // 2, W), (H // 2, H, 0, W // 2), (H // 2, H, W // 2, W)]:
        a2, d, e = A2[y0:y1, x0:x1], (A - B)[y0:y1, x0:x1], E[y0:y1, x0:x1]
        npx = a2.shape[0] * a2.shape[1]
        se = np.sqrt(2.0 * np.mean(d ** 2, axis=(0, 1)) / 4.0 / npx)
        dm = e.mean(axis=(0, 1)) - a2.mean(axis=(0, 1))
        print(f"   region {(y0, y1, x0, x1)}: mean default {a2.mean(axis=(0,1))}, mode - default {dm}, in se {dm / se}")
    print(f"   max pixel: default {A2.max():.1f}, mode {E.max():.1f}; rmse(mode, default 2N) {np.sqrt(np.mean((E - A2) ** 2)):.3f}, rmse(A, B) {np.sqrt(np.mean((A - B) ** 2)):.3f}")
    c.close()
