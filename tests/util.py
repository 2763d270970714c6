"""Shared helpers for the parity tests: run the same Scene through the HIP C-ABI and the oracle."""
from __future__ import annotations

import numpy as np

from oracle import binding
from sunvolumerender_amd import abi, host, scenes


import os

ORACLE_THREADS = int(os.environ.get("SVR_CPU_THREADS", "0")) or min(16, os.cpu_count() or 1)


def oracle_frames(scene, nframes, trace_depth=None, window=None, nthreads=None):
    """Progressive frames 0..nframes-1 through the oracle.  Returns (hdr, img, counters).
    Threads: SVR_CPU_THREADS or min(16, cores) -- one GPU's CPU share of the test box; an OpenMP team of all 256 cores of
    that box costs ~0.1 s per call on the small test frames."""
    nthreads = ORACLE_THREADS if nthreads is None else nthreads
    o = binding.OracleScene(scene)
    hdr = o.new_hdr()
    img = np.zeros((o.H, o.W, 4), dtype=np.uint8)
    total = None
    for f in range(nframes):
        c = o.render_pathtracer(hdr, f, trace_depth=trace_depth, window=window, img=img, nthreads=nthreads)
        total = c if total is None else {k: total[k] + c[k] for k in c}
    return hdr, img, total


def hip_frames(dev, scene, nframes, kernel=abi.KERNEL_AUTO, layout=abi.LAYOUT_AUTO, batch=False, count=None,
               shard=None, window=None, empty_skip=True, pipeline=True):
    """Progressive frames 0..nframes-1 through libsvr_hip.so, replaying the Canvas protocol.

    count=None (default) renders TWICE on the same canvas: first the production build (no counters: every early
    return of the walk is live -- the path bench.py times), then the counting build (every iteration runs so that
    the counters equal the oracle's); the two results must agree bit for bit, and the counting run's counters
    are returned.  count=True / False runs one of them."""
    canvas = host.Canvas(dev, scene.width, scene.height)
    try:
        scenes.apply_to_canvas(scene, canvas, layout)
        dev.set_option(abi.OPT_KERNEL, kernel)
        dev.set_option(abi.OPT_EMPTY_SKIP, 1 if empty_skip else 0)
        dev.set_option(abi.OPT_PIPELINE, 1 if pipeline else 0)
        if shard is not None:
            dev.check(dev.lib.svr_set_row_shard(*shard))
        if window is not None:
            dev.check(dev.lib.svr_set_render_window(*window))

        def run(cnt):
            dev.set_option(abi.OPT_COUNT, 1 if cnt else 0)
            dev.reset_counters()
            canvas.ReStartRender()
            if batch:
                canvas.paint_frames(nframes)
            else:
                for _ in range(nframes):
                    canvas.paint()
            dev.synchronize()
            return canvas.read_hdr(), canvas.read_img(), dev.counters()

        if count is None:
            p_hdr, p_img, _ = run(False)
            hdr, img, counters = run(True)
            assert_bit_exact(p_hdr, hdr, "production (non-counting) build vs counting build")
            assert np.array_equal(p_img, img), "production vs counting build: LDR image differs"
        else:
            hdr, img, counters = run(bool(count))
    finally:
        dev.lib.svr_set_row_shard(0, 0, 1)
        dev.lib.svr_set_render_window(0, 0, -1, -1)
        dev.set_option(abi.OPT_KERNEL, abi.KERNEL_AUTO)
        dev.set_option(abi.OPT_COUNT, 0)
        dev.set_option(abi.OPT_EMPTY_SKIP, 1)
        dev.set_option(abi.OPT_PIPELINE, 1)
        canvas.close()
    return hdr, img, counters


def bits(a: np.ndarray) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def assert_bit_exact(a, b, what=""):
    """Bit for bit, except that a NaN equals a NaN whatever its payload: the reference's own arithmetic yields 0/0 for a
    handful of paths in 10^8..10^9 (pathtracer.cu:106-131, core/bsdf/microfacet.h:52-68), and the default NaN of x86 (0xFFC00000)
    and of the GPU (0x7FC00000) differ in the sign bit.  NaN pixels are part of the comparison (both sides must have them in
    the same places) and are named in the failure message."""
    ba, bb = bits(a), bits(b)
    if np.array_equal(ba, bb):
        return
    fa, fb = np.ascontiguousarray(a, dtype=np.float32), np.ascontiguousarray(b, dtype=np.float32)
    na, nb = np.isnan(fa), np.isnan(fb)
    diff = (ba != bb) & ~(na & nb)
    if diff.any():
        n = int(diff.sum())
        idx = np.argwhere(diff)[:5]
        raise AssertionError(f"{what}: {n} of {diff.size} floats differ bitwise; first at {idx.tolist()}; "
                             f"a={fa[tuple(idx[0])]!r} b={fb[tuple(idx[0])]!r}; NaNs: {int(na.sum())} in a, {int(nb.sum())} in b"
                             + (f", first NaN of a at {np.argwhere(na)[0].tolist()}" if na.any() else ""))
