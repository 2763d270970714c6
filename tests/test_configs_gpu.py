"""GPU parity at the sizes of BASELINE.json's configs 2, 3 and 4 (c3, c4, c5), through the C ABI.

* c3 (512^3, 1024^2) is the benchmarked configuration: the bench's exact launch shape -- the full frame, ONE
  32-frame launch, the production (non-counting) build, group march on -- is compared with the oracle on every
  pixel, and the cheap GPU-only property (skipping on == off == reference-shaped kernel) is checked at the same shape.
* c4 (512^3, 2048^2): windows against the oracle, full-frame properties, 8-way row-strip composition.
* c5 (1024^3 u16, 1024^2; the only HBM-resident volume, 16-voxel macro-cells): windows against the oracle for the
  path tracer (with the oracle's counters) and the ray caster, skipping on == off.

The oracle runs on SVR_CPU_THREADS host threads (default 16 = one GPU's CPU share of the box).
"""
import ctypes as C
import os

import dataclasses

import numpy as np
import pytest

from oracle import binding
from sunvolumerender_amd import abi, dist, host, scenes
from tests.util import assert_bit_exact, oracle_frames

pytestmark = pytest.mark.gpu
THREADS = int(os.environ.get("SVR_CPU_THREADS", "0")) or min(16, os.cpu_count() or 1)


class Rig:
    """One canvas per configuration (the volume upload dominates the set-up), options restored after each run."""

    def __init__(self, dev, name, **kw):
        self.dev, self.sc = dev, scenes.make_scene(name, **kw)
        self.canvas = host.Canvas(dev, self.sc.width, self.sc.height)
        scenes.apply_to_canvas(self.sc, self.canvas)

    def run(self, frames, batch=True, count=False, kernel=abi.KERNEL_AUTO, skip=1, rayskip=1, shard=None, window=None,
            raycast=False):
        dev, canvas = self.dev, self.canvas
        try:
            dev.set_option(abi.OPT_ENV_ON_ESCAPE, 1 if self.sc.env_on_escape else 0)   # (library-global; conftest resets it after every test)
            dev.set_option(abi.OPT_KERNEL, kernel)
            dev.set_option(abi.OPT_EMPTY_SKIP, skip)
            dev.set_option(abi.OPT_RAY_SKIP, rayskip)
            dev.set_option(abi.OPT_COUNT, 1 if count else 0)
            if shard is not None:
                dev.check(dev.lib.svr_set_row_shard(*shard))
            if window is not None:
                dev.check(dev.lib.svr_set_render_window(*window))
            dev.reset_counters()
            # a window / shard only touches its own pixels: start from a clean accumulator and image
            dev.check(dev.lib.svr_memset_device(canvas.renderParams.hdrBuffer, 0, self.sc.width * self.sc.height * 12))
            dev.check(dev.lib.svr_memset_device(C.c_void_p(canvas.img), 0, self.sc.width * self.sc.height * 4))
            canvas.SetRenderMode(host.Canvas.RENDER_MODE_RAYCASTING if raycast else host.Canvas.RENDER_MODE_PATHTRACER)
            if raycast:
                canvas.paint()
            elif batch:
                canvas.paint_frames(frames)
            else:
                for _ in range(frames):
                    canvas.paint()
            dev.synchronize()
            return canvas.read_hdr(), canvas.read_img(), dev.counters()
        finally:
            dev.lib.svr_set_row_shard(0, 0, 1)
            dev.lib.svr_set_render_window(0, 0, -1, -1)
            dev.set_option(abi.OPT_KERNEL, abi.KERNEL_AUTO)
            dev.set_option(abi.OPT_EMPTY_SKIP, 1)
            dev.set_option(abi.OPT_RAY_SKIP, 1)
            dev.set_option(abi.OPT_COUNT, 0)
            canvas.SetRenderMode(host.Canvas.RENDER_MODE_PATHTRACER)

    def close(self):
        self.canvas.close()


def _oracle_windows(sc, wins, frames, count=False):
    o = binding.OracleScene(sc)
    ref = o.new_hdr()
    img = np.zeros((o.H, o.W, 4), dtype=np.uint8)
    total = {}
    for f in range(frames):
        for w in wins:
            c = o.render_pathtracer(ref, f, window=w, img=img, count=count, nthreads=THREADS)
            for k, v in c.items():
                total[k] = total.get(k, 0) + v
    return ref, img, total


# ------------------------------------------------------------------------------------------------------------
# c3: the benchmark's launch shape
# ------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def c3(hip_dev):
    r = Rig(hip_dev, "c3")
    yield r
    r.close()


def test_c3_bench_shape_full_frame_vs_oracle(c3):
    """bench.py's step: c3, the whole 1024^2 frame, one 32-frame launch of the production build (no counters, every
    early return live, group march on because the wave is full and holds 32 frames of a pixel).  Every pixel of
    the HDR accumulator and of the RGBA8 image against the oracle (33.5 M paths, ~11 s on 16 host threads)."""
    sc = c3.sc
    ref_hdr, ref_img, _ = oracle_frames(sc, 32, nthreads=THREADS)
    hdr, img, _ = c3.run(32, batch=True, count=False)
    assert_bit_exact(hdr, ref_hdr, "c3 full frame, one 32-frame launch, production build")
    assert np.array_equal(img, ref_img)
    # a second group continues the running mean from frame 32 (the accumulator is read back, not cleared)
    c3.canvas.renderParams.frameNo = 32          # (Rig.run's mode reset restarted the render; the accumulator still holds frames 0..31)
    c3.canvas.paint_frames(32, sync=True)
    hdr64 = c3.canvas.read_hdr()
    o = binding.OracleScene(sc)
    win = (448, 496, 576, 528)
    ref64 = ref_hdr.copy()
    for f in range(32, 64):
        o.render_pathtracer(ref64, f, window=win, count=False, nthreads=THREADS)
    x0, y0, x1, y1 = win
    assert_bit_exact(hdr64[y0:y1, x0:x1], ref64[y0:y1, x0:x1], "c3 frames 32..63 continue the running mean")


def test_c3_bench_step_256spp_vs_oracle(c3):
    """bench.py's step EXACTLY: c3, the whole frame, ONE svr_render_pathtracer_frames call of 256 frames = 4 folding launches
    of 64 frames (a wave = one pixel x 64 frames, queue machine on by the AUTO rule), production build.  Four 128x32
    windows (centre of the head, its silhouette, a corner with nothing but the environment, the lower edge) of the HDR
    accumulator and of the RGBA8 image against 256 oracle frames (4.2 M paths)."""
    sc = c3.sc
    hdr, img, _ = c3.run(256, batch=True, count=False)
    wins = [(448, 496, 576, 528), (256, 300, 384, 332), (0, 0, 128, 32), (512, 992, 640, 1024)]
    ref_hdr, ref_img, _ = _oracle_windows(sc, wins, 256)
    for (x0, y0, x1, y1) in wins:
        assert_bit_exact(hdr[y0:y1, x0:x1], ref_hdr[y0:y1, x0:x1], f"c3, one 256-frame call, window {(x0, y0, x1, y1)}")
        assert np.array_equal(img[y0:y1, x0:x1], ref_img[y0:y1, x0:x1]), f"LDR window {(x0, y0, x1, y1)}"
    assert np.isfinite(hdr).all() and hdr[512, 512].max() > 0


def test_c3_bench_shape_properties(c3):
    """At the same shape (full frame, one 32-frame launch): skipping on == empty-space skipping off == whole-ray
    skipping off == the reference-shaped one-thread-per-pixel kernel == the counting build == 32 per-frame calls
    (frame-ahead tracing)."""
    a, ai, _ = c3.run(32)
    for what, kw in (("EMPTY_SKIP=0", dict(skip=0)), ("RAY_SKIP=0", dict(rayskip=0)), ("KERNEL_PIXEL", dict(kernel=abi.KERNEL_PIXEL)),
                     ("counting build", dict(count=True)), ("32 render_pathtracer calls", dict(batch=False))):
        b, bi, _ = c3.run(32, **kw)
        assert_bit_exact(a, b, f"c3 32-frame launch vs {what}")
        assert np.array_equal(ai, bi), what


def test_c3_queue_machine_equals_straight_line_full_frame(c3, hip_dev):
    """The scatter-record queue + per-lane state machine at FULL size (every wave drains many batches of 32 tasks, the
    queues and radiance rows in global memory are reused batch after batch): forced on, it must reproduce the straight-line
    paths bit for bit on the whole 1024^2 frame at traceDepth 1 (merged service) and 4 (separate services, its default
    there), one 64-frame launch each; a window of the depth-4 frame is also checked against the oracle."""
    try:
        for depth in (1, 4):
            c3.canvas.SetScatterTimes(depth)
            hip_dev.set_option(abi.OPT_QUEUE, 0)
            a, ai, _ = c3.run(64)
            hip_dev.set_option(abi.OPT_QUEUE, 2)
            b, bi, _ = c3.run(64)
            assert_bit_exact(a, b, f"c3 depth {depth}: queue machine vs straight-line paths, full frame")
            assert np.array_equal(ai, bi)
        o = binding.OracleScene(c3.sc)
        ref = o.new_hdr()
        w = (480, 500, 544, 516)
        for f in range(64):
            o.render_pathtracer(ref, f, trace_depth=4, window=w, count=False, nthreads=THREADS)
        x0, y0, x1, y1 = w
        assert_bit_exact(b[y0:y1, x0:x1], ref[y0:y1, x0:x1], "c3 depth 4, queue machine vs oracle")
    finally:
        hip_dev.set_option(abi.OPT_QUEUE, 1)
        c3.canvas.SetScatterTimes(1)


def test_c3n_culling_and_queue_equal_plain_walks_full_frame(hip_dev):
    """c3 with noisy non-zero air (the bench's second workload): majorant-bound culling and the queue machine -- with the primary
    walks pooled and the five-iteration trips (SVR_OPT_POOL / SVR_OPT_TRIPS: all on by default there) -- against the same kernel with
    culling and the machine off, whole frame, one 64-frame launch; a window against the oracle."""
    r = Rig(hip_dev, "c3n")
    try:
        a, ai, _ = r.run(64)
        hip_dev.set_option(abi.OPT_QUEUE, 0)
        hip_dev.set_option(abi.OPT_BOUND_CULL, 0)
        b, bi, _ = r.run(64)
        assert_bit_exact(a, b, "c3n: culling + queue machine vs plain walks, full frame")
        assert np.array_equal(ai, bi)
        o = binding.OracleScene(r.sc)
        ref = o.new_hdr()
        w = (480, 500, 544, 508)
        for f in range(64):
            o.render_pathtracer(ref, f, window=w, count=False, nthreads=THREADS)
        x0, y0, x1, y1 = w
        assert_bit_exact(a[y0:y1, x0:x1], ref[y0:y1, x0:x1], "c3n vs oracle")
        # deeper paths through this medium: pooled primary walks, first scatter events queued unshaded (svr_trace_tile.hip): depth 3,
        # 16-frame launch, defaults against plain walks on the whole frame, and a window against the oracle
        hip_dev.set_option(abi.OPT_QUEUE, 1)
        hip_dev.set_option(abi.OPT_BOUND_CULL, 1)
        r.canvas.SetScatterTimes(3)
        sc3 = dataclasses.replace(r.sc, trace_depth=3)
        a3, a3i, _ = r.run(16)
        hip_dev.set_option(abi.OPT_QUEUE, 0)
        hip_dev.set_option(abi.OPT_BOUND_CULL, 0)
        b3, b3i, _ = r.run(16)
        assert_bit_exact(a3, b3, "c3n depth 3: culling + queue machine vs plain walks, full frame")
        assert np.array_equal(a3i, b3i)
        o3 = binding.OracleScene(sc3)
        ref3 = o3.new_hdr()
        w3 = (496, 502, 528, 506)
        for f in range(16):
            o3.render_pathtracer(ref3, f, window=w3, count=False, nthreads=THREADS)
        x0, y0, x1, y1 = w3
        assert_bit_exact(a3[y0:y1, x0:x1], ref3[y0:y1, x0:x1], "c3n depth 3 vs oracle")
    finally:
        hip_dev.set_option(abi.OPT_QUEUE, 1)
        hip_dev.set_option(abi.OPT_BOUND_CULL, 1)
        r.close()


# ------------------------------------------------------------------------------------------------------------
# c4: 512^3 volume, 2048^2 image (BASELINE config 3; tiled across 8 GPUs there)
# ------------------------------------------------------------------------------------------------------------
def test_c3_bone_transfer_function_vs_oracle(hip_dev):
    """c3 under the bone transfer function (c3b): exactly transparent air AND translucent tissue -- skipping, bound classes between 0 and
    1 and the queue machine all at work in one scene.  64-frame launch, full frame: defaults == culling and machine off; a window through
    the skull against the oracle with counters; depth 3, 16 frames: defaults == plain walks and a window against the oracle."""
    r = Rig(hip_dev, "c3b")
    try:
        a, ai, _ = r.run(64)
        assert a.max() > 0
        hip_dev.set_option(abi.OPT_QUEUE, 0)
        hip_dev.set_option(abi.OPT_BOUND_CULL, 0)
        b, bi, _ = r.run(64)
        hip_dev.set_option(abi.OPT_QUEUE, 1)
        hip_dev.set_option(abi.OPT_BOUND_CULL, 1)
        assert_bit_exact(a, b, "c3b: defaults vs plain walks, full frame")
        assert np.array_equal(ai, bi)
        w = (480, 400, 544, 408)
        ref, _, ref_c = _oracle_windows(r.sc, [w], 64, count=True)
        x0, y0, x1, y1 = w
        assert_bit_exact(a[y0:y1, x0:x1], ref[y0:y1, x0:x1], "c3b vs oracle")
        _, _, c = r.run(64, count=True, window=w)
        for k in ("paths", "vol_taps", "woodcock_iters", "scatter_events"):
            assert c[k] == ref_c[k], (k, c[k], ref_c[k])
        r.canvas.SetScatterTimes(3)
        sc3 = dataclasses.replace(r.sc, trace_depth=3)
        a3, _, _ = r.run(16)
        hip_dev.set_option(abi.OPT_QUEUE, 0)
        hip_dev.set_option(abi.OPT_BOUND_CULL, 0)
        b3, _, _ = r.run(16)
        assert_bit_exact(a3, b3, "c3b depth 3: defaults vs plain walks, full frame")
        o3 = binding.OracleScene(sc3)
        ref3 = o3.new_hdr()
        w3 = (496, 402, 528, 406)
        for f in range(16):
            o3.render_pathtracer(ref3, f, window=w3, count=False, nthreads=THREADS)
        x0, y0, x1, y1 = w3
        assert_bit_exact(a3[y0:y1, x0:x1], ref3[y0:y1, x0:x1], "c3b depth 3 vs oracle")
    finally:
        hip_dev.set_option(abi.OPT_QUEUE, 1)
        hip_dev.set_option(abi.OPT_BOUND_CULL, 1)
        r.close()


@pytest.fixture(scope="module")
def c4(hip_dev):
    r = Rig(hip_dev, "c4")
    yield r
    r.close()


C4_WINS = [(960, 1000, 1088, 1032), (192, 1200, 320, 1216), (1800, 80, 1928, 96), (1000, 2040, 1064, 2048)]


def test_c4_windows_vs_oracle(c4):
    """2048^2 windows (centre, limb, background, last rows) of a 16-frame launch and of 2 per-frame calls."""
    ref, ref_img, ref_c = _oracle_windows(c4.sc, C4_WINS, 16, count=True)
    hdr, img, _ = c4.run(16, batch=True)
    for (x0, y0, x1, y1) in C4_WINS:
        assert_bit_exact(hdr[y0:y1, x0:x1], ref[y0:y1, x0:x1], f"c4 window {(x0, y0, x1, y1)}")
        assert np.array_equal(img[y0:y1, x0:x1], ref_img[y0:y1, x0:x1])
    # counters of the counting build over exactly the oracle's windows
    tot = {}
    for w in C4_WINS:
        _, _, c = c4.run(16, batch=True, count=True, window=w)
        for k in ("paths", "vol_taps", "woodcock_iters", "scatter_events", "shadow_walks"):
            tot[k] = tot.get(k, 0) + c[k]
    for k, v in tot.items():
        assert v == ref_c[k], (k, v, ref_c[k])
    ref2, _, _ = _oracle_windows(c4.sc, C4_WINS[:2], 2)
    hdr2, _, _ = c4.run(2, batch=False)
    for (x0, y0, x1, y1) in C4_WINS[:2]:
        assert_bit_exact(hdr2[y0:y1, x0:x1], ref2[y0:y1, x0:x1], f"c4 per-frame calls, window {(x0, y0, x1, y1)}")


def test_c4_full_frame_properties_and_8_way_strips(c4):
    """Full 2048^2 frame: determinism, skipping on == off, and the 8 ranks' interleaved 16-row strips (each rendered
    alone into a zeroed buffer) sum to the single-GPU frame bit for bit -- the decomposition BASELINE config 3 names."""
    a, ai, _ = c4.run(8)
    b, _, _ = c4.run(8)
    assert_bit_exact(a, b, "c4 determinism")
    d, _, _ = c4.run(8, skip=0)
    assert_bit_exact(a, d, "c4 empty-space skipping off")
    assert np.isfinite(a).all() and (a >= 0).all()
    acc = np.zeros_like(a)
    img_acc = np.zeros_like(ai)
    for r in range(8):
        part, pimg, _ = c4.run(8, shard=(16, r, 8))
        rows = dist.owned_rows(c4.sc.height, 16, r, 8)
        other = np.setdiff1d(np.arange(c4.sc.height), rows)
        assert not part[other].any(), f"rank {r} wrote outside its strips"
        acc += part
        img_acc[rows] = pimg[rows]
    assert_bit_exact(acc, a, "c4 8-rank strip sum")
    assert np.array_equal(img_acc, ai)


# ------------------------------------------------------------------------------------------------------------
# c5: 1024^3 u16 volume (2 GiB, HBM-resident), 1024^2 image (BASELINE config 4)
# ------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def c5(hip_dev):
    r = Rig(hip_dev, "c5")
    yield r
    r.close()


C5_WINS = [(480, 500, 544, 516), (96, 600, 160, 608), (700, 300, 764, 308)]


def test_c5_windows_vs_oracle(c5):
    """Path tracer on the 1024^3 volume: windows of an 8-frame launch against the oracle, with the oracle's counters."""
    ref, ref_img, ref_c = _oracle_windows(c5.sc, C5_WINS, 8, count=True)
    hdr, img, _ = c5.run(8, batch=True)
    for (x0, y0, x1, y1) in C5_WINS:
        assert_bit_exact(hdr[y0:y1, x0:x1], ref[y0:y1, x0:x1], f"c5 window {(x0, y0, x1, y1)}")
        assert np.array_equal(img[y0:y1, x0:x1], ref_img[y0:y1, x0:x1])
    tot = {}
    for w in C5_WINS:
        part, _, c = c5.run(8, batch=True, count=True, window=w)
        x0, y0, x1, y1 = w
        assert_bit_exact(part[y0:y1, x0:x1], ref[y0:y1, x0:x1], f"c5 counting build, window {w}")
        for k in ("paths", "vol_taps", "woodcock_iters", "scatter_events", "shadow_walks"):
            tot[k] = tot.get(k, 0) + c[k]
    for k, v in tot.items():
        assert v == ref_c[k], (k, v, ref_c[k])
    assert tot["vol_taps"] / tot["paths"] > 50          # the long walks of a 1024-voxel box


def test_c5_depth2_window_vs_oracle(c5):
    c5.canvas.SetScatterTimes(2)
    try:
        o = binding.OracleScene(c5.sc)
        ref = o.new_hdr()
        w = C5_WINS[0]
        for f in range(4):
            o.render_pathtracer(ref, f, trace_depth=2, window=w, count=False, nthreads=THREADS)
        hdr, _, _ = c5.run(4, batch=True)
        x0, y0, x1, y1 = w
        assert_bit_exact(hdr[y0:y1, x0:x1], ref[y0:y1, x0:x1], "c5 depth 2")
    finally:
        c5.canvas.SetScatterTimes(1)


def test_c5_skip_on_equals_off_and_raycaster(c5):
    a, ai, _ = c5.run(8)
    b, bi, _ = c5.run(8, skip=0)
    assert_bit_exact(a, b, "c5 empty-space skipping off")
    e, _, _ = c5.run(8, rayskip=0)
    assert_bit_exact(a, e, "c5 whole-ray skipping off")
    assert np.array_equal(ai, bi)
    # ray caster: windows against the oracle (image and step count), full frame skipping on == off
    o = binding.OracleScene(c5.sc)
    rc_wins = [(480, 500, 544, 504), (96, 600, 160, 604)]
    _, full, _ = c5.run(1, raycast=True)
    steps = 0
    for w in rc_wins:
        ref, rc = o.render_raycasting(window=w, nthreads=THREADS)
        x0, y0, x1, y1 = w
        assert np.array_equal(full[y0:y1, x0:x1], ref[y0:y1, x0:x1]), f"c5 ray caster window {w}"
        _, _, c = c5.run(1, raycast=True, count=True, window=w)
        assert c["raycast_steps"] == rc["raycast_steps"], w
        steps += rc["raycast_steps"]
    assert steps > 0
    _, noskip, _ = c5.run(1, raycast=True, skip=0)
    assert np.array_equal(full, noskip)
