"""More GPU parity through the C ABI: golden fixtures (no oracle involved), ray caster, row-strip sharding,
scene edits through the Canvas protocol, error behaviour, and size-independent properties at the full
benchmark size (c3: 512^3 volume, 1024^2 image)."""
import ctypes as C
import dataclasses
from pathlib import Path

import numpy as np
import pytest

from oracle import binding
from sunvolumerender_amd import abi, dist, host, scenes
from tests.util import assert_bit_exact, hip_frames, oracle_frames

pytestmark = pytest.mark.gpu
GOLD = Path(__file__).resolve().parent / "golden"


@pytest.mark.parametrize("name,depth,frames", [("tiny", 1, 3), ("tiny_head", 4, 2), ("tiny_bone", 6, 1)])
def test_golden_fixtures_bit_exact(hip_dev, name, depth, frames):
    g = np.load(GOLD / f"render_{name}_d{depth}_f{frames}.npz")
    sc = scenes.make_scene(name, trace_depth=depth)
    hdr, img, c = hip_frames(hip_dev, sc, frames)
    assert_bit_exact(hdr, g["hdr"], f"{name} vs golden")
    assert np.array_equal(img, g["img"])
    gold = dict(zip(g["counter_names"].tolist(), g["counters"].tolist()))
    assert c["vol_taps"] == gold["vol_taps"] and c["woodcock_iters"] == gold["woodcock_iters"]


@pytest.mark.parametrize("layout", [abi.LAYOUT_LINEAR, abi.LAYOUT_BRICK], ids=["linear", "brick"])
def test_raycasting_bit_exact(hip_dev, layout):
    """render_raycasting (raycasting.cu:15-75) against the oracle and the golden image; RGBA8 bit-exact."""
    sc = scenes.make_scene("tiny_head")
    ref, rc = binding.OracleScene(sc).render_raycasting()
    canvas = host.Canvas(hip_dev, sc.width, sc.height)
    try:
        scenes.apply_to_canvas(sc, canvas, layout)
        canvas.SetRenderMode(host.Canvas.RENDER_MODE_RAYCASTING)
        hip_dev.set_option(abi.OPT_COUNT, 1)
        hip_dev.reset_counters()
        canvas.paint(sync=True)
        img = canvas.read_img()
        cnt = hip_dev.counters()
    finally:
        hip_dev.set_option(abi.OPT_COUNT, 0)
        canvas.close()
    assert np.array_equal(img, ref)
    assert np.array_equal(img, np.load(GOLD / "raycast_tiny_head.npz")["img"])
    assert cnt["raycast_steps"] == rc["raycast_steps"]


@pytest.mark.parametrize("world,strip", [(2, 8), (3, 16), (8, 8)])
def test_row_strip_shards_compose(hip_dev, world, strip):
    """Every rank's strips, rendered separately into zeroed buffers and summed, equal the single-GPU frame."""
    sc = scenes.make_scene("tiny_head", trace_depth=2)
    full, full_img, _ = hip_frames(hip_dev, sc, 2)
    acc = np.zeros_like(full)
    img_acc = np.zeros_like(full_img)
    for r in range(world):
        part, pimg, c = hip_frames(hip_dev, sc, 2, shard=(strip, r, world))
        rows = dist.owned_rows(sc.height, strip, r, world)
        other = np.setdiff1d(np.arange(sc.height), rows)
        assert not part[other].any(), "a rank wrote outside its strips"
        assert c["paths"] == 2 * len(rows) * sc.width
        acc += part
        img_acc[rows] = pimg[rows]
    assert_bit_exact(acc, full, f"{world}-rank strip sum")
    assert np.array_equal(img_acc, full_img)


def test_render_window(hip_dev):
    sc = scenes.make_scene("tiny_head")
    full, _, _ = hip_frames(hip_dev, sc, 1)
    part, _, c = hip_frames(hip_dev, sc, 1, window=(10, 20, 50, 61))
    assert_bit_exact(part[20:61, 10:50], full[20:61, 10:50], "window")
    mask = np.ones(full.shape[:2], bool)
    mask[20:61, 10:50] = False
    assert not part[mask].any() and c["paths"] == 40 * 41


def test_frame_zero_clears_the_whole_accumulator_under_a_window_or_shard(hip_dev):
    """clear_hdr_buffer (pathtracer.cu:86-94) zeroes the WHOLE accumulator at frame 0; under a window or a row shard the kernels
    touch only their own pixels, so the library clears the rest: a caller-supplied buffer with stale values must come out zero
    outside the owned pixels (the ranks' strips are summed into one frame afterwards)."""
    sc = scenes.make_scene("tiny_head")
    for kw in (dict(window=(10, 20, 50, 61)), dict(shard=(8, 1, 3))):
        canvas = host.Canvas(hip_dev, sc.width, sc.height)
        try:
            scenes.apply_to_canvas(sc, canvas)
            hip_dev.to_device(int(canvas.renderParams.hdrBuffer), np.full((sc.height, sc.width, 3), 7.5, dtype=np.float32))
            if "window" in kw:
                hip_dev.check(hip_dev.lib.svr_set_render_window(*kw["window"]))
                owned = np.zeros((sc.height, sc.width), bool)
                x0, y0, x1, y1 = kw["window"]
                owned[y0:y1, x0:x1] = True
            else:
                hip_dev.check(hip_dev.lib.svr_set_row_shard(*kw["shard"]))
                owned = np.zeros((sc.height, sc.width), bool)
                owned[dist.owned_rows(sc.height, *kw["shard"])] = True
            for batch in (False, True):
                canvas.ReStartRender()
                canvas.paint_frames(16) if batch else canvas.paint()
                hip_dev.synchronize()
                hdr = canvas.read_hdr()
                assert not hdr[~owned].any(), f"stale values outside the owned pixels ({kw}, batch={batch})"
                assert hdr[owned].any()
                hip_dev.to_device(int(canvas.renderParams.hdrBuffer), np.full((sc.height, sc.width, 3), 7.5, dtype=np.float32))
        finally:
            hip_dev.lib.svr_set_render_window(0, 0, -1, -1)
            hip_dev.lib.svr_set_row_shard(0, 0, 1)
            canvas.close()


def test_canvas_edits_restart_and_match_oracle(hip_dev):
    """Setter -> setup_* -> frameNo = 0 (gui/canvas.h:43-175): density scale, clip planes, lights, exposure."""
    base = scenes.make_scene("tiny_head", trace_depth=2)
    canvas = host.Canvas(hip_dev, base.width, base.height)
    try:
        scenes.apply_to_canvas(base, canvas)
        canvas.paint()
        canvas.paint()
        assert canvas.renderParams.frameNo == 2
        canvas.SetDensityScale(0.7)
        assert canvas.renderParams.frameNo == 0
        canvas.SetClipPlane((-0.5, 1.0), (-1.0, 0.8), (-1.0, 1.0))
        canvas.SetExposure(0.5)
        canvas.paint()
        canvas.paint(sync=True)
        hdr, img = canvas.read_hdr(), canvas.read_img()
    finally:
        canvas.close()
    edited = scenes.make_scene("tiny_head", trace_depth=2, density_scale=0.7, exposure=0.5,
                               clip=((-0.5, 1.0), (-1.0, 0.8), (-1.0, 1.0)))
    ref_hdr, ref_img, _ = oracle_frames(edited, 2)
    assert_bit_exact(hdr, ref_hdr, "after edits")
    assert np.array_equal(img, ref_img)


def test_transfer_function_edit_rebuilds_skip_mask(hip_dev):
    """A TF that makes 'air' slightly opaque must invalidate the empty-space mask (svr_update_tf_texture)."""
    sc = scenes.make_scene("tiny_head", trace_depth=1)
    canvas = host.Canvas(hip_dev, sc.width, sc.height)
    try:
        scenes.apply_to_canvas(sc, canvas)
        canvas.paint(sync=True)
        t2 = sc.tf_rgba.copy()
        t2[:, 3] = np.maximum(t2[:, 3], np.float32(0.02))          # haze everywhere: nothing is skippable
        hip_dev.check(hip_dev.lib.svr_update_tf_texture(canvas.transferFunction.tex, t2.ctypes.data_as(C.c_void_p), t2.shape[0], 0))
        canvas.ReStartRender()
        canvas.paint(sync=True)
        hdr = canvas.read_hdr()
    finally:
        canvas.close()
    hazy = scenes.make_scene("tiny_head", trace_depth=1)
    hazy.tf_rgba = t2
    ref, _, _ = oracle_frames(hazy, 1)
    assert_bit_exact(hdr, ref, "after TF edit")


def test_errors_are_reported_not_swallowed(hip_dev):
    lib = hip_dev.lib
    rp = abi.RenderParams(1, 0, None)
    lib.render_pathtracer(None, C.byref(rp))
    assert lib.svr_last_error_code() != 0
    lib.svr_clear_error()
    vol = abi.cudaVolume()
    vol.tex = 0xDEAD
    lib.setup_volume(C.byref(vol))
    tf = abi.cudaTransferFunction()
    tf.tex, tf.maxOpacity = 0xBEEF, 0.5
    cam = host.camera_setup((0, 0, 5), (0, 0, 0), (0, 1, 0), imageW=16, imageH=16)
    img = hip_dev.malloc(16 * 16 * 4)
    lib.render_raycasting(C.c_void_p(img), C.byref(vol), C.byref(tf), C.byref(cam), C.c_float(1.0))
    assert lib.svr_last_error_code() != 0 and b"handle" in lib.svr_last_error()
    lib.svr_clear_error()
    hip_dev.free(img)
    assert lib.svr_create_tf_texture(None, 16, 0) == 0 and lib.svr_last_error_code() != 0
    lib.svr_clear_error()


# ---------------- full benchmark size ----------------
@pytest.fixture(scope="module")
def c3_canvas(hip_dev):
    sc = scenes.make_scene("c3")
    canvas = host.Canvas(hip_dev, sc.width, sc.height)
    scenes.apply_to_canvas(sc, canvas)
    yield sc, canvas
    canvas.close()


def test_c3_window_matches_oracle(hip_dev, c3_canvas):
    """Bit-exact against the oracle at the benchmark's full sizes on windows the oracle finishes in seconds."""
    sc, canvas = c3_canvas
    canvas.ReStartRender()
    canvas.paint()
    canvas.paint(sync=True)
    hdr = canvas.read_hdr()
    o = binding.OracleScene(sc)
    ref = o.new_hdr()
    wins = [(480, 500, 544, 532), (96, 600, 160, 616), (900, 40, 964, 56)]
    for f in range(2):
        for w in wins:
            o.render_pathtracer(ref, f, window=w)
    for (x0, y0, x1, y1) in wins:
        assert_bit_exact(hdr[y0:y1, x0:x1], ref[y0:y1, x0:x1], f"c3 window {(x0, y0, x1, y1)}")


def test_c3_properties(hip_dev, c3_canvas):
    """Size-independent properties at 512^3 / 1024^2: determinism, batch == sequential, kernels agree,
    skipping changes nothing, running mean stays inside the per-frame extremes."""
    sc, canvas = c3_canvas

    def run(frames, batch=False, kernel=abi.KERNEL_AUTO, skip=1, rayskip=1):
        hip_dev.set_option(abi.OPT_KERNEL, kernel)
        hip_dev.set_option(abi.OPT_EMPTY_SKIP, skip)
        hip_dev.set_option(abi.OPT_RAY_SKIP, rayskip)
        canvas.ReStartRender()
        if batch:
            canvas.paint_frames(frames)
        else:
            for _ in range(frames):
                canvas.paint()
        hip_dev.synchronize()
        out = canvas.read_hdr(), canvas.read_img()
        hip_dev.set_option(abi.OPT_KERNEL, abi.KERNEL_AUTO)
        hip_dev.set_option(abi.OPT_EMPTY_SKIP, 1)
        hip_dev.set_option(abi.OPT_RAY_SKIP, 1)
        return out

    a, ai = run(3)
    b, bi = run(3)
    assert_bit_exact(a, b, "determinism")
    c, ci = run(3, batch=True)
    assert_bit_exact(a, c, "batch == sequential")
    d, _ = run(3, skip=0)
    assert_bit_exact(a, d, "empty-space skipping off")
    e, _ = run(3, rayskip=0)
    assert_bit_exact(a, e, "whole-ray skipping off")
    f, _ = run(3, kernel=abi.KERNEL_PIXEL)
    assert_bit_exact(a, f, "baseline kernel")
    assert np.array_equal(ai, bi) and np.array_equal(ai, ci)
    assert np.isfinite(a).all() and (a >= 0).all()
    one, one_img = run(1)
    assert (a.max(axis=(0, 1)) <= 3 * np.maximum(one.max(axis=(0, 1)), 1e-6) * 1e6).all()
    # tone map of the accumulated buffer (hdr_to_ldr alone) reproduces img
    img2 = hip_dev.malloc(sc.width * sc.height * 4)
    hip_dev.check(hip_dev.lib.svr_hdr_to_ldr(C.c_void_p(img2), C.byref(canvas.renderParams)))
    got = hip_dev.to_host(img2, (sc.height, sc.width, 4), np.uint8)
    hip_dev.free(img2)
    assert np.array_equal(got, one_img)


def _odd_scene(depth=2):
    """Non-cubic volume, anisotropic spacing, image size not a multiple of the tile size, off-axis camera,
    thin-lens aperture, density scale and clip planes all at once."""
    rs = np.random.RandomState(5)
    base = scenes.make_ct_head_volume(64)
    vox = np.ascontiguousarray(base[8:40, 4:60, 12:52])           # nz=32, ny=56, nx=40
    spacing = (1.0, 0.5, 2.0)
    tf, mo = scenes.bone_transfer_function()
    nx, ny, nz = vox.shape[2], vox.shape[1], vox.shape[0]
    W, H = 50, 37
    cam = host.camera_setup((30.0, 22.0, 95.0), (1.0, -2.0, 0.5), (0.1, 1.0, 0.0), 40.0, 0.8, 1.3, 1.5, W, H)
    R = host.bounding_sphere_radius((nx, ny, nz), spacing)
    lights = [host.place_area_light(20.0, 30.0, R * 1.5 + 1, 6.0, (1.0, 0.9, 0.8), 700.0),
              host.place_area_light(-70.0, -40.0, R * 1.5 + 1, 9.0, (0.7, 0.8, 1.0), 900.0)]
    return scenes.Scene(name="odd", vox=vox, spacing=spacing, max_magnitude=scenes.max_gradient_magnitude(vox, spacing),
                        tf_rgba=tf, max_opacity=mo, width=W, height=H, lights=lights, env_map=scenes.synthetic_env_map(64, 32),
                        env_offset=(0.3, 0.1), env_on_escape=True, trace_depth=depth, density_scale=1.2, gradient_factor=0.8,
                        clip=((-0.9, 1.0), (-1.0, 0.85), (-0.7, 1.0)), camera=cam)


@pytest.mark.parametrize("kernel", [abi.KERNEL_TILE, abi.KERNEL_PIXEL, abi.KERNEL_WAVEFRONT, abi.KERNEL_ULOOP],
                         ids=["tile", "pixel", "wavefront", "uloop"])
def test_non_cubic_anisotropic_scene(hip_dev, kernel):
    sc = _odd_scene()
    ref_hdr, ref_img, ref_c = oracle_frames(sc, 3)
    for layout in (abi.LAYOUT_BRICK, abi.LAYOUT_LINEAR, abi.LAYOUT_CELL):
        hdr, img, c = hip_frames(hip_dev, sc, 3, kernel=kernel, layout=layout)
        assert_bit_exact(hdr, ref_hdr, f"odd scene kernel {kernel} layout {layout}")
        assert np.array_equal(img, ref_img)
        assert c["vol_taps"] == ref_c["vol_taps"] and c["woodcock_iters"] == ref_c["woodcock_iters"]
    b_hdr, b_img, _ = hip_frames(hip_dev, sc, 3, kernel=kernel, batch=True)
    assert_bit_exact(b_hdr, ref_hdr, "odd scene, batch")
    # ray caster on the same scene
    ref_rc, _ = binding.OracleScene(sc).render_raycasting()
    canvas = host.Canvas(hip_dev, sc.width, sc.height)
    try:
        scenes.apply_to_canvas(sc, canvas)
        canvas.SetRenderMode(host.Canvas.RENDER_MODE_RAYCASTING)
        canvas.paint(sync=True)
        assert np.array_equal(canvas.read_img(), ref_rc)
    finally:
        canvas.close()


def _inside_scene():
    """Camera inside the volume (tNear = 0: the first sample sits exactly at the eye, where the head-light
    direction of raycasting.cu:45 is 0/0)."""
    sc = scenes.make_scene("tiny_head")
    cam = host.camera_setup((3.0, -2.0, 5.0), (0.0, 1.0, -4.0), (0.0, 1.0, 0.0), 60.0, 0.0, 1.0, 1.0, sc.width, sc.height)
    return dataclasses.replace(sc, name="inside", camera=cam)


@pytest.mark.parametrize("skip", [1, 0], ids=["skip", "noskip"])
@pytest.mark.parametrize("layout", [abi.LAYOUT_LINEAR, abi.LAYOUT_BRICK], ids=["linear", "brick"])
@pytest.mark.parametrize("case", ["tiny_bone", "odd", "inside"])
def test_raycasting_cases(hip_dev, case, layout, skip):
    """Empty-space skipping in k_raycast drops samples whose opacity is exactly 0; the image and the step
    count must not change, with the bitmask on or off."""
    sc = {"tiny_bone": lambda: scenes.make_scene("tiny_bone"), "odd": _odd_scene, "inside": _inside_scene}[case]()
    ref, rc = binding.OracleScene(sc).render_raycasting()
    canvas = host.Canvas(hip_dev, sc.width, sc.height)
    try:
        scenes.apply_to_canvas(sc, canvas, layout)
        canvas.SetRenderMode(host.Canvas.RENDER_MODE_RAYCASTING)
        hip_dev.set_option(abi.OPT_EMPTY_SKIP, skip)
        hip_dev.set_option(abi.OPT_COUNT, 1)
        hip_dev.reset_counters()
        canvas.paint(sync=True)
        img = canvas.read_img()
        cnt = hip_dev.counters()
    finally:
        hip_dev.set_option(abi.OPT_COUNT, 0)
        hip_dev.set_option(abi.OPT_EMPTY_SKIP, 1)
        canvas.close()
    assert np.array_equal(img, ref)
    assert cnt["raycast_steps"] == rc["raycast_steps"]
    assert cnt["vol_taps"] == 7 * rc["raycast_steps"]
    if skip:
        assert cnt["vol_taps_executed"] < cnt["vol_taps"]
    else:
        assert cnt["vol_taps_executed"] >= cnt["vol_taps"]      # plus samples evaluated past the early exit


@pytest.mark.parametrize("lanes_log2", [0, 1, 2, 4, 5])
def test_raycasting_lanes_per_ray(hip_dev, lanes_log2):
    """Any number of lanes per ray (samples of a chunk evaluated in parallel, composited in order) gives the
    reference's image and step count; 3 (the default) is covered by the tests above."""
    for sc in (_odd_scene(), scenes.make_scene("tiny_head")):
        ref, rc = binding.OracleScene(sc).render_raycasting()
        canvas = host.Canvas(hip_dev, sc.width, sc.height)
        try:
            scenes.apply_to_canvas(sc, canvas)
            canvas.SetRenderMode(host.Canvas.RENDER_MODE_RAYCASTING)
            hip_dev.set_option(abi.OPT_RAYCAST_LANES_LOG2, lanes_log2)
            hip_dev.set_option(abi.OPT_COUNT, 1)
            hip_dev.reset_counters()
            canvas.paint(sync=True)
            img = canvas.read_img()
            cnt = hip_dev.counters()
        finally:
            hip_dev.set_option(abi.OPT_COUNT, 0)
            hip_dev.set_option(abi.OPT_RAYCAST_LANES_LOG2, 3)
            canvas.close()
        assert np.array_equal(img, ref)
        assert cnt["raycast_steps"] == rc["raycast_steps"]


@pytest.mark.parametrize("name", ["tiny_head", "tiny_head_noisy", "small_head", "odd_thin_lens"])
def test_pooled_primary_walks_bit_exact(hip_dev, name):
    """SVR_OPT_POOL: at traceDepth 1 the queue machine can walk the PRIMARY rays too (a task only generates its camera rays as P
    records; the lane machine walks them, the collisions are shaded 64 at a time into C1 records, the machine walks those) -- the
    default for media without exactly transparent space.  Forced on (2) and off (0): the oracle's image and counters, production and
    counting builds, one many-frame call (the pool rides on folding launches) of 40 frames (32 frame lanes, 8 dead) and of 64."""
    sc = _odd_scene(depth=1) if name == "odd_thin_lens" else scenes.make_scene(name, trace_depth=1)
    for nframes in (40, 64):
        ref_hdr, ref_img, ref_c = oracle_frames(sc, nframes)
        for pool in (2, 0):
            hip_dev.set_option(abi.OPT_POOL, pool)
            hip_dev.set_option(abi.OPT_QUEUE, 2)
            hdr, img, c = hip_frames(hip_dev, sc, nframes, batch=True)
            hip_dev.set_option(abi.OPT_QUEUE, 1)
            assert_bit_exact(hdr, ref_hdr, f"{name}: pooled primary walks = {pool}, {nframes} frames")
            assert np.array_equal(img, ref_img)
            assert c["vol_taps"] == ref_c["vol_taps"] and c["woodcock_iters"] == ref_c["woodcock_iters"] and c["scatter_events"] == ref_c["scatter_events"]
    hip_dev.set_option(abi.OPT_POOL, 1)


@pytest.mark.parametrize("name,depth", [("tiny_head", 1), ("tiny_head_noisy", 1), ("small_head_noisy", 1), ("tiny_head", 3), ("tiny_head_noisy", 2), ("small_head", 2), ("odd", 1), ("odd", 3)])
def test_five_iteration_trips_bit_exact(hip_dev, name, depth):
    """SVR_OPT_TRIPS: the walking lanes of the lane machine run five Woodcock iterations per turn with the generator as a circular
    buffer (compile-time heads, one barrel rotation behind the trip); lanes that need a fetch, a re-march or are through wait for
    the end of the trip.  Every path executes the operations of the plain machine in the same order: forced on (2; with the primary
    walks pooled at depth 1) and off (0), the oracle's image and counters, production and counting builds, 40- and 64-frame calls."""
    sc = _odd_scene(depth=depth) if name == "odd" else scenes.make_scene(name, trace_depth=depth)
    try:
        for nframes in (40, 64):
            ref_hdr, ref_img, ref_c = oracle_frames(sc, nframes)
            for trips in (2, 0):
                hip_dev.set_option(abi.OPT_TRIPS, trips)
                hip_dev.set_option(abi.OPT_POOL, 2)
                hip_dev.set_option(abi.OPT_QUEUE, 2)
                hdr, img, c = hip_frames(hip_dev, sc, nframes, batch=True)
                assert_bit_exact(hdr, ref_hdr, f"{name} depth {depth}: trips = {trips}, {nframes} frames")
                assert np.array_equal(img, ref_img)
                assert c["vol_taps"] == ref_c["vol_taps"] and c["woodcock_iters"] == ref_c["woodcock_iters"] and c["scatter_events"] == ref_c["scatter_events"]
    finally:
        hip_dev.set_option(abi.OPT_TRIPS, 1)
        hip_dev.set_option(abi.OPT_POOL, 1)
        hip_dev.set_option(abi.OPT_QUEUE, 1)


def _odd_noisy_scene(depth=1):
    """tiny_head_noisy's medium (nothing exactly transparent) as a non-cubic crop with anisotropic spacing, an off-axis thin-lens camera, a
    density scale and clip planes: every axis of the fast bound look-up gets its own scale, offset and clamp."""
    base = scenes.make_scene("tiny_head_noisy", trace_depth=depth)
    vox = np.ascontiguousarray(scenes.make_ct_head_volume(64, noisy_air=True)[6:47, 3:62, 10:55])           # nz=41, ny=59, nx=45
    spacing = (1.0, 0.5, 2.0)
    W, H = 50, 37
    cam = host.camera_setup((30.0, 22.0, 95.0), (1.0, -2.0, 0.5), (0.1, 1.0, 0.0), 40.0, 0.8, 1.3, 1.5, W, H)
    return dataclasses.replace(base, name="odd_noisy", vox=vox, spacing=spacing, max_magnitude=scenes.max_gradient_magnitude(vox, spacing), width=W, height=H,
                               camera=cam, density_scale=1.2, clip=((-0.9, 1.0), (-1.0, 0.85), (-0.7, 1.0)))


@pytest.mark.parametrize("name,depth", [("tiny_head_noisy", 1), ("tiny_head_noisy", 3), ("small_head_noisy", 1), ("odd_noisy", 1), ("odd_noisy", 2)])
def test_fast_bound_lookup_bit_exact(hip_dev, name, depth):
    """SVR_OPT_FAST_BOUND (default on; media without exactly transparent space): the trips of the pooled lane machine take an iteration's fetch
    bound from one fma per axis on the ray parameter and a byte table whose cells cover one more voxel per side (svr_accel.hip, k_bound8), and
    compare it with the top 8 bits of the accept draw's random word.  Any valid bound culls correctly: the oracle's accumulator, image and
    algorithmic counters with the switch on and off -- while the EXECUTED fetches differ (the table is not the class table: it ran)."""
    sc = _odd_noisy_scene(depth) if name == "odd_noisy" else scenes.make_scene(name, trace_depth=depth)
    executed = {}
    try:
        for nframes in (40, 64):
            ref_hdr, ref_img, ref_c = oracle_frames(sc, nframes)
            for fast in (1, 0):
                hip_dev.set_option(abi.OPT_FAST_BOUND, fast)
                assert hip_dev.get_option(abi.OPT_FAST_BOUND) == fast
                hdr, img, c = hip_frames(hip_dev, sc, nframes, batch=True)
                assert_bit_exact(hdr, ref_hdr, f"{name} depth {depth}: fast bound look-up = {fast}, {nframes} frames")
                assert np.array_equal(img, ref_img)
                assert c["vol_taps"] == ref_c["vol_taps"] and c["woodcock_iters"] == ref_c["woodcock_iters"] and c["scatter_events"] == ref_c["scatter_events"]
                executed[(nframes, fast)] = c["vol_taps_executed"]
            assert executed[(nframes, 1)] != executed[(nframes, 0)], "the fast bound look-up fetched exactly what the class table does: did it run?"
            assert executed[(nframes, 1)] < ref_c["vol_taps"]
    finally:
        hip_dev.set_option(abi.OPT_FAST_BOUND, 1)


def _bound8_rays(sc, rs, n):
    """Rays for svr_selftest_bound8: camera rays through random pixels, rays between random points of the volume's box (incl. points ON its faces, edges
    and corners, and axis-parallel directions), each with u in [0, 1] -- a quarter of them exactly 0 or 1 (the box entry and exit)."""
    size = np.asarray(host.volume_size(sc.dim, sc.spacing), dtype=np.float64)
    cam = sc.resolved_camera()
    eye = np.array([cam.pos.x, cam.pos.y, cam.pos.z], dtype=np.float64)
    k = n // 4
    # (a) from the eye towards random points of a slightly larger box (some miss)
    tgt = (rs.rand(k, 3) - 0.5) * size * 1.1
    a = np.concatenate([np.broadcast_to(eye, (k, 3)), tgt - eye], axis=1)
    # (b) between two random interior points
    p, q = (rs.rand(k, 3) - 0.5) * size, (rs.rand(k, 3) - 0.5) * size
    b = np.concatenate([p, q - p], axis=1)
    # (c) from points with some coordinates snapped onto the faces
    p = (rs.rand(k, 3) - 0.5) * size
    snap = rs.rand(k, 3) < 0.4
    p = np.where(snap, np.sign(p) * 0.5 * size, p)
    q = (rs.rand(k, 3) - 0.5) * size
    c = np.concatenate([p, q - p], axis=1)
    # (d) axis-parallel and plane-parallel directions from interior points and from the eye
    p = (rs.rand(n - 3 * k, 3) - 0.5) * size
    d = rs.randn(n - 3 * k, 3)
    d[rs.rand(n - 3 * k, 3) < 0.5] = 0.0
    d[np.all(d == 0.0, axis=1)] = (0.0, 0.0, 1.0)
    dd = np.concatenate([p, d], axis=1)
    rays = np.concatenate([a, b, c, dd], axis=0)
    nrm = np.linalg.norm(rays[:, 3:6], axis=1, keepdims=True)
    rays[:, 3:6] /= np.where(nrm > 0, nrm, 1.0)
    u = rs.rand(n, 1)
    ends = rs.rand(n, 1)
    u = np.where(ends < 0.125, 0.0, np.where(ends < 0.25, 1.0, u))
    return np.ascontiguousarray(np.concatenate([rays, u], axis=1), dtype=np.float32)


@pytest.mark.parametrize("name", ["tiny_head_noisy", "small_head_noisy", "odd_noisy", "odd_noisy_bone_fog"])
def test_fast_bound_table_bounds_every_fetch(hip_dev, name):
    """The property the bit-exactness of SVR_OPT_FAST_BOUND rests on, tested directly (svr_selftest_bound8, csrc/svr_selftest.hip): at 2 million points of
    rays through the volume -- camera rays, rays between interior points, from points on the faces, edges and corners, axis-parallel ones, a quarter of the
    points exactly at the box entry or exit -- the byte the lane machine would read (the same expressions for the ray in table coordinates and the index)
    culls no draw that the reference's accept test at that point (exact trilinear cell, fetch, alpha, invSigmaMax) could accept, and the index stays
    inside the table."""
    if name == "odd_noisy_bone_fog":
        # a steeper transfer function over the same medium: bounds from 0.01 to 1 within a few cells
        sc = _odd_noisy_scene(1)
        tf = np.array(sc.tf_rgba, dtype=np.float32, copy=True)
        tf[:, 3] = np.maximum(tf[:, 3] ** 3 / max(float(tf[:, 3].max()), 1e-6) ** 2, 1e-3 * float(tf[:, 3].max()))
        sc = dataclasses.replace(sc, name=name, tf_rgba=tf, max_opacity=float(tf[:, 3].max()))
    else:
        sc = _odd_noisy_scene(1) if name == "odd_noisy" else scenes.make_scene(name, trace_depth=1)
    c = host.Canvas(hip_dev, sc.width, sc.height)
    try:
        scenes.apply_to_canvas(sc, c, 0)
        c.ReStartRender()
        c.paint()                                        # (the acceleration data is built with the first render)
        hip_dev.synchronize()
        rs = np.random.RandomState(11)
        n = 1 << 21
        rays = _bound8_rays(sc, rs, n)
        out = np.zeros(n, dtype=np.uint32)
        hip_dev.check(hip_dev.lib.svr_selftest_bound8(rays.ctypes.data_as(C.c_void_p), n, out.ctypes.data_as(C.c_void_p)))
    finally:
        c.close()
    tested = (out & 1) != 0
    assert tested.sum() > n // 2, f"{name}: only {tested.sum()} of {n} rays hit the box"
    assert not np.any(out & 4), f"{name}: {np.count_nonzero(out & 4)} look-ups outside the table"
    bad = np.flatnonzero(out & 2)
    assert bad.size == 0, f"{name}: {bad.size} points where the byte culls a draw the accept test could accept, e.g. ray {rays[bad[0]]} byte {out[bad[0]] >> 8}"
    byts = (out[tested] >> 8) & 0xff
    assert byts.min() < 255 and np.unique(byts).size > 4, f"{name}: the table holds no bound to speak of (bytes {np.unique(byts)[:8]})"


def test_fast_bound_lookup_far_camera_falls_back(hip_dev):
    """The look-up's table covers one voxel of rounding error between its fma and the reference's float chain, which holds while the camera is
    within 2^21 / (16 N) volume extents (svr_api.hip, ensure_mask); beyond, the kernel takes the exact cell again.  A camera 4 000 extents
    away looking at the volume through a very long lens: the oracle's image, and the executed fetches of the switch off."""
    base = scenes.make_scene("tiny_head_noisy", trace_depth=1)
    ext = max(host.volume_size(base.dim, base.spacing))
    cam = host.camera_setup((0.0, 0.0, 4000.0 * ext), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 0.02, 0.0, 1.0, 1.0, base.width, base.height)
    sc = dataclasses.replace(base, camera=cam)
    ref_hdr, ref_img, ref_c = oracle_frames(sc, 8)
    assert ref_c["scatter_events"] > 0
    executed = []
    try:
        for fast in (1, 0):
            hip_dev.set_option(abi.OPT_FAST_BOUND, fast)
            hdr, img, c = hip_frames(hip_dev, sc, 8, batch=True)
            assert_bit_exact(hdr, ref_hdr, f"camera 4 000 extents away, fast bound look-up = {fast}")
            assert np.array_equal(img, ref_img)
            executed.append(c["vol_taps_executed"])
        assert executed[0] == executed[1]
    finally:
        hip_dev.set_option(abi.OPT_FAST_BOUND, 1)


def test_pinhole_camera_fast_path(hip_dev):
    """With apeture == 0 (the reference's default) the lens sample is (+-0, +-0) and camera_ray skips its square root and sine / cosine
    (SVR_OPT_PINHOLE_FAST): the oracle's image bit for bit with the switch on and off, with a camera position that holds a -0 component
    (where +-0 can flip a sign bit, so the library must not take the fast path), and off the optical axis."""
    base = scenes.make_scene("tiny_head", trace_depth=2)
    eye = host.zoom_to_extent_eye_dist(host.volume_size(base.dim, base.spacing), base.fov)
    cams = [None,
            host.camera_setup((-0.0, 0.0, eye), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), base.fov, 0.0, 1.0, 1.0, base.width, base.height),
            host.camera_setup((0.3 * eye, -0.0, 0.8 * eye), (1.0, 2.0, 0.0), (0.0, 1.0, 0.1), 38.0, 0.0, 2.5, 1.0, base.width, base.height)]
    for cam in cams:
        sc = dataclasses.replace(base, camera=cam)
        ref_hdr, ref_img, _ = oracle_frames(sc, 4)
        for fast in (1, 0):
            hip_dev.set_option(abi.OPT_PINHOLE_FAST, fast)
            hdr, img, _ = hip_frames(hip_dev, sc, 4, batch=True)
            assert_bit_exact(hdr, ref_hdr, f"pinhole camera, fast path {fast}, camera {cam is not None}")
            assert np.array_equal(img, ref_img)
    hip_dev.set_option(abi.OPT_PINHOLE_FAST, 1)


@pytest.mark.parametrize("apeture", [0.0, 2.5])
def test_lights_in_and_around_the_view_frustum(hip_dev, apeture):
    """SVR_OPT_LIGHT_CULL drops lights no camera ray can reach from the primary rays' nearest-light test (a conservative
    host-side frustum test, lens and pixel jitter included).  Lights squarely in view, straddling the frustum's edge and its
    corner, just outside it, behind the camera and far off axis; pinhole and a wide thin lens: the oracle's image, bit for
    bit, with the culling on and off -- and the lights in view really are seen."""
    base = scenes.make_scene("tiny_head", trace_depth=2)
    eye = host.zoom_to_extent_eye_dist(host.volume_size(base.dim, base.spacing), base.fov)
    half_h = 0.45 * eye * np.tan(np.radians(base.fov) / 2)            # frustum half-height at depth 0.45 eye (the plane z = 0.55 eye, between camera and volume)
    half_w = half_h * base.width / base.height
    z = 0.55 * eye
    toward_eye = (0.0, 0.0, 1.0)
    lights = [host.make_area_light((0.3 * half_w, 0.2 * half_h, z), toward_eye, 2.0, (1.0, 0.8, 0.6), 40.0),          # in view
              host.make_area_light((half_w, -0.5 * half_h, z), toward_eye, 3.0, (0.6, 1.0, 0.8), 40.0),               # straddles the edge
              host.make_area_light((-half_w - 1.0, half_h + 1.0, z), toward_eye, 2.5, (0.7, 0.7, 1.0), 40.0),          # around the corner
              host.make_area_light((half_w + 9.0, 0.0, z), toward_eye, 2.0, (1.0, 1.0, 1.0), 40.0),                   # just outside (a wide lens reaches it)
              host.make_area_light((0.0, 0.0, eye + 6.0), (0.0, 0.0, -1.0), 5.0, (1.0, 1.0, 1.0), 40.0),              # behind the camera
              scenes.default_light(base.dim, base.spacing)]                                                            # far off axis (the GUI default)
    sc = dataclasses.replace(base, lights=lights, apeture=apeture, focal_length=1.0 if apeture == 0.0 else 0.8 * eye)
    ref_hdr, ref_img, _ = oracle_frames(sc, 6)
    for cull in (1, 0):
        hip_dev.set_option(abi.OPT_LIGHT_CULL, cull)
        for batch in (False, True):
            hdr, img, _ = hip_frames(hip_dev, sc, 6, batch=batch)
            assert_bit_exact(hdr, ref_hdr, f"lights around the frustum, apeture {apeture}, cull {cull}, batch {batch}")
            assert np.array_equal(img, ref_img)
    hip_dev.set_option(abi.OPT_LIGHT_CULL, 1)
    # the light in view is seen directly (its radiance is far above anything the volume scatters)
    assert ref_hdr.max() > 10.0 * np.median(ref_hdr[ref_hdr > 0])


@pytest.mark.parametrize("case", ["tiny_head", "small_head", "odd_thin_lens", "c3_window"])
@pytest.mark.parametrize("nframes", [8, 16, 40])
def test_frame_major_group_march(hip_dev, case, nframes):
    """Many-frame launches put the frames of a pixel into one wave, and the lanes of a pixel share one whole-ray
    test (first_occupied_group): counting and non-counting builds against the oracle.  tiny_head has one-voxel
    macro-cells (most lanes fail the quarter-cell test and fall back to their own march), the thin-lens scene has
    per-lane origins (every lane falls back), c3 is the benchmark configuration (every lane shares)."""
    window = None
    if case == "odd_thin_lens":
        sc = _odd_scene(depth=1)
    elif case == "c3_window":
        sc = scenes.make_scene("c3")
        window = (448, 480, 576, 520)
        if nframes != 16:
            pytest.skip("one frame count is enough at full size")
    else:
        sc = scenes.make_scene(case, trace_depth=2 if case == "tiny_head" else 1)
    ref_hdr, _, ref_c = oracle_frames(sc, nframes, window=window)
    for count in (False, True):
        hdr, _, c = hip_frames(hip_dev, sc, nframes, batch=True, count=count, window=window)
        if window:
            x0, y0, x1, y1 = window
            assert_bit_exact(hdr[y0:y1, x0:x1], ref_hdr[y0:y1, x0:x1], f"{case} {nframes} frames count={count}")
        else:
            assert_bit_exact(hdr, ref_hdr, f"{case} {nframes} frames count={count}")
        if count:
            assert c["vol_taps"] == ref_c["vol_taps"] and c["woodcock_iters"] == ref_c["woodcock_iters"]


def test_frame_ahead_matches_per_frame_calls(hip_dev):
    """render_pathtracer traces frames ahead of the calls that ask for them (batches 1, 2, 4 ... 32, the next batch
    started half-way through the current one).  70 per-frame calls, a scene edit in between, the feature on and off:
    always the oracle's accumulator, bit for bit."""
    sc = scenes.make_scene("tiny_head", trace_depth=1)
    ref70, ref_img70, _ = oracle_frames(sc, 70)
    sc2 = dataclasses.replace(sc, density_scale=1.7)
    ref45, _, _ = oracle_frames(sc2, 45)
    for ahead in (1, 0):
        canvas = host.Canvas(hip_dev, sc.width, sc.height)
        try:
            hip_dev.set_option(abi.OPT_FRAME_AHEAD, ahead)
            scenes.apply_to_canvas(sc, canvas)
            for f in range(70):
                canvas.paint()
            hip_dev.synchronize()
            assert_bit_exact(canvas.read_hdr(), ref70, f"70 per-frame calls, frame-ahead {ahead}")
            assert np.array_equal(canvas.read_img(), ref_img70)
            # an edit restarts the render: frames in stock for the old scene must not be used
            canvas.SetDensityScale(1.7)
            for f in range(45):
                canvas.paint()
            hip_dev.synchronize()
            assert_bit_exact(canvas.read_hdr(), ref45, f"45 calls after an edit, frame-ahead {ahead}")
            # back to the first scene, continuing from a frame number that is not a batch start
            canvas.SetDensityScale(1.0)
            for f in range(70):
                canvas.paint()
            hip_dev.synchronize()
            assert_bit_exact(canvas.read_hdr(), ref70, "70 calls after a second edit")
        finally:
            hip_dev.set_option(abi.OPT_FRAME_AHEAD, 1)
            canvas.close()


@pytest.mark.parametrize("name,depth", [("tiny_head", 1), ("tiny_head_noisy", 1), ("tiny_head", 2), ("tiny_bone", 4), ("tiny_head_noisy", 2)])
def test_frame_ahead_steady_state_with_a_sync_per_call(hip_dev, name, depth):
    """The reference's host protocol, gui/canvas.cpp:96-116: render_pathtracer, cudaDeviceSynchronize, frameNo++ -- 200 times, i.e. well
    into the steady state of frame-ahead tracing (64-frame batches traced by the queue builds of the tile kernel into scratch slots while
    the previous batch is consumed; svr_device_synchronize waits for the caller's stream only).  The accumulator and the image after
    frames 100 and 200 are the oracle's, bit for bit; tiny_head_noisy runs the pooled primary walks with the fast bound look-up (at depth 2: the
    deeper pooled machine writing scratch slots), depth 2 / 4 the lane machine."""
    sc = scenes.make_scene(name, trace_depth=depth)
    o = binding.OracleScene(sc)
    acc = o.new_hdr()
    img = np.zeros((o.H, o.W, 4), dtype=np.uint8)
    refs = {}
    for f in range(200):
        o.render_pathtracer(acc, f, trace_depth=depth, img=img, count=False)
        if f + 1 in (100, 200):
            refs[f + 1] = (acc.copy(), img.copy())
    canvas = host.Canvas(hip_dev, sc.width, sc.height)
    try:
        scenes.apply_to_canvas(sc, canvas)
        for f in range(200):
            canvas.paint(sync=True)
            if f + 1 in refs:
                assert_bit_exact(canvas.read_hdr(), refs[f + 1][0], f"{name} depth {depth}: {f + 1} calls, one sync per call")
                assert np.array_equal(canvas.read_img(), refs[f + 1][1])
    finally:
        canvas.close()


@pytest.mark.parametrize("eye", [30.0, 47.9, 130.0, 1000.0, 4100.0, 70000.0])
def test_raycasting_sample_chain_replay(hip_dev, eye):
    """The ray caster replays the float chain t += h in closed form to skip transparent stretches.  Eye distances
    put the samples in one binade, across one or across many binade boundaries (more than the closed form follows),
    step sizes include exact rounding ties (h = 0.5 with t on odd multiples of the ulp), steps far below and far
    above one voxel: image and step count are always the oracle's."""
    sc0 = scenes.make_scene("tiny_head")
    fov = 45.0 if eye < 500 else (4.0 if eye < 10000 else 0.25)
    cam = host.camera_setup((0.3 * eye, -0.2 * eye, eye), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), fov, 0.0, 1.0, 1.0, sc0.width, sc0.height)
    sc = dataclasses.replace(sc0, camera=cam)
    orc = binding.OracleScene(sc)
    canvas = host.Canvas(hip_dev, sc.width, sc.height)
    try:
        scenes.apply_to_canvas(sc, canvas)
        canvas.SetRenderMode(host.Canvas.RENDER_MODE_RAYCASTING)
        # 1 + 2^-7 and 1 + 2^-11: h = 0.5 + half an ulp of t around 70000 / 4100 -> every addition is a rounding tie
        for step in (None, 1.0, 0.0625, 5.0, 1.0 + 2.0 ** -22, 1.0 + 2.0 ** -7, 1.0 + 2.0 ** -11, 3.0e-3 if eye < 200 else 0.37):
            st = sc.step_size() if step is None else step
            ref, rc = orc.render_raycasting(step_size=st)
            canvas.stepSize = st
            hip_dev.set_option(abi.OPT_COUNT, 1)
            hip_dev.reset_counters()
            canvas.paint(sync=True)
            cnt = hip_dev.counters()
            hip_dev.set_option(abi.OPT_COUNT, 0)
            assert cnt["raycast_steps"] == rc["raycast_steps"], (eye, st)
            assert np.array_equal(canvas.read_img(), ref), (eye, st)
    finally:
        hip_dev.set_option(abi.OPT_COUNT, 0)
        canvas.close()


def test_sample_chain_primitives(hip_dev):
    """chain_count / chain_advance of svr_raycast.hip (closed-form replay of t += h) against the brute-force float32
    loop: counts are exact when flagged exact and never too large otherwise; advanced values are the chain's."""
    import random
    f = np.float32
    random.seed(5)
    items = []
    for _ in range(4000):
        t = random.choice([random.uniform(1, 9000), random.uniform(1, 9000), 2.0 ** random.randint(0, 20) - random.uniform(0, 3)])
        t = max(t, 1.0)
        h = random.choice([0.5, 0.43301, 0.5002441, 0.50390625, 0.03125, 2.5, 0.0015, 0.185, random.uniform(0.001, 3)])
        bound = t + random.uniform(0, 200) * h * random.choice([1, 1, 10])
        items.append((t, h, bound, float(random.randint(0, 3000))))
    a = np.array(items, dtype=np.float32)
    out = np.zeros_like(a)
    hip_dev.check(hip_dev.lib.svr_selftest_chain(a.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), len(a)))
    u = out.view(np.uint32)
    n_exact = 0
    for i in range(len(a)):
        t, h, bound, n = a[i]
        c0 = c1 = 0
        x = f(t)
        while x < bound and c0 < 10 ** 6:
            c0 += 1; x = f(x + h)
        x = f(t)
        while x <= bound and c1 < 10 ** 6:
            c1 += 1; x = f(x + h)
        x = f(t)
        for _ in range(int(n)):
            x = f(x + h)
        g0, g1, ta, fl = int(u[i, 0]), int(u[i, 1]), out[i, 2], int(u[i, 3])
        assert (g0 == c0) if fl & 1 else (g0 <= c0), (a[i], g0, c0)
        assert (g1 == c1) if fl & 2 else (g1 <= c1), (a[i], g1, c1)
        if fl & 4:
            assert ta == x, (a[i], ta, x)
        n_exact += bool(fl & 1) + bool(fl & 2) + bool(fl & 4)
    assert n_exact > 2 * len(a)          # the closed form covers most chains


@pytest.mark.parametrize("case", ["voxel_1", "vol_3x2x2", "image_2x2", "tf_transparent", "tf_opaque", "dense_scale", "shard_more_ranks_than_strips"])
def test_degenerate_scenes(hip_dev, case):
    """Corner cases of the acceleration data and the work distribution: one-voxel and few-voxel volumes (a 1x1x1
    macro grid), a 2x2-pixel image, a transfer function that is transparent everywhere (every ray is skipped) or
    opaque everywhere (no cell is empty: distance 0 everywhere), a density scale that saturates the table, and
    more ranks than strips (some ranks own nothing).  Path tracer (per-frame calls and one 16-frame launch) and ray
    caster against the oracle."""
    base = scenes.make_scene("tiny_head", trace_depth=2)
    sc = base
    shard = None
    if case == "voxel_1":
        sc = dataclasses.replace(base, vox=np.full((1, 1, 1), 40000, dtype=np.uint16), max_magnitude=100.0)
    elif case == "vol_3x2x2":
        v = (np.arange(12, dtype=np.uint16).reshape(2, 2, 3) * 5000 + 3000).astype(np.uint16)
        sc = dataclasses.replace(base, vox=v, spacing=(1.0, 2.0, 0.5), max_magnitude=scenes.max_gradient_magnitude(v, (1.0, 2.0, 0.5)))
    elif case == "image_2x2":
        sc = dataclasses.replace(base, width=2, height=2)      # (1x1 divides by W - 1 = 0, cuda_camera.h:68: NaN in the reference too)
    elif case == "tf_transparent":
        tf = base.tf_rgba.copy(); tf[:, 3] = 0.0
        sc = dataclasses.replace(base, tf_rgba=tf, max_opacity=0.5)
    elif case == "tf_opaque":
        tf = base.tf_rgba.copy(); tf[:, 3] = 1.0
        sc = dataclasses.replace(base, tf_rgba=tf, max_opacity=1.0)
    elif case == "dense_scale":
        sc = dataclasses.replace(base, density_scale=37.5)
    elif case == "shard_more_ranks_than_strips":
        shard = (32, 5, 7)                                   # 80 rows = 3 strips of 32 for 7 ranks: rank 5 owns nothing
    if sc.camera is None and case in ("voxel_1", "vol_3x2x2"):
        sc = dataclasses.replace(sc, lights=[host.place_area_light(20.0, 30.0, 12.0, 3.0, (1, 1, 1), 300.0)])
    for frames, batch in ((3, False), (16, True)):
        ref_hdr, ref_img, ref_c = oracle_frames(sc, frames)
        hdr, img, c = hip_frames(hip_dev, sc, frames, batch=batch, shard=shard)
        if shard is None:
            assert_bit_exact(hdr, ref_hdr, f"{case} {frames} frames batch={batch}")
            assert np.array_equal(img, ref_img)
            assert c["vol_taps"] == ref_c["vol_taps"] and c["woodcock_iters"] == ref_c["woodcock_iters"]
        else:
            assert not hdr.any() and c["paths"] == 0         # this rank owns no rows: nothing rendered, nothing touched
    if shard is None:
        ref_rc, rc = binding.OracleScene(sc).render_raycasting()
        canvas = host.Canvas(hip_dev, sc.width, sc.height)
        try:
            scenes.apply_to_canvas(sc, canvas)
            canvas.SetRenderMode(host.Canvas.RENDER_MODE_RAYCASTING)
            canvas.paint(sync=True)
            assert np.array_equal(canvas.read_img(), ref_rc)
        finally:
            canvas.close()


@pytest.mark.parametrize("case", ["tf_opaque", "dense_scale", "voxel_1", "image_2x2", "noisy", "odd", "window", "shard"])
def test_queue_machine_corner_cases(hip_dev, case):
    """The scatter-record queue + per-lane state machine (SVR_OPT_QUEUE = 2: forced on) where it is stressed: every path
    scatters (opaque table: the queue fills and is drained before the 32 tasks are through), saturated table, one-voxel volume,
    2x2 image (almost every lane of a task is dead), unskippable medium, a non-cubic thin-lens scene, a window and a row shard --
    at traceDepth 1 (merged service), 3 and 6 (separate services, roulette), launches of 40 frames (24 idle frame lanes), of
    64 + 3 and of 9; production and counting builds against the oracle."""
    base = scenes.make_scene("tiny_head")
    sc, kw = base, {}
    if case == "tf_opaque":
        tf = base.tf_rgba.copy(); tf[:, 3] = 1.0
        sc = dataclasses.replace(base, tf_rgba=tf, max_opacity=1.0)
    elif case == "dense_scale":
        sc = dataclasses.replace(base, density_scale=37.5)
    elif case == "voxel_1":
        sc = dataclasses.replace(base, vox=np.full((1, 1, 1), 40000, dtype=np.uint16), max_magnitude=100.0,
                                 lights=[host.place_area_light(20.0, 30.0, 12.0, 3.0, (1, 1, 1), 300.0)])
    elif case == "image_2x2":
        sc = dataclasses.replace(base, width=2, height=2)
    elif case == "noisy":
        sc = scenes.make_scene("tiny_head_noisy")
    elif case == "odd":
        sc = _odd_scene(depth=1)
    elif case == "window":
        kw = dict(window=(10, 20, 50, 61))
    elif case == "shard":
        kw = dict(shard=(8, 1, 3))
    hip_dev.set_option(abi.OPT_QUEUE, 2)
    try:
        for depth, frames in ((1, 40), (3, 67), (6, 9)):
            sd = dataclasses.replace(sc, trace_depth=depth)
            ref_hdr, _, ref_c = oracle_frames(sd, frames)
            hdr, _, c = hip_frames(hip_dev, sd, frames, batch=True, **kw)
            if "window" in kw:
                x0, y0, x1, y1 = kw["window"]
                assert_bit_exact(hdr[y0:y1, x0:x1], ref_hdr[y0:y1, x0:x1], f"queue {case} depth {depth}")
            elif "shard" in kw:
                rows = dist.owned_rows(sd.height, *kw["shard"])
                assert_bit_exact(hdr[rows], ref_hdr[rows], f"queue {case} depth {depth}")
            else:
                assert_bit_exact(hdr, ref_hdr, f"queue {case} depth {depth}, {frames} frames")
                assert c["vol_taps"] == ref_c["vol_taps"] and c["woodcock_iters"] == ref_c["woodcock_iters"]
                assert c["scatter_events"] == ref_c["scatter_events"] and c["shadow_walks"] == ref_c["shadow_walks"]
    finally:
        hip_dev.set_option(abi.OPT_QUEUE, 1)


@pytest.mark.parametrize("name,frames", [("c1", 4), ("c2", 2)])
def test_baseline_configs_full_frame(hip_dev, name, frames):
    """BASELINE configs 0 and 1 (64^3 sphere at 256^2; 256^3 head at 512^2, one light) at their full image sizes:
    path tracer (per-frame calls and one many-frame launch) and ray caster against the oracle, every pixel."""
    sc = scenes.make_scene(name)
    ref_hdr, ref_img, ref_c = oracle_frames(sc, frames)
    hdr, img, c = hip_frames(hip_dev, sc, frames)
    assert_bit_exact(hdr, ref_hdr, f"{name} {frames} frames")
    assert np.array_equal(img, ref_img)
    assert c["vol_taps"] == ref_c["vol_taps"] and c["woodcock_iters"] == ref_c["woodcock_iters"] and c["paths"] == ref_c["paths"]
    n = 8 if name == "c2" else 32
    ref_n, _, _ = oracle_frames(sc, n)
    got_n, _, _ = hip_frames(hip_dev, sc, n, batch=True, count=False)
    assert_bit_exact(got_n, ref_n, f"{name} one {n}-frame launch")
    ref_rc, rc = binding.OracleScene(sc).render_raycasting()
    canvas = host.Canvas(hip_dev, sc.width, sc.height)
    try:
        scenes.apply_to_canvas(sc, canvas)
        canvas.SetRenderMode(host.Canvas.RENDER_MODE_RAYCASTING)
        hip_dev.set_option(abi.OPT_COUNT, 1); hip_dev.reset_counters()
        canvas.paint(sync=True)
        assert np.array_equal(canvas.read_img(), ref_rc)
        assert hip_dev.counters()["raycast_steps"] == rc["raycast_steps"]
    finally:
        hip_dev.set_option(abi.OPT_COUNT, 0)
        canvas.close()


@pytest.mark.parametrize("depth", [1, 3])
def test_nan_guard_drops_non_finite_samples(hip_dev, depth):
    """The reference's running mean keeps a NaN for good (pathtracer.cu:81-84,279), and its own arithmetic yields one now and then
    (0/0 in the microfacet term).  Here EVERY next-event estimate is NaN: a light of radius 0 has area 0, radiance = x / 0 = inf and
    pdf = inf, so Ld = (B * kf) * inf / inf (pathtracer.cu:191-198).  Default mode: the HIP image has its NaNs exactly where the
    oracle has them (x86 and gfx950 disagree about the sign bit of a default NaN: assert_bit_exact treats NaN = NaN) and the same
    bits elsewhere.  SVR_OPT_NAN_GUARD = 1 (opt-in): no NaN is left, every pixel that had none keeps its bits, and the two
    accumulation paths (one 16-frame folding launch; 16 render_pathtracer calls = frames traced ahead + k_resolve) agree."""
    base = scenes.make_scene("tiny_head", trace_depth=depth)
    sc = dataclasses.replace(base, lights=[host.place_area_light(20.0, 30.0, 80.0, 0.0, (1, 1, 1), 300.0)])
    N = 16
    ref_hdr, ref_img, _ = oracle_frames(sc, N)
    n_nan = int(np.isnan(ref_hdr).any(axis=2).sum())
    assert n_nan > 100, n_nan
    hdr, img, _ = hip_frames(hip_dev, sc, N, batch=True)
    assert_bit_exact(hdr, ref_hdr, "radius-0 light, default mode (NaNs in the same pixels)")
    assert np.array_equal(img, ref_img)
    hip_dev.set_option(abi.OPT_NAN_GUARD, 1)
    try:
        g_hdr, g_img, _ = hip_frames(hip_dev, sc, N, batch=True)
        g2_hdr, g2_img, _ = hip_frames(hip_dev, sc, N, batch=False)
    finally:
        hip_dev.set_option(abi.OPT_NAN_GUARD, 0)
    assert np.isfinite(g_hdr).all()
    clean = ~np.isnan(ref_hdr).any(axis=2)
    assert_bit_exact(g_hdr[clean], ref_hdr[clean], "NAN_GUARD: pixels without a NaN keep their bits")
    assert_bit_exact(g_hdr, g2_hdr, "NAN_GUARD: folding launch vs per-frame calls")
    assert np.array_equal(g_img, g2_img)
    assert (g_hdr[~clean] >= 0).all()


@pytest.mark.parametrize("name,depth", [("tiny_head", 2), ("tiny_head", 4), ("tiny_bone", 6), ("small_head", 3), ("odd", 3), ("tiny_head_noisy", 3)])
def test_split_kernels_bit_exact(hip_dev, name, depth):
    """Deeper paths as two kernels (csrc/svr_trace_split.hip: front half -> chunks of path records -> lane machine -> scratch slots -> k_resolve),
    forced on (SVR_OPT_SPLIT = 2: from traceDepth 2) and off, one 70-frame call (a 64-frame and a 6-frame launch: the short one takes the ordinary
    path) and one 40-frame call; production and counting builds against the oracle: accumulator, image and the reference's counters.
    (tiny_head_noisy: its primary walks are pooled, so the fused kernel renders whatever the switch says.)"""
    from tests.test_local_majorant_gpu import _make
    sc = _make(name, trace_depth=depth)
    for frames in (70, 40):
        ref_hdr, ref_img, ref_c = oracle_frames(sc, frames)
        for split in (2, 0):
            hip_dev.set_option(abi.OPT_SPLIT, split)
            try:
                hdr, img, c = hip_frames(hip_dev, sc, frames, batch=True)
            finally:
                hip_dev.set_option(abi.OPT_SPLIT, 0)
            assert_bit_exact(hdr, ref_hdr, f"{name} depth {depth}, {frames} frames, SVR_OPT_SPLIT = {split}")
            assert np.array_equal(img, ref_img)
            assert c["vol_taps"] == ref_c["vol_taps"] and c["woodcock_iters"] == ref_c["woodcock_iters"]
            assert c["scatter_events"] == ref_c["scatter_events"] and c["shadow_walks"] == ref_c["shadow_walks"] and c["paths"] == ref_c["paths"]


@pytest.mark.parametrize("case", ["window", "shard", "window_direct", "shard_direct"])
def test_split_and_direct_builds_under_a_window_and_a_row_shard(hip_dev, case):
    """The round-4 launch forms where pixels are not the whole frame: the two-kernel form of deeper paths (path ids = frame << 26 | GLOBAL pixel index)
    and the DIRECT queue builds of frames traced ahead (a path writes its scratch slot itself), under a render window and under an interleaved row
    shard; against the oracle on the owned pixels, untouched elsewhere."""
    sc = scenes.make_scene("small_head", trace_depth=3 if "direct" not in case else 1)
    kw = dict(window=(37, 50, 201, 190)) if "window" in case else dict(shard=(8, 2, 3))
    frames = 72
    ref_hdr, _, _ = oracle_frames(sc, frames, window=kw.get("window"))
    hip_dev.set_option(abi.OPT_SPLIT, 2)
    try:
        # batch = one 72-frame call (a 64-frame and an 8-frame launch); per-frame calls = frames traced ahead (batches up to 64, DIRECT builds)
        hdr, _, _ = hip_frames(hip_dev, sc, frames, batch="direct" not in case, count=False, **kw)
    finally:
        hip_dev.set_option(abi.OPT_SPLIT, 0)
    if "window" in case:
        x0, y0, x1, y1 = kw["window"]
        assert_bit_exact(hdr[y0:y1, x0:x1], ref_hdr[y0:y1, x0:x1], case)
        mask = np.ones(hdr.shape[:2], bool)
        mask[y0:y1, x0:x1] = False
        assert not hdr[mask].any()
    else:
        rows = dist.owned_rows(sc.height, *kw["shard"])
        assert_bit_exact(hdr[rows], ref_hdr[rows], case)
        other = np.setdiff1d(np.arange(sc.height), rows)
        assert not hdr[other].any()
