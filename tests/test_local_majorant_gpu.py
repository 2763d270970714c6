"""SVR_OPT_LOCAL_MAJORANT (opt-in, default off): delta tracking against per-macro-cell majorants instead of the reference's
single global one (core/woodcock_tracking.h:29-31) -- BASELINE.json's "Woodcock max-density acceleration"
(sunvolumerender_amd/csrc/svr_trace_lm.hip).  The law of every collision point is the reference's, the consumption of
random numbers is not, so the mode is NOT bit-identical to the default one.  Its contract is the converged-image
tolerance BASELINE.json's north star states, written like tests/test_fast_math_gpu.py:

  A = default mode, frames 0..N-1;  B = default mode, frames N..2N-1 (an independent estimate: B = 2 mean(0..2N-1) - A);
  F = local-majorant mode, frames 0..N-1
  * rmse(F, B) <= 1.10 * rmse(A, B)   -- F is as close to an independent exact estimate as an exact render is
  * rmse(F, A) <= 1.10 * rmse(A, B)   -- (F and A share only the camera draws: they are nearly independent too)
  * |mean(F) - mean(A)| <= max(0.3 % of mean(A), 4 standard errors) per channel -- no bias
The per-pixel L2 is taken on the IMAGE, i.e. after the reference's exposure curve c = 1 - exp(-16 L exposure)
(core/tonemapping.h:13-21, before its pow): bounded, so the rare 1/pdf fireflies of deep paths -- which dominate an
L2 on raw radiance and make the ratio of two such L2s a coin toss -- count as the saturated pixels they are; where the
radiance itself is light-tailed (trace depth 1) the same inequalities are ALSO required on the raw HDR values.
The means are compared on the raw HDR values always; their standard error is rmse_channel(A, B) / sqrt(pixels).
On c2 (256^3, depth 2), c3n (noisy non-zero air: nothing is exactly transparent; depth 1 and 4), c5 (1024^3) and c3b (c3
under the bone transfer function: transparent air and translucent tissue, depth 2), full frames; and against the ORACLE's own 256-spp image on a window (the unbiasedness check does not rest on the HIP default
mode alone).
Inside the mode a frame's radiance is still a pure function of (scene, pixel, frame): per-frame calls, batches, row
shards and the counting build agree bit for bit.  The default mode is untouched by the switch."""
import numpy as np
import pytest

from oracle import binding
from sunvolumerender_amd import abi, host, scenes
from tests.util import assert_bit_exact

pytestmark = pytest.mark.gpu


def _make(name, **kw):
    """named scenes, plus 'odd': non-cubic volume, anisotropic spacing (1, 0.5, 2), off-axis thin-lens camera, clip planes, density scale 1.2, the bone
    transfer function (bound classes between 0 and 1), image size not a multiple of anything (tests/test_more_gpu.py::_odd_scene)"""
    if name == "odd":
        from tests.test_more_gpu import _odd_scene
        return _odd_scene(depth=kw.get("trace_depth", 2))
    return scenes.make_scene(name, **kw)


def _canvas(dev, name, macro_shift_min=0, **kw):
    """macro_shift_min: SVR_OPT_MACRO_SHIFT_MIN while the volume texture is created -- macro-cells of at least 2^v voxels, so that the
    coarse-grid code of the mode (sub-cell occupancy needs cells of >= 2 voxels; by default only volumes beyond 512^3 have it on) runs on
    the small scenes whose means can be resolved to 0.05 %"""
    sc = _make(name, **kw)
    canvas = host.Canvas(dev, sc.width, sc.height)
    dev.set_option(abi.OPT_MACRO_SHIFT_MIN, macro_shift_min)
    try:
        scenes.apply_to_canvas(sc, canvas)
    finally:
        dev.set_option(abi.OPT_MACRO_SHIFT_MIN, 0)
    return sc, canvas


def _render(dev, canvas, lm, frames, sub=1):
    """progressive render; returns the accumulator after each entry of `frames` (cumulative calls).  sub = SVR_OPT_LM_SUBCELLS
    (0 off, 1 where macro-cells are >= 16 voxels, 2 wherever a fine level exists): part of the mode's definition"""
    dev.set_option(abi.OPT_LOCAL_MAJORANT, 1 if lm else 0)
    dev.set_option(abi.OPT_LM_SUBCELLS, sub)
    canvas.ReStartRender()
    out = []
    for n in frames:
        canvas.paint_frames(n)
        dev.synchronize()
        out.append(canvas.read_hdr().astype(np.float64))
    dev.set_option(abi.OPT_LOCAL_MAJORANT, 0)
    dev.set_option(abi.OPT_LM_SUBCELLS, 1)
    return out


def _rmse(x, y):
    return float(np.sqrt(np.mean((x - y) ** 2)))


def _curve(x, exposure=1.0):
    """the reference's exposure curve, core/tonemapping.h:13-21 (what hdr_to_ldr shows, before its pow)"""
    return 1.0 - np.exp(-16.0 * np.maximum(x, 0.0) * exposure)


def _drop_reference_nans(*imgs):
    """The reference's own arithmetic yields NaN for a handful of paths in 10^8..10^9 -- e.g. a light direction exactly
    perpendicular to the shading normal: G = 0 over |n.wi| = 0 in the microfacet term times cos = 0 (pathtracer.cu:106-131,
    core/bsdf/microfacet.h:52-68; the oracle does the same) -- and the running mean keeps it for good.  Which path it hits depends
    on the random numbers, i.e. on the mode.  Such pixels (at most a few per image) are zeroed in every image compared."""
    bad = np.zeros(imgs[0].shape[:2], dtype=bool)
    for a in imgs:
        bad |= ~np.isfinite(a).all(axis=2)
    assert bad.sum() <= max(4, bad.size // 100000), int(bad.sum())
    return [np.where(bad[..., None], 0.0, a) for a in imgs]


def _check_converged(A, B, F, what, hdr_l2=True):
    assert not np.array_equal(F, A), f"{what}: the local-majorant mode produced the default mode's bits: did it run?"
    A, B, F = _drop_reference_nans(A, B, F)
    assert (F >= 0).all()
    spaces = [("image", _curve(A), _curve(B), _curve(F))] + ([("hdr", A, B, F)] if hdr_l2 else [])
    for space, a, b, f in spaces:
        noise = _rmse(a, b)
        assert noise > 0
        assert _rmse(f, b) <= 1.10 * noise, (what, space, _rmse(f, b), noise)
        assert _rmse(f, a) <= 1.10 * noise, (what, space, _rmse(f, a), noise)
    npx = A.shape[0] * A.shape[1]
    mA, mF = A.mean(axis=(0, 1)), F.mean(axis=(0, 1))
    se = np.sqrt(np.mean((A - B) ** 2, axis=(0, 1)) / npx)          # standard error of mean(F) - mean(A): var(A - B) = var(F - A) per pixel
    assert np.all(np.abs(mF - mA) <= np.maximum(3e-3 * mA, 4.0 * se)), (what, mA, mF, se)


@pytest.mark.parametrize("name,depth,spp,sub", [("c2", 2, 256, 1), ("c3n", 1, 128, 1), ("c3n", 4, 64, 1), ("c5", 1, 128, 1), ("c5", 1, 128, 0), ("c3b", 2, 128, 1), ("c3", 1, 128, 2)])
def test_local_majorant_converged_image_within_noise(hip_dev, name, depth, spp, sub):
    """sub: SVR_OPT_LM_SUBCELLS -- c5 (16-voxel macro-cells) has the sub-cell occupancy on by default and is also rendered without it,
    c3 (8-voxel cells) has it off by default and is also rendered with it forced on"""
    sc, canvas = _canvas(hip_dev, name, trace_depth=depth)
    try:
        A, A2 = _render(hip_dev, canvas, False, (spp, spp))
        B = 2.0 * A2 - A
        (F,) = _render(hip_dev, canvas, True, (spp,), sub=sub)
        _check_converged(A, B, F, f"{name} depth {depth} sub-cells {sub}", hdr_l2=depth <= 1)
        if sub != 1:
            (F1,) = _render(hip_dev, canvas, True, (spp,))
            assert not np.array_equal(F1, F), "SVR_OPT_LM_SUBCELLS changed nothing: did the sub-cell branch run?"
        # the default mode is untouched by the switch
        (A3,) = _render(hip_dev, canvas, False, (spp,))
        assert_bit_exact(A3.astype(np.float32), A.astype(np.float32), "default mode after the local-majorant mode was used")
    finally:
        hip_dev.set_option(abi.OPT_LOCAL_MAJORANT, 0)
        canvas.close()


_ORACLE_WINDOW = {}


@pytest.mark.parametrize("sub", [1, 2])
def test_local_majorant_unbiased_against_the_oracle(hip_dev, sub):
    """The same three inequalities with the ORACLE in the place of the default mode: a 96x48 window of small_head (128^3,
    256^2, 3 lights + env, depth 2) at 256 spp; O = oracle frames 0..255, O2 = oracle frames 256..511.  small_head has macro-cells of
    2 voxels: sub = 2 forces the sub-cell occupancy on (1-voxel fine cells), sub = 1 leaves it off there."""
    sc, canvas = _canvas(hip_dev, "small_head", trace_depth=2)
    try:
        N = 256
        win = (80, 100, 176, 148)
        x0, y0, x1, y1 = win
        if "O" not in _ORACLE_WINDOW:                          # (the oracle's 512 frames of the window are rendered once for both settings)
            o = binding.OracleScene(sc)
            acc = o.new_hdr()
            for f in range(N):
                o.render_pathtracer(acc, f, trace_depth=2, window=win, count=False)
            O = acc[y0:y1, x0:x1].astype(np.float64)
            for f in range(N, 2 * N):
                o.render_pathtracer(acc, f, trace_depth=2, window=win, count=False)
            _ORACLE_WINDOW["O"], _ORACLE_WINDOW["O2"] = O, 2.0 * acc[y0:y1, x0:x1].astype(np.float64) - O
        O, O2 = _ORACLE_WINDOW["O"], _ORACLE_WINDOW["O2"]
        (F,) = _render(hip_dev, canvas, True, (N,), sub=sub)
        _check_converged(O, O2, F[y0:y1, x0:x1], f"small_head window vs oracle, sub-cells {sub}", hdr_l2=False)
        if sub == 2:
            (F1,) = _render(hip_dev, canvas, True, (N,))
            assert not np.array_equal(F1, F), "SVR_OPT_LM_SUBCELLS = 2 changed nothing on small_head: did the sub-cell branch run?"
    finally:
        hip_dev.set_option(abi.OPT_LOCAL_MAJORANT, 0)
        canvas.close()


@pytest.mark.parametrize("name,depth,shift,sub", [("tiny_head", 1, 0, 1), ("tiny_head", 2, 0, 1), ("tiny_head_noisy", 1, 0, 1), ("tiny_bone", 2, 0, 1), ("odd", 1, 0, 1), ("odd", 3, 0, 1),
                                                  ("tiny_head", 1, 2, 2), ("tiny_head", 2, 3, 2), ("tiny_bone", 2, 2, 2), ("odd", 1, 2, 2), ("odd", 3, 1, 2), ("tiny_head", 1, 2, 0)])
def test_local_majorant_means_agree_at_high_sample_counts(hip_dev, name, depth, shift, sub):
    """A bias of a few 0.1 % hides in the noise of 256 spp.  Small frames at 8192 spp, default mode against local-majorant mode:
    the per-channel means of the frame and of its four quadrants within 4 standard errors (estimated from two independent halves
    of the default render), or 0.05 % where the noise is smaller than that.
    shift / sub: macro-cells of 2^shift voxels (SVR_OPT_MACRO_SHIFT_MIN) with the SUB-CELL OCCUPANCY forced on (sub = 2: the free path
    is spent only in the occupied eighths of a macro-cell, svr_trace_lm.hip lm_step -- on by default only for volumes beyond 512^3)
    or off (sub = 0) on the coarse grid: the branch c5's numbers rest on, held to the same test as the rest of the mode."""
    sc, canvas = _canvas(hip_dev, name, macro_shift_min=shift, trace_depth=depth)
    try:
        N = 4096
        A, A2 = _render(hip_dev, canvas, False, (N, N))          # frames 0..N-1, then the mean of 0..2N-1
        B = 2.0 * A2 - A
        (F2,) = _render(hip_dev, canvas, True, (2 * N,), sub=sub)
        if sub == 2:
            (G,) = _render(hip_dev, canvas, True, (64,), sub=0)
            (G2,) = _render(hip_dev, canvas, True, (64,), sub=2)
            assert not np.array_equal(G, G2), "SVR_OPT_LM_SUBCELLS = 2 changed nothing: did the sub-cell branch run?"
        H, W = A.shape[:2]
        A, A2, B, F2 = _drop_reference_nans(A, A2, B, F2)
        for (y0, y1, x0, x1) in [(0, H, 0, W), (0, H // 2, 0, W // 2), (0, H // 2, W // 2, W), (H // 2, H, 0, W // 2), (H // 2, H, W // 2, W)]:
            a, b, f = A2[y0:y1, x0:x1], (A - B)[y0:y1, x0:x1], F2[y0:y1, x0:x1]
            npx = a.shape[0] * a.shape[1]
            # var of a 2N-spp pixel = var(A - B) / 4; F2 and A2 are independent: var(mean F2 - mean A2) = 2 * that / npx
            se = np.sqrt(2.0 * np.mean(b ** 2, axis=(0, 1)) / 4.0 / npx)
            dm = np.abs(f.mean(axis=(0, 1)) - a.mean(axis=(0, 1)))
            assert np.all(dm <= np.maximum(4.0 * se, 5e-4 * a.mean(axis=(0, 1)))), (name, depth, (y0, y1, x0, x1), dm, se, a.mean(axis=(0, 1)))
    finally:
        hip_dev.set_option(abi.OPT_LOCAL_MAJORANT, 0)
        canvas.close()


@pytest.mark.parametrize("name,depth", [("tiny_head", 1), ("tiny_head", 3), ("tiny_head_noisy", 2), ("small_head", 1), ("odd", 1), ("odd", 4)])
def test_local_majorant_is_a_pure_function_of_scene_pixel_frame(hip_dev, name, depth):
    """Inside the mode: one 24-frame call == 24 render_pathtracer calls (frames traced ahead, scratch slots + k_resolve) ==
    the counting build == the union of 3 row shards; and rendering twice gives the same bits."""
    sc, canvas = _canvas(hip_dev, name, trace_depth=depth)
    dev = hip_dev
    try:
        N = 24
        dev.set_option(abi.OPT_LOCAL_MAJORANT, 1)

        def run(batch=True, count=False, shard=None):
            dev.set_option(abi.OPT_COUNT, 1 if count else 0)
            if shard is not None:
                dev.check(dev.lib.svr_set_row_shard(*shard))
            dev.reset_counters()
            canvas.ReStartRender()
            if batch:
                canvas.paint_frames(N)
            else:
                for _ in range(N):
                    canvas.paint()
            dev.synchronize()
            dev.lib.svr_set_row_shard(0, 0, 1)
            dev.set_option(abi.OPT_COUNT, 0)
            return canvas.read_hdr(), canvas.read_img(), dev.counters()

        a, ai, _ = run()
        b, bi, _ = run()
        assert_bit_exact(a, b, "local-majorant mode, rendered twice")
        c, ci, cnt = run(count=True)
        assert_bit_exact(a, c, "local-majorant mode: counting build")
        assert np.array_equal(ai, ci)
        assert cnt["paths"] == sc.width * sc.height * N and 0 < cnt["vol_taps_executed"] and cnt["scatter_events"] > 0
        d, di, _ = run(batch=False)
        assert_bit_exact(a, d, "local-majorant mode: 24 render_pathtracer calls vs one 24-frame call")
        assert np.array_equal(ai, di)
        union = np.zeros_like(a)
        for r in range(3):
            part, _, _ = run(shard=(8, r, 3))
            rows = [y for y in range(sc.height) if (y // 8) % 3 == r]
            union[rows] = part[rows]
        assert_bit_exact(a, union, "local-majorant mode: union of 3 row shards")
        # and it is not the default mode's image
        dev.set_option(abi.OPT_LOCAL_MAJORANT, 0)
        e, _, _ = run()
        assert not np.array_equal(a, e)
    finally:
        dev.set_option(abi.OPT_LOCAL_MAJORANT, 0)
        dev.set_option(abi.OPT_COUNT, 0)
        dev.lib.svr_set_row_shard(0, 0, 1)
        canvas.close()


@pytest.mark.parametrize("name,depth", [("c3", 1), ("c3n", 1), ("c3", 2), ("c3", 5), ("c3n", 4)])
def test_local_majorant_pool_equals_straight_line_full_frame(hip_dev, name, depth):
    """Folding launches run the POOL forms of the kernel (a wave takes a batch of 16 tasks through gen / walk pool / batched
    shading / walk pool / [batched BSDF sampling / ...] / fold, csrc/svr_trace_lm.hip: records at traceDepth 1, one slot per
    path beyond); SVR_OPT_LOCAL_MAJORANT = 2 forces the straight-line form.  Scheduling only: bit-identical on the whole 1024^2
    frame, one 128-frame call (two 64-frame launches: a wave = one pixel x 64 frames, shared whole-ray tests) and one 24-frame
    call (2 pixels x 32 frame lanes, 8 of them dead, per-lane tests)."""
    sc, canvas = _canvas(hip_dev, name, trace_depth=depth)
    dev = hip_dev
    try:
        for n in (128, 24):
            imgs = []
            for mode in (1, 2):
                dev.set_option(abi.OPT_LOCAL_MAJORANT, mode)
                canvas.ReStartRender()
                canvas.paint_frames(n, sync=True)
                imgs.append((canvas.read_hdr(), canvas.read_img()))
            assert_bit_exact(imgs[0][0], imgs[1][0], f"{name} depth {depth}: pool vs straight-line local-majorant paths, {n} frames")
            assert np.array_equal(imgs[0][1], imgs[1][1])
            assert imgs[0][0].max() > 0
    finally:
        dev.set_option(abi.OPT_LOCAL_MAJORANT, 0)
        canvas.close()


@pytest.mark.parametrize("name,depth", [("small_head", 1), ("small_head_noisy", 3), ("odd", 2)])
def test_local_majorant_image_does_not_depend_on_scheduling_or_layout(hip_dev, name, depth):
    """What is scheduling or storage stays scheduling or storage in this mode too: the pool's cells per turn and refill /
    settle thresholds (SVR_OPT_LM_TUNE), the number of persistent blocks, the light culling, the pinhole fast path and the
    four volume layouts give the same bits as the defaults (64-frame calls: the pool forms of the kernel)."""
    dev = hip_dev
    sc = _make(name, trace_depth=depth)
    imgs = {}
    for layout in (abi.LAYOUT_AUTO, abi.LAYOUT_LINEAR, abi.LAYOUT_BRICK, abi.LAYOUT_PAIR, abi.LAYOUT_CELL):
        canvas = host.Canvas(dev, sc.width, sc.height)
        scenes.apply_to_canvas(sc, canvas, layout)
        try:
            variants = [("defaults", [])]
            if layout == abi.LAYOUT_AUTO:
                variants += [("lm_tune 1/8/8", [(abi.OPT_LM_TUNE, 1 | (8 << 8) | (8 << 16))]), ("lm_tune 4/40/56", [(abi.OPT_LM_TUNE, 4 | (40 << 8) | (56 << 16))]),
                             ("2 blocks per CU", [(abi.OPT_BLOCKS_PER_CU, 2)]), ("no light culling", [(abi.OPT_LIGHT_CULL, 0)]), ("no pinhole fast path", [(abi.OPT_PINHOLE_FAST, 0)])]
            for what, opts in variants:
                old = [(o, dev.lib.svr_get_option(o)) for o, _ in opts]
                try:
                    dev.set_option(abi.OPT_LOCAL_MAJORANT, 1)
                    for o, v in opts:
                        dev.set_option(o, v)
                    canvas.ReStartRender()
                    canvas.paint_frames(64, sync=True)
                    imgs[(layout, what)] = (canvas.read_hdr(), canvas.read_img())
                finally:
                    for o, v in old:
                        dev.set_option(o, v)
                    dev.set_option(abi.OPT_LOCAL_MAJORANT, 0)
        finally:
            canvas.close()
    ref = imgs[(abi.LAYOUT_AUTO, "defaults")]
    assert ref[0].max() > 0
    for key, (hdr, img) in imgs.items():
        assert_bit_exact(hdr, ref[0], f"{name} depth {depth}, local majorants: layout {key[0]}, {key[1]}")
        assert np.array_equal(img, ref[1])


def test_local_majorant_falls_back_where_it_cannot_run(hip_dev):
    """Without the acceleration data (SVR_OPT_EMPTY_SKIP = 0) the switch is inert: the default kernel renders, bit-exact."""
    sc, canvas = _canvas(hip_dev, "tiny_head", trace_depth=2)
    dev = hip_dev
    try:
        canvas.ReStartRender()
        canvas.paint_frames(8, sync=True)
        ref = canvas.read_hdr()
        dev.set_option(abi.OPT_LOCAL_MAJORANT, 1)
        dev.set_option(abi.OPT_EMPTY_SKIP, 0)
        canvas.ReStartRender()
        canvas.paint_frames(8, sync=True)
        assert_bit_exact(canvas.read_hdr(), ref, "LOCAL_MAJORANT with EMPTY_SKIP = 0 renders with the default kernel")
    finally:
        dev.set_option(abi.OPT_LOCAL_MAJORANT, 0)
        dev.set_option(abi.OPT_EMPTY_SKIP, 1)
        canvas.close()
