"""SVR_OPT_ENV_NEE (opt-in, default off): importance sampling of the environment map (sunvolumerender_amd/csrc/svr_trace_env.hip) --
SURVEY 8(f) row N4's second half.  The reference only has the lat-long lookup (core/lights/cuda_environment_light.h:58-72), enabled here by
SVR_OPT_ENV_ON_ESCAPE; with it the environment reaches the medium only through the directions the BSDF / phase sampling picks.  The mode adds
one direction drawn from the map's luminance per scatter event and combines the two estimates with the balance heuristic; its contract is
"the escape-only estimator's image in expectation":

  A = default mode (escape only), frames 0..N-1;  B = default mode, frames N..2N-1 (independent);  E = the mode, frames 0..2N-1
  * per-channel means of the frame and of its quadrants: |mean(E) - mean(A u B)| <= 4 standard errors (estimated from A - B) or 0.3 %
  * E is (much) closer to the converged image than an escape-only render of the same length: rmse(E, A u B) < rmse(A, B) / 2
on a scene lit by a small bright sun in the map (where escape-only rendering is hopeless) at trace depth 2, 3 and 5 (roulette from the
fourth bounce on).  Inside the mode a frame is a pure function of (scene, pixel, frame); without a map, without the escape term or at trace
depth 1 the switch is inert and the default kernel renders, bit-exact."""
import dataclasses

import numpy as np
import pytest

from sunvolumerender_amd import abi, host, scenes
from tests.util import assert_bit_exact

pytestmark = pytest.mark.gpu


def sun_map(w=128, h=64, sun=2000.0, sky=0.1):
    """lat-long RGBA float32: dim sky, a 4 x 4 texel sun high above the volume (theta ~ 0.6 rad from +y)"""
    img = np.full((h, w, 4), sky, dtype=np.float32)
    img[..., 2] *= 1.5
    img[10:14, 20:24, :3] = np.array([sun, 0.9 * sun, 0.7 * sun], dtype=np.float32)
    img[..., 3] = 1.0
    return np.ascontiguousarray(img)


def _scene(name, depth, **kw):
    sc = scenes.make_scene(name, trace_depth=depth)
    return dataclasses.replace(sc, env_map=sun_map(), env_intensity=1.0, env_on_escape=True, **kw)


def _render(dev, canvas, nee, frames):
    dev.set_option(abi.OPT_ENV_NEE, 1 if nee else 0)
    canvas.ReStartRender()
    out = []
    for n in frames:
        canvas.paint_frames(n)
        dev.synchronize()
        out.append(canvas.read_hdr().astype(np.float64))
    dev.set_option(abi.OPT_ENV_NEE, 0)
    return out


@pytest.mark.parametrize("name,depth,offset,quadrants", [("tiny_head", 2, (0.0, 0.0), True), ("tiny_head", 3, (0.0, 0.1), True), ("tiny_bone", 3, (0.3, 0.0), True), ("tiny_head", 5, (0.0, 0.0), True),
                                                         ("tiny_head_noisy", 3, (0.55, 0.1), False)])
def test_env_nee_means_agree_with_the_escape_only_estimator(hip_dev, name, depth, offset, quadrants):
    """quadrants = False (tiny_head_noisy: thin fog everywhere): only the whole frame is compared.  In fog the reference picks the BRDF branch with the
    tiny probability Pbrdf and divides the throughput by it (pathtracer.cu:105, 264-268); such a path then hitting the sun by chance is a sample of ~10^6
    times the mean -- the escape-only render has pixels of 600 at a mean of 4 after 8 192 spp, a quadrant's mean moves by 5 % with one of them, and the
    standard error estimated from two halves says nothing about the ones that did not land.  (The mode samples the sun every time: its largest pixel is 3 x
    smaller.)"""
    sc = _scene(name, depth, env_offset=offset)
    canvas = host.Canvas(hip_dev, sc.width, sc.height)
    try:
        scenes.apply_to_canvas(sc, canvas)
        N = 4096
        A, A2 = _render(hip_dev, canvas, False, (N, N))            # frames 0..N-1, then the mean of 0..2N-1
        B = 2.0 * A2 - A
        (E,) = _render(hip_dev, canvas, True, (2 * N,))
        assert not np.array_equal(E, A2), "SVR_OPT_ENV_NEE produced the default mode's bits: did it run?"
        # (the reference's own 0/0 -- a light direction exactly perpendicular to the shading normal, pathtracer.cu:106-131 -- kills a pixel now and then
        # in either mode: such pixels are dropped from every image compared, as in tests/test_local_majorant_gpu.py)
        from tests.test_local_majorant_gpu import _drop_reference_nans
        A, A2, B, E = _drop_reference_nans(A, A2, B, E)
        assert (E >= 0).all()
        H, W = A.shape[:2]
        regions = [(0, H, 0, W)] + ([(0, H // 2, 0, W // 2), (0, H // 2, W // 2, W), (H // 2, H, 0, W // 2), (H // 2, H, W // 2, W)] if quadrants else [])
        for (y0, y1, x0, x1) in regions:
            a2, d, e = A2[y0:y1, x0:x1], (A - B)[y0:y1, x0:x1], E[y0:y1, x0:x1]
            npx = a2.shape[0] * a2.shape[1]
            # var of a 2N-spp pixel of the default mode = var(A - B) / 4; the mode's own variance is far smaller: bounded by the same figure
            se = np.sqrt(2.0 * np.mean(d ** 2, axis=(0, 1)) / 4.0 / npx)
            dm = np.abs(e.mean(axis=(0, 1)) - a2.mean(axis=(0, 1)))
            assert np.all(dm <= np.maximum(4.0 * se, 3e-3 * a2.mean(axis=(0, 1)))), (name, depth, (y0, y1, x0, x1), dm, se, a2.mean(axis=(0, 1)))
        # and the point of it: at equal length the mode is far closer to the converged image than the escape-only render is to itself
        curve = lambda x: 1.0 - np.exp(-16.0 * np.maximum(x, 0.0))     # the reference's exposure curve (core/tonemapping.h:13-21): bounded
        noise_default = float(np.sqrt(np.mean((curve(A) - curve(B)) ** 2)))
        err_mode = float(np.sqrt(np.mean((curve(E) - curve(A2)) ** 2)))
        assert err_mode < noise_default, (err_mode, noise_default)
    finally:
        hip_dev.set_option(abi.OPT_ENV_NEE, 0)
        canvas.close()


def test_env_nee_is_a_pure_function_of_scene_pixel_frame(hip_dev):
    """one 40-frame call == 40 render_pathtracer calls == the counting build == rendering twice; all four volume layouts agree"""
    sc = _scene("tiny_head", 3)
    dev = hip_dev
    ref = None
    for layout in (abi.LAYOUT_AUTO, abi.LAYOUT_LINEAR, abi.LAYOUT_BRICK, abi.LAYOUT_PAIR):
        canvas = host.Canvas(dev, sc.width, sc.height)
        try:
            scenes.apply_to_canvas(sc, canvas, layout)
            dev.set_option(abi.OPT_ENV_NEE, 1)

            def run(batch=True, count=False):
                dev.set_option(abi.OPT_COUNT, 1 if count else 0)
                canvas.ReStartRender()
                if batch:
                    canvas.paint_frames(40)
                else:
                    for _ in range(40):
                        canvas.paint()
                dev.synchronize()
                dev.set_option(abi.OPT_COUNT, 0)
                return canvas.read_hdr(), canvas.read_img()

            a, ai = run()
            if ref is None:
                ref = (a, ai)
                b, _ = run()
                assert_bit_exact(a, b, "env NEE, rendered twice")
                c, ci = run(count=True)
                assert_bit_exact(a, c, "env NEE: counting build")
                d, di = run(batch=False)
                assert_bit_exact(a, d, "env NEE: 40 render_pathtracer calls vs one 40-frame call")
                assert np.array_equal(ai, di)
            else:
                assert_bit_exact(a, ref[0], f"env NEE: layout {layout}")
                assert np.array_equal(ai, ref[1])
        finally:
            dev.set_option(abi.OPT_ENV_NEE, 0)
            dev.set_option(abi.OPT_COUNT, 0)
            canvas.close()


@pytest.mark.parametrize("case", ["depth_1", "no_map", "no_escape_term"])
def test_env_nee_is_inert_where_it_has_nothing_to_pair_with(hip_dev, case):
    """the env sample of event k pairs with the escape term of bounce k + 1: at trace depth 1, without a map (constant environment) and with the
    environment term off (the reference's behaviour, pathtracer.cu:233) the default kernel renders -- the oracle's bits"""
    from tests.util import hip_frames, oracle_frames
    sc = _scene("tiny_head", 1 if case == "depth_1" else 3)
    if case == "no_map":
        sc = dataclasses.replace(sc, env_map=None)
    if case == "no_escape_term":
        sc = dataclasses.replace(sc, env_on_escape=False)
    ref, ref_img, _ = oracle_frames(sc, 12)
    hip_dev.set_option(abi.OPT_ENV_NEE, 1)
    try:
        hdr, img, _ = hip_frames(hip_dev, sc, 12, batch=True)
    finally:
        hip_dev.set_option(abi.OPT_ENV_NEE, 0)
    assert_bit_exact(hdr, ref, f"SVR_OPT_ENV_NEE with {case}")
    assert np.array_equal(img, ref_img)
