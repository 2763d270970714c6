"""GPU parity: libsvr_hip.so (through the C ABI) against the CPU oracle, same seeded inputs.

Bar: BIT-EXACT float32 radiance (tolerance 0) -- the numeric contract makes a path's arithmetic a
pure function of (scene, pixel, frame), independent of lane scheduling -- and bit-exact RGBA8.
"""
import numpy as np
import pytest

from sunvolumerender_amd import abi, scenes
from tests.util import assert_bit_exact, hip_frames, oracle_frames

pytestmark = pytest.mark.gpu

CASES = [
    ("tiny", 1, 2),
    ("tiny_head", 1, 2),
    ("tiny_head", 4, 2),
    ("tiny_bone", 6, 1),
    ("tiny_head_noisy", 2, 2),        # non-zero air: no macro-cell is exactly transparent
]


KERNELS = [
    (abi.KERNEL_PIXEL, True, "pixel"),
    (abi.KERNEL_TILE, True, "tile"),
    (abi.KERNEL_TILE, False, "tile_noskip"),
    (abi.KERNEL_ULOOP, True, "uloop"),
    (abi.KERNEL_WAVEFRONT, True, "wavefront"),
    (abi.KERNEL_WAVEFRONT, False, "wavefront_noskip"),
]


@pytest.mark.parametrize("kernel,skip,kid", KERNELS, ids=[k[2] for k in KERNELS])
@pytest.mark.parametrize("layout", [abi.LAYOUT_LINEAR, abi.LAYOUT_BRICK, abi.LAYOUT_PAIR, abi.LAYOUT_CELL], ids=["linear", "brick", "pair", "cell"])
@pytest.mark.parametrize("name,depth,frames", CASES)
def test_pathtracer_bit_exact(hip_dev, name, depth, frames, kernel, skip, kid, layout):
    sc = scenes.make_scene(name, trace_depth=depth)
    ref_hdr, ref_img, ref_c = oracle_frames(sc, frames)
    hdr, img, c = hip_frames(hip_dev, sc, frames, kernel=kernel, layout=layout, empty_skip=skip)
    assert_bit_exact(hdr, ref_hdr, f"{name} depth {depth} hdr")
    assert np.array_equal(img, ref_img)
    assert c["paths"] == ref_c["paths"]
    assert c["woodcock_iters"] == ref_c["woodcock_iters"]
    assert c["scatter_events"] == ref_c["scatter_events"]
    assert c["shadow_walks"] == ref_c["shadow_walks"]
    # algorithmic taps = what the reference issues; executed taps are fewer (reused scatter tap, skipping)
    assert c["vol_taps"] == ref_c["vol_taps"]
    assert c["vol_taps_executed"] <= c["vol_taps"]
    if kernel in (abi.KERNEL_TILE, abi.KERNEL_WAVEFRONT) and skip and not name.endswith("_noisy"):
        assert c["vol_taps_executed"] < c["vol_taps"] - ref_c["scatter_events"]


@pytest.mark.parametrize("LAYOUT", [abi.LAYOUT_PAIR, abi.LAYOUT_CELL], ids=["pair", "cell"])
@pytest.mark.parametrize("name,depth,frames", CASES + [("small_head", 2, 9)])
def test_pair_layout_bit_exact(hip_dev, name, depth, frames, LAYOUT):
    """LAYOUT_PAIR (32-bit elements holding voxel x and x + 1: 4 gathers per fetch instead of 8) and LAYOUT_CELL (16-byte
    elements holding the 8 voxels of a trilinear cell: one load per fetch) feed the filter the same eight voxels: tile kernel
    (per-frame calls and one folding launch, queue machine on and off) and ray caster."""
    sc = scenes.make_scene(name, trace_depth=depth)
    ref_hdr, ref_img, ref_c = oracle_frames(sc, frames)
    hdr, img, c = hip_frames(hip_dev, sc, frames, kernel=abi.KERNEL_TILE, layout=LAYOUT)
    assert_bit_exact(hdr, ref_hdr, f"{name} layout {LAYOUT}")
    assert np.array_equal(img, ref_img)
    assert c["vol_taps"] == ref_c["vol_taps"] and c["woodcock_iters"] == ref_c["woodcock_iters"]
    for q in (0, 2):
        hip_dev.set_option(abi.OPT_QUEUE, q)
        b_hdr, _, _ = hip_frames(hip_dev, sc, frames, kernel=abi.KERNEL_TILE, layout=LAYOUT, batch=True)
        hip_dev.set_option(abi.OPT_QUEUE, 1)
        assert_bit_exact(b_hdr, ref_hdr, f"{name} layout {LAYOUT}, one launch, queue={q}")
    from oracle import binding
    from sunvolumerender_amd import host
    ref_rc, _ = binding.OracleScene(sc).render_raycasting()
    canvas = host.Canvas(hip_dev, sc.width, sc.height)
    try:
        scenes.apply_to_canvas(sc, canvas, LAYOUT)
        canvas.SetRenderMode(host.Canvas.RENDER_MODE_RAYCASTING)
        canvas.paint(sync=True)
        assert np.array_equal(canvas.read_img(), ref_rc)
    finally:
        canvas.close()


def test_batch_equals_sequential(hip_dev):
    """svr_render_pathtracer_frames(n) is bit-identical to n render_pathtracer calls (one group of 32 + remainder;
    lanes of a wave = pixels x frames in the batch form)."""
    sc = scenes.make_scene("tiny_head", trace_depth=2)
    n = 37
    a_hdr, a_img, _ = hip_frames(hip_dev, sc, n, batch=False)
    b_hdr, b_img, _ = hip_frames(hip_dev, sc, n, batch=True)
    c_hdr, c_img, _ = hip_frames(hip_dev, sc, n, batch=True, pipeline=False)
    assert_bit_exact(a_hdr, b_hdr, "batch vs sequential")
    assert_bit_exact(a_hdr, c_hdr, "pipelined vs single stream")
    assert np.array_equal(a_img, b_img) and np.array_equal(a_img, c_img)
    ref_hdr, ref_img, _ = oracle_frames(sc, n)
    assert_bit_exact(a_hdr, ref_hdr, f"{n} frames vs oracle")
    assert np.array_equal(a_img, ref_img)
