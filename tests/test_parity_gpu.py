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
]


@pytest.mark.parametrize("kernel", [abi.KERNEL_PIXEL, abi.KERNEL_PERSISTENT], ids=["pixel", "persistent"])
@pytest.mark.parametrize("layout", [abi.LAYOUT_LINEAR, abi.LAYOUT_BRICK], ids=["linear", "brick"])
@pytest.mark.parametrize("name,depth,frames", CASES)
def test_pathtracer_bit_exact(hip_dev, name, depth, frames, kernel, layout):
    sc = scenes.make_scene(name, trace_depth=depth)
    ref_hdr, ref_img, ref_c = oracle_frames(sc, frames)
    hdr, img, c = hip_frames(hip_dev, sc, frames, kernel=kernel, layout=layout)
    assert_bit_exact(hdr, ref_hdr, f"{name} depth {depth} hdr")
    assert np.array_equal(img, ref_img)
    assert c["paths"] == ref_c["paths"]
    assert c["woodcock_iters"] == ref_c["woodcock_iters"]
    assert c["scatter_events"] == ref_c["scatter_events"]
    assert c["shadow_walks"] == ref_c["shadow_walks"]
    # the persistent kernel reuses the last Woodcock tap as the scatter point's intensity
    expect_taps = ref_c["vol_taps"] - (ref_c["scatter_events"] if kernel == abi.KERNEL_PERSISTENT else 0)
    assert c["vol_taps"] == expect_taps
