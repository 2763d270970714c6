"""SVR_OPT_FAST_MATH (opt-in, default off): the tile kernel compiled with the hardware's approximate transcendentals and the
compiler's fast-math flags -- the reference's own build mode (-use_fast_math, CMakeLists.txt:9-10).  It is NOT bit-identical
to the default mode; what it must satisfy is the tolerance BASELINE.json's north star states: the converged image agrees
with the bit-exact mode within Monte-Carlo noise at a fixed seed.

Test (c2: 256^3 head, 512^2, 256 spp):  A = exact mode, frames 0..255;  B = exact mode, frames 256..511 (an independent
estimate of the same image: B = 2 * mean(0..511) - A);  F = fast mode, frames 0..255.
  * rmse(F, B) <= 1.10 * rmse(A, B)      -- the fast image is as close to an independent estimate as the exact one is
  * rmse(F, A) <= 1.00 * rmse(A, B)      -- same seeds: far closer to A than two independent renders are to each other
  * |mean(F) - mean(A)| <= 0.3 % of mean(A) per channel -- no bias
The default mode must not be affected by the switch (bit-exact again after turning it off)."""
import numpy as np
import pytest

from sunvolumerender_amd import abi, host, scenes
from tests.util import assert_bit_exact

pytestmark = pytest.mark.gpu


def test_fast_math_converged_image_within_noise(hip_dev):
    sc = scenes.make_scene("c2", trace_depth=2)
    canvas = host.Canvas(hip_dev, sc.width, sc.height)
    try:
        scenes.apply_to_canvas(sc, canvas)

        def render(fast, frames):
            hip_dev.set_option(abi.OPT_FAST_MATH, 1 if fast else 0)
            canvas.ReStartRender()
            out = []
            for n in frames:
                canvas.paint_frames(n)
                hip_dev.synchronize()
                out.append(canvas.read_hdr().astype(np.float64))
            hip_dev.set_option(abi.OPT_FAST_MATH, 0)
            return out

        A, A512 = render(False, (256, 256))
        B = 2.0 * A512 - A
        (F,) = render(True, (256,))
        (A2,) = render(False, (256,))
        assert_bit_exact(A2.astype(np.float32), A.astype(np.float32), "default mode after the fast mode was used")
        rmse = lambda x, y: float(np.sqrt(np.mean((x - y) ** 2)))
        noise = rmse(A, B)
        assert noise > 0
        assert not np.array_equal(F, A), "the fast build produced the exact build's bits: is it really the fast kernel?"
        assert rmse(F, B) <= 1.10 * noise, (rmse(F, B), noise)
        assert rmse(F, A) <= 1.00 * noise, (rmse(F, A), noise)
        mA, mF = A.mean(axis=(0, 1)), F.mean(axis=(0, 1))
        assert np.all(np.abs(mF - mA) <= 3e-3 * mA), (mA, mF)
        assert np.isfinite(F).all() and (F >= 0).all()
    finally:
        hip_dev.set_option(abi.OPT_FAST_MATH, 0)
        canvas.close()
