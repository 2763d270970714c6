"""Function-level known-answer tests ON THE DEVICE (svr_selftest_math): the device functions the render kernels call,
evaluated on the committed golden vectors -- no oracle in the loop, tolerance 0.

* schlick_fresnel against tests/golden/fresnel_ref.npz, whose outputs came from the reference's OWN
  core/bsdf/fresnel.h compiled where it lies (oracle/ref_fresnel.cpp): the one function of the path that is pinned
  to the reference's code, now pinned on the GPU too.
* wangHash against tests/golden/wanghash_ref.npz, likewise from the reference's own text (pathtracer.cu:70-79,
  oracle/ref_wanghash.cpp).
* libm (logf/expf/sinf/cosf/acosf/atan2f/powf), XORWOW uniforms and wangHash against tests/golden/kat.npz.
"""
import ctypes as C
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = Path(__file__).resolve().parent / "golden"

FN_SCHLICK, FN_LOG, FN_EXP, FN_SIN, FN_COS, FN_ACOS, FN_ATAN2, FN_POW, FN_UNIFORM, FN_WANG, FN_LOG_UNIT = range(11)


def _eval(dev, fn, *cols):
    a = np.ascontiguousarray(np.stack([np.asarray(c, dtype=np.float32) for c in cols], axis=1))
    out = np.zeros(a.shape[0], dtype=np.float32)
    dev.check(dev.lib.svr_selftest_math(fn, a.ctypes.data_as(C.c_void_p), a.shape[1], out.ctypes.data_as(C.c_void_p), a.shape[0]))
    return out


def _same_bits(got, want, what):
    g, w = got.view(np.uint32), np.ascontiguousarray(want, dtype=np.float32).view(np.uint32)
    nan = np.isnan(got) & np.isnan(want)
    bad = (g != w) & ~nan
    assert not bad.any(), f"{what}: {int(bad.sum())} of {bad.size} differ; first {np.argwhere(bad)[:3].ravel().tolist()}"


def test_device_schlick_fresnel_matches_the_references_own_code(hip_dev):
    z = np.load(GOLD / "fresnel_ref.npz")
    _same_bits(_eval(hip_dev, FN_SCHLICK, z["ni"], z["no"], z["cosin"]), z["out"], "schlick_fresnel vs core/bsdf/fresnel.h")


def test_device_libm_matches_golden(hip_dev):
    k = np.load(GOLD / "kat.npz")
    _same_bits(_eval(hip_dev, FN_LOG, k["x_log"]), k["y_log"], "logf")
    _same_bits(_eval(hip_dev, FN_EXP, k["x_exp"]), k["y_exp"], "expf")
    _same_bits(_eval(hip_dev, FN_SIN, k["x_trig"]), k["y_sin"], "sinf")
    _same_bits(_eval(hip_dev, FN_COS, k["x_trig"]), k["y_cos"], "cosf")
    _same_bits(_eval(hip_dev, FN_ACOS, k["x_acos"]), k["y_acos"], "acosf")
    _same_bits(_eval(hip_dev, FN_ATAN2, k["y_at"], k["x_at"]), k["r_atan2"], "atan2f")
    _same_bits(_eval(hip_dev, FN_POW, k["xp"], k["yp"]), k["r_pow"], "powf")
    # the Woodcock walk's log(1 - u) shortcut equals logf on its domain [0, 1)
    x = k["x_log"][(k["x_log"] >= 0) & (k["x_log"] < 1)]
    x = x[(x == 0) | (x >= np.float32(2.0 ** -126))]
    _same_bits(_eval(hip_dev, FN_LOG_UNIT, x), _eval(hip_dev, FN_LOG, x), "logf_unit == logf on [0, 1)")


def test_device_rng_and_hash_match_golden(hip_dev):
    k = np.load(GOLD / "kat.npz")
    seeds, uni = k["rng_seeds"], k["rng_uniform"]
    s = np.repeat(seeds, uni.shape[1]).astype(np.uint32).view(np.float32)
    j = np.tile(np.arange(uni.shape[1], dtype=np.float32), len(seeds))
    _same_bits(_eval(hip_dev, FN_UNIFORM, s, j), uni.ravel(), "curand_uniform sequence")
    got = _eval(hip_dev, FN_WANG, k["wang_in"].view(np.float32)).view(np.uint32)
    assert np.array_equal(got, k["wang_out"])


def test_device_wang_hash_matches_the_references_own_code(hip_dev):
    """tests/golden/wanghash_ref.npz came from the reference's own wangHash (pathtracer.cu:70-79, oracle/ref_wanghash.cpp)."""
    z = np.load(GOLD / "wanghash_ref.npz")
    got = _eval(hip_dev, FN_WANG, z["a"].view(np.float32)).view(np.uint32)
    assert np.array_equal(got, z["out"])
