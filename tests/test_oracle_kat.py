"""CPU tests of the oracle's contract pieces against INDEPENDENT restatements written here in Python
(double-entry bookkeeping: the oracle is C, these are numpy / pure-Python), and against the committed
golden vectors.  No GPU needed."""
import ctypes as C
import math
from pathlib import Path

import numpy as np
import pytest

from oracle import binding
from sunvolumerender_amd import scenes

GOLD = Path(__file__).resolve().parent / "golden"
M32 = 0xFFFFFFFF


# ---------- independent Python restatements ----------
def py_wang_hash(a):
    a = ((a ^ 61) ^ (a >> 16)) & M32
    a = (a + (a << 3)) & M32
    a = a ^ (a >> 4)
    a = (a * 0x27D4EB2D) & M32
    a = a ^ (a >> 15)
    return a & M32


class PyXorwow:
    """cuRAND XORWOW as published in curand_kernel.h: scramble of curand_init(seed, 0, 0) + recurrence."""

    def __init__(self, seed):
        s0 = (seed ^ 0xAAD26B49) & M32
        s1 = 0xF7DCEFDD
        t0 = (1099087573 * s0) & M32
        t1 = (2591861531 * s1) & M32
        self.d = (6615241 + t1 + t0) & M32
        self.v = [(123456789 + t0) & M32, (362436069 ^ t0) & M32, (521288629 + t1) & M32, (88675123 ^ t1) & M32, (5783321 + t0) & M32]

    def next(self):
        v = self.v
        t = (v[0] ^ (v[0] >> 2)) & M32
        v[0], v[1], v[2], v[3] = v[1], v[2], v[3], v[4]
        v[4] = ((v[4] ^ ((v[4] << 4) & M32)) ^ (t ^ ((t << 1) & M32))) & M32
        self.d = (self.d + 362437) & M32
        return (v[4] + self.d) & M32

    def uniform(self):
        x = self.next()
        return np.float32(np.float32(x) * np.float32(2.0 ** -32) + np.float32(2.0 ** -33))


def ulp_err(got, want64):
    got = np.asarray(got, dtype=np.float64)
    want64 = np.asarray(want64, dtype=np.float64)
    ulp = np.spacing(np.abs(want64).astype(np.float32)).astype(np.float64)
    return np.abs(got - want64) / np.maximum(ulp, 1e-45)


def test_wang_hash(oracle):
    for a in list(range(100)) + [0xFFFFFFFF, 0x80000000, 123456789]:
        assert oracle.svo_wang_hash(a) == py_wang_hash(a)


def test_xorwow_stream_and_uniform(oracle):
    for seed in [0, 1, 77, 0xDEADBEEF, 0xFFFFFFFF]:
        st = (C.c_uint32 * 6)()
        oracle.svo_xorwow_init(seed, st)
        ref = PyXorwow(seed)
        for _ in range(50):
            assert oracle.svo_xorwow_next(st) == ref.next()
        oracle.svo_xorwow_init(seed, st)
        ref = PyXorwow(seed)
        for _ in range(50):
            u = oracle.svo_xorwow_uniform(st)
            assert np.float32(u) == ref.uniform()
            assert 0.0 < u <= 1.0


def test_math_accuracy_vs_float64(oracle):
    """The portable libm is within a few ulp of the correctly rounded result on the path's domains."""
    rs = np.random.RandomState(7)
    x = (np.float32(1.0) - rs.rand(4000).astype(np.float32))
    x = x[x > 0]
    got = [oracle.svo_logf(float(v)) for v in x]
    assert ulp_err(got, np.log(x.astype(np.float64))).max() <= 3.0
    x = rs.uniform(-80, 80, 4000).astype(np.float32)
    got = [oracle.svo_expf(float(v)) for v in x]
    assert ulp_err(got, np.exp(x.astype(np.float64))).max() <= 3.0
    x = rs.uniform(0, 2 * math.pi, 4000).astype(np.float32)
    s = np.array([oracle.svo_sinf(float(v)) for v in x])
    c = np.array([oracle.svo_cosf(float(v)) for v in x])
    assert np.abs(s - np.sin(x.astype(np.float64))).max() <= 2.0e-7
    assert np.abs(c - np.cos(x.astype(np.float64))).max() <= 2.0e-7
    x = rs.uniform(-1, 1, 2000).astype(np.float32)
    got = np.array([oracle.svo_acosf(float(v)) for v in x])
    assert np.abs(got - np.arccos(x.astype(np.float64))).max() <= 1.0e-6
    y, xx = rs.uniform(-3, 3, 2000).astype(np.float32), rs.uniform(-3, 3, 2000).astype(np.float32)
    got = np.array([oracle.svo_atan2f(float(a), float(b)) for a, b in zip(y, xx)])
    assert np.abs(got - np.arctan2(y.astype(np.float64), xx.astype(np.float64))).max() <= 1.0e-6
    xp, yp = rs.rand(2000).astype(np.float32) + np.float32(1e-3), rs.uniform(0.1, 30, 2000).astype(np.float32)
    got = np.array([oracle.svo_powf(float(a), float(b)) for a, b in zip(xp, yp)])
    want = np.power(xp.astype(np.float64), yp.astype(np.float64))
    assert (np.abs(got - want) / np.maximum(want, 1e-30)).max() <= 2.0e-5


def test_math_special_values(oracle):
    assert oracle.svo_logf(0.0) == -math.inf and oracle.svo_logf(1.0) == 0.0 and math.isnan(oracle.svo_logf(-1.0))
    assert oracle.svo_expf(0.0) == 1.0 and oracle.svo_expf(-200.0) == 0.0 and oracle.svo_expf(100.0) == math.inf
    assert oracle.svo_powf(0.0, 2.2) == 0.0 and oracle.svo_powf(1.0, 2.2) == 1.0 and oracle.svo_powf(3.0, 0.0) == 1.0
    assert oracle.svo_sinf(0.0) == 0.0 and oracle.svo_cosf(0.0) == 1.0


def _numpy_trilinear(vox, u, v, w):
    """tex3D<float>: border addressing, linear filter, normalized coords, float weights -- float64 reference."""
    nz, ny, nx = vox.shape

    def at(i, j, k):
        if i < 0 or j < 0 or k < 0 or i >= nx or j >= ny or k >= nz:
            return 0.0
        return float(vox[k, j, i])

    xb, yb, zb = u * nx - 0.5, v * ny - 0.5, w * nz - 0.5
    i, j, k = math.floor(xb), math.floor(yb), math.floor(zb)
    a, b, g = xb - i, yb - j, zb - k
    c = 0.0
    for dk, wk in ((0, 1 - g), (1, g)):
        for dj, wj in ((0, 1 - b), (1, b)):
            for di, wi in ((0, 1 - a), (1, a)):
                c += wk * wj * wi * at(i + di, j + dj, k + dk)
    return c / 65535.0


def test_tex3d_vs_float64_reference(oracle):
    sc = scenes.make_scene("tiny_head")
    o = binding.OracleScene(sc)
    rs = np.random.RandomState(3)
    for u, v, w in rs.uniform(-0.1, 1.1, (400, 3)).astype(np.float32):
        got = oracle.svo_tex3d(o.ptr, float(u), float(v), float(w))
        want = _numpy_trilinear(sc.vox, float(u), float(v), float(w))
        assert abs(got - want) <= 2e-6 + 2e-6 * abs(want)
    # far outside: border zeros
    assert oracle.svo_tex3d(o.ptr, -5.0, 0.5, 0.5) == 0.0 and oracle.svo_tex3d(o.ptr, 0.5, 9.0, 0.5) == 0.0
    # exactly on a texel centre returns that voxel / 65535
    n = sc.vox.shape[0]
    got = oracle.svo_tex3d(o.ptr, (20 + 0.5) / n, (21 + 0.5) / n, (22 + 0.5) / n)
    assert abs(got - float(sc.vox[22, 21, 20]) / 65535.0) <= 1e-6


def test_tex1d_clamp_and_lerp(oracle):
    sc = scenes.make_scene("tiny_head")
    o = binding.OracleScene(sc)
    t = sc.tf_rgba
    buf = (C.c_float * 4)()
    oracle.svo_tex1d(o.ptr, -3.0, buf)
    assert np.allclose(list(buf), t[0])
    oracle.svo_tex1d(o.ptr, 7.0, buf)
    assert np.allclose(list(buf), t[-1])
    n = t.shape[0]
    oracle.svo_tex1d(o.ptr, (100 + 0.5) / n, buf)
    assert np.allclose(list(buf), t[100], atol=1e-6)
    oracle.svo_tex1d(o.ptr, (100 + 1.0) / n, buf)
    assert np.allclose(list(buf), 0.5 * (t[100] + t[101]), atol=1e-6)


def test_bbox_and_disk_and_tonemap(oracle):
    sc = scenes.make_scene("tiny")
    o = binding.OracleScene(sc)
    F3 = C.c_float * 3
    tn, tf = C.c_float(), C.c_float()
    hit = oracle.svo_volume_intersect(o.ptr, F3(0, 0, 100), F3(0, 0, -1), C.byref(tn), C.byref(tf))
    assert hit == 1 and tn.value == pytest.approx(100 - 16) and tf.value == pytest.approx(100 + 16)
    assert oracle.svo_volume_intersect(o.ptr, F3(100, 0, 100), F3(0, 0, -1), C.byref(tn), C.byref(tf)) == 0
    light = sc.lights[0]
    t = C.c_float()
    c = light.disk.center
    assert oracle.svo_disk_intersect(C.byref(light.disk), F3(c.x, c.y - 5, c.z), F3(0, 1, 0), C.byref(t)) == 1
    assert t.value == pytest.approx(5.0)
    assert oracle.svo_disk_intersect(C.byref(light.disk), F3(c.x + 11, c.y - 5, c.z), F3(0, 1, 0), C.byref(t)) == 0
    rad = F3()
    oracle.svo_light_radiance(C.byref(light), rad)
    assert rad[0] == pytest.approx(500.0 * 500.0 / math.pi / (math.pi * 100.0), rel=1e-5)
    out = F3()
    oracle.svo_tonemap(F3(0.01, 0.1, 10.0), 1.0, out)
    want = [(1 - math.exp(-16 * v)) ** 2.2 for v in (0.01, 0.1, 10.0)]
    assert np.allclose(list(out), want, rtol=2e-5)


def test_schlick_and_onb(oracle):
    r0 = ((1 - 2.5) / (1 + 2.5)) ** 2
    assert oracle.svo_schlick(1.0, 2.5, 1.0) == pytest.approx(r0, rel=1e-6)
    assert oracle.svo_schlick(1.0, 2.5, 0.0) == pytest.approx(1.0, rel=1e-6)
    F3 = C.c_float * 3
    for w in ([0, 0, 1], [1, 0, 0], [0.6, 0.0, 0.8], [0.1, -0.7, 0.7071]):
        w = np.array(w, dtype=np.float64)
        w /= np.linalg.norm(w)
        u, v = F3(), F3()
        oracle.svo_onb_from_w(F3(*w), u, v)
        u, v = np.array(list(u)), np.array(list(v))
        assert abs(np.dot(u, v)) < 1e-6 and abs(np.dot(u, w)) < 1e-6 and abs(np.dot(v, w)) < 1e-6
        assert np.linalg.norm(u) == pytest.approx(1, abs=1e-5) and np.linalg.norm(v) == pytest.approx(1, abs=1e-5)


def test_camera_ray_pinhole_geometry(oracle):
    sc = scenes.make_scene("tiny")
    o = binding.OracleScene(sc)
    F3 = C.c_float * 3
    orig, d = F3(), F3()
    W, H = sc.width, sc.height
    oracle.svo_camera_ray_pinhole(o.ptr, W // 2, H // 2, orig, d)
    assert np.linalg.norm(list(d)) == pytest.approx(1, abs=1e-6)
    assert d[2] < -0.999                       # looks down -z
    oracle.svo_camera_ray_pinhole(o.ptr, 0, 0, orig, d)
    assert d[0] < 0 and d[1] < 0               # no y flip (GL origin), cuda_camera.h:87-88
    st = (C.c_uint32 * 6)()
    oracle.svo_xorwow_init(5, st)
    oracle.svo_camera_ray(o.ptr, 3, 4, st, orig, d)
    assert np.linalg.norm(list(d)) == pytest.approx(1, abs=1e-6)
    assert list(orig) == [sc.resolved_camera().pos.x, sc.resolved_camera().pos.y, sc.resolved_camera().pos.z]   # aperture 0


# ---------- golden vectors ----------
def test_golden_function_vectors(oracle):
    g = np.load(GOLD / "kat.npz")
    f1 = lambda fn, xs: np.array([fn(float(v)) for v in xs], dtype=np.float32)
    eq = lambda a, b: np.array_equal(np.asarray(a, np.float32).view(np.uint32), np.asarray(b, np.float32).view(np.uint32))
    assert eq(f1(oracle.svo_logf, g["x_log"]), g["y_log"])
    assert eq(f1(oracle.svo_expf, g["x_exp"]), g["y_exp"])
    assert eq(f1(oracle.svo_sinf, g["x_trig"]), g["y_sin"]) and eq(f1(oracle.svo_cosf, g["x_trig"]), g["y_cos"])
    assert eq(f1(oracle.svo_acosf, g["x_acos"]), g["y_acos"])
    assert eq([oracle.svo_atan2f(float(a), float(b)) for a, b in zip(g["y_at"], g["x_at"])], g["r_atan2"])
    assert eq([oracle.svo_powf(float(a), float(b)) for a, b in zip(g["xp"], g["yp"])], g["r_pow"])
    for i, sd in enumerate(g["rng_seeds"]):
        st = (C.c_uint32 * 6)()
        oracle.svo_xorwow_init(int(sd), st)
        assert [oracle.svo_xorwow_next(st) for _ in range(16)] == g["rng_seq"][i].tolist()
    assert [oracle.svo_wang_hash(int(i)) for i in g["wang_in"]] == g["wang_out"].tolist()
    sc = scenes.make_scene("tiny_head")
    o = binding.OracleScene(sc)
    assert eq([oracle.svo_tex3d(o.ptr, float(a), float(b), float(c)) for a, b, c in g["tex_uvw"]], g["tex3d"])
    buf = (C.c_float * 4)()
    for x, want in zip(g["tex1d_x"], g["tex1d"]):
        oracle.svo_tex1d(o.ptr, float(x), buf)
        assert eq(list(buf), want)
    for (a, b), want in zip(g["tex2d_uv"], g["tex2d"]):
        oracle.svo_tex2d(o.ptr, float(a), float(b), buf)
        assert eq(list(buf), want)


@pytest.mark.parametrize("name,depth,frames", [("tiny", 1, 3), ("tiny_head", 4, 2), ("tiny_bone", 6, 1)])
def test_golden_renders(oracle, name, depth, frames):
    from tests.util import assert_bit_exact, oracle_frames

    g = np.load(GOLD / f"render_{name}_d{depth}_f{frames}.npz")
    sc = scenes.make_scene(name, trace_depth=depth)
    hdr, img, cnt = oracle_frames(sc, frames)
    assert_bit_exact(hdr, g["hdr"], f"oracle vs golden {name}")
    assert np.array_equal(img, g["img"])
    assert [cnt[k] for k in g["counter_names"].tolist()] == g["counters"].tolist()
    # thread count must not matter
    hdr1, _, _ = oracle_frames(sc, frames, nthreads=1)
    assert_bit_exact(hdr1, g["hdr"], "single-threaded oracle")


def test_golden_raycast(oracle):
    g = np.load(GOLD / "raycast_tiny_head.npz")
    sc = scenes.make_scene("tiny_head")
    img, c = binding.OracleScene(sc).render_raycasting()
    assert np.array_equal(img, g["img"]) and c["raycast_steps"] == int(g["steps"])


def test_oracle_windows_compose(oracle):
    """Tiles rendered separately equal the full frame (global seeds): the basis of multi-GPU sharding."""
    from tests.util import assert_bit_exact

    sc = scenes.make_scene("tiny_head", trace_depth=2)
    o = binding.OracleScene(sc)
    full = o.new_hdr()
    o.render_pathtracer(full, 0)
    parts = o.new_hdr()
    W, H = sc.width, sc.height
    for (x0, y0, x1, y1) in [(0, 0, W // 2, H // 3), (W // 2, 0, W, H // 3), (0, H // 3, W, H)]:
        o.render_pathtracer(parts, 0, window=(x0, y0, x1, y1))
    assert_bit_exact(parts, full, "window union")


def test_schlick_fresnel_against_the_reference_itself(oracle):
    """The one function of the render path that the reference's own source yields here without stand-ins:
    schlick_fresnel (core/bsdf/fresnel.h), compiled against the genuine cuda_runtime.h of the triton wheel
    (oracle/ref_fresnel.cpp).  The oracle matches its committed outputs bit for bit -- and the live library where
    /root/reference exists."""
    from pathlib import Path
    from oracle import binding
    g = np.load(Path(__file__).parent / "golden" / "fresnel_ref.npz")
    mine = np.array([oracle.svo_schlick(float(a), float(b), float(c)) for a, b, c in zip(g["ni"], g["no"], g["cosin"])], dtype=np.float32)
    assert np.array_equal(mine.view(np.uint32), g["out"].view(np.uint32))
    ref = binding.fresnel_ref()
    if ref is not None:
        rs = np.random.RandomState(3)
        for _ in range(2000):
            a, b, c = (float(np.float32(v)) for v in (rs.uniform(0.5, 3), rs.uniform(0.5, 3), rs.uniform(-1, 1)))
            assert np.float32(ref.ref_schlick_fresnel(a, b, c)).view(np.uint32) == np.float32(oracle.svo_schlick(a, b, c)).view(np.uint32)


def test_wang_hash_against_the_reference_itself(oracle):
    """wangHash (pathtracer.cu:70-79), the seed hash of every frame, from the reference's OWN text: the line range is cut
    out of /root/reference/pathtracer.cu at build time and compiled against the genuine cuda_runtime.h
    (oracle/ref_wanghash.cpp).  The oracle, the library's host-side hash (sunvolumerender_amd.scenes.wang_hash_np uses the
    same arithmetic for the synthetic volumes) and, where /root/reference exists, the live reference library agree with the
    committed outputs."""
    from pathlib import Path
    from oracle import binding
    from sunvolumerender_amd.scenes import wang_hash_np
    g = np.load(Path(__file__).parent / "golden" / "wanghash_ref.npz")
    mine = np.array([oracle.svo_wang_hash(int(a)) for a in g["a"]], dtype=np.uint32)
    assert np.array_equal(mine, g["out"])
    assert np.array_equal(wang_hash_np(g["a"]), g["out"])
    ref = binding.wanghash_ref()
    if ref is not None:
        rs = np.random.RandomState(4)
        for a in rs.randint(0, 2 ** 32, 2000, dtype=np.uint64):
            assert ref.ref_wang_hash(int(a)) == oracle.svo_wang_hash(int(a))


def test_config_c1_cpu_plumbing():
    """BASELINE config 0 -- the reference's own CPU-runnable case: ray casting (and one path-traced frame) of the 64^3
    sphere at 256^2 through the oracle alone, no GPU.  Deterministic, image properties as expected, and the images'
    digests are pinned so an accidental change of the numeric contract shows here first."""
    import hashlib
    from oracle import binding
    from sunvolumerender_amd import scenes
    sc = scenes.make_scene("c1")
    assert sc.dim == (64, 64, 64) and (sc.width, sc.height) == (256, 256)
    o = binding.OracleScene(sc)
    img, c = o.render_raycasting()
    img2, c2 = o.render_raycasting(nthreads=1)
    assert np.array_equal(img, img2) and c == c2                       # thread count does not matter
    assert c["raycast_steps"] == 3458072 and c["vol_taps"] == 7 * c["raycast_steps"]
    assert hashlib.sha256(img.tobytes()).hexdigest() == "597dc2725cd7303bf9bc1ce5710ca14acb19f06a0e92fc5e8f9728a609c5c405"
    cover = (img[..., 3] > 0)
    assert 0.15 < cover.mean() < 0.25 and cover[128, 128] and not cover[5, 5]     # the sphere in the middle of the frame
    hdr = o.new_hdr()
    cc = o.render_pathtracer(hdr, 0)
    assert cc["paths"] == 65536 and cc["vol_taps"] == 850779
    assert hashlib.sha256(hdr.tobytes()).hexdigest() == "36cdd1149c5d5485700536b91937fc78523038d1d002eec6d632c0287a3253e4"
