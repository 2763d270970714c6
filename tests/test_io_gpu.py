"""GPU tests of the host-side rows N1-N4: the volume-load preprocessing kernels against the CPU restatement
(bit-exact: integers, and float32 ops without contraction), and the file-driven Canvas protocol
(LoadVolume(.mhd), SetEnvLightMap(.hdr), the TGA frame dump) end to end against the oracle renderer."""
import ctypes as C
import dataclasses
from pathlib import Path

import numpy as np
import pytest

from oracle import binding
from sunvolumerender_amd import abi, host, io, scenes
from tests.io_util import float_to_rgbe, write_hdr, write_mhd
from tests.util import assert_bit_exact

pytestmark = pytest.mark.gpu


def _preprocess(dev, elems: np.ndarray, spacing, on_device=False, hist_capacity=65536):
    a = np.ascontiguousarray(elems)
    nz, ny, nx = a.shape
    out = dev.malloc(a.size * 2)
    src = None
    try:
        if on_device:
            src = dev.malloc(a.nbytes)
            dev.to_device(src, a)
        hist = np.zeros(hist_capacity, dtype=np.uint32)
        info = abi.VolumeInfo()
        sp = (C.c_double * 3)(*[float(s) for s in spacing])
        dev.check(dev.lib.svr_volume_preprocess(C.c_void_p(src) if on_device else a.ctypes.data_as(C.c_void_p), binding.elem_type_of(a.dtype),
                                                nx, ny, nz, sp, 1 if on_device else 0, C.c_void_p(out), hist.ctypes.data_as(C.c_void_p),
                                                hist_capacity, C.byref(info)))
        u16 = dev.to_host(out, a.shape, np.uint16)
    finally:
        dev.free(out)
        if src:
            dev.free(src)
    return {"u16": u16, "range": (info.range[0], info.range[1]), "hist_bins": int(info.hist_bins),
            "hist": hist[: min(int(info.hist_bins), hist_capacity)].copy(), "maxMagnitude": float(info.maxMagnitude),
            "dim": tuple(info.dim), "spacing": tuple(info.spacing)}


def _check(got, ref, what):
    assert got["range"] == ref["range"], what
    assert got["hist_bins"] == ref["hist_bins"], what
    assert np.array_equal(got["u16"], ref["u16"]), what
    assert np.array_equal(got["hist"], ref["hist"]), what
    assert got["maxMagnitude"] == ref["maxMagnitude"], (what, got["maxMagnitude"], ref["maxMagnitude"])


@pytest.mark.parametrize("dtype", [np.int8, np.uint8, np.int16, np.uint16, np.int32, np.uint32, np.float32, np.float64])
def test_preprocess_matches_oracle(hip_dev, dtype):
    """vtkImageCast -> range -> Rescale -> vtkImageAccumulate -> vtkImageGradientMagnitude (VolumeReader.cpp:41-76)."""
    rs = np.random.RandomState(17)
    base = scenes.make_ct_head_volume(40)[3:34, 1:38, :].astype(np.float64)        # nz=31, ny=37, nx=40
    scale = {np.int8: 1 / 600, np.uint8: 1 / 300, np.int16: 0.05, np.uint16: 0.9, np.int32: 3.0, np.uint32: 2.0,
             np.float32: 0.07, np.float64: 0.31}[dtype]
    vol = base * scale + rs.standard_normal(base.shape) * 2
    if np.issubdtype(dtype, np.signedinteger) or np.issubdtype(dtype, np.floating):
        vol -= vol.mean() * 0.5
    vol = vol.astype(dtype)
    spacing = (0.7, 1.3, 2.1)
    ref = binding.io_preprocess(vol, spacing)
    got = _preprocess(hip_dev, vol, spacing)
    _check(got, ref, np.dtype(dtype).name)
    assert got["dim"] == (40, 37, 31) and got["spacing"] == tuple(np.float32(s) for s in spacing)
    _check(_preprocess(hip_dev, vol, spacing, on_device=True), ref, "device-resident input")


def test_preprocess_edge_cases(hip_dev):
    rs = np.random.RandomState(23)
    # more than 16384 bins: global-atomic histogram path; zeros are ignored, the maximum has no bin
    wide = rs.randint(-30000, 30000, size=(9, 10, 33)).astype(np.int16)
    wide[0, 0, :5] = 0
    _check(_preprocess(hip_dev, wide, (1, 1, 1)), binding.io_preprocess(wide, (1, 1, 1)), "wide range")
    # histogram capacity smaller than the bin count
    ref = binding.io_preprocess(wide, (1, 1, 1), hist_capacity=1000)
    got = _preprocess(hip_dev, wide, (1, 1, 1), hist_capacity=1000)
    assert got["hist_bins"] == ref["hist_bins"] > 1000 and np.array_equal(got["hist"], ref["hist"])
    # gradient magnitudes beyond 32767: VTK's narrowing to short wraps, the exact per-voxel path runs
    steep = np.where(rs.uniform(size=(6, 7, 8)) < 0.5, -32768, 32767).astype(np.int16)
    for sp in ((0.25, 0.25, 0.25), (1.0, 1.0, 1.0), (0.1, 3.0, 0.5)):
        _check(_preprocess(hip_dev, steep, sp), binding.io_preprocess(steep, sp), f"steep {sp}")
    # constant volume: extent 0 -> 0/0 -> defined as 0; no bins; magnitude 0
    const = np.full((3, 4, 5), 7, dtype=np.int16)
    ref = binding.io_preprocess(const, (1, 1, 1))
    got = _preprocess(hip_dev, const, (1, 1, 1))
    _check(got, ref, "constant")
    assert got["hist_bins"] == 0 and got["maxMagnitude"] == 0.0 and not got["u16"].any()
    # unsigned and wide integers wrap modulo 2^16 like a C narrowing conversion; floats truncate toward zero,
    # NaN / inf / out-of-range follow x86's cvttsd2si
    u = np.array([0, 1, 32767, 32768, 65535, 40000], dtype=np.uint16).reshape(1, 2, 3)
    _check(_preprocess(hip_dev, u, (1, 1, 1)), binding.io_preprocess(u, (1, 1, 1)), "u16 wrap")
    i = np.array([0, -1, 65536, 65537, -70000, 2**31 - 1, -2**31, 123456], dtype=np.int32).reshape(2, 2, 2)
    _check(_preprocess(hip_dev, i, (1, 1, 1)), binding.io_preprocess(i, (1, 1, 1)), "i32 wrap")
    f = np.array([0.9, -0.9, 1.5, -1.5, np.nan, np.inf, -np.inf, 1e30, -1e30, 70000.7, -32768.99, 3.0], dtype=np.float32).reshape(2, 2, 3)
    _check(_preprocess(hip_dev, f, (1, 1, 1)), binding.io_preprocess(f, (1, 1, 1)), "f32 specials")
    _check(_preprocess(hip_dev, f.astype(np.float64), (1, 1, 1)), binding.io_preprocess(f.astype(np.float64), (1, 1, 1)), "f64 specials")
    # single voxel, single row
    one = np.array([[[5]]], dtype=np.int16)
    _check(_preprocess(hip_dev, one, (1, 1, 1)), binding.io_preprocess(one, (1, 1, 1)), "1x1x1")
    row = np.arange(-3, 300, dtype=np.int16).reshape(1, 1, -1)
    _check(_preprocess(hip_dev, row, (2, 1, 1)), binding.io_preprocess(row, (2, 1, 1)), "one row")
    # bad arguments
    info = abi.VolumeInfo()
    sp = (C.c_double * 3)(1, 1, 0)
    assert hip_dev.lib.svr_volume_preprocess(one.ctypes.data_as(C.c_void_p), abi.ELEM_I16, 1, 1, 1, sp, 0, C.c_void_p(8), None, 0, C.byref(info)) != 0
    hip_dev.lib.svr_clear_error()


def _ct_like(shape=(28, 36, 44), dtype=np.int16):
    v = scenes.make_ct_head_volume(48)[: shape[0], : shape[1], : shape[2]].astype(np.float64)
    hu = (v / 65535.0) * 3000.0 - 1000.0                      # air -1000 .. bone 2000
    return hu.astype(dtype)


@pytest.mark.parametrize("style", ["raw", "local_msb", "zlib", "list_f32"])
def test_load_mhd_end_to_end(hip_dev, tmp_path, style):
    """Canvas::LoadVolume(filename) (canvas.cpp:27-41) through svr_load_mhd, rendered, against the oracle renderer
    fed with the CPU restatement of the same load."""
    spacing = (0.9, 0.9, 1.5)
    if style == "list_f32":
        vol = _ct_like(dtype=np.float32) + np.float32(0.37)
        path = write_mhd(tmp_path / "ct.mhd", vol, spacing, slices=True, header_pad=3)
    else:
        vol = _ct_like()
        kw = {"raw": {}, "local_msb": {"local": True, "msb": True}, "zlib": {"compressed": True}}[style]
        path = write_mhd(tmp_path / ("ct.mha" if style == "local_msb" else "ct.mhd"), vol, spacing, **kw)
    ref = binding.io_preprocess(vol, spacing)
    tf, mo = scenes.bone_transfer_function()
    W, H = 72, 56
    canvas = host.Canvas(hip_dev, W, H)
    try:
        canvas.SetTransferFunctionTable(tf, mo)
        canvas.LoadVolumeFile(str(path))
        vr = canvas.volumeReader
        assert vr.dim == (44, 36, 28) and vr.range == ref["range"] and vr.maxMagnitude == ref["maxMagnitude"]
        assert np.array_equal(vr.histogram, ref["hist"])
        assert vr.prep_ms > 0 and vr.prep_bytes == vol.size * (vol.dtype.itemsize + (2 if vol.dtype != np.int16 else 0) + 4)
        # the cudaVolume the loader fills is the one VolumeReader::CreateDeviceVolume builds
        sp32 = tuple(np.float32(s) for s in spacing)
        want = host.create_device_volume(canvas.deviceVolume.tex, vr.dim, sp32, ref["maxMagnitude"])
        assert bytes(canvas.deviceVolume) == bytes(want)
        sc = scenes.Scene(name="mhd", vox=ref["u16"], spacing=tuple(float(s) for s in sp32), max_magnitude=ref["maxMagnitude"],
                          tf_rgba=tf, max_opacity=mo, width=W, height=H, lights=[host.place_area_light(30.0, 40.0, 120.0, 12.0, (1, 1, 1), 900.0)],
                          trace_depth=2)
        canvas.SetAreaLights(sc.lights)
        canvas.SetScatterTimes(2)
        orc = binding.OracleScene(sc)
        hdr_ref = np.zeros((H, W, 3), dtype=np.float32)
        img_ref = np.zeros((H, W, 4), dtype=np.uint8)
        for f in range(3):
            orc.render_pathtracer(hdr_ref, f, 2, img=img_ref)
            canvas.paint()
        hip_dev.synchronize()
        assert_bit_exact(canvas.read_hdr(), hdr_ref, f"mhd {style}")
        assert np.array_equal(canvas.read_img(), img_ref)
        # the frame dump is the file the reference's stb writer produces for this image
        canvas.SaveFrame(str(tmp_path / "0.tga"))
        if binding.stb_ref() is not None:
            assert (tmp_path / "0.tga").read_bytes() == binding.ref_write_tga(tmp_path / "ref.tga", img_ref)
        # ray caster on the loaded file
        canvas.SetRenderMode(host.Canvas.RENDER_MODE_RAYCASTING)
        canvas.paint(sync=True)
        rc_ref, _ = orc.render_raycasting()
        assert np.array_equal(canvas.read_img(), rc_ref)
    finally:
        canvas.close()


def test_env_map_file(hip_dev, tmp_path):
    """Lights::SetEnvironmentLight(filename) (lights.cpp:31-75): the .hdr decoded by the library lights the scene
    exactly as the same pixels given as a table."""
    rs = np.random.RandomState(4)
    img = (rs.uniform(0, 1, (16, 32, 3)) ** 3 * 20).astype(np.float32)
    img[:, 8:20] = img[:, 8:9]
    path = write_hdr(tmp_path / "sky.hdr", float_to_rgbe(img))
    table = io.hdr_load(hip_dev, str(path))
    assert table.shape == (16, 32, 4) and np.all(table[..., 3] == 0)
    if binding.stb_ref() is not None:
        assert np.array_equal(table[..., :3], binding.ref_loadf(path))
    sc = dataclasses.replace(scenes.make_scene("tiny_head", trace_depth=2), env_map=table, env_on_escape=True, env_offset=(0.2, 0.05))
    hdr_ref = np.zeros((sc.height, sc.width, 3), dtype=np.float32)
    orc = binding.OracleScene(sc)
    for f in range(2):
        orc.render_pathtracer(hdr_ref, f, 2)
    canvas = host.Canvas(hip_dev, sc.width, sc.height)
    try:
        scenes.apply_to_canvas(dataclasses.replace(sc, env_map=None), canvas)
        canvas.SetEnvLightMap(str(path))
        assert canvas.env.intensity == 1.0 and (canvas.env.offset.x, canvas.env.offset.y) == (0.0, 0.0)   # Set(tex) resets both
        canvas.SetEnvLightIntensity(sc.env_intensity)
        canvas.SetEnvLightOffset(sc.env_offset)
        for f in range(2):
            canvas.paint()
        hip_dev.synchronize()
        assert_bit_exact(canvas.read_hdr(), hdr_ref, "env map from .hdr")
        # Lights mirror: same texture contents through the class the reference's GUI uses
        lights = io.Lights(hip_dev)
        lights.SetEnvironmentLight(str(path))
        assert lights.environmentLight.tex != 0
        lights.close()
    finally:
        hip_dev.set_option(abi.OPT_ENV_ON_ESCAPE, 0)
        canvas.close()


def test_transfer_function_class(hip_dev, tmp_path):
    """io.TransferFunction (gui/transferfunction.cpp): nodes -> table -> texture, .tf round trip, edit + re-upload."""
    from tests.test_io_cpu import GUI_COLOR, GUI_OPACITY
    tf = io.TransferFunction(hip_dev, GUI_OPACITY, GUI_COLOR)
    try:
        tex = tf.Upload()
        ref, ref_mo = binding.io_tf_table(GUI_OPACITY, GUI_COLOR)
        assert tex != 0 and tf.maxOpacity == ref_mo and np.array_equal(tf.compositeTable, ref)
        tf.SaveCurrentTFConfiguration(str(tmp_path / "a.tf"))
        tf2 = io.TransferFunction(hip_dev)
        tf2.LoadExistingTFConfiguration(str(tmp_path / "a.tf"))
        assert tf2.opacity == tf.opacity and tf2.color == tf.color
        tf.AddPoint(0.05, 0.9, 0.3, 0.4)
        assert tf.Upload() == tex                              # edited in place (texture handle kept)
        nodes = sorted(GUI_OPACITY + [(0.05, 0.9, 0.3, 0.4)])
        ref, ref_mo = binding.io_tf_table(nodes, GUI_COLOR)
        assert np.array_equal(tf.compositeTable, ref) and tf.maxOpacity == ref_mo and 0.89 < ref_mo <= 0.9
    finally:
        tf.close()


def test_cpp_canvas_example_matches_python_canvas(hip_dev, tmp_path):
    """examples/render_mhd.cpp (C++ Canvas / VolumeReader / TransferFunction / Lights of include/sunvolumerender/canvas.hpp)
    run as its own process writes the same TGA as the Python Canvas replaying the same protocol, which the oracle
    confirms."""
    import shutil
    import subprocess
    from tests.test_io_cpu import GUI_COLOR, GUI_OPACITY

    if shutil.which("g++") is None:
        pytest.skip("no g++")
    root = Path(__file__).resolve().parents[1]
    exe = tmp_path / "render_mhd"
    libdir = abi.library_path().parent
    res = subprocess.run(["g++", "-std=c++14", "-O1", f"-I{root / 'include'}", str(root / "examples" / "render_mhd.cpp"), "-o", str(exe),
                          f"-L{libdir}", "-lsvr_hip", f"-Wl,-rpath,{libdir}"], capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    vol = _ct_like()
    spacing = (0.9, 0.9, 1.5)
    mhd = write_mhd(tmp_path / "ct.mhd", vol, spacing)
    rs = np.random.RandomState(8)
    sky = write_hdr(tmp_path / "sky.hdr", float_to_rgbe((rs.uniform(0, 1, (8, 16, 3)) ** 2 * 5).astype(np.float32)))
    W, H = 64, 48
    res = subprocess.run([str(exe), str(mhd), "-env", str(sky), "-frames", "3", "-depth", "2", "-size", str(W), str(H), "-o", str(tmp_path / "cpp.tga")],
                         capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stdout + res.stderr
    res2 = subprocess.run([str(exe), str(mhd), "-raycast", "-size", str(W), str(H), "-o", str(tmp_path / "cpp_rc.tga")], capture_output=True, text=True, timeout=120)
    assert res2.returncode == 0, res2.stdout + res2.stderr

    # the same protocol through the Python mirror
    tf = io.TransferFunction(hip_dev, GUI_OPACITY, GUI_COLOR)
    canvas = host.Canvas(hip_dev, W, H)
    try:
        canvas.SetTransferFunction(tf.Upload(), tf.maxOpacity)
        canvas.LoadVolumeFile(str(mhd))
        dist = np.float32(canvas.volumeReader.GetBoundingSphereRadius()) * np.float32(1.5) + np.float32(1.0)
        light = host.make_area_light((0.0, float(dist), 0.0), (0.0, -1.0, 0.0), 10.0, (1.0, 1.0, 1.0), 500.0)
        canvas.SetAreaLights([light])
        canvas.SetEnvLightMap(str(sky))
        hip_dev.set_option(abi.OPT_ENV_ON_ESCAPE, 1)
        canvas.SetScatterTimes(2)
        for f in range(3):
            canvas.paint()
        hip_dev.synchronize()
        img = canvas.read_img()
        assert (tmp_path / "cpp.tga").read_bytes() == io.tga_encode(hip_dev, img)
        # and the oracle agrees with both
        ref = binding.io_preprocess(vol, spacing)
        sc = scenes.Scene(name="cpp", vox=ref["u16"], spacing=tuple(float(np.float32(s)) for s in spacing), max_magnitude=ref["maxMagnitude"],
                          tf_rgba=tf.compositeTable, max_opacity=tf.maxOpacity, width=W, height=H, lights=[light], trace_depth=2,
                          env_map=io.hdr_load(hip_dev, str(sky)), env_on_escape=True, env_intensity=1.0)
        orc = binding.OracleScene(sc)
        hdr_ref = np.zeros((H, W, 3), dtype=np.float32)
        img_ref = np.zeros((H, W, 4), dtype=np.uint8)
        for f in range(3):
            orc.render_pathtracer(hdr_ref, f, 2, img=img_ref)
        assert np.array_equal(img, img_ref)
        canvas.SetRenderMode(host.Canvas.RENDER_MODE_RAYCASTING)
        canvas.paint(sync=True)
        assert (tmp_path / "cpp_rc.tga").read_bytes() == io.tga_encode(hip_dev, canvas.read_img())
    finally:
        hip_dev.set_option(abi.OPT_ENV_ON_ESCAPE, 0)
        canvas.close()
        tf.close()


def test_preprocess_full_size_512(hip_dev):
    """BASELINE-size volume (512^3, MET_SHORT): the GPU preprocessing against the CPU restatement, plus
    size-independent properties of the result."""
    u = scenes.make_scene("c3").vox                                            # cached 512^3 u16 phantom
    hu = (u.astype(np.int32) * 3000 // 65535 - 1000).astype(np.int16)           # -1000 .. 2000
    spacing = (0.8, 0.8, 1.25)
    got = _preprocess(hip_dev, hu, spacing, on_device=True)
    ref = binding.io_preprocess(hu, spacing)
    _check(got, ref, "512^3")
    assert got["u16"].min() == 0 and got["u16"].max() == 65535
    lo, hi = int(hu.min()), int(hu.max())
    assert got["range"] == (lo, hi) and got["hist_bins"] == hi - lo
    # every voxel lands in exactly one bin except zeros (IgnoreZero) and the maximum (outside the extent)
    assert int(got["hist"].astype(np.int64).sum()) == hu.size - int((hu == 0).sum()) - int((hu == hi).sum())
    # rescaling is monotone
    order = np.argsort(hu.ravel()[:: 4099], kind="stable")
    assert np.all(np.diff(got["u16"].ravel()[:: 4099][order].astype(np.int64)) >= 0)
