"""Host-side rows N1-N4 that need no GPU: MetaImage parsing, transfer-function tables and .tf files, the .hdr
loader and the TGA writer of libsvr_hip.so -- against independent restatements (oracle/svr_io_oracle.c, numpy),
against the reference's own stb headers (oracle/_ref/libstb_ref.so, built where they lie) and against the
golden files those produced (tests/golden/io_golden.npz)."""
import ctypes as C
import struct
from pathlib import Path

import numpy as np
import pytest

from oracle import binding
from sunvolumerender_amd import abi
from tests.io_util import float_to_rgbe, write_hdr, write_mhd

GOLD = Path(__file__).parent / "golden"


@pytest.fixture(scope="module")
def lib():
    lib = abi.load()
    lib.svr_set_error_mode(0)
    lib.svr_clear_error()
    return lib


def _err(lib):
    return (lib.svr_last_error() or b"").decode()


# ---------------------------------------------------------------- N2: MetaImage files ----
@pytest.mark.parametrize("dtype", [np.int8, np.uint8, np.int16, np.uint16, np.int32, np.uint32, np.float32, np.float64])
@pytest.mark.parametrize("style", ["raw", "local", "msb", "zlib", "zlib_local", "pad", "minus1", "list"])
def test_mhd_header_and_elements(lib, tmp_path, dtype, style):
    rs = np.random.RandomState(3)
    vol = (rs.standard_normal((5, 4, 7)) * 1000).astype(dtype)
    kw = {"raw": {}, "local": {"local": True}, "msb": {"msb": True}, "zlib": {"compressed": True},
          "zlib_local": {"compressed": True, "local": True}, "pad": {"header_pad": 19},
          "minus1": {"header_size_minus_one": True}, "list": {"slices": True, "header_pad": 5}}[style]
    path = write_mhd(tmp_path / "v.mhd", vol, spacing=(0.5, 1.25, 2.0), **kw)
    h = abi.MhdHeader()
    assert lib.svr_mhd_read_header(str(path).encode(), C.byref(h)) == 0, _err(lib)
    assert (h.ndims, tuple(h.dim), tuple(h.spacing)) == (3, (7, 4, 5), (0.5, 1.25, 2.0))
    assert h.elem_type == binding.elem_type_of(dtype) and h.elem_size == np.dtype(dtype).itemsize
    assert bool(h.msb) == (style == "msb") and bool(h.compressed) == style.startswith("zlib")
    out = np.zeros(vol.shape, dtype=dtype)
    rc = lib.svr_mhd_read_elements(C.byref(h), out.ctypes.data_as(C.c_void_p), out.nbytes)
    if style == "list":
        assert rc != 0 and "LIST" in _err(lib)            # per-slice files go through svr_load_mhd
        lib.svr_clear_error()
        return
    assert rc == 0, _err(lib)
    assert np.array_equal(out.view(np.uint8), vol.view(np.uint8))


def test_mhd_header_variants_and_errors(lib, tmp_path):
    vol = np.arange(24, dtype=np.int16).reshape(2, 3, 4)
    # ElementSize is the fallback for a missing ElementSpacing; negative spacings lose their sign; unknown keys are ignored
    p = write_mhd(tmp_path / "a.mhd", vol, spacing=(-2.0, 3.0, 4.0), spacing_key="ElementSize", extra_lines=["Comment = x = y", "Modality = MET_MOD_CT"])
    h = abi.MhdHeader()
    assert lib.svr_mhd_read_header(str(p).encode(), C.byref(h)) == 0, _err(lib)
    assert tuple(h.spacing) == (2.0, 3.0, 4.0)
    # a 2-D image is one slice
    (tmp_path / "b.mhd").write_text("ObjectType = Image\nNDims = 2\nDimSize = 4 3\nElementType = MET_UCHAR\nElementDataFile = b.raw\n")
    assert lib.svr_mhd_read_header(str(tmp_path / "b.mhd").encode(), C.byref(h)) == 0, _err(lib)
    assert tuple(h.dim) == (4, 3, 1) and tuple(h.spacing) == (1.0, 1.0, 1.0)
    bad = {
        "missing.mhd": None,
        "notimage.mhd": "ObjectType = Tube\nNDims = 3\nDimSize = 1 1 1\nElementType = MET_UCHAR\nElementDataFile = x.raw\n",
        "nodata.mhd": "ObjectType = Image\nNDims = 3\nDimSize = 1 1 1\nElementType = MET_UCHAR\n",
        "badtype.mhd": "ObjectType = Image\nNDims = 3\nDimSize = 1 1 1\nElementType = MET_STRING\nElementDataFile = x.raw\n",
        "ascii.mhd": "ObjectType = Image\nNDims = 3\nBinaryData = False\nDimSize = 1 1 1\nElementType = MET_UCHAR\nElementDataFile = x.raw\n",
        "dims.mhd": "ObjectType = Image\nNDims = 4\nDimSize = 1 1 1 1\nElementType = MET_UCHAR\nElementDataFile = x.raw\n",
        "garbage.mhd": "this is not a header\n",
        "huge.mhd": "ObjectType = Image\nNDims = 3\nDimSize = 4096 4096 4096\nElementType = MET_SHORT\nElementDataFile = x.raw\n",
        "negdim.mhd": "ObjectType = Image\nNDims = 3\nDimSize = 4 -3 2\nElementType = MET_SHORT\nElementDataFile = x.raw\n",
        "pattern.mhd": "ObjectType = Image\nNDims = 3\nDimSize = 2 2 2\nElementType = MET_UCHAR\nElementDataFile = s%03d.raw 0 1 1\n",
    }
    for name, text in bad.items():
        if text is not None:
            (tmp_path / name).write_text(text)
        assert lib.svr_mhd_read_header(str(tmp_path / name).encode(), C.byref(h)) != 0, name
        assert _err(lib)
        lib.svr_clear_error()
    # short data file
    p = write_mhd(tmp_path / "short.mhd", vol)
    p.with_suffix(".raw").write_bytes(b"123")
    assert lib.svr_mhd_read_header(str(p).encode(), C.byref(h)) == 0
    out = np.zeros(vol.shape, dtype=np.int16)
    assert lib.svr_mhd_read_elements(C.byref(h), out.ctypes.data_as(C.c_void_p), out.nbytes) != 0
    lib.svr_clear_error()
    # caller's buffer too small
    p = write_mhd(tmp_path / "ok.mhd", vol)
    assert lib.svr_mhd_read_header(str(p).encode(), C.byref(h)) == 0
    assert lib.svr_mhd_read_elements(C.byref(h), out.ctypes.data_as(C.c_void_p), 5) != 0
    lib.svr_clear_error()


# ---------------------------------------------------------------- N3: transfer function ----
GUI_OPACITY = [(0.0, 0.0, 0.5, 0.0)] + [(0.1 * i, 0.5, 0.5, 0.0) for i in range(1, 11)]
GUI_COLOR = [(0.0, 69 / 255, 199 / 255, 186 / 255, 0.5, 0.0), (0.2, 172 / 255, 3 / 255, 57 / 255, 0.5, 0.0),
             (0.4, 169 / 255, 83 / 255, 58 / 255, 0.5, 0.0), (0.6, 43 / 255, 32 / 255, 161 / 255, 0.5, 0.0),
             (0.8, 247 / 255, 158 / 255, 97 / 255, 0.5, 0.0), (1.0, 183 / 255, 7 / 255, 140 / 255, 0.5, 0.0)]


def _build(lib, opacity, color, size=1024):
    o = np.ascontiguousarray(np.array(opacity, dtype=np.float64).reshape(-1, 4))
    c = np.ascontiguousarray(np.array(color, dtype=np.float64).reshape(-1, 6))
    table = np.zeros((size, 4), dtype=np.float32)
    mo = C.c_float(-1)
    rc = lib.svr_tf_build_table(o.ctypes.data_as(C.c_void_p), o.shape[0], c.ctypes.data_as(C.c_void_p), c.shape[0], size,
                                table.ctypes.data_as(C.c_void_p), C.byref(mo))
    assert rc == 0, _err(lib)
    return table, float(mo.value)


def test_tf_table_linear_nodes_match_oracle_and_interp(lib):
    table, mo = _build(lib, GUI_OPACITY, GUI_COLOR)
    ref, ref_mo = binding.io_tf_table(GUI_OPACITY, GUI_COLOR)
    assert np.array_equal(table.view(np.uint32), ref.view(np.uint32)) and mo == ref_mo == 0.5
    # midpoint 0.5 / sharpness 0 is plain piecewise-linear interpolation at x_i = i / 1023
    x = np.arange(1024, dtype=np.float64) / 1023.0
    lin = np.interp(x, [p[0] for p in GUI_OPACITY], [p[1] for p in GUI_OPACITY])
    assert np.max(np.abs(table[:, 3] - lin)) < 1e-6
    for ch in range(3):
        lin = np.interp(x, [p[0] for p in GUI_COLOR], [p[1 + ch] for p in GUI_COLOR])
        assert np.max(np.abs(table[:, ch] - lin)) < 1e-6


def test_tf_table_midpoint_sharpness_clamping(lib):
    rs = np.random.RandomState(11)
    for trial in range(40):
        n, m = rs.randint(0, 7), rs.randint(0, 7)
        ox = np.sort(rs.uniform(-0.2, 1.2, n))
        cx = np.sort(rs.uniform(-0.2, 1.2, m))
        sharp = lambda: rs.choice([0.0, 0.005, 0.3, 0.7, 0.995, 1.0])
        opacity = [(ox[i], rs.uniform(0, 1), rs.choice([0.0, 0.2, 0.5, 0.9, 1.0]), sharp()) for i in range(n)]
        color = [(cx[i], rs.uniform(0, 1), rs.uniform(0, 1), rs.uniform(0, 1), rs.choice([0.0, 0.35, 0.5, 1.0]), sharp()) for i in range(m)]
        size = int(rs.choice([1, 2, 17, 1024]))
        table, mo = _build(lib, opacity, color, size)
        ref, ref_mo = binding.io_tf_table(opacity, color, size)
        assert np.array_equal(table.view(np.uint32), ref.view(np.uint32)), trial
        assert mo == ref_mo
        assert np.all(table >= 0.0) and np.all(table <= 1.0)
    # unsorted nodes are rejected
    o = np.array([(0.5, 1, 0.5, 0), (0.2, 0, 0.5, 0)], dtype=np.float64)
    t = np.zeros((8, 4), dtype=np.float32)
    assert lib.svr_tf_build_table(o.ctypes.data_as(C.c_void_p), 2, None, 0, 8, t.ctypes.data_as(C.c_void_p), None) != 0
    lib.svr_clear_error()


def test_tf_file_format_round_trip(lib, tmp_path):
    o = np.array(GUI_OPACITY, dtype=np.float64)
    c = np.array(GUI_COLOR, dtype=np.float64)
    path = tmp_path / "gui.tf"
    assert lib.svr_tf_save(str(path).encode(), o.ctypes.data_as(C.c_void_p), len(o), c.ctypes.data_as(C.c_void_p), len(c)) == 0, _err(lib)
    # transferfunction.cpp:67-87: int n; n x 4 doubles; int m; m x 6 doubles, native endianness, no padding
    raw = path.read_bytes()
    assert len(raw) == 4 + len(o) * 32 + 4 + len(c) * 48
    assert struct.unpack_from("<i", raw, 0)[0] == len(o)
    assert np.array_equal(np.frombuffer(raw, dtype="<f8", count=len(o) * 4, offset=4).reshape(-1, 4), o)
    off = 4 + len(o) * 32
    assert struct.unpack_from("<i", raw, off)[0] == len(c)
    assert np.array_equal(np.frombuffer(raw, dtype="<f8", count=len(c) * 6, offset=off + 4).reshape(-1, 6), c)
    o2, c2 = np.zeros((32, 4)), np.zeros((32, 6))
    n, m = C.c_int(32), C.c_int(32)
    assert lib.svr_tf_load(str(path).encode(), o2.ctypes.data_as(C.c_void_p), C.byref(n), c2.ctypes.data_as(C.c_void_p), C.byref(m)) == 0, _err(lib)
    assert (n.value, m.value) == (len(o), len(c)) and np.array_equal(o2[: len(o)], o) and np.array_equal(c2[: len(c)], c)
    # too little room, truncated file, missing file
    n, m = C.c_int(3), C.c_int(32)
    assert lib.svr_tf_load(str(path).encode(), o2.ctypes.data_as(C.c_void_p), C.byref(n), c2.ctypes.data_as(C.c_void_p), C.byref(m)) != 0
    (tmp_path / "cut.tf").write_bytes(raw[: len(raw) - 9])
    n, m = C.c_int(32), C.c_int(32)
    assert lib.svr_tf_load(str(tmp_path / "cut.tf").encode(), o2.ctypes.data_as(C.c_void_p), C.byref(n), c2.ctypes.data_as(C.c_void_p), C.byref(m)) != 0
    assert lib.svr_tf_load(str(tmp_path / "nope.tf").encode(), o2.ctypes.data_as(C.c_void_p), C.byref(n), c2.ctypes.data_as(C.c_void_p), C.byref(m)) != 0
    lib.svr_clear_error()


# ---------------------------------------------------------------- N4: Radiance .hdr ----
def _hdr_load(lib, path):
    w, h = C.c_int(0), C.c_int(0)
    rc = lib.svr_hdr_load(str(path).encode(), C.byref(w), C.byref(h), None, 0)
    if rc:
        return None
    out = np.zeros((h.value, w.value, 4), dtype=np.float32)
    assert lib.svr_hdr_load(str(path).encode(), C.byref(w), C.byref(h), out.ctypes.data_as(C.c_void_p), out.size) == 0, _err(lib)
    return out


def _hdr_cases(tmp_path):
    rs = np.random.RandomState(21)
    cases = {}
    img = (rs.uniform(0, 1, (9, 33, 3)) ** 4 * 50).astype(np.float32)
    img[2:5, 4:20] = img[2, 4]                      # runs
    img[6, :, :] = 0.0                              # exponent 0 pixels
    rgbe = float_to_rgbe(img)
    cases["rle"] = write_hdr(tmp_path / "rle.hdr", rgbe, rle=True)
    cases["flat_wide"] = write_hdr(tmp_path / "flat_wide.hdr", rgbe, rle=False)          # stb's "not RLE" re-entry path
    cases["narrow"] = write_hdr(tmp_path / "narrow.hdr", float_to_rgbe(img[:, :5]), rle=True)   # width < 8: always flat
    cases["long_header"] = write_hdr(tmp_path / "long.hdr", rgbe, header_extra=("# " + "x" * 2000, "GAMMA=1"))
    return cases


def test_hdr_loader_matches_the_reference_stb(lib, tmp_path):
    """The files are decoded by the reference's own stb_image.h v2.12 (stbi_loadf, lights.cpp:34) and by
    svr_hdr_load; the float4 expansion is lights.cpp:45-53."""
    if binding.stb_ref() is None:
        pytest.skip("oracle/_ref/libstb_ref.so not built (no /root/reference here); the golden test covers this row")
    cases = _hdr_cases(tmp_path)
    # also a file written by the reference's stb_image_write (its RLE encoder)
    rs = np.random.RandomState(5)
    img = (rs.uniform(0, 4, (6, 40, 3))).astype(np.float32)
    img[:, 10:30] = img[:, 10:11]
    binding.ref_write_hdr(tmp_path / "stbw.hdr", img)
    cases["stb_written"] = tmp_path / "stbw.hdr"
    for name, path in cases.items():
        ref = binding.ref_loadf(path)
        assert ref is not None and ref.shape[2] == 3, name
        got = _hdr_load(lib, path)
        assert got is not None, (name, _err(lib))
        assert got.shape[:2] == ref.shape[:2], name
        assert np.array_equal(got[..., :3].view(np.uint32), ref.view(np.uint32)), name
        assert np.all(got[..., 3] == 0.0)
    # files stb rejects are rejected here too
    bad = {"magic.hdr": b"#?RGBE\nFORMAT=32-bit_rle_rgbe\n\n-Y 1 +X 1\n\x01\x02\x03\x80",
           "format.hdr": b"#?RADIANCE\nFORMAT=32-bit_rle_xyze\n\n-Y 1 +X 1\n\x01\x02\x03\x80",
           "layout.hdr": b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n+Y 1 +X 1\n\x01\x02\x03\x80"}
    for name, data in bad.items():
        (tmp_path / name).write_bytes(data)
        assert binding.ref_loadf(tmp_path / name) is None, name
        assert _hdr_load(lib, tmp_path / name) is None, name
        lib.svr_clear_error()


def test_hdr_and_tga_golden(lib, tmp_path):
    """Golden vectors produced by the reference's stb headers (tests/golden/make_io_golden.py)."""
    g = np.load(GOLD / "io_golden.npz")
    for name in ("rle", "flat_wide", "narrow"):
        path = tmp_path / f"{name}.hdr"
        path.write_bytes(g[f"hdr_{name}_file"].tobytes())
        got = _hdr_load(lib, path)
        assert np.array_equal(got[..., :3].view(np.uint32), g[f"hdr_{name}_rgb"].view(np.uint32)), name
    for name in ("noise", "runs", "one", "wide"):
        img = g[f"tga_{name}_img"]
        size = C.c_size_t(0)
        assert lib.svr_tga_encode(img.shape[1], img.shape[0], None, None, 0, C.byref(size)) == 0
        buf = np.zeros(size.value, dtype=np.uint8)
        assert lib.svr_tga_encode(img.shape[1], img.shape[0], img.ctypes.data_as(C.c_void_p), buf.ctypes.data_as(C.c_void_p), buf.size, C.byref(size)) == 0, _err(lib)
        assert buf[: size.value].tobytes() == g[f"tga_{name}_file"].tobytes(), name


# ---------------------------------------------------------------- N1: TGA frame dump ----
def _tga_images():
    rs = np.random.RandomState(9)
    noise = rs.randint(0, 256, (7, 13, 4)).astype(np.uint8)
    runs = np.zeros((6, 300, 4), dtype=np.uint8)
    runs[..., 3] = 255
    runs[1, 10:150] = (1, 2, 3, 255)                 # run longer than 128
    runs[2, ::2] = (9, 9, 9, 9)                      # alternating: raw packets
    runs[3, :129] = rs.randint(0, 256, (129, 4))     # raw packet longer than 128
    runs[4, 5:7] = (7, 7, 7, 7)                      # shortest run
    one = np.array([[[10, 20, 30, 40]]], dtype=np.uint8)
    wide = rs.randint(0, 3, (2, 1000, 4)).astype(np.uint8) * 100
    return {"noise": noise, "runs": runs, "one": one, "wide": wide}


def test_tga_writer_matches_the_reference_stb(lib, tmp_path):
    """Byte-for-byte the file stbi_write_tga(name, W, H, 4, data) of the reference's stb_image_write.h v1.02
    writes (canvas.cpp:102), RLE packets included."""
    if binding.stb_ref() is None:
        pytest.skip("oracle/_ref/libstb_ref.so not built; the golden test covers this row")
    for name, img in _tga_images().items():
        ref = binding.ref_write_tga(tmp_path / f"ref_{name}.tga", img)
        assert lib.svr_tga_write(str(tmp_path / f"{name}.tga").encode(), img.shape[1], img.shape[0], img.ctypes.data_as(C.c_void_p)) == 0, _err(lib)
        assert (tmp_path / f"{name}.tga").read_bytes() == ref, name
    # decode check independent of stb: header fields and bottom-up BGRA order of a 1x1 image
    raw = (tmp_path / "one.tga").read_bytes()
    assert raw[:18] == bytes([0, 0, 10, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 0, 1, 0, 32, 8])
    assert raw[18:] == bytes([0, 30, 20, 10, 40])
