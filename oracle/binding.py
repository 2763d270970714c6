"""ctypes binding of the CPU oracle (oracle/libsvr_oracle.so).  TEST INFRASTRUCTURE ONLY:
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never by the
sunvolumerender_amd package."""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

ORACLE_DIR = Path(__file__).resolve().parent
LIB = ORACLE_DIR / "libsvr_oracle.so"


class ovec3(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float)]


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("paths", "vol_taps", "tf_taps", "rng_draws", "woodcock_iters",
                                          "scatter_events", "shadow_walks", "raycast_steps")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


def _scene_struct(abi):
    """svo_scene, laid out with the product's ctypes PODs (they are byte-identical by contract)."""

    class Scene(C.Structure):
        _fields_ = [
            ("vol", abi.cudaVolume),
            ("tf", abi.cudaTransferFunction),
            ("cam", abi.cudaCamera),
            ("env", abi.cudaEnvironmentLight),
            ("num_lights", C.c_uint32),
            ("env_on_escape", C.c_uint32),
            ("lights", abi.cudaAreaLight * 8),
            ("vox", C.c_void_p),
            ("nx", C.c_int32), ("ny", C.c_int32), ("nz", C.c_int32),
            ("tf_n", C.c_int32),
            ("tf_rgba", C.c_void_p),
            ("env_rgba", C.c_void_p),
            ("env_w", C.c_int32), ("env_h", C.c_int32),
        ]

    return Scene


_lib = None


REFERENCE = Path("/root/reference")
STB_REF = ORACLE_DIR / "_ref" / "libstb_ref.so"


def build(force: bool = False) -> Path:
    srcs = [ORACLE_DIR / f for f in ("svr_oracle.c", "svr_oracle.h", "svr_io_oracle.c", "svr_io_oracle.h")]
    if force or not LIB.exists() or any(LIB.stat().st_mtime < f.stat().st_mtime for f in srcs):
        subprocess.run(["make", "-C", str(ORACLE_DIR), "-B", "libsvr_oracle.so"], check=True, capture_output=True)
    build_ref(force)
    build_fresnel_ref(force)
    build_wanghash_ref(force)
    return LIB


def build_ref(force: bool = False):
    """oracle/_ref/libstb_ref.so: the reference's vendored stb headers compiled where they lie.  Only possible
    where /root/reference exists (this container); the GPU box uses the prebuilt file."""
    if (REFERENCE / "utils" / "stb_image.h").exists() and (force or not STB_REF.exists()
                                                         or STB_REF.stat().st_mtime < (ORACLE_DIR / "stb_ref.c").stat().st_mtime):
        subprocess.run(["make", "-C", str(ORACLE_DIR), "-B", "_ref/libstb_ref.so"], check=True, capture_output=True)
    return STB_REF if STB_REF.exists() else None


FRESNEL_REF = ORACLE_DIR / "_ref" / "libref_fresnel.so"


def _cuda_include_dir():
    """A directory with a genuine cuda_runtime.h (the triton wheel ships NVIDIA's headers), or None."""
    try:
        import importlib.util
        spec = importlib.util.find_spec("triton")
        if spec and spec.origin:
            d = Path(spec.origin).parent / "backends" / "nvidia" / "include"
            if (d / "cuda_runtime.h").exists():
                return d
    except Exception:
        pass
    return None


def build_fresnel_ref(force: bool = False):
    """oracle/_ref/libref_fresnel.so: the reference's own schlick_fresnel (core/bsdf/fresnel.h) compiled where it lies."""
    inc = _cuda_include_dir()
    if (REFERENCE / "core" / "bsdf" / "fresnel.h").exists() and inc is not None and (
            force or not FRESNEL_REF.exists() or FRESNEL_REF.stat().st_mtime < (ORACLE_DIR / "ref_fresnel.cpp").stat().st_mtime):
        subprocess.run(["make", "-C", str(ORACLE_DIR), "-B", "_ref/libref_fresnel.so", f"CUDA_INC={inc}"], check=True, capture_output=True)
    return FRESNEL_REF if FRESNEL_REF.exists() else None


WANGHASH_REF = ORACLE_DIR / "_ref" / "libref_wanghash.so"


def build_wanghash_ref(force: bool = False):
    """oracle/_ref/libref_wanghash.so: the reference's own wangHash (pathtracer.cu:70-79), cut out of the file where it lies at
    build time (oracle/ref_wanghash.cpp, oracle/Makefile)."""
    inc = _cuda_include_dir()
    if (REFERENCE / "pathtracer.cu").exists() and inc is not None and (
            force or not WANGHASH_REF.exists() or WANGHASH_REF.stat().st_mtime < (ORACLE_DIR / "ref_wanghash.cpp").stat().st_mtime):
        subprocess.run(["make", "-C", str(ORACLE_DIR), "-B", "_ref/libref_wanghash.so", f"CUDA_INC={inc}"], check=True, capture_output=True)
    return WANGHASH_REF if WANGHASH_REF.exists() else None


def wanghash_ref():
    path = build_wanghash_ref()
    if path is None:
        return None
    lib = C.CDLL(str(path))
    lib.ref_wang_hash.restype = C.c_uint32
    lib.ref_wang_hash.argtypes = [C.c_uint32]
    return lib


def fresnel_ref():
    path = build_fresnel_ref()
    if path is None:
        return None
    lib = C.CDLL(str(path))
    lib.ref_schlick_fresnel.restype = C.c_float
    lib.ref_schlick_fresnel.argtypes = [C.c_float, C.c_float, C.c_float]
    return lib


_stb = None


def stb_ref():
    """The reference's own stb_image / stb_image_write (None if oracle/_ref was not built and cannot be)."""
    global _stb
    if _stb is None:
        path = build_ref()
        if path is None:
            return None
        lib = C.CDLL(str(path))
        lib.stbi_loadf.restype = C.POINTER(C.c_float)
        lib.stbi_loadf.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int]
        lib.stbi_image_free.restype, lib.stbi_image_free.argtypes = None, [C.c_void_p]
        lib.stbi_write_tga.restype = C.c_int
        lib.stbi_write_tga.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        lib.stbi_write_hdr.restype = C.c_int
        lib.stbi_write_hdr.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        _stb = lib
    return _stb


def ref_loadf(path: str):
    """stbi_loadf(path, &w, &h, &n, 0) of the reference's stb_image.h -> ([h][w][n] float32) or None."""
    lib = stb_ref()
    w, h, n = C.c_int(0), C.c_int(0), C.c_int(0)
    p = lib.stbi_loadf(str(path).encode(), C.byref(w), C.byref(h), C.byref(n), 0)
    if not p:
        return None
    out = np.ctypeslib.as_array(p, shape=(h.value, w.value, n.value)).copy()
    lib.stbi_image_free(p)
    return out


def ref_write_tga(path: str, img: np.ndarray) -> bytes:
    a = np.ascontiguousarray(img, dtype=np.uint8)
    ok = stb_ref().stbi_write_tga(str(path).encode(), a.shape[1], a.shape[0], a.shape[2], a.ctypes.data_as(C.c_void_p))
    assert ok
    return Path(path).read_bytes()


def ref_write_hdr(path: str, img: np.ndarray):
    a = np.ascontiguousarray(img, dtype=np.float32)
    ok = stb_ref().stbi_write_hdr(str(path).encode(), a.shape[1], a.shape[0], a.shape[2], a.ctypes.data_as(C.c_void_p))
    assert ok


# ---- svr_io_oracle.c -------------------------------------------------------------------------------
_ELEM_DTYPES = [np.int8, np.uint8, np.int16, np.uint16, np.int32, np.uint32, np.float32, np.float64]


def elem_type_of(dtype) -> int:
    return [np.dtype(d) for d in _ELEM_DTYPES].index(np.dtype(dtype))


def io_preprocess(elems: np.ndarray, spacing, hist_capacity: int = 65536) -> dict:
    """VolumeReader::Read after the file is in memory (VolumeReader.cpp:41-76), step by step on the CPU."""
    lib = load()
    a = np.ascontiguousarray(elems)
    nz, ny, nx = a.shape
    n = a.size
    shorts = np.zeros(n, dtype=np.int16)
    lib.svo_cast_to_short(a.ctypes.data_as(C.c_void_p), elem_type_of(a.dtype), C.c_size_t(n), shorts.ctypes.data_as(C.c_void_p))
    rng = (C.c_double * 2)()
    lib.svo_scalar_range(shorts.ctypes.data_as(C.c_void_p), C.c_size_t(n), rng)
    u16 = np.zeros(n, dtype=np.uint16)
    lib.svo_rescale(shorts.ctypes.data_as(C.c_void_p), C.c_size_t(n), C.c_float(rng[0]), C.c_float(rng[1]), u16.ctypes.data_as(C.c_void_p))
    hist = np.zeros(hist_capacity, dtype=np.uint32)
    bins = lib.svo_histogram(shorts.ctypes.data_as(C.c_void_p), C.c_size_t(n), C.c_double(rng[0]), C.c_double(rng[1]),
                             hist.ctypes.data_as(C.c_void_p), hist_capacity)
    sp = (C.c_double * 3)(*[float(s) for s in spacing])
    mm = lib.svo_max_gradient_magnitude(shorts.ctypes.data_as(C.c_void_p), nx, ny, nz, sp)
    return {"shorts": shorts.reshape(a.shape), "range": (float(rng[0]), float(rng[1])), "u16": u16.reshape(a.shape),
            "hist": hist[: min(bins, hist_capacity)].copy(), "hist_bins": int(bins), "maxMagnitude": float(mm)}


def io_tf_table(opacity_nodes, color_nodes, size: int = 1024):
    """TransferFunction ctor (transferfunction.cpp:17-28): (table [size][4] float32, maxOpacity)."""
    lib = load()
    o = np.ascontiguousarray(np.array(opacity_nodes, dtype=np.float64).reshape(-1, 4))
    c = np.ascontiguousarray(np.array(color_nodes, dtype=np.float64).reshape(-1, 6))
    ot, ct = np.zeros(size, dtype=np.float32), np.zeros((size, 3), dtype=np.float32)
    lib.svo_piecewise_table(o.ctypes.data_as(C.c_void_p), o.shape[0], 1, size, ot.ctypes.data_as(C.c_void_p))
    lib.svo_color_table(c.ctypes.data_as(C.c_void_p), c.shape[0], 1, size, ct.ctypes.data_as(C.c_void_p))
    table = np.concatenate([ct, ot[:, None]], axis=1).astype(np.float32)
    mo = np.float32(0)
    for v in ot:
        mo = np.float32(max(mo, v))
    return table, float(mo)


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not LIB.exists():
        build()
    lib = C.CDLL(str(LIB))
    F, U32, I = C.c_float, C.c_uint32, C.c_int
    PF = C.POINTER(C.c_float)
    lib.svo_wang_hash.restype, lib.svo_wang_hash.argtypes = U32, [U32]
    lib.svo_xorwow_init.restype, lib.svo_xorwow_init.argtypes = None, [U32, C.POINTER(U32)]
    lib.svo_xorwow_next.restype, lib.svo_xorwow_next.argtypes = U32, [C.POINTER(U32)]
    lib.svo_xorwow_uniform.restype, lib.svo_xorwow_uniform.argtypes = F, [C.POINTER(U32)]
    for name in ("svo_logf", "svo_expf", "svo_sinf", "svo_cosf", "svo_acosf"):
        getattr(lib, name).restype = F
        getattr(lib, name).argtypes = [F]
    for name in ("svo_powf", "svo_atan2f"):
        getattr(lib, name).restype = F
        getattr(lib, name).argtypes = [F, F]
    lib.svo_tex3d.restype, lib.svo_tex3d.argtypes = F, [C.c_void_p, F, F, F]
    lib.svo_tex1d.restype, lib.svo_tex1d.argtypes = None, [C.c_void_p, F, PF]
    lib.svo_tex2d.restype, lib.svo_tex2d.argtypes = None, [C.c_void_p, F, F, PF]
    lib.svo_volume_intensity.restype, lib.svo_volume_intensity.argtypes = F, [C.c_void_p, PF]
    lib.svo_volume_gradient.restype, lib.svo_volume_gradient.argtypes = None, [C.c_void_p, PF, PF]
    lib.svo_volume_intersect.restype, lib.svo_volume_intersect.argtypes = I, [C.c_void_p, PF, PF, PF, PF]
    lib.svo_camera_ray.restype, lib.svo_camera_ray.argtypes = None, [C.c_void_p, U32, U32, C.POINTER(U32), PF, PF]
    lib.svo_camera_ray_pinhole.restype, lib.svo_camera_ray_pinhole.argtypes = None, [C.c_void_p, U32, U32, PF, PF]
    lib.svo_disk_intersect.restype, lib.svo_disk_intersect.argtypes = I, [C.c_void_p, PF, PF, PF]
    lib.svo_light_radiance.restype, lib.svo_light_radiance.argtypes = None, [C.c_void_p, PF]
    lib.svo_schlick.restype, lib.svo_schlick.argtypes = F, [F, F, F]
    lib.svo_microfacet_f.restype, lib.svo_microfacet_f.argtypes = F, [PF, PF, PF, F, F]
    lib.svo_tonemap.restype, lib.svo_tonemap.argtypes = None, [PF, F, PF]
    lib.svo_onb_from_w.restype, lib.svo_onb_from_w.argtypes = None, [PF, PF, PF]
    lib.svo_render_pathtracer.restype = None
    lib.svo_render_pathtracer.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, U32, U32, I, I, I, I, C.c_void_p, I]
    lib.svo_trace_path.restype = None
    lib.svo_trace_path.argtypes = [C.c_void_p, U32, U32, U32, U32, PF, C.c_void_p]
    lib.svo_hdr_to_ldr.restype, lib.svo_hdr_to_ldr.argtypes = None, [C.c_void_p, C.c_void_p, C.c_void_p, I, I, I, I]
    lib.svo_render_raycasting.restype = None
    lib.svo_render_raycasting.argtypes = [C.c_void_p, C.c_void_p, F, I, I, I, I, C.c_void_p, I]
    lib.svo_max_threads.restype, lib.svo_max_threads.argtypes = I, []
    lib.svo_sizeof.restype, lib.svo_sizeof.argtypes = I, [I]
    VP, SZ, D = C.c_void_p, C.c_size_t, C.c_double
    lib.svo_cast_to_short.restype, lib.svo_cast_to_short.argtypes = None, [VP, I, SZ, VP]
    lib.svo_scalar_range.restype, lib.svo_scalar_range.argtypes = None, [VP, SZ, C.POINTER(D)]
    lib.svo_rescale.restype, lib.svo_rescale.argtypes = None, [VP, SZ, F, F, VP]
    lib.svo_histogram.restype, lib.svo_histogram.argtypes = I, [VP, SZ, D, D, VP, I]
    lib.svo_max_gradient_magnitude.restype, lib.svo_max_gradient_magnitude.argtypes = F, [VP, I, I, I, C.POINTER(D)]
    lib.svo_piecewise_table.restype, lib.svo_piecewise_table.argtypes = None, [VP, I, I, I, VP]
    lib.svo_color_table.restype, lib.svo_color_table.argtypes = None, [VP, I, I, I, VP]
    _lib = lib
    return lib


class OracleScene:
    """svo_scene built from a sunvolumerender_amd.scenes.Scene (plain data) -- the oracle's view of
    exactly the inputs the HIP renderer gets."""

    def __init__(self, scene):
        from sunvolumerender_amd import abi, host

        self.lib = load()
        self.scene = scene
        S = _scene_struct(abi)
        assert C.sizeof(S) == self.lib.svo_sizeof(0), (C.sizeof(S), self.lib.svo_sizeof(0))
        s = S()
        nx, ny, nz = scene.dim
        vol = host.create_device_volume(1, (nx, ny, nz), scene.spacing, scene.max_magnitude)
        vol.densityScale = scene.density_scale
        vol.gradientFactor = scene.gradient_factor
        vol.x_clip = abi.vec2(*scene.clip[0])
        vol.y_clip = abi.vec2(*scene.clip[1])
        vol.z_clip = abi.vec2(*scene.clip[2])
        s.vol = vol
        s.tf.tex = 2
        s.tf.maxOpacity = scene.max_opacity
        s.cam = scene.resolved_camera()
        env = host.env_light_constant(scene.env_radiance, scene.env_intensity)
        env.offset = abi.vec2(*scene.env_offset)
        self._vox = np.ascontiguousarray(scene.vox, dtype=np.uint16)
        self._tf = np.ascontiguousarray(scene.tf_rgba, dtype=np.float32)
        self._env = None
        if scene.env_map is not None:
            self._env = np.ascontiguousarray(scene.env_map, dtype=np.float32)
            env.tex = 3
            s.env_rgba = self._env.ctypes.data
            s.env_h, s.env_w = self._env.shape[0], self._env.shape[1]
        s.env = env
        s.num_lights = len(scene.lights)
        for i, l in enumerate(scene.lights):
            s.lights[i] = l
        s.env_on_escape = 1 if scene.env_on_escape else 0
        s.vox = self._vox.ctypes.data
        s.nx, s.ny, s.nz = nx, ny, nz
        s.tf_n = self._tf.shape[0]
        s.tf_rgba = self._tf.ctypes.data
        self.s = s
        self.W, self.H = scene.width, scene.height

    @property
    def ptr(self):
        return C.addressof(self.s)

    def new_hdr(self):
        return np.zeros((self.H, self.W, 3), dtype=np.float32)

    def render_pathtracer(self, hdr, frame_no, trace_depth=None, window=None, img=None, count=True, nthreads=0):
        x0, y0, x1, y1 = window if window is not None else (0, 0, self.W, self.H)
        c = Counters()
        td = self.scene.trace_depth if trace_depth is None else trace_depth
        self.lib.svo_render_pathtracer(self.ptr, hdr.ctypes.data, img.ctypes.data if img is not None else None,
                                       td, frame_no, x0, y0, x1, y1, C.addressof(c) if count else None, nthreads)
        return c.as_dict()

    def render_raycasting(self, step_size=None, window=None, count=True, nthreads=0):
        x0, y0, x1, y1 = window if window is not None else (0, 0, self.W, self.H)
        img = np.zeros((self.H, self.W, 4), dtype=np.uint8)
        c = Counters()
        st = self.scene.step_size() if step_size is None else step_size
        self.lib.svo_render_raycasting(self.ptr, img.ctypes.data, st, x0, y0, x1, y1, C.addressof(c) if count else None, nthreads)
        return img, c.as_dict()

    def hdr_to_ldr(self, hdr, window=None):
        x0, y0, x1, y1 = window if window is not None else (0, 0, self.W, self.H)
        img = np.zeros((self.H, self.W, 4), dtype=np.uint8)
        self.lib.svo_hdr_to_ldr(self.ptr, hdr.ctypes.data, img.ctypes.data, x0, y0, x1, y1)
        return img
