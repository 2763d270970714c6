"""ctypes binding of libsvr_hip.so -- exactly the C ABI declared in include/svr_abi.h.

There is no fallback: if the library has not been built, `load()` raises.  Nothing in
this module (or anywhere in the package) imports or links the CPU oracle.
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path

from ._build import LIB_PATH

# ---------------------------------------------------------------------------
# POD layouts of the reference host API; names follow the reference
# ---------------------------------------------------------------------------


class vec2(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float)]

    def __init__(self, x=0.0, y=0.0):
        super().__init__(float(x), float(y))

    def tuple(self):
        return (self.x, self.y)


class vec3(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float)]

    def __init__(self, x=0.0, y=0.0, z=0.0):
        super().__init__(float(x), float(y), float(z))

    def tuple(self):
        return (self.x, self.y, self.z)


class cudaBBox(C.Structure):  # core/geometry/cuda_bbox.h:66-69
    _fields_ = [("vmin", vec3), ("vmax", vec3), ("invSize", vec3)]


class cudaVolume(C.Structure):  # core/cuda_volume.h:111-121
    _fields_ = [
        ("bbox", cudaBBox),
        ("_pad0", C.c_uint32),
        ("tex", C.c_uint64),
        ("densityScale", C.c_float),
        ("invMaxMagnitude", C.c_float),
        ("gradientFactor", C.c_float),
        ("spacing", vec3),
        ("invSpacing", vec3),
        ("x_clip", vec2),
        ("y_clip", vec2),
        ("z_clip", vec2),
        ("_pad1", C.c_uint32),
    ]


class cudaTransferFunction(C.Structure):  # core/cuda_transfer_function.h:57-59
    _fields_ = [("tex", C.c_uint64), ("maxOpacity", C.c_float), ("_pad", C.c_uint32)]


class cudaCamera(C.Structure):  # core/cuda_camera.h:98-106
    _fields_ = [
        ("imageW", C.c_uint32),
        ("imageH", C.c_uint32),
        ("exposure", C.c_float),
        ("apeture", C.c_float),
        ("focalLength", C.c_float),
        ("aspectRatio", C.c_float),
        ("tanFovxOverTwo", C.c_float),
        ("pos", vec3),
        ("u", vec3),
        ("v", vec3),
        ("w", vec3),
    ]


class cudaDisk(C.Structure):  # core/geometry/cuda_disk.h:58-61
    _fields_ = [("radius", C.c_float), ("center", vec3), ("normal", vec3)]


class cudaAreaLight(C.Structure):  # core/lights/cuda_arealight.h:68-71
    _fields_ = [("disk", cudaDisk), ("color", vec3), ("intensity", C.c_float)]


class cudaEnvironmentLight(C.Structure):  # core/lights/cuda_environment_light.h:74-78
    _fields_ = [("tex", C.c_uint64), ("defaultRadiance", vec3), ("intensity", C.c_float), ("offset", vec2)]


class RenderParams(C.Structure):  # core/render_parameters.h:34-37
    _fields_ = [("traceDepth", C.c_uint32), ("frameNo", C.c_uint32), ("hdrBuffer", C.c_void_p)]


class Counters(C.Structure):  # svr_counters
    _fields_ = [
        ("paths", C.c_uint64),
        ("vol_taps", C.c_uint64),
        ("woodcock_iters", C.c_uint64),
        ("scatter_events", C.c_uint64),
        ("shadow_walks", C.c_uint64),
        ("raycast_steps", C.c_uint64),
        ("loop_iters", C.c_uint64),
        ("vol_taps_executed", C.c_uint64),
        ("walks_ray_skipped", C.c_uint64),
        ("iters_ray_skipped", C.c_uint64),
        ("iters_prefix_skipped", C.c_uint64),
        ("taps_bound_culled", C.c_uint64),
    ]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_ }


EXPECTED_SIZES = {
    vec3: 12,
    cudaBBox: 36,
    cudaVolume: 112,
    cudaTransferFunction: 16,
    cudaCamera: 76,
    cudaDisk: 28,
    cudaAreaLight: 44,
    cudaEnvironmentLight: 32,
    RenderParams: 16,
    Counters: 96,
}
for _t, _n in EXPECTED_SIZES.items():
    assert C.sizeof(_t) == _n, (_t, C.sizeof(_t), _n)

MAX_LIGHT_SOURCES = 8
TF_TABLE_SIZE = 1024

LAYOUT_AUTO, LAYOUT_LINEAR, LAYOUT_BRICK, LAYOUT_PAIR, LAYOUT_CELL = 0, 1, 2, 3, 4
OPT_ENV_ON_ESCAPE, OPT_KERNEL, OPT_COUNT, OPT_TIMING, OPT_SKIP_TONEMAP, OPT_BLOCKS_PER_CU = 1, 2, 3, 4, 5, 6
OPT_PIPELINE, OPT_REFILL_MIN_IDLE, OPT_EMPTY_SKIP, OPT_RAY_SKIP, OPT_FRAMES_PER_WAVE_LOG2 = 7, 8, 9, 10, 11
OPT_RAYCAST_LANES_LOG2 = 12
OPT_FRAME_AHEAD = 13
OPT_FAST_MATH = 14
OPT_BOUND_CULL = 15
OPT_FOLD = 17
OPT_QUEUE = 18
OPT_PARK_END = 19
OPT_FINE_MASK = 20
OPT_ROW_ORDER = 21
OPT_GROUP_FRAMES = 22
OPT_LOCAL_MAJORANT = 23
OPT_LIGHT_CULL = 24
OPT_LM_TUNE = 25
OPT_LM_SUBCELLS = 26
OPT_PARK_CHEAP = 27
OPT_PINHOLE_FAST = 28
OPT_POOL = 29
OPT_TRIPS = 30
OPT_NAN_GUARD = 31
OPT_MACRO_SHIFT_MIN = 32
OPT_SPLIT = 33
OPT_ENV_NEE = 34
OPT_FAST_BOUND = 35
KERNEL_AUTO, KERNEL_PIXEL, KERNEL_TILE, KERNEL_ULOOP, KERNEL_WAVEFRONT = 0, 1, 2, 3, 4

ELEM_I8, ELEM_U8, ELEM_I16, ELEM_U16, ELEM_I32, ELEM_U32, ELEM_F32, ELEM_F64 = range(8)


class MhdHeader(C.Structure):          # svr_mhd_header, include/svr_io.h
    _fields_ = [
        ("ndims", C.c_int32), ("dim", C.c_int32 * 3), ("spacing", C.c_double * 3),
        ("elem_type", C.c_int32), ("elem_size", C.c_int32), ("channels", C.c_int32), ("msb", C.c_int32),
        ("compressed", C.c_int32), ("compressed_size", C.c_int64), ("header_size", C.c_int64),
        ("data_offset", C.c_int64), ("data_file", C.c_char * 1024),
    ]


class VolumeInfo(C.Structure):         # svr_volume_info, include/svr_io.h
    _fields_ = [
        ("dim", C.c_int32 * 3), ("spacing", C.c_float * 3), ("range", C.c_double * 2),
        ("maxMagnitude", C.c_float), ("hist_bins", C.c_uint32),
    ]


# every symbol include/svr_abi.h and include/svr_io.h declare: name -> (restype, argtypes)
_P = C.POINTER
PROTOTYPES = {
    # (A) the reference's entry points
    "render_pathtracer": (None, [C.c_void_p, _P(RenderParams)]),
    "setup_volume": (None, [_P(cudaVolume)]),
    "setup_transferfunction": (None, [_P(cudaTransferFunction)]),
    "setup_camera": (None, [_P(cudaCamera)]),
    "setup_env_lights": (None, [_P(cudaEnvironmentLight)]),
    "setup_area_lights": (None, [_P(cudaAreaLight), C.c_uint32]),
    "render_raycasting": (None, [C.c_void_p, _P(cudaVolume), _P(cudaTransferFunction), _P(cudaCamera), C.c_float]),
    # (B) helpers
    "svr_init": (C.c_int, [C.c_int]),
    "svr_shutdown": (None, []),
    "svr_set_stream": (C.c_int, [C.c_void_p]),
    "svr_device_synchronize": (C.c_int, []),
    "svr_set_error_mode": (None, [C.c_int]),
    "svr_last_error": (C.c_char_p, []),
    "svr_last_error_code": (C.c_int, []),
    "svr_clear_error": (None, []),
    "svr_create_volume_texture": (C.c_uint64, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "svr_create_tf_texture": (C.c_uint64, [C.c_void_p, C.c_int, C.c_int]),
    "svr_update_tf_texture": (C.c_int, [C.c_uint64, C.c_void_p, C.c_int, C.c_int]),
    "svr_create_env_texture": (C.c_uint64, [C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "svr_destroy_texture": (C.c_int, [C.c_uint64]),
    "svr_render_params_setup_hdr": (C.c_int, [_P(RenderParams), C.c_uint32, C.c_uint32]),
    "svr_render_params_clear": (C.c_int, [_P(RenderParams)]),
    "svr_device_malloc": (C.c_void_p, [C.c_size_t]),
    "svr_device_free": (C.c_int, [C.c_void_p]),
    "svr_memcpy_h2d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "svr_memcpy_d2h": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "svr_memset_device": (C.c_int, [C.c_void_p, C.c_int, C.c_size_t]),
    "svr_strip_rows_owned": (C.c_uint32, [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]),
    "svr_strip_row_to_y": (C.c_uint32, [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]),
    "svr_pack_strips": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]),
    "svr_unpack_strips": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]),
    "svr_assemble_frame": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]),
    "svr_set_row_shard": (C.c_int, [C.c_uint32, C.c_uint32, C.c_uint32]),
    "svr_set_render_window": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "svr_set_option": (C.c_int, [C.c_int, C.c_int]),
    "svr_get_option": (C.c_int, [C.c_int]),
    "svr_render_pathtracer_frames": (C.c_int, [C.c_void_p, _P(RenderParams), C.c_uint32]),
    "svr_hdr_to_ldr": (C.c_int, [C.c_void_p, _P(RenderParams)]),
    "svr_hdr_to_ldr_frame": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32]),
    "svr_selftest_chain": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32]),
    "svr_selftest_math": (C.c_int, [C.c_int, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32]),
    "svr_selftest_bound8": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p]),
    "svr_get_counters": (C.c_int, [_P(Counters)]),
    "svr_reset_counters": (C.c_int, []),
    "svr_get_kernel_time": (C.c_int, [_P(C.c_double), _P(C.c_uint64)]),
    "svr_reset_kernel_time": (C.c_int, []),
    "svr_device_info": (C.c_char_p, []),
    "svr_abi_version": (C.c_int, []),
    # include/svr_io.h: the host-side rows N1-N4
    "svr_mhd_read_header": (C.c_int, [C.c_char_p, _P(MhdHeader)]),
    "svr_mhd_read_elements": (C.c_int, [_P(MhdHeader), C.c_void_p, C.c_size_t]),
    "svr_volume_preprocess": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, _P(C.c_double), C.c_int, C.c_void_p,
                                        C.c_void_p, C.c_uint32, _P(VolumeInfo)]),
    "svr_load_mhd": (C.c_int, [C.c_char_p, C.c_int, _P(cudaVolume), _P(VolumeInfo), C.c_void_p, C.c_uint32]),
    "svr_volume_preprocess_last_ms": (C.c_int, [_P(C.c_float), _P(C.c_uint64)]),
    "svr_tf_build_table": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, _P(C.c_float)]),
    "svr_tf_save": (C.c_int, [C.c_char_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int]),
    "svr_tf_load": (C.c_int, [C.c_char_p, C.c_void_p, _P(C.c_int), C.c_void_p, _P(C.c_int)]),
    "svr_hdr_load": (C.c_int, [C.c_char_p, _P(C.c_int), _P(C.c_int), C.c_void_p, C.c_size_t]),
    "svr_load_env_map": (C.c_int, [C.c_char_p, _P(cudaEnvironmentLight)]),
    "svr_tga_write": (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.c_void_p]),
    "svr_tga_encode": (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, _P(C.c_size_t)]),
}

_lib = None


def library_path() -> Path:
    return LIB_PATH


def load() -> C.CDLL:
    """Load libsvr_hip.so and bind every prototype.  Raises if the library is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -m sunvolumerender_amd._build` "
            "(or __graft_entry__.build()).  There is no CPU fallback."
        )
    lib = C.CDLL(str(LIB_PATH))
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
