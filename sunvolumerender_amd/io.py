"""Host-side classes of the reference next to the render path, over the C ABI of include/svr_io.h
(SURVEY.md section 8(f), rows N1-N4).  Same names and call order as the reference:

  VolumeReader       core/VolumeReader.{h,cpp}        Read(.mhd) -> CreateDeviceVolume(volume)
  TransferFunction   gui/transferfunction.{h,cpp}     nodes -> 1024 x RGBA table, maxOpacity, .tf save / load
  Lights             core/lights/lights.{h,cpp}       SetEnvironmentLight(.hdr), area-light list
  save_tga           gui/canvas.cpp:97-104            the "0.tga" frame dump

All parsing and all per-voxel work happen inside libsvr_hip.so; this module only moves arguments.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import abi
from .abi import cudaAreaLight, cudaEnvironmentLight, cudaVolume, vec2, vec3

HIST_CAPACITY = 65536


class VolumeReader:
    """core/VolumeReader.h:24-55.  Read() parses the MetaImage file, uploads the elements and runs the
    reference's preprocessing (cast to short, rescale to u16, histogram, max gradient magnitude) on the GPU."""

    def __init__(self, dev, layout: int = abi.LAYOUT_AUTO):
        self.dev, self.lib, self.layout = dev, dev.lib, layout
        self.histogram = np.zeros(0, dtype=np.uint32)
        self.dim = (0, 0, 0)
        self.spacing = (0.0, 0.0, 0.0)
        self.maxMagnitude = 0.0
        self.range = (0.0, 0.0)
        self._volume: Optional[cudaVolume] = None
        self.prep_ms = 0.0
        self.prep_bytes = 0

    def Read(self, filename: str):
        self.ClearDevice()
        vol, info = cudaVolume(), abi.VolumeInfo()
        hist = np.zeros(HIST_CAPACITY, dtype=np.uint32)
        self.dev.check(self.lib.svr_load_mhd(str(filename).encode(), int(self.layout), C.byref(vol), C.byref(info),
                                             hist.ctypes.data_as(C.c_void_p), HIST_CAPACITY))
        self._volume = vol
        self.dim = tuple(int(d) for d in info.dim)
        self.spacing = tuple(float(s) for s in info.spacing)
        self.maxMagnitude = float(info.maxMagnitude)
        self.range = (float(info.range[0]), float(info.range[1]))
        self.histogram = hist[: min(int(info.hist_bins), HIST_CAPACITY)].copy()
        ms, nbytes = C.c_float(0), C.c_uint64(0)
        self.lib.svr_volume_preprocess_last_ms(C.byref(ms), C.byref(nbytes))
        self.prep_ms, self.prep_bytes = float(ms.value), int(nbytes.value)

    def CreateDeviceVolume(self, volume: cudaVolume):
        """VolumeReader.cpp:174-185: bbox, spacing, texture and invMaxMagnitude; nothing else is touched."""
        if self._volume is None:
            raise RuntimeError("VolumeReader.CreateDeviceVolume before Read")
        v = self._volume
        volume.bbox, volume.spacing, volume.invSpacing = v.bbox, v.spacing, v.invSpacing
        volume.tex, volume.invMaxMagnitude = v.tex, v.invMaxMagnitude

    def GetVolumeSize(self) -> np.ndarray:
        return np.array([np.float32(d) * np.float32(s) for d, s in zip(self.dim, self.spacing)], dtype=np.float32)

    def GetBoundingSphereRadius(self) -> float:
        s = self.GetVolumeSize()
        return float(np.sqrt((s[0] * s[0] + s[1] * s[1]) + s[2] * s[2]) * np.float32(0.5))

    def GetElementBoundingSphereRadius(self) -> float:
        s = np.asarray(self.spacing, dtype=np.float32)
        return float(np.sqrt((s[0] * s[0] + s[1] * s[1]) + s[2] * s[2]) * np.float32(0.5))

    def ClearDevice(self):
        if self._volume is not None and self._volume.tex:
            self.lib.svr_destroy_texture(self._volume.tex)
        self._volume = None


class TransferFunction:
    """gui/transferfunction.h: opacity nodes (x, y, midpoint, sharpness), colour nodes (x, r, g, b, midpoint,
    sharpness); the composite 1024 x RGBA table and maxOpacity are rebuilt on every change and uploaded as a
    clamp / linear 1-D texture (transferfunction.cpp:17-44, 128-175)."""

    TABLE_SIZE = abi.TF_TABLE_SIZE

    def __init__(self, dev, opacity_points: Sequence[Sequence[float]] = (), color_points: Sequence[Sequence[float]] = ()):
        self.dev, self.lib = dev, dev.lib
        self.opacity: List[Tuple[float, float, float, float]] = []
        self.color: List[Tuple[float, float, float, float, float, float]] = []
        self.compositeTex = 0
        self.maxOpacity = 0.0
        self.compositeTable = np.zeros((self.TABLE_SIZE, 4), dtype=np.float32)
        for p in opacity_points:
            self.AddPoint(*p)
        for p in color_points:
            self.AddRGBPoint(*p)

    # vtkPiecewiseFunction::AddPoint / vtkColorTransferFunction::AddRGBPoint (nodes are kept sorted by x;
    # a node at an existing x replaces it)
    def AddPoint(self, x, y, midpoint=0.5, sharpness=0.0):
        self.opacity = sorted([p for p in self.opacity if p[0] != x] + [(float(x), float(y), float(midpoint), float(sharpness))])

    def AddRGBPoint(self, x, r, g, b, midpoint=0.5, sharpness=0.0):
        self.color = sorted([p for p in self.color if p[0] != x] + [(float(x), float(r), float(g), float(b), float(midpoint), float(sharpness))])

    def RemoveAllPoints(self):
        self.opacity, self.color = [], []

    def _nodes(self):
        o = np.ascontiguousarray(np.array(self.opacity, dtype=np.float64).reshape(-1, 4))
        c = np.ascontiguousarray(np.array(self.color, dtype=np.float64).reshape(-1, 6))
        return o, c

    def BuildTable(self) -> np.ndarray:
        o, c = self._nodes()
        mo = C.c_float(0)
        self.dev.check(self.lib.svr_tf_build_table(o.ctypes.data_as(C.c_void_p), o.shape[0], c.ctypes.data_as(C.c_void_p), c.shape[0],
                                                   self.TABLE_SIZE, self.compositeTable.ctypes.data_as(C.c_void_p), C.byref(mo)))
        self.maxOpacity = float(mo.value)
        return self.compositeTable

    def Upload(self) -> int:
        """(Re)build the table and the texture object; returns the handle (onOpacityTFChanged / onColorTFChanged)."""
        self.BuildTable()
        if self.compositeTex:
            self.dev.check(self.lib.svr_update_tf_texture(self.compositeTex, self.compositeTable.ctypes.data_as(C.c_void_p), self.TABLE_SIZE, 0))
        else:
            self.compositeTex = self.lib.svr_create_tf_texture(self.compositeTable.ctypes.data_as(C.c_void_p), self.TABLE_SIZE, 0)
            self.dev.check()
        return self.compositeTex

    def SaveCurrentTFConfiguration(self, filename: str):
        o, c = self._nodes()
        self.dev.check(self.lib.svr_tf_save(str(filename).encode(), o.ctypes.data_as(C.c_void_p), o.shape[0],
                                            c.ctypes.data_as(C.c_void_p), c.shape[0]))

    def LoadExistingTFConfiguration(self, filename: str):
        cap = 4096
        o, c = np.zeros((cap, 4), dtype=np.float64), np.zeros((cap, 6), dtype=np.float64)
        n, m = C.c_int(cap), C.c_int(cap)
        self.dev.check(self.lib.svr_tf_load(str(filename).encode(), o.ctypes.data_as(C.c_void_p), C.byref(n),
                                            c.ctypes.data_as(C.c_void_p), C.byref(m)))
        self.RemoveAllPoints()
        for i in range(n.value):
            self.AddPoint(*o[i])
        for i in range(m.value):
            self.AddRGBPoint(*c[i])

    def close(self):
        if self.compositeTex:
            self.lib.svr_destroy_texture(self.compositeTex)
            self.compositeTex = 0


class Lights:
    """core/lights/lights.{h,cpp}: the environment light and the area-light list."""

    def __init__(self, dev):
        self.dev, self.lib = dev, dev.lib
        self.environmentLight = cudaEnvironmentLight()
        self.environmentLight.defaultRadiance = vec3(0.03, 0.03, 0.03)         # lights.cpp:13
        self.environmentLight.intensity = 1.0
        self.areaLights: List[cudaAreaLight] = []
        self._env_tex = 0

    def SetEnvironmentLight(self, filename: str):
        old = self._env_tex
        self.dev.check(self.lib.svr_load_env_map(str(filename).encode(), C.byref(self.environmentLight)))
        self._env_tex = int(self.environmentLight.tex)
        if old:
            self.lib.svr_destroy_texture(old)

    def SetEnvionmentLight(self, radiance):                                     # sic, lights.cpp:77
        self.environmentLight.tex = 0
        self.environmentLight.defaultRadiance = vec3(*[float(x) for x in radiance])

    def SetEnvironmentLightIntensity(self, intensity: float):
        self.environmentLight.intensity = float(intensity)

    def SetEnvironmentLightOffset(self, offset):
        self.environmentLight.offset = vec2(float(offset[0]), float(offset[1]))

    def AddAreaLights(self, areaLight: cudaAreaLight):
        if len(self.areaLights) <= abi.MAX_LIGHT_SOURCES:                       # lights.cpp:94 (sic: <=; setup clamps to 8)
            self.areaLights.append(areaLight)

    def RemoveLights(self, idx: int):
        if self.areaLights:
            del self.areaLights[idx]

    def close(self):
        if self._env_tex:
            self.lib.svr_destroy_texture(self._env_tex)
            self._env_tex = 0


def hdr_load(dev, filename: str) -> np.ndarray:
    """stbi_loadf + the float4 expansion of lights.cpp:45-53: [h][w][4] float32, w component 0."""
    w, h = C.c_int(0), C.c_int(0)
    dev.check(dev.lib.svr_hdr_load(str(filename).encode(), C.byref(w), C.byref(h), None, 0))
    out = np.zeros((h.value, w.value, 4), dtype=np.float32)
    dev.check(dev.lib.svr_hdr_load(str(filename).encode(), C.byref(w), C.byref(h), out.ctypes.data_as(C.c_void_p), out.size))
    return out


def tga_encode(dev, img: np.ndarray) -> bytes:
    a = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = a.shape[0], a.shape[1]
    size = C.c_size_t(0)
    dev.check(dev.lib.svr_tga_encode(w, h, None, None, 0, C.byref(size)))
    buf = np.zeros(size.value, dtype=np.uint8)
    dev.check(dev.lib.svr_tga_encode(w, h, a.ctypes.data_as(C.c_void_p), buf.ctypes.data_as(C.c_void_p), buf.size, C.byref(size)))
    return buf[: size.value].tobytes()


def save_tga(dev, filename: str, img: np.ndarray):
    a = np.ascontiguousarray(img, dtype=np.uint8)
    dev.check(dev.lib.svr_tga_write(str(filename).encode(), a.shape[1], a.shape[0], a.ctypes.data_as(C.c_void_p)))
