"""Host side of the reference's render API, above the C ABI.

The reference's host code is C++ (Qt `Canvas`, gui/canvas.{h,cpp}); this module mirrors the
parts of it that drive the render path -- same names, argument meaning and call protocol --
so tests and the benchmark read like the reference's own host code:

  cudaCamera.Setup            core/cuda_camera.h:35-63      -> camera_setup / camera_setup_uvw
  cudaBBox.Set                core/geometry/cuda_bbox.h:25-31 -> bbox_set
  cudaVolume.Set/...          core/cuda_volume.h:18-37      -> volume_set
  VolumeReader::CreateDeviceVolume  core/VolumeReader.cpp:174-201
  Canvas (LoadVolume, setters, paintGL, ReStartRender)  gui/canvas.cpp:8-117, gui/canvas.h:43-175

All arithmetic that the reference does in `float` is done in numpy float32 here.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Optional, Sequence

import numpy as np

from . import abi
from .abi import (RenderParams, cudaAreaLight, cudaBBox, cudaCamera, cudaDisk, cudaEnvironmentLight,
                  cudaTransferFunction, cudaVolume, vec2, vec3)

f32 = np.float32


def _v(a) -> np.ndarray:
    return np.asarray(a, dtype=np.float32).reshape(3)


def _to_vec3(a) -> vec3:
    a = _v(a)
    return vec3(a[0], a[1], a[2])


def _dot(a, b) -> np.float32:
    return f32(f32(f32(a[0] * b[0]) + f32(a[1] * b[1])) + f32(a[2] * b[2]))


def _normalize(a) -> np.ndarray:
    a = _v(a)
    s = f32(1.0) / np.sqrt(_dot(a, a), dtype=np.float32)
    return (a * s).astype(np.float32)


def _cross(x, y) -> np.ndarray:
    x, y = _v(x), _v(y)
    return np.array([f32(x[1] * y[2]) - f32(y[1] * x[2]),
                     f32(x[2] * y[0]) - f32(y[2] * x[0]),
                     f32(x[0] * y[1]) - f32(y[0] * x[1])], dtype=np.float32)


def _tan_fov_over_two(fovx: float) -> float:
    # cuda_camera.h:44: tanf(fovx * 0.5f * M_PI / 180.f) -- float * double / float -> double -> tanf(float)
    arg = f32(float(f32(fovx) * f32(0.5)) * math.pi / 180.0)
    return float(np.tan(arg, dtype=np.float32))


def camera_setup(pos, target, up, fovx=45.0, apeture=0.0, focalLength=1.0, exposure=1.0,
                 imageW=640, imageH=640) -> cudaCamera:
    """cudaCamera::Setup(pos, target, up, ...), core/cuda_camera.h:50-63."""
    pos, target, up = _v(pos), _v(target), _v(up)
    w = _normalize(pos - target)
    u = _cross(up, w)
    v = _cross(w, u)
    return camera_setup_uvw(pos, u, v, w, fovx, apeture, focalLength, exposure, imageW, imageH)


def camera_setup_uvw(pos, u, v, w, fovx=45.0, apeture=0.0, focalLength=1.0, exposure=1.0,
                     imageW=640, imageH=640) -> cudaCamera:
    """cudaCamera::Setup(pos, u, v, w, ...), core/cuda_camera.h:35-48."""
    cam = cudaCamera()
    cam.pos, cam.u, cam.v, cam.w = _to_vec3(pos), _to_vec3(u), _to_vec3(v), _to_vec3(w)
    cam.imageW, cam.imageH = int(imageW), int(imageH)
    cam.aspectRatio = float(f32(imageW) / f32(imageH))
    cam.tanFovxOverTwo = _tan_fov_over_two(fovx)
    cam.exposure, cam.focalLength, cam.apeture = float(exposure), float(focalLength), float(apeture)
    return cam


def zoom_to_extent_eye_dist(volume_size, fov=45.0) -> float:
    """Canvas::ZoomToExtent, gui/canvas.cpp:191-197."""
    e = _v(volume_size)
    max_span = f32(max(e[0], e[1], e[2])) * f32(1.5)
    half = f32(math.radians(float(f32(fov) * f32(0.5))))
    return float(max_span / f32(f32(2.0) * np.tan(half, dtype=np.float32)))


def bbox_set(vmin, vmax) -> cudaBBox:
    """cudaBBox::Set, core/geometry/cuda_bbox.h:25-31."""
    vmin, vmax = _v(vmin), _v(vmax)
    b = cudaBBox()
    b.vmin, b.vmax = _to_vec3(vmin), _to_vec3(vmax)
    b.invSize = _to_vec3(f32(1.0) / (vmax - vmin))
    return b


def volume_size(dim, spacing) -> np.ndarray:
    """VolumeReader::GetVolumeSize, core/VolumeReader.cpp:187-190 (dim is (nx, ny, nz))."""
    return (np.asarray(dim, dtype=np.float32) * _v(spacing)).astype(np.float32)


def bounding_sphere_radius(dim, spacing) -> float:
    """VolumeReader::GetBoundingSphereRadius, core/VolumeReader.cpp:192-196."""
    s = volume_size(dim, spacing)
    return float(np.sqrt(_dot(s, s), dtype=np.float32) * f32(0.5))


def element_bounding_sphere_radius(spacing) -> float:
    """VolumeReader::GetElementBoundingSphereRadius (the ray caster's stepSize), VolumeReader.cpp:198-201."""
    s = _v(spacing)
    return float(np.sqrt(_dot(s, s), dtype=np.float32) * f32(0.5))


def create_device_volume(tex_handle: int, dim, spacing, max_magnitude: float) -> cudaVolume:
    """VolumeReader::CreateDeviceVolume, core/VolumeReader.cpp:174-185 (+ Canvas::LoadVolume defaults,
    gui/canvas.cpp:30-32 and gui/canvas.cpp:19)."""
    ext = volume_size(dim, spacing)
    vmax = ext - ext * f32(0.5)
    vmin = -vmax
    vol = cudaVolume()
    vol.bbox = bbox_set(vmin, vmax)
    sp = _v(spacing)
    vol.spacing = _to_vec3(sp)
    vol.invSpacing = _to_vec3(f32(1.0) / sp)
    vol.tex = int(tex_handle)
    vol.invMaxMagnitude = float(f32(1.0) / f32(max_magnitude))
    vol.x_clip, vol.y_clip, vol.z_clip = vec2(-1.0, 1.0), vec2(-1.0, 1.0), vec2(-1.0, 1.0)
    vol.densityScale = 1.0
    vol.gradientFactor = 0.5
    return vol


def make_area_light(center, normal, radius=10.0, color=(1.0, 1.0, 1.0), intensity=500.0) -> cudaAreaLight:
    """cudaAreaLight::Set(cudaDisk(center, normal, radius), color, intensity), cuda_arealight.h:18-23."""
    l = cudaAreaLight()
    l.disk = cudaDisk(float(radius), _to_vec3(center), _to_vec3(normal))
    l.color = _to_vec3(color)
    l.intensity = float(intensity)
    return l


def place_area_light(latitude_deg: float, longitude_deg: float, distance: float, radius=10.0,
                     color=(1.0, 1.0, 1.0), intensity=500.0) -> cudaAreaLight:
    """Light placement of MainWindow (gui/mainwindow.cpp:229-238, 338-361): start at (0, distance, 0),
    rotate by latitude about X and by longitude about Z (sic), aim at the origin."""
    pos = np.array([0.0, distance, 0.0], dtype=np.float32)
    lat, lon = math.radians(latitude_deg), math.radians(longitude_deg)
    cz, sz = f32(math.cos(lon)), f32(math.sin(lon))
    p1 = np.array([cz * pos[0] - sz * pos[1], sz * pos[0] + cz * pos[1], pos[2]], dtype=np.float32)
    cx, sx = f32(math.cos(lat)), f32(math.sin(lat))
    p2 = np.array([p1[0], cx * p1[1] - sx * p1[2], sx * p1[1] + cx * p1[2]], dtype=np.float32)
    return make_area_light(p2, _normalize(-p2), radius, color, intensity)


def env_light_constant(radiance=(1.0, 1.0, 1.0), intensity=0.5) -> cudaEnvironmentLight:
    """Lights::SetEnvionmentLight(radiance) + SetEnvironmentLightIntensity, gui/canvas.cpp:11-12."""
    e = cudaEnvironmentLight()
    e.tex = 0
    e.defaultRadiance = _to_vec3(radiance)
    e.intensity = float(intensity)
    e.offset = vec2(0.0, 0.0)
    return e


class SvrError(RuntimeError):
    pass


class Device:
    """Thin RAII-style wrapper over the svr_* helper entry points (non-fatal error mode)."""

    def __init__(self, device: int = 0, fatal_errors: bool = False):
        self.lib = abi.load()
        self.lib.svr_set_error_mode(1 if fatal_errors else 0)
        self.check(self.lib.svr_init(int(device)))

    def check(self, rc=0):
        code = self.lib.svr_last_error_code()
        if rc != 0 or code != 0:
            msg = self.lib.svr_last_error().decode("utf-8", "replace")
            self.lib.svr_clear_error()
            raise SvrError(f"libsvr_hip error {code}: {msg}")

    def info(self) -> str:
        return self.lib.svr_device_info().decode()

    def malloc(self, nbytes: int) -> int:
        p = self.lib.svr_device_malloc(int(nbytes))
        self.check()
        if not p:
            raise SvrError("svr_device_malloc returned null")
        return int(p)

    def free(self, ptr: int):
        self.check(self.lib.svr_device_free(C.c_void_p(ptr)))

    def to_host(self, ptr: int, shape, dtype) -> np.ndarray:
        out = np.empty(shape, dtype=dtype)
        self.check(self.lib.svr_memcpy_d2h(out.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), out.nbytes))
        return out

    def to_device(self, ptr: int, arr: np.ndarray):
        arr = np.ascontiguousarray(arr)
        self.check(self.lib.svr_memcpy_h2d(C.c_void_p(ptr), arr.ctypes.data_as(C.c_void_p), arr.nbytes))

    def synchronize(self):
        self.check(self.lib.svr_device_synchronize())

    def set_option(self, key: int, value: int):
        self.check(self.lib.svr_set_option(int(key), int(value)))

    def get_option(self, key: int) -> int:
        return int(self.lib.svr_get_option(int(key)))

    def counters(self) -> dict:
        c = abi.Counters()
        self.check(self.lib.svr_get_counters(C.byref(c)))
        return c.as_dict()

    def reset_counters(self):
        self.check(self.lib.svr_reset_counters())

    def kernel_time(self):
        ms, n = C.c_double(0.0), C.c_uint64(0)
        self.check(self.lib.svr_get_kernel_time(C.byref(ms), C.byref(n)))
        return ms.value, int(n.value)


class Canvas:
    """Headless replay of the reference's Qt `Canvas` render protocol (gui/canvas.{h,cpp}).

    Owns RenderParams / cudaCamera / cudaVolume / cudaTransferFunction / lights exactly like the
    widget does; every setter re-uploads through the matching setup_* and restarts the progressive
    render (frameNo = 0, gui/canvas.h:43-47); paint() is paintGL's render branch
    (gui/canvas.cpp:90-116): render_*, then frameNo++.
    """

    RENDER_MODE_PATHTRACER, RENDER_MODE_RAYCASTING = 0, 1

    def __init__(self, dev: Device, width: int, height: int, img_ptr: Optional[int] = None,
                 hdr_ptr: Optional[int] = None):
        self.dev, self.lib = dev, dev.lib
        self.W, self.H = int(width), int(height)
        # Canvas::Canvas, gui/canvas.cpp:8-20
        self.env = env_light_constant((1.0, 1.0, 1.0), 0.5)
        self.lib.setup_env_lights(C.byref(self.env)); dev.check()
        self.renderParams = RenderParams(1, 0, None)
        self._own_hdr = hdr_ptr is None
        if hdr_ptr is None:
            dev.check(self.lib.svr_render_params_setup_hdr(C.byref(self.renderParams), self.W, self.H))
        else:
            self.renderParams.hdrBuffer = C.c_void_p(hdr_ptr)
        self._own_img = img_ptr is None
        self.img = dev.malloc(self.W * self.H * 4) if img_ptr is None else int(img_ptr)
        self.renderParams.traceDepth = 1
        self.deviceVolume = cudaVolume()
        self.deviceVolume.gradientFactor = 0.5
        self.transferFunction = cudaTransferFunction()
        self.camera = cudaCamera()
        self.areaLights: list = []
        self.renderMode = self.RENDER_MODE_PATHTRACER
        self.stepSize = 1.0
        self.fov, self.apeture, self.focalLength, self.exposure = 45.0, 0.0, 1.0, 1.0
        self.ready = False
        self._textures: list = []

    # ---- gui/canvas.h:43-47 ----
    def ReStartRender(self):
        self.renderParams.frameNo = 0

    # ---- Canvas::LoadVolume, gui/canvas.cpp:27-41 (the file reader is replaced by an array) ----
    def LoadVolume(self, voxels: np.ndarray, spacing, max_magnitude: float, layout: int = abi.LAYOUT_AUTO):
        vox = np.ascontiguousarray(voxels, dtype=np.uint16)
        nz, ny, nx = vox.shape
        h = self.lib.svr_create_volume_texture(vox.ctypes.data_as(C.c_void_p), nx, ny, nz, 0, int(layout))
        self.dev.check()
        old = getattr(self, "_volume_tex", 0)
        if old and old in self._textures:                      # a second LoadVolume replaces the texture: free the old one now
            self._textures.remove(old)
            self.lib.svr_destroy_texture(old)
        self._volume_tex = h
        self._textures.append(h)
        gf = self.deviceVolume.gradientFactor
        self.deviceVolume = create_device_volume(h, (nx, ny, nz), spacing, max_magnitude)
        self.deviceVolume.gradientFactor = gf
        self.lib.setup_volume(C.byref(self.deviceVolume)); self.dev.check()
        self.stepSize = element_bounding_sphere_radius(spacing)
        self.volumeSize = volume_size((nx, ny, nz), spacing)
        self.boundingSphereRadius = bounding_sphere_radius((nx, ny, nz), spacing)
        eye = zoom_to_extent_eye_dist(self.volumeSize, self.fov)
        self.eyeDist = eye
        self.camera = camera_setup((0.0, 0.0, eye), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), self.fov, self.apeture,
                                   self.focalLength, self.exposure, self.W, self.H)
        self.lib.setup_camera(C.byref(self.camera)); self.dev.check()
        self.ready = True
        self.ReStartRender()

    def LoadVolumeFile(self, filename: str, layout: int = abi.LAYOUT_AUTO):
        """Canvas::LoadVolume(filename), gui/canvas.cpp:27-41, with the reference's reader replaced by
        io.VolumeReader (MetaImage parse on the host, preprocessing on the GPU)."""
        from .io import VolumeReader
        if getattr(self, "volumeReader", None) is not None:
            self.volumeReader.ClearDevice()
        self.volumeReader = VolumeReader(self.dev, layout)
        self.volumeReader.Read(filename)
        self.volumeReader.CreateDeviceVolume(self.deviceVolume)
        self.deviceVolume.x_clip, self.deviceVolume.y_clip, self.deviceVolume.z_clip = vec2(-1, 1), vec2(-1, 1), vec2(-1, 1)
        self.deviceVolume.densityScale = 1.0
        self.lib.setup_volume(C.byref(self.deviceVolume)); self.dev.check()
        self.stepSize = self.volumeReader.GetElementBoundingSphereRadius()
        self.volumeSize = self.volumeReader.GetVolumeSize()
        self.boundingSphereRadius = self.volumeReader.GetBoundingSphereRadius()
        self.eyeDist = zoom_to_extent_eye_dist(self.volumeSize, self.fov)
        self.camera = camera_setup((0.0, 0.0, self.eyeDist), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), self.fov, self.apeture,
                                   self.focalLength, self.exposure, self.W, self.H)
        self.lib.setup_camera(C.byref(self.camera)); self.dev.check()
        self.ready = True
        self.ReStartRender()

    def SetEnvLightMap(self, filename: str):
        """gui/canvas.h:104-109: Lights::SetEnvironmentLight(filename) + setup_env_lights.  Like the reference's
        cudaEnvironmentLight::Set(tex), this resets the intensity to 1 and the offset to 0."""
        old = int(self.env.tex)
        self.dev.check(self.lib.svr_load_env_map(str(filename).encode(), C.byref(self.env)))
        self._textures.append(int(self.env.tex))
        if old in self._textures:
            self._textures.remove(old)
            self.lib.svr_destroy_texture(old)
        self.lib.setup_env_lights(C.byref(self.env)); self.dev.check()
        self.ReStartRender()

    def SaveFrame(self, filename: str = "0.tga"):
        """The frame dump of gui/canvas.cpp:97-104 (stbi_write_tga of the RGBA8 image)."""
        from .io import save_tga
        save_tga(self.dev, filename, self.read_img())

    def SetCamera(self, cam: cudaCamera):
        self.camera = cam
        self.lib.setup_camera(C.byref(self.camera)); self.dev.check()
        self.ReStartRender()

    # ---- gui/canvas.h:49-54 + TransferFunction ctor, gui/transferfunction.cpp:17-44 ----
    def SetTransferFunctionTable(self, rgba: np.ndarray, maxOpacity: float):
        t = np.ascontiguousarray(rgba, dtype=np.float32).reshape(-1, 4)
        h = self.lib.svr_create_tf_texture(t.ctypes.data_as(C.c_void_p), t.shape[0], 0)
        self.dev.check()
        old = getattr(self, "_tf_tex", 0)
        self._tf_tex = h
        self._textures.append(h)
        self.SetTransferFunction(h, maxOpacity)
        if old and old in self._textures:                      # replaced: free the previous table
            self._textures.remove(old)
            self.lib.svr_destroy_texture(old)

    def SetTransferFunction(self, tex: int, maxOpacity: float):
        self.transferFunction.tex = int(tex)
        self.transferFunction.maxOpacity = float(maxOpacity)
        self.lib.setup_transferfunction(C.byref(self.transferFunction)); self.dev.check()
        self.ReStartRender()

    # ---- gui/canvas.h:56-175 ----
    def SetDensityScale(self, s: float):
        self.deviceVolume.densityScale = float(s)
        self.lib.setup_volume(C.byref(self.deviceVolume)); self.dev.check()
        self.ReStartRender()

    def SetGradientFactor(self, g: float):
        self.deviceVolume.gradientFactor = float(g)
        self.lib.setup_volume(C.byref(self.deviceVolume)); self.dev.check()
        self.ReStartRender()

    def SetScatterTimes(self, val: int):
        self.renderParams.traceDepth = int(val)
        self.ReStartRender()

    def SetRenderMode(self, mode: int):
        self.renderMode = mode
        self.ReStartRender()

    def SetClipPlane(self, x_clip, y_clip, z_clip):
        self.deviceVolume.x_clip = vec2(*x_clip)
        self.deviceVolume.y_clip = vec2(*y_clip)
        self.deviceVolume.z_clip = vec2(*z_clip)
        self.lib.setup_volume(C.byref(self.deviceVolume)); self.dev.check()
        self.ReStartRender()

    def SetEnvLightBackground(self, color):
        self.env.tex = 0
        self.env.defaultRadiance = _to_vec3(color)
        self.lib.setup_env_lights(C.byref(self.env)); self.dev.check()
        self.ReStartRender()

    def SetEnvLightMapTable(self, rgba: np.ndarray):
        t = np.ascontiguousarray(rgba, dtype=np.float32)
        h, w = t.shape[0], t.shape[1]
        tex = self.lib.svr_create_env_texture(t.ctypes.data_as(C.c_void_p), w, h, 0)
        self.dev.check()
        self._textures.append(tex)
        self.env.tex = tex
        self.lib.setup_env_lights(C.byref(self.env)); self.dev.check()
        self.ReStartRender()

    def SetEnvLightIntensity(self, intensity: float):
        self.env.intensity = float(intensity)
        self.lib.setup_env_lights(C.byref(self.env)); self.dev.check()
        self.ReStartRender()

    def SetEnvLightOffset(self, offset):
        self.env.offset = vec2(*offset)
        self.lib.setup_env_lights(C.byref(self.env)); self.dev.check()
        self.ReStartRender()

    def SetAreaLights(self, lights: Sequence[cudaAreaLight]):
        self.areaLights = list(lights)
        n = len(self.areaLights)
        arr = (cudaAreaLight * max(n, 1))(*self.areaLights)
        self.lib.setup_area_lights(arr, n); self.dev.check()
        self.ReStartRender()

    def SetExposure(self, exposure: float):
        self.exposure = float(exposure)
        self.camera.exposure = float(exposure)
        self.lib.setup_camera(C.byref(self.camera)); self.dev.check()
        self.ReStartRender()

    # ---- Canvas::paintGL render branch, gui/canvas.cpp:90-116 ----
    def paint(self, sync: bool = False):
        if not self.ready:
            return
        if self.renderMode == self.RENDER_MODE_RAYCASTING:
            self.lib.render_raycasting(C.c_void_p(self.img), C.byref(self.deviceVolume), C.byref(self.transferFunction),
                                       C.byref(self.camera), C.c_float(self.stepSize))
        else:
            self.lib.render_pathtracer(C.c_void_p(self.img), C.byref(self.renderParams))
        self.dev.check()
        if sync:
            self.dev.synchronize()
        self.renderParams.frameNo += 1

    def paint_frames(self, nframes: int, sync: bool = False):
        """Extension: nframes progressive frames in one launch (svr_render_pathtracer_frames)."""
        self.dev.check(self.lib.svr_render_pathtracer_frames(C.c_void_p(self.img), C.byref(self.renderParams), int(nframes)))
        if sync:
            self.dev.synchronize()
        self.renderParams.frameNo += int(nframes)

    def read_hdr(self) -> np.ndarray:
        return self.dev.to_host(int(self.renderParams.hdrBuffer), (self.H, self.W, 3), np.float32)

    def read_img(self) -> np.ndarray:
        return self.dev.to_host(self.img, (self.H, self.W, 4), np.uint8)

    def close(self):
        self.dev.synchronize()
        for h in self._textures:
            self.lib.svr_destroy_texture(h)
        self._textures.clear()
        if getattr(self, "volumeReader", None) is not None:
            self.volumeReader.ClearDevice()
            self.volumeReader = None
        if self._own_hdr:
            self.lib.svr_render_params_clear(C.byref(self.renderParams))
        if self._own_img and self.img:
            self.lib.svr_device_free(C.c_void_p(self.img))
            self.img = 0
        self.lib.svr_clear_error()
