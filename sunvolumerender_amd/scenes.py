"""Deterministic synthetic scenes; no files, no network.

Volumes are generated analytically (+ integer-hash noise), transfer functions are the reference
GUI's defaults (gui/mainwindow.cpp:51-62), cameras/lights follow the reference's defaults
(gui/canvas.cpp:36,191-197; gui/mainwindow.cpp:229-238).  A `Scene` is plain data: numpy arrays
plus the reference's POD structs, consumable by the HIP renderer (through `Canvas`) and by the
test oracle alike.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import List, Optional, Tuple

import numpy as np

from . import host
from .abi import TF_TABLE_SIZE, cudaAreaLight, cudaCamera

f32 = np.float32


def wang_hash_np(a: np.ndarray) -> np.ndarray:
    """pathtracer.cu:70-79, vectorised (uint32 wrap-around)."""
    a = a.astype(np.uint32, copy=True)
    with np.errstate(over="ignore"):
        a = (a ^ np.uint32(61)) ^ (a >> np.uint32(16))
        a = a + (a << np.uint32(3))
        a = a ^ (a >> np.uint32(4))
        a = a * np.uint32(0x27D4EB2D)
        a = a ^ (a >> np.uint32(15))
    return a


def _coords(n: int, z0: int, z1: int):
    """voxel-centre coordinates in the unit cube centred at the origin, for z-slab [z0,z1)."""
    ax = ((np.arange(n, dtype=np.float32) + f32(0.5)) / f32(n) - f32(0.5)).astype(np.float32)
    az = ax[z0:z1]
    return az[:, None, None], ax[None, :, None], ax[None, None, :]


def make_sphere_volume(n: int) -> np.ndarray:
    """C1 volume: density 1 - r/(2*0.35) inside r < 0.35 (unit-cube coordinates), 0 outside."""
    out = np.empty((n, n, n), dtype=np.uint16)
    step = max(1, min(n, (1 << 22) // (n * n)))
    for z0 in range(0, n, step):
        z1 = min(n, z0 + step)
        z, y, x = _coords(n, z0, z1)
        r = np.sqrt(x * x + y * y + z * z, dtype=np.float32)
        d = np.where(r < f32(0.35), f32(1.0) - r / f32(0.7), f32(0.0)).astype(np.float32)
        out[z0:z1] = np.rint(d * f32(65535.0)).astype(np.uint16)
    return out


def _smooth(edge0, edge1, x):
    t = np.clip((x - edge0) / (edge1 - edge0), f32(0.0), f32(1.0)).astype(np.float32)
    return (t * t * (f32(3.0) - f32(2.0) * t)).astype(np.float32)


def make_ct_head_volume(n: int, seed: int = 1, noisy_air: bool = False) -> np.ndarray:
    """'CT-head-like' phantom: nested ellipsoids (skin 0.25, soft tissue 0.35, bone shell 0.8, brain 0.4
    with two dark ventricles) with ~1.5-voxel soft edges and low-amplitude hash noise inside the head;
    air is exactly 0 so the GUI-default transfer function leaves it transparent.
    noisy_air: every voxel is at least 64 + (hash & 127) raw units -- what real CT data looks like after the
    reference rescales it to the full u16 range (VolumeReader.cpp:124-136): air is noisy and NON-zero, the
    GUI-default transfer function (alpha > 0 above raw ~32) is nowhere exactly transparent, and no macro-cell can
    be skipped on the grounds that its opacity is exactly 0."""
    out = np.empty((n, n, n), dtype=np.uint16)
    step = max(1, min(n, (1 << 21) // (n * n)))
    e = f32(1.5 / n)  # edge softness in unit-cube units
    for z0 in range(0, n, step):
        z1 = min(n, z0 + step)
        z, y, x = _coords(n, z0, z1)

        def ell(cx, cy, cz, ax_, ay_, az_):
            return np.sqrt(((x - f32(cx)) / f32(ax_)) ** 2 + ((y - f32(cy)) / f32(ay_)) ** 2 +
                           ((z - f32(cz)) / f32(az_)) ** 2, dtype=np.float32)

        r_head = ell(0, 0, 0, 0.36, 0.44, 0.38)
        r_skull_o = ell(0, 0.01, 0, 0.335, 0.41, 0.355)
        r_skull_i = ell(0, 0.01, 0, 0.30, 0.37, 0.32)
        r_v1 = ell(-0.06, 0.03, 0.0, 0.035, 0.09, 0.05)
        r_v2 = ell(0.06, 0.03, 0.0, 0.035, 0.09, 0.05)
        s = e / f32(0.36)
        inside_head = f32(1.0) - _smooth(f32(1.0) - s, f32(1.0) + s, r_head)
        skin = f32(1.0) - _smooth(f32(0.965) - s, f32(0.965) + s, r_head)          # interior of skin layer
        bone_o = f32(1.0) - _smooth(f32(1.0) - s, f32(1.0) + s, r_skull_o)
        bone_i = f32(1.0) - _smooth(f32(1.0) - s, f32(1.0) + s, r_skull_i)
        vent = np.maximum(f32(1.0) - _smooth(f32(1.0) - 4 * s, f32(1.0) + 4 * s, r_v1),
                          f32(1.0) - _smooth(f32(1.0) - 4 * s, f32(1.0) + 4 * s, r_v2))
        d = f32(0.25) * inside_head
        d = d + f32(0.10) * skin                          # soft tissue 0.35
        d = d + f32(0.45) * bone_o                        # bone 0.80
        d = d - f32(0.40) * bone_i                        # brain 0.40
        d = d - f32(0.22) * vent * bone_i                 # ventricles 0.18
        nz_, ny_, nx_ = (z1 - z0), n, n
        zi = np.arange(z0, z1, dtype=np.uint32)[:, None, None]
        yi = np.arange(n, dtype=np.uint32)[None, :, None]
        xi = np.arange(n, dtype=np.uint32)[None, None, :]
        with np.errstate(over="ignore"):
            idx = xi + np.uint32(n) * (yi + np.uint32(n) * zi) + np.uint32(seed)
        noise = (wang_hash_np(np.broadcast_to(idx, (nz_, ny_, nx_))).astype(np.float32) / f32(4294967296.0) - f32(0.5))
        d = d + f32(0.03) * noise * inside_head
        d = np.clip(d, f32(0.0), f32(1.0))
        raw = np.rint(d * f32(65535.0)).astype(np.uint16)
        if noisy_air:
            with np.errstate(over="ignore"):
                floor = np.uint32(64) + (wang_hash_np(np.broadcast_to(idx + np.uint32(0x9E3779B9), (nz_, ny_, nx_))) & np.uint32(127))
            raw = np.maximum(raw, floor.astype(np.uint16))
        out[z0:z1] = raw
    return out


def max_gradient_magnitude(vox: np.ndarray, spacing=(1.0, 1.0, 1.0)) -> float:
    """Stand-in for vtkImageGradientMagnitude's range maximum (VolumeReader.cpp:70-76): central
    differences with replicated borders on the raw integer data, / (2*spacing)."""
    nz = vox.shape[0]
    sp = np.asarray(spacing, dtype=np.float32)
    best = 0.0
    step = max(1, (1 << 22) // (vox.shape[1] * vox.shape[2]))
    for z0 in range(0, nz, step):
        z1 = min(nz, z0 + step)
        lo, hi = max(0, z0 - 1), min(nz, z1 + 1)
        blk = vox[lo:hi].astype(np.float32)
        if z0 == 0:
            blk = np.concatenate([blk[:1], blk], axis=0)
        if z1 == nz:
            blk = np.concatenate([blk, blk[-1:]], axis=0)
        pad = np.pad(blk, ((0, 0), (1, 1), (1, 1)), mode="edge")
        gz = (pad[2:, 1:-1, 1:-1] - pad[:-2, 1:-1, 1:-1]) / (f32(2.0) * sp[2])
        gy = (pad[1:-1, 2:, 1:-1] - pad[1:-1, :-2, 1:-1]) / (f32(2.0) * sp[1])
        gx = (pad[1:-1, 1:-1, 2:] - pad[1:-1, 1:-1, :-2]) / (f32(2.0) * sp[0])
        m = float(np.sqrt(gx * gx + gy * gy + gz * gz).max())
        best = max(best, m)
    return best


def _piecewise(points: List[Tuple[float, ...]], x: np.ndarray) -> np.ndarray:
    xs = np.array([p[0] for p in points], dtype=np.float64)
    cols = []
    for c in range(1, len(points[0])):
        ys = np.array([p[c] for p in points], dtype=np.float64)
        cols.append(np.interp(x, xs, ys))
    return np.stack(cols, axis=-1)


def default_transfer_function() -> Tuple[np.ndarray, float]:
    """The GUI default (gui/mainwindow.cpp:51-62) sampled like TransferFunction's constructor
    (gui/transferfunction.cpp:17-28): 1024 x RGBA float32 at x_i = i/1023.  Returns (table, maxOpacity);
    maxOpacity is the 0.5 the GUI passes initially (gui/mainwindow.cpp:27)."""
    x = np.arange(TF_TABLE_SIZE, dtype=np.float64) / (TF_TABLE_SIZE - 1)
    opacity = _piecewise([(0.0, 0.0)] + [(0.1 * i, 0.5) for i in range(1, 11)], x)[:, 0]
    color = _piecewise([(0.0, 69 / 255, 199 / 255, 186 / 255), (0.2, 172 / 255, 3 / 255, 57 / 255),
                        (0.4, 169 / 255, 83 / 255, 58 / 255), (0.6, 43 / 255, 32 / 255, 161 / 255),
                        (0.8, 247 / 255, 158 / 255, 97 / 255), (1.0, 183 / 255, 7 / 255, 140 / 255)], x)
    table = np.concatenate([color, opacity[:, None]], axis=1).astype(np.float32)
    return np.ascontiguousarray(table), 0.5


def bone_transfer_function() -> Tuple[np.ndarray, float]:
    """Thresholded look: transparent below 0.3, ramp to 1.0 at 0.7; warm colours.  maxOpacity = table max."""
    x = np.arange(TF_TABLE_SIZE, dtype=np.float64) / (TF_TABLE_SIZE - 1)
    opacity = _piecewise([(0.0, 0.0), (0.3, 0.0), (0.7, 1.0), (1.0, 1.0)], x)[:, 0]
    color = _piecewise([(0.0, 0.8, 0.5, 0.4), (0.4, 0.9, 0.7, 0.5), (0.7, 1.0, 0.95, 0.85), (1.0, 1.0, 1.0, 1.0)], x)
    table = np.concatenate([color, opacity[:, None]], axis=1).astype(np.float32)
    return np.ascontiguousarray(table), float(table[:, 3].max())


def synthetic_env_map(w: int = 512, h: int = 256) -> np.ndarray:
    """Lat-long RGBA float32 sky gradient with a warm band; rows = v (theta/pi), cols = u (phi/2pi)."""
    v = (np.arange(h, dtype=np.float32) + f32(0.5)) / f32(h)
    u = (np.arange(w, dtype=np.float32) + f32(0.5)) / f32(w)
    sky = (f32(1.0) - v)[:, None] * np.ones((1, w), dtype=np.float32)
    band = (f32(0.5) + f32(0.5) * np.cos(f32(2.0 * math.pi) * u, dtype=np.float32))[None, :] * np.ones((h, 1), dtype=np.float32)
    img = np.empty((h, w, 4), dtype=np.float32)
    img[..., 0] = f32(0.35) + f32(0.45) * sky + f32(0.20) * band
    img[..., 1] = f32(0.40) + f32(0.45) * sky + f32(0.05) * band
    img[..., 2] = f32(0.45) + f32(0.55) * sky
    img[..., 3] = f32(1.0)
    return np.ascontiguousarray(img)


@dataclass
class Scene:
    name: str
    vox: np.ndarray                       # [nz][ny][nx] uint16
    spacing: Tuple[float, float, float]
    max_magnitude: float
    tf_rgba: np.ndarray                   # [n][4] float32
    max_opacity: float
    width: int
    height: int
    lights: List[cudaAreaLight] = field(default_factory=list)
    env_radiance: Tuple[float, float, float] = (1.0, 1.0, 1.0)
    env_intensity: float = 0.5
    env_map: Optional[np.ndarray] = None   # [h][w][4] float32
    env_offset: Tuple[float, float] = (0.0, 0.0)
    env_on_escape: bool = False
    trace_depth: int = 1
    density_scale: float = 1.0
    gradient_factor: float = 0.5
    clip: Tuple[Tuple[float, float], Tuple[float, float], Tuple[float, float]] = ((-1.0, 1.0), (-1.0, 1.0), (-1.0, 1.0))
    fov: float = 45.0
    apeture: float = 0.0
    focal_length: float = 1.0
    exposure: float = 1.0
    camera: Optional[cudaCamera] = None    # None -> Canvas::LoadVolume default (eye on +z, ZoomToExtent)
    spp: int = 1

    @property
    def dim(self):
        nz, ny, nx = self.vox.shape
        return (nx, ny, nz)

    def default_camera(self) -> cudaCamera:
        eye = host.zoom_to_extent_eye_dist(host.volume_size(self.dim, self.spacing), self.fov)
        return host.camera_setup((0.0, 0.0, eye), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), self.fov, self.apeture,
                                 self.focal_length, self.exposure, self.width, self.height)

    def resolved_camera(self) -> cudaCamera:
        return self.camera if self.camera is not None else self.default_camera()

    def step_size(self) -> float:
        return host.element_bounding_sphere_radius(self.spacing)


def default_light(dim, spacing) -> cudaAreaLight:
    """gui/mainwindow.cpp:229-238: disk radius 10, colour 1, intensity 500, on +y at 1.5*R + 1."""
    R = host.bounding_sphere_radius(dim, spacing)
    dist = float(f32(R) * f32(1.5) + f32(1.0))
    return host.place_area_light(0.0, 0.0, dist, 10.0, (1.0, 1.0, 1.0), 500.0)


def three_lights(dim, spacing) -> List[cudaAreaLight]:
    """C3: the GUI default light plus two large disks at latitude 60, longitude +-50 degrees (outside the
    camera frustum, so no primary ray is blocked by a light's back side); radius scales with the volume."""
    R = host.bounding_sphere_radius(dim, spacing)
    dist = float(f32(R) * f32(1.5) + f32(1.0))
    rad = 120.0 * max(dim) / 512.0
    return [default_light(dim, spacing),
            host.place_area_light(60.0, 50.0, dist, rad, (1.0, 0.95, 0.9), 5500.0),
            host.place_area_light(60.0, -50.0, dist, rad, (0.9, 0.95, 1.0), 5500.0)]


_VOLUME_CACHE: dict = {}


def _volume(kind: str, n: int) -> Tuple[np.ndarray, float]:
    key = (kind, n)
    if key not in _VOLUME_CACHE:
        import os
        cache_dir = os.environ.get("SVR_SCENE_CACHE")          # optional on-disk cache of generated volumes
        path = os.path.join(cache_dir, f"{kind}_{n}.npz") if cache_dir else None
        if path and os.path.exists(path):
            z = np.load(path)
            _VOLUME_CACHE[key] = (z["vox"], float(z["maxmag"]))
        else:
            vox = make_sphere_volume(n) if kind == "sphere" else make_ct_head_volume(n, noisy_air=(kind == "head_noisy"))
            mm = max_gradient_magnitude(vox)
            _VOLUME_CACHE[key] = (vox, mm)
            if path:
                os.makedirs(cache_dir, exist_ok=True)
                np.savez(path, vox=vox, maxmag=np.float64(mm))
    return _VOLUME_CACHE[key]


def make_scene(name: str, **overrides) -> Scene:
    """Named configurations.  c1..c5 are BASELINE.json's configs; the rest are small test scenes."""
    presets = {
        # name: (volume kind, N, W, H, lights, env map, env_on_escape, spp, tf)
        "c1": ("sphere", 64, 256, 256, 1, False, False, 1, "default"),
        "c2": ("head", 256, 512, 512, 1, False, False, 64, "default"),
        "c3": ("head", 512, 1024, 1024, 3, True, True, 256, "default"),
        "c4": ("head", 512, 2048, 2048, 3, True, True, 1024, "default"),
        "c5": ("head", 1024, 1024, 1024, 3, True, True, 512, "default"),
        # c3 / tiny_head with noisy, non-zero air: nothing is exactly transparent (see make_ct_head_volume)
        "c3n": ("head_noisy", 512, 1024, 1024, 3, True, True, 256, "default"),
        # c3 under the bone transfer function: exactly transparent air AND translucent tissue (bound classes between 0 and 1)
        "c3b": ("head", 512, 1024, 1024, 3, True, True, 256, "bone"),
        "tiny_head_noisy": ("head_noisy", 48, 96, 80, 3, True, True, 1, "default"),
        "small_head_noisy": ("head_noisy", 128, 256, 256, 3, True, True, 1, "default"),
        "tiny": ("sphere", 32, 64, 64, 1, False, False, 1, "default"),
        "tiny_head": ("head", 48, 96, 80, 3, True, True, 1, "default"),
        "small_head": ("head", 128, 256, 256, 3, True, True, 1, "default"),
        "tiny_bone": ("head", 48, 64, 64, 1, False, False, 1, "bone"),
    }
    if name not in presets:
        raise KeyError(f"unknown scene {name!r}; known: {sorted(presets)}")
    kind, n, W, H, nl, envmap, env_esc, spp, tfname = presets[name]
    n = overrides.pop("n", n)
    W = overrides.pop("width", W)
    H = overrides.pop("height", H)
    vox, maxmag = _volume(kind, n)
    tf, max_op = default_transfer_function() if tfname == "default" else bone_transfer_function()
    spacing = (1.0, 1.0, 1.0)
    dim = (n, n, n)
    lights = [default_light(dim, spacing)] if nl == 1 else three_lights(dim, spacing)
    sc = Scene(name=name, vox=vox, spacing=spacing, max_magnitude=maxmag, tf_rgba=tf, max_opacity=max_op,
               width=W, height=H, lights=lights, env_map=synthetic_env_map() if envmap else None,
               env_on_escape=env_esc, spp=spp)
    for k, v in overrides.items():
        if not hasattr(sc, k):
            raise AttributeError(k)
        setattr(sc, k, v)
    return sc


def apply_to_canvas(scene: Scene, canvas: "host.Canvas", layout: int = 0):
    """Replay the reference's start-up protocol for `scene` on a Canvas (SURVEY.md 8(b) call protocol)."""
    from . import abi
    canvas.fov, canvas.apeture, canvas.focalLength, canvas.exposure = scene.fov, scene.apeture, scene.focal_length, scene.exposure
    canvas.SetTransferFunctionTable(scene.tf_rgba, scene.max_opacity)          # mainwindow.cpp:27
    canvas.deviceVolume.gradientFactor = scene.gradient_factor
    canvas.LoadVolume(scene.vox, scene.spacing, scene.max_magnitude, layout)   # canvas.cpp:27-41
    if scene.camera is not None:
        canvas.SetCamera(scene.camera)
    if scene.density_scale != 1.0:
        canvas.SetDensityScale(scene.density_scale)
    if scene.clip != ((-1.0, 1.0), (-1.0, 1.0), (-1.0, 1.0)):
        canvas.SetClipPlane(*scene.clip)
    canvas.SetEnvLightBackground(scene.env_radiance)
    canvas.SetEnvLightIntensity(scene.env_intensity)
    if scene.env_map is not None:
        canvas.SetEnvLightMapTable(scene.env_map)
    if scene.env_offset != (0.0, 0.0):
        canvas.SetEnvLightOffset(scene.env_offset)
    canvas.SetAreaLights(scene.lights)
    canvas.SetScatterTimes(scene.trace_depth)
    canvas.dev.set_option(abi.OPT_ENV_ON_ESCAPE, 1 if scene.env_on_escape else 0)
    canvas.ReStartRender()
