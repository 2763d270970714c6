// svr_device.hpp -- device building blocks of the volumetric path tracer for gfx950:
// XORWOW RNG, software texture samplers, camera, box/disk intersection, BSDF/phase
// library, light sampling, Russian roulette, tone map.  Each block cites the reference
// function whose arithmetic it reproduces; the kernels in svr_kernels.hip arrange these
// blocks MI355X-first (persistent waves, lane regeneration, LDS-resident LUT).
#pragma once
#include "svr_math.hpp"
#include "svr_scene.hpp"

namespace svr {

// --------------------------------------------------------------------------
// cuRAND XORWOW, curand_init(seed,0,0) + curand_uniform  (call sites pathtracer.cu:206,
// woodcock_tracking.h:34,43, sampling.h, henyey_greenstein.h:32-36, microfacet.h:73-75)
// --------------------------------------------------------------------------
struct Rng { uint32_t v0, v1, v2, v3, v4, d; };

SVR_DEV void rng_init(Rng& r, uint32_t seed)
{
    uint32_t s0 = seed ^ 0xaad26b49u;
    uint32_t t0 = 1099087573u * s0;
    const uint32_t t1 = 2591861531u * 0xf7dcefddu;    // high seed word is zero
    r.d = 6615241u + t1 + t0;
    r.v0 = 123456789u + t0;
    r.v1 = 362436069u ^ t0;
    r.v2 = 521288629u + t1;
    r.v3 = 88675123u ^ t1;
    r.v4 = 5783321u + t0;
}

// curand_uniform's mapping of the draw's integer x = v4 + d
// _curand_uniform: x * 2^-32 + 2^-33, a multiply and an add.  The product is EXACT (a power-of-two scaling of a float in [0, 2^32], nowhere near the
// subnormals), so the one rounding of a fused multiply-add is the add's: the same bits in one instruction instead of two.
#ifndef SVR_UNIFORM_FMA
#define SVR_UNIFORM_FMA 1
#endif
SVR_DEV float rng_to_uniform(uint32_t x)
{
#if SVR_UNIFORM_FMA
    return __builtin_fmaf((float)x, 2.3283064365386963e-10f, 1.1641532182693481e-10f);
#else
    return (float)x * 2.3283064365386963e-10f + 1.1641532182693481e-10f;
#endif
}

SVR_DEV float rng_uniform(Rng& r)
{
    uint32_t t = r.v0 ^ (r.v0 >> 2);
    r.v0 = r.v1;
    r.v1 = r.v2;
    r.v2 = r.v3;
    r.v3 = r.v4;
    r.v4 = (r.v4 ^ (r.v4 << 4)) ^ (t ^ (t << 1));
    r.d += 362437u;
    return rng_to_uniform(r.v4 + r.d);
}

// advance the generator by one draw whose value is not needed
SVR_DEV void rng_skip(Rng& r)
{
    uint32_t t = r.v0 ^ (r.v0 >> 2);
    r.v0 = r.v1;
    r.v1 = r.v2;
    r.v2 = r.v3;
    r.v3 = r.v4;
    r.v4 = (r.v4 ^ (r.v4 << 4)) ^ (t ^ (t << 1));
    r.d += 362437u;
}

// The same generator as a CIRCULAR BUFFER with a compile-time head H: logical v_i = word (H + i) % 5.  A step overwrites
// the word that falls out of the recurrence (logical v0) with the new v4 and the head moves on, instead of shifting the
// other four words: no register moves.  Loops that hold k steps per iteration unroll 5 / gcd(5, k) times with
// H = 0, k, 2k ... (mod 5), after which the buffer is where it started; an exit in between records its head for rng_canon.
template <int H> SVR_DEV uint32_t& rng_word(Rng& r)
{
    static_assert(H >= 0 && H < 5, "head");
    if constexpr (H == 0) return r.v0;
    else if constexpr (H == 1) return r.v1;
    else if constexpr (H == 2) return r.v2;
    else if constexpr (H == 3) return r.v3;
    else return r.v4;
}
// the xorshift part of one step with head H (afterwards the head is (H + 1) % 5): returns the new v4; the Weyl word d is the caller's
template <int H> SVR_DEV uint32_t rng_xorshift_rot(Rng& r)
{
    uint32_t& w0 = rng_word<H>(r);
    const uint32_t w4 = rng_word<(H + 4) % 5>(r);
    const uint32_t t = w0 ^ (w0 >> 2);
    w0 = (w4 ^ (w4 << 4)) ^ (t ^ (t << 1));
    return w0;
}
constexpr uint32_t RNG_WEYL = 362437u;
// back to head 0 (logical v_i into field i) for a head known only at run time (the exit a lane took out of such a loop):
// a barrel rotation, 3 x 5 selects.  The exits themselves leave every word where it is, so the loop body carries no moves.
SVR_DEV void rng_canon(Rng& r, uint32_t head)
{
    uint32_t a0 = r.v0, a1 = r.v1, a2 = r.v2, a3 = r.v3, a4 = r.v4;
    if (head & 1u) { const uint32_t t0 = a0; a0 = a1; a1 = a2; a2 = a3; a3 = a4; a4 = t0; }
    if (head & 2u) { const uint32_t t0 = a0, t1 = a1; a0 = a2; a1 = a3; a2 = a4; a3 = t0; a4 = t1; }
    if (head & 4u) { const uint32_t t4 = a4; a4 = a3; a3 = a2; a2 = a1; a1 = a0; a0 = t4; }
    r.v0 = a0; r.v1 = a1; r.v2 = a2; r.v3 = a3; r.v4 = a4;
}
template <int H> struct RngHead { static constexpr int value = H; };

// pathtracer.cu:70-79 (host side too; see svr_api)
__host__ __device__ inline uint32_t wang_hash(uint32_t a)
{
    a = (a ^ 61u) ^ (a >> 16);
    a = a + (a << 3);
    a = a ^ (a >> 4);
    a = a * 0x27d4eb2du;
    a = a ^ (a >> 15);
    return a;
}

// --------------------------------------------------------------------------
// Software volume texture: tex3D<float>, border addressing, linear filter, normalized
// coordinates, normalized-float read (VolumeReader.cpp:159-167).  The array carries a
// 2-voxel zero apron, so border addressing is a clamp of the cell index, not 8 range
// tests.  Filtering runs on the raw integer values; one multiply by 1/65535 normalises.
// --------------------------------------------------------------------------
template <int LAYOUT>
SVR_DEV float tex3d(const DevScene& s, float u, float v, float w)
{
    float xb = fma_(u, s.fnx, -0.5f);
    float yb = fma_(v, s.fny, -0.5f);
    float zb = fma_(w, s.fnz, -0.5f);
    float fx = __builtin_floorf(xb), fy = __builtin_floorf(yb), fz = __builtin_floorf(zb);
    float a = xb - fx, b = yb - fy, g = zb - fz;
    fx = fmin_(fmax_(fx, -2.f), s.fnx);
    fy = fmin_(fmax_(fy, -2.f), s.fny);
    fz = fmin_(fmax_(fz, -2.f), s.fnz);
    int i = (int)fx + VOL_PAD, j = (int)fy + VOL_PAD, k = (int)fz + VOL_PAD;   // >= 0
    const uint16_t* __restrict__ vox = s.vox;
    float v000, v100, v010, v110, v001, v101, v011, v111;
    if (LAYOUT == LAYOUT_LINEAR) {
        int base = (k * s.sz + j * s.sy) + i;
        const uint16_t* p0 = vox + base;
        const uint16_t* p1 = p0 + s.sy;
        const uint16_t* p2 = p0 + s.sz;
        const uint16_t* p3 = p2 + s.sy;
        v000 = (float)p0[0]; v100 = (float)p0[1];
        v010 = (float)p1[0]; v110 = (float)p1[1];
        v001 = (float)p2[0]; v101 = (float)p2[1];
        v011 = (float)p3[0]; v111 = (float)p3[1];
    } else if (LAYOUT == LAYOUT_CELL) {
        // 16-byte elements: the 8 voxels of the cell (svr_walk.hpp, tex_fetch)
        const uint32_t ui = (uint32_t)i, uj = (uint32_t)j, uk = (uint32_t)k;
        const uint32_t e = (((ui >> 3) << 7) + (ui & 7u)) + ((uj >> 2) * ((uint32_t)s.bnx << 7) + ((uj & 3u) << 3)) +
                           ((uk >> 2) * ((uint32_t)(s.bny * s.bnx) << 7) + ((uk & 3u) << 5));
        const uint4 c4 = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(vox) + ((size_t)e << 4));
        v000 = (float)(c4.x & 0xffffu); v100 = (float)(c4.x >> 16);
        v010 = (float)(c4.y & 0xffffu); v110 = (float)(c4.y >> 16);
        v001 = (float)(c4.z & 0xffffu); v101 = (float)(c4.z >> 16);
        v011 = (float)(c4.w & 0xffffu); v111 = (float)(c4.w >> 16);
    } else if (LAYOUT == LAYOUT_PAIR) {
        // 32-bit elements: voxel x | voxel x + 1 << 16 (svr_walk.hpp, tex_fetch)
        int j1 = j + 1, k1 = k + 1;
        int X0 = ((i >> 3) << 7) + (i & 7);
        int Y0 = (j >> 2) * (s.bnx << 7) + ((j & 3) << 3), Y1 = (j1 >> 2) * (s.bnx << 7) + ((j1 & 3) << 3);
        int zs = (s.bny * s.bnx) << 7;
        int Z0 = (k >> 2) * zs + ((k & 3) << 5), Z1 = (k1 >> 2) * zs + ((k1 & 3) << 5);
        const uint32_t* __restrict__ pv = reinterpret_cast<const uint32_t*>(vox);
        const uint32_t p00 = pv[Y0 + Z0 + X0], p10 = pv[Y1 + Z0 + X0], p01 = pv[Y0 + Z1 + X0], p11 = pv[Y1 + Z1 + X0];
        v000 = (float)(p00 & 0xffffu); v100 = (float)(p00 >> 16);
        v010 = (float)(p10 & 0xffffu); v110 = (float)(p10 >> 16);
        v001 = (float)(p01 & 0xffffu); v101 = (float)(p01 >> 16);
        v011 = (float)(p11 & 0xffffu); v111 = (float)(p11 >> 16);
    } else {
        int i1 = i + 1, j1 = j + 1, k1 = k + 1;
        int X0 = ((i >> 3) << 7) + (i & 7), X1 = ((i1 >> 3) << 7) + (i1 & 7);
        int Y0 = (j >> 2) * (s.bnx << 7) + ((j & 3) << 3), Y1 = (j1 >> 2) * (s.bnx << 7) + ((j1 & 3) << 3);
        int zs = (s.bny * s.bnx) << 7;
        int Z0 = (k >> 2) * zs + ((k & 3) << 5), Z1 = (k1 >> 2) * zs + ((k1 & 3) << 5);
        int a00 = Y0 + Z0, a10 = Y1 + Z0, a01 = Y0 + Z1, a11 = Y1 + Z1;
        v000 = (float)vox[a00 + X0]; v100 = (float)vox[a00 + X1];
        v010 = (float)vox[a10 + X0]; v110 = (float)vox[a10 + X1];
        v001 = (float)vox[a01 + X0]; v101 = (float)vox[a01 + X1];
        v011 = (float)vox[a11 + X0]; v111 = (float)vox[a11 + X1];
    }
    float c00 = lerpf(v000, v100, a);
    float c10 = lerpf(v010, v110, a);
    float c01 = lerpf(v001, v101, a);
    float c11 = lerpf(v011, v111, a);
    float c0 = lerpf(c00, c10, b);
    float c1 = lerpf(c01, c11, b);
    return lerpf(c0, c1, g) * 1.5259021896696422e-05f;
}

// cudaVolume::GetIntensity, core/cuda_volume.h:87-100
template <int LAYOUT>
SVR_DEV float volume_intensity(const DevScene& s, v3 p)
{
    float u = (p.x - s.vmin[0]) * s.invSize[0];
    float v = (p.y - s.vmin[1]) * s.invSize[1];
    float w = (p.z - s.vmin[2]) * s.invSize[2];
    return tex3d<LAYOUT>(s, u, v, w) * s.densityScale;
}

// cudaVolume::Gradient_CentralDiff, core/cuda_volume.h:54-61
template <int LAYOUT>
SVR_DEV v3 volume_gradient(const DevScene& s, v3 p)
{
    float xd = volume_intensity<LAYOUT>(s, V3(p.x + s.spacing[0], p.y + 0.f, p.z + 0.f)) -
               volume_intensity<LAYOUT>(s, V3(p.x - s.spacing[0], p.y - 0.f, p.z - 0.f));
    float yd = volume_intensity<LAYOUT>(s, V3(p.x + 0.f, p.y + s.spacing[1], p.z + 0.f)) -
               volume_intensity<LAYOUT>(s, V3(p.x - 0.f, p.y - s.spacing[1], p.z - 0.f));
    float zd = volume_intensity<LAYOUT>(s, V3(p.x + 0.f, p.y + 0.f, p.z + s.spacing[2])) -
               volume_intensity<LAYOUT>(s, V3(p.x - 0.f, p.y - 0.f, p.z - s.spacing[2]));
    return V3((xd * 0.5f) * s.invSpacing[0], (yd * 0.5f) * s.invSpacing[1], (zd * 0.5f) * s.invSpacing[2]);
}

// --------------------------------------------------------------------------
// Transfer function: tex1D<float4>, clamp / linear / normalized coordinates
// (gui/transferfunction.cpp:38-42; cuda_transfer_function.h:22-30).  `lut` may point at
// LDS (the kernels stage the 16 KiB table there) or at global memory.
// --------------------------------------------------------------------------
SVR_DEV void tf_index(const DevScene& s, float x, int& i0, int& i1, float& a)
{
    float xb = fma_(x, s.tf_nf, -0.5f);
    xb = fmin_(fmax_(xb, -1.f), s.tf_nf);
    float fx = __builtin_floorf(xb);
    a = xb - fx;
    int i = (int)fx;
    int hi = s.tf_n - 1;
    i0 = min(max(i, 0), hi);
    i1 = min(max(i + 1, 0), hi);
}

template <typename LUT>
SVR_DEV float tf_alpha(const DevScene& s, const LUT* lut, float x)
{
    int i0, i1; float a;
    tf_index(s, x, i0, i1, a);
    return lerpf(lut[4 * i0 + 3], lut[4 * i1 + 3], a);
}

template <typename LUT>
SVR_DEV void tf_rgba(const DevScene& s, const LUT* lut, float x, float out[4])
{
    int i0, i1; float a;
    tf_index(s, x, i0, i1, a);
    const float4 t0 = *reinterpret_cast<const float4*>(lut + 4 * i0);
    const float4 t1 = *reinterpret_cast<const float4*>(lut + 4 * i1);
    out[0] = lerpf(t0.x, t1.x, a);
    out[1] = lerpf(t0.y, t1.y, a);
    out[2] = lerpf(t0.z, t1.z, a);
    out[3] = lerpf(t0.w, t1.w, a);
}

// --------------------------------------------------------------------------
// Environment light: cudaEnvironmentLight::GetEnvRadiance(dir),
// core/lights/cuda_environment_light.h:58-72; tex2D<float4> wrap/linear/normalized
// (lights.cpp:60-70).  The reference never calls it (pathtracer.cu:233 is commented
// out); SVR_OPT_ENV_ON_ESCAPE enables it as a documented extension.
// --------------------------------------------------------------------------
SVR_DEV v3 env_radiance(const DevScene& s, v3 dir)
{
    if (s.env == nullptr)
        return V3(s.env_default[0] * s.env_intensity, s.env_default[1] * s.env_intensity, s.env_default[2] * s.env_intensity);
    float theta = acosf_(dir.y);
    float phi = atan2f_(dir.x, dir.z);
    phi = (float)(phi < 0.f ? (double)phi + 6.283185307179586 : (double)phi);
    float u = (float)((double)(phi * 0.5f) * 0.31830988618379067154);
    float v = (float)((double)theta * 0.31830988618379067154);
    u = u + s.env_offset[0];
    v = v + s.env_offset[1];
    float W = (float)s.env_w, H = (float)s.env_h;
    u = u - __builtin_floorf(u);
    v = v - __builtin_floorf(v);
    float xb = fma_(u, W, -0.5f);
    float yb = fma_(v, H, -0.5f);
    float fx = __builtin_floorf(xb), fy = __builtin_floorf(yb);
    float a = xb - fx, b = yb - fy;
    int i0 = (int)fx, j0 = (int)fy;
    int i1 = i0 + 1, j1 = j0 + 1;
    // wrap addressing.  u, v are in [0, 1] here, so i0 is in [-1, W-1] and i1 in [0, W]: one conditional add / subtract is the
    // modulo (four integer divisions by a run-time value cost ~200 instructions per escaping path, 3 % of the headline
    // kernel); the clamp only keeps a NaN direction inside the table
    i0 = i0 < 0 ? i0 + s.env_w : i0;
    i1 = i1 >= s.env_w ? i1 - s.env_w : i1;
    j0 = j0 < 0 ? j0 + s.env_h : j0;
    j1 = j1 >= s.env_h ? j1 - s.env_h : j1;
    i0 = min(max(i0, 0), s.env_w - 1); i1 = min(max(i1, 0), s.env_w - 1);
    j0 = min(max(j0, 0), s.env_h - 1); j1 = min(max(j1, 0), s.env_h - 1);
    const float4* e = reinterpret_cast<const float4*>(s.env);
    float4 t00 = e[j0 * s.env_w + i0], t10 = e[j0 * s.env_w + i1];
    float4 t01 = e[j1 * s.env_w + i0], t11 = e[j1 * s.env_w + i1];
    v3 r = V3(lerpf(lerpf(t00.x, t10.x, a), lerpf(t01.x, t11.x, a), b),
              lerpf(lerpf(t00.y, t10.y, a), lerpf(t01.y, t11.y, a), b),
              lerpf(lerpf(t00.z, t10.z, a), lerpf(t01.z, t11.z, a), b));
    return r * s.env_intensity;
}

// --------------------------------------------------------------------------
// sampling.h:26-32 uniform_sample_disk
// --------------------------------------------------------------------------
SVR_DEV void uniform_sample_disk(Rng& rng, float r, float& ox, float& oy)
{
    r *= __builtin_sqrtf(rng_uniform(rng));
    float theta = (float)(6.283185307179586 * (double)rng_uniform(rng));   // 2.f * M_PI is a double
    float sn, cs;
    sincosf_(theta, &sn, &cs);
    ox = cs * r;
    oy = sn * r;
}

// direction from (sinTheta, cosTheta, phi) in the frame of `w` -- the expression shared by
// sampling.h:55, henyey_greenstein.h:47 and microfacet.h:78
SVR_DEV v3 frame_direction(v3 w, float sinTheta, float cosTheta, float phi)
{
    onb_t onb = onb_from_w(w);
    float sn, cs;
    sincosf_(phi, &sn, &cs);
    v3 d = (onb.u * (sinTheta * cs) + onb.v * (sinTheta * sn)) + onb.w * cosTheta;
    return normalize(d);
}

// --------------------------------------------------------------------------
// cudaCamera::GenerateRay (thin lens, jittered), core/cuda_camera.h:66-83
// --------------------------------------------------------------------------
SVR_DEV void camera_ray(const DevScene& s, uint32_t x, uint32_t y, Rng& rng, v3& orig, v3& dir)
{
    float nx = 2.f * (((float)x + rng_uniform(rng)) / s.wm1) - 1.f;
    float ny = 2.f * (((float)y + rng_uniform(rng)) / s.hm1) - 1.f;
    nx = nx * s.aspectRatio * s.tanFovxOverTwo;
    ny = ny * s.tanFovxOverTwo;
    nx = nx * s.focalLength;
    ny = ny * s.focalLength;
    float ax, ay;
    if (s.cam_pinhole) {
        // apeture == +0: the lens sample is (cos * 0, sin * 0) = (+-0, +-0), and with no -0 in the camera position and positive
        // image-plane scales (checked on the host) adding / subtracting it changes no bit of orig or dir.  The two draws are
        // still consumed (sampling.h:26-32).
        rng_skip(rng); rng_skip(rng);
        ax = 0.f; ay = 0.f;
    } else
        uniform_sample_disk(rng, s.apeture, ax, ay);
    v3 cu = V3(s.cam_u[0], s.cam_u[1], s.cam_u[2]);
    v3 cv = V3(s.cam_v[0], s.cam_v[1], s.cam_v[2]);
    v3 cw = V3(s.cam_w[0], s.cam_w[1], s.cam_w[2]);
    orig = (V3(s.cam_pos[0], s.cam_pos[1], s.cam_pos[2]) + cu * ax) + cv * ay;
    dir = normalize((cu * (nx - ax) + cv * (ny - ay)) - cw * s.focalLength);
}

// cudaCamera::GenerateRay (pinhole, pixel centre), core/cuda_camera.h:85-95
SVR_DEV void camera_ray_pinhole(const DevScene& s, uint32_t x, uint32_t y, v3& orig, v3& dir)
{
    float nx = 2.f * (((float)x + 0.5f) / s.wm1) - 1.f;
    float ny = 2.f * (((float)y + 0.5f) / s.hm1) - 1.f;
    nx = nx * s.aspectRatio * s.tanFovxOverTwo;
    ny = ny * s.tanFovxOverTwo;
    orig = V3(s.cam_pos[0], s.cam_pos[1], s.cam_pos[2]);
    v3 cu = V3(s.cam_u[0], s.cam_u[1], s.cam_u[2]);
    v3 cv = V3(s.cam_v[0], s.cam_v[1], s.cam_v[2]);
    v3 cw = V3(s.cam_w[0], s.cam_w[1], s.cam_w[2]);
    dir = normalize((cu * nx + cv * ny) - cw);
}

// --------------------------------------------------------------------------
// cudaBBox::Intersect with clip planes, core/geometry/cuda_bbox.h:33-54
// --------------------------------------------------------------------------
SVR_DEV bool volume_intersect(const DevScene& s, v3 orig, v3 dir, float& tNear, float& tFar)
{
    v3 invDir = V3(1.f / dir.x, 1.f / dir.y, 1.f / dir.z);
    v3 tbot = invDir * (V3(s.clip_vmin[0], s.clip_vmin[1], s.clip_vmin[2]) - orig);
    v3 ttop = invDir * (V3(s.clip_vmax[0], s.clip_vmax[1], s.clip_vmax[2]) - orig);
    v3 tmn = V3(gmin(tbot.x, ttop.x), gmin(tbot.y, ttop.y), gmin(tbot.z, ttop.z));
    v3 tmx = V3(gmax(tbot.x, ttop.x), gmax(tbot.y, ttop.y), gmax(tbot.z, ttop.z));
    float largest_tmin = fmax_(tmn.x, fmax_(tmn.y, tmn.z));
    float smallest_tmax = fmin_(tmx.x, fmin_(tmx.y, tmx.z));
    tNear = largest_tmin;
    tFar = smallest_tmax;
    return smallest_tmax > largest_tmin;
}

// --------------------------------------------------------------------------
// cudaDisk::Intersect, core/geometry/cuda_disk.h:32-51;
// get_nearest_light_sample, core/lights/light_sample.h:23-49
// --------------------------------------------------------------------------
SVR_DEV bool disk_intersect(const DevLight& l, v3 orig, v3 dir, float& t)
{
    v3 normal = V3(l.normal[0], l.normal[1], l.normal[2]);
    v3 center = V3(l.center[0], l.center[1], l.center[2]);
    float denom = dot(normal, dir);
    if ((double)__builtin_fabsf(denom) > 1e-6) {
        v3 co = center - orig;
        t = dot(co, normal) / denom;
        if (t >= 0.f) {
            v3 p = orig + dir * t;
            v3 co2 = p - center;
            return __builtin_sqrtf(dot(co2, co2)) <= l.radius;
        }
        return false;
    }
    return false;
}

// returns light index or -1; tHit = ls.t
SVR_DEV int nearest_light(const DevScene& s, v3 orig, v3 dir, float& tHit)
{
    float tNear = SVR_FLT_MAX;
    float t = SVR_FLT_MAX;
    int id = -1;
    for (uint32_t i = 0; i < s.num_lights; ++i) {
        // (lights no camera ray can reach -- outside the view frustum, lens included: svr_api.hip -- cannot change the result)
        if (!((s.primary_light_mask >> i) & 1u)) continue;
        if (disk_intersect(s.lights[i], orig, dir, t) && (t < tNear)) {
            tNear = t;
            id = (int)i;
        }
    }
    tHit = (id != -1) ? tNear : -SVR_FLT_MAX;
    return id;
}

// --------------------------------------------------------------------------
// BSDF / phase library
// --------------------------------------------------------------------------
#define SVR_IOR 2.5f          // pathtracer.cu:30
#define SVR_ALPHA 0.15f       // pathtracer.cu:31
#define SVR_HG_ISO 0.07957747154594767f   // (float)(M_1_PI * 0.25f), henyey_greenstein.h:17-18 with g == 0
#define SVR_PI_F 3.14159265358979323846f

// core/bsdf/fresnel.h:10-15
SVR_DEV float schlick_fresnel(float ni, float no, float cosin)
{
    float R0 = (ni - no) * (ni - no) / ((ni + no) * (ni + no));
    float c = 1.f - cosin;
    return R0 + (1.f - R0) * c * c * c * c * c;
}

// core/bsdf/microfacet.h:18-25
SVR_DEV float beckmann_distribution(v3 normal, v3 wh, float alpha)
{
    float c2 = dot(normal, wh);
    c2 *= c2;
    return expf_((c2 - 1.f) / (alpha * alpha * c2)) / (SVR_PI_F * alpha * alpha * c2 * c2);
}

// core/bsdf/microfacet.h:42-50
SVR_DEV float geometry_cook_torrance(v3 wi, v3 wo, v3 normal, v3 wh)
{
    float cosO = dot(wo, wh);
    float cosTerm = dot(normal, wh);
    float g1 = 2.f * cosTerm * dot(normal, wo) / cosO;
    float g2 = 2.f * cosTerm * dot(normal, wi) / cosO;
    return fmin_(1.f, fmin_(g1, g2));
}

// core/bsdf/microfacet.h:52-68 (DISTRIBUTION_BECKMANN)
SVR_DEV float microfacet_brdf_f(v3 wi, v3 wo, v3 normal, float ior, float alpha)
{
    if (dot(wi, normal) * dot(wo, normal) < 0.f) return 0.f;
    v3 wh = normalize(wi + wo);
    float F = schlick_fresnel(1.f, ior, __builtin_fabsf(dot(wh, wo)));
    float G = geometry_cook_torrance(wi, wo, normal, wh);
    float D = beckmann_distribution(normal, wh, alpha);
    return F * G * D / (4.f * __builtin_fabsf(dot(normal, wi)) * __builtin_fabsf(dot(normal, wo)));
}

// what survives of a VolumeSample (core/cuda_volume.h:124-132) between the stages of a bounce
struct Shade {
    v3 pt;            // ptInWorld
    v3 wo;
    v3 gradient;
    float color[4];   // color_opacity
    float Pbrdf;
    int st;           // 0 isotropic, 1 BRDF (pathtracer.cu:105)
};

// bsdf(), pathtracer.cu:106-131
SVR_DEV v3 bsdf_eval(const Shade& vs, v3 wi)
{
    v3 diffuseColor = V3(vs.color[0], vs.color[1], vs.color[2]);
    if (vs.st == 0)
        return diffuseColor * SVR_HG_ISO;
    v3 normal = normalize(vs.gradient);
    normal = dot(vs.wo, normal) < 0.f ? -normal : normal;
    float cosTerm = fmax_(0.f, dot(wi, normal));
    float ks = schlick_fresnel(1.0f, SVR_IOR, cosTerm);
    float kd = 1.f - ks;
    v3 diffuse = diffuseColor * (1.f / SVR_PI_F);
    v3 specular = V3(1.f, 1.f, 1.f) * microfacet_brdf_f(wi, vs.wo, normal, SVR_IOR, SVR_ALPHA);
    return (diffuse * kd + specular * ks) * cosTerm;
}

// sample_bsdf(), pathtracer.cu:133-169
SVR_DEV v3 bsdf_sample(const Shade& vs, v3& wi, float& pdf, Rng& rng)
{
    if (vs.st == 0) {
        // hg_phase_sample_f with g == 0, henyey_greenstein.h:29-51
        float phi = (float)(6.283185307179586 * (double)rng_uniform(rng));
        float cosTheta = 1.f - 2.f * rng_uniform(rng);
        float sinTheta = __builtin_sqrtf(fmax_(0.f, 1.f - cosTheta * cosTheta));
        wi = frame_direction(vs.wo, sinTheta, cosTheta, phi);
        pdf = SVR_HG_ISO;
        return V3(vs.color[0], vs.color[1], vs.color[2]) * SVR_HG_ISO;
    }
    v3 normal = normalize(vs.gradient);
    float cosTerm = dot(vs.wo, normal);
    if (cosTerm < 0.f) {
        cosTerm = -cosTerm;
        normal = -normal;
    }
    float ks = schlick_fresnel(1.f, SVR_IOR, cosTerm);
    float kd = 1.f - ks;
    float p = 0.25f + 0.5f * ks;
    if (rng_uniform(rng) < p) {
        // microfacet_brdf_sample_f + sample_beckmann, microfacet.h:70-79,95-111
        float phi = 2.f * SVR_PI_F * rng_uniform(rng);
        float cosTheta = 1.f / (1.f - SVR_ALPHA * SVR_ALPHA * logf_(1.f - rng_uniform(rng)));
        float sinTheta = __builtin_sqrtf(fmax_(0.f, 1.f - cosTheta * cosTheta));
        v3 wh = frame_direction(normal, sinTheta, cosTheta, phi);
        wh = dot(vs.wo, wh) >= 0.f ? wh : -wh;
        v3 I = -vs.wo;
        wi = I - (wh * dot(wh, I)) * 2.f;             // glm::reflect
        pdf = beckmann_distribution(normal, wh, SVR_ALPHA) / (4.f * __builtin_fabsf(dot(vs.wo, wh)));
        float f = microfacet_brdf_f(wi, vs.wo, normal, SVR_IOR, SVR_ALPHA);
        return ((V3(1.f, 1.f, 1.f) * f) * ks) / p;
    }
    // lambert_brdf_sample_f, lambert.h:20-24 + cosine_weightd_sample_hemisphere, sampling.h:47-56
    float phi = (float)(6.283185307179586 * (double)rng_uniform(rng));
    float sinTheta = __builtin_sqrtf(rng_uniform(rng));
    float cosTheta = __builtin_sqrtf(fmax_(0.f, 1.f - sinTheta * sinTheta));
    wi = frame_direction(normal, sinTheta, cosTheta, phi);
    pdf = __builtin_fabsf(dot(wi, normal)) / SVR_PI_F;
    float f = 1.f / SVR_PI_F;
    return ((V3(vs.color[0], vs.color[1], vs.color[2]) * f) * kd) / (1.f - p);
}

// sample_light, core/lights/light_sample.h:51-68.  Returns false when the reference's
// test `pdf > 0 && max(Li) > 0` (pathtracer.cu:189) fails, i.e. no shadow ray is traced.
SVR_DEV bool sample_light(const DevLight& l, v3 pos, Rng& rng, v3& wi, float& pdf, v3& Li)
{
    float lx, ly;
    uniform_sample_disk(rng, l.radius, lx, ly);
    v3 lightNormal = V3(l.normal[0], l.normal[1], l.normal[2]);
    onb_t onb = onb_from_w(lightNormal);
    v3 lightPos = (V3(l.center[0], l.center[1], l.center[2]) + onb.u * lx) + onb.v * ly;
    v3 shadowVec = lightPos - pos;
    wi = normalize(shadowVec);
    float cosTerm = dot(lightNormal, -wi);
    pdf = dot(shadowVec, shadowVec) / (__builtin_fabsf(cosTerm) * l.area);
    Li = cosTerm > 0.f ? V3(l.radiance[0], l.radiance[1], l.radiance[2]) : V3(0.f, 0.f, 0.f);
    // transmittance() builds its ray from normalize(lightPos - pos): the same value as wi
    return pdf > 0.f && fmax_(Li.x, fmax_(Li.y, Li.z)) > 0.f;
}

// terminate_with_raussian_roulette, pathtracer.cu:96-103 (0.0722 is a double literal)
SVR_DEV bool russian_roulette(v3& T, Rng& rng)
{
    float illum = (float)((double)(0.2126f * T.x + 0.7152f * T.y) + 0.0722 * (double)T.z);
    if (rng_uniform(rng) > illum) return true;
    T = T / illum;
    return false;
}

// reinhard_tone_mapping, core/tonemapping.h:13-27 (gamma = 1/2.2 -> pow(c, 1/(1/2.2)))
SVR_DEV float tonemap_channel(float L, float exposure)
{
    float gamma = 1.f / 2.2f;
    float invGamma = 1.f / gamma;
    float l = L * 16.f;
    l = 1.f - expf_(-l * exposure);
    return powf_(l, invGamma);
}

SVR_DEV uint32_t to_u8(float v)
{
    if (!(v > 0.f)) return 0u;
    if (v >= 255.f) return 255u;
    return (uint32_t)v;
}

// hdr_to_ldr body, pathtracer.cu:288-289
SVR_DEV uint32_t tonemap_pixel(v3 L, float exposure)
{
    uint32_t r = to_u8(tonemap_channel(L.x, exposure) * 255);
    uint32_t g = to_u8(tonemap_channel(L.y, exposure) * 255);
    uint32_t b = to_u8(tonemap_channel(L.z, exposure) * 255);
    return r | (g << 8) | (b << 16) | (255u << 24);
}

} // namespace svr
