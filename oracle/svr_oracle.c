/*
 * svr_oracle.c -- CPU ORACLE (plain C) for the SunVolumeRender render path.
 * TEST INFRASTRUCTURE ONLY -- see svr_oracle.h for the rules and for what is and is
 * not pinned ("parity unpinned": no reference fixtures exist; reference unbuildable here).
 *
 * Every function cites the reference file:line it restates (paths relative to the
 * reference checkout).  Operation order is the reference's, written out explicitly:
 * C promotes exactly like the C++ the reference is written in, so double literals
 * (M_PI, 1e-6, 0.0722, ...) promote here wherever they promote there.
 *
 * Build: see oracle/Makefile (-O2 -ffp-contract=off -fno-fast-math -fopenmp).
 */
#include "svr_oracle.h"

#include <float.h>
#include <math.h>
#include <string.h>
#include <stdlib.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif
#ifndef M_1_PI
#define M_1_PI 0.31830988618379067154
#endif

/* ------------------------------------------------------------------ */
/* bit casts                                                            */
/* ------------------------------------------------------------------ */
static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* ------------------------------------------------------------------ */
/* contract (iii): libm as fixed IEEE binary32 sequences                */
/* Cephes single-precision polynomials (public domain, S. Moshier),    */
/* evaluated with explicit fmaf.  Same sequences in csrc/svr_math.hpp. */
/* ------------------------------------------------------------------ */
float svo_logf(float x)
{
    if (x != x) return x;
    if (x < 0.f) return u2f(0x7fc00000u);
    if (x == 0.f) return u2f(0xff800000u);          /* -inf */
    uint32_t ix = f2u(x);
    if (ix == 0x7f800000u) return x;                /* +inf */
    int e = 0;
    if (ix < 0x00800000u) {                         /* subnormal: scale by 2^23 */
        x = x * 8388608.f;
        ix = f2u(x);
        e = -23;
    }
    e += (int)(ix >> 23) - 127;
    float m = u2f((ix & 0x007fffffu) | 0x3f800000u); /* [1,2) */
    if (m > 1.41421356f) { m = m * 0.5f; e += 1; }
    float y = m - 1.f;
    float z = y * y;
    float p = 7.0376836292E-2f;
    p = fmaf(p, y, -1.1514610310E-1f);
    p = fmaf(p, y, 1.1676998740E-1f);
    p = fmaf(p, y, -1.2420140846E-1f);
    p = fmaf(p, y, 1.4249322787E-1f);
    p = fmaf(p, y, -1.6668057665E-1f);
    p = fmaf(p, y, 2.0000714765E-1f);
    p = fmaf(p, y, -2.4999993993E-1f);
    p = fmaf(p, y, 3.3333331174E-1f);
    p = (p * y) * z;
    float fe = (float)e;
    p = fmaf(fe, -2.12194440e-4f, p);
    p = fmaf(-0.5f, z, p);
    float r = y + p;
    r = fmaf(fe, 0.693359375f, r);
    return r;
}

float svo_expf(float x)
{
    if (x != x) return x;
    if (x > 88.72283905f) return u2f(0x7f800000u);
    if (x < -86.6f) return 0.f;                     /* results below ~2^-125 flush to 0 (contract) */
    float fx = floorf(fmaf(x, 1.44269504088896341f, 0.5f));
    float r = fmaf(fx, -0.693359375f, x);
    r = fmaf(fx, 2.12194440e-4f, r);
    float z = r * r;
    float p = 1.9875691500E-4f;
    p = fmaf(p, r, 1.3981999507E-3f);
    p = fmaf(p, r, 8.3334519073E-3f);
    p = fmaf(p, r, 4.1665795894E-2f);
    p = fmaf(p, r, 1.6666665459E-1f);
    p = fmaf(p, r, 5.0000001201E-1f);
    float y = fmaf(p, z, r) + 1.f;
    int n = (int)fx;
    int n1 = n >> 1;                                /* arithmetic shift: floor(n/2) */
    int n2 = n - n1;
    y = y * u2f((uint32_t)(n1 + 127) << 23);
    y = y * u2f((uint32_t)(n2 + 127) << 23);
    return y;
}

/* shared range reduction + both polynomials; valid to full accuracy for |x| < 8192 */
static void svo_sincos_core(float ax, int* oct, float* ps, float* pc)
{
    int j = (int)(ax * 1.27323954473516f);          /* 4/pi */
    float y = (float)j;
    if (j & 1) { j += 1; y += 1.f; }
    float r = fmaf(y, -0.78515625f, ax);
    r = fmaf(y, -2.4187564849853515625e-4f, r);
    r = fmaf(y, -3.77489497744594108e-8f, r);
    float z = r * r;
    float s = -1.9515295891E-4f;
    s = fmaf(s, z, 8.3321608736E-3f);
    s = fmaf(s, z, -1.6666654611E-1f);
    s = fmaf(s * z, r, r);
    float c = 2.443315711809948E-005f;
    c = fmaf(c, z, -1.388731625493765E-003f);
    c = fmaf(c, z, 4.166664568298827E-002f);
    c = fmaf(c * z, z, fmaf(-0.5f, z, 1.f));
    *oct = j & 7;
    *ps = s;
    *pc = c;
}

float svo_sinf(float x)
{
    if (x != x || fabsf(x) == u2f(0x7f800000u)) return u2f(0x7fc00000u);
    int neg = x < 0.f;
    float ax = fabsf(x);
    int j; float s, c;
    svo_sincos_core(ax, &j, &s, &c);
    if (j > 3) { neg = !neg; j -= 4; }
    float r = (j == 1 || j == 2) ? c : s;
    return neg ? -r : r;
}

float svo_cosf(float x)
{
    if (x != x || fabsf(x) == u2f(0x7f800000u)) return u2f(0x7fc00000u);
    float ax = fabsf(x);
    int neg = 0;
    int j; float s, c;
    svo_sincos_core(ax, &j, &s, &c);
    if (j > 3) { neg = !neg; j -= 4; }
    if (j > 1) neg = !neg;
    float r = (j == 1 || j == 2) ? s : c;
    return neg ? -r : r;
}

float svo_powf(float x, float y)
{
    if (x != x || y != y) return u2f(0x7fc00000u);
    if (y == 0.f) return 1.f;
    if (x == 0.f) return y > 0.f ? 0.f : u2f(0x7f800000u);
    if (x < 0.f) return u2f(0x7fc00000u);
    return svo_expf(y * svo_logf(x));
}

static float svo_asinf_core(float a)               /* a in [0,1] */
{
    int flag = 0;
    float z, w;
    if (a > 0.5f) { z = 0.5f * (1.f - a); w = sqrtf(z); flag = 1; }
    else { w = a; z = w * w; }
    float p = 4.2163199048E-2f;
    p = fmaf(p, z, 2.4181311049E-2f);
    p = fmaf(p, z, 4.5470025998E-2f);
    p = fmaf(p, z, 7.4953002686E-2f);
    p = fmaf(p, z, 1.6666752422E-1f);
    p = fmaf(p * z, w, w);
    if (flag) { p = p + p; p = 1.5707963267948966192f - p; }
    return p;
}

float svo_acosf(float x)
{
    if (x != x) return x;
    if (x < -1.f || x > 1.f) return u2f(0x7fc00000u);
    if (x < -0.5f) return 3.14159265358979323846f - 2.f * svo_asinf_core(sqrtf(0.5f * (1.f + x)));
    if (x > 0.5f) return 2.f * svo_asinf_core(sqrtf(0.5f * (1.f - x)));
    float a = fabsf(x);
    float as = svo_asinf_core(a);
    if (x < 0.f) as = -as;
    return 1.5707963267948966192f - as;
}

static float svo_atanf_pos(float x)                 /* x >= 0 */
{
    float y;
    if (x > 2.414213562373095f) { y = 1.5707963267948966192f; x = -(1.f / x); }
    else if (x > 0.4142135623730950f) { y = 0.7853981633974483096f; x = (x - 1.f) / (x + 1.f); }
    else y = 0.f;
    float z = x * x;
    float p = 8.05374449538e-2f;
    p = fmaf(p, z, -1.38776856032E-1f);
    p = fmaf(p, z, 1.99777106478E-1f);
    p = fmaf(p, z, -3.33329491539E-1f);
    p = fmaf(p * z, x, x);
    return y + p;
}

float svo_atan2f(float y, float x)
{
    const float PI = 3.14159265358979323846f, PIO2 = 1.5707963267948966192f;
    if (x != x || y != y) return u2f(0x7fc00000u);
    if (x == 0.f) {
        if (y == 0.f) return 0.f;
        return y > 0.f ? PIO2 : -PIO2;
    }
    if (y == 0.f) return x > 0.f ? 0.f : PI;
    float a = svo_atanf_pos(fabsf(y / x));
    if (x < 0.f) a = PI - a;
    return y < 0.f ? -a : a;
}

/* ------------------------------------------------------------------ */
/* contract (ii): cuRAND XORWOW, curand_init(seed, 0, 0)               */
/* call sites: pathtracer.cu:206 (init), :99,:154,:179,:252;           */
/* woodcock_tracking.h:34,43; sampling.h:28-29,50-52;                  */
/* henyey_greenstein.h:32-36; microfacet.h:73-75; cuda_camera.h:68-69  */
/* state[0..4] = v[0..4], state[5] = d                                  */
/* ------------------------------------------------------------------ */
void svo_xorwow_init(uint32_t seed, uint32_t st[6])
{
    /* seed is "unsigned long long" in curand_init; pathtracer.cu:206 passes a
     * 32-bit sum, so the high word is zero. */
    uint32_t s0 = seed ^ 0xaad26b49u;
    uint32_t s1 = 0u ^ 0xf7dcefddu;
    uint32_t t0 = 1099087573u * s0;
    uint32_t t1 = 2591861531u * s1;
    st[5] = 6615241u + t1 + t0;
    st[0] = 123456789u + t0;
    st[1] = 362436069u ^ t0;
    st[2] = 521288629u + t1;
    st[3] = 88675123u ^ t1;
    st[4] = 5783321u + t0;
}

uint32_t svo_xorwow_next(uint32_t st[6])
{
    uint32_t t = st[0] ^ (st[0] >> 2);
    st[0] = st[1];
    st[1] = st[2];
    st[2] = st[3];
    st[3] = st[4];
    st[4] = (st[4] ^ (st[4] << 4)) ^ (t ^ (t << 1));
    st[5] += 362437u;
    return st[4] + st[5];
}

float svo_xorwow_uniform(uint32_t st[6])
{
    /* curand_uniform: x * 2^-32 + 2^-33, in (0, 1] */
    uint32_t x = svo_xorwow_next(st);
    return (float)x * 2.3283064365386963e-10f + 1.1641532182693481e-10f;
}

typedef struct { uint32_t st[6]; svo_counters* c; } rng_t;
static inline float rnd(rng_t* r)
{
    if (r->c) r->c->rng_draws++;
    return svo_xorwow_uniform(r->st);
}

/* pathtracer.cu:70-79 */
uint32_t svo_wang_hash(uint32_t a)
{
    a = (a ^ 61u) ^ (a >> 16);
    a = a + (a << 3);
    a = a ^ (a >> 4);
    a = a * 0x27d4eb2du;
    a = a ^ (a >> 15);
    return a;
}

/* ------------------------------------------------------------------ */
/* GLM restated (vector algebra only)                                   */
/* ------------------------------------------------------------------ */
typedef struct { float x, y, z; } v3;
static inline v3 V(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 vadd(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 vsub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 vmul(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 vscale(v3 a, float s) { return V(a.x * s, a.y * s, a.z * s); }
static inline v3 vdivs(v3 a, float s) { return V(a.x / s, a.y / s, a.z / s); }
static inline v3 vneg(v3 a) { return V(-a.x, -a.y, -a.z); }
static inline float vdot(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline v3 vnormalize(v3 a) { float s = 1.f / sqrtf(vdot(a, a)); return vscale(a, s); }
static inline v3 vcross(v3 x, v3 y)
{
    return V(x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y);
}
static inline float gmin(float a, float b) { return (b < a) ? b : a; }   /* glm::min */
static inline float gmax(float a, float b) { return (a < b) ? b : a; }   /* glm::max */
static inline v3 from3(const svo_vec3* p) { return V(p->x, p->y, p->z); }
static inline float rsqrtf_(float x) { return 1.f / sqrtf(x); }         /* CUDA rsqrtf, defined as exact-rounded 1/sqrt */

/* cuda_onb.h:26-40 */
typedef struct { v3 u, v, w; } onb_t;
static onb_t onb_from_w(v3 w)
{
    onb_t o;
    o.w = w;
    if (fabsf(w.x) > fabsf(w.y)) {
        float inv = rsqrtf_(w.x * w.x + w.z * w.z);
        o.v = V(-w.z * inv, 0.f, w.x * inv);
    } else {
        float inv = rsqrtf_(w.y * w.y + w.z * w.z);
        o.v = V(0.f, w.z * inv, -w.y * inv);
    }
    o.u = vcross(o.v, o.w);
    return o;
}
void svo_onb_from_w(const float w[3], float u[3], float v[3])
{
    onb_t o = onb_from_w(V(w[0], w[1], w[2]));
    u[0] = o.u.x; u[1] = o.u.y; u[2] = o.u.z;
    v[0] = o.v.x; v[1] = o.v.y; v[2] = o.v.z;
}

/* ------------------------------------------------------------------ */
/* contract (i): texture fetches                                        */
/* ------------------------------------------------------------------ */
static inline float lerpf(float p, float q, float t) { return fmaf(t, q - p, p); }

static inline float vox_border(const svo_scene* s, int i, int j, int k)
{
    /* cudaAddressModeBorder (VolumeReader.cpp:161-163): texels outside read 0 */
    if (i < 0 || j < 0 || k < 0 || i >= s->nx || j >= s->ny || k >= s->nz) return 0.f;
    return (float)s->vox[((size_t)k * (size_t)s->ny + (size_t)j) * (size_t)s->nx + (size_t)i];
}

/* tex3D<float>: linear filter, normalized coords, normalized-float read
 * (VolumeReader.cpp:159-167).  Filtering is done on the raw integer values and
 * normalised once by 1/65535. */
float svo_tex3d(const svo_scene* s, float u, float v, float w)
{
    float xb = fmaf(u, (float)s->nx, -0.5f);
    float yb = fmaf(v, (float)s->ny, -0.5f);
    float zb = fmaf(w, (float)s->nz, -0.5f);
    float fx = floorf(xb), fy = floorf(yb), fz = floorf(zb);
    float a = xb - fx, b = yb - fy, g = zb - fz;
    /* clamp the cell index into the all-zero border (keeps int conversion defined) */
    fx = fminf(fmaxf(fx, -2.f), (float)s->nx);
    fy = fminf(fmaxf(fy, -2.f), (float)s->ny);
    fz = fminf(fmaxf(fz, -2.f), (float)s->nz);
    int i = (int)fx, j = (int)fy, k = (int)fz;
    float c00 = lerpf(vox_border(s, i, j, k),         vox_border(s, i + 1, j, k),         a);
    float c10 = lerpf(vox_border(s, i, j + 1, k),     vox_border(s, i + 1, j + 1, k),     a);
    float c01 = lerpf(vox_border(s, i, j, k + 1),     vox_border(s, i + 1, j, k + 1),     a);
    float c11 = lerpf(vox_border(s, i, j + 1, k + 1), vox_border(s, i + 1, j + 1, k + 1), a);
    float c0 = lerpf(c00, c10, b);
    float c1 = lerpf(c01, c11, b);
    return lerpf(c0, c1, g) * 1.5259021896696422e-05f;   /* 1/65535 */
}

/* tex1D<float4>: clamp, linear, normalized coords (transferfunction.cpp:38-42) */
void svo_tex1d(const svo_scene* s, float x, float out[4])
{
    float n = (float)s->tf_n;
    float xb = fmaf(x, n, -0.5f);
    xb = fminf(fmaxf(xb, -1.f), n);                  /* NaN -> -1 -> texel 0 */
    float fx = floorf(xb);
    float a = xb - fx;
    int i0 = (int)fx, i1 = i0 + 1;
    if (i0 < 0) i0 = 0;
    if (i0 > s->tf_n - 1) i0 = s->tf_n - 1;
    if (i1 < 0) i1 = 0;
    if (i1 > s->tf_n - 1) i1 = s->tf_n - 1;
    const float* t0 = s->tf_rgba + 4 * (size_t)i0;
    const float* t1 = s->tf_rgba + 4 * (size_t)i1;
    for (int c = 0; c < 4; ++c) out[c] = lerpf(t0[c], t1[c], a);
}

/* tex2D<float4>: wrap, linear, normalized coords (lights.cpp:60-70) */
void svo_tex2d(const svo_scene* s, float u, float v, float out[4])
{
    float W = (float)s->env_w, H = (float)s->env_h;
    u = u - floorf(u);
    v = v - floorf(v);
    float xb = fmaf(u, W, -0.5f);
    float yb = fmaf(v, H, -0.5f);
    float fx = floorf(xb), fy = floorf(yb);
    float a = xb - fx, b = yb - fy;
    int i0 = (int)fx, j0 = (int)fy;
    int i1 = i0 + 1, j1 = j0 + 1;
    i0 = ((i0 % s->env_w) + s->env_w) % s->env_w;
    i1 = ((i1 % s->env_w) + s->env_w) % s->env_w;
    j0 = ((j0 % s->env_h) + s->env_h) % s->env_h;
    j1 = ((j1 % s->env_h) + s->env_h) % s->env_h;
    const float* t00 = s->env_rgba + 4 * ((size_t)j0 * s->env_w + i0);
    const float* t10 = s->env_rgba + 4 * ((size_t)j0 * s->env_w + i1);
    const float* t01 = s->env_rgba + 4 * ((size_t)j1 * s->env_w + i0);
    const float* t11 = s->env_rgba + 4 * ((size_t)j1 * s->env_w + i1);
    for (int c = 0; c < 4; ++c)
        out[c] = lerpf(lerpf(t00[c], t10[c], a), lerpf(t01[c], t11[c], a), b);
}

/* ------------------------------------------------------------------ */
/* cudaRay (cuda_ray.h:15-42)                                           */
/* ------------------------------------------------------------------ */
typedef struct { v3 orig, dir; float tMin, tMax; } ray_t;
static inline ray_t ray_default(void)
{
    ray_t r;
    r.orig = V(0, 0, 0); r.dir = V(0, 0, 0);
    r.tMin = (float)1e-6;                           /* cuda_ray.h:20 */
    r.tMax = FLT_MAX;
    return r;
}
static inline v3 point_on_ray(const ray_t* r, float t)
{
    return vadd(r->orig, vscale(r->dir, t));        /* orig + t * dir, cuda_ray.h:34 */
}

/* ------------------------------------------------------------------ */
/* cudaVolume (cuda_volume.h) + cudaBBox (cuda_bbox.h)                  */
/* ------------------------------------------------------------------ */
static float volume_intensity(const svo_scene* s, v3 p, svo_counters* c)
{
    /* cuda_volume.h:87-90: (p - vmin) * invSize ; :96 tex3D * densityScale */
    const svo_bbox* b = &s->vol.bbox;
    v3 tc = vmul(vsub(p, from3(&b->vmin)), from3(&b->invSize));
    if (c) c->vol_taps++;
    return svo_tex3d(s, tc.x, tc.y, tc.z) * s->vol.densityScale;
}
float svo_volume_intensity(const svo_scene* s, const float p[3])
{
    return volume_intensity(s, V(p[0], p[1], p[2]), NULL);
}

static v3 volume_gradient(const svo_scene* s, v3 p, svo_counters* c)
{
    /* cuda_volume.h:54-61 */
    v3 sp = from3(&s->vol.spacing);
    float xdiff = volume_intensity(s, vadd(p, V(sp.x, 0.f, 0.f)), c) - volume_intensity(s, vsub(p, V(sp.x, 0.f, 0.f)), c);
    float ydiff = volume_intensity(s, vadd(p, V(0.f, sp.y, 0.f)), c) - volume_intensity(s, vsub(p, V(0.f, sp.y, 0.f)), c);
    float zdiff = volume_intensity(s, vadd(p, V(0.f, 0.f, sp.z)), c) - volume_intensity(s, vsub(p, V(0.f, 0.f, sp.z)), c);
    return vmul(vscale(V(xdiff, ydiff, zdiff), 0.5f), from3(&s->vol.invSpacing));
}
void svo_volume_gradient(const svo_scene* s, const float p[3], float g[3])
{
    v3 r = volume_gradient(s, V(p[0], p[1], p[2]), NULL);
    g[0] = r.x; g[1] = r.y; g[2] = r.z;
}

static int volume_intersect(const svo_scene* s, const ray_t* ray, float* tNear, float* tFar)
{
    /* cuda_bbox.h:33-54, called through cuda_volume.h:49-52 */
    const svo_volume* vol = &s->vol;
    v3 invDir = V(1.f / ray->dir.x, 1.f / ray->dir.y, 1.f / ray->dir.z);
    v3 clip_vmin = vmul(from3(&vol->bbox.vmin), V(-vol->x_clip.x, -vol->y_clip.x, -vol->z_clip.x));
    v3 clip_vmax = vmul(from3(&vol->bbox.vmax), V(vol->x_clip.y, vol->y_clip.y, vol->z_clip.y));
    v3 tbot = vmul(invDir, vsub(clip_vmin, ray->orig));
    v3 ttop = vmul(invDir, vsub(clip_vmax, ray->orig));
    v3 tmin = V(gmin(tbot.x, ttop.x), gmin(tbot.y, ttop.y), gmin(tbot.z, ttop.z));
    v3 tmax = V(gmax(tbot.x, ttop.x), gmax(tbot.y, ttop.y), gmax(tbot.z, ttop.z));
    float largest_tmin = fmaxf(tmin.x, fmaxf(tmin.y, tmin.z));
    float smallest_tmax = fminf(tmax.x, fminf(tmax.y, tmax.z));
    *tNear = largest_tmin;
    *tFar = smallest_tmax;
    return smallest_tmax > largest_tmin;
}
int svo_volume_intersect(const svo_scene* s, const float orig[3], const float dir[3], float* tNear, float* tFar)
{
    ray_t r = ray_default();
    r.orig = V(orig[0], orig[1], orig[2]);
    r.dir = V(dir[0], dir[1], dir[2]);
    return volume_intersect(s, &r, tNear, tFar);
}

/* cuda_transfer_function.h:22-30 */
static void transfer_function(const svo_scene* s, float intensity, float out[4], svo_counters* c)
{
    if (c) c->tf_taps++;
    svo_tex1d(s, intensity, out);
}

/* ------------------------------------------------------------------ */
/* sampling.h                                                           */
/* ------------------------------------------------------------------ */
/* sampling.h:26-32 */
static void uniform_sample_disk(rng_t* rng, float r, float* ox, float* oy)
{
    r *= sqrtf(rnd(rng));
    float theta = (float)(2.f * M_PI * rnd(rng));   /* (2.f*M_PI) is double */
    *ox = svo_cosf(theta) * r;
    *oy = svo_sinf(theta) * r;
}

/* sampling.h:47-56 */
static v3 cosine_weighted_sample_hemisphere(rng_t* rng, v3 n)
{
    onb_t onb = onb_from_w(n);
    float phi = (float)(2.f * M_PI * rnd(rng));
    float sinTheta = sqrtf(rnd(rng));
    float cosTheta = sqrtf(fmaxf(0.f, 1.f - sinTheta * sinTheta));
    v3 d = vadd(vadd(vscale(onb.u, sinTheta * svo_cosf(phi)), vscale(onb.v, sinTheta * svo_sinf(phi))), vscale(onb.w, cosTheta));
    return vnormalize(d);
}

/* ------------------------------------------------------------------ */
/* cudaCamera (cuda_camera.h)                                           */
/* ------------------------------------------------------------------ */
static void camera_generate_ray(const svo_camera* cam, uint32_t x, uint32_t y, rng_t* rng, ray_t* ray)
{
    /* cuda_camera.h:66-83 */
    float nx = 2.f * (((float)x + rnd(rng)) / ((float)cam->imageW - 1.f)) - 1.f;
    float ny = 2.f * (((float)y + rnd(rng)) / ((float)cam->imageH - 1.f)) - 1.f;
    nx = nx * cam->aspectRatio * cam->tanFovxOverTwo;
    ny = ny * cam->tanFovxOverTwo;
    nx = nx * cam->focalLength;
    ny = ny * cam->focalLength;
    float ax, ay;
    uniform_sample_disk(rng, cam->apeture, &ax, &ay);
    v3 u = from3(&cam->u), v = from3(&cam->v), w = from3(&cam->w);
    ray->orig = vadd(vadd(from3(&cam->pos), vscale(u, ax)), vscale(v, ay));
    v3 d = vsub(vadd(vscale(u, nx - ax), vscale(v, ny - ay)), vscale(w, cam->focalLength));
    ray->dir = vnormalize(d);
}
void svo_camera_ray(const svo_scene* s, uint32_t x, uint32_t y, uint32_t st[6], float orig[3], float dir[3])
{
    rng_t rng; memcpy(rng.st, st, sizeof rng.st); rng.c = NULL;
    ray_t r = ray_default();
    camera_generate_ray(&s->cam, x, y, &rng, &r);
    memcpy(st, rng.st, sizeof rng.st);
    orig[0] = r.orig.x; orig[1] = r.orig.y; orig[2] = r.orig.z;
    dir[0] = r.dir.x; dir[1] = r.dir.y; dir[2] = r.dir.z;
}

static void camera_generate_ray_pinhole(const svo_camera* cam, uint32_t x, uint32_t y, ray_t* ray)
{
    /* cuda_camera.h:85-95 */
    float nx = 2.f * (((float)x + 0.5f) / ((float)cam->imageW - 1.f)) - 1.f;
    float ny = 2.f * (((float)y + 0.5f) / ((float)cam->imageH - 1.f)) - 1.f;
    nx = nx * cam->aspectRatio * cam->tanFovxOverTwo;
    ny = ny * cam->tanFovxOverTwo;
    ray->orig = from3(&cam->pos);
    v3 d = vsub(vadd(vscale(from3(&cam->u), nx), vscale(from3(&cam->v), ny)), from3(&cam->w));
    ray->dir = vnormalize(d);
}
void svo_camera_ray_pinhole(const svo_scene* s, uint32_t x, uint32_t y, float orig[3], float dir[3])
{
    ray_t r = ray_default();
    camera_generate_ray_pinhole(&s->cam, x, y, &r);
    orig[0] = r.orig.x; orig[1] = r.orig.y; orig[2] = r.orig.z;
    dir[0] = r.dir.x; dir[1] = r.dir.y; dir[2] = r.dir.z;
}

/* ------------------------------------------------------------------ */
/* lights                                                               */
/* ------------------------------------------------------------------ */
static float disk_area(const svo_disk* d)
{
    return (float)(M_PI * d->radius * d->radius);   /* cuda_disk.h:53-56, double product */
}

static int disk_intersect(const svo_disk* d, const ray_t* ray, float* t)
{
    /* cuda_disk.h:32-51 */
    v3 normal = from3(&d->normal), center = from3(&d->center);
    float denom = vdot(normal, ray->dir);
    if (fabsf(denom) > 1e-6) {                      /* double compare */
        v3 co = vsub(center, ray->orig);
        *t = vdot(co, normal) / denom;
        if (*t >= 0) {
            v3 p = vadd(ray->orig, vscale(ray->dir, *t));
            v3 co2 = vsub(p, center);
            return sqrtf(vdot(co2, co2)) <= d->radius;
        }
        return 0;
    }
    return 0;
}
int svo_disk_intersect(const svo_disk* d, const float orig[3], const float dir[3], float* t)
{
    ray_t r = ray_default();
    r.orig = V(orig[0], orig[1], orig[2]);
    r.dir = V(dir[0], dir[1], dir[2]);
    return disk_intersect(d, &r, t);
}

static v3 light_radiance(const svo_arealight* l)
{
    /* cuda_arealight.h:57: 500.f * color * intensity * float(M_1_PI) / disk.GetArea() */
    v3 r = vscale(from3(&l->color), 500.f);
    r = vscale(r, l->intensity);
    r = vscale(r, (float)M_1_PI);
    return vdivs(r, disk_area(&l->disk));
}
void svo_light_radiance(const svo_arealight* l, float out[3])
{
    v3 r = light_radiance(l);
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}

typedef struct { float t; v3 normal, radiance; } light_sample_t;

static int get_nearest_light_sample(const ray_t* ray, const svo_arealight* lights, uint32_t n, light_sample_t* ls)
{
    /* light_sample.h:23-49 */
    float tNear = FLT_MAX;
    float t = FLT_MAX;
    int id = -1;
    for (uint32_t i = 0; i < n; ++i) {
        if (disk_intersect(&lights[i].disk, ray, &t) && (t < tNear)) {
            tNear = t;
            id = (int)i;
        }
    }
    if (id != -1) {
        ls->t = tNear;
        ls->normal = from3(&lights[id].disk.normal);
        ls->radiance = light_radiance(&lights[id]);
        return 1;
    }
    ls->t = -FLT_MAX;
    return 0;
}

static v3 sample_light(const svo_arealight* light, v3 volSamplePos, rng_t* rng, v3* lightPos, v3* wi, float* pdf)
{
    /* light_sample.h:51-68 */
    float lx, ly;
    uniform_sample_disk(rng, light->disk.radius, &lx, &ly);
    v3 lightNormal = from3(&light->disk.normal);
    onb_t onb = onb_from_w(lightNormal);
    v3 lightCenter = from3(&light->disk.center);
    *lightPos = vadd(vadd(lightCenter, vscale(onb.u, lx)), vscale(onb.v, ly));
    v3 shadowVec = vsub(*lightPos, volSamplePos);
    *wi = vnormalize(shadowVec);
    float cosTerm = vdot(lightNormal, vneg(*wi));
    *pdf = vdot(shadowVec, shadowVec) / (fabsf(cosTerm) * disk_area(&light->disk));
    return cosTerm > 0.f ? light_radiance(light) : V(0.f, 0.f, 0.f);
}

/* cuda_environment_light.h:58-72 (the fetch-before-test quirk at :67-68 is not copied) */
static v3 env_radiance(const svo_scene* s, v3 dir)
{
    const svo_envlight* e = &s->env;
    if (e->tex == 0 || s->env_rgba == NULL)
        return vscale(from3(&e->defaultRadiance), e->intensity);
    float theta = svo_acosf(dir.y);
    float phi = svo_atan2f(dir.x, dir.z);
    phi = (float)(phi < 0.f ? phi + 2.f * M_PI : phi);
    float u = (float)(phi * 0.5f * M_1_PI);
    float v = (float)(theta * M_1_PI);
    float t[4];
    svo_tex2d(s, u + e->offset.x, v + e->offset.y, t);
    return vscale(V(t[0], t[1], t[2]), e->intensity);
}

/* ------------------------------------------------------------------ */
/* woodcock_tracking.h:20-51 and transmittance.h:10-17                  */
/* ------------------------------------------------------------------ */
static float sample_distance(const svo_scene* s, ray_t* ray, rng_t* rng, svo_counters* c)
{
    float tNear, tFar;
    if (volume_intersect(s, ray, &tNear, &tFar)) {
        ray->tMin = (float)(tNear < 0.f ? 1e-6 : tNear);
        ray->tMax = tFar;
        float t = ray->tMin;
        float sigmaMax = s->tf.maxOpacity;
        float invSigmaMax = 1.f / sigmaMax;
        float invSigmaMaxSampleInterval = 1.f / (sigmaMax * 1.f);   /* BASE_SAMPLE_STEP_SIZE 1.f */
        for (;;) {
            if (c) c->woodcock_iters++;
            t += -svo_logf(1.f - rnd(rng)) * invSigmaMaxSampleInterval;
            if (t > ray->tMax)
                return -FLT_MAX;
            v3 p = point_on_ray(ray, t);
            float intensity = volume_intensity(s, p, c);
            float co[4];
            transfer_function(s, intensity, co, c);
            float sigma_t = co[3];
            if (rnd(rng) < sigma_t * invSigmaMax || t > ray->tMax)
                break;
        }
        return t;
    }
    return -FLT_MAX;
}

static float transmittance(const svo_scene* s, v3 start, v3 end, rng_t* rng, svo_counters* c)
{
    ray_t ray = ray_default();
    ray.orig = start;
    ray.dir = vnormalize(vsub(end, start));
    if (c) c->shadow_walks++;
    float t = sample_distance(s, &ray, rng, c);
    int flag = (t > ray.tMin) && (t < ray.tMax);
    return flag ? 0.f : 1.f;
}

/* ------------------------------------------------------------------ */
/* BSDF library: fresnel.h, henyey_greenstein.h, lambert.h, microfacet.h */
/* ------------------------------------------------------------------ */
#define PHASE_FUNC_G (0.f)          /* pathtracer.cu:29 */
#define IOR (2.5f)                  /* pathtracer.cu:30 */
#define ALPHA (0.15f)               /* pathtracer.cu:31 */

float svo_schlick(float ni, float no, float cosin)
{
    /* fresnel.h:10-15 */
    float R0 = (ni - no) * (ni - no) / ((ni + no) * (ni + no));
    float c = 1.f - cosin;
    return R0 + (1.f - R0) * c * c * c * c * c;
}

static float hg_phase_f_iso(void)
{
    /* henyey_greenstein.h:15-18, g == 0: M_1_PI * 0.25f (double) -> float */
    return (float)(M_1_PI * 0.25f);
}

static float beckmann_distribution(v3 normal, v3 wh, float alpha)
{
    /* microfacet.h:18-25 */
    float cosTerm2 = vdot(normal, wh);
    cosTerm2 *= cosTerm2;
    return svo_expf((cosTerm2 - 1.f) / (alpha * alpha * cosTerm2)) / ((float)M_PI * alpha * alpha * cosTerm2 * cosTerm2);
}

static float geometry_cook_torrance(v3 wi, v3 wo, v3 normal, v3 wh)
{
    /* microfacet.h:42-50 */
    float cosO = vdot(wo, wh);
    float cosTerm = vdot(normal, wh);
    float g1 = 2.f * cosTerm * vdot(normal, wo) / cosO;
    float g2 = 2.f * cosTerm * vdot(normal, wi) / cosO;
    return fminf(1.f, fminf(g1, g2));
}

static float microfacet_brdf_f(v3 wi, v3 wo, v3 normal, float ior, float alpha)
{
    /* microfacet.h:52-68, DISTRIBUTION_BECKMANN (:16) */
    if (vdot(wi, normal) * vdot(wo, normal) < 0.f) return 0.f;
    v3 wh = vnormalize(vadd(wi, wo));
    float fresnelTerm = svo_schlick(1.f, ior, fabsf(vdot(wh, wo)));
    float geometryTerm = geometry_cook_torrance(wi, wo, normal, wh);
    float D = beckmann_distribution(normal, wh, alpha);
    return fresnelTerm * geometryTerm * D / (4.f * fabsf(vdot(normal, wi)) * fabsf(vdot(normal, wo)));
}
float svo_microfacet_f(const float wi[3], const float wo[3], const float n[3], float ior, float alpha)
{
    return microfacet_brdf_f(V(wi[0], wi[1], wi[2]), V(wo[0], wo[1], wo[2]), V(n[0], n[1], n[2]), ior, alpha);
}

static v3 sample_beckmann(v3 normal, float alpha, rng_t* rng)
{
    /* microfacet.h:70-79.  log(float) is CUDA's float overload. */
    onb_t onb = onb_from_w(normal);
    float phi = 2.f * (float)M_PI * rnd(rng);
    float cosTheta = 1.f / (1.f - alpha * alpha * svo_logf(1.f - rnd(rng)));
    float sinTheta = sqrtf(fmaxf(0.f, 1.f - cosTheta * cosTheta));
    v3 d = vadd(vadd(vscale(onb.u, sinTheta * svo_cosf(phi)), vscale(onb.v, sinTheta * svo_sinf(phi))), vscale(onb.w, cosTheta));
    return vnormalize(d);
}

static void microfacet_brdf_sample_f(v3 wo, v3 normal, float alpha, v3* wi, float* pdf, rng_t* rng)
{
    /* microfacet.h:95-111 */
    v3 wh = sample_beckmann(normal, alpha, rng);
    wh = vdot(wo, wh) >= 0.f ? wh : vneg(wh);
    /* glm::reflect(I, N) = I - N * dot(N, I) * 2 */
    v3 I = vneg(wo);
    *wi = vsub(I, vscale(vscale(wh, vdot(wh, I)), 2.f));
    *pdf = beckmann_distribution(normal, wh, alpha) / (4.f * fabsf(vdot(wo, wh)));
}

/* VolumeSample, cuda_volume.h:124-132 */
typedef struct {
    v3 ptInWorld, wo;
    float intensity;
    v3 gradient;
    float gradientMagnitude;
    float color_opacity[4];
} volume_sample_t;

enum { ST_ISOTROPIC = 0, ST_BRDF = 1 };            /* pathtracer.cu:105 */

static v3 bsdf(const volume_sample_t* vs, v3 wi, int st)
{
    /* pathtracer.cu:106-131 */
    v3 diffuseColor = V(vs->color_opacity[0], vs->color_opacity[1], vs->color_opacity[2]);
    v3 L = V(0.f, 0.f, 0.f);
    if (st == ST_ISOTROPIC) {
        L = vscale(diffuseColor, hg_phase_f_iso());
    } else {
        v3 normal = vnormalize(vs->gradient);
        normal = vdot(vs->wo, normal) < 0.f ? vneg(normal) : normal;
        float cosTerm = fmaxf(0.f, vdot(wi, normal));
        float ks = svo_schlick(1.0f, IOR, cosTerm);
        float kd = 1.f - ks;
        v3 diffuse = vscale(diffuseColor, 1.f / (float)M_PI);       /* lambert.h:15-18 */
        v3 specular = vscale(V(1.f, 1.f, 1.f), microfacet_brdf_f(wi, vs->wo, normal, IOR, ALPHA));
        L = vscale(vadd(vscale(diffuse, kd), vscale(specular, ks)), cosTerm);
    }
    return L;
}

static v3 sample_bsdf(const volume_sample_t* vs, v3* wi, float* pdf, rng_t* rng, int st)
{
    /* pathtracer.cu:133-169 */
    if (st == ST_ISOTROPIC) {
        /* henyey_greenstein.h:29-51 with g == 0 */
        float phi = (float)(2.f * M_PI * rnd(rng));
        float cosTheta = 1.f - 2.f * rnd(rng);
        float sinTheta = sqrtf(fmaxf(0.f, 1.f - cosTheta * cosTheta));
        onb_t onb = onb_from_w(vs->wo);
        v3 d = vadd(vadd(vscale(onb.u, sinTheta * svo_cosf(phi)), vscale(onb.v, sinTheta * svo_sinf(phi))), vscale(onb.w, cosTheta));
        *wi = vnormalize(d);
        *pdf = hg_phase_f_iso();
        return vscale(V(vs->color_opacity[0], vs->color_opacity[1], vs->color_opacity[2]), hg_phase_f_iso());
    } else {
        v3 normal = vnormalize(vs->gradient);
        float cosTerm = vdot(vs->wo, normal);
        if (cosTerm < 0.f) {
            cosTerm = -cosTerm;
            normal = vneg(normal);
        }
        float ks = svo_schlick(1.f, IOR, cosTerm);
        float kd = 1.f - ks;
        float p = 0.25f + 0.5f * ks;
        if (rnd(rng) < p) {
            microfacet_brdf_sample_f(vs->wo, normal, ALPHA, wi, pdf, rng);
            float f = microfacet_brdf_f(*wi, vs->wo, normal, IOR, ALPHA);
            return vdivs(vscale(vscale(V(1.f, 1.f, 1.f), f), ks), p);
        } else {
            /* lambert.h:20-24 */
            *wi = cosine_weighted_sample_hemisphere(rng, normal);
            *pdf = fabsf(vdot(*wi, normal)) / (float)M_PI;
            float f = 1.f / (float)M_PI;
            v3 col = V(vs->color_opacity[0], vs->color_opacity[1], vs->color_opacity[2]);
            return vdivs(vscale(vscale(col, f), kd), 1.f - p);
        }
    }
}

static v3 estimate_direct_light(const svo_scene* s, const volume_sample_t* vs, rng_t* rng, int st, svo_counters* c)
{
    /* pathtracer.cu:171-198 */
    v3 Li = V(0.f, 0.f, 0.f);
    if (s->num_lights == 0)
        return Li;
    int lightId = (int)((float)s->num_lights * rnd(rng));
    lightId = lightId < (int)s->num_lights ? lightId : (int)s->num_lights - 1;
    const svo_arealight* light = &s->lights[lightId];
    v3 lightPos, wi;
    float pdf;
    Li = sample_light(light, vs->ptInWorld, rng, &lightPos, &wi, &pdf);
    if (pdf > 0.f && fmaxf(Li.x, fmaxf(Li.y, Li.z)) > 0.f) {
        float Tr = transmittance(s, vs->ptInWorld, lightPos, rng, c);
        /* Tr * num_areaLights * bsdf(vs, wi, st) * Li / pdf */
        float k = Tr * (float)s->num_lights;
        Li = vdivs(vmul(vscale(bsdf(vs, wi, st), k), Li), pdf);
    } else
        Li = V(0.f, 0.f, 0.f);
    return Li;
}

static int terminate_with_russian_roulette(v3* throughput, rng_t* rng)
{
    /* pathtracer.cu:96-103; the 0.0722 literal is a double */
    float illum = (float)((double)(0.2126f * throughput->x + 0.7152f * throughput->y) + 0.0722 * (double)throughput->z);
    if (rnd(rng) > illum) return 1;
    *throughput = vdivs(*throughput, illum);
    return 0;
}

/* kernel_pathtracer, pathtracer.cu:200-278 (everything before running_estimate) */
void svo_trace_path(const svo_scene* s, uint32_t idx, uint32_t idy, uint32_t traceDepth,
                    uint32_t hashedFrameNo, float Lout[3], svo_counters* c)
{
    uint32_t offset = idy * s->cam.imageW + idx;     /* WIDTH -> camera.imageW (runtime resolution) */
    rng_t rng;
    rng.c = c;
    svo_xorwow_init(hashedFrameNo + offset, rng.st);
    if (c) c->paths++;

    v3 L = V(0.f, 0.f, 0.f);
    v3 T = V(1.f, 1.f, 1.f);

    ray_t ray = ray_default();
    camera_generate_ray(&s->cam, idx, idy, &rng, &ray);

    light_sample_t ls;
    ls.t = -1.f;                                      /* light_sample.h:18 */
    ls.normal = V(0.f, 0.f, 0.f);
    ls.radiance = V(0.f, 0.f, 0.f);
    int hitLight = get_nearest_light_sample(&ray, s->lights, s->num_lights, &ls);
    for (uint32_t k = 0; k < traceDepth; ++k) {
        float t = sample_distance(s, &ray, &rng, c);

        if ((k == 0) && hitLight) {
            t = t < 0.f ? FLT_MAX : t;
            if (ls.t < t) {
                float cosTerm = vdot(ls.normal, vneg(ray.dir));
                L = vadd(L, vscale(vmul(T, ls.radiance), (cosTerm <= 0.f ? 0.f : 1.f)));
                break;
            }
        }

        if (t < 0.f) {
            /* pathtracer.cu:233 is commented out in the reference; env_on_escape is the documented extension */
            if (s->env_on_escape)
                L = vadd(L, vmul(T, env_radiance(s, ray.dir)));
            break;
        }

        volume_sample_t vs;
        if (c) c->scatter_events++;
        vs.wo = vneg(ray.dir);
        vs.ptInWorld = point_on_ray(&ray, t);
        vs.intensity = volume_intensity(s, vs.ptInWorld, c);
        transfer_function(s, vs.intensity, vs.color_opacity, c);
        vs.gradient = volume_gradient(s, vs.ptInWorld, c);
        vs.gradientMagnitude = sqrtf(vdot(vs.gradient, vs.gradient));

        v3 wi;
        float pdf = 0.f;
        int st;

        float gradientFactor = s->vol.gradientFactor;
        float Pbrdf = vs.color_opacity[3] * (1.f - svo_expf(-25.f * gradientFactor * gradientFactor * gradientFactor * vs.gradientMagnitude * 65535.f * s->vol.invMaxMagnitude));
        if (rnd(&rng) < Pbrdf)
            st = ST_BRDF;
        else
            st = ST_ISOTROPIC;

        L = vadd(L, vmul(T, estimate_direct_light(s, &vs, &rng, st, c)));

        v3 f = sample_bsdf(&vs, &wi, &pdf, &rng, st);
        float cosTerm = fabsf(vdot(vnormalize(vs.gradient), wi));
        if (fmaxf(f.x, fmaxf(f.y, f.z)) > 0.f && pdf > 0.f) {
            if (st == ST_ISOTROPIC)
                T = vmul(T, vdivs(f, pdf * (1.f - Pbrdf)));
            else
                T = vmul(T, vdivs(vscale(f, cosTerm), pdf * Pbrdf));
        }

        ray.orig = vs.ptInWorld;
        ray.dir = wi;

        if (k >= 3) {
            if (terminate_with_russian_roulette(&T, &rng))
                break;
        }
    }
    Lout[0] = L.x; Lout[1] = L.y; Lout[2] = L.z;
}

/* tonemapping.h:13-27 */
void svo_tonemap(const float L[3], float exposure, float out[3])
{
    float gamma = 1.f / 2.2f;
    float invGamma = 1.f / gamma;
    for (int c = 0; c < 3; ++c) {
        float l = L[c] * 16.f;
        l = 1.f - svo_expf(-l * exposure);
        out[c] = svo_powf(l, invGamma);
    }
}

static inline uint8_t to_u8(float v)
{
    /* glm::u8vec4(float...) = static_cast<uint8>(float): truncation.  Out-of-range
     * and NaN are undefined in C++; the contract clamps to [0,255], NaN -> 0. */
    if (!(v > 0.f)) return 0;
    if (v >= 255.f) return 255;
    return (uint8_t)v;
}

/* hdr_to_ldr, pathtracer.cu:282-290 */
void svo_hdr_to_ldr(const svo_scene* s, const float* hdr, uint8_t* img, int x0, int y0, int x1, int y1)
{
    const uint32_t W = s->cam.imageW;
    for (int y = y0; y < y1; ++y)
        for (int x = x0; x < x1; ++x) {
            size_t off = (size_t)y * W + (size_t)x;
            float l[3];
            svo_tonemap(hdr + 3 * off, s->cam.exposure, l);
            img[4 * off + 0] = to_u8(l[0] * 255);
            img[4 * off + 1] = to_u8(l[1] * 255);
            img[4 * off + 2] = to_u8(l[2] * 255);
            img[4 * off + 3] = 255;
        }
}

static void counters_add(svo_counters* a, const svo_counters* b)
{
    a->paths += b->paths; a->vol_taps += b->vol_taps; a->tf_taps += b->tf_taps;
    a->rng_draws += b->rng_draws; a->woodcock_iters += b->woodcock_iters;
    a->scatter_events += b->scatter_events; a->shadow_walks += b->shadow_walks;
    a->raycast_steps += b->raycast_steps;
}

/* render_pathtracer, pathtracer.cu:292-304 */
void svo_render_pathtracer(const svo_scene* s, float* hdr, uint8_t* img,
                           uint32_t traceDepth, uint32_t frameNo,
                           int x0, int y0, int x1, int y1,
                           svo_counters* counters, int nthreads)
{
    const uint32_t W = s->cam.imageW;
    const uint32_t hashed = svo_wang_hash(frameNo);
    svo_counters total;
    memset(&total, 0, sizeof total);
#ifdef _OPENMP
    if (nthreads <= 0) nthreads = omp_get_max_threads();
#else
    nthreads = 1;
#endif
#pragma omp parallel num_threads(nthreads)
    {
        svo_counters local;
        memset(&local, 0, sizeof local);
#pragma omp for collapse(2) schedule(dynamic, 64)
        for (int y = y0; y < y1; ++y) {
            for (int x = x0; x < x1; ++x) {
                size_t off = (size_t)y * W + (size_t)x;
                float* acc = hdr + 3 * off;
                if (frameNo == 0) { acc[0] = 0.f; acc[1] = 0.f; acc[2] = 0.f; }   /* clear_hdr_buffer, pathtracer.cu:86-94,297-300 */
                float L[3];
                svo_trace_path(s, (uint32_t)x, (uint32_t)y, traceDepth, hashed, L, counters ? &local : NULL);
                /* running_estimate, pathtracer.cu:81-84 */
                float n1 = (float)frameNo + 1.f;
                acc[0] += (L[0] - acc[0]) / n1;
                acc[1] += (L[1] - acc[1]) / n1;
                acc[2] += (L[2] - acc[2]) / n1;
            }
        }
#pragma omp critical
        counters_add(&total, &local);
    }
    if (counters) counters_add(counters, &total);
    if (img) svo_hdr_to_ldr(s, hdr, img, x0, y0, x1, y1);
}

/* kernel_raycasting, raycasting.cu:15-67 */
static void raycast_pixel(const svo_scene* s, uint32_t idx, uint32_t idy, float stepSize, uint8_t out[4], svo_counters* c)
{
    ray_t ray = ray_default();
    camera_generate_ray_pinhole(&s->cam, idx, idy, &ray);
    float L[4] = {0.f, 0.f, 0.f, 0.f};
    float tNear, tFar, t;
    if (volume_intersect(s, &ray, &tNear, &tFar)) {
        t = tNear;
        while (t <= tFar) {
            if (c) c->raycast_steps++;
            v3 p = point_on_ray(&ray, t);
            float intensity = volume_intensity(s, p, c);
            float co[4];
            transfer_function(s, intensity, co, c);
            v3 gradient = volume_gradient(s, p, c);
            float gradientMagnitude = sqrtf(vdot(gradient, gradient));
            float cosTerm = 1.f;
            float specularTerm = 0.f;
            if (gradientMagnitude > 1e-3) {          /* double compare */
                v3 normal = vnormalize(gradient);
                v3 lightDir = vnormalize(vsub(from3(&s->cam.pos), p));
                cosTerm = fabsf(vdot(normal, lightDir));
                specularTerm = svo_powf(cosTerm, 30.f);
            }
            co[0] = co[0] * co[3] * cosTerm * 0.8f + co[3] * specularTerm * 0.2f;
            co[1] = co[1] * co[3] * cosTerm * 0.8f + co[3] * specularTerm * 0.2f;
            co[2] = co[2] * co[3] * cosTerm * 0.8f + co[3] * specularTerm * 0.2f;
            float w = 1.f - L[3];
            L[0] += w * co[0]; L[1] += w * co[1]; L[2] += w * co[2]; L[3] += w * co[3];
            if (L[3] > 0.95f) break;
            t += stepSize * 0.5f;
        }
    }
    L[0] = fminf(L[0], 1.f);
    L[1] = fminf(L[1], 1.f);
    L[2] = fminf(L[2], 1.f);
    out[0] = to_u8(L[0] * 255);
    out[1] = to_u8(L[1] * 255);
    out[2] = to_u8(L[2] * 255);
    out[3] = to_u8(255 * L[3]);
}

void svo_render_raycasting(const svo_scene* s, uint8_t* img, float stepSize,
                           int x0, int y0, int x1, int y1, svo_counters* counters, int nthreads)
{
    const uint32_t W = s->cam.imageW;
    svo_counters total;
    memset(&total, 0, sizeof total);
#ifdef _OPENMP
    if (nthreads <= 0) nthreads = omp_get_max_threads();
#else
    nthreads = 1;
#endif
#pragma omp parallel num_threads(nthreads)
    {
        svo_counters local;
        memset(&local, 0, sizeof local);
#pragma omp for collapse(2) schedule(dynamic, 16)
        for (int y = y0; y < y1; ++y)
            for (int x = x0; x < x1; ++x) {
                local.paths++;
                raycast_pixel(s, (uint32_t)x, (uint32_t)y, stepSize, img + 4 * ((size_t)y * W + (size_t)x), counters ? &local : NULL);
            }
#pragma omp critical
        counters_add(&total, &local);
    }
    if (counters) counters_add(counters, &total);
}

int svo_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

int svo_sizeof(int which)
{
    switch (which) {
    case 0: return (int)sizeof(svo_scene);
    case 1: return (int)sizeof(svo_volume);
    case 2: return (int)sizeof(svo_tf);
    case 3: return (int)sizeof(svo_camera);
    case 4: return (int)sizeof(svo_arealight);
    case 5: return (int)sizeof(svo_envlight);
    case 6: return (int)sizeof(svo_counters);
    default: return -1;
    }
}
