// svr_math.hpp -- device math for the gfx950 render kernels.
//
// Numeric contract: every transcendental is a fixed sequence of
// IEEE-754 binary32 operations (Cephes-style polynomials, explicit fma), compiled with
// -ffp-contract=off and correctly rounded divide/sqrt, so a path's radiance is a pure
// function of (scene, pixel, frame) and does not depend on how lanes are scheduled.
// The reference builds with -use_fast_math (CMakeLists.txt:9-10); its intrinsics are
// not reproducible off an NVIDIA GPU, so the contract replaces them.
//
// Vector algebra restates the GLM operations the reference uses (glm::dot, normalize,
// cross, reflect, min, max) with their evaluation order made explicit.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define SVR_DEV __device__ __forceinline__

namespace svr {

SVR_DEV uint32_t f2u(float f) { return __float_as_uint(f); }
SVR_DEV float u2f(uint32_t u) { return __uint_as_float(u); }
SVR_DEV float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
// the smallest float above x (x finite or +inf, not NaN): y > x  <=>  y >= next_up(x)
SVR_DEV float next_up(float x)
{
    const uint32_t b = __float_as_uint(x);
    return x > 0.f ? (b == 0x7f800000u ? x : __uint_as_float(b + 1u)) : (x < 0.f ? __uint_as_float(b - 1u) : __uint_as_float(1u));
}

#define SVR_INF_BITS 0x7f800000u
#define SVR_NAN_BITS 0x7fc00000u
#define SVR_FLT_MAX 3.402823466e+38f

#ifdef SVR_FAST_MATH
// OPT-IN fast-math build of the trace kernel (svr_trace_tile_fast.hip, SVR_OPT_FAST_MATH): the hardware's v_log_f32 for
// the logarithms (the Woodcock walk's -log(1 - xi) is the dominant transcendental: one per iteration) and, through the
// TU's compile flags, reciprocal-based division and fma contraction -- in the spirit of the reference's nvcc
// -use_fast_math (CMakeLists.txt:9-10).  Level 2 (SVR_FAST_MATH >= 2) also swaps exp / sin / cos / pow for v_exp_f32 /
// v_sin_f32 / v_cos_f32; it is not built: the shorter code lets the scheduler overlap more and spills ~50 VGPRs (-38 %).
// Results are no longer bit-identical to the oracle; the contract for this mode is the converged-image tolerance of
// tests/test_fast_math_gpu.py.
SVR_DEV float logf_(float x) { return __builtin_amdgcn_logf(x) * 0.6931471805599453f; }
SVR_DEV float logf_unit(float x) { return __builtin_amdgcn_logf(x) * 0.6931471805599453f; }

#else
SVR_DEV float logf_(float x)
{
    if (x != x) return x;
    if (x < 0.f) return u2f(SVR_NAN_BITS);
    if (x == 0.f) return u2f(0xff800000u);
    uint32_t ix = f2u(x);
    if (ix == SVR_INF_BITS) return x;
    int e = 0;
    if (ix < 0x00800000u) {
        x = x * 8388608.f;
        ix = f2u(x);
        e = -23;
    }
    e += (int)(ix >> 23) - 127;
    float m = u2f((ix & 0x007fffffu) | 0x3f800000u);
    if (m > 1.41421356f) { m = m * 0.5f; e += 1; }
    float y = m - 1.f;
    float z = y * y;
    float p = 7.0376836292E-2f;
    p = fma_(p, y, -1.1514610310E-1f);
    p = fma_(p, y, 1.1676998740E-1f);
    p = fma_(p, y, -1.2420140846E-1f);
    p = fma_(p, y, 1.4249322787E-1f);
    p = fma_(p, y, -1.6668057665E-1f);
    p = fma_(p, y, 2.0000714765E-1f);
    p = fma_(p, y, -2.4999993993E-1f);
    p = fma_(p, y, 3.3333331174E-1f);
    p = (p * y) * z;
    float fe = (float)e;
    p = fma_(fe, -2.12194440e-4f, p);
    p = fma_(-0.5f, z, p);
    float r = y + p;
    r = fma_(fe, 0.693359375f, r);
    return r;
}

// log of (1 - u) for u in (0,1]: the argument is in [0,1), never NaN/negative/inf/subnormal
// unless it is exactly 0.  Same value as logf_ on that domain, fewer branches.
SVR_DEV float logf_unit(float x)
{
    uint32_t ix = f2u(x);
    int e = (int)(ix >> 23) - 127;
    float m = u2f((ix & 0x007fffffu) | 0x3f800000u);
    bool hi = m > 1.41421356f;
    m = hi ? m * 0.5f : m;
    e = hi ? e + 1 : e;
    float y = m - 1.f;
    float z = y * y;
    float p = 7.0376836292E-2f;
    p = fma_(p, y, -1.1514610310E-1f);
    p = fma_(p, y, 1.1676998740E-1f);
    p = fma_(p, y, -1.2420140846E-1f);
    p = fma_(p, y, 1.4249322787E-1f);
    p = fma_(p, y, -1.6668057665E-1f);
    p = fma_(p, y, 2.0000714765E-1f);
    p = fma_(p, y, -2.4999993993E-1f);
    p = fma_(p, y, 3.3333331174E-1f);
    p = (p * y) * z;
    float fe = (float)e;
    p = fma_(fe, -2.12194440e-4f, p);
    p = fma_(-0.5f, z, p);
    float r = y + p;
    r = fma_(fe, 0.693359375f, r);
    // x == 0 -> -inf; subnormal x cannot occur (1-u is a multiple of 2^-25 or 0)
    return x == 0.f ? u2f(0xff800000u) : r;
}
#endif // SVR_FAST_MATH (logarithms)

#if defined(SVR_FAST_MATH) && SVR_FAST_MATH >= 2
SVR_DEV float expf_(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }
SVR_DEV void sincosf_(float x, float* s_out, float* c_out)
{
    const float r = x * 0.15915494309189535f;          // v_sin_f32 / v_cos_f32 take revolutions
    *s_out = __builtin_amdgcn_sinf(r);
    *c_out = __builtin_amdgcn_cosf(r);
}
SVR_DEV float powf_(float x, float y) { return y == 0.f ? 1.f : __builtin_amdgcn_exp2f(y * __builtin_amdgcn_logf(x)); }
#else

SVR_DEV float expf_(float x)
{
    if (x != x) return x;
    if (x > 88.72283905f) return u2f(SVR_INF_BITS);
    if (x < -86.6f) return 0.f;
    float fx = __builtin_floorf(fma_(x, 1.44269504088896341f, 0.5f));
    float r = fma_(fx, -0.693359375f, x);
    r = fma_(fx, 2.12194440e-4f, r);
    float z = r * r;
    float p = 1.9875691500E-4f;
    p = fma_(p, r, 1.3981999507E-3f);
    p = fma_(p, r, 8.3334519073E-3f);
    p = fma_(p, r, 4.1665795894E-2f);
    p = fma_(p, r, 1.6666665459E-1f);
    p = fma_(p, r, 5.0000001201E-1f);
    float y = fma_(p, z, r) + 1.f;
    int n = (int)fx;
    int n1 = n >> 1;
    int n2 = n - n1;
    y = y * u2f((uint32_t)(n1 + 127) << 23);
    y = y * u2f((uint32_t)(n2 + 127) << 23);
    return y;
}

// sin and cos of the same argument share the range reduction (|x| < 8192 for full accuracy)
SVR_DEV void sincosf_(float x, float* s_out, float* c_out)
{
    if (x != x || __builtin_fabsf(x) == u2f(SVR_INF_BITS)) {
        *s_out = u2f(SVR_NAN_BITS);
        *c_out = u2f(SVR_NAN_BITS);
        return;
    }
    bool sneg = x < 0.f;
    float ax = __builtin_fabsf(x);
    int j = (int)(ax * 1.27323954473516f);
    float y = (float)j;
    if (j & 1) { j += 1; y += 1.f; }
    float r = fma_(y, -0.78515625f, ax);
    r = fma_(y, -2.4187564849853515625e-4f, r);
    r = fma_(y, -3.77489497744594108e-8f, r);
    float z = r * r;
    float s = -1.9515295891E-4f;
    s = fma_(s, z, 8.3321608736E-3f);
    s = fma_(s, z, -1.6666654611E-1f);
    s = fma_(s * z, r, r);
    float c = 2.443315711809948E-005f;
    c = fma_(c, z, -1.388731625493765E-003f);
    c = fma_(c, z, 4.166664568298827E-002f);
    c = fma_(c * z, z, fma_(-0.5f, z, 1.f));
    j &= 7;
    bool cneg = false;
    if (j > 3) { sneg = !sneg; cneg = !cneg; j -= 4; }
    if (j > 1) cneg = !cneg;
    bool swap = (j == 1 || j == 2);
    float sv = swap ? c : s;
    float cv = swap ? s : c;
    *s_out = sneg ? -sv : sv;
    *c_out = cneg ? -cv : cv;
}

SVR_DEV float powf_(float x, float y)
{
    if (x != x || y != y) return u2f(SVR_NAN_BITS);
    if (y == 0.f) return 1.f;
    if (x == 0.f) return y > 0.f ? 0.f : u2f(SVR_INF_BITS);
    if (x < 0.f) return u2f(SVR_NAN_BITS);
    return expf_(y * logf_(x));
}
#endif // SVR_FAST_MATH

SVR_DEV float asinf_core(float a)
{
    bool flag = false;
    float z, w;
    if (a > 0.5f) { z = 0.5f * (1.f - a); w = __builtin_sqrtf(z); flag = true; }
    else { w = a; z = w * w; }
    float p = 4.2163199048E-2f;
    p = fma_(p, z, 2.4181311049E-2f);
    p = fma_(p, z, 4.5470025998E-2f);
    p = fma_(p, z, 7.4953002686E-2f);
    p = fma_(p, z, 1.6666752422E-1f);
    p = fma_(p * z, w, w);
    if (flag) { p = p + p; p = 1.5707963267948966192f - p; }
    return p;
}

SVR_DEV float acosf_(float x)
{
    if (x != x) return x;
    if (x < -1.f || x > 1.f) return u2f(SVR_NAN_BITS);
    if (x < -0.5f) return 3.14159265358979323846f - 2.f * asinf_core(__builtin_sqrtf(0.5f * (1.f + x)));
    if (x > 0.5f) return 2.f * asinf_core(__builtin_sqrtf(0.5f * (1.f - x)));
    float a = __builtin_fabsf(x);
    float as = asinf_core(a);
    if (x < 0.f) as = -as;
    return 1.5707963267948966192f - as;
}

SVR_DEV float atanf_pos(float x)
{
    float y;
    if (x > 2.414213562373095f) { y = 1.5707963267948966192f; x = -(1.f / x); }
    else if (x > 0.4142135623730950f) { y = 0.7853981633974483096f; x = (x - 1.f) / (x + 1.f); }
    else y = 0.f;
    float z = x * x;
    float p = 8.05374449538e-2f;
    p = fma_(p, z, -1.38776856032E-1f);
    p = fma_(p, z, 1.99777106478E-1f);
    p = fma_(p, z, -3.33329491539E-1f);
    p = fma_(p * z, x, x);
    return y + p;
}

SVR_DEV float atan2f_(float y, float x)
{
    const float PI = 3.14159265358979323846f, PIO2 = 1.5707963267948966192f;
    if (x != x || y != y) return u2f(SVR_NAN_BITS);
    if (x == 0.f) {
        if (y == 0.f) return 0.f;
        return y > 0.f ? PIO2 : -PIO2;
    }
    if (y == 0.f) return x > 0.f ? 0.f : PI;
    float a = atanf_pos(__builtin_fabsf(y / x));
    if (x < 0.f) a = PI - a;
    return y < 0.f ? -a : a;
}

// ---- GLM vector algebra, evaluation order explicit ----
struct v3 { float x, y, z; };
SVR_DEV v3 V3(float x, float y, float z) { v3 r; r.x = x; r.y = y; r.z = z; return r; }
SVR_DEV v3 operator+(v3 a, v3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
SVR_DEV v3 operator-(v3 a, v3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
SVR_DEV v3 operator*(v3 a, v3 b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
SVR_DEV v3 operator*(v3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
SVR_DEV v3 operator/(v3 a, float s) { return V3(a.x / s, a.y / s, a.z / s); }
SVR_DEV v3 operator-(v3 a) { return V3(-a.x, -a.y, -a.z); }
SVR_DEV float dot(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
SVR_DEV v3 normalize(v3 a) { float s = 1.f / __builtin_sqrtf(dot(a, a)); return a * s; }
SVR_DEV v3 cross(v3 x, v3 y)
{
    return V3(x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y);
}
SVR_DEV float gmin(float a, float b) { return (b < a) ? b : a; }
SVR_DEV float gmax(float a, float b) { return (a < b) ? b : a; }
SVR_DEV float fmin_(float a, float b) { return __builtin_fminf(a, b); }
SVR_DEV float fmax_(float a, float b) { return __builtin_fmaxf(a, b); }
SVR_DEV float rsqrtf_(float x) { return 1.f / __builtin_sqrtf(x); }
SVR_DEV float lerpf(float p, float q, float t) { return fma_(t, q - p, p); }

// cudaONB::InitFromW, core/cuda_onb.h:26-40
struct onb_t { v3 u, v, w; };
SVR_DEV onb_t onb_from_w(v3 w)
{
    onb_t o;
    o.w = w;
    if (__builtin_fabsf(w.x) > __builtin_fabsf(w.y)) {
        float inv = rsqrtf_(w.x * w.x + w.z * w.z);
        o.v = V3(-w.z * inv, 0.f, w.x * inv);
    } else {
        float inv = rsqrtf_(w.y * w.y + w.z * w.z);
        o.v = V3(0.f, w.z * inv, -w.y * inv);
    }
    o.u = cross(o.v, o.w);
    return o;
}

} // namespace svr
