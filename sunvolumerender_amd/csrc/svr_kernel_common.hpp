// svr_kernel_common.hpp -- pieces shared by the kernel translation units: LDS-resident transfer
// function, pixel/work enumeration, counters.
#pragma once
#include "svr_kernels.hpp"
#include "svr_device.hpp"

namespace svr {

#define SVR_TF_MAX 1024
// Hang guard, not part of the algorithm: a single Woodcock walk is abandoned (treated as leaving the
// volume) after 2^20 iterations.  Unreachable for sane scenes (expected iterations = sigma_max x chord
// length, ~10^2..10^3); it only bounds kernels fed degenerate majorants so a launch always drains.
#define SVR_WALK_GUARD (1u << 20)
#define SVR_TF_PAD 3

// ------------------------------------------------------------------------------------------
// LDS-resident transfer function.  Entry e of the padded tables holds texel clamp(e-1), so the
// clamp addressing of tex1D becomes plain adjacent reads (ds_read2_b32 for the alpha pair).
// ------------------------------------------------------------------------------------------
struct LdsTF {
    float4 rgba[SVR_TF_MAX + SVR_TF_PAD];
    float alpha[SVR_TF_MAX + SVR_TF_PAD];
};

SVR_DEV void lds_tf_load(LdsTF& L, const DevScene& s)
{
    const int n = s.tf_n;
    const float4* g = reinterpret_cast<const float4*>(s.tf);
    for (int e = threadIdx.x; e < n + SVR_TF_PAD; e += blockDim.x) {
        int t = min(max(e - 1, 0), n - 1);
        float4 v = g[t];
        L.rgba[e] = v;
        L.alpha[e] = v.w;
    }
    __syncthreads();
}

SVR_DEV void lds_tf_coord(const DevScene& s, float x, int& e, float& a)
{
    float xb = fma_(x, s.tf_nf, -0.5f);
    xb = fmin_(fmax_(xb, -1.f), s.tf_nf);
    float fx = __builtin_floorf(xb);
    a = xb - fx;
    e = (int)fx + 1;          // in [0, n+1]
}

SVR_DEV float lds_tf_alpha(const LdsTF& L, const DevScene& s, float x)
{
    int e; float a;
    lds_tf_coord(s, x, e, a);
    float t0 = L.alpha[e], t1 = L.alpha[e + 1];
    return lerpf(t0, t1, a);
}

SVR_DEV void lds_tf_rgba(const LdsTF& L, const DevScene& s, float x, float out[4])
{
    int e; float a;
    lds_tf_coord(s, x, e, a);
    float4 t0 = L.rgba[e], t1 = L.rgba[e + 1];
    out[0] = lerpf(t0.x, t1.x, a);
    out[1] = lerpf(t0.y, t1.y, a);
    out[2] = lerpf(t0.z, t1.z, a);
    out[3] = lerpf(t0.w, t1.w, a);
}

// ------------------------------------------------------------------------------------------
// pixel enumeration: owned rows (window or interleaved row strips), 8x8 tiles, 64 items per tile
// ------------------------------------------------------------------------------------------
SVR_DEV uint32_t owned_row_to_y(const DevWork& w, uint32_t r)
{
    if (w.world <= 1u) return w.y0 + r;
    uint32_t q = r / w.strip_rows;
    return (q * w.world + w.rank) * w.strip_rows + (r - q * w.strip_rows);
}

// item -> (pixel, frame slot).  Items enumerate wave-tasks tile-major: task = tile * nframes + slot,
// 64 items (one 8x8 tile) per task, so concurrently running waves work on neighbouring tubes of the
// volume.  false for the padding items of partial tiles.
SVR_DEV bool item_to_pixel(const DevWork& w, uint32_t item, uint32_t& x, uint32_t& y, uint32_t& slot)
{
    uint32_t wv = w.x1 - w.x0;
    uint32_t tiles_x = (wv + 7u) >> 3;
    uint32_t task = item >> 6, in = item & 63u;
    uint32_t tile = task / w.nframes;
    slot = task - tile * w.nframes;
    uint32_t ty = tile / tiles_x, tx = tile - ty * tiles_x;
    uint32_t px = (tx << 3) + (in & 7u);
    uint32_t r = (ty << 3) + (in >> 3);
    if (px >= wv || r >= w.n_rows) return false;
    x = w.x0 + px;
    y = owned_row_to_y(w, r);
    return true;
}

SVR_DEV unsigned long long wave_sum(unsigned long long v)
{
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

struct Cnt { uint32_t taps, iters, scatter, shadow, paths, loops, exec, wskip, iskip, ipre, cull; };

SVR_DEV void cnt_flush(const DevWork& w, const Cnt& c)
{
    unsigned long long a = wave_sum(c.taps), b = wave_sum(c.iters), d = wave_sum(c.scatter),
                       e = wave_sum(c.shadow), f = wave_sum(c.paths), x = wave_sum(c.exec);
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&w.counters[CNT_VOL_TAPS], a);
        atomicAdd(&w.counters[CNT_WOODCOCK], b);
        atomicAdd(&w.counters[CNT_SCATTER], d);
        atomicAdd(&w.counters[CNT_SHADOW], e);
        atomicAdd(&w.counters[CNT_PATHS], f);
        atomicAdd(&w.counters[CNT_LOOP], (unsigned long long)c.loops);
        atomicAdd(&w.counters[CNT_TAPS_EXEC], x);
    }
    unsigned long long ws = wave_sum(c.wskip), is = wave_sum(c.iskip), ip = wave_sum(c.ipre), cu = wave_sum(c.cull);
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&w.counters[CNT_WALKS_RAYSKIP], ws);
        atomicAdd(&w.counters[CNT_ITERS_RAYSKIP], is);
        atomicAdd(&w.counters[CNT_ITERS_PREFIX], ip);
        atomicAdd(&w.counters[CNT_CULLED], cu);
    }
}

} // namespace svr
