// svr_trace_tile.hip -- the default path-tracing kernel for gfx950.
//
// Shape (what the rocprof counters of the earlier kernels asked for):
//  * persistent waves pull (8x8 pixel tile, frame) tasks from one ticket; tasks are tile-major, so the
//    waves resident at any moment march through neighbouring tubes of the volume, and the 64 rays of a
//    wave start together and stay within a few dozen voxels of depth of each other (Woodcock steps have
//    mean 1/sigma_max): the 256-B bricks they touch are served by the vector L1 / per-XCD L2
//    (measured 90 % / 93 % hit), not by HBM;
//  * the transfer-function alpha LUT (the only table the Woodcock loop needs) and an empty-space
//    bitmask live in LDS;
//  * EMPTY-SPACE SKIPPING, bit-exact: a Woodcock iteration whose trilinear cell lies in a macro-cell
//    where the transfer function's alpha is exactly 0 for every reachable intensity has sigma_t == 0,
//    so the reference's accept test `xi < sigma_t / sigma_max` (woodcock_tracking.h:43) fails whatever
//    the fetch returns; the 8 voxel loads, the filter and the LUT read are skipped while both random
//    draws of the iteration are still consumed, so every path follows the reference's RNG stream and
//    produces the same bits.  ~90 % of the Woodcock iterations of a CT-like scene are in air;
//  * the scatter point's intensity is the last Woodcock fetch (same position, same arithmetic), not a
//    second fetch;
//  * voxel addresses are 32-bit byte offsets from a scalar base (saddr global loads), brick strides use
//    24-bit multiplies.
// Radiance goes to the scratch slots; k_resolve (svr_kernels.hip) folds it into the running mean.
#include "svr_kernel_common.hpp"

namespace svr {

struct LdsTile {
    float alpha[SVR_TF_MAX + SVR_TF_PAD];      // entry e = alpha of texel clamp(e-1)
    uint32_t mask[MASK_WORDS_MAX];             // deep-empty bits: the macro-cell and its 26 neighbours are transparent
};
struct LdsTileNoMask {
    float alpha[SVR_TF_MAX + SVR_TF_PAD];
    uint32_t mask[1];
};

template <typename LDS>
SVR_DEV void lds_tile_load(LDS& L, const DevScene& s, bool with_mask)
{
    const int n = s.tf_n;
    for (int e = threadIdx.x; e < n + SVR_TF_PAD; e += blockDim.x) {
        int t = min(max(e - 1, 0), n - 1);
        L.alpha[e] = s.tf[4 * t + 3];
    }
    if (with_mask) {
        const uint4* src = reinterpret_cast<const uint4*>(s.empty_mask);
        uint4* dst = reinterpret_cast<uint4*>(L.mask);
        for (uint32_t q = threadIdx.x; q < (s.mask_words + 3u) / 4u; q += blockDim.x) dst[q] = src[q];
    }
    __syncthreads();
}

// trilinear cell of a world-space point: cuda_volume.h:87-90 + the first half of tex3D
struct Cell { int cx, cy, cz; float a, b, g; };

SVR_DEV Cell cell_of(const DevScene& s, v3 p)
{
    float u = (p.x - s.vmin[0]) * s.invSize[0];
    float v = (p.y - s.vmin[1]) * s.invSize[1];
    float w = (p.z - s.vmin[2]) * s.invSize[2];
    float xb = fma_(u, s.fnx, -0.5f);
    float yb = fma_(v, s.fny, -0.5f);
    float zb = fma_(w, s.fnz, -0.5f);
    float fx = __builtin_floorf(xb), fy = __builtin_floorf(yb), fz = __builtin_floorf(zb);
    Cell c;
    c.a = xb - fx; c.b = yb - fy; c.g = zb - fz;
    fx = fmin_(fmax_(fx, -2.f), s.fnx);
    fy = fmin_(fmax_(fy, -2.f), s.fny);
    fz = fmin_(fmax_(fz, -2.f), s.fnz);
    c.cx = (int)fx; c.cy = (int)fy; c.cz = (int)fz;
    return c;
}

SVR_DEV float ld_u16(const uint16_t* base, uint32_t byte_off)
{
    return (float)*reinterpret_cast<const uint16_t*>(reinterpret_cast<const char*>(base) + byte_off);
}

// second half of tex3D<float>: 8 voxels + float-weight trilinear filter, normalised by 1/65535
template <int LAYOUT>
SVR_DEV float tex_fetch(const DevScene& s, const Cell& c)
{
    uint32_t i = (uint32_t)(c.cx + VOL_PAD), j = (uint32_t)(c.cy + VOL_PAD), k = (uint32_t)(c.cz + VOL_PAD);
    const uint16_t* vox = s.vox;
    float v000, v100, v010, v110, v001, v101, v011, v111;
    if (LAYOUT == LAYOUT_LINEAR) {
        uint32_t base = ((k * (uint32_t)s.sz + j * (uint32_t)s.sy) + i) << 1;
        uint32_t dy = (uint32_t)s.sy << 1, dz = (uint32_t)s.sz << 1;
        v000 = ld_u16(vox, base);           v100 = ld_u16(vox, base + 2u);
        v010 = ld_u16(vox, base + dy);      v110 = ld_u16(vox, base + dy + 2u);
        v001 = ld_u16(vox, base + dz);      v101 = ld_u16(vox, base + dz + 2u);
        v011 = ld_u16(vox, base + dz + dy); v111 = ld_u16(vox, base + dz + dy + 2u);
    } else {
        // brick = 8x4x4 voxels = 256 B; byte offsets
        uint32_t i1 = i + 1u, j1 = j + 1u, k1 = k + 1u;
        uint32_t X0 = ((i >> 3) << 8) + ((i & 7u) << 1), X1 = ((i1 >> 3) << 8) + ((i1 & 7u) << 1);
        uint32_t ys = (uint32_t)s.bnx << 8;
        uint32_t zs = (uint32_t)(s.bny * s.bnx) << 8;                     // < 2^24, checked on the host
        uint32_t Y0 = __umul24(j >> 2, ys) + ((j & 3u) << 4), Y1 = __umul24(j1 >> 2, ys) + ((j1 & 3u) << 4);
        uint32_t Z0 = __umul24(k >> 2, zs) + ((k & 3u) << 6), Z1 = __umul24(k1 >> 2, zs) + ((k1 & 3u) << 6);
        uint32_t a00 = Y0 + Z0, a10 = Y1 + Z0, a01 = Y0 + Z1, a11 = Y1 + Z1;
        v000 = ld_u16(vox, a00 + X0); v100 = ld_u16(vox, a00 + X1);
        v010 = ld_u16(vox, a10 + X0); v110 = ld_u16(vox, a10 + X1);
        v001 = ld_u16(vox, a01 + X0); v101 = ld_u16(vox, a01 + X1);
        v011 = ld_u16(vox, a11 + X0); v111 = ld_u16(vox, a11 + X1);
    }
    float c00 = lerpf(v000, v100, c.a);
    float c10 = lerpf(v010, v110, c.a);
    float c01 = lerpf(v001, v101, c.a);
    float c11 = lerpf(v011, v111, c.a);
    float c0 = lerpf(c00, c10, c.b);
    float c1 = lerpf(c01, c11, c.b);
    return lerpf(c0, c1, c.g) * 1.5259021896696422e-05f;
}

template <int LAYOUT>
SVR_DEV float intensity_at(const DevScene& s, v3 p)
{
    return tex_fetch<LAYOUT>(s, cell_of(s, p)) * s.densityScale;
}

template <typename LDS>
SVR_DEV float alpha_of(const LDS& L, const DevScene& s, float x)
{
    int e; float a;
    lds_tf_coord(s, x, e, a);
    return lerpf(L.alpha[e], L.alpha[e + 1], a);
}

// macro-cell bit of a trilinear cell.  The mask covers the cells c = -1 .. N-1 (c+1 in [0, N]: every cell
// a point of the texture domain maps to); cells further out (clip planes beyond the volume, gradient
// taps) always fetch.
template <typename LDS>
SVR_DEV bool cell_is_empty(const LDS& L, const DevScene& s, const Cell& c)
{
    uint32_t ux = (uint32_t)(c.cx + 1), uy = (uint32_t)(c.cy + 1), uz = (uint32_t)(c.cz + 1);
    bool inb = (ux <= (uint32_t)s.nx) & (uy <= (uint32_t)s.ny) & (uz <= (uint32_t)s.nz);
    uint32_t sh = (uint32_t)s.mc_shift;
    uint32_t qx = min(ux >> sh, (uint32_t)s.mc_gx - 1u), qy = min(uy >> sh, (uint32_t)s.mc_gy - 1u),
             qz = min(uz >> sh, (uint32_t)s.mc_gz - 1u);
    uint32_t m = qx + __umul24(qy, (uint32_t)s.mc_gx) + __umul24(qz, (uint32_t)s.mc_gxy);
    m = inb ? m : 0u;
    uint32_t word = L.mask[m >> 5];
    return inb && ((word >> (m & 31u)) & 1u);
}

// Conservative march of the ray segment [t0, t1] through the macro grid (3D-DDA): returns the ray
// parameter at which the segment first enters a macro-cell that is not deep-empty, or +inf if it
// never does.  Float error in the march is far below one macro-cell, and a deep-empty cell has only
// empty neighbours, so every point of the ray with t < result lies in an empty macro-cell.
template <typename LDS>
SVR_DEV float first_occupied(const DevScene& s, const LDS& L, v3 o, v3 d, float t0, float t1)
{
    const float INF = u2f(SVR_INF_BITS);
    float Ax = fma_(o.x - s.vmin[0], s.mc_scale[0], s.mc_off), Bx = d.x * s.mc_scale[0];
    float Ay = fma_(o.y - s.vmin[1], s.mc_scale[1], s.mc_off), By = d.y * s.mc_scale[1];
    float Az = fma_(o.z - s.vmin[2], s.mc_scale[2], s.mc_off), Bz = d.z * s.mc_scale[2];
    int gx = s.mc_gx, gy = s.mc_gy, gz = s.mc_gz;
    int ix = min(max((int)__builtin_floorf(fma_(Bx, t0, Ax)), 0), gx - 1);
    int iy = min(max((int)__builtin_floorf(fma_(By, t0, Ay)), 0), gy - 1);
    int iz = min(max((int)__builtin_floorf(fma_(Bz, t0, Az)), 0), gz - 1);
    int sx = Bx > 0.f ? 1 : -1, sy = By > 0.f ? 1 : -1, sz = Bz > 0.f ? 1 : -1;
    float rx = __builtin_amdgcn_rcpf(Bx), ry = __builtin_amdgcn_rcpf(By), rz = __builtin_amdgcn_rcpf(Bz);
    float dtx = __builtin_fabsf(rx), dty = __builtin_fabsf(ry), dtz = __builtin_fabsf(rz);
    float tnx = (Bx != 0.f) ? ((float)(ix + (Bx > 0.f ? 1 : 0)) - Ax) * rx : INF;
    float tny = (By != 0.f) ? ((float)(iy + (By > 0.f ? 1 : 0)) - Ay) * ry : INF;
    float tnz = (Bz != 0.f) ? ((float)(iz + (Bz > 0.f ? 1 : 0)) - Az) * rz : INF;
    dtx = (Bx != 0.f) ? dtx : INF; dty = (By != 0.f) ? dty : INF; dtz = (Bz != 0.f) ? dtz : INF;
    float t = t0;
    const uint32_t* deep = L.mask;
    int guard = gx + gy + gz + 4;
    for (int it = 0; it < guard; ++it) {
        uint32_t q = (uint32_t)ix + __umul24((uint32_t)iy, (uint32_t)gx) + __umul24((uint32_t)iz, (uint32_t)s.mc_gxy);
        if (!((deep[q >> 5] >> (q & 31u)) & 1u)) return t;
        float tn = fmin_(tnx, fmin_(tny, tnz));
        if (!(tn <= t1)) return INF;          // the segment ends inside this cell
        t = tn;
        if (tnx <= tny && tnx <= tnz) { ix += sx; tnx += dtx; if ((uint32_t)ix >= (uint32_t)gx) return INF; }
        else if (tny <= tnz) { iy += sy; tny += dty; if ((uint32_t)iy >= (uint32_t)gy) return INF; }
        else { iz += sz; tnz += dtz; if ((uint32_t)iz >= (uint32_t)gz) return INF; }
    }
    return t;   // guard exhausted (cannot happen: each step leaves a cell): treat the rest as occupied
}

// sample_distance, woodcock_tracking.h:20-51.  `val` returns the intensity fetched by the accepted
// iteration (= volume(PointOnRay(t)), the scatter point's intensity, pathtracer.cu:241).
// rng_live: a random draw of this path can follow the walk; if not, and the walk provably cannot
// collide, its result (-FLT_MAX) is known without running it.
template <int LAYOUT, bool COUNT, bool SKIP, typename LDS>
SVR_DEV float walk(const DevScene& s, const LDS& L, v3 orig, v3 dir, Rng& rng, float& tMin, float& tMax,
                   float& val, bool rng_live, Cnt& c)
{
    float tNear, tFar;
    if (!volume_intersect(s, orig, dir, tNear, tFar)) return -SVR_FLT_MAX;
    tMin = tNear < 0.f ? (float)1e-6 : tNear;
    tMax = tFar;
    float t = tMin;
    float t_occ = tMin;                 // fetches may be needed from here on
    if (SKIP && s.ray_skip) {
        t_occ = first_occupied(s, L, orig, dir, tMin, tMax);
        // COUNT builds run every walk so that the iteration/tap counters stay the reference's
        if (!COUNT && !rng_live && t_occ == u2f(SVR_INF_BITS)) return -SVR_FLT_MAX;
    }
    const bool ray_skippable = SKIP && s.ray_skip && !rng_live && t_occ == u2f(SVR_INF_BITS);
    if (COUNT && ray_skippable) c.wskip++;
    for (uint32_t guard = 0;; ++guard) {
        if (COUNT) { c.iters++; if (ray_skippable) c.iskip++; else if (SKIP && t < t_occ) c.ipre++; }
        t += -logf_unit(1.f - rng_uniform(rng)) * s.invSigmaMaxSI;
        if (t > tMax || guard >= SVR_WALK_GUARD) return -SVR_FLT_MAX;
        if (COUNT) c.taps++;
        float sigma_t = 0.f;
        if (!SKIP || t >= t_occ) {
            v3 p = orig + dir * t;
            Cell cell = cell_of(s, p);
            bool fetch = true;
            if (SKIP) fetch = !cell_is_empty(L, s, cell);
            if (fetch) {
                if (COUNT) c.exec++;
                val = tex_fetch<LAYOUT>(s, cell) * s.densityScale;
                sigma_t = alpha_of(L, s, val);
            }
        }
        // the accept draw is consumed either way; with sigma_t == 0 it cannot accept (xi > 0)
        if (rng_uniform(rng) < sigma_t * s.invSigmaMax) break;
    }
    return t;
}

// one path: kernel_pathtracer body, pathtracer.cu:205-277
template <int LAYOUT, bool COUNT, bool SKIP, typename LDS>
SVR_DEV v3 trace_path_tile(const DevScene& s, const LDS& L_, uint32_t x, uint32_t y, uint32_t traceDepth,
                           uint32_t hashed, Cnt& c)
{
    uint32_t offset = y * s.imageW + x;
    Rng rng;
    rng_init(rng, hashed + offset);
    if (COUNT) c.paths++;
    v3 L = V3(0.f, 0.f, 0.f), T = V3(1.f, 1.f, 1.f);
    v3 orig, dir;
    camera_ray(s, x, y, rng, orig, dir);
    float ls_t;
    int ls_id = nearest_light(s, orig, dir, ls_t);
    for (uint32_t k = 0; k < traceDepth; ++k) {
        float tMin = (float)1e-6, tMax = SVR_FLT_MAX, val = 0.f;
        float t = walk<LAYOUT, COUNT, SKIP>(s, L_, orig, dir, rng, tMin, tMax, val, false, c);
        if (k == 0 && ls_id >= 0) {
            t = t < 0.f ? SVR_FLT_MAX : t;
            if (ls_t < t) {
                const DevLight& l = s.lights[ls_id];
                float cosTerm = dot(V3(l.normal[0], l.normal[1], l.normal[2]), -dir);
                L = L + (T * V3(l.radiance[0], l.radiance[1], l.radiance[2])) * (cosTerm <= 0.f ? 0.f : 1.f);
                break;
            }
        }
        if (t < 0.f) {
            if (s.env_on_escape) L = L + T * env_radiance(s, dir);
            break;
        }
        // VolumeSample, pathtracer.cu:237-244
        Shade vs;
        if (COUNT) { c.scatter++; c.taps += 7; c.exec += 6; }
        vs.wo = -dir;
        vs.pt = orig + dir * t;
        tf_rgba(s, s.tf, val, vs.color);
        {
            // Gradient_CentralDiff, cuda_volume.h:54-61
            v3 q = vs.pt;
            float xd = intensity_at<LAYOUT>(s, V3(q.x + s.spacing[0], q.y + 0.f, q.z + 0.f)) -
                       intensity_at<LAYOUT>(s, V3(q.x - s.spacing[0], q.y - 0.f, q.z - 0.f));
            float yd = intensity_at<LAYOUT>(s, V3(q.x + 0.f, q.y + s.spacing[1], q.z + 0.f)) -
                       intensity_at<LAYOUT>(s, V3(q.x - 0.f, q.y - s.spacing[1], q.z - 0.f));
            float zd = intensity_at<LAYOUT>(s, V3(q.x + 0.f, q.y + 0.f, q.z + s.spacing[2])) -
                       intensity_at<LAYOUT>(s, V3(q.x - 0.f, q.y - 0.f, q.z - s.spacing[2]));
            vs.gradient = V3((xd * 0.5f) * s.invSpacing[0], (yd * 0.5f) * s.invSpacing[1], (zd * 0.5f) * s.invSpacing[2]);
        }
        float gradMag = __builtin_sqrtf(dot(vs.gradient, vs.gradient));
        vs.Pbrdf = vs.color[3] * (1.f - expf_(s.pbrdf_c * gradMag * 65535.f * s.invMaxMagnitude));
        vs.st = (rng_uniform(rng) < vs.Pbrdf) ? 1 : 0;
        // estimate_direct_light, pathtracer.cu:171-198
        v3 Ld = V3(0.f, 0.f, 0.f);
        if (s.num_lights != 0) {
            int lightId = (int)((float)s.num_lights * rng_uniform(rng));
            lightId = lightId < (int)s.num_lights ? lightId : (int)s.num_lights - 1;
            v3 wiL, Li; float pdfL;
            if (sample_light(s.lights[lightId], vs.pt, rng, wiL, pdfL, Li)) {
                float sMin = (float)1e-6, sMax = SVR_FLT_MAX, sval = 0.f;
                if (COUNT) c.shadow++;
                // the draws of sample_bsdf / roulette follow the shadow walk unless this is the last bounce
                float ts = walk<LAYOUT, COUNT, SKIP>(s, L_, vs.pt, wiL, rng, sMin, sMax, sval, k + 1u < traceDepth, c);
                float Tr = ((ts > sMin) && (ts < sMax)) ? 0.f : 1.f;          // transmittance.h:15-16
                float kf = Tr * (float)s.num_lights;
                Ld = ((bsdf_eval(vs, wiL) * kf) * Li) / pdfL;
            }
        }
        L = L + T * Ld;
        if (k + 1u >= traceDepth) break;      // sample_bsdf / roulette of the last bounce cannot reach L
        v3 wi; float pdf = 0.f;
        v3 f = bsdf_sample(vs, wi, pdf, rng);
        float cosTerm = __builtin_fabsf(dot(normalize(vs.gradient), wi));
        if (fmax_(f.x, fmax_(f.y, f.z)) > 0.f && pdf > 0.f) {
            if (vs.st == 0) T = T * (f / (pdf * (1.f - vs.Pbrdf)));
            else T = T * ((f * cosTerm) / (pdf * vs.Pbrdf));
        }
        orig = vs.pt;
        dir = wi;
        if (k >= 3) {
            if (russian_roulette(T, rng)) break;
        }
    }
    return L;
}

#ifndef SVR_TILE_WAVES_PER_EU
#define SVR_TILE_WAVES_PER_EU 4
#endif
#ifndef SVR_TILE_THREADS
#define SVR_TILE_THREADS 256
#endif

template <int LAYOUT, bool COUNT, bool SKIP>
__global__ __launch_bounds__(SVR_TILE_THREADS, SVR_TILE_WAVES_PER_EU) void k_trace_tile(const DevScene s, const DevWork w)
{
    using LDS = typename std::conditional<SKIP, LdsTile, LdsTileNoMask>::type;
    __shared__ LDS lds;
    lds_tile_load(lds, s, SKIP);

    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wv = w.x1 - w.x0;
    const uint32_t tiles_x = (wv + 7u) >> 3;
    const uint32_t n_tasks = tiles_x * ((w.n_rows + 7u) >> 3) * w.nframes;
    Cnt c = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

    for (;;) {
        uint32_t task = 0;
        if (lane == 0) task = atomicAdd(w.ticket, 1u);
        task = __builtin_amdgcn_readfirstlane(task);
        if (task >= n_tasks) break;
        if (COUNT) c.loops += (lane == 0);
        uint32_t tile = task / w.nframes;
        uint32_t slot = task - tile * w.nframes;
        uint32_t ty = tile / tiles_x, tx = tile - ty * tiles_x;
        uint32_t px = (tx << 3) + (lane & 7u);
        uint32_t r = (ty << 3) + (lane >> 3);
        if (px < wv && r < w.n_rows) {
            uint32_t x = w.x0 + px, y = owned_row_to_y(w, r);
            v3 L = trace_path_tile<LAYOUT, COUNT, SKIP>(s, lds, x, y, w.traceDepth, wang_hash(w.frame0 + slot), c);
            float* o = w.lbuf + (size_t)slot * w.slot_stride + 3 * ((size_t)y * s.imageW + x);
            o[0] = L.x; o[1] = L.y; o[2] = L.z;
        }
    }
    if (COUNT) cnt_flush(w, c);
}

template <int LAYOUT, bool COUNT>
static hipError_t launch_tile_t(const DevScene& s, const DevWork& w, const LaunchCfg& cfg, hipStream_t st)
{
    uint32_t wv = w.x1 - w.x0;
    if (wv == 0 || w.n_rows == 0) return hipSuccess;
    uint32_t n_tasks = ((wv + 7u) >> 3) * ((w.n_rows + 7u) >> 3) * w.nframes;
    constexpr uint32_t WPB = SVR_TILE_THREADS / 64;                       // waves per block
    uint32_t max_blocks = (uint32_t)(cfg.num_cus * cfg.blocks_per_cu) * 4u / WPB;
    uint32_t need = (n_tasks + WPB - 1u) / WPB;
    uint32_t blocks = need < max_blocks ? need : max_blocks;
    if (blocks == 0) blocks = 1;
    hipError_t e = hipMemsetAsync(w.ticket, 0, sizeof(uint32_t), st);
    if (e != hipSuccess) return e;
    if (s.empty_mask != nullptr)
        hipLaunchKernelGGL((k_trace_tile<LAYOUT, COUNT, true>), dim3(blocks), dim3(SVR_TILE_THREADS), 0, st, s, w);
    else
        hipLaunchKernelGGL((k_trace_tile<LAYOUT, COUNT, false>), dim3(blocks), dim3(SVR_TILE_THREADS), 0, st, s, w);
    return hipGetLastError();
}

hipError_t launch_trace_tile(const DevScene& s, const DevWork& w, const LaunchCfg& cfg, hipStream_t st)
{
    if (s.layout == LAYOUT_LINEAR)
        return cfg.count ? launch_tile_t<LAYOUT_LINEAR, true>(s, w, cfg, st) : launch_tile_t<LAYOUT_LINEAR, false>(s, w, cfg, st);
    return cfg.count ? launch_tile_t<LAYOUT_BRICK, true>(s, w, cfg, st) : launch_tile_t<LAYOUT_BRICK, false>(s, w, cfg, st);
}

} // namespace svr
