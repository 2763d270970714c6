// svr_walk.hpp -- the Woodcock-walk machinery shared by the tile kernel (svr_trace_tile.hip) and the
// wavefront kernels (svr_wavefront.hip): LDS-resident alpha LUT + empty bitmask + distance field, the software
// tex3D (cell + fetch), the conservative whole-ray march, and the walk loop itself.
#pragma once
#include "svr_kernel_common.hpp"
#include <type_traits>

namespace svr {

struct LdsTile {
    float alpha[SVR_TF_MAX + SVR_TF_PAD];      // entry e = alpha of texel clamp(e-1)
    uint32_t dist[DIST_WORDS_MAX];             // 4-bit distance (macro-cells, half resolution) to the nearest non-empty macro-cell (long leaps of a march)
    uint32_t mask[MASK_WORDS_MAX];             // deep-empty bits (distance >= 2 at full resolution): the macro-cell and its 26 neighbours are transparent
    uint32_t emask[MASK_WORDS_MAX];            // empty bits: the macro-cell itself is transparent (per-fetch test, exact cell index)
};
struct LdsTileNoMask {
    float alpha[SVR_TF_MAX + SVR_TF_PAD];
    uint32_t dist[1];
    uint32_t mask[1];
    uint32_t emask[1];
};
// LdsTile + the bound classes of the half-resolution macro-cells (svr_accel.hip, k_bound_class): the tile kernel
struct LdsTileCull : LdsTile {
    uint32_t cls[DIST_WORDS_MAX];              // 4 bits per half-resolution macro-cell
    float thr[BOUND_CLASSES];                  // class -> threshold on the accept draw
};
// LdsTileCull + the byte table of the fast bound look-up (svr_accel.hip, k_bound8): the POOL builds of the tile kernel.  The table comes FIRST:
// the object sits at LDS address 0, so the index is the address (no base to add in the walk loop; the other tables' offsets still fit their instructions'
// 16-bit offset field, except the `empty` bits and the classes, which media that take the fast look-up do not read)
struct LdsBound8 {
    uint8_t bnd[BOUND8_BYTES];                 // half-resolution macro-cell -> fetch iff (random word >> 24) <= byte
};
struct LdsTilePool : LdsBound8, LdsTileCull {};
template <typename LDS> struct lds_has_cull { static constexpr bool value = std::is_base_of<LdsTileCull, LDS>::value; };
template <typename LDS> struct lds_has_bnd8 { static constexpr bool value = std::is_base_of<LdsTilePool, LDS>::value; };

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
        uint4* dst = reinterpret_cast<uint4*>(L.dist);
        for (uint32_t q = threadIdx.x; q < (s.dist_words + 3u) / 4u; q += blockDim.x) dst[q] = src[q];
        const uint4* src1 = reinterpret_cast<const uint4*>(s.empty_mask + DIST_WORDS_MAX);
        uint4* dst1 = reinterpret_cast<uint4*>(L.mask);
        for (uint32_t q = threadIdx.x; q < (s.mask_words + 3u) / 4u; q += blockDim.x) dst1[q] = src1[q];
        const uint4* src2 = reinterpret_cast<const uint4*>(s.empty_mask + DIST_WORDS_MAX + MASK_WORDS_MAX);
        uint4* dst2 = reinterpret_cast<uint4*>(L.emask);
        for (uint32_t q = threadIdx.x; q < (s.mask_words + 3u) / 4u; q += blockDim.x) dst2[q] = src2[q];
        if constexpr (lds_has_cull<LDS>::value) {
            const uint4* src3 = reinterpret_cast<const uint4*>(s.empty_mask + ACCEL_CLASS_OFF);
            uint4* dst3 = reinterpret_cast<uint4*>(L.cls);
            for (uint32_t q = threadIdx.x; q < (s.dist_words + 3u) / 4u; q += blockDim.x) dst3[q] = src3[q];
            if (threadIdx.x < BOUND_CLASSES) L.thr[threadIdx.x] = reinterpret_cast<const float*>(s.empty_mask + ACCEL_THR_OFF)[threadIdx.x];
        }
        if constexpr (lds_has_bnd8<LDS>::value) {
            if (s.bnd8 != nullptr) {
                const uint4* src4 = reinterpret_cast<const uint4*>(s.bnd8);
                uint4* dst4 = reinterpret_cast<uint4*>(L.bnd);
                for (uint32_t q = threadIdx.x; q < BOUND8_BYTES / 16u; q += blockDim.x) dst4[q] = src4[q];
            }
        }
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
    } else if (LAYOUT == LAYOUT_CELL) {
        // brick = 8x4x4 elements of 16 bytes = 2 KB; element (i, j, k) = the 8 voxels of the trilinear cell (i, j, k), as four
        // PAIR words (y, z), (y + 1, z), (y, z + 1), (y + 1, z + 1): a fetch is ONE 16-byte load from ONE 32-byte sector instead of
        // 4 gathers from 4 sectors -- a quarter of the gather instructions, tag look-ups and HBM sectors of PAIR -- for 8 x the
        // memory of the u16 volume: 2.2 GB for 512^3, 17 GB for 1024^3 of a 288 GB device
        const uint32_t X0 = ((i >> 3) << 7) + (i & 7u);
        const uint32_t Y0 = __umul24(j >> 2, (uint32_t)s.bnx << 7) + ((j & 3u) << 3);
        const uint32_t Z0 = __umul24(k >> 2, (uint32_t)(s.bny * s.bnx) << 7) + ((k & 3u) << 5);      // < 2^24, checked on the host
        // (64-bit byte offset: 1024^3 is 17 GB in this layout)
        const uint4 c4 = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(vox) + ((uint64_t)(X0 + Y0 + Z0) << 4));
        v000 = (float)(c4.x & 0xffffu); v100 = (float)(c4.x >> 16);
        v010 = (float)(c4.y & 0xffffu); v110 = (float)(c4.y >> 16);
        v001 = (float)(c4.z & 0xffffu); v101 = (float)(c4.z >> 16);
        v011 = (float)(c4.w & 0xffffu); v111 = (float)(c4.w >> 16);
    } else if (LAYOUT == LAYOUT_PAIR) {
        // brick = 8x4x4 elements of 32 bits = 512 B; element (i, j, k) = voxel (i, j, k) | voxel (i + 1, j, k) << 16: the two
        // x-neighbours of a trilinear footprint arrive in ONE load, so a fetch is 4 gather instructions instead of 8 (the
        // software sampler is bound by the rate at which the texture-address / L1 path takes 64-lane gathers, not by bytes),
        // for twice the memory -- 552 MB for a 512^3 volume of a 288 GB device
        uint32_t j1 = j + 1u, k1 = k + 1u;
        uint32_t X0 = ((i >> 3) << 9) + ((i & 7u) << 2);
        uint32_t ys = (uint32_t)s.bnx << 9;
        uint32_t zs = (uint32_t)(s.bny * s.bnx) << 9;                     // < 2^24, checked on the host
        uint32_t Y0 = __umul24(j >> 2, ys) + ((j & 3u) << 5), Y1 = __umul24(j1 >> 2, ys) + ((j1 & 3u) << 5);
        uint32_t Z0 = __umul24(k >> 2, zs) + ((k & 3u) << 7), Z1 = __umul24(k1 >> 2, zs) + ((k1 & 3u) << 7);
        const char* b = reinterpret_cast<const char*>(vox);
        const uint32_t p00 = *reinterpret_cast<const uint32_t*>(b + (Y0 + Z0 + X0)), p10 = *reinterpret_cast<const uint32_t*>(b + (Y1 + Z0 + X0));
        const uint32_t p01 = *reinterpret_cast<const uint32_t*>(b + (Y0 + Z1 + X0)), p11 = *reinterpret_cast<const uint32_t*>(b + (Y1 + Z1 + X0));
        v000 = (float)(p00 & 0xffffu); v100 = (float)(p00 >> 16);
        v010 = (float)(p10 & 0xffffu); v110 = (float)(p10 >> 16);
        v001 = (float)(p01 & 0xffffu); v101 = (float)(p01 >> 16);
        v011 = (float)(p11 & 0xffffu); v111 = (float)(p11 >> 16);
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

// distance-field entry of macro-cell (ix, iy, iz) (in-grid): every macro-cell within Chebyshev distance d - 1 is empty
template <typename LDS>
SVR_DEV uint32_t dist_at(const LDS& L, const DevScene& s, uint32_t ix, uint32_t iy, uint32_t iz)
{
    const uint32_t hq = (ix >> 1) + __umul24(iy >> 1, (uint32_t)s.mc_hgx) + __umul24(iz >> 1, (uint32_t)s.mc_hgxy);
    return (L.dist[hq >> 3] >> ((hq & 7u) << 2)) & 15u;
}

// macro-cell of a trilinear cell.  The grid covers the cells c = -1 .. N-1 (c+1 in [0, N]: every cell
// a point of the texture domain maps to); cells further out (clip planes beyond the volume, gradient
// taps) always fetch.
// deep = false: the macro-cell is transparent (no fetch needed); deep = true: so are its 26 neighbours
template <bool DEEP, typename LDS>
SVR_DEV bool cell_is_empty(const LDS& L, const DevScene& s, const Cell& c)
{
    uint32_t ux = (uint32_t)(c.cx + 1), uy = (uint32_t)(c.cy + 1), uz = (uint32_t)(c.cz + 1);
    bool inb = (ux <= (uint32_t)s.nx) & (uy <= (uint32_t)s.ny) & (uz <= (uint32_t)s.nz);
    uint32_t sh = (uint32_t)s.mc_shift;
    uint32_t qx = min(ux >> sh, (uint32_t)s.mc_gx - 1u), qy = min(uy >> sh, (uint32_t)s.mc_gy - 1u),
             qz = min(uz >> sh, (uint32_t)s.mc_gz - 1u);
    uint32_t m = qx + __umul24(qy, (uint32_t)s.mc_gx) + __umul24(qz, (uint32_t)s.mc_gxy);
    m = inb ? m : 0u;
    uint32_t word = DEEP ? L.mask[m >> 5] : L.emask[m >> 5];
    return inb && ((word >> (m & 31u)) & 1u);
}

// One look at the macro grid for a trilinear cell: CELL_EMPTY (no fetch can return a non-zero opacity), CELL_DEEP
// (so are the 26 neighbours), and the bound on the accept draw below which a fetch is needed at all (+inf outside
// the grid and when culling is off).
struct CellInfo { bool empty, deep; float thr; };
template <bool WANT_DEEP, typename LDS>
SVR_DEV CellInfo cell_info(const LDS& L, const DevScene& s, const Cell& c)
{
    const uint32_t ux = (uint32_t)(c.cx + 1), uy = (uint32_t)(c.cy + 1), uz = (uint32_t)(c.cz + 1);
    const bool inb = (ux <= (uint32_t)s.nx) & (uy <= (uint32_t)s.ny) & (uz <= (uint32_t)s.nz);
    const uint32_t sh = (uint32_t)s.mc_shift;
    const uint32_t qx = min(ux >> sh, (uint32_t)s.mc_gx - 1u), qy = min(uy >> sh, (uint32_t)s.mc_gy - 1u),
                   qz = min(uz >> sh, (uint32_t)s.mc_gz - 1u);
    uint32_t m = qx + __umul24(qy, (uint32_t)s.mc_gx) + __umul24(qz, (uint32_t)s.mc_gxy);
    m = inb ? m : 0u;
    CellInfo r;
    r.empty = false; r.deep = false;
    if (s.has_empty) r.empty = inb && ((L.emask[m >> 5] >> (m & 31u)) & 1u);
    if (s.fine_mask != nullptr && inb && !r.empty) {
        // the coarse cell holds something somewhere: ask the fine level (cells of half the edge, global memory) about this spot
        const uint32_t fs = sh - 1u;
        const uint32_t fx = min(ux >> fs, (uint32_t)s.fg_x - 1u), fy = min(uy >> fs, (uint32_t)s.fg_y - 1u), fz = min(uz >> fs, (uint32_t)s.fg_z - 1u);
        const uint32_t f = fx + fy * (uint32_t)s.fg_x + fz * (uint32_t)s.fg_xy;
        r.empty = (s.fine_mask[f >> 5] >> (f & 31u)) & 1u;
    }
    if (WANT_DEEP && s.has_empty) r.deep = inb && ((L.mask[m >> 5] >> (m & 31u)) & 1u);
    r.thr = u2f(SVR_INF_BITS);
    if constexpr (lds_has_cull<LDS>::value) {
        if (s.bound_cull) {
            uint32_t hq = (qx >> 1) + __umul24(qy >> 1, (uint32_t)s.mc_hgx) + __umul24(qz >> 1, (uint32_t)s.mc_hgxy);
            hq = inb ? hq : 0u;
            const uint32_t cl = (L.cls[hq >> 3] >> ((hq & 7u) << 2)) & 15u;
            r.thr = inb ? L.thr[cl] : u2f(SVR_INF_BITS);
        }
    }
    return r;
}

// Conservative march of the ray segment [t0, t1] through the macro grid: returns a ray parameter before which
// every point of the segment lies in an empty macro-cell (and at which a cell next to a non-empty one is reached), or
// +inf if the whole segment is clear.  Sphere tracing on the distance field: from a point whose cell has distance d,
// the ray may advance until its largest-axis displacement is d - 1 cells (0.05 cell of margin covers the float error
// of the march, which is far below one macro-cell).  ~10 steps for a ray that crosses a 64^3 grid, against ~100 cell
// crossings of a 3D-DDA.
template <typename LDS>
SVR_DEV float first_occupied(const DevScene& s, const LDS& L, v3 o, v3 d, float t0, float t1)
{
    const float INF = u2f(SVR_INF_BITS);
    const float Ax = fma_(o.x - s.vmin[0], s.mc_scale[0], s.mc_off), Bx = d.x * s.mc_scale[0];
    const float Ay = fma_(o.y - s.vmin[1], s.mc_scale[1], s.mc_off), By = d.y * s.mc_scale[1];
    const float Az = fma_(o.z - s.vmin[2], s.mc_scale[2], s.mc_off), Bz = d.z * s.mc_scale[2];
    const float bmax = fmax_(__builtin_fabsf(Bx), fmax_(__builtin_fabsf(By), __builtin_fabsf(Bz)));
    if (!(bmax > 0.f)) return t0;                 // degenerate direction: no skipping
    const float inv = __builtin_amdgcn_rcpf(bmax) * 0.999f;
    const int gx = s.mc_gx, gy = s.mc_gy, gz = s.mc_gz;
    float t = t0;
    for (int it = 0; it < 512; ++it) {
        const uint32_t ix = (uint32_t)min(max((int)__builtin_floorf(fma_(Bx, t, Ax)), 0), gx - 1);
        const uint32_t iy = (uint32_t)min(max((int)__builtin_floorf(fma_(By, t, Ay)), 0), gy - 1);
        const uint32_t iz = (uint32_t)min(max((int)__builtin_floorf(fma_(Bz, t, Az)), 0), gz - 1);
        uint32_t dd = dist_at(L, s, ix, iy, iz);
        if (dd < 2u) {
            // close to something at half resolution: decide on the full-resolution deep-empty bit
            const uint32_t q = ix + __umul24(iy, (uint32_t)gx) + __umul24(iz, (uint32_t)s.mc_gxy);
            if (!((L.mask[q >> 5] >> (q & 31u)) & 1u)) return t;
            dd = 2u;
        }
        t = fma_((float)dd - 1.05f, inv, t);
        if (t > t1) return INF;
    }
    return t;   // step budget exhausted: treat the rest as occupied
}

// The same question for a GROUP of lanes at once.  With frame-major lanes (svr_trace_tile.hip) the 1 << fl2 lanes
// pl, pl + 2^P2, pl + 2 * 2^P2 ... of a wave trace one pixel in different frames: primary rays from one origin whose
// directions differ by sub-pixel jitter -- a small fraction of a macro-cell over the whole box.  Instead of 2^fl2
// identical sequential DDAs the group samples its FIRST lane's ray at steps of 0.7 macro-cell (largest axis), one
// sample per lane and round, and takes the first sample that is not deep-empty.  A lane may use the result if it has
// the same origin and stays within 0.25 macro-cell of the reference ray up to the far end of the group's segment:
// every point of its ray before the returned parameter is then within 0.95 cell (per axis) of an earlier, deep-empty
// sample, i.e. in that cell or one of its 26 neighbours, which are all empty.  Lanes that fail the test (`ok` false:
// thin lens, wide pixels on a coarse grid) fall back to their own DDA.  Must be called by all 64 lanes of the wave.
//
// The samples of the first GROUP_MAP_ROUNDS rounds are also kept as a bitmap (GroupMap): if they reach the end of the
// segment, a lane can later ask in O(1) where the next possibly-occupied stretch after its current position starts
// (group_map_next) -- the cheap stand-in for re-marching a primary walk that has come out of an occupied stretch.
constexpr uint32_t GROUP_MAP_ROUNDS = 3;
constexpr uint32_t GROUP_MAPS_PER_WAVE = 8;     // a shared march needs >= 8 frames of a pixel in the wave: at most 8 pixel groups
// One map per PIXEL GROUP, in LDS (every lane of the group computes the same values and stores them to the group's
// slot; a lane reads back what it -- or a lane with identical data -- wrote, so no ordering between lanes is needed).
struct GroupMapShared {
    uint64_t w[GROUP_MAP_ROUNDS];      // round r: bit (j << P2) = sample r * Lg + j is not deep-empty
    uint64_t pad;
};
struct GroupMap {
    GroupMapShared* g;                 // the group's slot (the sample bits: indexed by the round counter, so they live in memory)
    float lo, dt, inv_dt;              // sample k sits at lo + k * dt (per-lane copies: registers)
    bool valid;                        // the map covers the whole segment and this lane may use it
};

SVR_DEV float group_map_next(const GroupMap& gm, uint32_t P2, float t)
{
    struct { float lo, dt, inv_dt; const uint64_t* w; } g = {gm.lo, gm.dt, gm.inv_dt, gm.g->w};
    const uint32_t fl2 = 6u - P2, Lg = 64u >> P2;
    // a sample at or before t (one earlier than the quotient says, against rounding): coverage from t_k includes t
    int ki = (int)((t - g.lo) * g.inv_dt) - 1;
    const uint32_t k = (uint32_t)(ki < 0 ? 0 : ki);
    const uint32_t r = k >> fl2, j = k & (Lg - 1u);
    uint64_t w0 = r == 0u ? g.w[0] : (r == 1u ? g.w[1] : (r == 2u ? g.w[2] : 0ull));
    w0 >>= (j << P2);
    if (w0) return fma_((float)(k + ((uint32_t)__builtin_ctzll(w0) >> P2)), g.dt, g.lo);
    const uint64_t w1 = r == 0u ? g.w[1] : (r == 1u ? g.w[2] : 0ull);
    if (w1) return fma_((float)(((r + 1u) << fl2) + ((uint32_t)__builtin_ctzll(w1) >> P2)), g.dt, g.lo);
    const uint64_t w2 = r == 0u ? g.w[2] : 0ull;
    if (w2) return fma_((float)(((r + 2u) << fl2) + ((uint32_t)__builtin_ctzll(w2) >> P2)), g.dt, g.lo);
    return u2f(SVR_INF_BITS);
}

template <typename LDS>
SVR_DEV float first_occupied_group(const DevScene& s, const LDS& L, uint32_t P2, v3 o, v3 d, bool hit, float tMin, float tMax, bool& ok,
                                   GroupMap& map)
{
    GroupMapShared& ms = *map.g;
    map.valid = false; map.lo = 0.f; map.dt = 1.f; map.inv_dt = 1.f;
    for (uint32_t r = 0; r < GROUP_MAP_ROUNDS; ++r) ms.w[r] = 0ull;
    const float INF = u2f(SVR_INF_BITS);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t pl = lane & ((1u << P2) - 1u), m = lane >> P2, Lg = 64u >> P2;
    const int leader = (int)pl;
    const v3 oc = V3(__shfl(o.x, leader, 64), __shfl(o.y, leader, 64), __shfl(o.z, leader, 64));
    const v3 dc = V3(__shfl(d.x, leader, 64), __shfl(d.y, leader, 64), __shfl(d.z, leader, 64));
    float lo = hit ? tMin : INF, hi = hit ? tMax : -INF;
    for (uint32_t off = 32u; off >= (1u << P2); off >>= 1) {
        lo = fmin_(lo, __shfl_xor(lo, (int)off, 64));
        hi = fmax_(hi, __shfl_xor(hi, (int)off, 64));
    }
    const float dev = fmax_(__builtin_fabsf((d.x - dc.x) * s.mc_scale[0]),
                            fmax_(__builtin_fabsf((d.y - dc.y) * s.mc_scale[1]), __builtin_fabsf((d.z - dc.z) * s.mc_scale[2])));
    ok = (o.x == oc.x) && (o.y == oc.y) && (o.z == oc.z) && (dev * hi <= 0.25f);
    if (!(lo <= hi)) return INF;                                   // no lane of the group hits the box
    const float Ax = fma_(oc.x - s.vmin[0], s.mc_scale[0], s.mc_off), Bx = dc.x * s.mc_scale[0];
    const float Ay = fma_(oc.y - s.vmin[1], s.mc_scale[1], s.mc_off), By = dc.y * s.mc_scale[1];
    const float Az = fma_(oc.z - s.vmin[2], s.mc_scale[2], s.mc_off), Bz = dc.z * s.mc_scale[2];
    const float bmax = fmax_(__builtin_fabsf(Bx), fmax_(__builtin_fabsf(By), __builtin_fabsf(Bz)));
    if (!(bmax > 0.f)) { ok = false; return INF; }
    const float dt = 0.7f / bmax;
    // bits of this group in a ballot: lanes pl + j * 2^P2
    const uint64_t every[7] = {~0ull, 0x5555555555555555ull, 0x1111111111111111ull, 0x0101010101010101ull,
                               0x0001000100010001ull, 0x0000000100000001ull, 1ull};
    const uint64_t gmask = every[P2] << pl;
    const int gx = s.mc_gx, gy = s.mc_gy, gz = s.mc_gz;
    float result = INF;
    bool found = false, complete = false;
    uint32_t round = 0;
    for (uint32_t k0 = 0u; k0 < 4096u; k0 += Lg, ++round) {
        const float tk = fma_((float)(k0 + m), dt, lo);
        const int ix = min(max((int)__builtin_floorf(fma_(Bx, tk, Ax)), 0), gx - 1);
        const int iy = min(max((int)__builtin_floorf(fma_(By, tk, Ay)), 0), gy - 1);
        const int iz = min(max((int)__builtin_floorf(fma_(Bz, tk, Az)), 0), gz - 1);
        const uint32_t q = (uint32_t)(ix + iy * gx + iz * s.mc_gxy);
        const bool occupied = (tk <= hi) && !((L.mask[q >> 5] >> (q & 31u)) & 1u);
        const uint64_t b = __ballot(occupied) & gmask;
        if (round < GROUP_MAP_ROUNDS) ms.w[round] = b >> pl;
        if (b && !found) { found = true; result = fma_((float)(k0 + (((uint32_t)__builtin_ctzll(b) - pl) >> P2)), dt, lo); }
        const bool at_end = fma_((float)(k0 + Lg - 1u), dt, lo) > hi;      // the round reached the end of the segment
        if (at_end) { complete = round < GROUP_MAP_ROUNDS; break; }
        if (found && round + 1u >= GROUP_MAP_ROUNDS) break;                // the map cannot be completed: stop at the first hit
    }
    map.lo = lo; map.dt = dt; map.inv_dt = bmax * (1.f / 0.7f);
    map.valid = complete && ok;
    // the group's slot was written by every lane of the group (identical values) and is read by all of them in group_map_next
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    return result;
}

// sample_distance, woodcock_tracking.h:20-51.  `val` returns the intensity fetched by the accepted
// iteration (= volume(PointOnRay(t)), the scatter point's intensity, pathtracer.cu:241).
// rng_live: a random draw of this path can follow the walk; if not, and the walk provably cannot
// collide, its result (-FLT_MAX) is known without running it.
// Walk set-up: box intersection (woodcock_tracking.h:22-27) + whole-ray test.
//   0 = the walk cannot collide and nothing consumes random numbers after it: result is -FLT_MAX, do not run;
//   1 = run walk_run(tMin, tMax, t_occ);  -1 = the ray misses the (clipped) box: result is -FLT_MAX.
template <bool COUNT, bool SKIP, typename LDS>
SVR_DEV int walk_setup(const DevScene& s, const LDS& L, v3 orig, v3 dir, bool rng_live, float& tMin, float& tMax, float& t_occ)
{
    float tNear, tFar;
    if (!volume_intersect(s, orig, dir, tNear, tFar)) return -1;
    tMin = tNear < 0.f ? (float)1e-6 : tNear;
    tMax = tFar;
    t_occ = tMin;                       // fetches may be needed from here on
    if (SKIP && s.ray_skip) {
        t_occ = first_occupied(s, L, orig, dir, tMin, tMax);
        // COUNT builds run every walk so that the iteration/tap counters stay the reference's
        if (!COUNT && !rng_live && t_occ == u2f(SVR_INF_BITS)) return 0;
    }
    return 1;
}

// walk_setup for the primary walks of a full wave in frame-major order: the whole-ray test is shared by the lanes
// that trace the same pixel (first_occupied_group).  Every lane of the wave must call it.
template <bool COUNT, bool SKIP, typename LDS>
SVR_DEV int walk_setup_group(const DevScene& s, const LDS& L, uint32_t P2, v3 orig, v3 dir, bool rng_live, float& tMin, float& tMax, float& t_occ,
                             GroupMap& map)
{
    map.valid = false;
    float tNear, tFar;
    const bool hit = volume_intersect(s, orig, dir, tNear, tFar);
    tMin = tNear < 0.f ? (float)1e-6 : tNear;
    tMax = tFar;
    t_occ = tMin;
    if (SKIP && s.ray_skip) {
        bool ok;
        const float g = first_occupied_group(s, L, P2, orig, dir, hit, tMin, tMax, ok, map);
        if (hit) {
            t_occ = ok ? g : first_occupied(s, L, orig, dir, tMin, tMax);
            if (!COUNT && !rng_live && t_occ == u2f(SVR_INF_BITS)) return 0;
        }
    }
    return hit ? 1 : -1;
}

// REMARCH: a walk that comes out of an occupied stretch into clear space (two consecutive iterations in
// deep-empty cells) PARKS: it leaves the iteration loop.  The lanes of a wave reconverge at the loop exit, so the
// parked lanes march again TOGETHER (a per-lane march inside the loop is serialised by divergence and was
// slower than not marching at all).  If nothing occupied lies ahead and no draw follows the walk, its result
// (-FLT_MAX) is known; otherwise it resumes with the new t_occ.  COUNT builds keep iterating instead so that
// the iteration/tap counters stay the reference's.
// MAP: the walk belongs to a pixel group with a GroupMap (primary walks of frame-major launches): an iteration that
// lands in an empty cell asks the map for the next possibly-occupied stretch and, if that lies ahead, goes back to
// fetch-free iterations until then (or ends, if nothing lies ahead and no draw follows the walk).
template <int LAYOUT, bool COUNT, bool SKIP, bool REMARCH, bool MAP, typename LDS>
SVR_DEV float walk_run(const DevScene& s, const LDS& L, v3 orig, v3 dir, Rng& rng, float tMin, float tMax, float t_occ,
                       float& val, bool rng_live, Cnt& c, const GroupMap* map = nullptr, uint32_t P2 = 0u)
{
    float t = tMin;
    const bool ray_skippable = SKIP && s.ray_skip && !rng_live && t_occ == u2f(SVR_INF_BITS);
    if (COUNT && ray_skippable) c.wskip++;
    bool tail_counted = false;          // COUNT builds only: a non-counting build would have ended the walk
    uint32_t guard = 0;
    #pragma nounroll
    for (;;) {
        // ---- prefix: iterations before the first possibly-occupied macro-cell.  No fetch, sigma_t = 0: each is a
        //      distance draw + log, the exit test, and the state update of the accept draw (its value is unused) ----
        bool pending = false;           // t has been advanced and still needs its tap and accept draw
        if (SKIP) {
            // Two generator steps per iteration (distance draw, accept draw): the loop runs 5 iterations per trip with the
            // generator as a circular buffer (rng_xorshift_rot, heads 0 2 4 1 3), so no state word is ever moved -- the
            // shifting form spent 14-19 of its ~65 vector instructions per iteration on v_mov -- and with the Weyl word
            // advanced once per trip.  One exit test: t >= t_stop covers `t > tMax` (woodcock_tracking.h:36) and `the first
            // possibly-occupied cell is reached`.  A lane leaves from iteration `jx` of a trip after 2 jx + 1 steps.
            const float t_stop = t_occ <= tMax ? t_occ : next_up(tMax);
            uint32_t jx = 0u, d0 = rng.d;
            auto prefix_iter = [&](auto hd, auto jj) -> bool {
                constexpr int H = decltype(hd)::value;
                constexpr uint32_t J = (uint32_t)decltype(jj)::value;
                if (COUNT) { c.iters++; if (ray_skippable || tail_counted) c.iskip++; else c.ipre++; }
                const uint32_t x = rng_xorshift_rot<H>(rng) + (d0 + (2u * J + 1u) * RNG_WEYL);
                t += -logf_unit(1.f - rng_to_uniform(x)) * s.invSigmaMaxSI;
                jx = J;
                if (t >= t_stop) return true;
                if (COUNT) c.taps++;
                rng_xorshift_rot<(H + 1) % 5>(rng);
                return false;
            };
            #pragma nounroll
            for (;;) {
                if ((guard += 5u) > SVR_WALK_GUARD) { rng.d = d0; return -SVR_FLT_MAX; }       // (the hang guard counts trips here: checked every 5th iteration)
                if (prefix_iter(RngHead<0>{}, RngHead<0>{})) break;
                if (prefix_iter(RngHead<2>{}, RngHead<1>{})) break;
                if (prefix_iter(RngHead<4>{}, RngHead<2>{})) break;
                if (prefix_iter(RngHead<1>{}, RngHead<3>{})) break;
                if (prefix_iter(RngHead<3>{}, RngHead<4>{})) break;
                d0 += 10u * RNG_WEYL;
            }
            {
                // back to the shifting form (dead code when no draw of the path follows a walk that ends here)
                const uint32_t steps = 2u * jx + 1u;
                rng.d = d0 + steps * RNG_WEYL;
                rng_canon(rng, steps >= 5u ? steps - 5u : steps);
            }
            if (t > tMax) return -SVR_FLT_MAX;
            if (COUNT) c.taps++;
            pending = true;
            if (COUNT) { c.ipre -= !(ray_skippable || tail_counted); }
        }
        // ---- general iterations (loop rotated: tap first, then advance) ----
        uint32_t clear_run = 0;
        #pragma nounroll
        for (;;) {
            if (!pending) {
                if (COUNT) { c.iters++; if (tail_counted) c.iskip++; }
                t += -logf_unit(1.f - rng_uniform(rng)) * s.invSigmaMaxSI;
                if (t > tMax || guard++ >= SVR_WALK_GUARD) return -SVR_FLT_MAX;
                if (COUNT) c.taps++;
            }
            pending = false;
            float sigma_t = 0.f;
            bool park_now = false;
            // the accept draw does not depend on the fetch: draw it first (same position in the stream), so that a fetch
            // the draw rules out (majorant-bound culling, svr_accel.hip) is never issued
            const float xi = rng_uniform(rng);
            if (!MAP || t >= t_occ) {
                v3 p = orig + dir * t;
                Cell cell = cell_of(s, p);
                CellInfo ci;
                ci.empty = false; ci.deep = false; ci.thr = u2f(SVR_INF_BITS);
                if (SKIP) ci = cell_info<REMARCH && !MAP>(L, s, cell);
                if (!ci.empty) {
                    clear_run = 0;
                    if (xi < ci.thr) {
                        if (COUNT) c.exec++;
                        val = tex_fetch<LAYOUT>(s, cell) * s.densityScale;
                        sigma_t = alpha_of(L, s, val);
                    } else if (COUNT) c.cull++;
                } else if (SKIP && MAP) {
                    if (map->valid) {
                        t_occ = group_map_next(*map, P2, t);          // <= t while the walk is in or next to an occupied stretch
                        if (t_occ == u2f(SVR_INF_BITS) && !rng_live) {
                            if (!COUNT) {
                                // consume nothing further: the walk cannot collide any more and no draw follows it
                                return -SVR_FLT_MAX;
                            }
                            if (!tail_counted) { tail_counted = true; c.wskip++; }
                        }
                    }
                } else if (SKIP && REMARCH) {
                    // park after two consecutive iterations in DEEP-empty cells (where a march can start)
                    clear_run = ci.deep ? clear_run + 1u : 0u;
                    park_now = clear_run == 2u;
                }
            }
            // the accept draw is consumed either way; with sigma_t == 0 it cannot accept (xi > 0)
            if (xi < sigma_t * s.invSigmaMax) return t;
            if (park_now) break;
        }
        // ---- parked lanes of the wave march together ----
        t_occ = first_occupied(s, L, orig, dir, t, tMax);
        if (t_occ == u2f(SVR_INF_BITS) && !rng_live) {
            if (!COUNT) return -SVR_FLT_MAX;
            if (!tail_counted) { tail_counted = true; c.wskip++; }
        }
    }
}

// sample_distance in one piece (tile kernel)
template <int LAYOUT, bool COUNT, bool SKIP, bool REMARCH, typename LDS>
SVR_DEV float walk(const DevScene& s, const LDS& L, v3 orig, v3 dir, Rng& rng, float& tMin, float& tMax,
                   float& val, bool rng_live, Cnt& c)
{
    float t_occ;
    int r = walk_setup<COUNT, SKIP>(s, L, orig, dir, rng_live, tMin, tMax, t_occ);
    if (r <= 0) return -SVR_FLT_MAX;
    return walk_run<LAYOUT, COUNT, SKIP, REMARCH, false>(s, L, orig, dir, rng, tMin, tMax, t_occ, val, rng_live, c);
}

} // namespace svr
