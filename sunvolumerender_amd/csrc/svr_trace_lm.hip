// svr_trace_lm.hip -- OPT-IN path tracing with LOCAL majorants (SVR_OPT_LOCAL_MAJORANT, default off): BASELINE.json's
// "Woodcock max-density acceleration".
//
// The reference's sample_distance (core/woodcock_tracking.h:20-51) steps with ONE global majorant sigma_max =
// tf.GetMaxOpacity() (:29-31): every path pays sigma_max x chord iterations, also through air.  The bit-exact kernel
// (svr_trace_tile.hip) removes the FETCHES of the iterations that provably cannot collide but must still run them --
// they consume the path's random numbers.  Here the free-flight sampler is delta tracking against a piecewise-constant
// majorant instead (Szirmay-Kalos et al. 2011): the half-resolution macro grid already holds, per cell, a bound
// b = (largest transfer-function alpha any fetch in the cell can return) / sigma_max as a 4-bit class (svr_accel.hip,
// k_bound_class).  A walk draws its free path S = -log(1 - xi) / sigma_max ONCE (woodcock_tracking.h:34), then a 3D-DDA
// spends it at rate b per unit length cell by cell; where it runs out there is a tentative collision: one fetch, accepted
// with probability (sigma_t / sigma_max) / b (woodcock_tracking.h:43 with the local majorant); on rejection a new S is
// drawn and the walk goes on from there.  Cells of class 0 (alpha exactly 0) cost one DDA step and no draw at all.
//
// What is preserved: the law of the collision point (first event of the inhomogeneous Poisson process with rate
// min(sigma_t, sigma_max) -- the global-majorant walk realises exactly the same law), hence every estimator of the
// reference built on it (binary transmittance to the box exit, transmittance.h:10-17; next-event estimation; the path
// loop of pathtracer.cu:216-277) keeps its distribution, and the converged image is the reference's.
// What is NOT preserved: the consumption of random numbers -- a path's radiance is no longer the oracle's bit for bit.
// The contract of this mode is therefore the converged-image tolerance of tests/test_local_majorant_gpu.py, like
// SVR_OPT_FAST_MATH.  Everything else (camera, shading, lights, BSDF, accumulation) is the bit-exact code.
//
// Three forms of one algorithm, bit-identical to each other: straight-line paths (k_trace_lm: non-folding launches, cross-check),
// the record pool of traceDepth 1 (k_trace_lm_pool) and the slot-per-path pool of deeper paths (k_trace_lm_pool_deep).
#include "svr_walk.hpp"
#include "svr_lanes.hpp"
#include "svr_tile_tasks.hpp"

#ifndef SVR_LM_DEEP_COLD
#define SVR_LM_DEEP_COLD 1           // the slot-per-path pool of deeper paths: lights in LDS, set-up / shading / settling constants through the laundered kernarg pointer
#endif
#ifndef SVR_LM_LDS_LIGHTS
#define SVR_LM_LDS_LIGHTS 1          // the depth-1 pool reads the lights from a copy in LDS (light sampling and settling index them per lane)
#endif

namespace svr {

// bound of a class relative to sigma_max, and its reciprocal: b = min(bound_thr(c), 1) = 2^((c - 15) / 2) (svr_accel.hip)
SVR_DEV void class_bound(uint32_t cl, float& b, float& rb)
{
    const int k = (int)cl - 15;                         // -14 .. 0
    const int e = k >> 1;                               // floor(k / 2)
    const uint32_t odd = (uint32_t)k & 1u;
    const uint32_t man = odd ? 0x3504f3u : 0u;          // mantissa of sqrt(2)
    b = u2f(((uint32_t)(127 + e) << 23) | man);
    rb = u2f(((uint32_t)(127 - e - (int)odd) << 23) | man);
}

// The grid a walk steps through (wave-uniform).  Scenes with exactly transparent space walk the FULL-resolution macro grid:
// a macro-cell whose `empty` bit is set has bound 0, any other the bound class of its half-resolution parent -- half the
// wasted tentative collisions in the cells that straddle a surface, and a walk may start anywhere in the empty space in
// front of the first occupied macro-cell with the same result (see lm_begin).  Scenes without (noisy air: c3n) walk the
// half-resolution grid of the classes themselves: half the steps.
struct LmGrid { float sc; int gx, gy, gz, sy, sz; bool fine; };
SVR_DEV LmGrid lm_grid(const DevScene& s)
{
    LmGrid g;
    g.fine = s.has_empty != 0u;
    g.sc = g.fine ? 1.f : 0.5f;
    g.gx = g.fine ? s.mc_gx : s.mc_hgx;
    g.gy = g.fine ? s.mc_gy : (s.mc_gy + 1) >> 1;
    g.gz = g.fine ? s.mc_gz : (s.mc_gz + 1) >> 1;
    g.sy = g.gx;
    g.sz = g.fine ? s.mc_gxy : s.mc_hgxy;
    return g;
}

// a walk in flight: grid coordinate = A + t / r per axis; (tx, ty, tz) = ray parameter of the next cell boundary per axis
struct LmWalk {
    float t, tMax, S, tx, ty, tz, Ax, Ay, Az, rx, ry, rz, b, rb;
    int ix, iy, iz, q;
    uint32_t run;              // consecutive empty cells crossed (a leap is tried from the third on)
};

// cell of the point at ray parameter t, and the (absolute) ray parameters of its far boundaries
SVR_DEV void lm_locate(const LmGrid& g, LmWalk& w, float t)
{
    const float INF = u2f(SVR_INF_BITS);
    // grid coordinate = A + t / r (r = 1 / B; an axis the ray does not move along has r = +-inf and a boundary at infinity)
    const float Bx = __builtin_amdgcn_rcpf(w.rx), By = __builtin_amdgcn_rcpf(w.ry), Bz = __builtin_amdgcn_rcpf(w.rz);
    w.ix = min(max((int)__builtin_floorf(fma_(Bx, t, w.Ax)), 0), g.gx - 1);
    w.iy = min(max((int)__builtin_floorf(fma_(By, t, w.Ay)), 0), g.gy - 1);
    w.iz = min(max((int)__builtin_floorf(fma_(Bz, t, w.Az)), 0), g.gz - 1);
    w.tx = Bx != 0.f ? ((float)(w.ix + (Bx > 0.f ? 1 : 0)) - w.Ax) * w.rx : INF;
    w.ty = By != 0.f ? ((float)(w.iy + (By > 0.f ? 1 : 0)) - w.Ay) * w.ry : INF;
    w.tz = Bz != 0.f ? ((float)(w.iz + (Bz > 0.f ? 1 : 0)) - w.Az) * w.rz : INF;
    w.q = w.ix + w.iy * g.sy + w.iz * g.sz;
}

// Start a walk along (o, d) at t0.  Cell boundaries are ABSOLUTE -- functions of the integer cell index, never accumulated --
// so a walk that starts at any t0 of the empty space in front of its first occupied cell spends its free path exactly as
// one started at the box entry: the shared and the per-lane whole-ray tests (which return different, equally valid t0)
// give the same bits, and a frame's radiance stays a pure function of (scene, pixel, frame) in this mode too.
SVR_DEV void lm_begin(const DevScene& s, const LmGrid& g, LmWalk& w, v3 o, v3 d, float t0, float tMax, Rng& rng)
{
    w.Ax = g.sc * fma_(o.x - s.vmin[0], s.mc_scale[0], s.mc_off);
    w.Ay = g.sc * fma_(o.y - s.vmin[1], s.mc_scale[1], s.mc_off);
    w.Az = g.sc * fma_(o.z - s.vmin[2], s.mc_scale[2], s.mc_off);
    const float Bx = g.sc * (d.x * s.mc_scale[0]), By = g.sc * (d.y * s.mc_scale[1]), Bz = g.sc * (d.z * s.mc_scale[2]);
    const float INF = u2f(SVR_INF_BITS);
    w.rx = Bx != 0.f ? __builtin_amdgcn_rcpf(Bx) : INF; w.ry = By != 0.f ? __builtin_amdgcn_rcpf(By) : INF; w.rz = Bz != 0.f ? __builtin_amdgcn_rcpf(Bz) : INF;
    w.t = t0; w.tMax = tMax;
    w.ix = min(max((int)__builtin_floorf(fma_(Bx, t0, w.Ax)), 0), g.gx - 1);
    w.iy = min(max((int)__builtin_floorf(fma_(By, t0, w.Ay)), 0), g.gy - 1);
    w.iz = min(max((int)__builtin_floorf(fma_(Bz, t0, w.Az)), 0), g.gz - 1);
    w.tx = Bx != 0.f ? ((float)(w.ix + (Bx > 0.f ? 1 : 0)) - w.Ax) * w.rx : INF;
    w.ty = By != 0.f ? ((float)(w.iy + (By > 0.f ? 1 : 0)) - w.Ay) * w.ry : INF;
    w.tz = Bz != 0.f ? ((float)(w.iz + (Bz > 0.f ? 1 : 0)) - w.Az) * w.rz : INF;
    w.q = w.ix + w.iy * g.sy + w.iz * g.sz;
    w.b = 1.f; w.rb = 1.f;
    w.run = 0u;
    // free path at the global majorant (woodcock_tracking.h:34); a cell of bound b spends it at rate b
    w.S = -logf_unit(1.f - rng_uniform(rng)) * s.invSigmaMaxSI;
}

// One cell of the DDA.  0 = moved on to the next cell, 1 = tentative collision at w.t (the free path ran out in this cell),
// 2 = the walk has left the box.
template <bool COUNT, typename LDS>
SVR_DEV int lm_step(const DevScene& s, const LDS& L, const LmGrid& g, LmWalk& w, Cnt& c)
{
    const float te = fmin_(fmin_(w.tx, w.ty), fmin_(w.tz, w.tMax));
    if (COUNT) c.ipre++;
    uint32_t cl;
    if (g.fine) {
        // (the three table words are read together -- one LDS round trip per cell instead of two dependent ones)
        const uint32_t hq = (uint32_t)((w.ix >> 1) + (w.iy >> 1) * s.mc_hgx + (w.iz >> 1) * s.mc_hgxy);
#ifndef SVR_LM_LDS_TOGETHER
#define SVR_LM_LDS_TOGETHER 0        // (reading the three table words up front -- one LDS round trip per cell instead of two dependent ones -- was 0.5 % SLOWER on c3 / c5)
#endif
#if SVR_LM_LDS_TOGETHER
        const uint32_t ew = L.emask[(uint32_t)w.q >> 5], cw = L.cls[hq >> 3], dw = L.dist[hq >> 3];
#else
        const uint32_t ew = L.emask[(uint32_t)w.q >> 5];
#define cw L.cls[hq >> 3]
#define dw L.dist[hq >> 3]
#endif
        const bool empty = (ew >> ((uint32_t)w.q & 31u)) & 1u;
        cl = 0u;
        if (!empty) { cl = (cw >> ((hq & 7u) << 2)) & 15u; w.run = 0u; }
        else if (++w.run >= 3u) {
            // Empty space: no free path is spent, so the walk may LEAP.  Every macro-cell within Chebyshev distance dd - 1 of this
            // one is empty (distance field, svr_accel.hip): the ray may advance until its largest-axis displacement is dd - 1
            // cells (first_occupied's sphere tracing, svr_walk.hpp) and pick up the DDA in the cell it lands in.  Boundaries are
            // absolute, so where exactly a leap lands does not change what follows.  (Tried from the third empty cell in a row on:
            // the gaps between the cells of a surface are shorter than that.)
            const uint32_t dd = (dw >> ((hq & 7u) << 2)) & 15u;
#if !SVR_LM_LDS_TOGETHER
#undef cw
#undef dw
#endif
            if (dd >= 4u) {
                const float inv = fmin_(__builtin_fabsf(w.rx), fmin_(__builtin_fabsf(w.ry), __builtin_fabsf(w.rz))) * 0.999f;      // 1 / largest |B|
                const float tl = fma_((float)dd - 1.05f, inv, w.t);
                if (tl >= w.tMax) return 2;
                if (tl > te) { w.t = tl; lm_locate(g, w, tl); return 0; }
            }
        }
    } else
        cl = (L.cls[(uint32_t)w.q >> 3] >> (((uint32_t)w.q & 7u) << 2)) & 15u;
    if (cl != 0u) {
        class_bound(cl, w.b, w.rb);
        uint32_t sub = 0xffu;
        if (g.fine && s.sub8 != nullptr) sub = s.sub8[w.q];
        if (sub == 0xffu) {
            const float room = (te - w.t) * w.b;                 // (<= 0 for the part of a cell behind the start of the walk)
            if (w.S <= room) { w.t = fma_(w.S, w.rb, w.t); return 1; }
            w.S -= fmax_(room, 0.f);
        } else {
            // A surface cuts this macro-cell: the free path is spent only in its occupied eighths (k_sub8).  The cell's three
            // mid-planes cut the segment [t, te] into at most 4 pieces, each inside one fine cell (found from its midpoint).
            const float INF = u2f(SVR_INF_BITS);
            const float Bx = __builtin_amdgcn_rcpf(w.rx), By = __builtin_amdgcn_rcpf(w.ry), Bz = __builtin_amdgcn_rcpf(w.rz);     // (1 / inf = 0: an axis the ray does not move along)
            const float hx = (float)w.ix + 0.5f, hy = (float)w.iy + 0.5f, hz = (float)w.iz + 0.5f;
            const float mx = Bx != 0.f ? (hx - w.Ax) * w.rx : INF, my = By != 0.f ? (hy - w.Ay) * w.ry : INF, mz = Bz != 0.f ? (hz - w.Az) * w.rz : INF;
            float ta = w.t;
            #pragma nounroll
            for (int k = 0; k < 4 && ta < te; ++k) {
                float tb = te;
                tb = mx > ta ? fmin_(tb, mx) : tb;
                tb = my > ta ? fmin_(tb, my) : tb;
                tb = mz > ta ? fmin_(tb, mz) : tb;
                const float tm = 0.5f * (ta + tb);
                const uint32_t child = (fma_(Bx, tm, w.Ax) >= hx ? 1u : 0u) | (fma_(By, tm, w.Ay) >= hy ? 2u : 0u) | (fma_(Bz, tm, w.Az) >= hz ? 4u : 0u);
                if ((sub >> child) & 1u) {
                    const float room = (tb - ta) * w.b;
                    if (w.S <= room) { w.t = fma_(w.S, w.rb, ta); return 1; }
                    w.S -= room;
                }
                ta = tb;
            }
        }
    }
    if (te >= w.tMax) return 2;
    w.t = fmax_(te, w.t);
    // The LAST cell of an axis also holds what lies beyond it (svr_accel.hip: its bound covers the trilinear cell N - 1, the
    // half voxel between the last voxel centre and the box face): a walk that crosses the grid's upper end stays in that cell
    // until it leaves the box.  (Below 0 there is nothing: the box starts inside cell 0.)
    const float INF = u2f(SVR_INF_BITS);
    if (w.tx <= w.ty && w.tx <= w.tz) {
        const int sx = w.rx > 0.f ? 1 : -1;
        if (w.ix + sx >= g.gx) w.tx = INF;
        else {
            w.ix += sx; w.q += sx;
            if (w.ix < 0) return 2;
            w.tx = ((float)(w.ix + (sx > 0 ? 1 : 0)) - w.Ax) * w.rx;
        }
    } else if (w.ty <= w.tz) {
        const int sy = w.ry > 0.f ? 1 : -1;
        if (w.iy + sy >= g.gy) w.ty = INF;
        else {
            w.iy += sy; w.q += sy * g.sy;
            if (w.iy < 0) return 2;
            w.ty = ((float)(w.iy + (sy > 0 ? 1 : 0)) - w.Ay) * w.ry;
        }
    } else {
        const int sz = w.rz > 0.f ? 1 : -1;
        if (w.iz + sz >= g.gz) w.tz = INF;
        else {
            w.iz += sz; w.q += sz * g.sz;
            if (w.iz < 0) return 2;
            w.tz = ((float)(w.iz + (sz > 0 ? 1 : 0)) - w.Az) * w.rz;
        }
    }
    return 0;
}

// The tentative collision at w.t: fetch, accept with the reference's probability (woodcock_tracking.h:43) over the local
// majorant's share b of sigma_max.  true = collision (val = the fetched intensity); false = the next free path is drawn.
template <int LAYOUT, bool COUNT, typename LDS>
SVR_DEV bool lm_tentative(const DevScene& s, const LDS& L, const LmGrid& g, LmWalk& w, v3 o, v3 d, Rng& rng, float& val, Cnt& c)
{
    if (COUNT) c.iters++;
    const Cell cell = cell_of(s, o + d * w.t);
    bool fetch = true;
    if (!g.fine && s.has_empty) fetch = !cell_is_empty<false>(L, s, cell);       // (the fine grid has the `empty` bit in its bounds already)
    if (fetch) {
        if (COUNT) { c.exec++; c.taps++; }
        val = tex_fetch<LAYOUT>(s, cell) * s.densityScale;
        const float ratio = alpha_of(L, s, val) * s.invSigmaMax;
        if (rng_uniform(rng) * w.b < ratio) return true;
    }
    w.S = -logf_unit(1.f - rng_uniform(rng)) * s.invSigmaMaxSI;
    return false;
}

// sample_distance (woodcock_tracking.h:20-51) against the local majorants, from ray parameter t0 (>= the reference's
// tMin, in the empty space in front of the first occupied cell) to tMax: the collision's t, or -FLT_MAX.  val = the
// intensity fetched at the collision (pathtracer.cu:241).
template <int LAYOUT, bool COUNT, typename LDS>
SVR_DEV float walk_lm(const DevScene& s, const LDS& L, v3 orig, v3 dir, Rng& rng, float t0, float tMax, float& val, Cnt& c, bool dbg = false)
{
    const LmGrid g = lm_grid(s);
    LmWalk w;
    lm_begin(s, g, w, orig, dir, t0, tMax, rng);
    for (uint32_t guard = 0; guard < SVR_WALK_GUARD; ++guard) {               // (hang guard, as in svr_walk.hpp: unreachable for sane scenes)
        // DDA: on to this walk's next tentative collision (all lanes of the wave, each through its own cells) ...
        int r;
        #pragma nounroll
        do {
            r = lm_step<COUNT>(s, L, g, w, c);
#ifdef SVR_LM_DEBUG
            if (dbg) printf("  step r=%d t=%g S=%g cell=(%d,%d,%d) q=%d tx=%g ty=%g tz=%g tMax=%g b=%g\n", r, w.t, w.S, w.ix, w.iy, w.iz, w.q, w.tx, w.ty, w.tz, w.tMax, w.b);
#endif
        } while (r == 0);
        if (r == 2) return -SVR_FLT_MAX;
        // ... then the lanes that found one fetch together
        const bool acc = lm_tentative<LAYOUT, COUNT>(s, L, g, w, orig, dir, rng, val, c);
#ifdef SVR_LM_DEBUG
        if (dbg) printf("  tentative t=%g val=%g accepted=%d newS=%g\n", w.t, val, (int)acc, w.S);
#endif
        if (acc) return w.t;
    }
    return -SVR_FLT_MAX;
}

// Phase profile of experiment builds (-DSVR_TEST_HOOKS; tools/lm_phase_prof.py): wave time (s_memtime) per phase of the pool kernel and the same
// weighted by the lanes that had work in it, accumulated PER WAVE and added to the counters once, when the wave ends (an atomic per phase
// and turn would be what the profile measures).
#if SVR_PROF
struct ProfLocal { unsigned long long cyc[PH_N], lc[PH_N]; };
#define LPROF_BEGIN(name) const uint64_t name = __builtin_amdgcn_s_memtime()
#define LPROF_END(name, ph, lanes) do { const uint64_t dt_ = __builtin_amdgcn_s_memtime() - name; pl.cyc[ph] += dt_; pl.lc[ph] += dt_ * (uint64_t)(lanes); } while (0)
#else
struct ProfLocal {};
#define LPROF_BEGIN(name)
#define LPROF_END(name, ph, lanes)
#endif

#ifndef SVR_LM_WAVES_PER_EU
#define SVR_LM_WAVES_PER_EU 4
#endif
constexpr uint32_t LM_THREADS = 1024;
static_assert(TILE_WAVES == LM_THREADS / 64, "the pending-radiance rows (svr_kernels.hpp) are sized for 16 waves per block");

// one path: kernel_pathtracer's body (pathtracer.cu:205-277) with walk_lm as its sample_distance
template <int LAYOUT, bool COUNT, bool DEPTH1, typename LDS>
SVR_DEV v3 trace_path_lm(const DevScene& s, const LDS& L_, uint32_t x, uint32_t y, uint32_t traceDepth_, uint32_t hashed, bool group_march,
                         uint32_t P2, GroupMapShared* gslot, Cnt& c)
{
    const float INF = u2f(SVR_INF_BITS);
    const uint32_t traceDepth = DEPTH1 ? 1u : traceDepth_;
    Rng rng;
    rng_init(rng, hashed + (y * s.imageW + x));
    if (COUNT) c.paths++;
    v3 L = V3(0.f, 0.f, 0.f), T = V3(1.f, 1.f, 1.f);
    v3 orig, dir;
    camera_ray(s, x, y, rng, orig, dir);
    float ls_t;
    const int ls_id = nearest_light(s, orig, dir, ls_t);
#ifdef SVR_LM_DEBUG
    const bool dbg = x == SVR_LM_DEBUG_X && y == SVR_LM_DEBUG_Y && hashed == wang_hash(SVR_LM_DEBUG_F);
    if (dbg) printf("path (%u,%u) orig=(%g,%g,%g) dir=(%g,%g,%g) ls_id=%d\n", x, y, orig.x, orig.y, orig.z, dir.x, dir.y, dir.z, ls_id);
#else
    const bool dbg = false;
#endif
    for (uint32_t k = 0; k < traceDepth; ++k) {
        float tMin = (float)1e-6, tMax = SVR_FLT_MAX, val = 0.f, t = -SVR_FLT_MAX;
        if (k == 0 && group_march) {
            // the wave is full and its lanes are pixels x frames: one shared whole-ray test per pixel (svr_walk.hpp).  No draw
            // has to be consumed for parity here, so a ray with nothing occupied ahead is simply over
            float t_occ;
            GroupMap map;
            map.g = gslot + ((threadIdx.x & 63u) & ((1u << P2) - 1u) & (GROUP_MAPS_PER_WAVE - 1u));
            const int r = walk_setup_group<true, true>(s, L_, P2, orig, dir, false, tMin, tMax, t_occ, map);
            if (r > 0 && t_occ != INF) t = walk_lm<LAYOUT, COUNT>(s, L_, orig, dir, rng, fmax_(t_occ, tMin), tMax, val, c);      // (the shared test starts at the group's earliest box entry)
        } else {
            float tNear, tFar;
            if (volume_intersect(s, orig, dir, tNear, tFar)) {                // woodcock_tracking.h:22-27
                tMin = tNear < 0.f ? (float)1e-6 : tNear;
                tMax = tFar;
                // primary rays start at the first possibly-occupied parameter (per-lane whole-ray test), like those of a full wave
                const float t0 = k == 0 ? first_occupied(s, L_, orig, dir, tMin, tMax) : tMin;
                if (t0 != INF) t = walk_lm<LAYOUT, COUNT>(s, L_, orig, dir, rng, t0, tMax, val, c, dbg);
            }
        }
        if (k == 0 && ls_id >= 0) {                                           // pathtracer.cu:220-229
            t = t < 0.f ? SVR_FLT_MAX : t;
            if (ls_t < t) {
                const DevLight& l = s.lights[ls_id];
                const float cosTerm = dot(V3(l.normal[0], l.normal[1], l.normal[2]), -dir);
                L = L + (T * V3(l.radiance[0], l.radiance[1], l.radiance[2])) * (cosTerm <= 0.f ? 0.f : 1.f);
                break;
            }
        }
        if (t < 0.f) {                                                        // pathtracer.cu:231-235
            if (s.env_on_escape) L = L + T * env_radiance(s, dir);
            break;
        }
        Shade vs;
        vs.wo = -dir;
        vs.pt = orig + dir * t;
        Nee ne;
        shade_event<LAYOUT, COUNT>(s, vs, val, rng, ne, c);                   // VolumeSample + light sampling, pathtracer.cu:237-257
#ifdef SVR_LM_DEBUG
        if (dbg) printf(" hit t=%g pt=(%g,%g,%g) val=%g color=(%g,%g,%g,%g) grad=(%g,%g,%g) Pbrdf=%g st=%d have=%d wi=(%g,%g,%g) B=(%g,%g,%g) pdf=%g light=%u\n", t, vs.pt.x, vs.pt.y, vs.pt.z, val,
                        vs.color[0], vs.color[1], vs.color[2], vs.color[3], vs.gradient.x, vs.gradient.y, vs.gradient.z, vs.Pbrdf, vs.st, (int)ne.have, ne.wi.x, ne.wi.y, ne.wi.z, ne.B.x, ne.B.y, ne.B.z, ne.pdf, ne.light);
#endif
        if (ne.have) {
            // transmittance (transmittance.h:10-17): a walk along the light direction to the box exit, 0 if it collides
            float sNear, sFar, sval = 0.f, ts = -SVR_FLT_MAX;
            float sMin = (float)1e-6, sMax = SVR_FLT_MAX;
            if (volume_intersect(s, vs.pt, ne.wi, sNear, sFar)) {
                sMin = sNear < 0.f ? (float)1e-6 : sNear;
                sMax = sFar;
                ts = walk_lm<LAYOUT, COUNT>(s, L_, vs.pt, ne.wi, rng, sMin, sMax, sval, c, dbg);
            }
            const float Tr = ((ts > sMin) && (ts < sMax)) ? 0.f : 1.f;
            const float kf = Tr * (float)s.num_lights;
            const DevLight& l = s.lights[ne.light];
            L = L + T * (((ne.B * kf) * V3(l.radiance[0], l.radiance[1], l.radiance[2])) / ne.pdf);
        }
        if (k + 1u >= traceDepth) break;
        v3 wi; float pdf = 0.f;
        const v3 f = bsdf_sample(vs, wi, pdf, rng);
        const float cosTerm = __builtin_fabsf(dot(normalize(vs.gradient), wi));
        if (fmax_(f.x, fmax_(f.y, f.z)) > 0.f && pdf > 0.f) {
            if (vs.st == 0) T = T * (f / (pdf * (1.f - vs.Pbrdf)));
            else T = T * ((f * cosTerm) / (pdf * vs.Pbrdf));
        }
        orig = vs.pt;
        dir = wi;
        if (k >= 3 && russian_roulette(T, rng)) break;
    }
    return L;
}


// ------------------------------------------------------------------------------------------------------------------
// The POOL form of the same algorithm (traceDepth 1, folding launches): a wave-sized wavefront path tracer.
//
// Why.  In the straight-line form a wave's 64 paths advance in lockstep: a walk round (DDA to the next tentative collision,
// one fetch) is paid by the whole wave until its slowest lane is through, and the counts are geometric -- PMC: as many
// vector instructions as the global-majorant kernel at 35 % (c3) and 21 % (c3n) lane utilisation.  So a wave takes a
// BATCH of LM_BATCH tasks (<= 1024 paths) through four stages, each of which keeps its lanes busy:
//   gen     per task, in lockstep (the lanes are frames of the same pixels): generator, camera ray, light hit test, box,
//           shared whole-ray test.  Rays with nothing occupied ahead end here (light / environment); the others become
//           RAY RECORDS in the wave's queue memory;
//   walk    a pool: every lane pops a ray record, walks it (walk state in registers), settles it, pops the next.  A
//           collision becomes a HIT RECORD, a miss ends the path;
//   shade   64 hit records at a time: VolumeSample, gradient, light sampling (pathtracer.cu:237-257) -- each a shadow RAY RECORD;
//   walk    the pool again, over the shadow rays: transmittance x prepared estimate -> the path's radiance;
//   fold    the batch's tasks into the accumulator (svr_tile_tasks.hpp).
// Records live in the wave's slice of DevWork.queue (written and read back once by the same wave: L2 / Infinity Cache).
// The radiance of a path goes to its task's row by id = (task-in-batch << 6 | lane), like the QUEUE builds of
// svr_trace_tile.hip.  Every path executes the operations of trace_path_lm in the same order on its own generator, so the
// pool is scheduling only: bit-identical to the straight-line form (tests/test_local_majorant_gpu.py).
// ------------------------------------------------------------------------------------------------------------------
#ifndef SVR_LM_BATCH
#define SVR_LM_BATCH 16
#endif
constexpr uint32_t LM_BATCH = SVR_LM_BATCH;               // tasks per batch: <= LM_CAP ray records
constexpr uint32_t LM_CAP = LM_BATCH * 64;
// the record pool of traceDepth 1: up to 23 tasks per batch (the record memory of a wave); how many it takes is a run-time choice (DevScene.lm_tune bits 24-31:
// scenes with transparent space do best with ~10 -- their records stay closer to the L2 --, fog with 21 = 63 of 64 lanes in the fold)
#ifndef SVR_LM_BATCH1
#define SVR_LM_BATCH1 23
#endif
constexpr uint32_t LM_BATCH1 = SVR_LM_BATCH1;
constexpr uint32_t LM_CAP1 = LM_BATCH1 * 64;
static_assert(LM_CAP1 <= 2048, "a record's path id has 11 bits");
constexpr uint32_t LM_RAY_WORDS = 17;                     // o(3) d(3) rng(6) meta p0..p3
constexpr uint32_t LM_HIT_WORDS = 14;                     // pt(3) wo(3) val rng(6) meta
static_assert((LM_RAY_WORDS + LM_HIT_WORDS) * LM_CAP1 <= REC_WORDS * QUEUE_CAP, "the pool's records fit the wave's queue slice");
static_assert(LM_BATCH <= QUEUE_TASKS && LM_BATCH1 <= QUEUE_TASKS, "pending-radiance rows");
// meta: id (11 bits: task-in-batch << 6 | lane) | light or (nearest light + 1) << 12
SVR_DEV uint32_t lm_meta(uint32_t id, uint32_t light) { return id | (light << 12); }

template <int LAYOUT, bool COUNT, uint32_t NB, typename LDS>
SVR_DEV void lm_walk_pool(const DevScene& s, const LDS& L_, const uint32_t* R, uint32_t n, const bool shadows, uint32_t* H, uint32_t& nH,
                          float* pendL, Cnt& c, ProfLocal& pl, const DevScene* scp = nullptr, const DevLight* lts = nullptr)
{
    constexpr uint32_t cap = NB * 64u;                                 // records per stage of a batch = the stride of a record's words
    const DevScene& sc = scp ? *scp : s;                               // (lights / environment of the settling: svr_lanes.hpp, shade_event)
    const DevLight* const LT = SVR_LM_LDS_LIGHTS ? lts : sc.lights;   // (the kernel's copy of the lights in LDS: indexed per lane)
    enum : uint32_t { IDLE = 0u, WALK = 1u, TENT = 2u, END = 3u };
    const LmGrid g = lm_grid(s);
    const uint32_t steps_per_turn = s.lm_tune & 0xffu, refill_min = (s.lm_tune >> 8) & 0xffu, ended_min = (s.lm_tune >> 16) & 0xffu;
    uint32_t st = IDLE, next = 0u;
    // walk state
    Rng rng = {0u, 0u, 0u, 0u, 0u, 0u};
    v3 o = V3(0.f, 0.f, 0.f), d = V3(0.f, 0.f, 1.f);
    LmWalk wk;
    wk.t = wk.tMax = wk.S = wk.tx = wk.ty = wk.tz = wk.Ax = wk.Ay = wk.Az = 0.f; wk.rx = wk.ry = wk.rz = wk.b = wk.rb = 1.f;
    wk.ix = wk.iy = wk.iz = wk.q = 0; wk.run = 0u;
    float tMin = 0.f, val = 0.f;
    bool hit = false;
    uint32_t meta = 0u;
    float p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f;
    auto put = [&](v3 L) {
        const uint32_t id = meta & 0x7ffu;
        float* p = pendL + (id >> 6) * (3u * 64u) + (id & 63u);
        p[0] = L.x; p[64] = L.y; p[128] = L.z;
        st = IDLE;
    };

    for (uint32_t turn = 0; turn < (SVR_WALK_GUARD << 4); ++turn) {           // (hang guard: unreachable for sane scenes)
        // ---- refill: idle lanes pop the next ray records ----
        {
            const uint64_t idle = __ballot(st == IDLE);
            const uint32_t n_idle = (uint32_t)__popcll(idle);
            if (next < n && (n_idle >= refill_min || __ballot(st == WALK || st == TENT) == 0ull)) {
                LPROF_BEGIN(prf);
                const uint32_t i = next + lane_rank(idle);
                if (st == IDLE && i < n) {
                    const uint32_t* r = R + i;
                    o = rec_v3_load(r, cap); d = rec_v3_load(r + 3 * cap, cap);
                    rec_rng_load(r + 6 * cap, cap, rng);
                    meta = r[12 * cap];
                    p0 = u2f(r[13 * cap]); p1 = u2f(r[14 * cap]); p2 = u2f(r[15 * cap]); p3 = u2f(r[16 * cap]);
                    if (shadows) {
                        // transmittance (transmittance.h:10-17): to the box exit along the light direction
                        float sNear, sFar;
                        hit = false;
                        tMin = (float)1e-6; wk.tMax = SVR_FLT_MAX;
                        if (volume_intersect(s, o, d, sNear, sFar)) {
                            tMin = sNear < 0.f ? (float)1e-6 : sNear;
                            lm_begin(s, g, wk, o, d, tMin, sFar, rng);
                            st = WALK;
                        } else st = END;
                    } else {
                        // primary: the gen stage has intersected the box and found the first possibly-occupied parameter
                        hit = false;
                        lm_begin(s, g, wk, o, d, p1, p2, rng);
                        st = WALK;
                    }
                }
                next = min(n, next + n_idle);
                LPROF_END(prf, PH_REFILL, min(n_idle, 64u));
            }
        }
        if (__ballot(st != IDLE) == 0ull) break;                          // no record left, no walk in flight
        // ---- DDA: the walking lanes go on towards their next tentative collision (a few cells per turn) ----
        {
            LPROF_BEGIN(pdd);
#if SVR_PROF
            const uint32_t n_walk = (uint32_t)__popcll(__ballot(st == WALK));
#endif
            #pragma nounroll
            for (uint32_t k = 0; k < steps_per_turn && __ballot(st == WALK) != 0ull; ++k)
                if (st == WALK) {
                    const int r = lm_step<COUNT>(s, L_, g, wk, c);
                    st = r == 0 ? WALK : (r == 1 ? TENT : END);
                }
            LPROF_END(pdd, PH_CHEAP, n_walk);
        }
        // ---- tentative collisions: fetch + accept test ----
        if (__ballot(st == TENT) != 0ull) {
            LPROF_BEGIN(pte);
#if SVR_PROF
            const uint32_t n_tent = (uint32_t)__popcll(__ballot(st == TENT));
#endif
            if (st == TENT) {
                hit = lm_tentative<LAYOUT, COUNT>(s, L_, g, wk, o, d, rng, val, c);
                st = hit ? END : WALK;
            }
            LPROF_END(pte, PH_FETCH, n_tent);
        }
        // ---- settle the walks that are over (when they are many, or nothing else is left to do) ----
        {
            const uint64_t ended = __ballot(st == END);
            if (ended != 0ull && ((uint32_t)__popcll(ended) >= ended_min || __ballot(st == WALK || st == TENT) == 0ull)) {
                LPROF_BEGIN(pse);
                if (shadows) {
                    if (st == END) {
                        // estimate_direct_light's tail (pathtracer.cu:191-198): p0..p2 = bsdf, p3 = pdf
                        const float ts = hit ? wk.t : -SVR_FLT_MAX;
                        const float Tr = ((ts > tMin) && (ts < wk.tMax)) ? 0.f : 1.f;
                        const float kf = Tr * (float)sc.num_lights;
                        const DevLight& l = LT[(meta >> 12) & 15u];
                        put(((V3(p0, p1, p2) * kf) * V3(l.radiance[0], l.radiance[1], l.radiance[2])) / p3);
                    }
                } else {
                    // pathtracer.cu:220-235: the nearest light in front of the collision, else environment, else a scatter event
                    const uint32_t ls = (meta >> 12) & 15u;                    // nearest light + 1, 0 = none
                    bool to_hit = false;
                    if (st == END) {
                        const float tt = hit ? wk.t : SVR_FLT_MAX;
                        if (ls != 0u && p0 < tt) {
                            const DevLight& l = LT[ls - 1u];
                            const float cosTerm = dot(V3(l.normal[0], l.normal[1], l.normal[2]), -d);
                            put(V3(l.radiance[0], l.radiance[1], l.radiance[2]) * (cosTerm <= 0.f ? 0.f : 1.f));
                        } else if (!hit) {
                            put(sc.env_on_escape ? env_radiance(sc, d) : V3(0.f, 0.f, 0.f));
                        } else to_hit = true;
                    }
                    const uint64_t mh = __ballot(to_hit);
                    if (to_hit) {
                        uint32_t* h = H + nH + lane_rank(mh);
                        rec_v3_store(h, cap, o + d * wk.t); rec_v3_store(h + 3 * cap, cap, -d);
                        h[6 * cap] = f2u(val);
                        rec_rng_store(h + 7 * cap, cap, rng);
                        h[13 * cap] = meta & 0x7ffu;
                        st = IDLE;
                    }
                    nH += (uint32_t)__popcll(mh);
                }
                LPROF_END(pse, PH_END, (uint32_t)__popcll(ended));
            }
        }
    }
}

// NB: tasks per batch (compile time: the record stride NB x 64 sits in every address of the pool's loads and stores; as a run-time value it cost 1.4-5 %)
template <int LAYOUT, bool COUNT, uint32_t NB>
__global__ __launch_bounds__(LM_THREADS, SVR_LM_WAVES_PER_EU) void k_trace_lm_pool(const DevScene s, const DevWork w)
{
    static_assert(NB >= 1 && NB <= LM_BATCH1, "tasks per batch");
    __shared__ LdsTileCull lds;
    __shared__ GroupMapShared gmaps[TILE_WAVES][GROUP_MAPS_PER_WAVE];
    __shared__ uint32_t pend_task[TILE_WAVES][QUEUE_TASKS];
    __shared__ DevLight lds_lights[8];                                   // (light sampling and settling index the lights per lane: svr_trace_tile.hip)
    if (threadIdx.x < 8u * (sizeof(DevLight) / 4u)) reinterpret_cast<float*>(lds_lights)[threadIdx.x] = reinterpret_cast<const float*>(s.lights)[threadIdx.x];
    const DevLight* const lts = lds_lights;
    lds_tile_load(lds, s, true);

    const float INF = u2f(SVR_INF_BITS);
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const TaskShape ts = task_shape(w);
    const uint32_t fl2 = ts.fl2, P2 = ts.P2, tw2 = ts.tw2, th2 = ts.th2, wv = ts.wv;
    const uint32_t n_tasks = ts.tiles_x * ts.tiles_y * ts.fgroups;
    const uint32_t shard0 = blockIdx.x % TICKET_SHARDS;
    const size_t wslot = (size_t)(blockIdx.x * TILE_WAVES + wave);
    float* const gpend = w.pend + wslot * (QUEUE_TASKS * 3u * 64u);
    uint32_t* const R = w.queue + wslot * (REC_WORDS * QUEUE_CAP);       // ray records
    constexpr uint32_t batch_tasks = NB, cap = NB * 64u;
    // set-up / shading / settling constants through a laundered pointer into the kernarg segment (svr_trace_tile.hip, cold_scene)
#ifndef SVR_LM_COLD_SCENE
#define SVR_LM_COLD_SCENE 1
#endif
    auto cold_scene = [&]() -> const DevScene* {
#if SVR_LM_COLD_SCENE
        auto p = __builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(p));
        return (const DevScene*)p;
#else
        return nullptr;
#endif
    };
    uint32_t* const H = R + (size_t)LM_RAY_WORDS * cap;                  // hit records
    Cnt c = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    ProfLocal pl = {};
    auto fence = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");            // records and radiance rows are read back by other lanes of this wave
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    };
    uint32_t si = 0u;                                                     // ticket counters visited so far
    for (;;) {
        // ---- gen: up to batch_tasks tasks ----
        LPROF_BEGIN(pgen);
        uint32_t nb = 0u, nR = 0u;
        while (nb < batch_tasks && si < TICKET_SHARDS) {
            const uint32_t shard = (shard0 + si) % TICKET_SHARDS;
            uint32_t* ticket = w.ticket + shard * TICKET_STRIDE;
            // away from the home counter, look before taking (a plain load of a drained counter is free)
            if (si != 0u && __atomic_load_n(ticket, __ATOMIC_RELAXED) * TICKET_SHARDS + shard >= n_tasks) { ++si; continue; }
            uint32_t u = 0;
            if (lane == 0) u = atomicAdd(ticket, 1u);
            u = __builtin_amdgcn_readfirstlane(u);
            const uint32_t k = u * TICKET_SHARDS + shard;
            if (k >= n_tasks) { ++si; continue; }
            uint32_t tx, ty, fg;
            task_decode(ts, k, tx, ty, fg);
            if (COUNT) c.loops += (lane == 0);
            const uint32_t pl = lane & ((1u << P2) - 1u);
            const uint32_t slot = (fg << fl2) + (lane >> P2);
            const uint32_t px = (tx << tw2) + (pl & ((1u << tw2) - 1u));
            const uint32_t r = (ty << th2) + (pl >> tw2);
            const bool live = px < wv && r < w.n_rows && slot < w.nframes;
            const bool group_march = fl2 >= 3u && __ballot(live) == ~0ull;
            // kernel_pathtracer up to the primary walk (pathtracer.cu:205-218), as in trace_path_lm
            bool queued = false;
            v3 L = V3(0.f, 0.f, 0.f), orig = L, dir = V3(0.f, 0.f, 1.f);
            Rng rng = {0u, 0u, 0u, 0u, 0u, 0u};
            float ls_t = 0.f, t0 = 0.f, tMax = 0.f;
            int ls_id = -1;
            if (live) {
                const DevScene* scp = cold_scene();
                const DevScene& sc = scp ? *scp : s;
                const uint32_t x = w.x0 + px, y = owned_row_to_y(w, r);
                rng_init(rng, wang_hash(w.frame0 + slot) + (y * sc.imageW + x));
                if (COUNT) c.paths++;
                camera_ray(sc, x, y, rng, orig, dir);
                ls_id = nearest_light(sc, orig, dir, ls_t);
                float tMin = (float)1e-6;
                tMax = SVR_FLT_MAX;
                bool run = false;
                if (group_march) {
                    float t_occ;
                    GroupMap map;
                    map.g = &gmaps[wave][0] + ((threadIdx.x & 63u) & ((1u << P2) - 1u) & (GROUP_MAPS_PER_WAVE - 1u));
                    const int rr = walk_setup_group<true, true>(s, lds, P2, orig, dir, false, tMin, tMax, t_occ, map);
                    run = rr > 0 && t_occ != INF;
                    t0 = fmax_(t_occ, tMin);                                   // (the shared test starts at the group's earliest box entry)
                } else {
                    float tNear, tFar;
                    if (volume_intersect(s, orig, dir, tNear, tFar)) {
                        tMin = tNear < 0.f ? (float)1e-6 : tNear;
                        tMax = tFar;
                        // (per-lane whole-ray test: the same early end as under the shared one)
                        t0 = first_occupied(s, lds, orig, dir, tMin, tMax);
                        run = t0 != INF;
                    }
                }
                if (run) queued = true;
                else if (ls_id >= 0) {                                        // t = FLT_MAX > ls.t: the light is seen (pathtracer.cu:220-229)
                    const DevLight& l = sc.lights[ls_id];
                    const float cosTerm = dot(V3(l.normal[0], l.normal[1], l.normal[2]), -dir);
                    L = V3(l.radiance[0], l.radiance[1], l.radiance[2]) * (cosTerm <= 0.f ? 0.f : 1.f);
                } else if (sc.env_on_escape) L = env_radiance(sc, dir);
            }
            {
                float* p = gpend + (size_t)nb * (3u * 64u) + lane;           // (queued paths overwrite theirs when they end)
                p[0] = L.x; p[64] = L.y; p[128] = L.z;
            }
            const uint64_t mq = __ballot(queued);
            if (queued) {
                uint32_t* rr = R + nR + lane_rank(mq);
                rec_v3_store(rr, cap, orig); rec_v3_store(rr + 3 * cap, cap, dir);
                rec_rng_store(rr + 6 * cap, cap, rng);
                rr[12 * cap] = lm_meta((nb << 6) | lane, (uint32_t)(ls_id + 1));
                rr[13 * cap] = f2u(ls_t); rr[14 * cap] = f2u(t0); rr[15 * cap] = f2u(tMax); rr[16 * cap] = 0u;
            }
            nR += (uint32_t)__popcll(mq);
            if (lane == 0) pend_task[wave][nb] = k;
            ++nb;
        }
        LPROF_END(pgen, PH_PRIMARY, 64u);
        if (nb == 0u) break;
        // ---- walk: primary rays ----
        uint32_t nH = 0u;
        fence();
        lm_walk_pool<LAYOUT, COUNT, NB>(s, lds, R, nR, false, H, nH, gpend, c, pl, cold_scene(), lts);
        fence();
        // ---- shade the collisions, 64 at a time: each becomes a shadow ray (or ends with L = 0) ----
        LPROF_BEGIN(psh);
        uint32_t nS = 0u;
        for (uint32_t i0 = 0u; i0 < nH; i0 += 64u) {
            const uint32_t i = i0 + lane;
            bool have = false;
            Shade vs;
            vs.pt = V3(0.f, 0.f, 0.f); vs.wo = vs.pt; vs.gradient = vs.pt; vs.color[0] = vs.color[1] = vs.color[2] = vs.color[3] = 0.f; vs.Pbrdf = 0.f; vs.st = 0;
            Nee ne;
            ne.wi = V3(0.f, 0.f, 1.f); ne.B = vs.pt; ne.pdf = 1.f; ne.light = 0u; ne.have = false;
            Rng rng = {0u, 0u, 0u, 0u, 0u, 0u};
            uint32_t id = 0u;
            if (i < nH) {
                const uint32_t* h = H + i;
                vs.pt = rec_v3_load(h, cap); vs.wo = rec_v3_load(h + 3 * cap, cap);
                const float val = u2f(h[6 * cap]);
                rec_rng_load(h + 7 * cap, cap, rng);
                id = h[13 * cap];
                shade_event<LAYOUT, COUNT, SVR_LM_LDS_LIGHTS != 0>(s, vs, val, rng, ne, c, cold_scene(), lts);
                have = ne.have;
                if (!have) {                                                  // no light sample reaches the event: L = 0
                    float* p = gpend + (id >> 6) * (3u * 64u) + (id & 63u);
                    p[0] = 0.f; p[64] = 0.f; p[128] = 0.f;
                }
            }
            const uint64_t ms = __ballot(have);
            if (have) {
                uint32_t* rr = R + nS + lane_rank(ms);                        // (the primary ray records have all been consumed)
                rec_v3_store(rr, cap, vs.pt); rec_v3_store(rr + 3 * cap, cap, ne.wi);
                rec_rng_store(rr + 6 * cap, cap, rng);
                rr[12 * cap] = lm_meta(id, ne.light);
                rr[13 * cap] = f2u(ne.B.x); rr[14 * cap] = f2u(ne.B.y); rr[15 * cap] = f2u(ne.B.z); rr[16 * cap] = f2u(ne.pdf);
            }
            nS += (uint32_t)__popcll(ms);
        }
        LPROF_END(psh, PH_SHADE, nH ? min(64u, (nH + ((nH + 63u) >> 6) - 1u) / ((nH + 63u) >> 6)) : 0u);
        // ---- walk: shadow rays ----
        fence();
        uint32_t none = 0u;
        lm_walk_pool<LAYOUT, COUNT, NB>(s, lds, R, nS, true, H, none, gpend, c, pl, cold_scene(), lts);
        fence();
        // ---- fold the batch ----
        LPROF_BEGIN(pfo);
        fold_pending(s, w, gpend, 64u, &pend_task[wave][0], nb);
        fence();
        LPROF_END(pfo, PH_FOLD, 48u);
    }
#if SVR_PROF
    if (lane == 0u)
        for (uint32_t ph = 0; ph < PH_N; ++ph) { atomicAdd(&w.counters[CNT_N + 2 * ph], pl.cyc[ph]); atomicAdd(&w.counters[CNT_N + 2 * ph + 1], pl.lc[ph]); }
#endif
    if (COUNT) cnt_flush(w, c);
}

// ------------------------------------------------------------------------------------------------------------------
// The pool form for traceDepth > 1: ONE SLOT PER PATH.  A path of the batch owns slot id = (task-in-batch << 6 | lane) of the wave's
// queue slice for its whole life; the slot holds everything that survives between the stages of a bounce (SoA, stride LM_CAP
// words), and the stages hand each other LISTS of slot numbers (dense: ballot + mbcnt):
//   gen -> W | per bounce k:  walk(W) -> H | shade(H) -> S (or A: no light sample) | walk(S) -> A | bsdf(A) -> W | ... | fold
// Every stage runs full waves: the walks as a pool (lm_walk_pool_deep: a lane pops a slot, walks with its state in registers, settles
// the result into the slot, pops the next), shading and BSDF sampling 64 slots at a time.  Each path executes trace_path_lm's
// operations in trace_path_lm's order on its own generator (the shadow walk's draws come before sample_bsdf's, the generator
// lives in the slot between stages), so this, too, is scheduling only: bit-identical to the straight-line form.
// ------------------------------------------------------------------------------------------------------------------
enum : uint32_t { SF_O = 0, SF_D = 3, SF_RNG = 6, SF_META = 12, SF_L = 13, SF_T = 16, SF_WO = 19, SF_GRAD = 22, SF_COLOR = 25, SF_PBRDF = 28, SF_B = 29,
                  SF_PDF = 32, SF_VAL = 33, SF_LST = 34, SF_T0 = 35, SF_TMAX = 36, SF_WORDS = 37 };
constexpr uint32_t LM_LISTS = 4;                                                  // W (walks), H (hits), S (shadow walks), A (BSDF sampling)
static_assert((SF_WORDS + LM_LISTS) * LM_CAP <= REC_WORDS * QUEUE_CAP, "slots + lists fit the wave's queue slice");
// meta: nearest light + 1 (bits 0-3, primary rays) | sampled light (bits 4-7) | shading type (bit 8)
enum : uint32_t { LMK_PRIMARY = 0u, LMK_CONT = 1u, LMK_SHADOW = 2u };

template <int LAYOUT, bool COUNT, typename LDS>
SVR_DEV void lm_walk_pool_deep(const DevScene& s, const LDS& L_, uint32_t* F, const uint32_t* list, uint32_t n, const uint32_t kind, const bool last_bounce,
                               uint32_t* out, uint32_t& n_out, float* pendL, Cnt& c, const DevScene* scp = nullptr, const DevLight* lts = nullptr)
{
    const DevScene& sc = scp ? *scp : s;                               // (lights / environment of the settling, as in lm_walk_pool)
    const DevLight* const LT = SVR_LM_DEEP_COLD ? lts : sc.lights;
    enum : uint32_t { IDLE = 0u, WALK = 1u, TENT = 2u, END = 3u };
    const LmGrid g = lm_grid(s);
    const uint32_t steps_per_turn = s.lm_tune & 0xffu, refill_min = (s.lm_tune >> 8) & 0xffu, ended_min = (s.lm_tune >> 16) & 0xffu;
    uint32_t st = IDLE, next = 0u, slot = 0u;
    Rng rng = {0u, 0u, 0u, 0u, 0u, 0u};
    v3 o = V3(0.f, 0.f, 0.f), d = V3(0.f, 0.f, 1.f);
    LmWalk wk;
    wk.t = wk.tMax = wk.S = wk.tx = wk.ty = wk.tz = wk.Ax = wk.Ay = wk.Az = 0.f; wk.rx = wk.ry = wk.rz = wk.b = wk.rb = 1.f;
    wk.ix = wk.iy = wk.iz = wk.q = 0; wk.run = 0u;
    float tMin = 0.f, val = 0.f;
    bool hit = false;
    auto put = [&](v3 L) {
        float* p = pendL + (slot >> 6) * (3u * 64u) + (slot & 63u);
        p[0] = L.x; p[64] = L.y; p[128] = L.z;
        st = IDLE;
    };
    for (uint32_t turn = 0; turn < (SVR_WALK_GUARD << 4); ++turn) {           // (hang guard: unreachable for sane scenes)
        // ---- refill: idle lanes pop the next slots of the list ----
        {
            const uint64_t idle = __ballot(st == IDLE);
            const uint32_t n_idle = (uint32_t)__popcll(idle);
            if (next < n && (n_idle >= refill_min || __ballot(st == WALK || st == TENT) == 0ull)) {
                const uint32_t i = next + lane_rank(idle);
                if (st == IDLE && i < n) {
                    slot = list[i];
                    const uint32_t* f = F + slot;
                    o = rec_v3_load(f + SF_O * LM_CAP, LM_CAP); d = rec_v3_load(f + SF_D * LM_CAP, LM_CAP);
                    rec_rng_load(f + SF_RNG * LM_CAP, LM_CAP, rng);
                    hit = false;
                    if (kind == LMK_PRIMARY) {
                        // the gen stage has intersected the box and found the first possibly-occupied parameter
                        lm_begin(s, g, wk, o, d, u2f(f[SF_T0 * LM_CAP]), u2f(f[SF_TMAX * LM_CAP]), rng);
                        st = WALK;
                    } else {
                        // sample_distance's box (woodcock_tracking.h:22-27) / transmittance's (transmittance.h:10-17): to the box exit
                        float sNear, sFar;
                        tMin = (float)1e-6; wk.tMax = SVR_FLT_MAX;
                        if (volume_intersect(s, o, d, sNear, sFar)) {
                            tMin = sNear < 0.f ? (float)1e-6 : sNear;
                            lm_begin(s, g, wk, o, d, tMin, sFar, rng);
                            st = WALK;
                        } else st = END;
                    }
                }
                next = min(n, next + n_idle);
            }
        }
        if (__ballot(st != IDLE) == 0ull) break;
        // ---- DDA, tentative collisions (as in lm_walk_pool) ----
        #pragma nounroll
        for (uint32_t k = 0; k < steps_per_turn && __ballot(st == WALK) != 0ull; ++k)
            if (st == WALK) {
                const int r = lm_step<COUNT>(s, L_, g, wk, c);
                st = r == 0 ? WALK : (r == 1 ? TENT : END);
            }
        if (__ballot(st == TENT) != 0ull) {
            if (st == TENT) {
                hit = lm_tentative<LAYOUT, COUNT>(s, L_, g, wk, o, d, rng, val, c);
                st = hit ? END : WALK;
            }
        }
        // ---- settle the walks that are over ----
        {
            const uint64_t ended = __ballot(st == END);
            if (ended != 0ull && ((uint32_t)__popcll(ended) >= ended_min || __ballot(st == WALK || st == TENT) == 0ull)) {
                bool to_out = false;
                if (st == END) {
                    uint32_t* f = F + slot;
                    if (kind == LMK_SHADOW) {
                        // estimate_direct_light's tail (pathtracer.cu:191-198) and L = L + T * Ld (:257)
                        const float ts = hit ? wk.t : -SVR_FLT_MAX;
                        const float Tr = ((ts > tMin) && (ts < wk.tMax)) ? 0.f : 1.f;
                        const float kf = Tr * (float)sc.num_lights;
                        const DevLight& l = LT[(f[SF_META * LM_CAP] >> 4) & 15u];
                        const v3 B = rec_v3_load(f + SF_B * LM_CAP, LM_CAP), T = rec_v3_load(f + SF_T * LM_CAP, LM_CAP);
                        const float pdf = u2f(f[SF_PDF * LM_CAP]);
                        const v3 L = rec_v3_load(f + SF_L * LM_CAP, LM_CAP) + T * (((B * kf) * V3(l.radiance[0], l.radiance[1], l.radiance[2])) / pdf);
                        if (last_bounce) put(L);
                        else {
                            rec_v3_store(f + SF_L * LM_CAP, LM_CAP, L);
                            rec_rng_store(f + SF_RNG * LM_CAP, LM_CAP, rng);
                            to_out = true;                                      // on to the BSDF sampling
                        }
                    } else {
                        // pathtracer.cu:220-235: (k = 0) the nearest light in front of the collision; no collision: environment; else a scatter event
                        const uint32_t ls = kind == LMK_PRIMARY ? (f[SF_META * LM_CAP] & 15u) : 0u;
                        const float tt = hit ? wk.t : SVR_FLT_MAX;
                        if (ls != 0u && u2f(f[SF_LST * LM_CAP]) < tt) {
                            const DevLight& l = LT[ls - 1u];
                            const float cosTerm = dot(V3(l.normal[0], l.normal[1], l.normal[2]), -d);
                            put(V3(l.radiance[0], l.radiance[1], l.radiance[2]) * (cosTerm <= 0.f ? 0.f : 1.f));      // (L = 0, T = 1 at k = 0)
                        } else if (!hit) {
                            v3 L = rec_v3_load(f + SF_L * LM_CAP, LM_CAP);
                            if (sc.env_on_escape) L = L + rec_v3_load(f + SF_T * LM_CAP, LM_CAP) * env_radiance(sc, d);
                            put(L);
                        } else {
                            rec_v3_store(f + SF_O * LM_CAP, LM_CAP, o + d * wk.t); rec_v3_store(f + SF_WO * LM_CAP, LM_CAP, -d);
                            f[SF_VAL * LM_CAP] = f2u(val);
                            rec_rng_store(f + SF_RNG * LM_CAP, LM_CAP, rng);
                            to_out = true;                                      // on to the shading
                        }
                    }
                }
                const uint64_t mo = __ballot(to_out);
                if (to_out) { out[n_out + lane_rank(mo)] = slot; st = IDLE; }
                n_out += (uint32_t)__popcll(mo);
            }
        }
    }
}

// kernel_pathtracer up to the primary walk (pathtracer.cu:205-218), as in trace_path_lm: generator, camera ray, nearest light, box, whole-ray
// test (shared by the pixel's frames when the wave is full).  run = the ray has something occupied ahead; otherwise L is final.
struct LmGen { bool run; v3 L, orig, dir; Rng rng; float ls_t, t0, tMax; int ls_id; };
template <bool COUNT, typename LDS>
SVR_DEV LmGen lm_gen(const DevScene& s, const LDS& lds, GroupMapShared* gm, bool live, bool group_march, uint32_t P2, uint32_t x, uint32_t y, uint32_t hashed, Cnt& c,
                     const DevScene* scp = nullptr)
{
    const DevScene& sc = scp ? *scp : s;
    const float INF = u2f(SVR_INF_BITS);
    LmGen g;
    g.run = false; g.L = V3(0.f, 0.f, 0.f); g.orig = g.L; g.dir = V3(0.f, 0.f, 1.f);
    g.rng = Rng{0u, 0u, 0u, 0u, 0u, 0u};
    g.ls_t = 0.f; g.t0 = 0.f; g.tMax = 0.f; g.ls_id = -1;
    if (live) {
        rng_init(g.rng, hashed + (y * sc.imageW + x));
        if (COUNT) c.paths++;
        camera_ray(sc, x, y, g.rng, g.orig, g.dir);
        g.ls_id = nearest_light(sc, g.orig, g.dir, g.ls_t);
        float tMin = (float)1e-6;
        g.tMax = SVR_FLT_MAX;
        if (group_march) {
            float t_occ;
            GroupMap map;
            map.g = gm + ((threadIdx.x & 63u) & ((1u << P2) - 1u) & (GROUP_MAPS_PER_WAVE - 1u));
            const int rr = walk_setup_group<true, true>(s, lds, P2, g.orig, g.dir, false, tMin, g.tMax, t_occ, map);
            g.run = rr > 0 && t_occ != INF;
            g.t0 = fmax_(t_occ, tMin);                                   // (the shared test starts at the group's earliest box entry)
        } else {
            float tNear, tFar;
            if (volume_intersect(s, g.orig, g.dir, tNear, tFar)) {
                tMin = tNear < 0.f ? (float)1e-6 : tNear;
                g.tMax = tFar;
                g.t0 = first_occupied(s, lds, g.orig, g.dir, tMin, g.tMax);      // (per-lane whole-ray test: the same early end as under the shared one)
                g.run = g.t0 != INF;
            }
        }
        if (!g.run) {
            if (g.ls_id >= 0) {                                           // t = FLT_MAX > ls.t: the light is seen (pathtracer.cu:220-229)
                const DevLight& l = sc.lights[g.ls_id];
                const float cosTerm = dot(V3(l.normal[0], l.normal[1], l.normal[2]), -g.dir);
                g.L = V3(l.radiance[0], l.radiance[1], l.radiance[2]) * (cosTerm <= 0.f ? 0.f : 1.f);
            } else if (sc.env_on_escape) g.L = env_radiance(sc, g.dir);
        }
    }
    return g;
}

template <int LAYOUT, bool COUNT>
__global__ __launch_bounds__(LM_THREADS, SVR_LM_WAVES_PER_EU) void k_trace_lm_pool_deep(const DevScene s, const DevWork w)
{
    __shared__ LdsTileCull lds;
    __shared__ GroupMapShared gmaps[TILE_WAVES][GROUP_MAPS_PER_WAVE];
    __shared__ uint32_t pend_task[TILE_WAVES][QUEUE_TASKS];
    __shared__ DevLight lds_lights[8];                                   // (as in k_trace_lm_pool)
    if (threadIdx.x < 8u * (sizeof(DevLight) / 4u)) reinterpret_cast<float*>(lds_lights)[threadIdx.x] = reinterpret_cast<const float*>(s.lights)[threadIdx.x];
    const DevLight* const lts = lds_lights;
    auto cold_scene = [&]() -> const DevScene* {
#if SVR_LM_DEEP_COLD
        auto p = __builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(p));
        return (const DevScene*)p;
#else
        return nullptr;
#endif
    };
    lds_tile_load(lds, s, true);

    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const TaskShape ts = task_shape(w);
    const uint32_t fl2 = ts.fl2, P2 = ts.P2, tw2 = ts.tw2, th2 = ts.th2, wv = ts.wv;
    const uint32_t n_tasks = ts.tiles_x * ts.tiles_y * ts.fgroups;
    const uint32_t shard0 = blockIdx.x % TICKET_SHARDS;
    const uint32_t depth = w.traceDepth;
    const size_t wslot = (size_t)(blockIdx.x * TILE_WAVES + wave);
    float* const gpend = w.pend + wslot * (QUEUE_TASKS * 3u * 64u);
    uint32_t* const F = w.queue + wslot * (REC_WORDS * QUEUE_CAP);         // slot fields
    uint32_t* const LW = F + (size_t)SF_WORDS * LM_CAP;                    // lists
    uint32_t* const LH = LW + LM_CAP;
    uint32_t* const LS = LH + LM_CAP;
    uint32_t* const LA = LS + LM_CAP;
    Cnt c = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    auto fence = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");            // slots, lists and radiance rows are read back by other lanes of this wave
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    };
    uint32_t si = 0u;
    for (;;) {
        // ---- gen: up to LM_BATCH tasks ----
        uint32_t nb = 0u, nW = 0u;
        while (nb < LM_BATCH && si < TICKET_SHARDS) {
            const uint32_t shard = (shard0 + si) % TICKET_SHARDS;
            uint32_t* ticket = w.ticket + shard * TICKET_STRIDE;
            if (si != 0u && __atomic_load_n(ticket, __ATOMIC_RELAXED) * TICKET_SHARDS + shard >= n_tasks) { ++si; continue; }
            uint32_t u = 0;
            if (lane == 0) u = atomicAdd(ticket, 1u);
            u = __builtin_amdgcn_readfirstlane(u);
            const uint32_t k = u * TICKET_SHARDS + shard;
            if (k >= n_tasks) { ++si; continue; }
            uint32_t tx, ty, fg;
            task_decode(ts, k, tx, ty, fg);
            if (COUNT) c.loops += (lane == 0);
            const uint32_t pl = lane & ((1u << P2) - 1u);
            const uint32_t fslot = (fg << fl2) + (lane >> P2);
            const uint32_t px = (tx << tw2) + (pl & ((1u << tw2) - 1u));
            const uint32_t r = (ty << th2) + (pl >> tw2);
            const bool live = px < wv && r < w.n_rows && fslot < w.nframes;
            const bool group_march = fl2 >= 3u && __ballot(live) == ~0ull;
            const LmGen g = lm_gen<COUNT>(s, lds, &gmaps[wave][0], live, group_march, P2, w.x0 + px, live ? owned_row_to_y(w, r) : 0u, wang_hash(w.frame0 + fslot), c, cold_scene());
            {
                float* p = gpend + (size_t)nb * (3u * 64u) + lane;       // (queued paths overwrite theirs when they end)
                p[0] = g.L.x; p[64] = g.L.y; p[128] = g.L.z;
            }
            const uint32_t id = (nb << 6) | lane;
            const uint64_t mq = __ballot(g.run);
            if (g.run) {
                uint32_t* f = F + id;
                rec_v3_store(f + SF_O * LM_CAP, LM_CAP, g.orig); rec_v3_store(f + SF_D * LM_CAP, LM_CAP, g.dir);
                rec_rng_store(f + SF_RNG * LM_CAP, LM_CAP, g.rng);
                f[SF_META * LM_CAP] = (uint32_t)(g.ls_id + 1);
                rec_v3_store(f + SF_L * LM_CAP, LM_CAP, V3(0.f, 0.f, 0.f)); rec_v3_store(f + SF_T * LM_CAP, LM_CAP, V3(1.f, 1.f, 1.f));
                f[SF_LST * LM_CAP] = f2u(g.ls_t); f[SF_T0 * LM_CAP] = f2u(g.t0); f[SF_TMAX * LM_CAP] = f2u(g.tMax);
                LW[nW + lane_rank(mq)] = id;
            }
            nW += (uint32_t)__popcll(mq);
            if (lane == 0) pend_task[wave][nb] = k;
            ++nb;
        }
        if (nb == 0u) break;
        for (uint32_t k = 0; k < depth && nW != 0u; ++k) {
            const bool last = k + 1u >= depth;
            // ---- walk: this bounce's rays -> hits ----
            uint32_t nH = 0u;
            fence();
            lm_walk_pool_deep<LAYOUT, COUNT>(s, lds, F, LW, nW, k == 0u ? LMK_PRIMARY : LMK_CONT, last, LH, nH, gpend, c, cold_scene(), lts);
            fence();
            // ---- shade the hits, 64 at a time (VolumeSample + light sampling, pathtracer.cu:237-257) ----
            uint32_t nS = 0u, nA = 0u;
            for (uint32_t i0 = 0u; i0 < nH; i0 += 64u) {
                const uint32_t i = i0 + lane;
                bool to_s = false, to_a = false;
                uint32_t id = 0u;
                if (i < nH) {
                    id = LH[i];
                    uint32_t* f = F + id;
                    Shade vs;
                    vs.pt = rec_v3_load(f + SF_O * LM_CAP, LM_CAP); vs.wo = rec_v3_load(f + SF_WO * LM_CAP, LM_CAP);
                    const float val = u2f(f[SF_VAL * LM_CAP]);
                    Rng rng;
                    rec_rng_load(f + SF_RNG * LM_CAP, LM_CAP, rng);
                    Nee ne;
                    shade_event<LAYOUT, COUNT, SVR_LM_DEEP_COLD != 0>(s, vs, val, rng, ne, c, cold_scene(), lts);
                    rec_rng_store(f + SF_RNG * LM_CAP, LM_CAP, rng);
                    if (!last) {                                            // what sample_bsdf needs of the event
                        rec_v3_store(f + SF_GRAD * LM_CAP, LM_CAP, vs.gradient);
                        f[SF_COLOR * LM_CAP] = f2u(vs.color[0]); f[(SF_COLOR + 1) * LM_CAP] = f2u(vs.color[1]); f[(SF_COLOR + 2) * LM_CAP] = f2u(vs.color[2]);
                        f[SF_PBRDF * LM_CAP] = f2u(vs.Pbrdf);
                    }
                    f[SF_META * LM_CAP] = (ne.light << 4) | (vs.st ? 0x100u : 0u);
                    if (ne.have) {
                        rec_v3_store(f + SF_D * LM_CAP, LM_CAP, ne.wi);
                        rec_v3_store(f + SF_B * LM_CAP, LM_CAP, ne.B);
                        f[SF_PDF * LM_CAP] = f2u(ne.pdf);
                        to_s = true;
                    } else if (last) {                                      // no light sample reaches the event and nothing follows: L is final
                        const v3 L = rec_v3_load(f + SF_L * LM_CAP, LM_CAP);
                        float* p = gpend + (id >> 6) * (3u * 64u) + (id & 63u);
                        p[0] = L.x; p[64] = L.y; p[128] = L.z;
                    } else to_a = true;
                }
                const uint64_t ms = __ballot(to_s), ma = __ballot(to_a);
                if (to_s) LS[nS + lane_rank(ms)] = id;
                if (to_a) LA[nA + lane_rank(ma)] = id;
                nS += (uint32_t)__popcll(ms); nA += (uint32_t)__popcll(ma);
            }
            // ---- walk: the shadow rays -> A (or final, at the last bounce) ----
            fence();
            lm_walk_pool_deep<LAYOUT, COUNT>(s, lds, F, LS, nS, LMK_SHADOW, last, LA, nA, gpend, c, cold_scene(), lts);
            fence();
            nW = 0u;
            if (last) break;
            // ---- sample_bsdf, throughput, roulette (pathtracer.cu:258-276), 64 at a time -> the next bounce's rays ----
            for (uint32_t i0 = 0u; i0 < nA; i0 += 64u) {
                const uint32_t i = i0 + lane;
                bool to_w = false;
                uint32_t id = 0u;
                if (i < nA) {
                    id = LA[i];
                    uint32_t* f = F + id;
                    Shade vs;
                    vs.pt = rec_v3_load(f + SF_O * LM_CAP, LM_CAP); vs.wo = rec_v3_load(f + SF_WO * LM_CAP, LM_CAP);
                    vs.gradient = rec_v3_load(f + SF_GRAD * LM_CAP, LM_CAP);
                    vs.color[0] = u2f(f[SF_COLOR * LM_CAP]); vs.color[1] = u2f(f[(SF_COLOR + 1) * LM_CAP]); vs.color[2] = u2f(f[(SF_COLOR + 2) * LM_CAP]); vs.color[3] = 0.f;
                    vs.Pbrdf = u2f(f[SF_PBRDF * LM_CAP]);
                    vs.st = (f[SF_META * LM_CAP] & 0x100u) ? 1 : 0;
                    v3 T = rec_v3_load(f + SF_T * LM_CAP, LM_CAP);
                    Rng rng;
                    rec_rng_load(f + SF_RNG * LM_CAP, LM_CAP, rng);
                    v3 wi; float pdf = 0.f;
                    const v3 fr = bsdf_sample(vs, wi, pdf, rng);
                    const float cosTerm = __builtin_fabsf(dot(normalize(vs.gradient), wi));
                    if (fmax_(fr.x, fmax_(fr.y, fr.z)) > 0.f && pdf > 0.f) {
                        if (vs.st == 0) T = T * (fr / (pdf * (1.f - vs.Pbrdf)));
                        else T = T * ((fr * cosTerm) / (pdf * vs.Pbrdf));
                    }
                    if (k >= 3u && russian_roulette(T, rng)) {              // the path ends: L is final
                        const v3 L = rec_v3_load(f + SF_L * LM_CAP, LM_CAP);
                        float* p = gpend + (id >> 6) * (3u * 64u) + (id & 63u);
                        p[0] = L.x; p[64] = L.y; p[128] = L.z;
                    } else {
                        rec_v3_store(f + SF_D * LM_CAP, LM_CAP, wi);        // (the next ray starts at the event: SF_O already holds it)
                        rec_v3_store(f + SF_T * LM_CAP, LM_CAP, T);
                        rec_rng_store(f + SF_RNG * LM_CAP, LM_CAP, rng);
                        to_w = true;
                    }
                }
                const uint64_t mw = __ballot(to_w);
                if (to_w) LW[nW + lane_rank(mw)] = id;
                nW += (uint32_t)__popcll(mw);
            }
        }
        // ---- fold the batch ----
        fence();
        fold_pending(s, w, gpend, 64u, &pend_task[wave][0], nb);
        fence();
    }
    if (COUNT) cnt_flush(w, c);
}

// Persistent 1024-thread blocks, one task = (64 >> f) pixels x (1 << f) frames per wave from 8 sharded tickets, centre-out
// order, the frames of a launch folded into the accumulator in the kernel (svr_tile_tasks.hpp) -- the work distribution of
// k_trace_tile (svr_trace_tile.hip has the measurements behind it).  The radiance of up to QUEUE_TASKS tasks waits in the
// wave's rows in global memory (DevWork.pend), so that a fold keeps 63 or 64 lanes busy.
template <int LAYOUT, bool COUNT, bool DEPTH1>
__global__ __launch_bounds__(LM_THREADS, SVR_LM_WAVES_PER_EU) void k_trace_lm(const DevScene s, const DevWork w)
{
    __shared__ LdsTileCull lds;
    __shared__ GroupMapShared gmaps[TILE_WAVES][GROUP_MAPS_PER_WAVE];
    __shared__ uint32_t pend_task[TILE_WAVES][QUEUE_TASKS];
    lds_tile_load(lds, s, true);

    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const TaskShape ts = task_shape(w);
    const uint32_t fl2 = ts.fl2, P2 = ts.P2, tw2 = ts.tw2, th2 = ts.th2, wv = ts.wv;
    const uint32_t n_tasks = ts.tiles_x * ts.tiles_y * ts.fgroups;
    const uint32_t shard0 = blockIdx.x % TICKET_SHARDS;
    const bool fold = w.fold != 0u;                                   // the host guarantees one frame group then
    const uint32_t pend_max = P2 == 0u ? 21u : QUEUE_TASKS;           // 21 tasks x 3 channels = 63 fold lanes when a wave is one pixel
    float* const gpend = fold ? w.pend + (size_t)(blockIdx.x * TILE_WAVES + wave) * (QUEUE_TASKS * 3u * 64u) : nullptr;
    Cnt c = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t npend = 0;
    auto flush = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");        // the rows are read back by other lanes of this wave
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        fold_pending(s, w, gpend, 64u, &pend_task[wave][0], npend);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        npend = 0;
    };
    for (uint32_t si = 0; si < TICKET_SHARDS; ++si) {
        const uint32_t shard = (shard0 + si) % TICKET_SHARDS;
        uint32_t* ticket = w.ticket + shard * TICKET_STRIDE;
        for (;;) {
            // away from the home counter, look before taking (a plain load of a drained counter is free)
            if (si != 0u && __atomic_load_n(ticket, __ATOMIC_RELAXED) * TICKET_SHARDS + shard >= n_tasks) break;
            uint32_t u = 0;
            if (lane == 0) u = atomicAdd(ticket, 1u);
            u = __builtin_amdgcn_readfirstlane(u);
            const uint32_t k = u * TICKET_SHARDS + shard;
            if (k >= n_tasks) break;
            uint32_t tx, ty, fg;
            task_decode(ts, k, tx, ty, fg);
            if (COUNT) c.loops += (lane == 0);
            const uint32_t pl = lane & ((1u << P2) - 1u);
            const uint32_t slot = (fg << fl2) + (lane >> P2);
            const uint32_t px = (tx << tw2) + (pl & ((1u << tw2) - 1u));
            const uint32_t r = (ty << th2) + (pl >> tw2);
            const bool live = px < wv && r < w.n_rows && slot < w.nframes;
            const bool group_march = fl2 >= 3u && __ballot(live) == ~0ull;
            v3 L = V3(0.f, 0.f, 0.f);
            if (live) {
                const uint32_t x = w.x0 + px, y = owned_row_to_y(w, r);
                L = trace_path_lm<LAYOUT, COUNT, DEPTH1>(s, lds, x, y, w.traceDepth, wang_hash(w.frame0 + slot), group_march, P2, &gmaps[wave][0], c);
                if (!fold) {
                    float* o = w.lbuf + (size_t)slot * w.slot_stride + 3 * ((size_t)y * s.imageW + x);
                    o[0] = L.x; o[1] = L.y; o[2] = L.z;
                }
            }
            if (fold) {
                float* o = gpend + (size_t)npend * (3u * 64u) + lane;
                o[0] = L.x; o[64] = L.y; o[128] = L.z;
                if (lane == 0) pend_task[wave][npend] = k;
                if (++npend == pend_max) flush();
            }
        }
    }
    if (npend) flush();
    if (COUNT) cnt_flush(w, c);
}

template <int LAYOUT, bool COUNT>
static hipError_t launch_lm_t(const DevScene& s, const DevWork& w, const LaunchCfg& cfg, hipStream_t st)
{
    const uint32_t wv = w.x1 - w.x0;
    if (wv == 0 || w.n_rows == 0) return hipSuccess;
    // frames per wave: as many as the launch holds (<= 64); a folding launch keeps every frame of a pixel in one wave
    uint32_t fl2 = 0;
    while (fl2 < 6u && (2u << fl2) <= w.nframes) ++fl2;
    if (cfg.frames_log2 >= 0 && (uint32_t)cfg.frames_log2 < fl2) fl2 = (uint32_t)cfg.frames_log2;
    if (w.fold) {
        if (w.nframes > 64u || w.pend == nullptr) return hipErrorInvalidValue;
        fl2 = 0;
        while ((1u << fl2) < w.nframes) ++fl2;
    }
    const uint32_t P2 = 6u - fl2, tw2 = (P2 + 1u) >> 1, th2 = P2 >> 1;
    const uint32_t fgroups = (w.nframes + (1u << fl2) - 1u) >> fl2;
    const uint32_t n_tasks = ((wv + (1u << tw2) - 1u) >> tw2) * ((w.n_rows + (1u << th2) - 1u) >> th2) * fgroups;
    constexpr uint32_t WPB = LM_THREADS / 64;
    const uint32_t max_blocks = (uint32_t)(cfg.num_cus * cfg.blocks_per_cu) * 4u / WPB;
    uint32_t blocks = (n_tasks + WPB - 1u) / WPB;
    if (blocks > max_blocks) blocks = max_blocks;
    if (w.fold && blocks > w.queue_blocks) return hipErrorInvalidValue;       // the rows are sized for queue_blocks blocks
    if (blocks == 0) blocks = 1;
    DevWork w2 = w;
    w2.unit = 1u;
    w2.frames_log2 = fl2;
    hipError_t e = hipMemsetAsync(w.ticket, 0, sizeof(uint32_t) * TICKET_SHARDS * TICKET_STRIDE, st);
    if (e != hipSuccess) return e;
    // traceDepth 1 (the reference's default), folding launch, queue memory at hand: the pool form
    if (w.traceDepth == 1u && w.fold && w.queue != nullptr && !cfg.lm_straight) {
        // tasks per batch (DevScene.lm_tune bits 24-31): two builds -- 10 (scenes with transparent space) and 21 (fog: 63 of 64 lanes in the fold)
        if (((s.lm_tune >> 24) & 0xffu) >= 16u) hipLaunchKernelGGL((k_trace_lm_pool<LAYOUT, COUNT, 21>), dim3(blocks), dim3(LM_THREADS), 0, st, s, w2);
        else hipLaunchKernelGGL((k_trace_lm_pool<LAYOUT, COUNT, 10>), dim3(blocks), dim3(LM_THREADS), 0, st, s, w2);
    }
    else if (w.traceDepth > 1u && w.fold && w.queue != nullptr && !cfg.lm_straight) hipLaunchKernelGGL((k_trace_lm_pool_deep<LAYOUT, COUNT>), dim3(blocks), dim3(LM_THREADS), 0, st, s, w2);
    else if (w.traceDepth == 1u) hipLaunchKernelGGL((k_trace_lm<LAYOUT, COUNT, true>), dim3(blocks), dim3(LM_THREADS), 0, st, s, w2);
    else hipLaunchKernelGGL((k_trace_lm<LAYOUT, COUNT, false>), dim3(blocks), dim3(LM_THREADS), 0, st, s, w2);
    return hipGetLastError();
}

// needs the acceleration data (s.empty_mask with the class table) and whole-ray validity (s.ray_skip: the clipped box lies
// inside the texture domain, so every point of a walk maps into the macro grid); the caller checks both
hipError_t launch_trace_lm(const DevScene& s, const DevWork& w, const LaunchCfg& cfg, hipStream_t st)
{
    if (s.empty_mask == nullptr || !s.ray_skip) return hipErrorInvalidValue;
    if (s.layout == LAYOUT_CELL) return cfg.count ? launch_lm_t<LAYOUT_CELL, true>(s, w, cfg, st) : launch_lm_t<LAYOUT_CELL, false>(s, w, cfg, st);
    if (s.layout == LAYOUT_PAIR) return cfg.count ? launch_lm_t<LAYOUT_PAIR, true>(s, w, cfg, st) : launch_lm_t<LAYOUT_PAIR, false>(s, w, cfg, st);
    if (s.layout == LAYOUT_BRICK) return cfg.count ? launch_lm_t<LAYOUT_BRICK, true>(s, w, cfg, st) : launch_lm_t<LAYOUT_BRICK, false>(s, w, cfg, st);
    return cfg.count ? launch_lm_t<LAYOUT_LINEAR, true>(s, w, cfg, st) : launch_lm_t<LAYOUT_LINEAR, false>(s, w, cfg, st);
}

} // namespace svr
