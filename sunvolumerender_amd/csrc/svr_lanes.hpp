// svr_lanes.hpp -- what follows the FIRST scatter event of a path: records in the wave's queue memory and a per-lane state
// machine that drains them (svr_trace_tile.hip, QUEUE builds).
//
// Why.  The primary walks of a wave are coherent (its lanes are frames of the same pixels: one shared whole-ray test,
// similar walk lengths), and so is the shading of their hits.  What follows is not: only some lanes scatter, a shadow
// walk through a medium that cannot be skipped takes ~10^2 iterations while the other lanes of the wave wait, and after
// a bounce every lane goes its own way (straight-line code: 32 % lane utilisation at traceDepth 2, 22 % at depth 4).  So
// the wave shades the hits of a task in place and then hands the paths over:
//
//   traceDepth 1   record C1 = a shaded event's prepared next-event estimate, waiting for its shadow walk.  After
//                  QUEUE_TASKS tasks (2048 paths) the wave drains its records with all 64 lanes:
//                  IDLE -> (pop) -> WALK -> END -> (radiance = estimate x transmittance) -> IDLE
//   deeper         the first shadow walk and its estimate run in place too; record A = a path waiting for its BSDF
//                  sampling.  In the machine a lane holds a path only while it WALKs; a finished walk is settled and the
//                  path waits on the stack of the service it needs -- A (BSDF sampling, throughput, roulette, next walk's
//                  set-up) or B (shading, light sampling, shadow walk's set-up) -- and a round of services takes all
//                  non-walking lanes through one of them.
//
// WALK is the cheap Woodcock iteration of svr_walk.hpp (FREE / EMPTY / CULLED); the fetch an iteration asks for (8
// voxels + filter + LUT) and the re-march after an occupied stretch are served inside the walk loop.
// All of it is scheduling: each path executes the reference's operations (pathtracer.cu:216-277) in the reference's
// order on its own generator, so the radiance is bit-identical to the straight-line code of trace_path_tile.
#pragma once
#include "svr_walk.hpp"
#include "svr_tile_tasks.hpp"

namespace svr {


// Phase profile of experiment builds (-DSVR_TEST_HOOKS): per phase, shader cycles the waves spent in it and the same
// weighted by the lanes that had work there (-> lane utilisation per phase); read with svr_debug_phase_profile.
enum { PH_PRIMARY = 0, PH_REFILL, PH_SHADE, PH_CHEAP, PH_FETCH, PH_MARCH, PH_END, PH_FOLD, PH_N };
constexpr uint32_t PROF_WORDS = 2 * PH_N + 2;     // + cheap-loop iterations, + walking lanes summed over them
#ifdef SVR_TEST_HOOKS
#define SVR_PROF 1
struct ProfScope {
    unsigned long long* acc; uint32_t ph; uint64_t t0;
    __device__ ProfScope(unsigned long long* a, uint32_t p) : acc(a), ph(p), t0(__builtin_amdgcn_s_memtime()) {}
    __device__ void end(uint32_t lanes)
    {
        const uint64_t dt = __builtin_amdgcn_s_memtime() - t0;
        if ((threadIdx.x & 63u) == 0u) { atomicAdd(&acc[2 * ph], (unsigned long long)dt); atomicAdd(&acc[2 * ph + 1], (unsigned long long)dt * lanes); }
    }
};
#define PROF_BEGIN(name, ph) ProfScope name(c_prof, ph)
#define PROF_END(name, lanes) name.end(lanes)
#else
#define SVR_PROF 0
#define PROF_BEGIN(name, ph)
#define PROF_END(name, lanes)
#endif
SVR_DEV uint32_t lane_rank(uint64_t m)  // number of set bits of m below this lane
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// queue memory of one wave: word j of record i of a stack at base[j * cap + i]
struct LaneQueue { uint32_t* q; uint32_t cap; };

// next-event estimate a shaded scatter event has prepared; its shadow walk decides whether the light arrives
struct Nee { v3 wi; v3 B; float pdf; uint32_t light; bool have; };

// VolumeSample + the light sampling of estimate_direct_light (pathtracer.cu:237-257, 171-190) for the scatter event at
// vs.pt / vs.wo with intensity val: fills vs and the estimate; what follows is the shadow walk along ne.wi
// scp (here and in trace_primary / gen_primary / drain_queue): where to read the scene constants that only set-up, shading and settling use -- transfer
// function, spacing, lights, camera, environment.  Null: from `s` like everything else.  The queue builds of the tile kernel pass a pointer into the kernarg
// segment that the compiler cannot see through (svr_trace_tile.hip, cold_scene), so those ~60 constants are scalar-loaded where they are used instead of
// staying in (spilled) scalar registers across the walk loops.
// LDSL / lts: the lights come from the caller's copy in LDS (the tile kernel's queue builds), where a per-lane index is a ds_read.  A compile-time
// switch: a run-time choice between an LDS and a kernarg pointer makes every access a flat load (measured: - 10 %)
template <int LAYOUT, bool COUNT, bool LDSL = false>
SVR_DEV void shade_event(const DevScene& s, Shade& vs, float val, Rng& rng, Nee& ne, Cnt& c, const DevScene* scp = nullptr, const DevLight* lts = nullptr)
{
    const DevScene& sc = scp ? *scp : s;
    const DevLight* const LT = LDSL ? lts : sc.lights;
    if (COUNT) { c.scatter++; c.taps += 7; c.exec += 6; }
    tf_rgba(sc, sc.tf, val, vs.color);
    {
        // Gradient_CentralDiff, cuda_volume.h:54-61
        const v3 q = vs.pt;
        float xd = intensity_at<LAYOUT>(s, V3(q.x + sc.spacing[0], q.y + 0.f, q.z + 0.f)) -
                   intensity_at<LAYOUT>(s, V3(q.x - sc.spacing[0], q.y - 0.f, q.z - 0.f));
        float yd = intensity_at<LAYOUT>(s, V3(q.x + 0.f, q.y + sc.spacing[1], q.z + 0.f)) -
                   intensity_at<LAYOUT>(s, V3(q.x - 0.f, q.y - sc.spacing[1], q.z - 0.f));
        float zd = intensity_at<LAYOUT>(s, V3(q.x + 0.f, q.y + 0.f, q.z + sc.spacing[2])) -
                   intensity_at<LAYOUT>(s, V3(q.x - 0.f, q.y - 0.f, q.z - sc.spacing[2]));
        vs.gradient = V3((xd * 0.5f) * sc.invSpacing[0], (yd * 0.5f) * sc.invSpacing[1], (zd * 0.5f) * sc.invSpacing[2]);
    }
    const float gradMag = __builtin_sqrtf(dot(vs.gradient, vs.gradient));
    vs.Pbrdf = vs.color[3] * (1.f - expf_(sc.pbrdf_c * gradMag * 65535.f * sc.invMaxMagnitude));
    vs.st = (rng_uniform(rng) < vs.Pbrdf) ? 1 : 0;
    ne.have = false;
    if (sc.num_lights != 0) {
        int li = (int)((float)sc.num_lights * rng_uniform(rng));
        li = li < (int)sc.num_lights ? li : (int)sc.num_lights - 1;
        v3 Li;
        if (sample_light(LT[li], vs.pt, rng, ne.wi, ne.pdf, Li)) {
            ne.have = true;
            ne.light = (uint32_t)li;
            ne.B = bsdf_eval(vs, ne.wi);
            if (COUNT) c.shadow++;
        }
    }
}

// ---- records.  meta = id (12 bits: task << 6 | lane) | bounce k (15 bits) | light (4 bits) | shading type (1 bit)
SVR_DEV uint32_t rec_meta(uint32_t id, uint32_t k, uint32_t light, int st) { return id | (k << 12) | (light << 27) | (st ? 0x80000000u : 0u); }
SVR_DEV uint32_t meta_id(uint32_t m) { return m & 0xfffu; }
SVR_DEV uint32_t meta_k(uint32_t m) { return (m >> 12) & 0x7fffu; }
SVR_DEV uint32_t meta_light(uint32_t m) { return (m >> 27) & 0xfu; }
SVR_DEV void rec_rng_store(uint32_t* p, uint32_t cap, const Rng& rng)
{
    p[0] = rng.v0; p[cap] = rng.v1; p[2 * cap] = rng.v2; p[3 * cap] = rng.v3; p[4 * cap] = rng.v4; p[5 * cap] = rng.d;
}
SVR_DEV void rec_rng_load(const uint32_t* p, uint32_t cap, Rng& rng)
{
    rng.v0 = p[0]; rng.v1 = p[cap]; rng.v2 = p[2 * cap]; rng.v3 = p[3 * cap]; rng.v4 = p[4 * cap]; rng.d = p[5 * cap];
}
SVR_DEV void rec_v3_store(uint32_t* p, uint32_t cap, v3 a) { p[0] = f2u(a.x); p[cap] = f2u(a.y); p[2 * cap] = f2u(a.z); }
SVR_DEV v3 rec_v3_load(const uint32_t* p, uint32_t cap) { return V3(u2f(p[0]), u2f(p[cap]), u2f(p[2 * cap])); }
// the 13 words of a shaded scatter event: pt wo gradient colour Pbrdf
SVR_DEV void rec_shade_store(uint32_t* p, uint32_t cap, const Shade& vs)
{
    rec_v3_store(p, cap, vs.pt); rec_v3_store(p + 3 * cap, cap, vs.wo); rec_v3_store(p + 6 * cap, cap, vs.gradient);
    p[9 * cap] = f2u(vs.color[0]); p[10 * cap] = f2u(vs.color[1]); p[11 * cap] = f2u(vs.color[2]);
    p[12 * cap] = f2u(vs.Pbrdf);
}
SVR_DEV void rec_shade_load(const uint32_t* p, uint32_t cap, Shade& vs)
{
    vs.pt = rec_v3_load(p, cap); vs.wo = rec_v3_load(p + 3 * cap, cap); vs.gradient = rec_v3_load(p + 6 * cap, cap);
    vs.color[0] = u2f(p[9 * cap]); vs.color[1] = u2f(p[10 * cap]); vs.color[2] = u2f(p[11 * cap]);
    vs.Pbrdf = u2f(p[12 * cap]);
}
// A: a path after a next-event estimate; continues at the BSDF sampling (pathtracer.cu:259)
SVR_DEV void rec_a_store(uint32_t* p, uint32_t cap, const Shade& vs, v3 L, v3 T, const Rng& rng, uint32_t meta)
{
    rec_shade_store(p, cap, vs);
    rec_v3_store(p + 13 * cap, cap, L); rec_v3_store(p + 16 * cap, cap, T);
    rec_rng_store(p + 19 * cap, cap, rng);
    p[25 * cap] = meta;
}
SVR_DEV uint32_t rec_a_load(const uint32_t* p, uint32_t cap, Shade& vs, v3& L, v3& T, Rng& rng)
{
    rec_shade_load(p, cap, vs);
    L = rec_v3_load(p + 13 * cap, cap); T = rec_v3_load(p + 16 * cap, cap);
    rec_rng_load(p + 19 * cap, cap, rng);
    const uint32_t meta = p[25 * cap];
    vs.st = (int)(meta >> 31);
    return meta;
}
// B: a path whose continuation walk found a collision; continues at the shading (pathtracer.cu:237)
SVR_DEV void rec_b_store(uint32_t* p, uint32_t cap, v3 pt, v3 wo, float val, v3 L, v3 T, const Rng& rng, uint32_t meta)
{
    rec_v3_store(p, cap, pt); rec_v3_store(p + 3 * cap, cap, wo);
    p[6 * cap] = f2u(val);
    rec_v3_store(p + 7 * cap, cap, L); rec_v3_store(p + 10 * cap, cap, T);
    rec_rng_store(p + 13 * cap, cap, rng);
    p[19 * cap] = meta;
}
SVR_DEV uint32_t rec_b_load(const uint32_t* p, uint32_t cap, v3& pt, v3& wo, float& val, v3& L, v3& T, Rng& rng)
{
    pt = rec_v3_load(p, cap); wo = rec_v3_load(p + 3 * cap, cap);
    val = u2f(p[6 * cap]);
    L = rec_v3_load(p + 7 * cap, cap); T = rec_v3_load(p + 10 * cap, cap);
    rec_rng_load(p + 13 * cap, cap, rng);
    return p[19 * cap];
}
// C1 (traceDepth 1): the first scatter event of a path, shaded, waiting for its shadow walk -- nothing follows that walk but the estimate itself
SVR_DEV void rec_c1_store(uint32_t* p, uint32_t cap, v3 pt, const Nee& ne, const Rng& rng, uint32_t id)
{
    rec_v3_store(p, cap, pt); rec_v3_store(p + 3 * cap, cap, ne.wi); rec_v3_store(p + 6 * cap, cap, ne.B);
    p[9 * cap] = f2u(ne.pdf);
    rec_rng_store(p + 10 * cap, cap, rng);
    p[16 * cap] = rec_meta(id, 0u, ne.light, 0);
}
SVR_DEV uint32_t rec_c1_load(const uint32_t* p, uint32_t cap, v3& pt, Nee& ne, Rng& rng)
{
    pt = rec_v3_load(p, cap); ne.wi = rec_v3_load(p + 3 * cap, cap); ne.B = rec_v3_load(p + 6 * cap, cap);
    ne.pdf = u2f(p[9 * cap]);
    rec_rng_load(p + 10 * cap, cap, rng);
    const uint32_t meta = p[16 * cap];
    ne.light = meta_light(meta);
    return meta;
}
// P (traceDepth 1, POOL builds): a camera ray that has something possibly occupied ahead, waiting for its primary walk -- in the words
// of a C1 record: origin, direction, (nearest light's t, first possibly-occupied parameter, tMin), tMax, generator, id | nearest light + 1
SVR_DEV void rec_p_store(uint32_t* p, uint32_t cap, v3 o, v3 d, float ls_t, float t_occ, float tMin, float tMax, const Rng& rng, uint32_t id, uint32_t ls1)
{
    rec_v3_store(p, cap, o); rec_v3_store(p + 3 * cap, cap, d);
    p[6 * cap] = f2u(ls_t); p[7 * cap] = f2u(t_occ); p[8 * cap] = f2u(tMin); p[9 * cap] = f2u(tMax);
    rec_rng_store(p + 10 * cap, cap, rng);
    p[16 * cap] = rec_meta(id, 0u, ls1, 0);
}
// H (POOL builds): the collision of a primary walk, waiting to be shaded: pt(3) wo(3) val rng(6) id
constexpr uint32_t REC_H_WORDS = 14;
SVR_DEV uint32_t* queue_h(const LaneQueue& Q) { return Q.q + (size_t)REC_C1_WORDS * Q.cap; }
static_assert(REC_C1_WORDS + REC_H_WORDS <= REC_WORDS, "P / C1 records and H records share a wave's queue slice");

// the wave's queue memory: traceDepth 1: C1 records; deeper: stacks A | B
SVR_DEV uint32_t* queue_c(const LaneQueue& Q) { return Q.q; }
SVR_DEV uint32_t* queue_a(const LaneQueue& Q) { return Q.q; }
SVR_DEV uint32_t* queue_b(const LaneQueue& Q) { return Q.q + (size_t)REC_A_WORDS * Q.cap; }

// The wave hands its first scatter events to the queue: ballot + mbcnt prefix sum, so the records are dense and the stores
// coalesce.  traceDepth 1: shaded events that have a light sample (the others are over with L = 0: the caller stores it).
SVR_DEV void queue_push_c1(const LaneQueue& Q, uint32_t& nC, bool live, v3 pt, const Nee& ne, const Rng& rng, uint32_t id)
{
    const uint64_t m = __ballot(live);
    if (live) rec_c1_store(queue_c(Q) + nC + lane_rank(m), Q.cap, pt, ne, rng, id);
    nC += (uint32_t)__popcll(m);
}
// deeper, media without exactly transparent space (every walk long, the hits of a task far apart in time): paths at their first
// scatter event, for the machine to shade
SVR_DEV void queue_push_b(const LaneQueue& Q, uint32_t& nB, bool live, v3 pt, v3 wo, float val, const Rng& rng, uint32_t id)
{
    const uint64_t m = __ballot(live);
    if (live) rec_b_store(queue_b(Q) + nB + lane_rank(m), Q.cap, pt, wo, val, V3(0.f, 0.f, 0.f), V3(1.f, 1.f, 1.f), rng, rec_meta(id, 0u, 0u, 0));
    nB += (uint32_t)__popcll(m);
}
// deeper: paths after the next-event estimate of their first scatter event
SVR_DEV void queue_push_a(const LaneQueue& Q, uint32_t& nA, bool live, const Shade& vs, v3 L, const Rng& rng, uint32_t id)
{
    const uint64_t m = __ballot(live);
    if (live) rec_a_store(queue_a(Q) + nA + lane_rank(m), Q.cap, vs, L, V3(1.f, 1.f, 1.f), rng, rec_meta(id, 0u, 0u, vs.st));
    nA += (uint32_t)__popcll(m);
}

#ifndef SVR_PARK_CHEAP
#define SVR_PARK_CHEAP 16
#endif
#ifndef SVR_FREE_MIN
#define SVR_FREE_MIN 16      // deeper paths: lanes in fetch-free iterations that make a loop of their own worthwhile ...
#endif
#ifndef SVR_FREE_MIN_D1
#define SVR_FREE_MIN_D1 32   // the same at traceDepth 1 (rarely met there: walks without a live generator end behind their last occupied stretch)
#endif
#ifndef SVR_FREE_ROUNDS
#define SVR_FREE_ROUNDS 8    // ... and its rounds per turn
#endif

// Drain the wave's records (traceDepth 1: nC shaded first events; deeper: nA paths waiting for the BSDF sampling, nB0 for the shading) with all 64 lanes.  pendL: the
// wave's pending-radiance rows ([task * 3 + channel] of pend_row floats); a finished path with id = (task << 6 | lane)
// stores its radiance at row (id >> 6) * 3 + channel, column id & 63.
// POOL builds (traceDepth 1): `primary` = the records are P records (camera rays): a lane pops one, walks it, and settles it --
// the nearest light or the environment (the path is over) or a collision, which becomes an H record (nH counts them).
// OUT: where a finished path's radiance goes.  0: its task's row (pendL).  1 (launches that do not fold: frames traced ahead): straight to its scratch
// slot, found from the wave's pending task numbers (wk / tasks; svr_tile_tasks.hpp direct_put).  2 (the split kernels of deeper paths,
// svr_trace_split.hip): straight to its scratch slot, found from the chunk's table tasks[id] = frame slot << 26 | pixel index.
template <int LAYOUT, bool COUNT, bool SKIP, bool DEPTH1, typename LDS, bool POOL = false, bool HIT_B = false, int OUT = 0, bool LDSL = false>
SVR_DEV void drain_queue(const DevScene& s, const LDS& L_, const LaneQueue& Q, uint32_t nC, uint32_t nA, uint32_t nB0, uint32_t traceDepth_, float* pendL, uint32_t pend_row, Cnt& c,
                         unsigned long long* c_prof = nullptr, const bool primary = false, uint32_t* nH = nullptr, const DevWork* wk = nullptr, const uint32_t* tasks = nullptr,
                         const DevScene* scp = nullptr, const DevLight* lts = nullptr)
{
    const DevScene& sc = scp ? *scp : s;
    const DevLight* const LT = LDSL ? lts : sc.lights;
    enum : uint32_t { IDLE = 0u, CELL = 1u, WALK = 2u, FETCH = 3u, MARCH = 4u, END = 5u, WANT_A = 6u, WANT_B = 7u };
    const float INF = u2f(SVR_INF_BITS);
    const uint32_t traceDepth = DEPTH1 ? 1u : traceDepth_;
    const bool fast_bound = lds_has_bnd8<LDS>::value && s.bnd8 != nullptr;      // (SVR_OPT_FAST_BOUND; media without exactly transparent space)
    uint32_t st = IDLE;
    // walk state
    Rng rng = {0u, 0u, 0u, 0u, 0u, 0u};
    v3 orig = V3(0.f, 0.f, 0.f), dir = V3(0.f, 0.f, 1.f);
    float t = 0.f, tMin = 0.f, tMax = 0.f, t_occ = 0.f, xi = 0.f, val = 0.f;
    uint32_t clear_run = 0u, guard = 0u;
    bool shadow = false, rng_live = false, hit = false, tail_counted = false, ray_skippable = false;
    // path state
    uint32_t id = 0u, k = 0u;
    v3 L = V3(0.f, 0.f, 0.f), T = V3(1.f, 1.f, 1.f);
    Nee ne;
    ne.wi = dir; ne.B = L; ne.pdf = 1.f; ne.light = 0u; ne.have = false;
    Shade vs;
    vs.pt = orig; vs.wo = dir; vs.gradient = dir; vs.color[0] = vs.color[1] = vs.color[2] = vs.color[3] = 0.f; vs.Pbrdf = 0.f; vs.st = 0;

    // begin a walk from `orig` along `dir`: WALK, or END with hit = false when its result is known
    auto begin_walk = [&](bool is_shadow, bool live) {
        shadow = is_shadow; rng_live = live; hit = false;
        tMin = (float)1e-6; tMax = SVR_FLT_MAX;
        const int r = walk_setup<COUNT, SKIP>(s, L_, orig, dir, live, tMin, tMax, t_occ);
        t = tMin; clear_run = 0u; guard = 0u; tail_counted = false;
        ray_skippable = SKIP && s.ray_skip && !live && t_occ == INF;
        if (COUNT && r > 0 && ray_skippable) c.wskip++;
        st = r > 0 ? WALK : END;
    };

    // one Woodcock iteration of a walking lane without a fetch (woodcock_tracking.h:32-45): FREE / EMPTY / CULLED keep it
    // in WALK, anything else parks it (FETCH, MARCH) or ends the walk (END)
    auto iterate = [&]() {
        if (st != CELL) {
            if (COUNT) { c.iters++; if (ray_skippable || tail_counted) c.iskip++; }
            t += -logf_unit(1.f - rng_uniform(rng)) * s.invSigmaMaxSI;
            if (t > tMax || guard++ >= SVR_WALK_GUARD) { st = END; return; }
            if (COUNT) c.taps++;
            if (SKIP && t < t_occ) {
                if (COUNT && !(ray_skippable || tail_counted)) c.ipre++;
                rng_skip(rng);                                  // the accept draw of a FREE iteration
                return;
            }
        }
        st = WALK;
        const Cell cell = cell_of(s, orig + dir * t);
        CellInfo ci;
        ci.empty = false; ci.deep = false; ci.thr = INF;
        if (SKIP) ci = cell_info<true>(L_, s, cell);
        if (ci.empty) {
            rng_skip(rng);                                  // sigma_t = 0: the draw is consumed, the test fails
            clear_run = ci.deep ? clear_run + 1u : 0u;
            if (clear_run == 2u) st = MARCH;
        } else {
            clear_run = 0u;
            xi = rng_uniform(rng);
            if (xi < ci.thr) st = FETCH;                   // else CULLED: xi >= bound >= sigma_t * invSigmaMax
            else if (COUNT) c.cull++;
        }
    };
    // TRIPS (POOL builds: media without exactly transparent space, where a walk is tens of iterations long): five iterations of the
    // walking lanes in a row with the generator as a circular buffer (svr_device.hpp: heads 0 2 4 1 3, no register moves, the
    // Weyl word advanced once), the lanes that leave the WALK state on the way (FETCH / MARCH / END) recording after how many
    // generator steps they left -- one barrel rotation behind the trip puts every generator back into the shifting form.  The
    // operations of a lane are those of iterate(), in the same order: scheduling only.  A lane that left waits for the end of
    // the trip (13 % of the iterations of c3n ask for a fetch: 3.85 of 5 iteration slots are used), and the fetch service runs
    // once per trip for half of the lanes instead of once per iteration for an eighth.
    auto iterate_rot = [&](auto hd, auto jj, bool& in, uint32_t& steps, const uint32_t d0, const v3 hd_, const v3 ho_) {
        constexpr int H = decltype(hd)::value;
        constexpr uint32_t J = (uint32_t)decltype(jj)::value;
        if (!in) return;
        if (COUNT) { c.iters++; if (ray_skippable || tail_counted) c.iskip++; }
        t += -logf_unit(1.f - rng_to_uniform(rng_xorshift_rot<H>(rng) + (d0 + (2u * J + 1u) * RNG_WEYL))) * s.invSigmaMaxSI;
        if (t > tMax || guard++ >= SVR_WALK_GUARD) { st = END; in = false; steps = 2u * J + 1u; return; }
        if (COUNT) c.taps++;
        if constexpr (lds_has_bnd8<LDS>::value) {
            if (fast_bound) {
                // The bound of the tap's neighbourhood from the ray parameter (svr_accel.hip, k_bound8): a culled draw is a rejection under ANY valid
                // bound.  (An iteration before the walk's first possibly-occupied cell needs no test of its own: it is culled or fetches a zero opacity,
                // two draws either way; counting builds keep the test for their prefix counter.)
                if (COUNT && SKIP && t < t_occ) {
                    if (!(ray_skippable || tail_counted)) c.ipre++;
                    rng_xorshift_rot<(H + 1) % 5>(rng);
                    return;
                }
                const uint32_t hx = (uint32_t)fma_(t, hd_.x, ho_.x), hy = (uint32_t)fma_(t, hd_.y, ho_.y), hz = (uint32_t)fma_(t, hd_.z, ho_.z);
                // (hz * 34 + hy) * 34 + hx as two 24-bit multiply-adds with the inline constant (the compiler makes the first a 64-bit v_mad_u64_u32)
                static_assert(BOUND8_DIM == 34u, "the inline constant of the index");
                uint32_t zy, e;
                asm("v_mad_u32_u24 %0, %1, 34, %2" : "=v"(zy) : "v"(hz), "v"(hy));
                asm("v_mad_u32_u24 %0, %1, 34, %2" : "=v"(e) : "v"(zy), "v"(hx));
                const uint32_t B = L_.bnd[e];
                const uint32_t x = rng_xorshift_rot<(H + 1) % 5>(rng) + (d0 + (2u * J + 2u) * RNG_WEYL);
                if ((x >> 24) <= B) { xi = rng_to_uniform(x); st = FETCH; in = false; steps = 2u * J + 2u; }
                else if (COUNT) c.cull++;
                return;
            }
        }
        if (SKIP && t < t_occ) {
            if (COUNT && !(ray_skippable || tail_counted)) c.ipre++;
            rng_xorshift_rot<(H + 1) % 5>(rng);
            return;
        }
        const Cell cell = cell_of(s, orig + dir * t);
        CellInfo ci;
        ci.empty = false; ci.deep = false; ci.thr = INF;
        if (SKIP) ci = cell_info<true>(L_, s, cell);
        if (ci.empty) {
            rng_xorshift_rot<(H + 1) % 5>(rng);
            clear_run = ci.deep ? clear_run + 1u : 0u;
            if (clear_run == 2u) { st = MARCH; in = false; steps = 2u * J + 2u; }
        } else {
            clear_run = 0u;
            xi = rng_to_uniform(rng_xorshift_rot<(H + 1) % 5>(rng) + (d0 + (2u * J + 2u) * RNG_WEYL));
            if (xi < ci.thr) { st = FETCH; in = false; steps = 2u * J + 2u; }
            else if (COUNT) c.cull++;
        }
    };
    auto trip = [&]() {
        const bool was = st == WALK;
        bool in = was;
        uint32_t steps = 10u;
        const uint32_t d0 = rng.d;
        // the ray in half-resolution macro-grid coordinates (the fast bound look-up: one fma per axis and iteration)
        v3 hd_ = dir, ho_ = orig;
        if constexpr (lds_has_bnd8<LDS>::value) {
            if (fast_bound) {
                hd_ = V3(dir.x * s.hc_scale[0], dir.y * s.hc_scale[1], dir.z * s.hc_scale[2]);
                ho_ = V3(fma_(orig.x - s.vmin[0], s.hc_scale[0], s.hc_off), fma_(orig.y - s.vmin[1], s.hc_scale[1], s.hc_off), fma_(orig.z - s.vmin[2], s.hc_scale[2], s.hc_off));
            }
        }
        iterate_rot(RngHead<0>{}, RngHead<0>{}, in, steps, d0, hd_, ho_);
        iterate_rot(RngHead<2>{}, RngHead<1>{}, in, steps, d0, hd_, ho_);
        iterate_rot(RngHead<4>{}, RngHead<2>{}, in, steps, d0, hd_, ho_);
        iterate_rot(RngHead<1>{}, RngHead<3>{}, in, steps, d0, hd_, ho_);
        iterate_rot(RngHead<3>{}, RngHead<4>{}, in, steps, d0, hd_, ho_);
        if (was) {
            rng.d = d0 + steps * RNG_WEYL;
            rng_canon(rng, steps >= 10u ? 0u : (steps >= 5u ? steps - 5u : steps));
        }
    };
    // the same iteration for a lane that is before its first possibly-occupied cell (t < t_occ), without the rest: the lane
    // stays such a lane, ends its walk, or stops with the iteration's cell test pending (CELL)
    auto free_iterate = [&]() {
        if (COUNT) { c.iters++; if (ray_skippable || tail_counted) c.iskip++; }
        t += -logf_unit(1.f - rng_uniform(rng)) * s.invSigmaMaxSI;
        if (t > tMax || guard++ >= SVR_WALK_GUARD) { st = END; return; }
        if (COUNT) c.taps++;
        if (t < t_occ) {
            if (COUNT && !(ray_skippable || tail_counted)) c.ipre++;
            rng_skip(rng);
            return;
        }
        st = CELL;
    };

    // the services of a walking lane that cannot wait: the fetch an iteration asked for, the re-march after an occupied stretch
    auto serve_fetch_march = [&]() {
        // FETCH: 8 voxels + filter + LUT, then the accept test with the draw the lane kept
        if (__ballot(st == FETCH) != 0ull) {
            if (st == FETCH) {
                if (COUNT) c.exec++;
                val = tex_fetch<LAYOUT>(s, cell_of(s, orig + dir * t)) * s.densityScale;
                const float sigma_t = alpha_of(L_, s, val);
                if (xi < sigma_t * s.invSigmaMax) { st = END; hit = true; }
                else st = WALK;
            }
        }
        // MARCH: the walk has left an occupied stretch: where is the next one?
        if (SKIP && __ballot(st == MARCH) != 0ull) {
            if (st == MARCH) {
                t_occ = first_occupied(s, L_, orig, dir, t, tMax);
                clear_run = 0u;
                st = WALK;
                if (t_occ == INF && !rng_live) {
                    if (!COUNT) st = END;                               // nothing ahead and no draw follows the walk: it ends without a collision
                    else if (!tail_counted) { tail_counted = true; c.wskip++; }
                }
            }
        }
    };
    // a finished path hands its radiance to its task's row
    auto finish = [&]() {
        if constexpr (OUT == 1) direct_put(s, *wk, tasks, id, L);
        else if constexpr (OUT == 2) {
            const uint32_t gid = tasks[id];
            float* o = wk->lbuf + (size_t)(gid >> 26) * wk->slot_stride + 3 * (size_t)(gid & 0x3ffffffu);
            o[0] = L.x; o[1] = L.y; o[2] = L.z;
        } else {
            float* o = pendL + (id >> 6) * 3u * pend_row + (id & 63u);
            o[0] = L.x; o[pend_row] = L.y; o[2u * pend_row] = L.z;
        }
        st = IDLE;
    };
    // END of a SHADOW walk (pathtracer.cu:191-198): transmittance -> direct light
    auto nee = [&]() {
        // transmittance.h:15-16 on the walk's result ts (t, or -FLT_MAX), with the box interval of the shadow ray
        const float ts = hit ? t : -SVR_FLT_MAX;
        const float Tr = ((ts > tMin) && (ts < tMax)) ? 0.f : 1.f;
        const float kf = Tr * (float)sc.num_lights;
        const DevLight& l = LT[ne.light];
        const v3 Li = V3(l.radiance[0], l.radiance[1], l.radiance[2]);     // sample_light returned true: cosTerm > 0
        L = L + T * (((ne.B * kf) * Li) / ne.pdf);
    };
    // the shadow walk of the prepared estimate; the draws of sample_bsdf / roulette follow it unless this is the last bounce
    auto begin_shadow = [&]() {
        orig = vs.pt;
        dir = ne.wi;
        begin_walk(true, k + 1u < traceDepth);
    };

    const bool trips = s.trips != 0u;                  // (SVR_OPT_TRIPS)
    const uint32_t park_cheap = s.park_cheap;          // (SVR_OPT_PARK_CHEAP; SVR_PARK_CHEAP = 16 is the default)
    if constexpr (DEPTH1) {
        // traceDepth 1: the machine only walks -- pop a shaded event, shadow walk, radiance -- so its rounds are cheap
        // (~10^2 instructions) and run as soon as a few lanes wait
        uint32_t next = 0u;
        for (;;) {
            PROF_BEGIN(pw, PH_CHEAP);
            while (__ballot(st == WALK || st == CELL) != 0ull) {
                if ((uint32_t)__popcll(__ballot(st == END)) + min((uint32_t)__popcll(__ballot(st == IDLE)), nC - next) >= park_cheap) break;
                if (SKIP) {
#pragma nounroll
                    for (uint32_t rounds = 0u; rounds < SVR_FREE_ROUNDS && (uint32_t)__popcll(__ballot(st == WALK && t < t_occ)) >= SVR_FREE_MIN_D1; ++rounds)
                        if (st == WALK && t < t_occ) free_iterate();
                }
                if (POOL && trips) {
                    if (st == CELL) iterate();
                    trip();
                } else if (st == WALK || st == CELL) iterate();
                serve_fetch_march();
            }
            PROF_END(pw, 32u);
            PROF_BEGIN(pe, PH_END);
            if (POOL && primary) {
                // the end of a primary walk (pathtracer.cu:220-235): the nearest light in front of the collision, no collision: the
                // environment -- the path is over -- else a scatter event for the wave to shade (H record)
                bool to_hit = false;
                if (st == END) {
                    const float tt = hit ? t : SVR_FLT_MAX;
                    const uint32_t ls1 = ne.light;                          // nearest light + 1 (0 = none); ne.pdf holds its t
                    if (ls1 != 0u && ne.pdf < tt) {
                        const DevLight& l = LT[ls1 - 1u];
                        const float cosTerm = dot(V3(l.normal[0], l.normal[1], l.normal[2]), -dir);
                        L = L + (T * V3(l.radiance[0], l.radiance[1], l.radiance[2])) * (cosTerm <= 0.f ? 0.f : 1.f);
                        finish();
                    } else if (!hit) {
                        if (sc.env_on_escape) L = L + T * env_radiance(sc, dir);
                        finish();
                    } else to_hit = true;
                }
                const uint64_t mh = __ballot(to_hit);
                if (to_hit) {
                    if constexpr (HIT_B) {
                        // deeper paths: the collision is a B record of the machine that follows (first scatter event, unshaded)
                        rec_b_store(queue_b(Q) + *nH + lane_rank(mh), Q.cap, orig + dir * t, -dir, val, V3(0.f, 0.f, 0.f), V3(1.f, 1.f, 1.f), rng, rec_meta(id, 0u, 0u, 0));
                    } else {
                        uint32_t* h = queue_h(Q) + *nH + lane_rank(mh);
                        rec_v3_store(h, Q.cap, orig + dir * t); rec_v3_store(h + 3 * Q.cap, Q.cap, -dir);
                        h[6 * Q.cap] = f2u(val);
                        rec_rng_store(h + 7 * Q.cap, Q.cap, rng);
                        h[13 * Q.cap] = id;
                    }
                    st = IDLE;
                }
                *nH += (uint32_t)__popcll(mh);
            } else if (st == END) { nee(); finish(); }
            if (next < nC) {
                const uint64_t idle = __ballot(st == IDLE);
                const uint32_t i = next + lane_rank(idle);
                if (st == IDLE && i < nC) {
                    if (POOL && primary) {
                        // a P record: the walk's set-up was done when the ray was generated (box, shared whole-ray test)
                        const uint32_t* r = queue_c(Q) + i;
                        orig = rec_v3_load(r, Q.cap); dir = rec_v3_load(r + 3 * Q.cap, Q.cap);
                        ne.pdf = u2f(r[6 * Q.cap]); t_occ = u2f(r[7 * Q.cap]); tMin = u2f(r[8 * Q.cap]); tMax = u2f(r[9 * Q.cap]);
                        rec_rng_load(r + 10 * Q.cap, Q.cap, rng);
                        const uint32_t meta = r[16 * Q.cap];
                        id = meta_id(meta); ne.light = meta_light(meta);
                        L = V3(0.f, 0.f, 0.f); T = V3(1.f, 1.f, 1.f); k = 0u;
                        shadow = false; rng_live = false; hit = false;
                        t = tMin; clear_run = 0u; guard = 0u; tail_counted = false;
                        ray_skippable = SKIP && s.ray_skip && t_occ == INF;          // (counting builds only: the others never queue such a ray)
                        if (COUNT && ray_skippable) c.wskip++;
                        st = WALK;
                    } else {
                        id = meta_id(rec_c1_load(queue_c(Q) + i, Q.cap, vs.pt, ne, rng));
                        L = V3(0.f, 0.f, 0.f); T = V3(1.f, 1.f, 1.f); k = 0u;
                        begin_shadow();
                    }
                }
                next = min(nC, next + (uint32_t)__popcll(idle));
            }
            PROF_END(pe, 32u);
            if (next >= nC && __ballot(st != IDLE) == 0ull) break;
        }
    } else {
        // Deeper paths: a lane holds a path only while it walks.  When the walk is over the lane settles its result (radiance
        // of the shadow ray; collision or escape of the continuation) and the path WAITS FOR A SERVICE, of which there are
        // two: A = BSDF sampling + throughput + roulette + the next walk's set-up, B = shading + light sampling + the shadow
        // walk's set-up (~10^3 instructions, the same for 1 lane or 64).  With the waiting paths kept in their lanes, the
        // lanes of a wave were split between walking, waiting for A, waiting for B and waiting for a record, each service
        // was triggered on its own count, and 15 of 64 lanes walked on average (c3, traceDepth 4).  So waiting paths go
        // to memory instead -- one stack per service -- and a round of services takes ALL non-walking lanes through the
        // service with more paths waiting (lanes that already hold such a path keep it, the rest push theirs and pop).
        uint32_t nB = nB0;
        uint32_t* const qa = queue_a(Q);
        uint32_t* const qb = queue_b(Q);
        // after the next-event estimate of bounce k (pathtracer.cu:258-276): the next direction, the throughput, roulette,
        // and the next bounce's walk
        auto bounce = [&]() {
            v3 wi; float pdf = 0.f;
            const v3 f = bsdf_sample(vs, wi, pdf, rng);
            const float cosTerm = __builtin_fabsf(dot(normalize(vs.gradient), wi));
            if (fmax_(f.x, fmax_(f.y, f.z)) > 0.f && pdf > 0.f) {
                if (vs.st == 0) T = T * (f / (pdf * (1.f - vs.Pbrdf)));
                else T = T * ((f * cosTerm) / (pdf * vs.Pbrdf));
            }
            orig = vs.pt;
            dir = wi;
            if (k >= 3u && russian_roulette(T, rng)) { finish(); return; }
            ++k;
            begin_walk(false, false);                                       // the next bounce's walk (pathtracer.cu:218)
            if (st == END) {                                                // its result is known: no collision
                if (sc.env_on_escape) L = L + T * env_radiance(sc, dir);
                finish();
            }
        };
        // a shaded event without a light sample has no shadow walk: its estimate is settled
        auto after_shade = [&]() {
            if (ne.have) begin_shadow();
            else if (k + 1u >= traceDepth) finish();
            else st = WANT_A;
        };
        auto fence = [&]() {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");      // records are written and read by different lanes of this wave
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        };
        auto push_a = [&]() {
            const uint64_t m = __ballot(st == WANT_A);
            if (m == 0ull) return;
            if (st == WANT_A) { rec_a_store(qa + nA + lane_rank(m), Q.cap, vs, L, T, rng, rec_meta(id, k, 0u, vs.st)); st = IDLE; }
            nA += (uint32_t)__popcll(m);
        };
        auto push_b = [&]() {
            const uint64_t m = __ballot(st == WANT_B);
            if (m == 0ull) return;
            if (st == WANT_B) { rec_b_store(qb + nB + lane_rank(m), Q.cap, vs.pt, vs.wo, val, L, T, rng, rec_meta(id, k, 0u, 0)); st = IDLE; }
            nB += (uint32_t)__popcll(m);
        };
        // pops: the idle lanes take the newest records (still in the L2)
        auto pop_a = [&]() {
            const uint64_t idle = __ballot(st == IDLE);
            const uint32_t r = lane_rank(idle);
            fence();
            if (st == IDLE && r < nA) {
                const uint32_t meta = rec_a_load(qa + (nA - 1u - r), Q.cap, vs, L, T, rng);
                id = meta_id(meta); k = meta_k(meta);
                st = WANT_A;
            }
            nA -= min(nA, (uint32_t)__popcll(idle));
        };
        auto pop_b = [&]() {
            const uint64_t idle = __ballot(st == IDLE);
            const uint32_t r = lane_rank(idle);
            fence();
            if (st == IDLE && r < nB) {
                const uint32_t meta = rec_b_load(qb + (nB - 1u - r), Q.cap, vs.pt, vs.wo, val, L, T, rng);
                id = meta_id(meta); k = meta_k(meta);
                st = WANT_B;
            }
            nB -= min(nB, (uint32_t)__popcll(idle));
        };
        auto serve_a = [&]() {
            const uint64_t m = __ballot(st == WANT_A);
            if (m == 0ull) return;
            PROF_BEGIN(pe, PH_END);
            if (st == WANT_A) bounce();
            PROF_END(pe, (uint32_t)__popcll(m));
        };
        auto serve_b = [&]() {
            const uint64_t m = __ballot(st == WANT_B);
            if (m == 0ull) return;
            PROF_BEGIN(ps, PH_SHADE);
            if (st == WANT_B) { shade_event<LAYOUT, COUNT, LDSL>(s, vs, val, rng, ne, c, scp, lts); after_shade(); }
            PROF_END(ps, (uint32_t)__popcll(m));
        };
        const uint32_t park_end = s.park_end;
        for (;;) {
            PROF_BEGIN(pw, PH_CHEAP);
#if SVR_PROF
            uint32_t pc_it = 0u, pc_walk = 0u;
#endif
            while (__ballot(st == WALK || st == CELL) != 0ull) {
                // lanes a round of services could put to work
                const uint32_t n_end = (uint32_t)__popcll(__ballot(st == END)), n_idle = (uint32_t)__popcll(__ballot(st == IDLE));
                if (n_end + min(n_idle, nA + nB) >= park_end) break;
#if SVR_PROF
                pc_it++; pc_walk += (uint32_t)__popcll(__ballot(st == WALK));
#endif
                // A turn pays for the cell tests, fetches and re-marches of whichever lanes need them.  While most lanes are between
                // occupied stretches (or behind the last one, consuming the draws a later sample needs), they run their ~50-instruction
                // fetch-free iterations in a loop of their own, a few rounds at a time
                if (SKIP) {
#pragma nounroll
                    for (uint32_t rounds = 0u; rounds < SVR_FREE_ROUNDS && (uint32_t)__popcll(__ballot(st == WALK && t < t_occ)) >= SVR_FREE_MIN; ++rounds)
                        if (st == WALK && t < t_occ) free_iterate();
                }
                if (trips) {
                    if (st == CELL) iterate();
                    trip();
                } else if (st == WALK || st == CELL) iterate();
                serve_fetch_march();
            }
#if SVR_PROF
            PROF_END(pw, pc_it ? pc_walk / pc_it : 0u);
            if ((threadIdx.x & 63u) == 0u) { atomicAdd(&c_prof[2 * PH_N], (unsigned long long)pc_it); atomicAdd(&c_prof[2 * PH_N + 1], (unsigned long long)pc_walk); }
#endif
            // settle the walks that are over
            if (__ballot(st == END) != 0ull) {
                PROF_BEGIN(pr, PH_REFILL);
                if (st == END) {
                    if (shadow) {
                        nee();
                        if (k + 1u >= traceDepth) finish();            // sample_bsdf / roulette of the last bounce cannot reach L
                        else st = WANT_A;
                    } else if (!hit) {                                  // pathtracer.cu:231-236
                        if (sc.env_on_escape) L = L + T * env_radiance(sc, dir);
                        finish();
                    } else {
                        vs.wo = -dir;
                        vs.pt = orig + dir * t;
                        st = WANT_B;
                    }
                }
                PROF_END(pr, 32u);
            }
            const uint32_t cA = nA + (uint32_t)__popcll(__ballot(st == WANT_A));
            const uint32_t cB = nB + (uint32_t)__popcll(__ballot(st == WANT_B));
            if (cA + cB == 0u) {
                if (__ballot(st == WALK || st == CELL) == 0ull) break;   // no walk, no waiting path, no record: the queue is drained
                continue;
            }
            if (cB >= cA) {
                push_a(); pop_b(); serve_b();
                // the other service, for the lanes still idle, if they are many or nothing else can run
                if (nA != 0u && ((uint32_t)__popcll(__ballot(st == IDLE)) >= park_end || __ballot(st == WALK || st == CELL) == 0ull)) { pop_a(); serve_a(); }
            } else {
                push_b(); pop_a(); serve_a();
                if (nB != 0u && ((uint32_t)__popcll(__ballot(st == IDLE)) >= park_end || __ballot(st == WALK || st == CELL) == 0ull)) { pop_b(); serve_b(); }
            }
        }
    }
}

} // namespace svr
