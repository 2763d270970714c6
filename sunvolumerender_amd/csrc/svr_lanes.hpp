// svr_lanes.hpp -- everything of a path that follows its FIRST scatter event, as a per-lane state machine over a queue
// of scatter records (svr_trace_tile.hip, QUEUE builds).
//
// Why.  The primary walks of a wave are coherent (its lanes are frames of the same pixels: one shared whole-ray test,
// similar walk lengths).  What follows a scatter event is not: only some lanes scatter, a shadow walk through a medium
// that cannot be skipped takes ~10^2 iterations while the other lanes of the wave wait, and after a bounce every lane
// goes its own way (23 % lane utilisation at traceDepth 4, 17 % in the shadow-walk phase at depth 1:
// profiles/r01_notes_experiments.txt).  So the wave does not shade its hits in place: a lane that finds its first
// collision pushes a 14-word RECORD (position, incoming direction, intensity, generator state, path id) onto the wave's
// queue in global memory -- ballot + mbcnt prefix sum, so the records are dense and the stores coalesce -- and after
// QUEUE_TASKS tasks (2048 paths: enough records to refill the lanes many times over, so that the tail of the last,
// longest paths is a small part of a drain) the wave drains the queue with all 64 lanes: every lane runs a state machine
//
//      IDLE -> (pop a record) -> SHADE -> WALK (shadow) -> END -> [bounce: WALK (continuation) -> END -> SHADE ...] -> IDLE
//
// in which WALK is the cheap Woodcock iteration of svr_walk.hpp (FREE / EMPTY / CULLED), and everything expensive is a
// SERVICE the wave runs for the lanes that wait for it: FETCH (8 voxels + filter + LUT), MARCH (whole-ray re-march),
// SHADE (transfer function, 6 gradient fetches, light sampling, BSDF), END (transmittance -> radiance, BSDF sampling,
// roulette, next walk's set-up, or the path's end).  A lane that needs a service parks; the wave leaves the iteration
// loop when enough lanes are parked (or nobody can iterate), serves, refills idle lanes from the queue and goes on.
// All of it is scheduling: each path executes the reference's operations (pathtracer.cu:216-277) in the reference's
// order on its own generator, so the radiance is bit-identical to the straight-line code of trace_path_tile.
#pragma once
#include "svr_walk.hpp"

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

// queue of one wave: word j of record i at q[j * cap + i]
struct LaneQueue { uint32_t* q; uint32_t cap; };

SVR_DEV void queue_push(const LaneQueue& Q, uint32_t& count, bool hit, v3 pt, v3 wo, float val, const Rng& rng, uint32_t id)
{
    const uint64_t m = __ballot(hit);
    if (hit) {
        uint32_t* p = Q.q + count + lane_rank(m);
        const uint32_t cap = Q.cap;
        p[0] = f2u(pt.x); p[cap] = f2u(pt.y); p[2 * cap] = f2u(pt.z);
        p[3 * cap] = f2u(wo.x); p[4 * cap] = f2u(wo.y); p[5 * cap] = f2u(wo.z);
        p[6 * cap] = f2u(val);
        p[7 * cap] = rng.v0; p[8 * cap] = rng.v1; p[9 * cap] = rng.v2; p[10 * cap] = rng.v3; p[11 * cap] = rng.v4; p[12 * cap] = rng.d;
        p[13 * cap] = id;
    }
    count += (uint32_t)__popcll(m);
}

// Drain the wave's `count` records with all 64 lanes.  pendL: the wave's pending-radiance rows ([task * 3 + channel] of
// pend_row floats, LDS or global); a finished path with id = (task << 6 | lane) stores its radiance at row
// (id >> 6) * 3 + channel, column id & 63.
template <int LAYOUT, bool COUNT, bool SKIP, bool DEPTH1, typename LDS>
SVR_DEV void drain_queue(const DevScene& s, const LDS& L_, const LaneQueue& Q, uint32_t count, uint32_t traceDepth_, float* pendL, uint32_t pend_row, Cnt& c,
                         unsigned long long* c_prof = nullptr)
{
    enum : uint32_t { IDLE = 0u, SHADE = 1u, WALK = 2u, FETCH = 3u, MARCH = 4u, END = 5u };
    const float INF = u2f(SVR_INF_BITS);
    const uint32_t traceDepth = DEPTH1 ? 1u : traceDepth_;
    uint32_t next = 0u;                       // wave-uniform: next record to pop
    uint32_t st = IDLE;
    // walk state
    Rng rng = {0u, 0u, 0u, 0u, 0u, 0u};
    v3 orig = V3(0.f, 0.f, 0.f), dir = V3(0.f, 0.f, 1.f);
    float t = 0.f, tMin = 0.f, tMax = 0.f, t_occ = 0.f, xi = 0.f, val = 0.f;
    uint32_t clear_run = 0u, guard = 0u;
    bool shadow = false, rng_live = false, hit = false, have_light = false, tail_counted = false, ray_skippable = false;
    // path state
    uint32_t id = 0u, k = 0u, lightId = 0u;
    v3 L = V3(0.f, 0.f, 0.f), T = V3(1.f, 1.f, 1.f), B = V3(0.f, 0.f, 0.f);
    float pdfL = 1.f;
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
        if (COUNT) { c.iters++; if (ray_skippable || tail_counted) c.iskip++; }
        t += -logf_unit(1.f - rng_uniform(rng)) * s.invSigmaMaxSI;
        if (t > tMax || guard++ >= SVR_WALK_GUARD) { st = END; return; }
        if (COUNT) c.taps++;
        if (SKIP && t < t_occ) {
            if (COUNT && !(ray_skippable || tail_counted)) c.ipre++;
            rng_skip(rng);                                  // the accept draw of a FREE iteration
            return;
        }
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

    // the services of a walking lane that cannot wait: the fetch an iteration asked for, the re-march after an occupied stretch
    auto serve_fetch_march = [&](bool mine) {
        // FETCH: 8 voxels + filter + LUT, then the accept test with the draw the lane kept
        if (__ballot(mine && st == FETCH) != 0ull) {
            if (mine && st == FETCH) {
                if (COUNT) c.exec++;
                val = tex_fetch<LAYOUT>(s, cell_of(s, orig + dir * t)) * s.densityScale;
                const float sigma_t = alpha_of(L_, s, val);
                if (xi < sigma_t * s.invSigmaMax) { st = END; hit = true; }
                else st = WALK;
            }
        }
        // MARCH: the walk has left an occupied stretch: where is the next one?
        if (SKIP && __ballot(mine && st == MARCH) != 0ull) {
            if (mine && st == MARCH) {
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
        float* o = pendL + (id >> 6) * 3u * pend_row + (id & 63u);
        o[0] = L.x; o[pend_row] = L.y; o[2u * pend_row] = L.z;
        st = IDLE;
    };
    // END of a CONTINUATION walk (pathtracer.cu:231-244): no collision -> the path is over; collision -> the next scatter point
    auto end_continuation = [&]() {
        if (!hit) {
            if (s.env_on_escape) L = L + T * env_radiance(s, dir);
            finish();
        } else {
            vs.wo = -dir;
            vs.pt = orig + dir * t;
            st = SHADE;
        }
    };
    // END of a SHADOW walk (pathtracer.cu:191-198, 258-276): transmittance -> direct light, then the next bounce's direction and walk
    auto end_shadow = [&]() {
        v3 Ld = V3(0.f, 0.f, 0.f);
        if (have_light) {
            // transmittance.h:15-16 on the walk's result ts (t, or -FLT_MAX), with the box interval of the shadow ray
            const float ts = hit ? t : -SVR_FLT_MAX;
            const float Tr = ((ts > tMin) && (ts < tMax)) ? 0.f : 1.f;
            const float kf = Tr * (float)s.num_lights;
            const DevLight& l = s.lights[lightId];
            const v3 Li = V3(l.radiance[0], l.radiance[1], l.radiance[2]);     // sample_light returned true: cosTerm > 0
            Ld = ((B * kf) * Li) / pdfL;
        }
        L = L + T * Ld;
        if (k + 1u >= traceDepth) { finish(); return; }                 // sample_bsdf / roulette of the last bounce cannot reach L
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
            if (s.env_on_escape) L = L + T * env_radiance(s, dir);
            finish();
        }
    };
    // SHADE: VolumeSample + next-event estimation up to the shadow walk (pathtracer.cu:237-257, 171-191)
    auto shade = [&]() {
        if (COUNT) { c.scatter++; c.taps += 7; c.exec += 6; }
        tf_rgba(s, s.tf, val, vs.color);
        {
            // Gradient_CentralDiff, cuda_volume.h:54-61
            const v3 q = vs.pt;
            float xd = intensity_at<LAYOUT>(s, V3(q.x + s.spacing[0], q.y + 0.f, q.z + 0.f)) -
                       intensity_at<LAYOUT>(s, V3(q.x - s.spacing[0], q.y - 0.f, q.z - 0.f));
            float yd = intensity_at<LAYOUT>(s, V3(q.x + 0.f, q.y + s.spacing[1], q.z + 0.f)) -
                       intensity_at<LAYOUT>(s, V3(q.x - 0.f, q.y - s.spacing[1], q.z - 0.f));
            float zd = intensity_at<LAYOUT>(s, V3(q.x + 0.f, q.y + 0.f, q.z + s.spacing[2])) -
                       intensity_at<LAYOUT>(s, V3(q.x - 0.f, q.y - 0.f, q.z - s.spacing[2]));
            vs.gradient = V3((xd * 0.5f) * s.invSpacing[0], (yd * 0.5f) * s.invSpacing[1], (zd * 0.5f) * s.invSpacing[2]);
        }
        const float gradMag = __builtin_sqrtf(dot(vs.gradient, vs.gradient));
        vs.Pbrdf = vs.color[3] * (1.f - expf_(s.pbrdf_c * gradMag * 65535.f * s.invMaxMagnitude));
        vs.st = (rng_uniform(rng) < vs.Pbrdf) ? 1 : 0;
        // estimate_direct_light, pathtracer.cu:171-198
        have_light = false;
        orig = vs.pt;
        st = END; shadow = true; hit = false;
        if (s.num_lights != 0) {
            int li = (int)((float)s.num_lights * rng_uniform(rng));
            li = li < (int)s.num_lights ? li : (int)s.num_lights - 1;
            v3 wiL, Li;
            if (sample_light(s.lights[li], vs.pt, rng, wiL, pdfL, Li)) {
                have_light = true;
                lightId = (uint32_t)li;
                B = bsdf_eval(vs, wiL);
                if (COUNT) c.shadow++;
                dir = wiL;
                // the draws of sample_bsdf / roulette follow the shadow walk unless this is the last bounce
                begin_walk(true, k + 1u < traceDepth);
            }
        }
    };

    // refill the idle lanes from the queue: a popped record is a path at its first scatter point
    auto refill = [&]() {
        const uint64_t idle = __ballot(st == IDLE);
        if (idle == 0ull || next >= count) return;
        PROF_BEGIN(pr, PH_REFILL);
        const uint32_t n_idle = (uint32_t)__popcll(idle);
        const uint32_t i = next + lane_rank(idle);
        if (st == IDLE && i < count) {
            const uint32_t* p = Q.q + i;
            const uint32_t cap = Q.cap;
            vs.pt = V3(u2f(p[0]), u2f(p[cap]), u2f(p[2 * cap]));
            vs.wo = V3(u2f(p[3 * cap]), u2f(p[4 * cap]), u2f(p[5 * cap]));
            val = u2f(p[6 * cap]);
            rng.v0 = p[7 * cap]; rng.v1 = p[8 * cap]; rng.v2 = p[9 * cap]; rng.v3 = p[10 * cap]; rng.v4 = p[11 * cap]; rng.d = p[12 * cap];
            id = p[13 * cap];
            L = V3(0.f, 0.f, 0.f); T = V3(1.f, 1.f, 1.f); k = 0u;
            st = SHADE;
        }
        PROF_END(pr, min(n_idle, count - next));
        next = min(count, next + n_idle);
    };

    // Scheduling (phase profiles of experiment builds, c3 at traceDepth 4): the services (END: end of a shadow walk + BSDF
    // sampling + next walk's set-up; SHADE) cost ~10^3 instructions each, the same for 1 lane or 64.  Serving as soon as 16
    // lanes waited ran them at 30-40 % lane utilisation and 65 % of the drain's time; serving only when NO lane can walk
    // (strict generations) ran them at 90-99 % but left the wave waiting for its few longest walks (51 % of the time at 7
    // walking lanes).  So a service runs when park_end lanes wait for it, or when nobody walks.
    const uint32_t park_end = s.park_end;
    if constexpr (DEPTH1) {
        // traceDepth 1: a path is over when its shadow walk is, so ONE service takes every waiting lane through
        // [end of walk -> radiance -> next record -> shading -> next shadow walk] (c3: 8032 against 7336 Msamples/s for two
        // separately triggered services, 7763 for straight-line paths)
        for (;;) {
            PROF_BEGIN(pw, PH_CHEAP);
            while (__ballot(st == WALK) != 0ull) {
                if ((uint32_t)__popcll(__ballot(st == END || st == SHADE || (st == IDLE && next < count))) >= park_end) break;
                if (st == WALK) iterate();
                serve_fetch_march(true);
            }
            PROF_END(pw, 32u);
            if (__ballot(st == END) != 0ull) {
                PROF_BEGIN(pe, PH_END);
                if (st == END) end_shadow();
                PROF_END(pe, 32u);
            }
            refill();
            {
                const uint64_t m = __ballot(st == SHADE);
                if (m != 0ull) {
                    PROF_BEGIN(ps, PH_SHADE);
                    if (st == SHADE) shade();
                    PROF_END(ps, (uint32_t)__popcll(m));
                }
            }
            if (__ballot(st != IDLE) == 0ull) break;
        }
    } else {
        // deeper paths: the two services are triggered separately and the continuation walks (a few iterations inside the
        // medium) run in the common walk loop -- taking the served lanes through [end -> continuation walk -> end -> shading]
        // in one go with the walks in flight paused was slower (c3 depth 2 / 4: 2977 / 1996 against 3784 / 2491 Msamples/s)
        for (;;) {
            PROF_BEGIN(pw, PH_CHEAP);
#if SVR_PROF
            uint32_t pc_it = 0u, pc_walk = 0u;
#endif
            while (__ballot(st == WALK) != 0ull) {
                if ((uint32_t)__popcll(__ballot(st == END)) >= park_end) break;
                if ((uint32_t)__popcll(__ballot(st == SHADE || (st == IDLE && next < count))) >= park_end) break;
#if SVR_PROF
                pc_it++; pc_walk += (uint32_t)__popcll(__ballot(st == WALK));
#endif
                if (st == WALK) iterate();
                serve_fetch_march(true);
            }
#if SVR_PROF
            PROF_END(pw, pc_it ? pc_walk / pc_it : 0u);
            if ((threadIdx.x & 63u) == 0u) { atomicAdd(&c_prof[2 * PH_N], (unsigned long long)pc_it); atomicAdd(&c_prof[2 * PH_N + 1], (unsigned long long)pc_walk); }
#endif
            const bool walking = __ballot(st == WALK) != 0ull;
            {
                const uint64_t m = __ballot(st == END);
                if (m != 0ull && (!walking || (uint32_t)__popcll(m) >= park_end)) {
                    PROF_BEGIN(pe, PH_END);
                    if (st == END) { if (shadow) end_shadow(); else end_continuation(); }
                    PROF_END(pe, (uint32_t)__popcll(m));
                    if (__ballot(st == WALK) != 0ull) continue;        // the next bounce's walks first: their hits join the shading below
                }
            }
            const bool shade_now = !walking || (uint32_t)__popcll(__ballot(st == SHADE || (st == IDLE && next < count))) >= park_end;
            if (shade_now) refill();
            {
                const uint64_t m = __ballot(st == SHADE);
                if (m == 0ull && !walking && __ballot(st != IDLE) == 0ull) break;   // no walk, no end, no record, nothing to shade: the queue is drained
                if (m == 0ull || !shade_now) continue;
                PROF_BEGIN(ps, PH_SHADE);
                if (st == SHADE) shade();
                PROF_END(ps, (uint32_t)__popcll(m));
            }
        }
    }
}

} // namespace svr
