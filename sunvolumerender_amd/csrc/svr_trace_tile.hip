// svr_trace_tile.hip -- the default path-tracing kernel for gfx950.
//
// Shape (what the rocprof counters of the earlier kernels asked for; DESIGN.md 5.1 has the measurements):
//  * persistent 1024-thread blocks, one per CU; a wave pulls one task = (64 >> f) pixels x (1 << f) frames of a
//    launch group from sharded tickets.  With many frames per launch the lanes of a wave are the same two pixels in
//    32 different frames: they start in the same voxels, share their whole-ray test (svr_walk.hpp,
//    first_occupied_group) and stay within a few dozen voxels of depth of each other, so the 256-B bricks they touch
//    come from the vector L1 / L2, not from HBM;
//  * the transfer-function alpha LUT (the only table the Woodcock loop needs), the `empty` and `deep-empty`
//    macro-cell bitmasks and a half-resolution distance field live in LDS (84 KB);
//  * EMPTY-SPACE SKIPPING, bit-exact: a Woodcock iteration whose trilinear cell lies in a macro-cell where the
//    transfer function's alpha is exactly 0 for every reachable intensity has sigma_t == 0, so the reference's
//    accept test `xi < sigma_t / sigma_max` (woodcock_tracking.h:43) fails whatever the fetch returns; the 8 voxel
//    loads, the filter and the LUT read are skipped while both random draws of the iteration are still consumed, so
//    every path follows the reference's RNG stream and produces the same bits.  Whole rays and stretches of rays
//    that cannot collide are found by sphere tracing on the distance field (svr_walk.hpp) and never iterate at all
//    when no random draw follows them.  On c3, 68 % of the reference's Woodcock iterations never run and 95 % of its
//    taps are never fetched;
//  * the scatter point's intensity is the last Woodcock fetch (same position, same arithmetic), not a second fetch;
//  * voxel addresses are 32-bit byte offsets from a scalar base (saddr global loads), brick strides use 24-bit
//    multiplies.
// Radiance goes to the scratch slots; k_resolve (svr_kernels.hip) folds it into the running mean.
#include "svr_walk.hpp"
#include "svr_lanes.hpp"
#include "svr_tile_tasks.hpp"
#include "svr_primary.hpp"

namespace svr {

// timing-ablation stops (DevWork.debug_stop, wrong images) exist only in experiment builds
#ifdef SVR_TEST_HOOKS
#define SVR_DEBUG_STOPS 1
#else
#define SVR_DEBUG_STOPS 0
#endif

// wave-synchronous re-marching of walks that leave an occupied stretch (svr_walk.hpp, REMARCH).  Measured on c3
// (ms per frame): none 0.323, shadow walks only 0.292, primary only 0.400, both 0.376 -- shadow walks start inside
// the medium and almost always end at the march; a primary march is paid by every tile that has one grazing ray.
#ifndef SVR_SHADOW_REMARCH
#define SVR_SHADOW_REMARCH true
#endif
#ifndef SVR_PRIMARY_REMARCH
#define SVR_PRIMARY_REMARCH false
#endif

// one path: kernel_pathtracer body, pathtracer.cu:205-277
// DEPTH1: traceDepth == 1 (the reference's default, gui/canvas.cpp:17) known at compile time: throughput and
// accumulated radiance are constants when the shadow walk runs and nothing of the VolumeSample outlives
// the next-event estimate, which leaves registers for the re-marching shadow walk.
template <int LAYOUT, bool COUNT, bool SKIP, bool DEPTH1, typename LDS>
SVR_DEV v3 trace_path_tile(const DevScene& s, const LDS& L_, uint32_t x, uint32_t y, uint32_t traceDepth_,
                           uint32_t hashed, uint32_t debug_stop, bool group_march, uint32_t P2, GroupMapShared* gslot, Cnt& c)
{
    const uint32_t traceDepth = DEPTH1 ? 1u : traceDepth_;
    uint32_t offset = y * s.imageW + x;
    Rng rng;
    rng_init(rng, hashed + offset);
    if (COUNT) c.paths++;
    v3 L = V3(0.f, 0.f, 0.f), T = V3(1.f, 1.f, 1.f);
    v3 orig, dir;
    camera_ray(s, x, y, rng, orig, dir);
    float ls_t;
    int ls_id = nearest_light(s, orig, dir, ls_t);
    if (SVR_DEBUG_STOPS && debug_stop == 1u) return V3(ls_t, orig.x, dir.x);
    if (SVR_DEBUG_STOPS && debug_stop == 2u) {
        float tn, tf;
        if (!volume_intersect(s, orig, dir, tn, tf)) return V3(0.f, 0.f, 0.f);
        return V3(first_occupied(s, L_, orig, dir, tn < 0.f ? 1e-6f : tn, tf), 0.f, 0.f);
    }
    for (uint32_t k = 0; k < traceDepth; ++k) {
        float tMin = (float)1e-6, tMax = SVR_FLT_MAX, val = 0.f;
        float t;
        if (SKIP && k == 0 && group_march) {
            // the wave is full and its lanes are pixels x frames: one shared whole-ray test per pixel
            float t_occ;
            GroupMap map;
            map.g = gslot + ((threadIdx.x & 63u) & ((1u << P2) - 1u) & (GROUP_MAPS_PER_WAVE - 1u));      // this wave's slot of the lane's pixel group
            int r = walk_setup_group<COUNT, SKIP>(s, L_, P2, orig, dir, false, tMin, tMax, t_occ, map);
            t = r <= 0 ? -SVR_FLT_MAX
                       : walk_run<LAYOUT, COUNT, SKIP, false, true>(s, L_, orig, dir, rng, tMin, tMax, t_occ, val, false, c, &map, P2);
        } else
            t = walk<LAYOUT, COUNT, SKIP, SVR_PRIMARY_REMARCH>(s, L_, orig, dir, rng, tMin, tMax, val, false, c);
        if (SVR_DEBUG_STOPS && debug_stop == 3u) return V3(t, val, 0.f);
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
                if (SVR_DEBUG_STOPS && debug_stop == 4u) return V3(wiL.x, pdfL, Li.x + vs.Pbrdf);
                // the draws of sample_bsdf / roulette follow the shadow walk unless this is the last bounce
                float ts = walk<LAYOUT, COUNT, SKIP, SVR_SHADOW_REMARCH>(s, L_, vs.pt, wiL, rng, sMin, sMax, sval, k + 1u < traceDepth, c);
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
#define SVR_TILE_THREADS 1024     // 16 waves share one 84 KB LDS image (alpha LUT, two 32 KB bitmasks, 16 KB distance field): one block per CU
#endif

// Radiance of the last PEND_TASKS tasks of a wave, in LDS, when the launch folds the frames of a pixel into the
// accumulator itself (DevWork.fold): [task][channel][lane], rows padded to 65 words so that the fold lanes (one per
// pixel and channel) read conflict-free.
constexpr uint32_t PEND_TASKS = 4;
constexpr uint32_t PEND_ROW = 65;
// QUEUE builds keep the radiance of the last QUEUE_TASKS tasks in a per-wave buffer in global memory instead (rows of 64
// floats; written and read back once by the same wave), and up to QUEUE_CAP path records per wave
// (QUEUE_TASKS = 32 and QUEUE_CAP = 1024: svr_kernels.hpp, shared with the allocation in svr_api.hip)
static_assert(TILE_WAVES == SVR_TILE_THREADS / 64, "svr_kernels.hpp sizes the queues for 16 waves per block");
struct LdsPend {
    float L[TILE_WAVES][PEND_TASKS * 3][PEND_ROW];
    uint32_t task[TILE_WAVES][PEND_TASKS];
};
struct LdsPendQueue {                              // QUEUE builds: only the task numbers stay in LDS
    uint32_t task[TILE_WAVES][QUEUE_TASKS];
};

// POOL (QUEUE builds at traceDepth 1, media where the primary walks of a wave are NOT coherent -- fog-like data without exactly
// transparent space, c3n): the primary walks are pooled too.  A task only generates its camera rays (P records); at a flush the lane
// machine walks them (a lane pops a ray, walks, settles, pops the next), the collisions are shaded 64 at a time into C1 records, and
// the machine walks those -- the wave-sized wavefront of svr_trace_lm.hip with the bit-exact walk.  Scheduling only.
// DIRECT (QUEUE builds, launches that do not fold: frames traced ahead of the calls that ask for them): a path's radiance goes straight to its scratch slot
// lbuf[frame][pixel] when it is final, instead of through the wave's rows and a fold.
template <int LAYOUT, bool COUNT, bool SKIP, bool DEPTH1, bool QUEUE, bool POOL = false, bool DIRECT = false>
__global__ __launch_bounds__(SVR_TILE_THREADS, SVR_TILE_WAVES_PER_EU) void k_trace_tile(const DevScene s, const DevWork w)
{
    static_assert(!POOL || (QUEUE && SKIP), "the pool form exists for skipping builds with the queue machine");
    static_assert(!DIRECT || QUEUE, "the straight-line builds write their scratch slots anyway");
    using LDS = typename std::conditional<SKIP, typename std::conditional<POOL, LdsTilePool, LdsTileCull>::type, LdsTileNoMask>::type;
    __shared__ LDS lds;
    __shared__ GroupMapShared gmaps[TILE_WAVES][GROUP_MAPS_PER_WAVE];
    __shared__ typename std::conditional<QUEUE, LdsPendQueue, LdsPend>::type pend;
    // QUEUE builds: the lights in LDS -- light sampling and the settling of a shadow walk index them PER LANE (a vector load from the kernarg
    // segment otherwise: a dependent memory round trip in every settling round of the lane machine)
#ifndef SVR_LDS_LIGHTS
#define SVR_LDS_LIGHTS 1
#endif
#ifndef SVR_COLD_DEEP_POOL
#define SVR_COLD_DEEP_POOL 1           // the deeper build with pooled primary walks (fog: c3n at depth >= 2) gains 12.5 % with both (1 255 -> 1 412, 835 -> 940)
#endif
#ifndef SVR_COLD_DEEP_ALL
#define SVR_COLD_DEEP_ALL 1            // ... and so does the fused deeper build without pooled walks -- with BOTH: 4 897 -> 5 115 at depth 2, 2 995 -> 3 176 at depth 4 (same box),
                                       // while the laundered constants alone cost it 8 % and the LDS lights alone 2.5 %
#endif
    __shared__ DevLight lds_lights[QUEUE && SVR_LDS_LIGHTS ? 8 : 1];
    if constexpr (QUEUE && SVR_LDS_LIGHTS) {
        if (threadIdx.x < 8u * (sizeof(DevLight) / 4u))
            reinterpret_cast<float*>(lds_lights)[threadIdx.x] = reinterpret_cast<const float*>(s.lights)[threadIdx.x];
    }
    // (traceDepth-1 builds: same-box A/B c3 9 626-9 805 against 9 522-9 526, c3n 2 403 / 2 386, c5 6 549 / 6 489; the deeper build loses 2.5 % with it)
    constexpr bool LDSL = QUEUE && (DEPTH1 || (POOL && SVR_COLD_DEEP_POOL) || SVR_COLD_DEEP_ALL) && SVR_LDS_LIGHTS;
    const DevLight* const lts = lds_lights;
    lds_tile_load(lds, s, SKIP);

    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = threadIdx.x >> 6;
    // QUEUE builds: the scene constants that only set-up, shading and settling read (camera, lights, environment, transfer-function base, spacing: ~60 of
    // the ~150 the kernel touches) are read through a pointer INTO THE KERNARG SEGMENT that is laundered per use, so the compiler scalar-loads them where
    // they are used instead of keeping them in scalar registers -- i.e. in spilled ones: v_writelane / v_readlane -- across the walk loops
    // (svr_lanes.hpp, shade_event).  DevScene is the kernel's first argument: offset 0 of the segment.
#ifndef SVR_COLD_SCENE
#define SVR_COLD_SCENE 1
#endif
    auto cold_scene = [&]() -> const DevScene* {
        // (traceDepth-1 builds only: same-box A/B c3 9 690 against 9 500, c3n 2 381 / 2 351, c5 6 561 / 6 438; the deeper build LOSES 8 % with it --
        // 4 508 against 4 884 at depth 2 -- its services settle walks one scalar load latency at a time)
        if constexpr (QUEUE && (DEPTH1 || (POOL && SVR_COLD_DEEP_POOL) || SVR_COLD_DEEP_ALL) && SVR_COLD_SCENE) {
            auto p = __builtin_amdgcn_kernarg_segment_ptr();          // (a pointer into the constant address space: the loads stay scalar)
            asm volatile("" : "+s"(p));
            return (const DevScene*)p;
        } else return nullptr;
    };
    // Lanes of a wave = (64 >> fl2) pixels x (1 << fl2) frames of the group: rays of one pixel in different frames
    // share origin, box segment and whole-ray test result up to sub-pixel jitter, so a wave is far more uniform
    // (skip / walk / hit, walk lengths) than 64 different pixels of one frame, and it touches fewer bricks.
    const TaskShape ts = task_shape(w);
    const uint32_t fl2 = ts.fl2, P2 = ts.P2, tw2 = ts.tw2, th2 = ts.th2, wv = ts.wv, fgroups = ts.fgroups;
    const uint32_t n_tasks = ts.tiles_x * ts.tiles_y * fgroups;
    Cnt c = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

    // Work distribution.  A single returning atomic saturates near 88 dequeues/us chip-wide
    // (MI355X_MICROARCH.md "dequeue"), which capped this kernel at 0.19 ms per 1024^2 frame.  So tasks are handed
    // out (in units of w.unit consecutive tasks) from TICKET_SHARDS counters; counter j owns the units j, j + 8,
    // j + 16 ... of ONE global order, and a block starts on counter blockIdx % 8 and moves on when that is drained.
    // The global order runs over the tile rows from the image centre outwards: the expensive tiles (the object is
    // where the camera looks) are started first and the launch ends on cheap, mostly skipped ones, instead of
    // waiting for a few long tasks -- the tail was 0.15 ms of every launch, a fifth of a launch on one eighth of
    // the image.  (Placement affects speed only, never results.)
    const uint32_t unit = w.unit;
    const uint32_t n_units = (n_tasks + unit - 1u) / unit;
    const uint32_t shard0 = blockIdx.x % TICKET_SHARDS;
    const bool fold = w.fold != 0u;                               // host guarantees fgroups == 1 then
    uint32_t npend = 0, qC = 0, qA = 0, qB = 0;
    LaneQueue Q;
    Q.cap = QUEUE_CAP;
    Q.q = QUEUE ? w.queue + (size_t)(blockIdx.x * TILE_WAVES + wave) * (REC_WORDS * QUEUE_CAP) : nullptr;
    float* const gpend = QUEUE ? w.pend + (size_t)(blockIdx.x * TILE_WAVES + wave) * (QUEUE_TASKS * 3u * 64u) : nullptr;
    // hand the pending tasks' radiance on.  QUEUE: first drain the scatter records with all 64 lanes (svr_lanes.hpp)
    auto flush = [&]() {
        if constexpr (QUEUE) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");      // records and radiance are read back by other lanes of this wave
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            if constexpr (POOL && !DEPTH1) {
                // deeper paths: qC camera rays (P records) -> collisions = B records (first scatter events, unshaded) of the machine below
                static_assert(REC_C1_WORDS <= REC_A_WORDS, "P records lie in front of the B stack");
                uint32_t nH = 0u;
                drain_queue<LAYOUT, COUNT, SKIP, true, LDS, true, true, DIRECT, LDSL>(s, lds, Q, qC, 0u, 0u, 1u, gpend, 64u, c, w.counters + CNT_N, true, &nH, &w, &pend.task[wave][0], cold_scene(), lts);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                qC = 0u; qA = 0u; qB = nH;
            } else if constexpr (POOL) {
                // qC camera rays (P records) -> collisions (H records) -> shaded, 64 at a time -> C1 records -> their shadow walks
                uint32_t nH = 0u;
                drain_queue<LAYOUT, COUNT, SKIP, DEPTH1, LDS, true, false, DIRECT, LDSL>(s, lds, Q, qC, 0u, 0u, 1u, gpend, 64u, c, w.counters + CNT_N, true, &nH, &w, &pend.task[wave][0], cold_scene(), lts);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                qC = 0u;
                for (uint32_t i0 = 0u; i0 < nH; i0 += 64u) {
                    const uint32_t i = i0 + lane;
                    bool have = false;
                    Shade vs;
                    vs.pt = V3(0.f, 0.f, 0.f); vs.wo = vs.pt; vs.gradient = vs.pt; vs.color[0] = vs.color[1] = vs.color[2] = vs.color[3] = 0.f; vs.Pbrdf = 0.f; vs.st = 0;
                    Nee ne;
                    ne.wi = vs.pt; ne.B = vs.pt; ne.pdf = 1.f; ne.light = 0u; ne.have = false;
                    Rng rng = {0u, 0u, 0u, 0u, 0u, 0u};
                    uint32_t id = 0u;
                    if (i < nH) {
                        const uint32_t* h = queue_h(Q) + i;
                        vs.pt = rec_v3_load(h, Q.cap); vs.wo = rec_v3_load(h + 3 * Q.cap, Q.cap);
                        const float val = u2f(h[6 * Q.cap]);
                        rec_rng_load(h + 7 * Q.cap, Q.cap, rng);
                        id = h[13 * Q.cap];
                        shade_event<LAYOUT, COUNT, LDSL>(s, vs, val, rng, ne, c, cold_scene(), lts);
                        have = ne.have;
                        if (!have) {                                          // a first event no light sample reaches: L = 0
                            if constexpr (DIRECT) direct_put(s, w, &pend.task[wave][0], id, V3(0.f, 0.f, 0.f));
                            else {
                                float* o = gpend + (id >> 6) * (3u * 64u) + (id & 63u);
                                o[0] = 0.f; o[64] = 0.f; o[128] = 0.f;
                            }
                        }
                    }
                    queue_push_c1(Q, qC, have, vs.pt, ne, rng, id);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            }
            drain_queue<LAYOUT, COUNT, SKIP, DEPTH1, LDS, false, false, DIRECT, LDSL>(s, lds, Q, qC, qA, qB, w.traceDepth, gpend, 64u, c, w.counters + CNT_N, false, nullptr, &w, &pend.task[wave][0], cold_scene(), lts);
            qC = qA = qB = 0;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#if SVR_PROF
            unsigned long long* c_prof = w.counters + CNT_N;
#endif
            PROF_BEGIN(pfo, PH_FOLD);
            if constexpr (!DIRECT) fold_pending(s, w, gpend, 64u, &pend.task[wave][0], npend);
            PROF_END(pfo, min(64u, npend * (3u << ts.P2)));
        } else {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // pend.L / pend.task are written by one lane and read by another lane of this wave
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            fold_pending(s, w, &pend.L[wave][0][0], PEND_ROW, &pend.task[wave][0], npend);
        }
        npend = 0;
    };

    for (uint32_t si = 0; si < TICKET_SHARDS; ++si) {
        const uint32_t shard = (shard0 + si) % TICKET_SHARDS;
        uint32_t* ticket = w.ticket + shard * TICKET_STRIDE;
        for (;;) {
            // away from the home counter, look before taking: when the launch runs out, every wave visits every
            // counter once, and 4096 x 8 returning atomics alone took ~45 us (a plain load of a drained counter
            // is free; the counters only grow, so a stale value just means one atomic more)
            // w.row_order: counter j owns whole TILE ROWS (rows j, j + 8 ... of the centre-out order) instead of every 8th task:
            // blocks b and b + 8 share an XCD (MI355X_MICROARCH.md), a block starts on counter b % 8, so consecutive tiles of
            // a row -- whose ray tubes run through the same bricks -- are traced on one XCD and find them in its L2
            const uint32_t rows_s = (w.row_order && ts.tiles_y > shard) ? (ts.tiles_y - shard + TICKET_SHARDS - 1u) / TICKET_SHARDS : 0u;
            const uint32_t limit = w.row_order ? rows_s * ts.row_tasks : n_units;
            if (si != 0u && (w.row_order ? __atomic_load_n(ticket, __ATOMIC_RELAXED) : __atomic_load_n(ticket, __ATOMIC_RELAXED) * TICKET_SHARDS + shard) >= limit) break;
            uint32_t u = 0;
            if (lane == 0) u = atomicAdd(ticket, 1u);
            u = __builtin_amdgcn_readfirstlane(u);
            uint32_t t_begin, t_end;
            if (w.row_order) {
                if (u >= limit) break;
                const uint32_t rq = u / ts.row_tasks;
                t_begin = (shard + TICKET_SHARDS * rq) * ts.row_tasks + (u - rq * ts.row_tasks);
                t_end = t_begin + 1u;
            } else {
                const uint32_t unit_id = u * TICKET_SHARDS + shard;
                if (unit_id >= n_units) break;
                t_begin = unit_id * unit;
                t_end = min(t_begin + unit, n_tasks);
            }
            for (uint32_t k = t_begin; k < t_end; ++k) {
                // k-th task of the centre-out order -> (tile row, tile column, frame group)
                uint32_t tx, ty, fg;
                task_decode(ts, k, tx, ty, fg);
                if (COUNT) c.loops += (lane == 0);
                uint32_t pl = lane & ((1u << P2) - 1u);
                uint32_t slot = (fg << fl2) + (lane >> P2);
                uint32_t px = (tx << tw2) + (pl & ((1u << tw2) - 1u));
                uint32_t r = (ty << th2) + (pl >> tw2);
                const bool live = px < wv && r < w.n_rows && slot < w.nframes;
                // shared whole-ray test: >= 8 frames of a pixel in the wave, every lane alive (the group shuffles)
                const bool group_march = SKIP && fl2 >= 3u && (!SVR_DEBUG_STOPS || w.debug_stop == 0u || w.debug_stop >= 3u) && __ballot(live) == ~0ull;
                if constexpr (POOL) {
                    Rng rng = {0u, 0u, 0u, 0u, 0u, 0u};
                    v3 L = V3(0.f, 0.f, 0.f), orig = L, dir = V3(0.f, 0.f, 1.f);
                    float tMin = 0.f, tMax = 0.f, t_occ = 0.f, ls_t = 0.f;
                    int ls_id = -1;
                    bool queued = false;
                    if (live) {
                        uint32_t x = w.x0 + px, y = owned_row_to_y(w, r);
                        queued = gen_primary<COUNT, SKIP>(s, lds, x, y, wang_hash(w.frame0 + slot), group_march, P2, &gmaps[wave][0], c, rng, L, orig, dir, tMin, tMax, t_occ, ls_t, ls_id, cold_scene());
                    }
                    {
                        const uint64_t m = __ballot(queued);
                        if (queued) rec_p_store(queue_c(Q) + qC + lane_rank(m), Q.cap, orig, dir, ls_t, t_occ, tMin, tMax, rng, (npend << 6) | lane, (uint32_t)(ls_id + 1));
                        qC += (uint32_t)__popcll(m);
                    }
                    if (!queued) {
                        if constexpr (DIRECT) {
                            if (live) {
                                float* o = w.lbuf + (size_t)slot * w.slot_stride + 3 * ((size_t)owned_row_to_y(w, r) * s.imageW + (w.x0 + px));
                                o[0] = L.x; o[1] = L.y; o[2] = L.z;
                            }
                        } else {
                            float* o = gpend + (size_t)npend * (3u * 64u) + lane;
                            o[0] = L.x; o[64] = L.y; o[128] = L.z;
                        }
                    }
                    if (lane == 0) pend.task[wave][npend] = k;
                    if (++npend == QUEUE_TASKS || qC + 64u > QUEUE_CAP) flush();
                    continue;
                } else if constexpr (QUEUE) {
                    // primary walk, then the hits' first scatter events are shaded in place -- the lanes are frames of the
                    // same pixels and scatter together.  traceDepth 1: the shaded events are queued for their shadow walks.
                    // Deeper: the shadow walk runs in place too (queueing it as well was slower: c3 depth 2 / 4, 4174 / 2602
                    // against 4380 / 2720 Msamples/s) and the path is queued at its BSDF sampling.
                    Rng rng = {0u, 0u, 0u, 0u, 0u, 0u};
                    v3 L = V3(0.f, 0.f, 0.f);
                    bool hit = false;
                    float val = 0.f;
                    Shade vs;
                    vs.pt = L; vs.wo = L; vs.gradient = L; vs.color[0] = vs.color[1] = vs.color[2] = vs.color[3] = 0.f; vs.Pbrdf = 0.f; vs.st = 0;
                    Nee ne;
                    ne.wi = L; ne.B = L; ne.pdf = 1.f; ne.light = 0u; ne.have = false;
#if SVR_PROF
                    unsigned long long* c_prof = w.counters + CNT_N;
#endif
                    PROF_BEGIN(pa, PH_PRIMARY);
#if SVR_PROF
                    const uint32_t prof_i0 = c.iters - c.iskip;
#endif
                    if (live) {
                        uint32_t x = w.x0 + px, y = owned_row_to_y(w, r);
                        hit = trace_primary<LAYOUT, COUNT, SKIP>(s, lds, x, y, wang_hash(w.frame0 + slot), group_march, P2, &gmaps[wave][0], c, rng, L, vs.pt, vs.wo, val, cold_scene());
                    }
                    PROF_END(pa, (uint32_t)__popcll(__ballot(live)));
#if SVR_PROF
                    if (COUNT && DEPTH1) {
                        // counting experiment builds: executed iterations of the primary walks, and 64 x the longest of the wave
                        const uint32_t d = (c.iters - c.iskip) - prof_i0;
                        uint32_t m = d;
                        for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o));
                        const unsigned long long sum = wave_sum(d);
                        if (lane == 0) { atomicAdd(&c_prof[2 * PH_N], (unsigned long long)m * 64ull); atomicAdd(&c_prof[2 * PH_N + 1], sum); }
                    }
#endif
                    // (deeper paths through a medium without exactly transparent space are queued unshaded: every walk is long
                    // there and the hits of a task lie far apart in time; c3n depth 4: 606 against 558 Msamples/s)
                    const bool shade_here = DEPTH1 || !s.bound_cull;
                    {
                        const uint64_t mh = __ballot(hit);
                        if (shade_here && mh != 0ull) {
                            PROF_BEGIN(psh, PH_SHADE);
                            if (hit) {
                                const DevScene* scp = cold_scene();
                                shade_event<LAYOUT, COUNT, LDSL>(s, vs, val, rng, ne, c, scp, lts);
                                if constexpr (!DEPTH1) {
                                    if (ne.have) {
                                        // estimate_direct_light, pathtracer.cu:191-198; the draws of sample_bsdf follow the
                                        // shadow walk, so it consumes every draw up to the box exit
                                        float sMin = (float)1e-6, sMax = SVR_FLT_MAX, sval = 0.f;
                                        const float ts = walk<LAYOUT, COUNT, SKIP, SVR_SHADOW_REMARCH>(s, lds, vs.pt, ne.wi, rng, sMin, sMax, sval, true, c);
                                        const float Tr = ((ts > sMin) && (ts < sMax)) ? 0.f : 1.f;          // transmittance.h:15-16
                                        const float kf = Tr * (float)s.num_lights;
                                        const DevLight& l = LDSL ? lts[ne.light] : s.lights[ne.light];
                                        L = L + V3(1.f, 1.f, 1.f) * (((ne.B * kf) * V3(l.radiance[0], l.radiance[1], l.radiance[2])) / ne.pdf);
                                    }
                                }
                            }
                            PROF_END(psh, (uint32_t)__popcll(mh));
                        }
                    }
                    bool over;
                    if constexpr (DEPTH1) {
                        queue_push_c1(Q, qC, hit && ne.have, vs.pt, ne, rng, (npend << 6) | lane);
                        over = !(hit && ne.have);                     // the camera ray's radiance, or a first event no light sample reaches (L = 0)
                    } else {
                        if (shade_here) queue_push_a(Q, qA, hit, vs, L, rng, (npend << 6) | lane);
                        else queue_push_b(Q, qB, hit, vs.pt, vs.wo, val, rng, (npend << 6) | lane);
                        over = !hit;
                    }
                    if (over) {
                        if constexpr (DIRECT) {
                            if (live) {
                                float* o = w.lbuf + (size_t)slot * w.slot_stride + 3 * ((size_t)owned_row_to_y(w, r) * s.imageW + (w.x0 + px));
                                o[0] = L.x; o[1] = L.y; o[2] = L.z;
                            }
                        } else {
                            float* o = gpend + (size_t)npend * (3u * 64u) + lane;
                            o[0] = L.x; o[64] = L.y; o[128] = L.z;
                        }
                    }
                    if (lane == 0) pend.task[wave][npend] = k;
                    if (++npend == QUEUE_TASKS || qC + qA + qB + 64u > QUEUE_CAP) flush();
                    continue;
                } else {
                    v3 L = V3(0.f, 0.f, 0.f);
                    if (live) {
                        uint32_t x = w.x0 + px, y = owned_row_to_y(w, r);
                        L = trace_path_tile<LAYOUT, COUNT, SKIP, DEPTH1>(s, lds, x, y, w.traceDepth, wang_hash(w.frame0 + slot), w.debug_stop, group_march, P2,
                                                                         &gmaps[wave][0], c);
                        if (!fold) {
                            float* o = w.lbuf + (size_t)slot * w.slot_stride + 3 * ((size_t)y * s.imageW + x);
                            o[0] = L.x; o[1] = L.y; o[2] = L.z;
                        }
                    }
                    if (fold) {
                        float* pl_row = &pend.L[wave][npend * 3u][lane];
                        pl_row[0] = L.x; pl_row[PEND_ROW] = L.y; pl_row[2u * PEND_ROW] = L.z;
                        if (lane == 0) pend.task[wave][npend] = k;
                        if (++npend == PEND_TASKS) flush();
                    }
                }
            }
        }
    }
    if (npend) flush();
    if (COUNT) cnt_flush(w, c);
}

template <int LAYOUT, bool COUNT>
static hipError_t launch_tile_t(const DevScene& s, const DevWork& w, const LaunchCfg& cfg, hipStream_t st)
{
    uint32_t wv = w.x1 - w.x0;
    if (wv == 0 || w.n_rows == 0) return hipSuccess;
    // frames per wave: the largest power of two <= min(nframes, 64), unless overridden
    uint32_t fl2 = 0;
    while (fl2 < 6u && (2u << fl2) <= w.nframes) ++fl2;
    if (cfg.frames_log2 >= 0 && (uint32_t)cfg.frames_log2 < fl2) fl2 = (uint32_t)cfg.frames_log2;
    if (w.fold) {
        // in-kernel accumulation: every frame of a pixel must sit in ONE wave (frame lanes 0 .. nframes-1 of its group)
        if (w.nframes > 64u) return hipErrorInvalidValue;
        fl2 = 0;
        while ((1u << fl2) < w.nframes) ++fl2;
    }
    const uint32_t P2 = 6u - fl2, tw2 = (P2 + 1u) >> 1, th2 = P2 >> 1;
    const uint32_t fgroups = (w.nframes + (1u << fl2) - 1u) >> fl2;
    uint32_t n_tasks = ((wv + (1u << tw2) - 1u) >> tw2) * ((w.n_rows + (1u << th2) - 1u) >> th2) * fgroups;
    constexpr uint32_t WPB = SVR_TILE_THREADS / 64;                       // waves per block
    uint32_t max_blocks = (uint32_t)(cfg.num_cus * cfg.blocks_per_cu) * 4u / WPB;
    uint32_t need = (n_tasks + WPB - 1u) / WPB;
    uint32_t blocks = need < max_blocks ? need : max_blocks;
    if (blocks == 0) blocks = 1;
    // ticket unit: ONE task.  Task costs differ by two orders of magnitude (a block of skipped rays vs a block of
    // grazing rays), so coarser units leave most waves idle behind the last heavy unit: measured 0.183 / 0.189 /
    // 0.208 / 0.230 / 0.279 ms per frame at 1 / 2 / 4 / 8 / 16 tasks per ticket (8-frame groups).  With 8 ticket
    // shards the atomics are not a limit (~130 k per launch).
    DevWork w2 = w;
    uint32_t unit = 1u;
    if (cfg.unit_override > 0) unit = (uint32_t)cfg.unit_override;
    w2.unit = unit;
    w2.frames_log2 = fl2;
    hipError_t e = hipMemsetAsync(w.ticket, 0, sizeof(uint32_t) * TICKET_SHARDS * TICKET_STRIDE, st);
    if (e != hipSuccess) return e;
    const bool skip = s.empty_mask != nullptr, d1 = w.traceDepth == 1u;
    // QUEUE builds need the per-wave record queues (DevWork.queue, sized for `queue_blocks` blocks) and exist for the BRICK layout only
    // (launches that do not fold -- frames traced ahead -- have the queue builds in their DIRECT form, which is not built with counters)
    const bool queue = LAYOUT != LAYOUT_LINEAR && w.queue != nullptr && w.pend != nullptr && blocks <= w.queue_blocks && w.traceDepth < 32768u &&   // a record's bounce counter has 15 bits
                       (w.fold || !COUNT);
#define SVR_LAUNCH_TILE(SK, D1, QU) hipLaunchKernelGGL((k_trace_tile<LAYOUT, COUNT, SK, D1, QU>), dim3(blocks), dim3(SVR_TILE_THREADS), 0, st, s, w2)
    if constexpr (LAYOUT != LAYOUT_LINEAR && !COUNT) {
        if (queue && !w.fold) {
#define SVR_LAUNCH_DIRECT(SK, D1, PO) hipLaunchKernelGGL((k_trace_tile<LAYOUT, false, SK, D1, true, PO, true>), dim3(blocks), dim3(SVR_TILE_THREADS), 0, st, s, w2)
            if (skip && cfg.pool_primary) { if (d1) SVR_LAUNCH_DIRECT(true, true, true); else SVR_LAUNCH_DIRECT(true, false, true); }
            else if (skip && d1) SVR_LAUNCH_DIRECT(true, true, false);
            else if (skip) SVR_LAUNCH_DIRECT(true, false, false);
            else if (d1) SVR_LAUNCH_DIRECT(false, true, false);
            else SVR_LAUNCH_DIRECT(false, false, false);
#undef SVR_LAUNCH_DIRECT
            return hipGetLastError();
        }
    }
    if constexpr (LAYOUT != LAYOUT_LINEAR) {
        if (queue && skip && cfg.pool_primary) {
            if (d1) hipLaunchKernelGGL((k_trace_tile<LAYOUT, COUNT, true, true, true, true>), dim3(blocks), dim3(SVR_TILE_THREADS), 0, st, s, w2);
            else hipLaunchKernelGGL((k_trace_tile<LAYOUT, COUNT, true, false, true, true>), dim3(blocks), dim3(SVR_TILE_THREADS), 0, st, s, w2);
            return hipGetLastError();
        }
        if (queue) {
            if (skip && d1) SVR_LAUNCH_TILE(true, true, true);
            else if (skip) SVR_LAUNCH_TILE(true, false, true);
            else if (d1) SVR_LAUNCH_TILE(false, true, true);
            else SVR_LAUNCH_TILE(false, false, true);
            return hipGetLastError();
        }
    }
    if (skip && d1) SVR_LAUNCH_TILE(true, true, false);
    else if (skip) SVR_LAUNCH_TILE(true, false, false);
    else if (d1) SVR_LAUNCH_TILE(false, true, false);
    else SVR_LAUNCH_TILE(false, false, false);
#undef SVR_LAUNCH_TILE
    return hipGetLastError();
}

hipError_t launch_trace_tile(const DevScene& s, const DevWork& w, const LaunchCfg& cfg, hipStream_t st)
{
    if (s.layout == LAYOUT_CELL)
        return cfg.count ? launch_tile_t<LAYOUT_CELL, true>(s, w, cfg, st) : launch_tile_t<LAYOUT_CELL, false>(s, w, cfg, st);
    if (s.layout == LAYOUT_PAIR)
        return cfg.count ? launch_tile_t<LAYOUT_PAIR, true>(s, w, cfg, st) : launch_tile_t<LAYOUT_PAIR, false>(s, w, cfg, st);
    if (s.layout == LAYOUT_LINEAR)
        return cfg.count ? launch_tile_t<LAYOUT_LINEAR, true>(s, w, cfg, st) : launch_tile_t<LAYOUT_LINEAR, false>(s, w, cfg, st);
    return cfg.count ? launch_tile_t<LAYOUT_BRICK, true>(s, w, cfg, st) : launch_tile_t<LAYOUT_BRICK, false>(s, w, cfg, st);
}

} // namespace svr
