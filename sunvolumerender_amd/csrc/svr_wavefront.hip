// svr_wavefront.hip -- wavefront form of the path tracer: the stages of a path run as separate kernels
// over DENSE queues, with wave-level ballot + mbcnt prefix-sum compaction between them.
//
//   k_wf_gen    one lane per (pixel, frame): RNG init, camera ray, light hit test, box intersection and the
//               conservative whole-ray test (pathtracer.cu:205-215 + the set-up of sample_distance).  Paths
//               whose primary walk cannot collide are finished here; the others are appended to a ray queue.
//   k_wf_walk   one lane per queued ray: the Woodcock walk (woodcock_tracking.h:32-45).  Misses / light
//               hits are finished; collisions are appended to a hit queue.
//   k_wf_shade  one lane per queued hit: VolumeSample (6 gradient taps), shading decision, next-event
//               estimation with its shadow walk, then either the path's radiance or -- for deeper
//               bounces -- BSDF sampling, roulette and a continuation ray appended to the next ray queue
//               (pathtracer.cu:237-276).
//
// Why: in the one-kernel form only ~45 % of the lanes of a wave are active on average (rays of a tile end
// at different stages); here every stage starts with full waves, and each kernel keeps only its own
// stage's state in registers.  Records are appended in tile order, so neighbouring lanes still march
// through neighbouring voxels.  Every path executes exactly the arithmetic of the one-kernel form, so the
// radiance is bit-identical (the queues only change WHICH lane runs a path).
#include "svr_walk.hpp"

namespace svr {

constexpr int WF_PLANES = 6;
constexpr int WF_MAX_DEPTH = 15;
constexpr int WF_HIT_COUNT0 = 16;      // counts[k] = rays of bounce k, counts[16 + k] = hits of bounce k
constexpr int WF_HEAD0 = 32;           // counts[32 + k] = chunk ticket of the walk kernel of bounce k

struct DevQueues {
    float4* ray[2][WF_PLANES];         // ping-pong by bounce parity
    float4* hit[WF_PLANES];
    uint32_t* counts;
    uint32_t capacity;
};

struct PathRec {
    v3 a; float aw;        // ray: origin, tMin        | hit: point, intensity
    v3 b; float bw;        // ray: direction, tMax     | hit: wo, -
    Rng rng;
    uint32_t pix, meta;    // meta = slot | k << 8 | (ls_id + 1) << 16
    v3 T; float tw;        // ray: t_occ
    v3 L; float lw;        // ray: ls_t
};

SVR_DEV void rec_store(float4* const* q, uint32_t i, const PathRec& r)
{
    q[0][i] = make_float4(r.a.x, r.a.y, r.a.z, r.aw);
    q[1][i] = make_float4(r.b.x, r.b.y, r.b.z, r.bw);
    q[2][i] = make_float4(u2f(r.rng.v0), u2f(r.rng.v1), u2f(r.rng.v2), u2f(r.rng.v3));
    q[3][i] = make_float4(u2f(r.rng.v4), u2f(r.rng.d), u2f(r.pix), u2f(r.meta));
    q[4][i] = make_float4(r.T.x, r.T.y, r.T.z, r.tw);
    q[5][i] = make_float4(r.L.x, r.L.y, r.L.z, r.lw);
}

SVR_DEV PathRec rec_load(float4* const* q, uint32_t i)
{
    PathRec r;
    float4 v = q[0][i]; r.a = V3(v.x, v.y, v.z); r.aw = v.w;
    v = q[1][i]; r.b = V3(v.x, v.y, v.z); r.bw = v.w;
    v = q[2][i]; r.rng.v0 = f2u(v.x); r.rng.v1 = f2u(v.y); r.rng.v2 = f2u(v.z); r.rng.v3 = f2u(v.w);
    v = q[3][i]; r.rng.v4 = f2u(v.x); r.rng.d = f2u(v.y); r.pix = f2u(v.z); r.meta = f2u(v.w);
    v = q[4][i]; r.T = V3(v.x, v.y, v.z); r.tw = v.w;
    v = q[5][i]; r.L = V3(v.x, v.y, v.z); r.lw = v.w;
    return r;
}

// append the records of the lanes with `push` set: one atomic per wave, slots by prefix sum over the ballot
SVR_DEV void queue_push(float4* const* q, uint32_t* count, uint32_t capacity, bool push, const PathRec& r)
{
    unsigned long long m = __ballot(push);
    if (m == 0ull) return;
    uint32_t base = 0;
    if ((threadIdx.x & 63u) == (uint32_t)__builtin_ctzll(m)) base = atomicAdd(count, (uint32_t)__popcll(m));
    base = __shfl(base, __builtin_ctzll(m), 64);
    uint32_t idx = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
    if (push && idx < capacity) rec_store(q, idx, r);
}

SVR_DEV void write_radiance(const DevWork& w, uint32_t pix, uint32_t slot, v3 L)
{
    float* o = w.lbuf + (size_t)slot * w.slot_stride + 3 * (size_t)pix;
    o[0] = L.x; o[1] = L.y; o[2] = L.z;
}

// a walk that did not collide: pathtracer.cu:220-235
SVR_DEV v3 finish_miss(const DevScene& s, uint32_t k, int ls_id, v3 dir, v3 T, v3 L)
{
    if (k == 0u && ls_id >= 0) {
        // t = FLT_MAX, ls.t < t always
        const DevLight& l = s.lights[ls_id];
        float cosTerm = dot(V3(l.normal[0], l.normal[1], l.normal[2]), -dir);
        return L + (T * V3(l.radiance[0], l.radiance[1], l.radiance[2])) * (cosTerm <= 0.f ? 0.f : 1.f);
    }
    if (s.env_on_escape) L = L + T * env_radiance(s, dir);
    return L;
}

// ---------------------------------------------------------------------------------------------------
template <bool COUNT, bool SKIP>
__global__ __launch_bounds__(256) void k_wf_gen(const DevScene s, const DevWork w, const DevQueues q)
{
    using LDS = typename std::conditional<SKIP, LdsTile, LdsTileNoMask>::type;
    __shared__ LDS lds;
    lds_tile_load(lds, s, SKIP);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wv = w.x1 - w.x0;
    const uint32_t tiles_x = (wv + 7u) >> 3;
    const uint32_t n_tasks = tiles_x * ((w.n_rows + 7u) >> 3) * w.nframes;
    const uint32_t n_waves = gridDim.x * (blockDim.x >> 6);
    Cnt c = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (uint32_t task = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); task < n_tasks; task += n_waves) {
        uint32_t tile = task / w.nframes, slot = task - tile * w.nframes;
        uint32_t ty = tile / tiles_x, tx = tile - ty * tiles_x;
        uint32_t px = (tx << 3) + (lane & 7u), r = (ty << 3) + (lane >> 3);
        bool push = false;
        PathRec rec;
        if (px < wv && r < w.n_rows) {
            uint32_t x = w.x0 + px, y = owned_row_to_y(w, r);
            uint32_t offset = y * s.imageW + x;
            rng_init(rec.rng, wang_hash(w.frame0 + slot) + offset);
            if (COUNT) c.paths++;
            v3 orig, dir;
            camera_ray(s, x, y, rec.rng, orig, dir);
            float ls_t;
            int ls_id = nearest_light(s, orig, dir, ls_t);
            float tMin = (float)1e-6, tMax = SVR_FLT_MAX, t_occ = 0.f;
            int st = walk_setup<COUNT, SKIP>(s, lds, orig, dir, false, tMin, tMax, t_occ);
            if (st <= 0) {
                write_radiance(w, offset, slot, finish_miss(s, 0u, ls_id, dir, V3(1.f, 1.f, 1.f), V3(0.f, 0.f, 0.f)));
            } else {
                push = true;
                rec.a = orig; rec.aw = tMin; rec.b = dir; rec.bw = tMax;
                rec.pix = offset; rec.meta = slot | ((uint32_t)(ls_id + 1) << 16);
                rec.T = V3(1.f, 1.f, 1.f); rec.tw = t_occ; rec.L = V3(0.f, 0.f, 0.f); rec.lw = ls_t;
            }
        }
        queue_push(q.ray[0], q.counts + 0, q.capacity, push, rec);
    }
    if (COUNT) cnt_flush(w, c);
}

// ---------------------------------------------------------------------------------------------------
#ifndef SVR_WF_WALK_WAVES_PER_EU
#define SVR_WF_WALK_WAVES_PER_EU 4
#endif
#ifndef SVR_WF_WALK_THREADS
#define SVR_WF_WALK_THREADS 256
#endif
constexpr uint32_t WF_CHUNK = 512;     // consecutive ray records a wave takes per ticket (8 tiles' worth: stays coherent)

// Woodcock walks with LANE REGENERATION.  Ray lengths vary from a handful to hundreds of iterations, so a
// wave that simply ran 64 records to completion kept 44 % of its lanes busy (rocprof: SQ_THREAD_CYCLES_VALU /
// (64 x SQ_INSTS_VALU)).  Here a wave owns a contiguous chunk of the queue and every lane is a slot: once
// `refill_min_idle` lanes have finished their ray, the finished lanes retire it (radiance or hit record,
// one ballot + one atomic for the whole wave) and take the next records of the chunk, assigned by a prefix
// sum over the ballot.  Records are in tile order, so the lanes of a wave keep marching through neighbouring
// voxels.  Each ray's arithmetic is unchanged -> bit-identical results.
template <int LAYOUT, bool COUNT, bool SKIP>
__global__ __launch_bounds__(SVR_WF_WALK_THREADS, SVR_WF_WALK_WAVES_PER_EU) void k_wf_walk(const DevScene s, const DevWork w, const DevQueues q, uint32_t bounce)
{
    uint32_t n = q.counts[bounce];
    n = n < q.capacity ? n : q.capacity;
    if (n == 0u) return;
    using LDS = typename std::conditional<SKIP, LdsTile, LdsTileNoMask>::type;
    __shared__ LDS lds;
    lds_tile_load(lds, s, SKIP);
    float4* const* rq = q.ray[bounce & 1u];
    uint32_t* head = q.counts + WF_HEAD0 + bounce;
    Cnt c = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t th = w.refill_min_idle < 1u ? 1u : (w.refill_min_idle > 64u ? 64u : w.refill_min_idle);

    uint32_t next = 0, end = 0;            // wave-uniform: unassigned records of the current chunk
    bool exhausted = false;
    // lane slot
    enum : uint32_t { EMPTY = 0, ACTIVE = 1, FINISHED = 2 };
    uint32_t state = EMPTY;
    PathRec rec;
    float t = 0.f, val = 0.f;
    uint32_t guard = 0;
    bool ray_skippable = false;

    for (;;) {
        unsigned long long m_act = __ballot(state == ACTIVE);
        uint32_t n_free = 64u - (uint32_t)__popcll(m_act);
        if (n_free >= th || m_act == 0ull) {
            // ---- retire finished rays: pathtracer.cu:220-235, or hand the collision to the shade stage ----
            bool push = false;
            if (state == FINISHED) {
                const uint32_t k = (rec.meta >> 8) & 0xffu, slot = rec.meta & 0xffu;
                const int ls_id = (int)((rec.meta >> 16) & 0xffu) - 1;
                bool done = false;
                v3 L = rec.L;
                if (k == 0u && ls_id >= 0) {
                    float tt = t < 0.f ? SVR_FLT_MAX : t;
                    if (rec.lw < tt) { L = finish_miss(s, 0u, ls_id, rec.b, rec.T, L); done = true; }
                }
                if (!done && t < 0.f) { L = finish_miss(s, 1u, -1, rec.b, rec.T, L); done = true; }
                if (done) write_radiance(w, rec.pix, slot, L);
                else {
                    push = true;
                    rec.a = rec.a + rec.b * t; rec.aw = val;    // ptInWorld, intensity (pathtracer.cu:240-241)
                    rec.b = -rec.b; rec.bw = 0.f;               // wo
                }
                state = EMPTY;
            }
            queue_push(q.hit, q.counts + WF_HIT_COUNT0 + bounce, q.capacity, push, rec);
            // ---- take the next records ----
            if (!exhausted && next == end) {
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(head, WF_CHUNK);
                base = __builtin_amdgcn_readfirstlane(base);
                if (base >= n) exhausted = true;
                else { next = base; end = min(base + WF_CHUNK, n); }
            }
            unsigned long long m_empty = __ballot(state == EMPTY);
            if (!exhausted) {
                uint32_t avail = end - next;
                uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m_empty >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m_empty, 0u));
                if (state == EMPTY && rank < avail) {
                    rec = rec_load(rq, next + rank);
                    t = rec.aw;                                // tMin
                    guard = 0;
                    ray_skippable = COUNT && SKIP && s.ray_skip && rec.tw == u2f(SVR_INF_BITS);
                    if (COUNT && ray_skippable) c.wskip++;
                    state = ACTIVE;
                }
                next += min((uint32_t)__popcll(m_empty), avail);
            } else if (m_act == 0ull) {
                break;                                          // nothing running, nothing left
            }
        }
        // ---- one Woodcock iteration per active lane: woodcock_tracking.h:32-45 ----
        if (state == ACTIVE) {
            if (COUNT) { c.iters++; if (ray_skippable) c.iskip++; else if (SKIP && t < rec.tw) c.ipre++; }
            t += -logf_unit(1.f - rng_uniform(rec.rng)) * s.invSigmaMaxSI;
            if (t > rec.bw || guard++ >= SVR_WALK_GUARD) { t = -SVR_FLT_MAX; state = FINISHED; }
            else {
                if (COUNT) c.taps++;
                float sigma_t = 0.f;
                if (!SKIP || t >= rec.tw) {
                    Cell cell = cell_of(s, rec.a + rec.b * t);
                    bool fetch = true;
                    if (SKIP) fetch = !cell_is_empty<false>(lds, s, cell);
                    if (fetch) {
                        if (COUNT) c.exec++;
                        val = tex_fetch<LAYOUT>(s, cell) * s.densityScale;
                        sigma_t = alpha_of(lds, s, val);
                    }
                }
                if (rng_uniform(rec.rng) < sigma_t * s.invSigmaMax) state = FINISHED;
            }
        }
    }
    if (COUNT) cnt_flush(w, c);
}

// ---------------------------------------------------------------------------------------------------
template <int LAYOUT, bool COUNT, bool SKIP>
__global__ __launch_bounds__(256) void k_wf_shade(const DevScene s, const DevWork w, const DevQueues q, uint32_t bounce)
{
    uint32_t n = q.counts[WF_HIT_COUNT0 + bounce];
    n = n < q.capacity ? n : q.capacity;
    if (blockIdx.x * blockDim.x >= n) return;
    using LDS = typename std::conditional<SKIP, LdsTile, LdsTileNoMask>::type;
    __shared__ LDS lds;
    lds_tile_load(lds, s, SKIP);
    float4* const* nq = q.ray[(bounce + 1u) & 1u];
    Cnt c = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const uint32_t stride = gridDim.x * blockDim.x;
    const uint32_t n_up = (n + 63u) & ~63u;
    const uint32_t traceDepth = w.traceDepth;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_up; i += stride) {
        bool push = false;
        PathRec rec;
        if (i < n) {
            rec = rec_load(q.hit, i);
            const uint32_t k = (rec.meta >> 8) & 0xffu, slot = rec.meta & 0xffu;
            Rng& rng = rec.rng;
            v3 L = rec.L, T = rec.T;
            // VolumeSample, pathtracer.cu:237-244
            Shade vs;
            if (COUNT) { c.scatter++; c.taps += 7; c.exec += 6; }
            vs.pt = rec.a;
            vs.wo = rec.b;
            tf_rgba(s, s.tf, rec.aw, vs.color);
            {
                v3 p = vs.pt;
                float xd = intensity_at<LAYOUT>(s, V3(p.x + s.spacing[0], p.y + 0.f, p.z + 0.f)) -
                           intensity_at<LAYOUT>(s, V3(p.x - s.spacing[0], p.y - 0.f, p.z - 0.f));
                float yd = intensity_at<LAYOUT>(s, V3(p.x + 0.f, p.y + s.spacing[1], p.z + 0.f)) -
                           intensity_at<LAYOUT>(s, V3(p.x - 0.f, p.y - s.spacing[1], p.z - 0.f));
                float zd = intensity_at<LAYOUT>(s, V3(p.x + 0.f, p.y + 0.f, p.z + s.spacing[2])) -
                           intensity_at<LAYOUT>(s, V3(p.x - 0.f, p.y - 0.f, p.z - s.spacing[2]));
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
                    float ts = walk<LAYOUT, COUNT, SKIP, false>(s, lds, vs.pt, wiL, rng, sMin, sMax, sval, k + 1u < traceDepth, c);
                    float Tr = ((ts > sMin) && (ts < sMax)) ? 0.f : 1.f;
                    float kf = Tr * (float)s.num_lights;
                    Ld = ((bsdf_eval(vs, wiL) * kf) * Li) / pdfL;
                }
            }
            L = L + T * Ld;
            bool done = k + 1u >= traceDepth;      // sample_bsdf / roulette of the last bounce cannot reach L
            if (!done) {
                v3 wi; float pdf = 0.f;
                v3 f = bsdf_sample(vs, wi, pdf, rng);
                float cosTerm = __builtin_fabsf(dot(normalize(vs.gradient), wi));
                if (fmax_(f.x, fmax_(f.y, f.z)) > 0.f && pdf > 0.f) {
                    if (vs.st == 0) T = T * (f / (pdf * (1.f - vs.Pbrdf)));
                    else T = T * ((f * cosTerm) / (pdf * vs.Pbrdf));
                }
                if (k >= 3u && russian_roulette(T, rng)) done = true;
                if (!done) {
                    float tMin = (float)1e-6, tMax = SVR_FLT_MAX, t_occ = 0.f;
                    int st = walk_setup<COUNT, SKIP>(s, lds, vs.pt, wi, false, tMin, tMax, t_occ);
                    if (st <= 0) { L = finish_miss(s, 1u, -1, wi, T, L); done = true; }
                    else {
                        push = true;
                        rec.a = vs.pt; rec.aw = tMin; rec.b = wi; rec.bw = tMax;
                        rec.meta = slot | ((k + 1u) << 8);
                        rec.T = T; rec.tw = t_occ; rec.L = L; rec.lw = 0.f;
                    }
                }
            }
            if (done) write_radiance(w, rec.pix, slot, L);
        }
        queue_push(nq, q.counts + bounce + 1u, q.capacity, push, rec);
    }
    if (COUNT) cnt_flush(w, c);
}

// ---------------------------------------------------------------------------------------------------
template <int LAYOUT, bool COUNT, bool SKIP>
static hipError_t launch_wf_t(const DevScene& s, const DevWork& w, const LaunchCfg& cfg, const DevQueues& q, hipStream_t st)
{
    uint32_t wv = w.x1 - w.x0;
    if (wv == 0 || w.n_rows == 0) return hipSuccess;
    uint32_t blocks = (uint32_t)(cfg.num_cus * cfg.blocks_per_cu);
    hipError_t e = hipMemsetAsync(q.counts, 0, sizeof(uint32_t) * 48, st);
    if (e != hipSuccess) return e;
    uint32_t n_tasks = ((wv + 7u) >> 3) * ((w.n_rows + 7u) >> 3) * w.nframes;
    uint32_t gen_blocks = (n_tasks + 3u) / 4u < blocks ? (n_tasks + 3u) / 4u : blocks;
    hipLaunchKernelGGL((k_wf_gen<COUNT, SKIP>), dim3(gen_blocks ? gen_blocks : 1), dim3(256), 0, st, s, w, q);
    for (uint32_t k = 0; k < w.traceDepth; ++k) {
        {
            // waves per CU of the walk kernel: blocks_per_cu x 4 waves for 256-thread blocks; scaled for larger blocks
            uint32_t wb = blocks * 256u / SVR_WF_WALK_THREADS * (SVR_WF_WALK_WAVES_PER_EU > 4 ? SVR_WF_WALK_WAVES_PER_EU : 4) / 4u;
            hipLaunchKernelGGL((k_wf_walk<LAYOUT, COUNT, SKIP>), dim3(wb ? wb : 1), dim3(SVR_WF_WALK_THREADS), 0, st, s, w, q, k);
        }
        hipLaunchKernelGGL((k_wf_shade<LAYOUT, COUNT, SKIP>), dim3(blocks), dim3(256), 0, st, s, w, q, k);
    }
    return hipGetLastError();
}

hipError_t launch_wavefront(const DevScene& s, const DevWork& w, const LaunchCfg& cfg, float4* const* planes,
                            uint32_t* counts, uint32_t capacity, hipStream_t st)
{
    if (w.traceDepth > (uint32_t)WF_MAX_DEPTH) return hipErrorInvalidValue;
    DevQueues q;
    for (int p = 0; p < WF_PLANES; ++p) {
        q.ray[0][p] = planes[p];
        q.ray[1][p] = planes[WF_PLANES + p];
        q.hit[p] = planes[2 * WF_PLANES + p];
    }
    q.counts = counts;
    q.capacity = capacity;
    const bool skip = s.empty_mask != nullptr;
#define SVR_WF_DISPATCH(LAY)                                                                       \
    if (cfg.count) return skip ? launch_wf_t<LAY, true, true>(s, w, cfg, q, st) : launch_wf_t<LAY, true, false>(s, w, cfg, q, st); \
    return skip ? launch_wf_t<LAY, false, true>(s, w, cfg, q, st) : launch_wf_t<LAY, false, false>(s, w, cfg, q, st);
    if (s.layout == LAYOUT_CELL) { SVR_WF_DISPATCH(LAYOUT_CELL) }
    if (s.layout == LAYOUT_PAIR) { SVR_WF_DISPATCH(LAYOUT_PAIR) }
    if (s.layout == LAYOUT_LINEAR) { SVR_WF_DISPATCH(LAYOUT_LINEAR) }
    SVR_WF_DISPATCH(LAYOUT_BRICK)
#undef SVR_WF_DISPATCH
}

} // namespace svr
