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
#include "svr_walk.hpp"
#include "svr_lanes.hpp"
#include "svr_tile_tasks.hpp"

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

// sample_distance (woodcock_tracking.h:20-51) against the local majorants, from ray parameter t0 (>= the reference's
// tMin) to tMax: the collision's t, or -FLT_MAX.  val = the intensity fetched at the collision (pathtracer.cu:241).
template <int LAYOUT, bool COUNT, typename LDS>
SVR_DEV float walk_lm(const DevScene& s, const LDS& L, v3 orig, v3 dir, Rng& rng, float t0, float tMax, float& val, Cnt& c)
{
    const float INF = u2f(SVR_INF_BITS);
    // half-resolution macro grid: h(t) = A + B t per axis, cell = floor(h)
    const float Ax = 0.5f * fma_(orig.x - s.vmin[0], s.mc_scale[0], s.mc_off), Bx = 0.5f * (dir.x * s.mc_scale[0]);
    const float Ay = 0.5f * fma_(orig.y - s.vmin[1], s.mc_scale[1], s.mc_off), By = 0.5f * (dir.y * s.mc_scale[1]);
    const float Az = 0.5f * fma_(orig.z - s.vmin[2], s.mc_scale[2], s.mc_off), Bz = 0.5f * (dir.z * s.mc_scale[2]);
    const int hgx = s.mc_hgx, hgy = (s.mc_gy + 1) >> 1, hgz = (s.mc_gz + 1) >> 1;
    float t = t0;
    int ix = min(max((int)__builtin_floorf(fma_(Bx, t, Ax)), 0), hgx - 1);
    int iy = min(max((int)__builtin_floorf(fma_(By, t, Ay)), 0), hgy - 1);
    int iz = min(max((int)__builtin_floorf(fma_(Bz, t, Az)), 0), hgz - 1);
    // ray parameter of the next cell boundary per axis, and between boundaries
    const float rx = __builtin_amdgcn_rcpf(Bx), ry = __builtin_amdgcn_rcpf(By), rz = __builtin_amdgcn_rcpf(Bz);
    float tx = Bx != 0.f ? fmax_(((float)(ix + (Bx > 0.f ? 1 : 0)) - Ax) * rx, t) : INF;
    float ty = By != 0.f ? fmax_(((float)(iy + (By > 0.f ? 1 : 0)) - Ay) * ry, t) : INF;
    float tz = Bz != 0.f ? fmax_(((float)(iz + (Bz > 0.f ? 1 : 0)) - Az) * rz, t) : INF;
    const float dtx = __builtin_fabsf(rx), dty = __builtin_fabsf(ry), dtz = __builtin_fabsf(rz);
    const int sx = Bx > 0.f ? 1 : -1, sy = By > 0.f ? 1 : -1, sz = Bz > 0.f ? 1 : -1;
    const int qsy = sy * hgx, qsz = sz * (int)s.mc_hgxy;
    int q = ix + iy * hgx + iz * (int)s.mc_hgxy;
    // free path at the global majorant (woodcock_tracking.h:34); a cell of bound b spends it at rate b
    float S = -logf_unit(1.f - rng_uniform(rng)) * s.invSigmaMaxSI;
    float b = 1.f, rb = 1.f;
    for (uint32_t guard = 0; guard < SVR_WALK_GUARD; ++guard) {               // (hang guard, as in svr_walk.hpp: unreachable for sane scenes)
        // ---- DDA: on to this walk's next tentative collision (all lanes of the wave, each through its own cells) ----
        bool found = false;
        #pragma nounroll
        for (;;) {
            const uint32_t cl = (L.cls[(uint32_t)q >> 3] >> (((uint32_t)q & 7u) << 2)) & 15u;
            const float te = fmin_(fmin_(tx, ty), fmin_(tz, tMax));
            if (COUNT) c.ipre++;
            if (cl != 0u) {
                class_bound(cl, b, rb);
                const float room = (te - t) * b;
                if (S <= room) { t = fma_(S, rb, t); found = true; break; }
                S -= room;
            }
            if (te >= tMax) break;                                        // the walk leaves the box
            t = te;
            if (tx <= ty && tx <= tz) { ix += sx; q += sx; tx += dtx; if ((uint32_t)ix >= (uint32_t)hgx) break; }
            else if (ty <= tz) { iy += sy; q += qsy; ty += dty; if ((uint32_t)iy >= (uint32_t)hgy) break; }
            else { iz += sz; q += qsz; tz += dtz; if ((uint32_t)iz >= (uint32_t)hgz) break; }
        }
        if (!found) return -SVR_FLT_MAX;
        // ---- tentative collision (the lanes that found one, together): full-resolution `empty` bit, else fetch + accept test ----
        if (COUNT) c.iters++;
        const Cell cell = cell_of(s, orig + dir * t);
        bool fetch = true;
        if (s.has_empty) fetch = !cell_is_empty<false>(L, s, cell);
        if (fetch) {
            if (COUNT) { c.exec++; c.taps++; }
            val = tex_fetch<LAYOUT>(s, cell) * s.densityScale;
            const float ratio = alpha_of(L, s, val) * s.invSigmaMax;      // the reference's acceptance probability (woodcock_tracking.h:43) ...
            if (rng_uniform(rng) * b < ratio) return t;                  // ... over the local majorant's share b of sigma_max
        }
        S = -logf_unit(1.f - rng_uniform(rng)) * s.invSigmaMaxSI;
    }
    return -SVR_FLT_MAX;
}

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
    for (uint32_t k = 0; k < traceDepth; ++k) {
        float tMin = (float)1e-6, tMax = SVR_FLT_MAX, val = 0.f, t = -SVR_FLT_MAX;
        if (k == 0 && group_march) {
            // the wave is full and its lanes are pixels x frames: one shared whole-ray test per pixel (svr_walk.hpp).  No draw
            // has to be consumed for parity here, so a ray with nothing occupied ahead is simply over
            float t_occ;
            GroupMap map;
            map.g = gslot + ((threadIdx.x & 63u) & ((1u << P2) - 1u) & (GROUP_MAPS_PER_WAVE - 1u));
            const int r = walk_setup_group<true, true>(s, L_, P2, orig, dir, false, tMin, tMax, t_occ, map);
            if (r > 0 && t_occ != INF) t = walk_lm<LAYOUT, COUNT>(s, L_, orig, dir, rng, t_occ, tMax, val, c);
        } else {
            float tNear, tFar;
            if (volume_intersect(s, orig, dir, tNear, tFar)) {                // woodcock_tracking.h:22-27
                tMin = tNear < 0.f ? (float)1e-6 : tNear;
                tMax = tFar;
                t = walk_lm<LAYOUT, COUNT>(s, L_, orig, dir, rng, tMin, tMax, val, c);
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
        if (ne.have) {
            // transmittance (transmittance.h:10-17): a walk along the light direction to the box exit, 0 if it collides
            float sNear, sFar, sval = 0.f, ts = -SVR_FLT_MAX;
            float sMin = (float)1e-6, sMax = SVR_FLT_MAX;
            if (volume_intersect(s, vs.pt, ne.wi, sNear, sFar)) {
                sMin = sNear < 0.f ? (float)1e-6 : sNear;
                sMax = sFar;
                ts = walk_lm<LAYOUT, COUNT>(s, L_, vs.pt, ne.wi, rng, sMin, sMax, sval, c);
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

#ifndef SVR_LM_WAVES_PER_EU
#define SVR_LM_WAVES_PER_EU 4
#endif
constexpr uint32_t LM_THREADS = 1024;
static_assert(TILE_WAVES == LM_THREADS / 64, "the pending-radiance rows (svr_kernels.hpp) are sized for 16 waves per block");

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
    if (w.traceDepth == 1u) hipLaunchKernelGGL((k_trace_lm<LAYOUT, COUNT, true>), dim3(blocks), dim3(LM_THREADS), 0, st, s, w2);
    else hipLaunchKernelGGL((k_trace_lm<LAYOUT, COUNT, false>), dim3(blocks), dim3(LM_THREADS), 0, st, s, w2);
    return hipGetLastError();
}

// needs the acceleration data (s.empty_mask with the class table) and whole-ray validity (s.ray_skip: the clipped box lies
// inside the texture domain, so every point of a walk maps into the macro grid); the caller checks both
hipError_t launch_trace_lm(const DevScene& s, const DevWork& w, const LaunchCfg& cfg, hipStream_t st)
{
    if (s.empty_mask == nullptr || !s.ray_skip) return hipErrorInvalidValue;
    if (s.layout == LAYOUT_PAIR) return cfg.count ? launch_lm_t<LAYOUT_PAIR, true>(s, w, cfg, st) : launch_lm_t<LAYOUT_PAIR, false>(s, w, cfg, st);
    if (s.layout == LAYOUT_BRICK) return cfg.count ? launch_lm_t<LAYOUT_BRICK, true>(s, w, cfg, st) : launch_lm_t<LAYOUT_BRICK, false>(s, w, cfg, st);
    return cfg.count ? launch_lm_t<LAYOUT_LINEAR, true>(s, w, cfg, st) : launch_lm_t<LAYOUT_LINEAR, false>(s, w, cfg, st);
}

} // namespace svr
