// svr_trace_split.hip -- deeper paths (traceDepth >= 2), bit-exact, as TWO kernels with a register budget each.
//
// The queue builds of the tile kernel (svr_trace_tile.hip) trace the primary walks of a task, shade the first scatter events in place,
// run the first shadow walk in place, queue the paths, and after 32 tasks drain the queue with the lane machine (svr_lanes.hpp) -- ONE
// kernel, whose register allocation has to serve both halves: 128 VGPRs with 60 of them spilled and 168 bytes of scratch, 200 spilled
// SGPRs.  Compiled alone the machine needs 128 registers and spills none.  So:
//   k_split_front   the tile kernel's front half: per task the primary walks (shared whole-ray test, pixel group's occupancy map), the
//                   first scatter events shaded in place, their shadow walks in place (pathtracer.cu:216-257 for k = 0).  A path
//                   that goes on becomes an A record (or a B record: unshaded, media without exactly transparent space) in the
//                   wave's current CHUNK of a launch-wide pool; the radiance of a path that is over goes straight to its scratch slot;
//   k_split_machine persistent waves take chunks off one counter and drain them with the lane machine; a finished path writes its
//                   radiance to its scratch slot (the chunk's table says which: frame slot << 26 | pixel);
//   k_resolve       (svr_kernels.hip) folds the launch's slots into the accumulator in frame order and tone-maps.
// Every path executes the reference's operations (pathtracer.cu:216-277) in the reference's order on its own generator: scheduling only,
// bit-identical to the fused kernel and the oracle.
#include "svr_walk.hpp"
#include "svr_lanes.hpp"
#include "svr_tile_tasks.hpp"
#include "svr_primary.hpp"

namespace svr {

constexpr uint32_t SPLIT_THREADS = 1024;
static_assert(TILE_WAVES == SPLIT_THREADS / 64, "16 waves per block share the LDS image");
// a chunk: REC_WORDS x QUEUE_CAP record words (the A stack, then the B stack: the layout drain_queue expects), a table of QUEUE_CAP path
// ids, and two counts
SVR_DEV uint32_t* chunk_records(const DevWork& w, uint32_t chunk) { return w.queue + (size_t)chunk * SPLIT_CHUNK_WORDS; }
SVR_DEV uint32_t* chunk_gids(const DevWork& w, uint32_t chunk) { return chunk_records(w, chunk) + (size_t)REC_WORDS * QUEUE_CAP; }
SVR_DEV uint32_t* chunk_counts(const DevWork& w, uint32_t chunk) { return chunk_gids(w, chunk) + QUEUE_CAP; }

template <int LAYOUT, bool COUNT, bool SKIP>
__global__ __launch_bounds__(SPLIT_THREADS, 4) void k_split_front(const DevScene s, const DevWork w)
{
    using LDS = typename std::conditional<SKIP, LdsTileCull, LdsTileNoMask>::type;
    __shared__ LDS lds;
    __shared__ GroupMapShared gmaps[TILE_WAVES][GROUP_MAPS_PER_WAVE];
    // the front half is the tile kernel's traceDepth-1 front half: its set-up / shading constants through the laundered kernarg pointer and its
    // lights in LDS (svr_trace_tile.hip, cold_scene / lds_lights)
#ifndef SVR_SPLIT_COLD
#define SVR_SPLIT_COLD 1
#endif
    __shared__ DevLight lds_lights[8];
    if (threadIdx.x < 8u * (sizeof(DevLight) / 4u)) reinterpret_cast<float*>(lds_lights)[threadIdx.x] = reinterpret_cast<const float*>(s.lights)[threadIdx.x];
    const DevLight* const lts = lds_lights;
    auto cold_scene = [&]() -> const DevScene* {
#if SVR_SPLIT_COLD
        auto p = __builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(p));
        return (const DevScene*)p;
#else
        return nullptr;
#endif
    };
    lds_tile_load(lds, s, SKIP);

    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const TaskShape ts = task_shape(w);
    const uint32_t fl2 = ts.fl2, P2 = ts.P2, tw2 = ts.tw2, th2 = ts.th2, wv = ts.wv;
    const uint32_t n_tasks = ts.tiles_x * ts.tiles_y * ts.fgroups;
    const uint32_t shard0 = blockIdx.x % TICKET_SHARDS;
    Cnt c = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t* const chunk_counter = w.ticket + TICKET_SHARDS * TICKET_STRIDE;          // chunks handed out so far (the launch zeroes it)
    const bool shade_here = !s.bound_cull;                // (media without exactly transparent space are queued unshaded: svr_trace_tile.hip)
    uint32_t cur = 0xffffffffu, nA = 0u, nB = 0u;         // the wave's current chunk and the records in it
    LaneQueue Q;
    Q.cap = QUEUE_CAP;
    Q.q = nullptr;
    auto publish = [&]() {
        if (cur != 0xffffffffu && lane == 0u) { uint32_t* cc = chunk_counts(w, cur); cc[0] = nA; cc[1] = nB; }
        cur = 0xffffffffu; nA = nB = 0u;
    };
    for (uint32_t si = 0; si < TICKET_SHARDS; ++si) {
        const uint32_t shard = (shard0 + si) % TICKET_SHARDS;
        uint32_t* ticket = w.ticket + shard * TICKET_STRIDE;
        for (;;) {
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
            const bool group_march = SKIP && fl2 >= 3u && __ballot(live) == ~0ull;
            Rng rng = {0u, 0u, 0u, 0u, 0u, 0u};
            v3 L = V3(0.f, 0.f, 0.f);
            bool hit = false;
            float val = 0.f;
            Shade vs;
            vs.pt = L; vs.wo = L; vs.gradient = L; vs.color[0] = vs.color[1] = vs.color[2] = vs.color[3] = 0.f; vs.Pbrdf = 0.f; vs.st = 0;
            Nee ne;
            ne.wi = L; ne.B = L; ne.pdf = 1.f; ne.light = 0u; ne.have = false;
            uint32_t pix = 0u;
            if (live) {
                const uint32_t x = w.x0 + px, y = owned_row_to_y(w, r);
                pix = y * s.imageW + x;
                hit = trace_primary<LAYOUT, COUNT, SKIP>(s, lds, x, y, wang_hash(w.frame0 + slot), group_march, P2, &gmaps[wave][0], c, rng, L, vs.pt, vs.wo, val, cold_scene());
            }
            // the first scatter events are shaded in place and their shadow walks run in place (the lanes are frames of the same pixels).  Handing the
            // shaded events to the machine for their shadow walks instead (S records: an A record + the prepared estimate) was measured and lost:
            // c3 depth 2 / 4 3 776 / 2 427 against 4 750 / 3 029 Msamples/s (profiles/r04_notes_experiments.txt)
            if (shade_here && __ballot(hit) != 0ull) {
                if (hit) {
                    shade_event<LAYOUT, COUNT, SVR_SPLIT_COLD != 0>(s, vs, val, rng, ne, c, cold_scene(), lts);
                    if (ne.have) {
                        // estimate_direct_light, pathtracer.cu:191-198; the draws of sample_bsdf follow the shadow walk, so it consumes every draw up to the box exit
                        float sMin = (float)1e-6, sMax = SVR_FLT_MAX, sval = 0.f;
                        const float tsh = walk<LAYOUT, COUNT, SKIP, SVR_SHADOW_REMARCH>(s, lds, vs.pt, ne.wi, rng, sMin, sMax, sval, true, c);
                        const float Tr = ((tsh > sMin) && (tsh < sMax)) ? 0.f : 1.f;          // transmittance.h:15-16
                        const float kf = Tr * (float)s.num_lights;
                        const DevLight& l = SVR_SPLIT_COLD ? lts[ne.light] : s.lights[ne.light];
                        L = L + V3(1.f, 1.f, 1.f) * (((ne.B * kf) * V3(l.radiance[0], l.radiance[1], l.radiance[2])) / ne.pdf);
                    }
                }
            }
            // the paths that go on: records of the wave's current chunk (a fresh one when this task might not fit)
            const uint64_t mh = __ballot(hit);
            if (mh != 0ull) {
                if (cur != 0xffffffffu && nA + nB + 64u > QUEUE_CAP) publish();
                if (cur == 0xffffffffu) {
                    uint32_t ch = 0u;
                    if (lane == 0u) ch = atomicAdd(chunk_counter, 1u);
                    cur = __builtin_amdgcn_readfirstlane(ch);
                    Q.q = chunk_records(w, cur);
                }
                const uint32_t id = nA + nB + lane_rank(mh);
                if (hit) chunk_gids(w, cur)[id] = (slot << 26) | pix;
                if (shade_here) queue_push_a(Q, nA, hit, vs, L, rng, id);
                else queue_push_b(Q, nB, hit, vs.pt, vs.wo, val, rng, id);
            }
            if (live && !hit) {
                float* o = w.lbuf + (size_t)slot * w.slot_stride + 3 * (size_t)pix;
                o[0] = L.x; o[1] = L.y; o[2] = L.z;
            }
        }
    }
    publish();
    if (COUNT) cnt_flush(w, c);
}

template <int LAYOUT, bool COUNT, bool SKIP>
__global__ __launch_bounds__(SPLIT_THREADS, 4) void k_split_machine(const DevScene s, const DevWork w)
{
    using LDS = typename std::conditional<SKIP, LdsTileCull, LdsTileNoMask>::type;
    __shared__ LDS lds;
    // the machine, too, settles and shades with its lights in LDS and its constants through the laundered kernarg pointer: same-box A/B depth 2 / 3 / 4
    // 5 025 / 3 745 / 3 164 against 4 931 / 3 655 / 3 088 (LDS lights alone: 4 940 / 3 669 / 3 101).  (The FUSED deeper kernel loses with either: svr_trace_tile.hip.)
#ifndef SVR_SPLIT_MACHINE_COLD
#define SVR_SPLIT_MACHINE_COLD 1
#endif
#ifndef SVR_SPLIT_MACHINE_LDSL
#define SVR_SPLIT_MACHINE_LDSL 1
#endif
    __shared__ DevLight lds_lights[8];
    if (threadIdx.x < 8u * (sizeof(DevLight) / 4u)) reinterpret_cast<float*>(lds_lights)[threadIdx.x] = reinterpret_cast<const float*>(s.lights)[threadIdx.x];
    const DevLight* const lts = lds_lights;
    auto cold_scene = [&]() -> const DevScene* {
#if SVR_SPLIT_MACHINE_COLD
        auto p = __builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(p));
        return (const DevScene*)p;
#else
        return nullptr;
#endif
    };
    lds_tile_load(lds, s, SKIP);
    const uint32_t lane = threadIdx.x & 63u;
    Cnt c = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t* const counters = w.ticket + TICKET_SHARDS * TICKET_STRIDE;                // [0] chunks the front kernel filled, [1] chunks taken here
    const uint32_t n_chunks = __atomic_load_n(counters, __ATOMIC_RELAXED);
    for (;;) {
        uint32_t u = 0;
        if (lane == 0) u = atomicAdd(counters + 1, 1u);
        const uint32_t chunk = __builtin_amdgcn_readfirstlane(u);
        if (chunk >= n_chunks) break;
        LaneQueue Q;
        Q.cap = QUEUE_CAP;
        Q.q = chunk_records(w, chunk);
        const uint32_t* cc = chunk_counts(w, chunk);
        const uint32_t nA = __builtin_amdgcn_readfirstlane(cc[0]), nB = __builtin_amdgcn_readfirstlane(cc[1]);
        if (COUNT) c.loops += (lane == 0);
        drain_queue<LAYOUT, COUNT, SKIP, false, LDS, false, false, 2, SVR_SPLIT_MACHINE_LDSL != 0>(s, lds, Q, 0u, nA, nB, w.traceDepth, nullptr, 64u, c, w.counters + CNT_N, false, nullptr, &w, chunk_gids(w, chunk),
                                                                                                     cold_scene(), lts);
        // (records a later chunk's machine writes must not be read by this wave's earlier loads: the queue memory is per chunk, nothing to fence)
    }
    if (COUNT) cnt_flush(w, c);
}

template <int LAYOUT, bool COUNT>
static hipError_t launch_split_t(const DevScene& s, const DevWork& w, const LaunchCfg& cfg, hipStream_t st)
{
    const uint32_t wv = w.x1 - w.x0;
    if (wv == 0 || w.n_rows == 0) return hipSuccess;
    if (w.nframes == 0u || w.nframes > 64u || w.traceDepth < 2u || w.traceDepth >= 32768u || w.queue == nullptr || w.lbuf == nullptr) return hipErrorInvalidValue;
    if ((uint64_t)s.imageW * s.imageH > (1ull << 26)) return hipErrorInvalidValue;         // a path id holds the pixel index in 26 bits
    uint32_t fl2 = 0;
    while ((1u << fl2) < w.nframes) ++fl2;
    const uint32_t P2 = 6u - fl2, tw2 = (P2 + 1u) >> 1, th2 = P2 >> 1;
    const uint32_t n_tasks = ((wv + (1u << tw2) - 1u) >> tw2) * ((w.n_rows + (1u << th2) - 1u) >> th2);
    constexpr uint32_t WPB = SPLIT_THREADS / 64;
    const uint32_t max_blocks = (uint32_t)(cfg.num_cus * cfg.blocks_per_cu) * 4u / WPB;
    uint32_t blocks = (n_tasks + WPB - 1u) / WPB;
    if (blocks > max_blocks) blocks = max_blocks;
    if (blocks == 0) blocks = 1;
    // worst case: every path goes on; a wave opens a new chunk when fewer than 64 record slots are left, and leaves one partly filled behind
    const uint64_t worst = ((uint64_t)n_tasks * 64u + (QUEUE_CAP - 64u) - 1u) / (QUEUE_CAP - 64u) + (uint64_t)blocks * WPB;
    if (worst > w.queue_blocks) return hipErrorInvalidValue;                               // (queue_blocks = chunks the pool holds)
    DevWork w2 = w;
    w2.unit = 1u;
    w2.frames_log2 = fl2;
    hipError_t e = hipMemsetAsync(w.ticket, 0, sizeof(uint32_t) * (TICKET_SHARDS * TICKET_STRIDE + 2), st);
    if (e != hipSuccess) return e;
    const bool skip = s.empty_mask != nullptr;
    if (skip) hipLaunchKernelGGL((k_split_front<LAYOUT, COUNT, true>), dim3(blocks), dim3(SPLIT_THREADS), 0, st, s, w2);
    else hipLaunchKernelGGL((k_split_front<LAYOUT, COUNT, false>), dim3(blocks), dim3(SPLIT_THREADS), 0, st, s, w2);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (skip) hipLaunchKernelGGL((k_split_machine<LAYOUT, COUNT, true>), dim3(max_blocks), dim3(SPLIT_THREADS), 0, st, s, w2);
    else hipLaunchKernelGGL((k_split_machine<LAYOUT, COUNT, false>), dim3(max_blocks), dim3(SPLIT_THREADS), 0, st, s, w2);
    return hipGetLastError();
}

// w.queue = the chunk pool (w.queue_blocks chunks of SPLIT_CHUNK_WORDS words), w.lbuf = the launch's scratch slots, w.ticket = TICKET_SHARDS
// sharded task counters + 2 words; the caller folds the slots afterwards (launch_resolve)
hipError_t launch_trace_split(const DevScene& s, const DevWork& w, const LaunchCfg& cfg, hipStream_t st)
{
    if (s.layout == LAYOUT_CELL) return cfg.count ? launch_split_t<LAYOUT_CELL, true>(s, w, cfg, st) : launch_split_t<LAYOUT_CELL, false>(s, w, cfg, st);
    if (s.layout == LAYOUT_PAIR) return cfg.count ? launch_split_t<LAYOUT_PAIR, true>(s, w, cfg, st) : launch_split_t<LAYOUT_PAIR, false>(s, w, cfg, st);
    if (s.layout == LAYOUT_BRICK) return cfg.count ? launch_split_t<LAYOUT_BRICK, true>(s, w, cfg, st) : launch_split_t<LAYOUT_BRICK, false>(s, w, cfg, st);
    return hipErrorInvalidValue;
}

// chunks a launch of n_tasks tasks on `blocks` blocks can need at most
uint64_t split_chunks_worst_case(uint64_t n_paths_padded, uint32_t waves)
{
    return (n_paths_padded + (QUEUE_CAP - 64u) - 1u) / (QUEUE_CAP - 64u) + waves;
}

} // namespace svr
