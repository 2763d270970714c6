// svr_kernels.hpp -- launch interface between the C-ABI layer (svr_api.hip) and the
// gfx950 kernels (svr_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include "svr_scene.hpp"

namespace svr {

enum { KERNEL_AUTO = 0, KERNEL_PIXEL = 1, KERNEL_TILE = 2, KERNEL_ULOOP = 3, KERNEL_WAVEFRONT = 4, KERNEL_ENV_NEE = 5 /* internal: SVR_OPT_ENV_NEE */ };
constexpr int WF_QUEUE_PLANES = 18;         // 2 ray queues + 1 hit queue, 6 float4 planes each
constexpr uint32_t TICKET_SHARDS = 8;       // sharded work counters (one per XCD group), 128 B apart
constexpr uint32_t TICKET_STRIDE = 32;      // in uint32 words
constexpr uint32_t MASK_WORDS_MAX = 8192;   // words of the LDS-resident `empty` bitmask (32 KiB): up to 64^3 macro-cells
constexpr uint32_t DIST_WORDS_MAX = 4096;   // words of the half-resolution 4-bit distance field (16 KiB): up to 32^3 coarse cells
constexpr int DIST_CAP = 15;
// QUEUE builds of the tile kernel (svr_trace_tile.hip, svr_lanes.hpp): per wave, up to QUEUE_CAP scatter records of REC_WORDS
// words and the radiance of QUEUE_TASKS tasks (3 channels x 64 lanes), in global memory; TILE_WAVES waves per block
constexpr uint32_t REC_C1_WORDS = 17;   // traceDepth 1, a shaded scatter event waiting for its shadow walk: pt(3) wi(3) B(3) pdf rng(6) id|light
constexpr uint32_t REC_A_WORDS = 26;    // deeper paths, after a next-event estimate: pt wo gradient colour(3 each) Pbrdf L(3) T(3) rng(6) meta
constexpr uint32_t REC_B_WORDS = 20;    // deeper paths, at a scatter point: pt(3) wo(3) val L(3) T(3) rng(6) meta
constexpr uint32_t REC_WORDS = REC_A_WORDS + REC_B_WORDS;   // words per record slot of a wave (the larger of the two uses)
#ifndef SVR_QUEUE_TASKS
#define SVR_QUEUE_TASKS 32
#endif
constexpr uint32_t QUEUE_TASKS = SVR_QUEUE_TASKS;
constexpr uint32_t QUEUE_CAP = 32 * QUEUE_TASKS;
constexpr uint32_t TILE_WAVES = 16;
constexpr size_t QUEUE_WORDS_PER_BLOCK = (size_t)REC_WORDS * QUEUE_CAP * TILE_WAVES;
constexpr size_t PEND_FLOATS_PER_BLOCK = (size_t)QUEUE_TASKS * 3 * 64 * TILE_WAVES;
// the split kernels of deeper paths (svr_trace_split.hip): a chunk of the launch-wide record pool = the records, a table of QUEUE_CAP path ids, two counts (+ padding)
constexpr size_t SPLIT_CHUNK_WORDS = (size_t)REC_WORDS * QUEUE_CAP + QUEUE_CAP + 32;
// Bound classes (svr_accel.hip, k_bound_class): 4 bits per half-resolution macro-cell = smallest class c whose threshold
// BOUND_THR(c) is >= (largest transfer-function alpha any fetch in the cell can return) x invSigmaMax.
constexpr uint32_t BOUND_CLASSES = 16;
constexpr uint32_t ACCEL_WORDS = DIST_WORDS_MAX + 2 * MASK_WORDS_MAX + DIST_WORDS_MAX + BOUND_CLASSES + 4;   // device buffer: dist | deep | empty | class | thresholds | census
constexpr uint32_t ACCEL_CLASS_OFF = DIST_WORDS_MAX + 2 * MASK_WORDS_MAX;
constexpr uint32_t ACCEL_THR_OFF = ACCEL_CLASS_OFF + DIST_WORDS_MAX;
constexpr uint32_t ACCEL_CENSUS_OFF = ACCEL_THR_OFF + BOUND_CLASSES;   // [0] coarse cells with a class in 1..14 (culling pays there), [1] cells of class 15, [2] `empty` macro-cells

// counter slots (unsigned long long each) -- order of svr_counters in include/svr_abi.h
enum { CNT_PATHS = 0, CNT_VOL_TAPS, CNT_WOODCOCK, CNT_SCATTER, CNT_SHADOW, CNT_RAYCAST, CNT_LOOP, CNT_TAPS_EXEC,
       CNT_WALKS_RAYSKIP, CNT_ITERS_RAYSKIP, CNT_ITERS_PREFIX, CNT_CULLED, CNT_N };

struct LaunchCfg {
    int kernel;            // KERNEL_*
    bool count;
    int num_cus;
    int blocks_per_cu;     // persistent kernel
    int unit_override;     // tile kernel: tasks per ticket (0 = heuristic)
    int frames_log2;       // tile kernel: log2(frames per wave), -1 = as many as the group allows (<= 8)
    bool lm_straight;      // local-majorant kernel: straight-line paths also where the pool form applies (cross-check)
    bool pool_primary;     // tile kernel, QUEUE builds at traceDepth 1: the primary walks go through the lane machine too (POOL)
};

// trace work.nframes paths per owned pixel into the scratch slots work.lbuf
hipError_t launch_pathtrace(const DevScene& scene, const DevWork& work, const LaunchCfg& cfg, hipStream_t stream);
// fold the scratch slots into the running mean (frame order) and, if work.img, tone-map
hipError_t launch_resolve(const DevScene& scene, const DevWork& work, hipStream_t stream);
// default trace kernel (svr_trace_tile.hip): persistent waves, one 8x8 tile-task per wave, empty-space skipping
hipError_t launch_trace_tile(const DevScene& scene, const DevWork& work, const LaunchCfg& cfg, hipStream_t stream);
// deeper paths as two kernels (svr_trace_split.hip): front half (primary walks, first events in place) -> chunks of records -> lane machine; radiance to the scratch slots
hipError_t launch_trace_split(const DevScene& scene, const DevWork& work, const LaunchCfg& cfg, hipStream_t stream);
uint64_t split_chunks_worst_case(uint64_t n_paths_padded, uint32_t waves);
// OPT-IN importance sampling of the environment map (svr_trace_env.hip): straight-line paths with one env sample per scatter event, into the scratch slots
hipError_t launch_trace_env(const DevScene& scene, const DevWork& work, const LaunchCfg& cfg, hipStream_t stream);
// its sampling table: cdf = h * (w + 1) + h + 1 floats, tmp = w * h + 1 floats of scratch
hipError_t launch_env_cdf(const float* env_rgba, int w, int h, float* cdf, float* tmp, hipStream_t stream);
// OPT-IN local-majorant kernel (svr_trace_lm.hip): needs scene.empty_mask (class table) and scene.ray_skip; folding launches need work.pend
hipError_t launch_trace_lm(const DevScene& scene, const DevWork& work, const LaunchCfg& cfg, hipStream_t stream);
// wavefront kernels (svr_wavefront.hip): gen -> [walk -> shade] x depth over dense queues with ballot/prefix-sum
// compaction; planes = WF_QUEUE_PLANES device arrays of `capacity` float4, counts = 32 words
hipError_t launch_wavefront(const DevScene& scene, const DevWork& work, const LaunchCfg& cfg, float4* const* planes,
                            uint32_t* counts, uint32_t capacity, hipStream_t stream);
// acceleration data (svr_accel.hip): per-macro-cell min/max of the raw voxels, and the empty bitmask
// pad: the footprint of every cell grown by that many voxels per side (1: the wide table of launch_bound8)
hipError_t launch_minmax(const uint16_t* src_linear, uint16_t* mm, int nx, int ny, int nz, int shift,
                         int gx, int gy, int gz, hipStream_t stream, int pad = 0);
// mask: [0, DIST_WORDS_MAX) = half-resolution 4-bit distance field (distance to the nearest non-empty macro-cell),
// then MASK_WORDS_MAX words of deep-empty bits (distance >= 2, full resolution), then mask_words words of `empty`
// bits; tmp: 2 * gx*gy*gz bytes of scratch
hipError_t launch_empty_mask(const uint16_t* mm, int gx, int gy, int gz, const uint32_t* tf_zero_prefix, int tf_n,
                             float densityScale, uint32_t* mask, uint32_t mask_words, uint8_t* tmp, hipStream_t stream);
// `empty` bits of a (fine) macro-cell level into plain global memory: mask[n_cells / 32]
hipError_t launch_fine_mask(const uint16_t* mm, uint32_t n_cells, const uint32_t* tf_zero_prefix, int tf_n, float densityScale,
                            uint32_t* mask, uint32_t words, hipStream_t stream);
// one byte per macro-cell: which of its 2 x 2 x 2 fine cells are NOT `empty` (fine_empty = the fine level's bits of launch_fine_mask)
hipError_t launch_sub8(const uint32_t* fine_empty, int fgx, int fgy, int fgz, int gx, int gy, int gz, uint8_t* sub8, hipStream_t stream);
// bound classes of the half-resolution macro-cells (4 bit each) + the BOUND_CLASSES thresholds, into accel + ACCEL_CLASS_OFF
hipError_t launch_bound_class(const uint16_t* mm, int gx, int gy, int gz, const float* tf_rgba, int tf_n, float densityScale,
                              float invSigmaMax, uint32_t* accel, hipStream_t stream);
// fast bound look-up of the lane machine (svr_lanes.hpp, iterate_rot): one byte per half-resolution macro-cell from the WIDE min/max table
// (footprints grown by one voxel: launch_minmax with shift + 1, pad 1) -- a fetch is needed only if (accept draw's random word >> 24) <= byte.
// The table has one more cell on EVERY side of the grid (copies of the edge cells), so the look-up needs no clamp: cell (x, y, z) of the grid is
// entry (x + 1, y + 1, z + 1) of a table with FIXED rows of BOUND8_DIM entries and slices of BOUND8_DIM^2 (the grid has at most 32 cells per axis; 34
// is an inline constant of the two multiply-adds that form the index).
constexpr uint32_t BOUND8_DIM = 34u;
constexpr uint32_t BOUND8_BYTES = 40960u;          // >= 34^3
hipError_t launch_bound8(const uint16_t* mm_wide, int hgx, int hgy, int hgz, const float* tf_rgba, int tf_n, float densityScale,
                         float invSigmaMax, uint8_t* bnd8, hipStream_t stream);
// rows of one rank of a row-strip shard: packed (n_rows x row_floats) <-> full frame; to_packed: 1 = pack, 0 = unpack
hipError_t launch_strips(float* packed, float* frame, uint32_t row_floats, uint32_t n_rows, uint32_t strip_rows, uint32_t rank, uint32_t world,
                         int to_packed, hipStream_t stream);
// hdr_to_ldr over the owned pixels
hipError_t launch_tonemap(const DevScene& scene, const DevWork& work, hipStream_t stream);
// self-test of the ray caster's sample-chain replay (tests only): in = (t, h, bound, n) per item
hipError_t launch_chain_selftest(const float4* in, float4* out, uint32_t n, hipStream_t stream);
// device-side known-answer tests (tests only): out[i] = fn(in[i * in_stride ...]); fn ids in svr_selftest.hip
hipError_t launch_math_selftest(int fn, const float* in, uint32_t in_stride, float* out, uint32_t n, hipStream_t stream);
// property test of the fast bound look-up (tests only): rays = n x (origin, direction, u); out = n flag words (svr_selftest.hip)
hipError_t launch_bound8_selftest(const DevScene& scene, const float* rays, uint32_t n, uint32_t* out, hipStream_t stream);
// kernel_raycasting over the owned pixels
hipError_t launch_raycast(const DevScene& scene, const DevWork& work, float stepSize, bool count, int num_cus, int lanes_log2, hipStream_t stream);
// repack a [nz][ny][nx] u16 volume (device) into the padded LINEAR or BRICK layout (device)
hipError_t launch_repack(const uint16_t* src, uint16_t* dst, int nx, int ny, int nz, int layout,
                         int sy, int sz, int bnx, int bny, hipStream_t stream);

} // namespace svr
