// svr_scene.hpp -- the device-side scene: what the reference keeps in __constant__
// globals (pathtracer.cu:34-68), resolved on the host into one POD that is passed BY
// VALUE as a kernel argument (kernarg segment -> scalar loads; wave-uniform, no
// hipMemcpyToSymbol, no device sync in setup_*).
#pragma once
#include <stdint.h>

namespace svr {

// volume memory layouts (software "texture"; gfx950 exposes no image/sampler path to HIP)
enum { LAYOUT_LINEAR = 1, LAYOUT_BRICK = 2, LAYOUT_PAIR = 3, LAYOUT_CELL = 4 };   // PAIR: BRICK with 32-bit elements (voxel x | voxel x + 1 << 16); CELL: 16-byte elements = the 8 voxels of a trilinear cell
constexpr int VOL_PAD = 2;         // zero apron, voxels, each side (border addressing)
constexpr int BRICK_X = 8, BRICK_Y = 4, BRICK_Z = 4;   // 8*4*4 u16 = 256 B

struct DevLight {                  // cudaAreaLight + values its getters compute
    float radius;
    float center[3];
    float normal[3];
    float radiance[3];             // cuda_arealight.h:57, evaluated on the host in the same float ops
    float area;                    // cuda_disk.h:53-56
};

struct DevScene {
    // ---- cudaVolume (core/cuda_volume.h) ----
    float vmin[3];
    float invSize[3];
    float clip_vmin[3];            // vmin * (-clip[0])   cuda_bbox.h:38
    float clip_vmax[3];            // vmax * ( clip[1])   cuda_bbox.h:39
    float densityScale;
    float invMaxMagnitude;
    float gradientFactor;
    float pbrdf_c;                 // ((-25*gf)*gf)*gf    pathtracer.cu:251 prefix
    float spacing[3];
    float invSpacing[3];
    // software volume texture
    const uint16_t* vox;
    int32_t nx, ny, nz;
    int32_t layout;
    float fnx, fny, fnz;
    int32_t sy, sz;                // LINEAR: element strides of the padded array
    int32_t bnx, bny;              // BRICK: bricks per row / per slab-row
    // empty-space bitmask (one bit per macro-cell of 2^mc_shift cells per axis; bit set = every
    // trilinear fetch whose cell lies in the macro-cell has transfer-function alpha exactly 0)
    const uint32_t* empty_mask;    // device: DIST_WORDS_MAX words of packed half-resolution distances, MASK_WORDS_MAX words of deep-empty bits, mask_words words of empty bits; null = no skipping
    int32_t mc_shift, mc_gx, mc_gy, mc_gz, mc_gxy;
    int32_t mc_hgx, mc_hgxy;       // half-resolution grid of the distance field: row and slice strides
    uint32_t mask_words, dist_words;
    uint32_t ray_skip;             // 1: the clipped box lies inside the texture domain, so whole-ray tests are valid
    const uint32_t* fine_mask;     // `empty` bits of the fine level (cells of 2^(mc_shift-1)), global memory; null = not used
    int32_t fg_x, fg_y, fg_z, fg_xy;   // its grid and slice stride (cells)
    const uint8_t* sub8;           // per macro-cell: occupancy bits of its 2 x 2 x 2 fine cells (svr_accel.hip, k_sub8); null = none (local-majorant walks)
    uint32_t has_empty;            // 0: not one macro-cell is `empty` (media without exactly transparent space): the walks skip the mask look-ups
    uint32_t bound_cull;           // 1: the bound-class table behind the masks is valid (majorant-bound fetch culling)
    uint32_t park_end;             // lane machine: lanes waiting for shading / a walk's end / a new record before the wave serves them
    uint32_t park_cheap;           // lane machine at traceDepth 1: ended walks + idle lanes with a record waiting before the wave serves them
    uint32_t lm_tune;              // local-majorant pool (svr_trace_lm.hip): cells per turn | idle lanes before a refill << 8 | ended walks before they are settled << 16
    float mc_scale[3];             // macro-grid coordinate = (p - vmin) * mc_scale + mc_off
    float mc_off;
    // ---- cudaTransferFunction ----
    const float* tf;               // tf_n x float4
    int32_t tf_n;
    float tf_nf;
    float sigmaMax;
    float invSigmaMax;             // 1/sigmaMax                      woodcock_tracking.h:30
    float invSigmaMaxSI;           // 1/(sigmaMax*BASE_SAMPLE_STEP)   woodcock_tracking.h:31
    // ---- cudaCamera ----
    uint32_t imageW, imageH;
    float exposure, apeture, focalLength, aspectRatio, tanFovxOverTwo;
    float wm1, hm1;                // imageW - 1.f, imageH - 1.f
    uint32_t cam_pinhole;          // 1: apeture is +0 and the lens sample provably cannot change a bit of the ray (svr_api.hip): camera_ray skips its sqrt / sincos
    float cam_pos[3], cam_u[3], cam_v[3], cam_w[3];
    // ---- cudaEnvironmentLight ----
    const float* env;              // env_h x env_w x float4, or null
    int32_t env_w, env_h;
    float env_default[3];
    float env_intensity;
    float env_offset[2];
    uint32_t env_on_escape;
    // ---- area lights ----
    uint32_t num_lights;
    uint32_t primary_light_mask;   // bit i: a camera ray can reach light i (host-side conservative frustum test); the others are skipped in the nearest-light test
    DevLight lights[8];
    uint32_t trips;                // lane machine of the tile kernel (svr_lanes.hpp): walking lanes run five iterations per turn (SVR_OPT_TRIPS).  LAST, so that the layout
                                   // of everything else -- and with it the code of every kernel that does not read it -- stays what it was
    // (appended behind it for the same reason) SVR_OPT_ENV_NEE, svr_trace_env.hip: the env map's sampling table -- env_h rows of env_w + 1 prefix sums of the
    // texel weights, then env_h + 1 prefix sums of the row sums -- or null
    const float* env_cdf;
    // (appended) fast bound look-up of the lane machine's POOL builds (svr_accel.hip, k_bound8; svr_lanes.hpp, iterate_rot): the byte table (BOUND8_BYTES, global
    // memory; the kernel keeps it in LDS) or null = not used, and the table coordinate of a world point, (p - vmin) * hc_scale + hc_off (half-resolution
    // macro-grid coordinate + 1: the table has one more cell around the grid)
    const uint8_t* bnd8;
    float hc_scale[3];
    float hc_off;
};

// per-launch work description.  Tracing and accumulation are decoupled: the trace kernel writes the
// radiance of each path into a scratch slot lbuf[slot][pixel] (one slot per frame of the group),
// and the resolve kernel folds the slots into the running mean in frame order (bit-identical to
// the reference's sequential running_estimate) and tone-maps, with coalesced row-contiguous access.
struct DevWork {
    float* hdr;                    // W*H packed float3 (resolve only)
    uint8_t* img;                  // W*H RGBA8 or null (resolve only)
    float* lbuf;                   // scratch radiance: nframes slots of slot_stride floats
    uint32_t slot_stride;          // floats per slot = 3*W*H
    uint32_t traceDepth;
    uint32_t frame0;               // first frame number of this group
    uint32_t nframes;              // frames in this group (= slots used)
    uint32_t x0, y0, x1, y1;       // pixel window
    uint32_t strip_rows, rank, world;   // interleaved row-strip shard (world<=1: off)
    uint32_t n_rows;               // number of owned rows inside the window
    uint32_t n_items;              // owned pixels = n_rows * (x1-x0)
    uint32_t unit;                 // tile kernel: consecutive tasks handed out per ticket
    uint32_t frames_log2;          // tile kernel: log2(frames per wave); a wave = (64 >> f) pixels x (1 << f) frames
    uint32_t fold;                 // tile kernel: 1 = fold the group's frames into hdr in the kernel (running mean in frame order;
                                   // needs nframes <= 64 so that a pixel's frames sit in one wave); 0 = write the scratch slots
    uint32_t row_order;            // tile kernel: 1 = a ticket counter owns whole tile rows (XCD-local rows), 0 = every 8th task
    uint32_t debug_stop;           // timing ablation only (0 = off): 1 stop after set-up, 2 after the whole-ray test, 3 after the primary walk
    uint32_t refill_min_idle;      // persistent kernel: regenerate lanes once this many are idle (64 = tile-synchronous)
    unsigned long long* counters;  // svr_counters on the device, or null
    uint32_t* ticket;              // persistent kernel work counter
    uint32_t* queue;               // tile kernel, QUEUE builds: scatter-record queues, REC_WORDS * QUEUE_CAP words per wave; null = straight-line paths
    float* pend;                   // ... and the waves' pending-radiance rows, QUEUE_TASKS * 3 * 64 floats per wave
    uint32_t queue_blocks;         // blocks the queue memory is sized for
    uint32_t nan_guard;            // SVR_OPT_NAN_GUARD: the running mean skips non-finite samples (default 0 = the reference's behaviour)
};

} // namespace svr
