/*
 * svr_abi.h -- C ABI of libsvr_hip.so, the MI355X (gfx950) drop-in for the render
 * launch of sunwj/SunVolumeRender.
 *
 * Plain C: POD structs, plain pointers and sizes, no C++/torch types.  Two groups:
 *
 *  (A) The seven entry points of the reference's device layer, with the SAME symbol
 *      names, so host code written against the reference (gui/canvas.cpp) links
 *      unchanged.  In the reference they are `extern "C"` functions taking C++
 *      references (pathtracer.h:17-24, raycasting.h:8); a reference is a pointer at
 *      the ABI level, so the C prototypes below are binary-identical.  The POD
 *      structs reproduce the reference classes byte for byte;
 *      include/sunvolumerender/host_api.hpp gives C++ hosts the reference's class
 *      names and methods over these layouts.
 *
 *  (B) svr_* helpers that stand in for the CUDA runtime calls the reference's HOST
 *      code makes around that layer (texture objects, cudaMalloc of the HDR buffer,
 *      device selection), plus documented extensions (render window for tile
 *      sharding, multi-frame batches, counters, timing).  gfx950 has no texture
 *      hardware exposed to HIP, so a "texture object" here is an opaque handle to a
 *      software-sampler descriptor; it is carried in the same 64-bit field the
 *      reference uses for cudaTextureObject_t.
 *
 * Error behaviour: the reference's functions return void and die through
 * checkCudaErrors (utils/helper_cuda.h:966-977: print, cudaDeviceReset, exit).  The
 * default here is the same (print to stderr, exit(EXIT_FAILURE)); with
 * svr_set_error_mode(0) errors are recorded instead and readable through
 * svr_last_error()/svr_last_error_code(), and svr_* functions return non-zero.
 *
 * Threading: like the reference (global __constant__ scene, pathtracer.cu:34-68) the
 * scene is per-process, per-device state; calls are not re-entrant.
 */
#ifndef SVR_ABI_H
#define SVR_ABI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SVR_MAX_LIGHT_SOURCES 8            /* common.h:11 */
#define SVR_TF_TABLE_SIZE 1024             /* gui/transferfunction.h:29 */

/* ---- POD layouts (all 4-byte floats; glm::vec3 is a packed 12-byte triple) ---- */
typedef struct svr_vec2 { float x, y; } svr_vec2;
typedef struct svr_vec3 { float x, y, z; } svr_vec3;

typedef struct svr_bbox {                  /* cudaBBox, core/geometry/cuda_bbox.h:66-69; 36 B */
    svr_vec3 vmin, vmax, invSize;
} svr_bbox;

typedef struct svr_volume {                /* cudaVolume, core/cuda_volume.h:111-121; 112 B, align 8 */
    svr_bbox bbox;                         /*   0 */
    uint32_t _pad0;                        /*  36 */
    uint64_t tex;                          /*  40  handle from svr_create_volume_texture */
    float densityScale;                    /*  48 */
    float invMaxMagnitude;                 /*  52 */
    float gradientFactor;                  /*  56 */
    svr_vec3 spacing;                      /*  60 */
    svr_vec3 invSpacing;                   /*  72 */
    svr_vec2 x_clip, y_clip, z_clip;       /*  84, 92, 100 */
    uint32_t _pad1;                        /* 108 */
} svr_volume;

typedef struct svr_transfer_function {     /* cudaTransferFunction, core/cuda_transfer_function.h:57-59; 16 B */
    uint64_t tex;                          /* handle from svr_create_tf_texture */
    float maxOpacity;                      /* Woodcock majorant */
    uint32_t _pad;
} svr_transfer_function;

typedef struct svr_camera {                /* cudaCamera, core/cuda_camera.h:98-106; 76 B */
    uint32_t imageW, imageH;
    float exposure, apeture, focalLength, aspectRatio, tanFovxOverTwo;
    svr_vec3 pos, u, v, w;
} svr_camera;

typedef struct svr_disk {                  /* cudaDisk, core/geometry/cuda_disk.h:58-61; 28 B */
    float radius;
    svr_vec3 center, normal;
} svr_disk;

typedef struct svr_area_light {            /* cudaAreaLight, core/lights/cuda_arealight.h:68-71; 44 B */
    svr_disk disk;
    svr_vec3 color;
    float intensity;
} svr_area_light;

typedef struct svr_environment_light {     /* cudaEnvironmentLight, core/lights/cuda_environment_light.h:74-78; 32 B */
    uint64_t tex;                          /* 0 = constant defaultRadiance; else handle from svr_create_env_texture */
    svr_vec3 defaultRadiance;
    float intensity;
    svr_vec2 offset;
} svr_environment_light;

typedef struct svr_render_params {         /* RenderParams, core/render_parameters.h:34-37; 16 B */
    uint32_t traceDepth;                   /* default 1 */
    uint32_t frameNo;                      /* default 0 */
    void* hdrBuffer;                       /* device pointer, W*H packed float3 (glm::vec3*) */
} svr_render_params;

/* =====================================================================
 * (A) the reference's device-layer entry points (same symbol names)
 * ===================================================================== */

#ifndef SVR_ABI_NO_REFERENCE_PROTOTYPES   /* host_api.hpp re-declares these seven with the reference's C++ signatures */
/* pathtracer.h:17 / pathtracer.cu:292-304.  One call = one sample per pixel: clears the
 * accumulator iff frameNo==0, traces one path per pixel with seed wangHash(frameNo) +
 * y*W + x, folds it into the running mean in hdrBuffer and tone-maps into img (device
 * pointer, W*H RGBA8).  W,H come from the last setup_camera (the reference bakes 640x640,
 * common.h:8-9).  Asynchronous on the launch stream, like the reference. */
void render_pathtracer(void* img, const svr_render_params* renderParams);

/* pathtracer.h:20-24 / pathtracer.cu:34-68: copy the POD into the per-device scene. */
void setup_volume(const svr_volume* vol);
void setup_transferfunction(const svr_transfer_function* tf);
void setup_camera(const svr_camera* cam);
void setup_env_lights(const svr_environment_light* light);
void setup_area_lights(svr_area_light* lights, uint32_t n);   /* n is clamped to SVR_MAX_LIGHT_SOURCES */

/* raycasting.h:8 / raycasting.cu:15-75: emission-absorption ray caster; scene passed by
 * argument, not through the setup_* state, as in the reference. */
void render_raycasting(void* img, svr_volume* volume, svr_transfer_function* transferFunction,
                       svr_camera* camera, float stepSize);
#endif /* SVR_ABI_NO_REFERENCE_PROTOTYPES */

/* =====================================================================
 * (B) helpers replacing the CUDA runtime calls of the reference's host code
 * ===================================================================== */

/* main.cpp:7-33 chooseBestDevice + cudaSetDevice: select the HIP device for this process. */
int svr_init(int device);
void svr_shutdown(void);
/* launch stream for every kernel (a hipStream_t; NULL = the null stream).  Lets a host
 * that owns streams (e.g. PyTorch) order the renderer with its own work. */
int svr_set_stream(void* hip_stream);
int svr_device_synchronize(void);                     /* canvas.cpp:106 cudaDeviceSynchronize */

void svr_set_error_mode(int fatal);                   /* 1 (default): checkCudaErrors behaviour */
const char* svr_last_error(void);
int svr_last_error_code(void);
void svr_clear_error(void);

/* VolumeReader.cpp:138-172 (cudaMalloc3DArray + cudaMemcpy3D + cudaCreateTextureObject,
 * border / linear / normalized-float / normalized coords): voxels is [nz][ny][nx] u16.
 * src_is_device: voxels is a device pointer.  layout: SVR_LAYOUT_*.  Returns 0 on error. */
#define SVR_LAYOUT_AUTO 0
#define SVR_LAYOUT_LINEAR 1          /* [z][y][x] with a 2-voxel zero apron */
#define SVR_LAYOUT_BRICK 2           /* 8x4x4-voxel bricks (256 B), 2-voxel zero apron */
#define SVR_LAYOUT_PAIR 3            /* the same bricks with 32-bit elements: voxel x | voxel x+1 << 16 (half the gather instructions per
                                        trilinear fetch, twice the memory; 32-bit byte offsets: volumes up to ~1000^3); what AUTO falls back to
                                        when CELL does not fit the addressing limits or the device memory */
#define SVR_LAYOUT_CELL 4            /* the same bricks with 16-byte elements: the 8 voxels of a trilinear cell -- one 16-byte load per fetch, one
                                     * sector instead of four; 8 x the memory of the u16 volume (2.2 GB for 512^3, 17 GB for 1024^3); 32-bit element
                                     * index, 64-bit byte offsets.  What AUTO picks FIRST (then PAIR, then BRICK, then LINEAR: the next one when
                                     * the addressing limits or hipErrorOutOfMemory rule one out; csrc/svr_api.hip, create_volume_texture) */
uint64_t svr_create_volume_texture(const uint16_t* voxels, int nx, int ny, int nz,
                                   int src_is_device, int layout);
/* gui/transferfunction.cpp:30-44 (1D float4 array, clamp / linear / normalized coords). */
uint64_t svr_create_tf_texture(const float* rgba, int n, int src_is_device);
/* gui/transferfunction.cpp:128-176: re-upload after an edit (same n). */
int svr_update_tf_texture(uint64_t handle, const float* rgba, int n, int src_is_device);
/* core/lights/lights.cpp:41-74 (2D float4 array, wrap / linear / normalized coords). */
uint64_t svr_create_env_texture(const float* rgba, int w, int h, int src_is_device);
int svr_destroy_texture(uint64_t handle);             /* cudaDestroyTextureObject + cudaFreeArray */

/* core/render_parameters.h:17-32: RenderParams::SetupHDRBuffer / Clear. */
int svr_render_params_setup_hdr(svr_render_params* p, uint32_t w, uint32_t h);
int svr_render_params_clear(svr_render_params* p);

void* svr_device_malloc(size_t bytes);
int svr_device_free(void* p);
int svr_memcpy_h2d(void* dst_device, const void* src_host, size_t bytes);
int svr_memcpy_d2h(void* dst_host, const void* src_device, size_t bytes);   /* canvas.cpp:100 */
int svr_memset_device(void* dst_device, int value, size_t bytes);

/* ---- documented extensions (not in the reference) ---- */

/* Restrict render_pathtracer / render_raycasting to the rows y with
 * (y / strip_rows) % world == rank (interleaved row strips; multi-GPU tile sharding).
 * Seeds stay global (y*W+x), so the union over ranks is bit-identical to one GPU.
 * strip_rows=0 or world<=1 resets to the full frame. */
int svr_set_row_shard(uint32_t strip_rows, uint32_t rank, uint32_t world);
/* Restrict to the pixel window [x0,x1) x [y0,y1) (combined with the row shard). Negative x1/y1 = full. */
int svr_set_render_window(int x0, int y0, int x1, int y1);

/* ---- frame assembly for row-sharded renders: the native counterpart of sunvolumerender_amd/dist.py's FrameAssembler ----
 * A rank's accumulator holds the rows it owns (svr_set_row_shard) and zeros elsewhere.  The rows of rank r, in increasing
 * y, form its PACKED buffer: svr_strip_rows_owned() rows of W x float3.  One collective per output frame: every rank packs
 * and sends its rows to `root`, root unpacks them into the full frame -- with RCCL one point-to-point transfer per peer
 * over its own xGMI link (1.5 MB per peer at 1024^2 on 8 GPUs).
 * The two index functions are plain host code (no GPU needed). */
uint32_t svr_strip_rows_owned(uint32_t H, uint32_t strip_rows, uint32_t rank, uint32_t world);
/* y of the p-th owned row (p < svr_strip_rows_owned); 0xffffffff if p is out of range */
uint32_t svr_strip_row_to_y(uint32_t p, uint32_t H, uint32_t strip_rows, uint32_t rank, uint32_t world);
/* device buffers, on the library's stream: packed <- the rows of `rank` out of the full-size W x H x float3 `hdr`, and back */
int svr_pack_strips(void* packed, const void* hdr, uint32_t W, uint32_t H, uint32_t strip_rows, uint32_t rank, uint32_t world);
int svr_unpack_strips(void* frame, const void* packed, uint32_t W, uint32_t H, uint32_t strip_rows, uint32_t rank, uint32_t world);
/* The whole exchange, for a host that holds an RCCL communicator (one process per GPU, as with sunvolumerender_amd.dist):
 * nccl_comm = the host's ncclComm_t (world ranks; rank / world must be its rank / size).  Every rank packs its rows of
 * hdr_local (its accumulator; not modified) and ncclSend()s them to `root`; root ncclRecv()s the peers' rows and unpacks all of
 * them into frame_on_root (W x H x float3, device; ignored on the other ranks).  Enqueued on the library's stream
 * (svr_set_stream); the staging buffer is the library's.  RCCL is resolved at the first call from the library the process
 * has already loaded (dlopen("librccl.so")), so libsvr_hip.so itself does not link against it.
 * Afterwards svr_hdr_to_ldr_frame(img, frame_on_root, W, H) tone-maps the assembled frame on root. */
int svr_assemble_frame(void* nccl_comm, void* frame_on_root, const void* hdr_local, uint32_t W, uint32_t H,
                       uint32_t strip_rows, uint32_t rank, uint32_t world, uint32_t root);

#define SVR_OPT_ENV_ON_ESCAPE 1   /* 1: add T*env(dir) when a path leaves the volume (the line the reference
                                     comments out, pathtracer.cu:233).  default 0 = reference behaviour */
#define SVR_OPT_KERNEL 2          /* 0 auto (= 2); 1 one block per 16x16 tile (reference-shaped baseline);
                                     2 persistent waves, one 8x8 tile-task per wave, empty-space skipping (default);
                                     3 persistent waves with per-lane state machine and ballot/mbcnt lane regeneration;
                                     4 wavefront: gen / walk / shade kernels over dense queues with ballot + prefix-sum compaction */
#define SVR_OPT_COUNT 3           /* 1: count volume taps etc. (slower; for roofline accounting) */
#define SVR_OPT_TIMING 4          /* 1: bracket the path-tracing kernel with HIP events */
#define SVR_OPT_SKIP_TONEMAP 5    /* 1: render_pathtracer does not run hdr_to_ldr (batch rendering) */
#define SVR_OPT_BLOCKS_PER_CU 6   /* persistent kernel: workgroups per CU (0 = default) */
#define SVR_OPT_PIPELINE 7        /* 1 (default): trace kernels of consecutive frame groups run on internal streams
                                     and overlap; the accumulator is still updated in frame order on the caller's
                                     stream.  0: everything on the caller's stream */
#define SVR_OPT_EMPTY_SKIP 9       /* 1 (default): skip the voxel fetches of Woodcock iterations in macro-cells where the
                                     transfer function is exactly transparent (bit-identical results; RNG still advanced) */
#define SVR_OPT_RAY_SKIP 10        /* 1 (default): per-ray conservative march over the dilated empty mask: a walk that can
                                     never meet a non-transparent macro-cell ends at once when no random draw follows it,
                                     and other walks skip cell tests up to their first possibly-occupied cell (bit-identical) */
#define SVR_OPT_BOUND_CULL 15       /* 0 off, 1 (default) where it pays (>= 2 % of the occupied coarse macro-cells have a bound below 1), 2 always:
                                     majorant-bound fetch culling -- a Woodcock iteration draws its accept number first and
                                     fetches only if it is below (largest alpha reachable in the macro-cell) / sigma_max; bit-identical
                                     results, same random-number stream (csrc/svr_accel.hip, k_bound_class) */
#define SVR_OPT_PARK_END 19         /* lane machine of the tile kernel at traceDepth > 1 (csrc/svr_lanes.hpp): a round of services (BSDF sampling /
                                     shading of the waiting paths) runs when this many lanes of the wave could be put to work by it (or none
                                     walks); 1..64, default 32.  Speed only */
#define SVR_OPT_QUEUE 18            /* the tile kernel shades the first scatter events of a task in place, queues the paths and continues them on a
                                     per-lane state machine that keeps the 64 lanes of a wave walking (csrc/svr_lanes.hpp): 0 never, 2 always (on
                                     folding launches), 1 (default) where it pays: empty-space skipping on and traceDepth >= 2, a medium without
                                     exactly transparent space, or a launch large enough to drain every wave's queue 4 times.  Results identical */
#define SVR_OPT_FOLD 17             /* 1 (default): a many-frame launch of the tile kernel folds its frames into the HDR accumulator itself (running
                                     mean in frame order, in the wave that traced them); 0: scratch slot per frame + resolve kernel */
#define SVR_OPT_FAST_MATH 14        /* OPT-IN, default 0: the tile kernel's fast-math build (v_log_f32 in the walk, reciprocal division, fma contraction;
                                     in the spirit of the reference's -use_fast_math, CMakeLists.txt:9-10).  NOT bit-identical to the default mode:
                                     converged images agree within Monte-Carlo noise (tests/test_fast_math_gpu.py) */
#define SVR_OPT_FINE_MASK 20        /* second, finer level of `empty` macro-cells (half the edge) in global memory for the per-fetch test:
                                     0 (default) off, 1 when the LDS-resident cells are >= 16 voxels (volumes beyond 512^3), 2 whenever it exists.
                                     Fewer fetches (c5: 3.9 instead of 5.9 per path), same speed: the test costs a dependent cached load */
#define SVR_OPT_ROW_ORDER 21        /* tile kernel work distribution: 1 = each of the 8 ticket counters (one per XCD group of blocks) owns whole tile
                                     rows, so neighbouring tiles share an XCD's L2; 0 = every 8th task.  Speed only */
#define SVR_OPT_GROUP_FRAMES 22     /* frames per folding launch of svr_render_pathtracer_frames: 8, 16, 32 or 64 (default: a wave = one pixel x 64 frames).  Speed only */
#define SVR_OPT_LOCAL_MAJORANT 23   /* OPT-IN, default 0: "Woodcock max-density acceleration" -- the free-flight sampler tracks against per-macro-cell
                                     * (local) majorants instead of the reference's single global one (core/woodcock_tracking.h:29-31): same law of the
                                     * collision point, far fewer iterations, but a different consumption of random numbers.  NOT bit-identical to
                                     * the default mode; converged images agree within Monte-Carlo noise (tests/test_local_majorant_gpu.py).  Needs
                                     * SVR_OPT_EMPTY_SKIP = 1 and clip planes inside the volume; otherwise the default kernel renders.
                                     * Meant for volumes >= 512^3 at >= 1024^2 pixels, fog-like media and deeper paths (c3n 2.1 x, c5 + 34 %, c3 + 8 %, depth 4 + 9 %); on a
                                     * 256^3 volume at 512^2 (c2) the default kernel has little left to skip and the pool's batches cost more than they save: - 24 % there.
                                     * 1 = the pool kernels; 2 = the straight-line form of the same algorithm (identical results, slower: the
                                     * reference the pool kernels are tested against) */
#define SVR_OPT_LIGHT_CULL 24       /* 1 (default): area lights that no camera ray can reach (behind the lens plane or outside the view frustum, lens
                                     * and pixel jitter included; conservative host-side test) are skipped by the primary rays' nearest-light test
                                     * (core/lights/light_sample.h:23-49).  Results unchanged */
#define SVR_OPT_LM_TUNE 25          /* local-majorant pool kernel, speed only: macro-cells a walking lane may cross per turn | idle lanes that trigger a
                                     * refill << 8 | ended walks that trigger their settling << 16 (each 1..64) | tasks per batch of the traceDepth-1 pool << 24
                                     * (0 = the default for the scene; < 16 selects the 10-task build, >= 16 the 21-task build); 0 (default) = everything chosen per scene */
#define SVR_OPT_LM_SUBCELLS 26      /* local-majorant mode: a walk spends its free path only in the occupied eighths (2 x 2 x 2 fine cells) of a macro-cell --
                                     * fewer wasted tentative collisions where a surface cuts a cell.  0 off, 1 (default) where macro-cells are >= 16 voxels
                                     * (volumes beyond 512^3), 2 always.  Changes the random numbers a path
                                     * consumes (another, equally valid estimate), so it is part of the mode's definition, not a speed-only switch */
#define SVR_OPT_PARK_CHEAP 27       /* lane machine of the tile kernel at traceDepth 1: ended shadow walks + idle lanes with a record waiting before the wave
                                     * settles / refills them (1..64, default 16).  Speed only */
#define SVR_OPT_PINHOLE_FAST 28     /* 1 (default): with apeture == 0 (the reference's default) the camera ray skips the square root and the sine / cosine of
                                     * the lens sample, which is (+-0, +-0) and provably changes no bit of the ray; the draws are consumed.  Results unchanged */
#define SVR_OPT_POOL 29             /* tile kernel with the queue machine: the primary walks are pooled too (a task only generates its camera
                                     * rays; the lane machine walks them, the collisions are shaded 64 at a time).  0 off, 1 (default) for media without
                                     * exactly transparent space (their walks are not coherent within a wave), 2 always.  Results unchanged */
#define SVR_OPT_TRIPS 30            /* lane machine of the tile kernel (pooled walks at traceDepth 1, every walk of deeper paths): the walking lanes run FIVE
                                     * Woodcock iterations per turn with the generator as a circular buffer (no register moves), a lane that needs a fetch waits
                                     * for the end of the trip.  0 off, 1 (default) for media without exactly transparent space, 2 always.  Results unchanged */
#define SVR_OPT_NAN_GUARD 31        /* OPT-IN, default 0 = the reference's behaviour: running_estimate (pathtracer.cu:81-84,279) keeps a NaN for good, and the
                                     * reference's own arithmetic yields one (0/0 in the microfacet term, pathtracer.cu:106-131, core/bsdf/microfacet.h:52-68) for a
                                     * handful of paths in 10^8..10^9 -- a permanently dead pixel.  1: a non-finite sample (per channel) is replaced by the running mean
                                     * before it is folded in, i.e. dropped.  Every other sample and pixel keeps its bits */
#define SVR_OPT_MACRO_SHIFT_MIN 32  /* volume textures created from now on get macro-cells of at least 2^v voxels per axis (0..6; default 0 = the smallest cells
                                     * whose grid fits 64^3).  Coarser acceleration data; the default mode's results are unchanged (bit-exact), the local-majorant
                                     * mode's estimate changes with its grid.  For tests of the coarse-grid code paths on small volumes */
#define SVR_OPT_SPLIT 33            /* OPT-IN, default 0: many-frame launches of deeper paths (traceDepth >= 2) as TWO kernels (csrc/svr_trace_split.hip): the front half (primary walks,
                                     * first scatter events and their shadow walks in place) writes the paths that go on into a launch-wide pool of record chunks, the lane machine
                                     * drains them in a kernel with a register budget of its own; the frames of the launch go through scratch slots and the resolve kernel.  Built to
                                     * give the machine its own registers (+ 3-6 % over the fused kernel of the time); the fused kernel has since caught up and is 0.5-1.5 % ahead
                                     * without the pool, so this is off by default.  1 / 2: on.  Needs room for the worst-case pool (14 GB for 64 frames at 1024^2; else the fused
                                     * kernel renders; also for media whose primary walks are pooled and for launches of < 8 frames).  Results unchanged (bit-exact) */
#define SVR_OPT_ENV_NEE 34          /* OPT-IN, default 0: importance sampling of the environment MAP (csrc/svr_trace_env.hip).  With SVR_OPT_ENV_ON_ESCAPE the environment lights the
                                     * medium only through the directions the BSDF / phase sampling picks (core/lights/cuda_environment_light.h:58-72 is a lookup, nothing more).
                                     * 1: every scatter event that is followed by a bounce also draws one direction from the map's luminance (a table built on the GPU by
                                     * svr_create_env_texture), walks a shadow ray along it and combines the two estimates with the balance heuristic.  Same image in expectation
                                     * -- the reference's throughput update, reported pdfs and roulette are reproduced term by term (tests/test_env_nee_gpu.py: means within 4
                                     * standard errors at 4 096 spp) -- far less noise under a small bright sun; NOT bit-identical (the path's random stream shifts).  Needs an
                                     * env texture, SVR_OPT_ENV_ON_ESCAPE = 1 and traceDepth >= 2; otherwise inert.  Straight-line paths: about half the speed of the default kernel */
#define SVR_OPT_FAST_BOUND 35       /* default 1: media without exactly transparent space (walks of tens of iterations under bound culling: the pooled lane machine of the tile
                                     * kernel) look the fetch bound of an iteration up from the ray parameter -- one fma per axis into a byte table whose cells cover one more
                                     * voxel per side, compared with the top 8 bits of the accept draw's random word (csrc/svr_accel.hip k_bound8) -- instead of from the exact
                                     * trilinear cell and a class threshold.  Any valid bound culls correctly: results unchanged (bit-exact); c3n +17 %, c3n at depth 2 / 4 +32 % / +31 % (same box).  Needs the clipped box inside the volume and the camera within 2^21 / (16 N) volume extents (else, and with 0: the exact cell) */
#define SVR_OPT_FRAME_AHEAD 13           /* render_pathtracer traces frames ahead of the calls that ask for them (batches of 1, 2, 4 ... 32 frames; results unchanged); default 1 */
#define SVR_OPT_RAYCAST_LANES_LOG2 12   /* ray caster: 1 << v adjacent lanes share one ray (samples of a chunk in parallel, composited in order); 0..5, default 3 */
#define SVR_OPT_FRAMES_PER_WAVE_LOG2 11 /* tile kernel: a wave traces (64 >> f) pixels x (1 << f) frames of a group; -1 (default) = up to 8 frames */
#define SVR_OPT_REFILL_MIN_IDLE 8 /* persistent kernel: regenerate lanes once this many of a wave's 64 lanes are idle
                                     (64 = a wave finishes its 8x8 tile before taking the next; default) */
int svr_set_option(int key, int value);
int svr_get_option(int key);

/* render_pathtracer for nframes consecutive frame numbers (renderParams->frameNo ...
 * +nframes-1) in ONE launch; bit-identical to nframes calls, tone-maps once at the end. */
int svr_render_pathtracer_frames(void* img, const svr_render_params* renderParams, uint32_t nframes);

/* hdr_to_ldr alone (pathtracer.cu:282-290), over the pixels this process owns (row shard / window). */
int svr_hdr_to_ldr(void* img, const svr_render_params* renderParams);
/* hdr_to_ldr over a WHOLE w x h frame, whatever shard or window is set: the tone map of the frame rank 0 has
 * assembled from the ranks' strips (hdr: device pointer, w*h packed float3; exposure of the last setup_camera). */
int svr_hdr_to_ldr_frame(void* img, const void* hdr, uint32_t w, uint32_t h);

typedef struct svr_counters {
    uint64_t paths;
    uint64_t vol_taps;         /* 8-voxel trilinear fetches of the algorithm (what the reference issues) */
    uint64_t woodcock_iters;
    uint64_t scatter_events;
    uint64_t shadow_walks;
    uint64_t raycast_steps;
    uint64_t loop_iters;       /* persistent kernels: wave-level scheduler iterations / tile tasks */
    uint64_t vol_taps_executed; /* fetches actually issued (<= vol_taps: empty-space skipping, reused scatter tap) */
    uint64_t walks_ray_skipped; /* Woodcock walks a non-counting build ends at once (whole-ray test) */
    uint64_t iters_ray_skipped; /* ... and the Woodcock iterations those walks contain */
    uint64_t iters_prefix_skipped; /* iterations before a walk's first possibly-occupied macro-cell (no cell test) */
    uint64_t taps_bound_culled; /* fetches of non-empty cells not issued because the accept draw was above the cell's bound */
} svr_counters;
/* test hook: the ray caster's sample-chain replay on n items (t, h, bound, steps) -> (count <, count <=, t after steps, flags) */
int svr_selftest_chain(const float* items, float* results, uint32_t n);
/* test hook: device-side known-answer tests of the numeric contract.  out[i] = fn(in[i*in_stride ...]) evaluated by the
 * device functions the kernels use: 0 schlick_fresnel(ni, no, cos)  1 logf  2 expf  3 sinf  4 cosf  5 acosf
 * 6 atan2f(y, x)  7 powf(x, y)  8 k-th curand_uniform of curand_init(seed bits, 0, 0): in = (seed bits, k)
 * 9 wangHash(bits) as bits  10 log(1 - u) of the Woodcock walk (logf_unit) */
int svr_selftest_math(int fn, const float* in, uint32_t in_stride, float* out, uint32_t n);
/* test hook: property test of SVR_OPT_FAST_BOUND's table on the scene set up as for render_pathtracer.  rays = n x 7 floats (origin, direction, u):
 * at the point t = tMin + u (tMax - tMin) of the ray's box interval, the byte the lane machine would read against the product the reference's
 * accept test forms there (exact cell, fetch, alpha, invSigmaMax).  out[i] = 1 tested | 2 VIOLATION (a draw the byte culls could be accepted)
 * | 4 index outside the table | byte << 8; 0 = the ray misses the box.  Fails (-3) when the look-up is not in use for the scene */
int svr_selftest_bound8(const float* rays, uint32_t n, uint32_t* out);
int svr_get_counters(svr_counters* out);              /* synchronises the launch stream */
int svr_reset_counters(void);

/* accumulated HIP-event time of the path-tracing kernel since the last reset (SVR_OPT_TIMING) */
int svr_get_kernel_time(double* total_ms, uint64_t* launches);
int svr_reset_kernel_time(void);

/* "name major.minor gcnArch CUs" of the active device, for logs */
const char* svr_device_info(void);
int svr_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* SVR_ABI_H */
