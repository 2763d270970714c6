// svr_api.hip -- C-ABI layer of libsvr_hip.so (include/svr_abi.h).
//
// Holds what the reference keeps in __constant__ globals (pathtracer.cu:34-68) as a
// per-process host-side scene, resolves the opaque texture handles into software-sampler
// descriptors, and launches the gfx950 kernels.  No CPU rendering path exists here: every
// entry point either launches HIP work or reports an error.
#include "../../include/svr_abi.h"
#include "svr_kernels.hpp"
#include "svr_device.hpp"   // wang_hash (host)

#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

static_assert(sizeof(svr_vec3) == 12, "glm::vec3 layout");
static_assert(sizeof(svr_bbox) == 36, "cudaBBox layout");
static_assert(sizeof(svr_volume) == 112 && offsetof(svr_volume, tex) == 40 && offsetof(svr_volume, densityScale) == 48 &&
              offsetof(svr_volume, spacing) == 60 && offsetof(svr_volume, invSpacing) == 72 && offsetof(svr_volume, x_clip) == 84 &&
              offsetof(svr_volume, z_clip) == 100, "cudaVolume layout");
static_assert(sizeof(svr_transfer_function) == 16 && offsetof(svr_transfer_function, maxOpacity) == 8, "cudaTransferFunction layout");
static_assert(sizeof(svr_camera) == 76 && offsetof(svr_camera, pos) == 28 && offsetof(svr_camera, w) == 64, "cudaCamera layout");
static_assert(sizeof(svr_disk) == 28, "cudaDisk layout");
static_assert(sizeof(svr_area_light) == 44 && offsetof(svr_area_light, color) == 28 && offsetof(svr_area_light, intensity) == 40, "cudaAreaLight layout");
static_assert(sizeof(svr_environment_light) == 32 && offsetof(svr_environment_light, defaultRadiance) == 8 &&
              offsetof(svr_environment_light, intensity) == 20 && offsetof(svr_environment_light, offset) == 24, "cudaEnvironmentLight layout");
static_assert(sizeof(svr_render_params) == 16 && offsetof(svr_render_params, hdrBuffer) == 8, "RenderParams layout");
static_assert(sizeof(svr_counters) == 8 * svr::CNT_N, "counter block");

namespace svr_fast { hipError_t launch_trace_tile_raw(const void* scene, const void* work, const void* cfg, hipStream_t st); }   // svr_trace_tile_fast.hip

namespace {

enum TexKind { TEX_VOLUME = 1, TEX_TF = 2, TEX_ENV = 3 };
constexpr size_t DEBUG_WORDS = 32;

struct Texture {
    uint32_t magic;
    int kind;
    void* data;          // device
    int nx, ny, nz;      // volume dims / tf n / env w,h
    int layout;
    int sy, sz, bnx, bny;
    size_t bytes;
    // volume: per-macro-cell min/max of the raw voxels (empty-space skipping)
    uint16_t* mm = nullptr;
    int mc_shift = 0, mc_gx = 0, mc_gy = 0, mc_gz = 0;
    // a second, finer level (macro-cells of half the edge) when the LDS-resident grid is coarse (mc_shift >= 1): its `empty`
    // bits live in global memory and are consulted for fetches in cells the coarse level cannot rule out
    uint16_t* mm_fine = nullptr;
    uint16_t* mm_wide = nullptr;       // min/max per HALF-resolution macro-cell over a footprint one voxel wider per side (the fast bound look-up, svr_accel.hip k_bound8)
    int fg_x = 0, fg_y = 0, fg_z = 0;
    // environment map: sampling table of SVR_OPT_ENV_NEE (svr_kernels.hip launch_env_cdf), built with the texture
    float* env_cdf = nullptr;
    // transfer function: prefix count of exactly-zero alphas of the padded table, and an edit counter
    uint32_t* zero_prefix = nullptr;
    uint64_t version = 0;
};
constexpr uint32_t TEX_MAGIC = 0x53565254u;   // "SVRT"

struct Context {
    bool inited = false;
    int device = -1;
    hipStream_t stream = nullptr;
    int num_cus = 256;
    std::string info;
    // scene (the reference's __constant__ globals)
    bool have_vol = false, have_tf = false, have_cam = false;
    svr_volume vol{};
    svr_transfer_function tf{};
    svr_camera cam{};
    svr_environment_light env{};
    uint32_t num_lights = 0;
    svr_area_light lights[SVR_MAX_LIGHT_SOURCES]{};
    // options
    int opt_env_on_escape = 0, opt_kernel = 0, opt_count = 0, opt_timing = 0, opt_skip_tonemap = 0, opt_blocks_per_cu = 0;
    // sharding
    uint32_t strip_rows = 0, rank = 0, world = 1;
    int wx0 = 0, wy0 = 0, wx1 = -1, wy1 = -1;
    // device scratch
    unsigned long long* d_counters = nullptr;
    uint32_t* d_ticket = nullptr;
    uint32_t* d_queue = nullptr;        // scatter-record queues of the tile kernel's folding launches (they run one at a time)
    float* d_pend = nullptr;            // ... and the waves' pending-radiance rows
    uint32_t queue_blocks = 0;          // blocks both are sized for
    bool queue_alloc_failed = false;    // SVR_OPT_QUEUE = 1 and the device had no room for them: straight-line launches
    hipEvent_t prev_traced = nullptr;   // `traced` event of the latest trace launch (owned by its set)
    uint32_t* d_split_pool = nullptr;   // chunks of path records of the split kernels of deeper paths (svr_trace_split.hip)
    uint64_t split_chunks = 0;
    bool split_alloc_failed = false;
    hipEvent_t queue_done = nullptr;    // recorded behind every launch that uses d_queue / d_pend (there is ONE such memory: its users run one after the other, on whatever stream)
    bool queue_used = false;
    // frames traced ahead of the host's render_pathtracer calls (render_frames)
    struct Ahead {
        bool valid = false;
        svr::DevScene scene;
        svr::DevWork shape;             // window / shard of the launch
        uint64_t content = 0;
        uint32_t first = 0, count = 0, depth = 0;
        int set = -1;
    } ahead[2];                         // the batch being consumed and the one traced while it is consumed
    uint64_t content_version = 0;       // bumped whenever texture contents or handles change
    int opt_frame_ahead = 1;
    // Frame pipelining: a render call is cut into groups of <= GROUP frames; each group is traced on
    // one of NSETS internal streams into that set's scratch slots and resolved (running mean + tone map)
    // on the caller's stream, so the trace kernels of consecutive frames/calls overlap on the GPU while
    // the accumulator is still updated strictly in frame order.
#ifndef SVR_GROUP
#define SVR_GROUP 32     // frames per trace launch: 8 / 16 / 32 / 64 measured 0.185 / 0.178 / 0.166 / 0.161 ms per frame on c3
#endif
    static constexpr int NSETS = 4, GROUP = SVR_GROUP;
    static constexpr int AHEAD_MAX = 64;                     // frames per launch of the steady state of frame-ahead tracing (a wave = one pixel x 64 frames, like a folding launch)
    static constexpr uint64_t CHAIN_MIN_PATHS = 12u << 20;   // paths of a trace launch from which launches are chained
    struct SlotSet {
        float* lbuf = nullptr;
        hipStream_t stream = nullptr;
        hipEvent_t traced = nullptr, resolved = nullptr;
        bool used = false;
        float4* planes[svr::WF_QUEUE_PLANES] = {};   // wavefront queues (allocated on first use)
        uint32_t* wf_counts = nullptr;
    } sets[NSETS];
    size_t queue_capacity = 0;
    size_t slot_floats = 0;        // floats per scratch slot currently allocated (3*W*H)
    uint32_t slots_per_set = 0;    // slots currently allocated per set
    int next_set = 0;
    int opt_pipeline = 1, opt_refill = 16, opt_empty_skip = 1, opt_ray_skip = 1, opt_debug_stop = 0, opt_frames_log2 = -1, opt_unit = 0, opt_rc_lanes = 3, opt_bound_cull = 1, opt_park_end = 32, opt_fold = 1, opt_queue = 1, opt_fast_math = 0, opt_fine_mask = 0, opt_row_order = 0, opt_group_frames = 64, opt_local_majorant = 0, opt_light_cull = 1, opt_lm_tune = 0, opt_lm_sub = 1, opt_park_cheap = 16, opt_pinhole_fast = 1, opt_pool = 1, opt_trips = 1, opt_nan_guard = 0, opt_macro_shift_min = 0, opt_split = 0, opt_env_nee = 0, opt_fast_bound = 1;
    // empty-space bitmask of the current (volume, transfer function, densityScale)
    uint32_t* d_mask = nullptr;
    uint32_t* d_fine_mask = nullptr;   // `empty` bits of the fine level (global memory), sized for the current volume
    size_t fine_mask_words = 0;
    uint8_t* d_bnd8 = nullptr;         // byte table of the fast bound look-up (svr::BOUND8_BYTES)
    uint8_t* d_sub8 = nullptr;         // occupancy of the fine cells inside every macro-cell (local-majorant walks), MASK_WORDS_MAX * 32 bytes
    bool sub8_valid = false;
    bool bnd8_valid = false;
    uint8_t* d_mask_tmp = nullptr;     // scratch of the distance transform
    bool mask_valid = false;
    uint64_t mask_vol = 0, mask_tf = 0, mask_tf_version = 0;
    uint32_t mask_ds_bits = 0, mask_sigma_bits = 0;
    bool mask_cull_useful = false, mask_has_empty = true;
    uint32_t mask_words = 0;
    // ring of HIP event pairs around the path-tracing kernel (SVR_OPT_TIMING); drained lazily so the
    // timed launches never synchronise with the host
    static constexpr int EV_RING = 512;
    hipEvent_t ev0[EV_RING] = {}, ev1[EV_RING] = {};
    int ev_head = 0, ev_count = 0;
    double kernel_ms = 0.0;
    uint64_t kernel_launches = 0;
    // textures
    std::unordered_map<uint64_t, Texture*> textures;
    // errors
    int fatal = 1;
    int err_code = 0;
    std::string err_msg;
};

Context g;
std::mutex g_mu;
float* g_stage = nullptr;          // svr_assemble_frame: packed rows -- this rank's, and on root every rank's
size_t g_stage_floats = 0;

int fail(int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g.err_code = code ? code : -1;
    g.err_msg = buf;
    if (g.fatal) {
        // utils/helper_cuda.h:966-977 behaviour: report and die
        fprintf(stderr, "libsvr_hip error %d: %s\n", g.err_code, buf);
        hipDeviceReset();
        exit(EXIT_FAILURE);
    }
    return g.err_code;
}

// record an error without the fatal-mode exit: misuse that cannot have corrupted anything (e.g. destroying a
// handle twice from a copied host object)
int soft_fail(int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g.err_code = code ? code : -1;
    g.err_msg = buf;
    if (g.fatal) fprintf(stderr, "libsvr_hip warning %d: %s\n", g.err_code, buf);
    return g.err_code;
}

#define HIP_TRY(expr)                                                                      \
    do {                                                                                   \
        hipError_t _e = (expr);                                                            \
        if (_e != hipSuccess)                                                              \
            return fail((int)_e, "HIP error at %s:%d code=%d(%s) \"%s\"", __FILE__, __LINE__, \
                        (int)_e, hipGetErrorName(_e), #expr);                              \
    } while (0)

int ensure_init()
{
    if (g.inited) return 0;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return fail((int)e, "no HIP device available (hipGetDevice: %s)", hipGetErrorName(e));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, dev));
    g.device = dev;
    g.num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    char buf[256];
    snprintf(buf, sizeof buf, "%s %d.%d %s CUs=%d", prop.name, prop.major, prop.minor, prop.gcnArchName, g.num_cus);
    g.info = buf;
    HIP_TRY(hipMalloc((void**)&g.d_counters, sizeof(svr_counters) + DEBUG_WORDS * 8));       // + the phase profile of experiment builds
    HIP_TRY(hipMemset(g.d_counters, 0, sizeof(svr_counters) + DEBUG_WORDS * 8));
    HIP_TRY(hipMalloc((void**)&g.d_ticket, sizeof(uint32_t) * (svr::TICKET_SHARDS * svr::TICKET_STRIDE * (Context::NSETS + 1) + 64)));       // (+ the chunk counters of the split kernels behind the last set of task counters)
    HIP_TRY(hipMemset(g.d_ticket, 0, sizeof(uint32_t) * (svr::TICKET_SHARDS * svr::TICKET_STRIDE * (Context::NSETS + 1) + 64)));
    HIP_TRY(hipMalloc((void**)&g.d_mask, svr::ACCEL_WORDS * sizeof(uint32_t)));
    HIP_TRY(hipMalloc((void**)&g.d_mask_tmp, 2 * (size_t)svr::MASK_WORDS_MAX * 32));
    for (int i = 0; i < Context::NSETS; ++i) {
        HIP_TRY(hipStreamCreateWithFlags(&g.sets[i].stream, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&g.sets[i].traced, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&g.sets[i].resolved, hipEventDisableTiming));
    }
    HIP_TRY(hipEventCreateWithFlags(&g.queue_done, hipEventDisableTiming));
    for (int i = 0; i < Context::EV_RING; ++i) {
        HIP_TRY(hipEventCreate(&g.ev0[i]));
        HIP_TRY(hipEventCreate(&g.ev1[i]));
    }
    g.inited = true;
    return 0;
}

Texture* find_tex(uint64_t h, int kind)
{
    auto it = g.textures.find(h);
    if (it == g.textures.end()) return nullptr;
    Texture* t = it->second;
    if (t->magic != TEX_MAGIC || t->kind != kind) return nullptr;
    return t;
}

bool finite3(const svr_vec3& v) { return std::isfinite(v.x) && std::isfinite(v.y) && std::isfinite(v.z); }

// Build the device scene from the PODs.  All derived values use the same float operations, in
// the same order, as the reference's device code would (they are part of the numeric contract).
int build_scene(const svr_volume& vol, const svr_transfer_function& tf, const svr_camera& cam,
                svr::DevScene& s)
{
    memset(&s, 0, sizeof s);
    Texture* tv = find_tex(vol.tex, TEX_VOLUME);
    if (!tv) return fail(-2, "cudaVolume.tex (0x%llx) is not a live volume texture handle", (unsigned long long)vol.tex);
    Texture* tt = find_tex(tf.tex, TEX_TF);
    if (!tt) return fail(-2, "cudaTransferFunction.tex (0x%llx) is not a live transfer-function handle", (unsigned long long)tf.tex);
    if (!(tf.maxOpacity > 0.f) || !std::isfinite(tf.maxOpacity))
        return fail(-3, "transfer function maxOpacity (Woodcock majorant) must be finite and > 0, got %g", (double)tf.maxOpacity);
    if (!finite3(vol.bbox.vmin) || !finite3(vol.bbox.vmax) || !finite3(vol.bbox.invSize) || !finite3(vol.spacing) ||
        !finite3(vol.invSpacing) || !std::isfinite(vol.densityScale))
        return fail(-3, "cudaVolume has non-finite fields");
    if (cam.imageW == 0 || cam.imageH == 0) return fail(-3, "camera image size is zero");

    s.vmin[0] = vol.bbox.vmin.x; s.vmin[1] = vol.bbox.vmin.y; s.vmin[2] = vol.bbox.vmin.z;
    s.invSize[0] = vol.bbox.invSize.x; s.invSize[1] = vol.bbox.invSize.y; s.invSize[2] = vol.bbox.invSize.z;
    // cuda_bbox.h:38-39
    s.clip_vmin[0] = vol.bbox.vmin.x * (-vol.x_clip.x);
    s.clip_vmin[1] = vol.bbox.vmin.y * (-vol.y_clip.x);
    s.clip_vmin[2] = vol.bbox.vmin.z * (-vol.z_clip.x);
    s.clip_vmax[0] = vol.bbox.vmax.x * vol.x_clip.y;
    s.clip_vmax[1] = vol.bbox.vmax.y * vol.y_clip.y;
    s.clip_vmax[2] = vol.bbox.vmax.z * vol.z_clip.y;
    s.densityScale = vol.densityScale;
    s.invMaxMagnitude = vol.invMaxMagnitude;
    s.gradientFactor = vol.gradientFactor;
    {
        volatile float c = -25.f * vol.gradientFactor;     // pathtracer.cu:251, left to right
        c = c * vol.gradientFactor;
        c = c * vol.gradientFactor;
        s.pbrdf_c = c;
    }
    s.spacing[0] = vol.spacing.x; s.spacing[1] = vol.spacing.y; s.spacing[2] = vol.spacing.z;
    s.invSpacing[0] = vol.invSpacing.x; s.invSpacing[1] = vol.invSpacing.y; s.invSpacing[2] = vol.invSpacing.z;
    s.vox = (const uint16_t*)tv->data;
    s.nx = tv->nx; s.ny = tv->ny; s.nz = tv->nz;
    s.layout = tv->layout;
    s.fnx = (float)tv->nx; s.fny = (float)tv->ny; s.fnz = (float)tv->nz;
    s.sy = tv->sy; s.sz = tv->sz; s.bnx = tv->bnx; s.bny = tv->bny;

    s.tf = (const float*)tt->data;
    s.tf_n = tt->nx;
    s.tf_nf = (float)tt->nx;
    s.sigmaMax = tf.maxOpacity;
    {
        volatile float a = 1.f / tf.maxOpacity;            // woodcock_tracking.h:30
        volatile float b = tf.maxOpacity * 1.f;            // BASE_SAMPLE_STEP_SIZE 1.f, :18
        volatile float c2 = 1.f / b;                       // :31
        s.invSigmaMax = a;
        s.invSigmaMaxSI = c2;
    }

    s.imageW = cam.imageW; s.imageH = cam.imageH;
    s.exposure = cam.exposure; s.apeture = cam.apeture; s.focalLength = cam.focalLength;
    s.aspectRatio = cam.aspectRatio; s.tanFovxOverTwo = cam.tanFovxOverTwo;
    s.wm1 = (float)cam.imageW - 1.f;
    s.hm1 = (float)cam.imageH - 1.f;
    {
        // cudaCamera::GenerateRay with apeture == +0 (the reference's default, gui/canvas.h:218): lens sample = (cos(theta) * 0, sin(theta) * 0)
        // = (+-0, +-0).  orig = pos + u * (+-0) + v * (+-0) is pos bit for bit unless a component of pos is -0 (-0 + +0 = +0); the image-plane
        // coordinates are +0 or non-zero when their scales are positive, so subtracting +-0 changes nothing.  Then the kernels may skip the
        // sqrt and the sincos of the lens sample (the two draws are still consumed).
        uint32_t ap_bits, p_bits[3];
        memcpy(&ap_bits, &cam.apeture, 4);
        memcpy(p_bits, &cam.pos, 12);
        const float scale_x = cam.aspectRatio * cam.tanFovxOverTwo * cam.focalLength, scale_y = cam.tanFovxOverTwo * cam.focalLength;
        s.cam_pinhole = (g.opt_pinhole_fast && ap_bits == 0u && p_bits[0] != 0x80000000u && p_bits[1] != 0x80000000u && p_bits[2] != 0x80000000u &&
                         cam.aspectRatio > 1e-20f && cam.tanFovxOverTwo > 1e-20f && cam.focalLength > 1e-20f && scale_x > 1e-20f && scale_y > 1e-20f &&
                         std::isfinite(scale_x) && std::isfinite(scale_y)) ? 1u : 0u;
    }
    s.cam_pos[0] = cam.pos.x; s.cam_pos[1] = cam.pos.y; s.cam_pos[2] = cam.pos.z;
    s.cam_u[0] = cam.u.x; s.cam_u[1] = cam.u.y; s.cam_u[2] = cam.u.z;
    s.cam_v[0] = cam.v.x; s.cam_v[1] = cam.v.y; s.cam_v[2] = cam.v.z;
    s.cam_w[0] = cam.w.x; s.cam_w[1] = cam.w.y; s.cam_w[2] = cam.w.z;
    return 0;
}

// Can a camera ray (cudaCamera::GenerateRay, core/cuda_camera.h:66-83) reach the disk?  Conservative: true unless the disk's
// bounding sphere lies behind the lens plane or outside the view frustum widened by the lens.  A ray leaves the lens point a
// (|a| <= aperture) towards the image-plane point n at depth f = focalLength; at depth z > 0 its lateral offset is
// a (1 - z / f) + n z / f, with |n_x| <= (1 + 2 / (W - 1)) aspect tan(fov / 2) f (pixel jitter included).
bool light_reachable_by_camera_rays(const svr::DevScene& s, const svr_area_light& l)
{
    const double f = s.focalLength, A = std::fabs((double)s.apeture);
    if (!(f > 0.0) || !std::isfinite(f) || !std::isfinite(A) || s.imageW < 2 || s.imageH < 2) return true;
    const double u[3] = {s.cam_u[0], s.cam_u[1], s.cam_u[2]}, v[3] = {s.cam_v[0], s.cam_v[1], s.cam_v[2]}, w[3] = {s.cam_w[0], s.cam_w[1], s.cam_w[2]};
    auto dot3 = [](const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; };
    // the argument needs an orthonormal camera frame (cudaCamera::Setup builds one); anything else: no culling
    if (std::fabs(dot3(u, u) - 1.0) > 1e-3 || std::fabs(dot3(v, v) - 1.0) > 1e-3 || std::fabs(dot3(w, w) - 1.0) > 1e-3 ||
        std::fabs(dot3(u, v)) > 1e-3 || std::fabs(dot3(u, w)) > 1e-3 || std::fabs(dot3(v, w)) > 1e-3) return true;
    const double c[3] = {(double)l.disk.center.x - s.cam_pos[0], (double)l.disk.center.y - s.cam_pos[1], (double)l.disk.center.z - s.cam_pos[2]};
    const double r = std::fabs((double)l.disk.radius) * 1.01 + 1e-3;
    const double cx = dot3(c, u), cy = dot3(c, v), cz = -dot3(c, w);
    if (!std::isfinite(cx) || !std::isfinite(cy) || !std::isfinite(cz) || !std::isfinite(r)) return true;
    if (cz + r <= 0.0) return false;                                        // behind the lens plane: rays only go forward (dir . -w = f > 0)
    const double z0 = cz - r > 0.0 ? cz - r : 0.0, z1 = cz + r;
    const double k = std::fmax(std::fabs(1.0 - z0 / f), std::fabs(1.0 - z1 / f));
    const double tanh_ = std::fabs((double)s.tanFovxOverTwo);
    const double nx = (1.0 + 2.0 / ((double)s.imageW - 1.0)) * std::fabs((double)s.aspectRatio) * tanh_ * f;
    const double ny = (1.0 + 2.0 / ((double)s.imageH - 1.0)) * tanh_ * f;
    const double bx = (A * k + nx * z1 / f) * 1.01 + 1e-3, by = (A * k + ny * z1 / f) * 1.01 + 1e-3;
    if (!std::isfinite(bx) || !std::isfinite(by)) return true;
    return std::fabs(cx) - r <= bx && std::fabs(cy) - r <= by;
}

int add_lights_env(svr::DevScene& s)
{
    s.env = nullptr;
    if (g.env.tex != 0) {
        Texture* te = find_tex(g.env.tex, TEX_ENV);
        if (!te) return fail(-2, "cudaEnvironmentLight.tex (0x%llx) is not a live environment texture handle", (unsigned long long)g.env.tex);
        s.env = (const float*)te->data;
        s.env_w = te->nx; s.env_h = te->ny;
        s.env_cdf = te->env_cdf;
    }
    s.env_default[0] = g.env.defaultRadiance.x; s.env_default[1] = g.env.defaultRadiance.y; s.env_default[2] = g.env.defaultRadiance.z;
    s.env_intensity = g.env.intensity;
    s.env_offset[0] = g.env.offset.x; s.env_offset[1] = g.env.offset.y;
    s.env_on_escape = (uint32_t)g.opt_env_on_escape;
    s.num_lights = g.num_lights;
    s.primary_light_mask = 0u;
    for (uint32_t i = 0; i < g.num_lights; ++i) {
        const svr_area_light& l = g.lights[i];
        svr::DevLight& d = s.lights[i];
        if (!g.opt_light_cull || light_reachable_by_camera_rays(s, l)) s.primary_light_mask |= 1u << i;
        d.radius = l.disk.radius;
        d.center[0] = l.disk.center.x; d.center[1] = l.disk.center.y; d.center[2] = l.disk.center.z;
        d.normal[0] = l.disk.normal.x; d.normal[1] = l.disk.normal.y; d.normal[2] = l.disk.normal.z;
        // cuda_disk.h:53-56: M_PI * radius * radius (double) -> float
        volatile float area = (float)(3.14159265358979323846 * (double)l.disk.radius * (double)l.disk.radius);
        d.area = area;
        // cuda_arealight.h:57: 500.f * color * intensity * float(M_1_PI) / area, left to right per component
        const float c3[3] = {l.color.x, l.color.y, l.color.z};
        for (int c = 0; c < 3; ++c) {
            volatile float r = 500.f * c3[c];
            r = r * l.intensity;
            r = r * (float)0.31830988618379067154;
            r = r / area;
            d.radiance[c] = r;
        }
    }
    return 0;
}

int fill_work(svr::DevWork& w, uint32_t W, uint32_t H)
{
    memset(&w, 0, sizeof w);
    w.counters = g.d_counters;
    w.ticket = g.d_ticket;
    w.refill_min_idle = (uint32_t)g.opt_refill;
    w.debug_stop = (uint32_t)g.opt_debug_stop;
    w.row_order = (uint32_t)g.opt_row_order;
    w.nan_guard = (uint32_t)g.opt_nan_guard;
    w.strip_rows = g.strip_rows ? g.strip_rows : 1;
    w.rank = g.rank;
    w.world = g.world;
    if (g.world > 1) {
        // interleaved strips over the full frame
        w.x0 = 0; w.x1 = W; w.y0 = 0; w.y1 = H;
        uint32_t rows = 0;
        uint32_t nstrips = (H + w.strip_rows - 1) / w.strip_rows;
        for (uint32_t sidx = g.rank; sidx < nstrips; sidx += g.world) {
            uint32_t ys = sidx * w.strip_rows;
            uint32_t ye = ys + w.strip_rows < H ? ys + w.strip_rows : H;
            rows += ye - ys;
        }
        // the strip->row formula in the kernels needs whole strips except possibly the last owned one
        w.n_rows = rows;
    } else {
        int x0 = g.wx0 < 0 ? 0 : g.wx0, y0 = g.wy0 < 0 ? 0 : g.wy0;
        int x1 = (g.wx1 < 0 || g.wx1 > (int)W) ? (int)W : g.wx1;
        int y1 = (g.wy1 < 0 || g.wy1 > (int)H) ? (int)H : g.wy1;
        if (x0 > x1) x0 = x1;
        if (y0 > y1) y0 = y1;
        w.x0 = (uint32_t)x0; w.x1 = (uint32_t)x1; w.y0 = (uint32_t)y0; w.y1 = (uint32_t)y1;
        w.n_rows = w.y1 - w.y0;
    }
    w.n_items = w.n_rows * (w.x1 - w.x0);
    return 0;
}

// the whole frame, whatever shard or window is set
void fill_work_full(svr::DevWork& w, uint32_t W, uint32_t H)
{
    memset(&w, 0, sizeof w);
    w.counters = g.d_counters;
    w.ticket = g.d_ticket;
    w.strip_rows = 1; w.rank = 0; w.world = 1;
    w.x0 = 0; w.x1 = W; w.y0 = 0; w.y1 = H;
    w.n_rows = H;
    w.n_items = W * H;
}

bool partial_frame(uint32_t W, uint32_t H)
{
    svr::DevWork w;
    fill_work(w, W, H);
    return w.n_items != W * H;
}

void collect_timing()
{
    while (g.ev_count > 0) {
        int i = (g.ev_head - g.ev_count + 2 * Context::EV_RING) % Context::EV_RING;
        float ms = 0.f;
        if (hipEventSynchronize(g.ev1[i]) == hipSuccess && hipEventElapsedTime(&ms, g.ev0[i], g.ev1[i]) == hipSuccess) {
            g.kernel_ms += (double)ms;
            g.kernel_launches += 1;
        }
        g.ev_count--;
    }
}

// (Re)build the empty-space bitmask when the volume, the transfer-function table or densityScale
// changed.  Scene edits are rare (UI events), so the rebuild simply drains the device first.
int ensure_mask(svr::DevScene& s, const svr_volume& vol, const svr_transfer_function& tf)
{
    s.empty_mask = nullptr;
    if (!g.opt_empty_skip) return 0;
    Texture* tv = find_tex(vol.tex, TEX_VOLUME);
    Texture* tt = find_tex(tf.tex, TEX_TF);
    if (!tv || !tt || !tv->mm || !tt->zero_prefix) return 0;
    uint32_t ds_bits, sg_bits;
    memcpy(&ds_bits, &vol.densityScale, 4);
    memcpy(&sg_bits, &tf.maxOpacity, 4);
    uint32_t n_cells = (uint32_t)tv->mc_gx * (uint32_t)tv->mc_gy * (uint32_t)tv->mc_gz;
    uint32_t words = (n_cells + 31u) / 32u;
    if (!(g.mask_valid && g.mask_vol == vol.tex && g.mask_tf == tf.tex && g.mask_tf_version == tt->version &&
          g.mask_ds_bits == ds_bits && g.mask_sigma_bits == sg_bits)) {
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(svr::launch_empty_mask(tv->mm, tv->mc_gx, tv->mc_gy, tv->mc_gz, tt->zero_prefix, tt->nx,
                                       vol.densityScale, g.d_mask, words, g.d_mask_tmp, g.stream));
        if (tv->mm_fine) {
            const size_t fcells = (size_t)tv->fg_x * tv->fg_y * tv->fg_z, fwords = (fcells + 31) / 32;
            if (fwords > g.fine_mask_words) {
                if (g.d_fine_mask) HIP_TRY(hipFree(g.d_fine_mask));
                g.d_fine_mask = nullptr; g.fine_mask_words = 0;
                HIP_TRY(hipMalloc((void**)&g.d_fine_mask, fwords * sizeof(uint32_t)));
                g.fine_mask_words = fwords;
            }
            HIP_TRY(svr::launch_fine_mask(tv->mm_fine, (uint32_t)fcells, tt->zero_prefix, tt->nx, vol.densityScale, g.d_fine_mask, (uint32_t)fwords, g.stream));
            if (!g.d_sub8) HIP_TRY(hipMalloc((void**)&g.d_sub8, (size_t)svr::MASK_WORDS_MAX * 32));
            HIP_TRY(svr::launch_sub8(g.d_fine_mask, tv->fg_x, tv->fg_y, tv->fg_z, tv->mc_gx, tv->mc_gy, tv->mc_gz, g.d_sub8, g.stream));
        }
        g.sub8_valid = tv->mm_fine != nullptr;
        HIP_TRY(svr::launch_bound_class(tv->mm, tv->mc_gx, tv->mc_gy, tv->mc_gz, (const float*)tt->data, tt->nx, vol.densityScale,
                                        s.invSigmaMax, g.d_mask, g.stream));
        g.bnd8_valid = false;
        if (tv->mm_wide) {
            if (!g.d_bnd8) HIP_TRY(hipMalloc((void**)&g.d_bnd8, svr::BOUND8_BYTES));
            HIP_TRY(svr::launch_bound8(tv->mm_wide, (tv->mc_gx + 1) / 2, (tv->mc_gy + 1) / 2, (tv->mc_gz + 1) / 2, (const float*)tt->data, tt->nx, vol.densityScale,
                                       s.invSigmaMax, g.d_bnd8, g.stream));
            g.bnd8_valid = true;
        }
        HIP_TRY(hipStreamSynchronize(g.stream));
        // the bound test costs two dependent LDS reads per tested cell (4 % on a scene where it never rejects anything): use
        // it where at least 2 % of the coarse cells that can hold a collision have a bound below 1
        uint32_t census[3] = {0u, 0u, 0u};
        HIP_TRY(hipMemcpy(census, g.d_mask + svr::ACCEL_CENSUS_OFF, sizeof census, hipMemcpyDeviceToHost));
        g.mask_has_empty = census[2] != 0u;
        g.mask_cull_useful = (uint64_t)census[0] * 50u >= (uint64_t)census[0] + census[1] && census[0] != 0u;
        g.mask_valid = true; g.mask_vol = vol.tex; g.mask_tf = tf.tex; g.mask_tf_version = tt->version;
        g.mask_ds_bits = ds_bits; g.mask_sigma_bits = sg_bits; g.mask_words = words;
    }
    s.empty_mask = g.d_mask;
    s.mask_words = g.mask_words;
    {
        const int hgx = (tv->mc_gx + 1) / 2, hgy = (tv->mc_gy + 1) / 2, hgz = (tv->mc_gz + 1) / 2;
        s.mc_hgx = hgx; s.mc_hgxy = hgx * hgy;
        s.dist_words = ((uint32_t)hgx * (uint32_t)hgy * (uint32_t)hgz + 7u) / 8u;
    }
    s.mc_shift = tv->mc_shift;
    s.mc_gx = tv->mc_gx; s.mc_gy = tv->mc_gy; s.mc_gz = tv->mc_gz;
    s.mc_gxy = tv->mc_gx * tv->mc_gy;
    // macro-grid coordinate of a world point: ((p - vmin) * invSize * N + 0.5) / S  (cell c' = c + 1)
    float invS = 1.f / (float)(1 << tv->mc_shift);
    s.mc_scale[0] = s.invSize[0] * s.fnx * invS;
    s.mc_scale[1] = s.invSize[1] * s.fny * invS;
    s.mc_scale[2] = s.invSize[2] * s.fnz * invS;
    s.mc_off = 0.5f * invS;
    // whole-ray tests need every fetch of a walk inside the texture domain: clipped box within the bbox
    bool inside = true;
    const float lo[3] = {vol.bbox.vmin.x, vol.bbox.vmin.y, vol.bbox.vmin.z};
    const float hi[3] = {vol.bbox.vmax.x, vol.bbox.vmax.y, vol.bbox.vmax.z};
    for (int a = 0; a < 3; ++a) {
        float cl = s.clip_vmin[a] < s.clip_vmax[a] ? s.clip_vmin[a] : s.clip_vmax[a];
        float ch = s.clip_vmin[a] < s.clip_vmax[a] ? s.clip_vmax[a] : s.clip_vmin[a];
        inside = inside && cl >= lo[a] && ch <= hi[a] && lo[a] < hi[a];
    }
    s.ray_skip = (inside && g.opt_ray_skip) ? 1u : 0u;
    // the fine level costs a dependent (cached) global load per tested cell, which is what it saves: c5 fetches 3.9 instead of 5.9
    // taps per path at 5.18 instead of 5.27 Gsamples/s, c3 3.0 instead of 4.0 at 7.89 instead of 8.06 -> off by default (1 = auto
    // would turn it on for cells >= 16 voxels)
    s.fine_mask = (tv->mm_fine && g.d_fine_mask && (g.opt_fine_mask == 2 || (g.opt_fine_mask == 1 && tv->mc_shift >= 4))) ? g.d_fine_mask : nullptr;
    s.fg_x = tv->fg_x; s.fg_y = tv->fg_y; s.fg_z = tv->fg_z; s.fg_xy = tv->fg_x * tv->fg_y;
    // sub-cell occupancy pays where macro-cells are large (c5, 16 voxels: +6 %) and costs where they are small (c3, 8 voxels: -4 %)
    s.sub8 = (g.sub8_valid && g.d_sub8 && (g.opt_lm_sub == 2 || (g.opt_lm_sub == 1 && tv->mc_shift >= 4))) ? g.d_sub8 : nullptr;
    s.has_empty = g.mask_has_empty ? 1u : 0u;
    s.bound_cull = (g.opt_bound_cull == 2 || (g.opt_bound_cull == 1 && g.mask_cull_useful)) ? 1u : 0u;
    s.park_end = (uint32_t)g.opt_park_end;
    s.park_cheap = (uint32_t)g.opt_park_cheap;
    // pool tuning (same-box sweeps, gpurun_out/r03p_lm_tune.log): on the full-resolution grid (scenes with empty space) one cell per turn
    // (c3 9 920 / c5 8 430 Msamples/s against 9 480 / 7 630 with three), on the half-resolution grid of fog-like media three cells
    // and later refills (c3n 4 860 against 4 290 with one cell)
    // tasks per batch of the traceDepth-1 pool (bits 24-31; same-box A/B, 10 / 16 / 21 tasks: c3 10 215 / 10 064 / 9 821, c5 8 611 / 8 383 / 8 173, c3n 4 747 / 4 932 / 4 972)
    s.lm_tune = g.opt_lm_tune ? (uint32_t)g.opt_lm_tune : (g.mask_has_empty ? (1u | (16u << 8) | (16u << 16) | (10u << 24)) : (3u | (24u << 8) | (24u << 16) | (21u << 24)));
    if ((s.lm_tune >> 24) == 0u) s.lm_tune |= (g.mask_has_empty ? 10u : 21u) << 24;
    // five-iteration trips of the lane machine: walks of tens of iterations with few fetches -- media without exactly transparent space under
    // bound culling (c3n: +20 %; c3 / c5 at depth 2: -1..2 %)
    s.trips = (g.opt_trips == 2 || (g.opt_trips == 1 && s.bound_cull && !s.has_empty)) ? 1u : 0u;
    // Fast bound look-up of the lane machine's trips (svr_lanes.hpp iterate_rot; svr_accel.hip k_bound8): media without exactly transparent space under
    // bound culling.  The look-up takes the macro-cell from one fma per axis on the ray parameter; the table covers one voxel of error, the float chains
    // differ by ~8 ulp of the largest intermediate -- |origin - volume| * N voxels * 2^-21 -- so the camera may be up to 2^21 / (16 N) volume extents
    // away (128 at N = 1024; scatter points and shadow rays start inside the volume) before the kernel falls back to the exact cell.  The look-up
    // does not clamp: the clipped box must lie inside the texture domain (`inside`), as for the whole-ray tests.
    s.bnd8 = nullptr;
    s.hc_scale[0] = 0.5f * s.mc_scale[0]; s.hc_scale[1] = 0.5f * s.mc_scale[1]; s.hc_scale[2] = 0.5f * s.mc_scale[2];
    s.hc_off = 0.5f * s.mc_off + 1.f;
    if (g.opt_fast_bound && g.bnd8_valid && s.bound_cull && !s.has_empty && s.fine_mask == nullptr && s.trips && inside) {
        bool near_enough = true;
        const int nmax = tv->nx > tv->ny ? (tv->nx > tv->nz ? tv->nx : tv->nz) : (tv->ny > tv->nz ? tv->ny : tv->nz);
        const double reach = 2097152.0 / (16.0 * (double)nmax);
        for (int a = 0; a < 3; ++a) {
            const double c = 0.5 * ((double)lo[a] + (double)hi[a]), ext = (double)hi[a] - (double)lo[a];
            const double d = ((double)s.cam_pos[a] - c) / ext;
            near_enough = near_enough && ext > 0.0 && d == d && (d < 0 ? -d : d) <= reach;
        }
        if (near_enough) s.bnd8 = g.d_bnd8;
    }
    return 0;
}

// scratch radiance slots: sized for the frame and for the largest group requested so far (a host that only
// ever calls render_pathtracer keeps 1 slot per set; svr_render_pathtracer_frames grows it up to GROUP)
int ensure_slots(uint32_t W, uint32_t H, uint32_t nslots)
{
    size_t need = (size_t)3 * W * H;
    if (g.slot_floats == need && g.slots_per_set >= nslots) return 0;
    HIP_TRY(hipDeviceSynchronize());
    g.ahead[0].valid = g.ahead[1].valid = false;
    for (auto& st : g.sets) {
        if (st.lbuf) { HIP_TRY(hipFree(st.lbuf)); st.lbuf = nullptr; }
        st.used = false;
    }
    if (g.slot_floats != need || nslots > g.slots_per_set) g.slots_per_set = nslots;
    for (auto& st : g.sets) HIP_TRY(hipMalloc((void**)&st.lbuf, need * sizeof(float) * g.slots_per_set));
    g.slot_floats = need;
    return 0;
}

// per-wave path-record queues (REC_WORDS x QUEUE_CAP words) and pending-radiance rows (QUEUE_TASKS x 3 x 64 floats) of the
// tile kernel's QUEUE builds (svr_trace_tile.hip, svr_lanes.hpp): 184 KB + 24 KB per wave, 0.85 GB for 256 blocks.
// soft: the caller can do without them (SVR_OPT_QUEUE = 1): running out of device memory is then not an error, the launches
// use the straight-line kernel and the allocation is not tried again until the option is set anew.
using svr::QUEUE_WORDS_PER_BLOCK;
using svr::PEND_FLOATS_PER_BLOCK;
int ensure_record_queues(uint32_t blocks, bool soft, bool& available)
{
    available = true;
    if (g.d_queue && g.d_pend && g.queue_blocks >= blocks) return 0;
    if (soft && g.queue_alloc_failed) { available = false; return 0; }
    HIP_TRY(hipDeviceSynchronize());
    if (g.d_queue) { HIP_TRY(hipFree(g.d_queue)); g.d_queue = nullptr; }
    if (g.d_pend) { HIP_TRY(hipFree(g.d_pend)); g.d_pend = nullptr; }
    g.queue_blocks = 0;
    hipError_t e = hipMalloc((void**)&g.d_queue, QUEUE_WORDS_PER_BLOCK * sizeof(uint32_t) * blocks);
    if (e == hipSuccess) e = hipMalloc((void**)&g.d_pend, PEND_FLOATS_PER_BLOCK * sizeof(float) * blocks);
    if (e != hipSuccess) {
        if (g.d_queue) { (void)hipFree(g.d_queue); g.d_queue = nullptr; }
        g.d_pend = nullptr;
        if (soft && e == hipErrorOutOfMemory) {
            (void)hipGetLastError();
            g.queue_alloc_failed = true;
            available = false;
            return 0;
        }
        return fail((int)e, "HIP error allocating the tile kernel's queue memory (%zu MB): %s", ((QUEUE_WORDS_PER_BLOCK + PEND_FLOATS_PER_BLOCK) * 4 * blocks) >> 20, hipGetErrorName(e));
    }
    g.queue_blocks = blocks;
    return 0;
}

// the chunk pool of the split kernels (svr_trace_split.hip): worst case of the largest launch so far; soft -- without room for it the fused kernel renders
int ensure_split_pool(uint64_t chunks, bool& available)
{
    available = true;
    if (g.d_split_pool && g.split_chunks >= chunks) return 0;
    if (g.split_alloc_failed) { available = false; return 0; }
    HIP_TRY(hipDeviceSynchronize());
    if (g.d_split_pool) { HIP_TRY(hipFree(g.d_split_pool)); g.d_split_pool = nullptr; g.split_chunks = 0; }
    hipError_t e = hipMalloc((void**)&g.d_split_pool, (size_t)chunks * svr::SPLIT_CHUNK_WORDS * sizeof(uint32_t));
    if (e != hipSuccess) {
        g.d_split_pool = nullptr;
        if (e == hipErrorOutOfMemory) { (void)hipGetLastError(); g.split_alloc_failed = true; available = false; return 0; }
        return fail((int)e, "HIP error allocating the record pool of the split kernels (%zu MB): %s", ((size_t)chunks * svr::SPLIT_CHUNK_WORDS * 4) >> 20, hipGetErrorName(e));
    }
    g.split_chunks = chunks;
    return 0;
}

int ensure_queues(uint32_t W, uint32_t H)
{
    size_t need = (size_t)W * H * Context::GROUP;
    if (g.queue_capacity == need) return 0;
    HIP_TRY(hipDeviceSynchronize());
    for (auto& st : g.sets) {
        for (auto& p : st.planes) if (p) { HIP_TRY(hipFree(p)); p = nullptr; }
        if (!st.wf_counts) HIP_TRY(hipMalloc((void**)&st.wf_counts, 64 * sizeof(uint32_t)));
    }
    if (need >= ((size_t)1 << 31)) return fail(-3, "frame too large for the wavefront queues");
    for (auto& st : g.sets)
        for (auto& p : st.planes) HIP_TRY(hipMalloc((void**)&p, need * sizeof(float4)));
    g.queue_capacity = need;
    return 0;
}

int render_frames(void* img, const svr_render_params* rp, uint32_t nframes, bool tonemap)
{
    if (ensure_init()) return g.err_code;
    if (!rp) return fail(-4, "render_pathtracer: renderParams is null");
    if (!g.have_vol || !g.have_tf || !g.have_cam)
        return fail(-4, "render_pathtracer before setup_volume/setup_transferfunction/setup_camera");
    if (!rp->hdrBuffer) return fail(-4, "render_pathtracer: renderParams.hdrBuffer is null (call SetupHDRBuffer)");
    if (nframes == 0) return 0;
    svr::DevScene s;
    if (build_scene(g.vol, g.tf, g.cam, s)) return g.err_code;
    if (add_lights_env(s)) return g.err_code;
    if ((size_t)3 * s.imageW * s.imageH >= ((size_t)1 << 32)) return fail(-3, "image too large");
    // clear_hdr_buffer zeroes the WHOLE accumulator at frame 0 (pathtracer.cu:86-94,297-300); under a row shard or a
    // window the kernels only touch their own pixels, so the rest is cleared here (the strips of the ranks are then
    // summed into one frame: stale values outside the owned rows would corrupt it)
    if (rp->frameNo == 0 && partial_frame(s.imageW, s.imageH))
        HIP_TRY(hipMemsetAsync(rp->hdrBuffer, 0, sizeof(float) * 3 * (size_t)s.imageW * s.imageH, g.stream));
    svr::LaunchCfg cfg;
    cfg.kernel = g.opt_kernel == svr::KERNEL_AUTO ? svr::KERNEL_TILE : g.opt_kernel;
    // OPT-IN importance sampling of the environment map (svr_trace_env.hip): where there is a map, the environment term is on and a bounce follows the
    // first scatter event (the env sample of event k pairs with the escape term of bounce k + 1)
    if (g.opt_env_nee && cfg.kernel == svr::KERNEL_TILE && s.env != nullptr && s.env_cdf != nullptr && s.env_on_escape && rp->traceDepth >= 2 && !g.opt_debug_stop)
        cfg.kernel = svr::KERNEL_ENV_NEE;
    if ((cfg.kernel == svr::KERNEL_TILE || cfg.kernel == svr::KERNEL_WAVEFRONT || cfg.kernel == svr::KERNEL_ENV_NEE) && ensure_mask(s, g.vol, g.tf)) return g.err_code;
    if (cfg.kernel == svr::KERNEL_WAVEFRONT) {
        if (rp->traceDepth > 15) return fail(-3, "the wavefront kernels support traceDepth <= 15 (got %u)", rp->traceDepth);
        if (ensure_queues(s.imageW, s.imageH)) return g.err_code;
    }
    cfg.count = g.opt_count != 0;
    cfg.num_cus = g.num_cus;
    cfg.blocks_per_cu = g.opt_blocks_per_cu > 0 ? g.opt_blocks_per_cu : 4;
    cfg.frames_log2 = g.opt_frames_log2;
    cfg.unit_override = g.opt_unit;
    cfg.lm_straight = g.opt_local_majorant == 2;
    cfg.pool_primary = false;            // (decided below, once the acceleration data says what kind of medium this is)
    // The tile kernel folds the frames of a launch into the accumulator itself (running mean in frame order, 12 B per
    // pixel per launch, svr_trace_tile.hip); the scratch slots + k_resolve remain for frames traced AHEAD of the calls
    // that ask for them (their radiance is folded later, one frame per call) and for the other kernels.
    const bool fold_batch = cfg.kernel == svr::KERNEL_TILE && g.opt_fold && !g.opt_debug_stop && cfg.frames_log2 < 0;
    // QUEUE builds (the hits of a task are shaded in place, then their paths continue on a per-lane state machine,
    // svr_lanes.hpp) ride on the folding launches.  Auto: with empty-space skipping on (without it every walk is long and the
    // straight-line code wins: c3, 1660 against 1452 Msamples/s), for deep paths and media without exactly transparent space
    // always (c3 depth 2 / 4: +10 % / +30 %, c3n: +34 %), otherwise when the launch gives every wave at least 4 drains of
    // QUEUE_TASKS tasks (c3 / c4 / c5: +4 % / +3 % / +12 %; c2, 512^2 pixels: -2 %)
    bool use_queue = fold_batch && g.opt_queue == 2;
    if (fold_batch && g.opt_queue == 1 && g.opt_empty_skip) {
        svr::DevWork wq;
        fill_work(wq, s.imageW, s.imageH);
        const uint64_t waves = (uint64_t)(cfg.num_cus * cfg.blocks_per_cu) / 4u * svr::TILE_WAVES;    // blocks of 1024 threads
        use_queue = rp->traceDepth >= 2 || s.bound_cull || (uint64_t)wq.n_items >= 4u * svr::QUEUE_TASKS * waves;
    }
    // OPT-IN local majorants (svr_trace_lm.hip): needs the class table and whole-ray validity; its folding launches keep the waves'
    // pending radiance in the rows next to the record queues
    const bool local_majorant = g.opt_local_majorant && cfg.kernel == svr::KERNEL_TILE && s.empty_mask != nullptr && s.ray_skip && !g.opt_debug_stop;
    // the slot-per-path pool of deeper paths settles its walks (slot loads and stores) and refills less eagerly: 2 cells per turn, a
    // refill from 32 idle lanes, settling from 48 ended walks (c3 depth 4: 3 390 Msamples/s against 2 630 with the depth-1 setting,
    // c3n depth 4: 1 080 against 810; gpurun_out/r03u_tune.log)
    if (local_majorant && !g.opt_lm_tune && rp->traceDepth > 1) s.lm_tune = 2u | (32u << 8) | (48u << 16) | (s.lm_tune & 0xff000000u);
    if (local_majorant) use_queue = false;
    // POOL (svr_trace_tile.hip): pooled primary walks pay where the walks of a wave are not coherent -- media without exactly transparent
    // space under bound culling (c3n) -- and cost where they are (c3): auto = such media only
    cfg.pool_primary = use_queue && (g.opt_pool == 2 || (g.opt_pool == 1 && s.bound_cull && !s.has_empty));
    if (use_queue || (local_majorant && fold_batch)) {          // (a frame-ahead call traces its batches with the same kernels)
        bool available = true;
        if (ensure_record_queues((uint32_t)(cfg.num_cus * cfg.blocks_per_cu) * 4u / 16u, g.opt_queue == 1 && !local_majorant, available)) return g.err_code;
        use_queue = use_queue && available;
    }
    const bool frame_ahead_possible = nframes == 1 && g.opt_frame_ahead && g.opt_pipeline && !g.opt_count && !g.opt_debug_stop && cfg.kernel == svr::KERNEL_TILE;
    // Deeper paths, bit-exact, as TWO kernels (svr_trace_split.hip: front half -> chunks of records -> lane machine; the launch's frames go through
    // the scratch slots and k_resolve): where the fused queue kernel would run and its primary walks are not pooled, the image fits a 26-bit pixel
    // index, and the device has room for the worst-case record pool of a launch (<= 24 GB; else the fused kernel)
    bool use_split = false;
    const uint32_t split_group = (uint32_t)g.opt_group_frames;
    // OPT-IN since the fused kernel caught up: the two changes that made the split machine fast (lights in LDS + laundered scene constants) also fix the fused
    // deeper kernel when applied TOGETHER (each alone costs it 2.5-8 %).  Same box, two kernels / fused: depth 2 5 048 / 5 120, depth 3 3 735 / 3 782,
    // depth 4 3 152 / 3 171, depth 6 2 844 / 2 876, c5 depth 2 3 306 / 3 333, c3b depth 4 1 846 / 1 830 -- and the fused form needs no 14-GB pool.
    // (Against the fused kernel as it was, the two-kernel form won 3-6 %: 5 054 / 4 893, 3 728 / 3 555, 3 161 / 2 988, 2 859 / 2 693.)
    if (g.opt_split && use_queue && rp->traceDepth >= 2u && rp->traceDepth < 32768u && !cfg.pool_primary && !local_majorant && !g.opt_fast_math && nframes >= 8 &&
        s.layout != svr::LAYOUT_LINEAR && (uint64_t)s.imageW * s.imageH <= (1ull << 26) && !frame_ahead_possible) {
        svr::DevWork wq;
        fill_work(wq, s.imageW, s.imageH);
        const uint32_t n_max = nframes < split_group ? nframes : split_group;
        uint32_t fl2 = 0;
        while ((1u << fl2) < n_max) ++fl2;
        const uint32_t P2 = 6u - fl2, tw2 = (P2 + 1u) >> 1, th2 = P2 >> 1;
        const uint64_t n_tasks = (uint64_t)((wq.x1 - wq.x0 + (1u << tw2) - 1u) >> tw2) * ((wq.n_rows + (1u << th2) - 1u) >> th2);
        const uint64_t chunks = svr::split_chunks_worst_case(n_tasks * 64u, (uint32_t)(cfg.num_cus * cfg.blocks_per_cu) * 4u);
        if (chunks * svr::SPLIT_CHUNK_WORDS * sizeof(uint32_t) <= (24ull << 30) && chunks < (1ull << 31)) {
            bool available = true;
            if (ensure_split_pool(chunks, available)) return g.err_code;
            use_split = available;
        }
        if (use_split && ensure_slots(s.imageW, s.imageH, n_max < (uint32_t)Context::GROUP ? (uint32_t)Context::GROUP : n_max)) return g.err_code;
    }
    auto launch_tile = [&](const svr::DevWork& w, hipStream_t st) -> hipError_t {
        if (local_majorant) return svr::launch_trace_lm(s, w, cfg, st);
        return g.opt_fast_math ? svr_fast::launch_trace_tile_raw(&s, &w, &cfg, st) : svr::launch_trace_tile(s, w, cfg, st);
    };
    const bool frame_ahead_call = frame_ahead_possible;
    // short launches (< FOLD_MIN frames) keep the slots: they end in a tail of a few long tasks that only overlapping launches
    // on several streams hide, and a folding launch cannot overlap its predecessor (measured, 1 frame per call: 0.276 vs 0.366 ms)
    constexpr uint32_t FOLD_MIN = 8;
    const uint32_t tail_frames = nframes % (uint32_t)(fold_batch ? g.opt_group_frames : Context::GROUP);
    if ((!fold_batch || frame_ahead_call || (tail_frames != 0 && tail_frames < FOLD_MIN)) &&
        ensure_slots(s.imageW, s.imageH, nframes < (uint32_t)Context::GROUP ? nframes : (uint32_t)Context::GROUP)) return g.err_code;
    // one folding launch: trace + accumulate on the caller's stream (the launches of a render update the same accumulator,
    // so they run in order anyway), tone map behind the last one
    auto trace_fold = [&](uint32_t first, uint32_t n, bool want_img) -> int {
        svr::DevWork w;
        fill_work(w, s.imageW, s.imageH);
        w.hdr = (float*)rp->hdrBuffer;
        w.img = want_img ? (uint8_t*)img : nullptr;
        w.ticket = g.d_ticket + (size_t)svr::TICKET_SHARDS * svr::TICKET_STRIDE * Context::NSETS;   // (the sets' own counters may be in use by frames traced ahead)
        w.traceDepth = rp->traceDepth;
        w.frame0 = first;
        w.nframes = n;
        w.fold = 1u;
        w.queue = (use_queue || local_majorant) ? g.d_queue : nullptr;
        w.pend = (use_queue || local_majorant) ? g.d_pend : nullptr;
        w.queue_blocks = g.queue_blocks;
        int slot = -1;
        if (g.opt_timing) {
            if (g.ev_count == Context::EV_RING) collect_timing();
            slot = g.ev_head;
            HIP_TRY(hipEventRecord(g.ev0[slot], g.stream));
        }
        if (w.queue != nullptr && g.queue_used) HIP_TRY(hipStreamWaitEvent(g.stream, g.queue_done, 0));
        HIP_TRY(launch_tile(w, g.stream));
        if (w.queue != nullptr) { HIP_TRY(hipEventRecord(g.queue_done, g.stream)); g.queue_used = true; }
        if (g.opt_timing) {
            HIP_TRY(hipEventRecord(g.ev1[slot], g.stream));
            g.ev_head = (g.ev_head + 1) % Context::EV_RING;
            g.ev_count++;
        }
        if (want_img) HIP_TRY(svr::launch_tonemap(s, w, g.stream));
        return 0;
    };
    // one trace launch of n_trace frames starting at frame `first` into scratch set `si`, then the resolve of its
    // first n_resolve frames
    auto trace_group = [&](int si, uint32_t first, uint32_t n_trace, uint32_t n_resolve, bool want_img, bool with_queue = false) -> int {
        Context::SlotSet& set = g.sets[si];
        for (auto& a : g.ahead) if (a.valid && a.set == si) a.valid = false;     // its slots are about to be overwritten
        hipStream_t ts = g.opt_pipeline ? set.stream : g.stream;
        svr::DevWork w;
        fill_work(w, s.imageW, s.imageH);
        w.hdr = (float*)rp->hdrBuffer;
        w.img = want_img ? (uint8_t*)img : nullptr;
        w.lbuf = set.lbuf;
        w.slot_stride = (uint32_t)g.slot_floats;
        w.ticket = g.d_ticket + (size_t)svr::TICKET_SHARDS * svr::TICKET_STRIDE * si;
        w.traceDepth = rp->traceDepth;
        w.frame0 = first;
        w.nframes = n_trace;
        if (with_queue) { w.queue = g.d_queue; w.pend = g.d_pend; w.queue_blocks = g.queue_blocks; }
        // the scratch slots of this set are free again once their previous resolve has run
        if (g.opt_pipeline && set.used) HIP_TRY(hipStreamWaitEvent(ts, set.resolved, 0));
        // A large trace launch fills the chip by itself; running two of them at once only makes them share L2
        // and stretches both.  They are chained (the resolve of one still overlaps the trace of the next).
        // Smaller launches (one frame per call; a GPU's share of the frame under row sharding) end in a ~0.1 ms
        // tail of a few long tasks and overlap freely to hide it.
        if (g.opt_pipeline && (uint64_t)w.n_items * n_trace >= Context::CHAIN_MIN_PATHS && g.prev_traced)
            HIP_TRY(hipStreamWaitEvent(ts, g.prev_traced, 0));
        int slot = -1;
        if (g.opt_timing) {
            if (g.ev_count == Context::EV_RING) collect_timing();
            slot = g.ev_head;
            HIP_TRY(hipEventRecord(g.ev0[slot], ts));
        }
        if (cfg.kernel == svr::KERNEL_WAVEFRONT)
            HIP_TRY(svr::launch_wavefront(s, w, cfg, set.planes, set.wf_counts, (uint32_t)g.queue_capacity, ts));
        else if (cfg.kernel == svr::KERNEL_ENV_NEE) HIP_TRY(svr::launch_trace_env(s, w, cfg, ts));
        else if (cfg.kernel == svr::KERNEL_TILE) {
            if (w.queue != nullptr && g.queue_used) HIP_TRY(hipStreamWaitEvent(ts, g.queue_done, 0));
            HIP_TRY(launch_tile(w, ts));
            if (w.queue != nullptr) { HIP_TRY(hipEventRecord(g.queue_done, ts)); g.queue_used = true; }
        } else HIP_TRY(svr::launch_pathtrace(s, w, cfg, ts));
        if (g.opt_timing) {
            HIP_TRY(hipEventRecord(g.ev1[slot], ts));
            g.ev_head = (g.ev_head + 1) % Context::EV_RING;
            g.ev_count++;
        }
        if (g.opt_pipeline) {
            HIP_TRY(hipEventRecord(set.traced, ts));
            g.prev_traced = set.traced;
            // (a batch traced purely AHEAD -- nothing of it is resolved by this call -- must not hold the caller's stream: the calls that
            // consume the batch in stock run beside it and wait for `traced` when they get to this one)
            if (n_resolve) HIP_TRY(hipStreamWaitEvent(g.stream, set.traced, 0));
        }
        if (n_resolve) {
            w.nframes = n_resolve;
            HIP_TRY(svr::launch_resolve(s, w, g.stream));
            if (g.opt_pipeline) HIP_TRY(hipEventRecord(set.resolved, g.stream));
        }
        set.used = true;
        if (n_resolve < n_trace) {
            // a free entry, else the older batch
            Context::Ahead& a = !g.ahead[0].valid ? g.ahead[0] : (!g.ahead[1].valid ? g.ahead[1] : (g.ahead[0].first < g.ahead[1].first ? g.ahead[0] : g.ahead[1]));
            a.valid = true; a.scene = s; a.shape = w; a.content = g.content_version;
            a.first = first; a.count = n_trace; a.depth = rp->traceDepth; a.set = si;
        }
        return 0;
    };
    auto next_set = [&]() { int si = g.next_set; g.next_set = (g.next_set + 1) % Context::NSETS; return si; };
    const bool want_img = tonemap && !g.opt_skip_tonemap;
    // one launch of the split kernels (n <= 64 frames of every owned pixel) + the fold of its scratch slots, all on the caller's stream
    auto trace_split = [&](uint32_t first, uint32_t n, bool img_now) -> int {
        const int si = next_set();
        Context::SlotSet& set = g.sets[si];
        for (auto& a : g.ahead) if (a.valid && a.set == si) a.valid = false;     // its slots are about to be overwritten
        if (set.used) HIP_TRY(hipStreamWaitEvent(g.stream, set.traced, 0));       // (a batch still being traced ahead into this set)
        svr::DevWork w;
        fill_work(w, s.imageW, s.imageH);
        w.hdr = (float*)rp->hdrBuffer;
        w.img = img_now ? (uint8_t*)img : nullptr;
        w.lbuf = set.lbuf;
        w.slot_stride = (uint32_t)g.slot_floats;
        w.ticket = g.d_ticket + (size_t)svr::TICKET_SHARDS * svr::TICKET_STRIDE * Context::NSETS;
        w.traceDepth = rp->traceDepth;
        w.frame0 = first;
        w.nframes = n;
        w.queue = g.d_split_pool;
        w.queue_blocks = (uint32_t)g.split_chunks;
        int slot = -1;
        if (g.opt_timing) {
            if (g.ev_count == Context::EV_RING) collect_timing();
            slot = g.ev_head;
            HIP_TRY(hipEventRecord(g.ev0[slot], g.stream));
        }
        HIP_TRY(svr::launch_trace_split(s, w, cfg, g.stream));
        if (g.opt_timing) {
            HIP_TRY(hipEventRecord(g.ev1[slot], g.stream));
            g.ev_head = (g.ev_head + 1) % Context::EV_RING;
            g.ev_count++;
        }
        HIP_TRY(svr::launch_resolve(s, w, g.stream));
        HIP_TRY(hipEventRecord(set.traced, g.stream));
        HIP_TRY(hipEventRecord(set.resolved, g.stream));
        set.used = true;
        return 0;
    };

    // The reference's host calls render_pathtracer once per frame (gui/canvas.cpp:96).  A frame's radiance is a pure
    // function of (scene, pixel, frame number), so frames can be traced AHEAD of the calls that ask for them: a
    // 32-frame launch costs 0.13 ms per frame on c3, a 1-frame launch 0.23.  The batch doubles with the frame number
    // (1, 2, 4 ... GROUP), so a host that restarts the render on every mouse event never traces more than twice what
    // it shows.  Anything that could change a frame -- scene PODs, texture contents, window, shard, trace depth --
    // invalidates the frames in stock.
    if (frame_ahead_call) {
        const uint32_t n = rp->frameNo;
        svr::DevWork shape;
        fill_work(shape, s.imageW, s.imageH);
        // at most 4 GB of scratch frames per set (64 frames up to ~2300^2 pixels; fewer for larger images)
        const uint64_t slot_bytes = (uint64_t)3 * s.imageW * s.imageH * sizeof(float);
        uint32_t batch_max = 1;
        while (batch_max < (uint32_t)Context::AHEAD_MAX && 2ull * batch_max * slot_bytes <= (4ull << 30)) batch_max *= 2u;
        // the queue builds of the tile kernel (first scatter events shaded in place, then the lane machine) for the batches too: they hand
        // their radiance rows to the scratch slots instead of folding them (svr_tile_tasks.hpp, scatter_pending)
        const bool ahead_queue = use_queue && !local_majorant && !g.opt_fast_math;
        auto in_stock = [&](const Context::Ahead& a, uint32_t frame) {
            return a.valid && a.content == g.content_version && a.depth == rp->traceDepth && frame >= a.first && frame - a.first < a.count &&
                   memcmp(&a.scene, &s, sizeof s) == 0 && a.shape.x0 == shape.x0 && a.shape.x1 == shape.x1 && a.shape.y0 == shape.y0 &&
                   a.shape.y1 == shape.y1 && a.shape.strip_rows == shape.strip_rows && a.shape.rank == shape.rank && a.shape.world == shape.world;
        };
        for (Context::Ahead& a : g.ahead) {
            if (!in_stock(a, n)) continue;
            // in stock: fold slot n - first into the accumulator
            Context::SlotSet& set = g.sets[a.set];
            svr::DevWork w = shape;
            w.hdr = (float*)rp->hdrBuffer;
            w.img = want_img ? (uint8_t*)img : nullptr;
            w.slot_stride = (uint32_t)g.slot_floats;
            w.lbuf = set.lbuf + (size_t)(n - a.first) * g.slot_floats;
            w.traceDepth = rp->traceDepth;
            w.frame0 = n;
            w.nframes = 1;
            HIP_TRY(hipStreamWaitEvent(g.stream, set.traced, 0));
            HIP_TRY(svr::launch_resolve(s, w, g.stream));
            HIP_TRY(hipEventRecord(set.resolved, g.stream));
            // steady state: a full batch takes as long to trace as to consume, so the NEXT one starts when this one is first used (it runs on
            // its set's stream beside the per-call resolves: the queue builds of the tile kernel leave the register room a resolve wave needs)
            // (while the batches still grow -- 1, 2, 4 ... -- the next one, twice as long, is started the same way: the ramp's traces then run back to
            // back instead of each waiting for its predecessor to be consumed; a host that restarts the render has at most two batches traced in vain)
            const uint32_t next_first = a.first + a.count;
            const uint32_t next_count = 2u * a.count < batch_max ? 2u * a.count : batch_max;
            if (next_count > 1u && !in_stock(g.ahead[0], next_first) && !in_stock(g.ahead[1], next_first))
                return trace_group(next_set(), next_first, next_count, 0, false, ahead_queue && next_count >= 16u);
            return 0;
        }
        uint32_t batch = 1;
        while (batch < batch_max && 2u * batch <= n + 1u) batch *= 2u;
        if (batch_max > 1 && ensure_slots(s.imageW, s.imageH, batch_max)) return g.err_code;
        if (trace_group(next_set(), n, batch, 1, want_img, ahead_queue && batch >= 16u)) return g.err_code;
        if (batch_max > 1) {
            const uint32_t next_count = 2u * batch < batch_max ? 2u * batch : batch_max;
            return trace_group(next_set(), n + batch, next_count, 0, false, ahead_queue && next_count >= 16u);
        }
        return 0;
    }

    // folding launches may take 64 frames (a wave = ONE pixel x 64 frames): half as many launch boundaries
    const uint32_t group = fold_batch ? (uint32_t)g.opt_group_frames : (uint32_t)Context::GROUP;
    for (uint32_t g0 = 0; g0 < nframes; g0 += group) {
        uint32_t n = nframes - g0 < group ? nframes - g0 : group;
        bool last = g0 + n >= nframes;
        if (fold_batch && n >= FOLD_MIN) {
            if (use_split ? trace_split(rp->frameNo + g0, n, want_img && last) : trace_fold(rp->frameNo + g0, n, want_img && last)) return g.err_code;
        } else if (trace_group(next_set(), rp->frameNo + g0, n, n, want_img && last)) return g.err_code;
    }
    return 0;
}

} // namespace

// hooks for the other translation units of the library (svr_internal.hpp)
namespace svr {
int report_error(int code, const char* msg) { return fail(code, "%s", msg); }
int ensure_ready() { return ensure_init(); }
hipStream_t current_stream() { return g.stream; }
}

extern "C" {

int svr_abi_version(void) { return 1; }

int svr_init(int device)
{
    std::lock_guard<std::mutex> lk(g_mu);
    if (g.inited && g.device == device) return 0;
    if (g.inited) return fail(-5, "svr_init(%d): already initialised on device %d (one scene per process, like the reference)", device, g.device);
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) return fail((int)e, "hipSetDevice(%d) failed: %s", device, hipGetErrorName(e));
    return ensure_init();
}

void svr_shutdown(void)
{
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g.inited) return;
    hipDeviceSynchronize();
    for (auto& kv : g.textures) {
        if (kv.second->data) hipFree(kv.second->data);
        if (kv.second->mm) hipFree(kv.second->mm);
        if (kv.second->mm_fine) hipFree(kv.second->mm_fine);
        if (kv.second->mm_wide) hipFree(kv.second->mm_wide);
        if (kv.second->zero_prefix) hipFree(kv.second->zero_prefix);
        if (kv.second->env_cdf) hipFree(kv.second->env_cdf);
        delete kv.second;
    }
    g.textures.clear();
    for (auto& st : g.sets) {
        if (st.lbuf) hipFree(st.lbuf);
        for (auto& p : st.planes) if (p) hipFree(p);
        if (st.wf_counts) hipFree(st.wf_counts);
        if (st.stream) hipStreamDestroy(st.stream);
        if (st.traced) hipEventDestroy(st.traced);
        if (st.resolved) hipEventDestroy(st.resolved);
    }
    if (g.d_mask) hipFree(g.d_mask);
    if (g.d_mask_tmp) hipFree(g.d_mask_tmp);
    if (g.d_fine_mask) hipFree(g.d_fine_mask);
    if (g.d_sub8) hipFree(g.d_sub8);
    if (g.d_bnd8) hipFree(g.d_bnd8);
    if (g.d_counters) hipFree(g.d_counters);
    if (g.d_ticket) hipFree(g.d_ticket);
    if (g.d_queue) hipFree(g.d_queue);
    if (g.d_pend) hipFree(g.d_pend);
    if (g.queue_done) hipEventDestroy(g.queue_done);
    if (g.d_split_pool) hipFree(g.d_split_pool);
    if (g_stage) { hipFree(g_stage); g_stage = nullptr; g_stage_floats = 0; }
    for (int i = 0; i < Context::EV_RING; ++i) {
        if (g.ev0[i]) hipEventDestroy(g.ev0[i]);
        if (g.ev1[i]) hipEventDestroy(g.ev1[i]);
    }
    int fatal = g.fatal;
    g = Context();
    g.fatal = fatal;
}

int svr_set_stream(void* hip_stream)
{
    if (ensure_init()) return g.err_code;
    HIP_TRY(hipDeviceSynchronize());
    collect_timing();
    for (auto& st : g.sets) st.used = false;
    g.stream = (hipStream_t)hip_stream;
    return 0;
}

int svr_device_synchronize(void)
{
    // canvas.cpp:106 synchronises so that the image can be shown: everything that touches the CALLER's buffers (accumulator, image, counters)
    // runs on the launch stream or is ordered into it with events.  Frames being traced ahead on the library's own streams write only the
    // library's scratch slots and are NOT waited for -- a host that synchronises after every render_pathtracer call (the reference's
    // paintGL) would otherwise stall on the next batch at every batch boundary.
    if (ensure_init()) return g.err_code;
    HIP_TRY(hipStreamSynchronize(g.stream));
    return 0;
}

void svr_set_error_mode(int fatal) { g.fatal = fatal ? 1 : 0; }
const char* svr_last_error(void) { return g.err_msg.c_str(); }
int svr_last_error_code(void) { return g.err_code; }
void svr_clear_error(void) { g.err_code = 0; g.err_msg.clear(); }
const char* svr_device_info(void) { ensure_init(); return g.info.c_str(); }

// ---------------- textures ----------------
static uint64_t create_volume_texture(const uint16_t* voxels, int nx, int ny, int nz, int src_is_device, int layout, int auto_rank, bool* oom);

uint64_t svr_create_volume_texture(const uint16_t* voxels, int nx, int ny, int nz, int src_is_device, int layout)
{
    // AUTO prefers the layouts that trade memory for fewer gathers (CELL: 8 x the u16 volume, PAIR: 2 x); if the device has no
    // room for one, the next smaller one is tried before giving up
    if (layout != SVR_LAYOUT_AUTO) return create_volume_texture(voxels, nx, ny, nz, src_is_device, layout, 0, nullptr);
    uint64_t h = 0;
    for (int rank = 0; rank <= 2 && h == 0; ++rank) {
        bool oom = false;
        h = create_volume_texture(voxels, nx, ny, nz, src_is_device, SVR_LAYOUT_AUTO, rank, rank < 2 ? &oom : nullptr);
        if (h == 0 && !oom) break;
    }
    return h;
}

// oom != null: running out of device memory is reported through *oom instead of as an error (the caller retries with a smaller layout)
// auto_rank: under AUTO, 0 = best layout that fits the addressing limits, 1 = skip CELL, 2 = skip CELL and PAIR
static uint64_t create_volume_texture(const uint16_t* voxels, int nx, int ny, int nz, int src_is_device, int layout, int auto_rank, bool* oom)
{
    if (ensure_init()) return 0;
    if (!voxels || nx <= 0 || ny <= 0 || nz <= 0) { fail(-6, "svr_create_volume_texture: bad arguments (%p, %d, %d, %d)", (const void*)voxels, nx, ny, nz); return 0; }
    if (layout != SVR_LAYOUT_AUTO && layout != SVR_LAYOUT_LINEAR && layout != SVR_LAYOUT_BRICK && layout != SVR_LAYOUT_PAIR && layout != SVR_LAYOUT_CELL) { fail(-6, "svr_create_volume_texture: unknown layout %d", layout); return 0; }
    {
        // BRICK needs 24-bit brick-row / brick-slab strides (svr_trace_tile.hip); AUTO picks it when they fit
        size_t bx = ((size_t)nx + 2 * svr::VOL_PAD + svr::BRICK_X - 1) / svr::BRICK_X;
        size_t by = ((size_t)ny + 2 * svr::VOL_PAD + svr::BRICK_Y - 1) / svr::BRICK_Y;
        bool brick_ok = (bx * by * 256) < ((size_t)1 << 24);
        // PAIR: the same bricks with 32-bit elements -- strides and byte offsets double
        const size_t bz_ = ((size_t)nz + 2 * svr::VOL_PAD + svr::BRICK_Z - 1) / svr::BRICK_Z;
        const bool pair_ok = (bx * by * 512) < ((size_t)1 << 24) && bx * by * bz_ * 512 < ((size_t)1 << 32);
        // CELL: 16-byte elements -- the element index uses 24-bit multiplies by the brick-row / brick-slab strides, the byte offset 32 bits
        const bool cell_ok = (bx * by * 128) < ((size_t)1 << 24) && bx * by * bz_ * 128 < ((size_t)1 << 32);      // 32-bit element index, 64-bit byte offset
        // AUTO: CELL (one 16-byte load per fetch; 8 x the memory of the u16 volume -- 2.2 GB for 512^3, 17 GB for 1024^3 of a 288 GB
        // device), else PAIR, else BRICK; on hipErrorOutOfMemory the next smaller one is tried (svr_create_volume_texture)
        if (layout == SVR_LAYOUT_AUTO) layout = auto_rank <= 0 && cell_ok ? SVR_LAYOUT_CELL : (auto_rank <= 1 && pair_ok ? SVR_LAYOUT_PAIR : (brick_ok ? SVR_LAYOUT_BRICK : SVR_LAYOUT_LINEAR));
        if (layout == SVR_LAYOUT_CELL && !cell_ok) { fail(-6, "svr_create_volume_texture: %dx%dx%d is too large for the CELL layout; use PAIR or BRICK", nx, ny, nz); return 0; }
        if (layout == SVR_LAYOUT_PAIR && !pair_ok) { fail(-6, "svr_create_volume_texture: %dx%dx%d is too large for the PAIR layout (32-bit byte offsets); use BRICK", nx, ny, nz); return 0; }
        if (layout == SVR_LAYOUT_BRICK && !brick_ok) { fail(-6, "svr_create_volume_texture: %dx%d slices are too large for the BRICK layout; use LINEAR", nx, ny); return 0; }
    }
    Texture* t = new Texture();
    t->magic = TEX_MAGIC; t->kind = TEX_VOLUME; t->nx = nx; t->ny = ny; t->nz = nz; t->layout = layout;
    size_t px = (size_t)nx + 2 * svr::VOL_PAD, py = (size_t)ny + 2 * svr::VOL_PAD, pz = (size_t)nz + 2 * svr::VOL_PAD;
    size_t elems;
    if (layout == SVR_LAYOUT_LINEAR) {
        t->sy = (int)px; t->sz = (int)(px * py); t->bnx = 0; t->bny = 0;
        elems = px * py * pz;
    } else {
        size_t bx = (px + svr::BRICK_X - 1) / svr::BRICK_X, by = (py + svr::BRICK_Y - 1) / svr::BRICK_Y, bz = (pz + svr::BRICK_Z - 1) / svr::BRICK_Z;
        t->bnx = (int)bx; t->bny = (int)by; t->sy = 0; t->sz = 0;
        elems = bx * by * bz * (size_t)(svr::BRICK_X * svr::BRICK_Y * svr::BRICK_Z);
    }
    if (elems >= ((size_t)1 << 32) || (layout != SVR_LAYOUT_CELL && elems >= ((size_t)1 << 31))) { delete t; fail(-6, "svr_create_volume_texture: %zu padded voxels exceed 32-bit byte offsets", elems); return 0; }
    // macro-cell grid for empty-space skipping: smallest cell size whose bitmask fits MASK_WORDS_MAX words
    {
        int sh = g.opt_macro_shift_min;                       // (SVR_OPT_MACRO_SHIFT_MIN: coarser cells on request)
        for (;; ++sh) {
            size_t gx = (((size_t)nx - 1) >> sh) + 1, gy = (((size_t)ny - 1) >> sh) + 1, gz = (((size_t)nz - 1) >> sh) + 1;
            if (gx * gy * gz <= (size_t)svr::MASK_WORDS_MAX * 32) { t->mc_shift = sh; t->mc_gx = (int)gx; t->mc_gy = (int)gy; t->mc_gz = (int)gz; break; }
        }
    }
    t->bytes = elems * (layout == SVR_LAYOUT_CELL ? 16u : (layout == SVR_LAYOUT_PAIR ? sizeof(uint32_t) : sizeof(uint16_t)));
    size_t src_bytes = (size_t)nx * ny * nz * sizeof(uint16_t);
    const uint16_t* d_src = voxels;
    uint16_t* staged = nullptr;
    hipError_t e;
    if (!src_is_device) {
        e = hipMalloc((void**)&staged, src_bytes);
        if (e == hipSuccess) e = hipMemcpy(staged, voxels, src_bytes, hipMemcpyHostToDevice);
        if (e != hipSuccess) { if (staged) hipFree(staged); delete t; fail((int)e, "volume upload failed: %s", hipGetErrorName(e)); return 0; }
        d_src = staged;
    }
    e = hipMalloc(&t->data, t->bytes);
    if (e == hipSuccess && layout != SVR_LAYOUT_CELL) e = hipMemsetAsync(t->data, 0, t->bytes, g.stream);      // (the CELL repack writes every element)
    if (e == hipSuccess) e = svr::launch_repack(d_src, (uint16_t*)t->data, nx, ny, nz, layout, t->sy, t->sz, t->bnx, t->bny, g.stream);
    if (e == hipSuccess) e = hipMalloc((void**)&t->mm, (size_t)t->mc_gx * t->mc_gy * t->mc_gz * 2 * sizeof(uint16_t));
    if (e == hipSuccess) e = svr::launch_minmax(d_src, t->mm, nx, ny, nz, t->mc_shift, t->mc_gx, t->mc_gy, t->mc_gz, g.stream);
    if (e == hipSuccess && t->mc_shift >= 1) {
        const int fs = t->mc_shift - 1;
        t->fg_x = ((nx - 1) >> fs) + 1; t->fg_y = ((ny - 1) >> fs) + 1; t->fg_z = ((nz - 1) >> fs) + 1;
        e = hipMalloc((void**)&t->mm_fine, (size_t)t->fg_x * t->fg_y * t->fg_z * 2 * sizeof(uint16_t));
        if (e == hipSuccess) e = svr::launch_minmax(d_src, t->mm_fine, nx, ny, nz, fs, t->fg_x, t->fg_y, t->fg_z, g.stream);
    }
    if (e == hipSuccess) {
        // the wide table of the fast bound look-up: half-resolution macro-cells (cells of 2^(shift + 1)), footprints one voxel wider per side
        const int hgx = (t->mc_gx + 1) / 2, hgy = (t->mc_gy + 1) / 2, hgz = (t->mc_gz + 1) / 2;
        e = hipMalloc((void**)&t->mm_wide, (size_t)hgx * hgy * hgz * 2 * sizeof(uint16_t));
        if (e == hipSuccess) e = svr::launch_minmax(d_src, t->mm_wide, nx, ny, nz, t->mc_shift + 1, hgx, hgy, hgz, g.stream, 1);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(g.stream);
    if (staged) hipFree(staged);
    if (e != hipSuccess) {
        const bool retry = oom != nullptr && e == hipErrorOutOfMemory && (layout == SVR_LAYOUT_PAIR || layout == SVR_LAYOUT_CELL);
        if (t->data) hipFree(t->data);
        if (t->mm) hipFree(t->mm);
        if (t->mm_fine) hipFree(t->mm_fine);
        if (t->mm_wide) hipFree(t->mm_wide);
        delete t;
        if (retry) { (void)hipGetLastError(); *oom = true; return 0; }
        fail((int)e, "volume texture creation failed: %s", hipGetErrorName(e));
        return 0;
    }
    uint64_t h = (uint64_t)(uintptr_t)t;
    g.textures[h] = t;
    return h;
}

static uint64_t create_float4_texture(int kind, const float* rgba, int w, int h, int src_is_device)
{
    if (ensure_init()) return 0;
    if (!rgba || w <= 0 || h <= 0) { fail(-6, "texture creation: bad arguments"); return 0; }
    Texture* t = new Texture();
    t->magic = TEX_MAGIC; t->kind = kind; t->nx = w; t->ny = h; t->nz = 1; t->layout = 0;
    t->bytes = (size_t)w * h * 4 * sizeof(float);
    hipError_t e = hipMalloc(&t->data, t->bytes);
    // a device-side table may still be being written by work queued on the caller's stream (torch pool streams do not
    // synchronise with the null stream): order the copy after it
    if (e == hipSuccess && src_is_device) e = hipStreamSynchronize(g.stream);
    if (e == hipSuccess) e = hipMemcpy(t->data, rgba, t->bytes, src_is_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice);
    if (e != hipSuccess) { if (t->data) hipFree(t->data); delete t; fail((int)e, "texture upload failed: %s", hipGetErrorName(e)); return 0; }
    uint64_t hd = (uint64_t)(uintptr_t)t;
    g.textures[hd] = t;
    return hd;
}

// zero_prefix[e] = number of entries < e of the padded alpha table (entry e = alpha of texel clamp(e-1),
// e in [0, n+2]) that are exactly zero; n+4 words
static int upload_zero_prefix(Texture* t)
{
    int n = t->nx;
    std::vector<float> host((size_t)n * 4);
    hipError_t e = hipMemcpy(host.data(), t->data, host.size() * sizeof(float), hipMemcpyDeviceToHost);
    if (e != hipSuccess) return fail((int)e, "transfer-function read-back failed: %s", hipGetErrorName(e));
    std::vector<uint32_t> pre((size_t)n + 4, 0u);
    for (int k = 0; k < n + 3; ++k) {
        int tex = k - 1 < 0 ? 0 : (k - 1 > n - 1 ? n - 1 : k - 1);
        pre[(size_t)k + 1] = pre[(size_t)k] + (host[(size_t)tex * 4 + 3] == 0.f ? 1u : 0u);
    }
    if (!t->zero_prefix) {
        e = hipMalloc((void**)&t->zero_prefix, pre.size() * sizeof(uint32_t));
        if (e != hipSuccess) return fail((int)e, "hipMalloc failed: %s", hipGetErrorName(e));
    }
    e = hipMemcpy(t->zero_prefix, pre.data(), pre.size() * sizeof(uint32_t), hipMemcpyHostToDevice);
    if (e != hipSuccess) return fail((int)e, "zero-prefix upload failed: %s", hipGetErrorName(e));
    t->version++;
    g.content_version++;
    return 0;
}

uint64_t svr_create_tf_texture(const float* rgba, int n, int src_is_device)
{
    if (n > SVR_TF_TABLE_SIZE) { ensure_init(); fail(-6, "transfer-function table of %d entries exceeds the LDS-resident limit of %d", n, SVR_TF_TABLE_SIZE); return 0; }
    uint64_t h = create_float4_texture(TEX_TF, rgba, n, 1, src_is_device);
    if (h && upload_zero_prefix(find_tex(h, TEX_TF))) return 0;
    return h;
}

int svr_update_tf_texture(uint64_t handle, const float* rgba, int n, int src_is_device)
{
    if (ensure_init()) return g.err_code;
    Texture* t = find_tex(handle, TEX_TF);
    if (!t) return fail(-2, "svr_update_tf_texture: bad handle");
    if (n != t->nx || !rgba) return fail(-6, "svr_update_tf_texture: size mismatch (%d vs %d)", n, t->nx);
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(t->data, rgba, t->bytes, src_is_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice));
    return upload_zero_prefix(t);
}

uint64_t svr_create_env_texture(const float* rgba, int w, int h, int src_is_device)
{
    const uint64_t hd = create_float4_texture(TEX_ENV, rgba, w, h, src_is_device);
    if (!hd) return 0;
    // the sampling table of SVR_OPT_ENV_NEE (a luminance distribution over the texels), built on the device
    Texture* t = find_tex(hd, TEX_ENV);
    float* tmp = nullptr;
    hipError_t e = hipMalloc((void**)&t->env_cdf, ((size_t)h * (w + 1) + h + 1) * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void**)&tmp, ((size_t)w * h + 1) * sizeof(float));
    if (e == hipSuccess) e = svr::launch_env_cdf((const float*)t->data, w, h, t->env_cdf, tmp, g.stream);
    if (e == hipSuccess) e = hipStreamSynchronize(g.stream);
    if (tmp) hipFree(tmp);
    if (e != hipSuccess) { svr_destroy_texture(hd); fail((int)e, "environment sampling table: %s", hipGetErrorName(e)); return 0; }
    return hd;
}

int svr_destroy_texture(uint64_t handle)
{
    if (ensure_init()) return g.err_code;
    auto it = g.textures.find(handle);
    if (it == g.textures.end()) return soft_fail(-2, "svr_destroy_texture: 0x%llx is not a live texture handle (destroyed twice?)", (unsigned long long)handle);
    HIP_TRY(hipDeviceSynchronize());
    Texture* t = it->second;
    if (t->data) hipFree(t->data);
    if (t->mm) hipFree(t->mm);
    if (t->mm_fine) hipFree(t->mm_fine);
    if (t->mm_wide) hipFree(t->mm_wide);
    if (t->zero_prefix) hipFree(t->zero_prefix);
    if (t->env_cdf) hipFree(t->env_cdf);
    if (g.mask_vol == handle || g.mask_tf == handle) g.mask_valid = false;
    t->magic = 0;
    delete t;
    g.textures.erase(it);
    g.content_version++;
    return 0;
}

// ---------------- memory helpers ----------------
void* svr_device_malloc(size_t bytes)
{
    if (ensure_init()) return nullptr;
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) { fail((int)e, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorName(e)); return nullptr; }
    return p;
}
int svr_device_free(void* p) { if (ensure_init()) return g.err_code; HIP_TRY(hipFree(p)); return 0; }
int svr_memcpy_h2d(void* dst, const void* src, size_t bytes) { if (ensure_init()) return g.err_code; HIP_TRY(hipStreamSynchronize(g.stream)); HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice)); return 0; }
int svr_memcpy_d2h(void* dst, const void* src, size_t bytes) { if (ensure_init()) return g.err_code; HIP_TRY(hipStreamSynchronize(g.stream)); HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost)); return 0; }
int svr_memset_device(void* dst, int value, size_t bytes) { if (ensure_init()) return g.err_code; HIP_TRY(hipMemsetAsync(dst, value, bytes, g.stream)); return 0; }

int svr_render_params_setup_hdr(svr_render_params* p, uint32_t w, uint32_t h)
{
    // render_parameters.h:17-23
    if (ensure_init()) return g.err_code;
    if (!p) return fail(-4, "svr_render_params_setup_hdr: null");
    if (svr_render_params_clear(p)) return g.err_code;
    size_t bytes = sizeof(float) * 3 * (size_t)w * h;
    HIP_TRY(hipMalloc(&p->hdrBuffer, bytes));
    HIP_TRY(hipMemset(p->hdrBuffer, 0, bytes));
    return 0;
}

int svr_render_params_clear(svr_render_params* p)
{
    // render_parameters.h:25-32
    if (ensure_init()) return g.err_code;
    if (p && p->hdrBuffer) {
        HIP_TRY(hipStreamSynchronize(g.stream));
        HIP_TRY(hipFree(p->hdrBuffer));
        p->hdrBuffer = nullptr;
    }
    return 0;
}

// ---------------- the reference's setup_* (pathtracer.cu:34-68) ----------------
void setup_volume(const svr_volume* vol)
{
    if (ensure_init()) return;
    if (!vol) { fail(-4, "setup_volume: null"); return; }
    g.vol = *vol; g.have_vol = true;
}
void setup_transferfunction(const svr_transfer_function* tf)
{
    if (ensure_init()) return;
    if (!tf) { fail(-4, "setup_transferfunction: null"); return; }
    g.tf = *tf; g.have_tf = true;
}
void setup_camera(const svr_camera* cam)
{
    if (ensure_init()) return;
    if (!cam) { fail(-4, "setup_camera: null"); return; }
    g.cam = *cam; g.have_cam = true;
}
void setup_env_lights(const svr_environment_light* light)
{
    if (ensure_init()) return;
    if (!light) { fail(-4, "setup_env_lights: null"); return; }
    g.env = *light;
}
void setup_area_lights(svr_area_light* lights, uint32_t n)
{
    if (ensure_init()) return;
    if (n > SVR_MAX_LIGHT_SOURCES) n = SVR_MAX_LIGHT_SOURCES;     // the reference's host guard is off by one (lights.cpp:94)
    if (n && !lights) { fail(-4, "setup_area_lights: null"); return; }
    g.num_lights = n;
    for (uint32_t i = 0; i < n; ++i) g.lights[i] = lights[i];
}

// ---------------- render entry points ----------------
void render_pathtracer(void* img, const svr_render_params* renderParams)
{
    render_frames(img, renderParams, 1, true);
}

int svr_render_pathtracer_frames(void* img, const svr_render_params* renderParams, uint32_t nframes)
{
    return render_frames(img, renderParams, nframes, true);
}

int svr_hdr_to_ldr(void* img, const svr_render_params* rp)
{
    if (ensure_init()) return g.err_code;
    if (!rp || !rp->hdrBuffer || !img) return fail(-4, "svr_hdr_to_ldr: null argument");
    if (!g.have_vol || !g.have_tf || !g.have_cam) return fail(-4, "svr_hdr_to_ldr before setup_*");
    svr::DevScene s;
    if (build_scene(g.vol, g.tf, g.cam, s)) return g.err_code;
    svr::DevWork w;
    fill_work(w, s.imageW, s.imageH);
    w.hdr = (float*)rp->hdrBuffer;
    w.img = (uint8_t*)img;
    HIP_TRY(svr::launch_tonemap(s, w, g.stream));
    return 0;
}

int svr_hdr_to_ldr_frame(void* img, const void* hdr, uint32_t w_, uint32_t h_)
{
    if (ensure_init()) return g.err_code;
    if (!hdr || !img || w_ == 0 || h_ == 0) return fail(-4, "svr_hdr_to_ldr_frame: bad argument");
    if (!g.have_cam) return fail(-4, "svr_hdr_to_ldr_frame before setup_camera (the exposure comes from the camera)");
    svr::DevScene s;
    memset(&s, 0, sizeof s);
    s.imageW = w_; s.imageH = h_;
    s.exposure = g.cam.exposure;
    svr::DevWork w;
    fill_work_full(w, w_, h_);
    w.hdr = (float*)const_cast<void*>(hdr);
    w.img = (uint8_t*)img;
    HIP_TRY(svr::launch_tonemap(s, w, g.stream));
    return 0;
}

void render_raycasting(void* img, svr_volume* volume, svr_transfer_function* transferFunction, svr_camera* camera, float stepSize)
{
    if (ensure_init()) return;
    if (!img || !volume || !transferFunction || !camera) { fail(-4, "render_raycasting: null argument"); return; }
    if (!(stepSize > 0.f) || !std::isfinite(stepSize)) { fail(-3, "render_raycasting: stepSize must be finite and > 0 (got %g)", (double)stepSize); return; }
    svr::DevScene s;
    if (build_scene(*volume, *transferFunction, *camera, s)) return;
    svr::DevWork w;
    fill_work(w, s.imageW, s.imageH);
    w.img = (uint8_t*)img;
    if (ensure_mask(s, *volume, *transferFunction)) return;
    hipError_t e = svr::launch_raycast(s, w, stepSize, g.opt_count != 0, g.num_cus, g.opt_rc_lanes, g.stream);
    if (e != hipSuccess) fail((int)e, "render_raycasting launch failed: %s", hipGetErrorName(e));
}

// ---------------- extensions ----------------
int svr_set_row_shard(uint32_t strip_rows, uint32_t rank, uint32_t world)
{
    if (world <= 1 || strip_rows == 0) { g.strip_rows = 0; g.rank = 0; g.world = 1; return 0; }
    if (rank >= world) return fail(-6, "svr_set_row_shard: rank %u >= world %u", rank, world);
    if (strip_rows % 8 != 0) return fail(-6, "svr_set_row_shard: strip_rows must be a multiple of 8 (got %u)", strip_rows);
    g.strip_rows = strip_rows; g.rank = rank; g.world = world;
    return 0;
}

// ---------------- frame assembly (native counterpart of sunvolumerender_amd/dist.py) ----------------
uint32_t svr_strip_rows_owned(uint32_t H, uint32_t strip_rows, uint32_t rank, uint32_t world)
{
    if (world <= 1 || strip_rows == 0) return rank == 0 || world <= 1 ? H : 0;
    if (rank >= world) return 0;
    uint32_t rows = 0;
    const uint32_t nstrips = (H + strip_rows - 1) / strip_rows;
    for (uint32_t sidx = rank; sidx < nstrips; sidx += world) {
        const uint32_t ys = sidx * strip_rows, ye = ys + strip_rows < H ? ys + strip_rows : H;
        rows += ye - ys;
    }
    return rows;
}

uint32_t svr_strip_row_to_y(uint32_t p, uint32_t H, uint32_t strip_rows, uint32_t rank, uint32_t world)
{
    if (p >= svr_strip_rows_owned(H, strip_rows, rank, world)) return 0xffffffffu;
    if (world <= 1 || strip_rows == 0) return p;
    const uint32_t q = p / strip_rows;
    return (q * world + rank) * strip_rows + (p - q * strip_rows);
}

static int strips_call(void* packed, void* frame, uint32_t W, uint32_t H, uint32_t strip_rows, uint32_t rank, uint32_t world, int to_packed, const char* who)
{
    if (ensure_init()) return g.err_code;
    if (!packed || !frame || W == 0 || H == 0) return fail(-4, "%s: bad argument", who);
    if (world > 1 && (rank >= world || strip_rows == 0)) return fail(-6, "%s: rank %u of %u, strip_rows %u", who, rank, world, strip_rows);
    HIP_TRY(svr::launch_strips((float*)packed, (float*)frame, 3u * W, svr_strip_rows_owned(H, strip_rows, rank, world), strip_rows, rank, world, to_packed, g.stream));
    return 0;
}
int svr_pack_strips(void* packed, const void* hdr, uint32_t W, uint32_t H, uint32_t strip_rows, uint32_t rank, uint32_t world)
{
    return strips_call(packed, const_cast<void*>(hdr), W, H, strip_rows, rank, world, 1, "svr_pack_strips");
}
int svr_unpack_strips(void* frame, const void* packed, uint32_t W, uint32_t H, uint32_t strip_rows, uint32_t rank, uint32_t world)
{
    return strips_call(const_cast<void*>(packed), frame, W, H, strip_rows, rank, world, 0, "svr_unpack_strips");
}

namespace {
// the four RCCL entry points of the exchange, resolved from the RCCL the process has loaded (rccl.h: ncclResult_t is an int, 0 = success)
struct Rccl {
    int (*group_start)() = nullptr;
    int (*group_end)() = nullptr;
    int (*send)(const void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*recv)(void*, size_t, int, int, void*, hipStream_t) = nullptr;
    bool tried = false;
} g_rccl;
constexpr int RCCL_FLOAT32 = 7;                          // ncclFloat32
bool rccl_resolve()
{
    if (!g_rccl.tried) {
        g_rccl.tried = true;
        void* h = nullptr;
        for (const char* name : {"librccl.so", "librccl.so.1", "libnccl.so", "libnccl.so.2"})
            if ((h = dlopen(name, RTLD_NOW | RTLD_GLOBAL)) != nullptr) break;
        if (h) {
            g_rccl.group_start = (int (*)())dlsym(h, "ncclGroupStart");
            g_rccl.group_end = (int (*)())dlsym(h, "ncclGroupEnd");
            g_rccl.send = (int (*)(const void*, size_t, int, int, void*, hipStream_t))dlsym(h, "ncclSend");
            g_rccl.recv = (int (*)(void*, size_t, int, int, void*, hipStream_t))dlsym(h, "ncclRecv");
        }
    }
    return g_rccl.group_start && g_rccl.group_end && g_rccl.send && g_rccl.recv;
}
}

int svr_assemble_frame(void* nccl_comm, void* frame_on_root, const void* hdr_local, uint32_t W, uint32_t H,
                       uint32_t strip_rows, uint32_t rank, uint32_t world, uint32_t root)
{
    if (ensure_init()) return g.err_code;
    if (!hdr_local || W == 0 || H == 0 || world == 0 || rank >= world || root >= world) return fail(-4, "svr_assemble_frame: bad argument");
    if (rank == root && !frame_on_root) return fail(-4, "svr_assemble_frame: frame_on_root is null on the root rank");
    const size_t row = (size_t)3 * W;
    if (world == 1) {
        HIP_TRY(hipMemcpyAsync(frame_on_root, hdr_local, row * H * sizeof(float), hipMemcpyDeviceToDevice, g.stream));
        return 0;
    }
    if (!nccl_comm) return fail(-4, "svr_assemble_frame: nccl_comm is null");
    if (strip_rows == 0) return fail(-6, "svr_assemble_frame: strip_rows is 0");
    if (!rccl_resolve()) return fail(-7, "svr_assemble_frame: no RCCL in this process (dlopen(\"librccl.so\") / ncclSend / ncclRecv not found)");
    // staging: on root the packed rows of every rank back to back (= one frame), elsewhere this rank's rows
    const size_t need = rank == root ? row * H : row * svr_strip_rows_owned(H, strip_rows, rank, world);
    if (need > g_stage_floats) {
        HIP_TRY(hipStreamSynchronize(g.stream));
        if (g_stage) HIP_TRY(hipFree(g_stage));
        g_stage = nullptr; g_stage_floats = 0;
        HIP_TRY(hipMalloc((void**)&g_stage, need * sizeof(float)));
        g_stage_floats = need;
    }
    if (rank != root) {
        const uint32_t mine = svr_strip_rows_owned(H, strip_rows, rank, world);
        if (svr_pack_strips(g_stage, hdr_local, W, H, strip_rows, rank, world)) return g.err_code;
        if (mine) {
            const int rc = g_rccl.send(g_stage, row * mine, RCCL_FLOAT32, (int)root, nccl_comm, g.stream);
            if (rc != 0) return fail(-7, "svr_assemble_frame: ncclSend failed (%d)", rc);
        }
        return 0;
    }
    // root: its own rows straight into the frame, the peers' rows through the staging buffer
    if (frame_on_root != hdr_local) {
        if (svr_pack_strips(g_stage, hdr_local, W, H, strip_rows, rank, world)) return g.err_code;
        if (svr_unpack_strips(frame_on_root, g_stage, W, H, strip_rows, rank, world)) return g.err_code;
    }
    int rc = g_rccl.group_start();
    size_t off = 0;
    std::vector<size_t> offs(world, 0);
    for (uint32_t r = 0; r < world && rc == 0; ++r) {
        const uint32_t n = svr_strip_rows_owned(H, strip_rows, r, world);
        offs[r] = off;
        if (r != root && n) rc = g_rccl.recv(g_stage + off, row * n, RCCL_FLOAT32, (int)r, nccl_comm, g.stream);
        off += row * n;
    }
    const int rc2 = g_rccl.group_end();
    if (rc != 0 || rc2 != 0) return fail(-7, "svr_assemble_frame: ncclRecv failed (%d, %d)", rc, rc2);
    for (uint32_t r = 0; r < world; ++r)
        if (r != root && svr_unpack_strips(frame_on_root, g_stage + offs[r], W, H, strip_rows, r, world)) return g.err_code;
    return 0;
}

int svr_set_render_window(int x0, int y0, int x1, int y1)
{
    g.wx0 = x0; g.wy0 = y0; g.wx1 = x1; g.wy1 = y1;
    return 0;
}

int svr_set_option(int key, int value)
{
    switch (key) {
    case SVR_OPT_ENV_ON_ESCAPE: g.opt_env_on_escape = value ? 1 : 0; return 0;
    case SVR_OPT_KERNEL:
        if (value < 0 || value > 4) return fail(-6, "SVR_OPT_KERNEL: bad value %d", value);
        g.opt_kernel = value; return 0;
    case SVR_OPT_COUNT: g.opt_count = value ? 1 : 0; return 0;
    case SVR_OPT_TIMING: g.opt_timing = value ? 1 : 0; return 0;
    case SVR_OPT_SKIP_TONEMAP: g.opt_skip_tonemap = value ? 1 : 0; return 0;
    case SVR_OPT_BLOCKS_PER_CU:
        if (value < 0 || value > 8) return fail(-6, "SVR_OPT_BLOCKS_PER_CU: bad value %d", value);
        g.opt_blocks_per_cu = value; return 0;
    case SVR_OPT_PIPELINE: g.opt_pipeline = value ? 1 : 0; return 0;
    case SVR_OPT_EMPTY_SKIP: g.opt_empty_skip = value ? 1 : 0; return 0;
    case SVR_OPT_RAY_SKIP: g.opt_ray_skip = value ? 1 : 0; return 0;
    case SVR_OPT_BOUND_CULL:
        if (value < 0 || value > 2) return fail(-6, "SVR_OPT_BOUND_CULL: bad value %d (0 off, 1 auto, 2 always)", value);
        g.opt_bound_cull = value; return 0;
    case SVR_OPT_FOLD: g.opt_fold = value ? 1 : 0; return 0;
    case SVR_OPT_GROUP_FRAMES:
        if (value != 8 && value != 16 && value != 32 && value != 64) return fail(-6, "SVR_OPT_GROUP_FRAMES: bad value %d (8, 16, 32, 64)", value);
        g.opt_group_frames = value; return 0;
    case SVR_OPT_ROW_ORDER: g.opt_row_order = value ? 1 : 0; return 0;
    case SVR_OPT_FINE_MASK:
        if (value < 0 || value > 2) return fail(-6, "SVR_OPT_FINE_MASK: bad value %d (0 off, 1 auto, 2 always)", value);
        g.opt_fine_mask = value; return 0;
    case SVR_OPT_FAST_MATH: g.opt_fast_math = value ? 1 : 0; g.ahead[0].valid = g.ahead[1].valid = false; return 0;
    case SVR_OPT_LIGHT_CULL: g.opt_light_cull = value ? 1 : 0; return 0;
    case SVR_OPT_LM_SUBCELLS:
        if (value < 0 || value > 2) return fail(-6, "SVR_OPT_LM_SUBCELLS: bad value %d (0 off, 1 auto, 2 always)", value);
        g.opt_lm_sub = value; g.ahead[0].valid = g.ahead[1].valid = false; return 0;
    case SVR_OPT_LM_TUNE: {
        const int steps = value & 0xff, refill = (value >> 8) & 0xff, ended = (value >> 16) & 0xff, tasks = (value >> 24) & 0x7f;
        if (value != 0 && (steps < 1 || steps > 64 || refill < 1 || refill > 64 || ended < 1 || ended > 64 || tasks > 23 || value < 0)) return fail(-6, "SVR_OPT_LM_TUNE: bad value 0x%x (cells per turn | idle lanes << 8 | ended walks << 16, each 1..64, | tasks per batch of the depth-1 pool << 24, 0 = default .. 23)", value);
        g.opt_lm_tune = value; return 0;
    }
    case SVR_OPT_LOCAL_MAJORANT:
        if (value < 0 || value > 2) return fail(-6, "SVR_OPT_LOCAL_MAJORANT: bad value %d (0 off, 1 on, 2 on with straight-line paths only)", value);
        g.opt_local_majorant = value; g.ahead[0].valid = g.ahead[1].valid = false; return 0;
    case SVR_OPT_QUEUE:
        if (value < 0 || value > 2) return fail(-6, "SVR_OPT_QUEUE: bad value %d (0 off, 1 auto, 2 always)", value);
        g.opt_queue = value; g.queue_alloc_failed = false; return 0;
    case SVR_OPT_PINHOLE_FAST: g.opt_pinhole_fast = value ? 1 : 0; return 0;
    case SVR_OPT_POOL:
        if (value < 0 || value > 2) return fail(-6, "SVR_OPT_POOL: bad value %d (0 off, 1 auto, 2 always)", value);
        g.opt_pool = value; return 0;
    case SVR_OPT_TRIPS:
        if (value < 0 || value > 2) return fail(-6, "SVR_OPT_TRIPS: bad value %d (0 off, 1 auto, 2 always)", value);
        g.opt_trips = value; return 0;
    case SVR_OPT_PARK_CHEAP:
        if (value < 1 || value > 64) return fail(-6, "SVR_OPT_PARK_CHEAP: bad value %d (1..64)", value);
        g.opt_park_cheap = value; return 0;
    case SVR_OPT_PARK_END:
        if (value < 1 || value > 64) return fail(-6, "SVR_OPT_PARK_END: bad value %d (1..64)", value);
        g.opt_park_end = value; return 0;
    case SVR_OPT_SPLIT:
        if (value < 0 || value > 2) return fail(-6, "SVR_OPT_SPLIT: bad value %d (0 off, 1 or 2 on)", value);
        g.opt_split = value; g.split_alloc_failed = false; return 0;
    case SVR_OPT_ENV_NEE: g.opt_env_nee = value ? 1 : 0; g.ahead[0].valid = g.ahead[1].valid = false; return 0;
    case SVR_OPT_NAN_GUARD: g.opt_nan_guard = value ? 1 : 0; return 0;
    case SVR_OPT_FAST_BOUND: g.opt_fast_bound = value ? 1 : 0; return 0;
    case SVR_OPT_MACRO_SHIFT_MIN:
        if (value < 0 || value > 6) return fail(-6, "SVR_OPT_MACRO_SHIFT_MIN: bad value %d (0..6)", value);
        g.opt_macro_shift_min = value; return 0;
#ifdef SVR_TEST_HOOKS
    // experiment builds only (tools/exp.py; `SVR_EXTRA_HIPCC_FLAGS=-DSVR_TEST_HOOKS python -m sunvolumerender_amd._build --force`)
    case 100: g.opt_debug_stop = value; return 0;      // timing ablation: stop every path after a phase (wrong images)
    case 101: g.opt_unit = value; return 0;            // tasks per ticket of the tile kernel
#endif
    case SVR_OPT_FRAME_AHEAD: g.opt_frame_ahead = value ? 1 : 0; g.ahead[0].valid = g.ahead[1].valid = false; return 0;
    case SVR_OPT_RAYCAST_LANES_LOG2:
        if (value < 0 || value > 5) return fail(-6, "SVR_OPT_RAYCAST_LANES_LOG2: bad value %d (0..5)", value);
        g.opt_rc_lanes = value; return 0;
    case SVR_OPT_FRAMES_PER_WAVE_LOG2:
        if (value < -1 || value > 6) return fail(-6, "SVR_OPT_FRAMES_PER_WAVE_LOG2: bad value %d (-1..6)", value);
        g.opt_frames_log2 = value; return 0;
    case SVR_OPT_REFILL_MIN_IDLE:
        if (value < 1 || value > 64) return fail(-6, "SVR_OPT_REFILL_MIN_IDLE: bad value %d (1..64)", value);
        g.opt_refill = value; return 0;
    default: return fail(-6, "svr_set_option: unknown key %d", key);
    }
}

int svr_get_option(int key)
{
    switch (key) {
    case SVR_OPT_ENV_ON_ESCAPE: return g.opt_env_on_escape;
    case SVR_OPT_KERNEL: return g.opt_kernel;
    case SVR_OPT_COUNT: return g.opt_count;
    case SVR_OPT_TIMING: return g.opt_timing;
    case SVR_OPT_SKIP_TONEMAP: return g.opt_skip_tonemap;
    case SVR_OPT_BLOCKS_PER_CU: return g.opt_blocks_per_cu;
    case SVR_OPT_PIPELINE: return g.opt_pipeline;
    case SVR_OPT_EMPTY_SKIP: return g.opt_empty_skip;
    case SVR_OPT_RAY_SKIP: return g.opt_ray_skip;
    case SVR_OPT_BOUND_CULL: return g.opt_bound_cull;
    case SVR_OPT_FOLD: return g.opt_fold;
    case SVR_OPT_GROUP_FRAMES: return g.opt_group_frames;
    case SVR_OPT_ROW_ORDER: return g.opt_row_order;
    case SVR_OPT_FINE_MASK: return g.opt_fine_mask;
    case SVR_OPT_FAST_MATH: return g.opt_fast_math;
    case SVR_OPT_LOCAL_MAJORANT: return g.opt_local_majorant;
    case SVR_OPT_LIGHT_CULL: return g.opt_light_cull;
    case SVR_OPT_LM_TUNE: return g.opt_lm_tune;
    case SVR_OPT_LM_SUBCELLS: return g.opt_lm_sub;
    case SVR_OPT_QUEUE: return g.opt_queue;
    case SVR_OPT_PARK_END: return g.opt_park_end;
    case SVR_OPT_PARK_CHEAP: return g.opt_park_cheap;
    case SVR_OPT_PINHOLE_FAST: return g.opt_pinhole_fast;
    case SVR_OPT_POOL: return g.opt_pool;
    case SVR_OPT_TRIPS: return g.opt_trips;
    case SVR_OPT_SPLIT: return g.opt_split;
    case SVR_OPT_ENV_NEE: return g.opt_env_nee;
    case SVR_OPT_FAST_BOUND: return g.opt_fast_bound;
    case SVR_OPT_NAN_GUARD: return g.opt_nan_guard;
    case SVR_OPT_MACRO_SHIFT_MIN: return g.opt_macro_shift_min;
    case SVR_OPT_REFILL_MIN_IDLE: return g.opt_refill;
    case SVR_OPT_FRAMES_PER_WAVE_LOG2: return g.opt_frames_log2;
    case SVR_OPT_RAYCAST_LANES_LOG2: return g.opt_rc_lanes;
    case SVR_OPT_FRAME_AHEAD: return g.opt_frame_ahead;
    default: return -1;
    }
}

// test hook (not part of the rendering API): runs the ray caster's chain primitives on n items of (t, h, bound, steps)
int svr_selftest_chain(const float* items, float* results, uint32_t n)
{
    if (ensure_init()) return g.err_code;
    if (!items || !results || n == 0) return fail(-4, "svr_selftest_chain: bad arguments");
    float4* d_buf = nullptr;                                  // items, then results
    HIP_TRY(hipMalloc((void**)&d_buf, (size_t)n * 32));
    hipError_t e = hipMemcpy(d_buf, items, (size_t)n * 16, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = svr::launch_chain_selftest(d_buf, d_buf + n, n, g.stream);
    if (e == hipSuccess) e = hipStreamSynchronize(g.stream);
    if (e == hipSuccess) e = hipMemcpy(results, d_buf + n, (size_t)n * 16, hipMemcpyDeviceToHost);
    hipFree(d_buf);
    if (e != hipSuccess) return fail((int)e, "svr_selftest_chain failed: %s", hipGetErrorName(e));
    return 0;
}

// test hook: device-side known-answer tests of the numeric contract (csrc/svr_selftest.hip)
int svr_selftest_math(int fn, const float* in, uint32_t in_stride, float* out, uint32_t n)
{
    if (ensure_init()) return g.err_code;
    if (!in || !out || n == 0 || in_stride == 0 || in_stride > 4) return fail(-4, "svr_selftest_math: bad arguments");
    float* d_in = nullptr; float* d_out = nullptr;
    HIP_TRY(hipMalloc((void**)&d_in, (size_t)n * in_stride * sizeof(float)));
    hipError_t e = hipMalloc((void**)&d_out, (size_t)n * sizeof(float));
    if (e == hipSuccess) e = hipMemcpy(d_in, in, (size_t)n * in_stride * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = svr::launch_math_selftest(fn, d_in, in_stride, d_out, n, g.stream);
    if (e == hipSuccess) e = hipStreamSynchronize(g.stream);
    if (e == hipSuccess) e = hipMemcpy(out, d_out, (size_t)n * sizeof(float), hipMemcpyDeviceToHost);
    hipFree(d_in);
    if (d_out) hipFree(d_out);
    if (e != hipSuccess) return fail((int)e, "svr_selftest_math(%d) failed: %s", fn, hipGetErrorName(e));
    return 0;
}

// test hook: property test of the fast bound look-up on the CURRENT scene (set up as for render_pathtracer): csrc/svr_selftest.hip
int svr_selftest_bound8(const float* rays, uint32_t n, uint32_t* out)
{
    if (ensure_init()) return g.err_code;
    if (!rays || !out || n == 0) return fail(-4, "svr_selftest_bound8: bad arguments");
    if (!g.have_vol || !g.have_tf || !g.have_cam) return fail(-4, "svr_selftest_bound8 before setup_volume/setup_transferfunction/setup_camera");
    svr::DevScene s;
    if (build_scene(g.vol, g.tf, g.cam, s)) return g.err_code;
    if (ensure_mask(s, g.vol, g.tf)) return g.err_code;
    if (s.bnd8 == nullptr) return fail(-3, "svr_selftest_bound8: the fast bound look-up is not in use for this scene");
    float* d_rays = nullptr; uint32_t* d_out = nullptr;
    HIP_TRY(hipMalloc((void**)&d_rays, (size_t)n * 7 * sizeof(float)));
    hipError_t e = hipMalloc((void**)&d_out, (size_t)n * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMemcpy(d_rays, rays, (size_t)n * 7 * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = svr::launch_bound8_selftest(s, d_rays, n, d_out, g.stream);
    if (e == hipSuccess) e = hipStreamSynchronize(g.stream);
    if (e == hipSuccess) e = hipMemcpy(out, d_out, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost);
    hipFree(d_rays);
    if (d_out) hipFree(d_out);
    if (e != hipSuccess) return fail((int)e, "svr_selftest_bound8 failed: %s", hipGetErrorName(e));
    return 0;
}

#ifdef SVR_TEST_HOOKS
// experiment builds: the lane machine's phase profile (svr_lanes.hpp), n <= 32 words, cleared by svr_reset_counters
extern "C" int svr_debug_phase_profile(uint64_t* out, int n)
{
    if (ensure_init()) return g.err_code;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, g.d_counters + svr::CNT_N, sizeof(uint64_t) * (size_t)(n < (int)DEBUG_WORDS ? n : (int)DEBUG_WORDS), hipMemcpyDeviceToHost));
    return 0;
}
#endif

int svr_get_counters(svr_counters* out)
{
    if (ensure_init()) return g.err_code;
    if (!out) return fail(-4, "svr_get_counters: null");
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, g.d_counters, sizeof(svr_counters), hipMemcpyDeviceToHost));
    return 0;
}

int svr_reset_counters(void)
{
    if (ensure_init()) return g.err_code;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemset(g.d_counters, 0, sizeof(svr_counters) + DEBUG_WORDS * 8));
    return 0;
}

int svr_get_kernel_time(double* total_ms, uint64_t* launches)
{
    if (ensure_init()) return g.err_code;
    collect_timing();
    if (total_ms) *total_ms = g.kernel_ms;
    if (launches) *launches = g.kernel_launches;
    return 0;
}

int svr_reset_kernel_time(void)
{
    if (ensure_init()) return g.err_code;
    collect_timing();
    g.kernel_ms = 0.0;
    g.kernel_launches = 0;
    return 0;
}

} // extern "C"
