/*
 * svr_oracle.h -- CPU ORACLE for the SunVolumeRender render path.  TEST INFRASTRUCTURE ONLY.
 *
 * This directory is a checker: only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it.  The product (sunvolumerender_amd/, libsvr_hip.so)
 * never includes, links or calls anything in here.
 *
 * PARITY UNPINNED: the reference (sunwj/SunVolumeRender) ships no tests, golden
 * images or fixtures for this path, and it cannot be built in this image (it needs
 * cuRAND device headers and GLM, which do not exist here, and writing stand-ins for
 * them is not allowed).  One function is the exception: schlick_fresnel
 * (core/bsdf/fresnel.h) needs only <cuda_runtime.h>, a genuine copy of which ships
 * inside the triton wheel; oracle/ref_fresnel.cpp builds it where it lies and
 * svo_schlick matches it bit for bit (tests/test_oracle_kat.py).  Everything else
 * is unpinned.  The oracle is therefore a plain-C
 * restatement of the reference's arithmetic and control flow, function by function,
 * each citing the reference file:line it follows.  Three things the reference
 * delegates to absent third parties are *defined* here and shared, as a written
 * contract, with the HIP kernels:
 *   (i)   texture filtering   (CUDA texture unit; published algorithm: CUDA C
 *         Programming Guide, appendix "Texture Fetching", linear filtering with
 *         texel centres at i+0.5, border/clamp/wrap addressing) -- with float
 *         weights instead of the hardware's 9-bit fixed-point weights;
 *   (ii)  cuRAND XORWOW       (curand_init(seed,0,0) scramble + xorwow recurrence
 *         + curand_uniform mapping, restated from the published curand_kernel.h;
 *         CUDA toolkit version unpinned by the reference, CMakeLists.txt:8);
 *   (iii) libm                (expf/logf/powf/sinf/cosf/acosf/atan2f: fixed
 *         sequences of IEEE-754 binary32 operations (Cephes-style polynomials),
 *         so that CPU and GPU results agree bit for bit; the reference builds with
 *         -use_fast_math, whose intrinsics are not reproducible off an NVIDIA GPU).
 *   GLM (vector algebra, version unpinned; pre-0.9.8 because of glm::uninitialize,
 *   light_sample.h:41) is restated operation by operation (dot = (x*x'+y*y')+z*z',
 *   normalize = v * (1/sqrt(dot)), etc.).
 * Floating-point contraction is OFF (-ffp-contract=off); every fused multiply-add
 * is written explicitly as fmaf() and is part of the contract.
 */
#ifndef SVR_ORACLE_H
#define SVR_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- POD layouts of the reference host API (byte-for-byte) ---- */
typedef struct { float x, y, z; } svo_vec3;          /* glm::vec3, packed 12 B            */
typedef struct { float x, y; } svo_vec2;             /* glm::vec2                          */

typedef struct {                                     /* cudaBBox   core/geometry/cuda_bbox.h:66-69 */
    svo_vec3 vmin, vmax, invSize;
} svo_bbox;                                          /* 36 B */

typedef struct {                                     /* cudaVolume core/cuda_volume.h:111-121 */
    svo_bbox bbox;                                   /* 0   */
    uint32_t _pad0;                                  /* 36  */
    uint64_t tex;                                    /* 40  */
    float densityScale;                              /* 48  */
    float invMaxMagnitude;                           /* 52  */
    float gradientFactor;                            /* 56  */
    svo_vec3 spacing;                                /* 60  */
    svo_vec3 invSpacing;                             /* 72  */
    svo_vec2 x_clip, y_clip, z_clip;                 /* 84, 92, 100 */
    uint32_t _pad1;                                  /* 108 */
} svo_volume;                                        /* 112 B */

typedef struct {                                     /* cudaTransferFunction core/cuda_transfer_function.h:57-59 */
    uint64_t tex;
    float maxOpacity;
    uint32_t _pad;
} svo_tf;                                            /* 16 B */

typedef struct {                                     /* cudaCamera core/cuda_camera.h:98-106 */
    uint32_t imageW, imageH;
    float exposure, apeture, focalLength, aspectRatio, tanFovxOverTwo;
    svo_vec3 pos, u, v, w;
} svo_camera;                                        /* 76 B */

typedef struct {                                     /* cudaDisk core/geometry/cuda_disk.h:58-61 */
    float radius;
    svo_vec3 center, normal;
} svo_disk;                                          /* 28 B */

typedef struct {                                     /* cudaAreaLight core/lights/cuda_arealight.h:68-71 */
    svo_disk disk;
    svo_vec3 color;
    float intensity;
} svo_arealight;                                     /* 44 B */

typedef struct {                                     /* cudaEnvironmentLight core/lights/cuda_environment_light.h:74-78 */
    uint64_t tex;
    svo_vec3 defaultRadiance;
    float intensity;
    svo_vec2 offset;
} svo_envlight;                                      /* 32 B */

#define SVO_MAX_LIGHT_SOURCES 8                      /* common.h:11 */

/* The oracle's scene = the reference's __constant__ globals (pathtracer.cu:34-68)
 * plus host arrays standing in for the three texture objects. */
typedef struct {
    svo_volume vol;
    svo_tf tf;
    svo_camera cam;
    svo_envlight env;
    uint32_t num_lights;
    uint32_t env_on_escape;       /* 0 = reference behaviour (pathtracer.cu:233 commented out); 1 = documented extension */
    svo_arealight lights[SVO_MAX_LIGHT_SOURCES];
    const uint16_t* vox;          /* [nz][ny][nx] u16, as uploaded by VolumeReader.cpp:138-172 */
    int32_t nx, ny, nz;
    int32_t tf_n;                 /* 1024 in the reference (transferfunction.h:29) */
    const float* tf_rgba;         /* tf_n x float4 */
    const float* env_rgba;        /* env_h x env_w x float4 (lat-long), may be NULL */
    int32_t env_w, env_h;
} svo_scene;

typedef struct {
    uint64_t paths;
    uint64_t vol_taps;            /* tex3D fetches (8 voxels each)              */
    uint64_t tf_taps;             /* tex1D fetches                              */
    uint64_t rng_draws;           /* curand_uniform calls                       */
    uint64_t woodcock_iters;      /* iterations of woodcock_tracking.h:32-45    */
    uint64_t scatter_events;      /* VolumeSample fills (pathtracer.cu:237-244) */
    uint64_t shadow_walks;        /* transmittance() calls                      */
    uint64_t raycast_steps;       /* raycasting.cu:30-59 iterations             */
} svo_counters;

/* ---- contract pieces, exported for known-answer tests ---- */
uint32_t svo_wang_hash(uint32_t a);                                   /* pathtracer.cu:70-79 */
void     svo_xorwow_init(uint32_t seed, uint32_t state[6]);           /* curand_init(seed,0,0) */
uint32_t svo_xorwow_next(uint32_t state[6]);                          /* curand()            */
float    svo_xorwow_uniform(uint32_t state[6]);                       /* curand_uniform()    */

float svo_logf(float x);
float svo_expf(float x);
float svo_sinf(float x);
float svo_cosf(float x);
float svo_powf(float x, float y);
float svo_acosf(float x);
float svo_atan2f(float y, float x);

float svo_tex3d(const svo_scene* s, float u, float v, float w);       /* tex3D<float>, border/linear/normalized  */
void  svo_tex1d(const svo_scene* s, float x, float out[4]);           /* tex1D<float4>, clamp/linear/normalized  */
void  svo_tex2d(const svo_scene* s, float u, float v, float out[4]);  /* tex2D<float4>, wrap/linear/normalized   */

float svo_volume_intensity(const svo_scene* s, const float p[3]);     /* cuda_volume.h:92-100 */
void  svo_volume_gradient(const svo_scene* s, const float p[3], float g[3]); /* cuda_volume.h:54-61 */
int   svo_volume_intersect(const svo_scene* s, const float orig[3], const float dir[3],
                           float* tNear, float* tFar);                /* cuda_bbox.h:33-54 */
void  svo_camera_ray(const svo_scene* s, uint32_t x, uint32_t y, uint32_t rng[6],
                     float orig[3], float dir[3]);                    /* cuda_camera.h:66-83 */
void  svo_camera_ray_pinhole(const svo_scene* s, uint32_t x, uint32_t y,
                             float orig[3], float dir[3]);            /* cuda_camera.h:85-95 */
int   svo_disk_intersect(const svo_disk* d, const float orig[3], const float dir[3], float* t); /* cuda_disk.h:32-51 */
void  svo_light_radiance(const svo_arealight* l, float out[3]);       /* cuda_arealight.h:57 */
float svo_schlick(float ni, float no, float c);                       /* fresnel.h:10-15 */
float svo_microfacet_f(const float wi[3], const float wo[3], const float n[3], float ior, float alpha); /* microfacet.h:52-68 */
void  svo_tonemap(const float L[3], float exposure, float out[3]);    /* tonemapping.h:13-27 */
void  svo_onb_from_w(const float w[3], float u[3], float v[3]);       /* cuda_onb.h:26-40 */

/* ---- the hot path ---- */
/* One call = render_pathtracer (pathtracer.cu:292-304): clear iff frameNo==0, one
 * sample per pixel into the running mean, tone-map to img.  The pixel window
 * [x0,x1) x [y0,y1) restricts the work (tiles for the multi-GPU tests); seeds
 * always use the global offset y*W+x.  hdr: W*H*3 floats, img: W*H*4 bytes (may be
 * NULL to skip tone mapping).  nthreads<=0 -> all OpenMP threads. */
void svo_render_pathtracer(const svo_scene* s, float* hdr, uint8_t* img,
                           uint32_t traceDepth, uint32_t frameNo,
                           int x0, int y0, int x1, int y1,
                           svo_counters* counters, int nthreads);

/* The radiance of ONE path (kernel_pathtracer body before running_estimate). */
void svo_trace_path(const svo_scene* s, uint32_t x, uint32_t y, uint32_t traceDepth,
                    uint32_t hashedFrameNo, float L[3], svo_counters* c);

/* hdr_to_ldr (pathtracer.cu:282-290) over a window. */
void svo_hdr_to_ldr(const svo_scene* s, const float* hdr, uint8_t* img,
                    int x0, int y0, int x1, int y1);

/* kernel_raycasting (raycasting.cu:15-67) over a window. */
void svo_render_raycasting(const svo_scene* s, uint8_t* img, float stepSize,
                           int x0, int y0, int x1, int y1,
                           svo_counters* counters, int nthreads);

int svo_max_threads(void);
int svo_sizeof(int which);  /* 0 scene,1 volume,2 tf,3 camera,4 arealight,5 envlight,6 counters */

#ifdef __cplusplus
}
#endif
#endif
