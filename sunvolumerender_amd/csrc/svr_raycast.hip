// svr_raycast.hip -- kernel_raycasting (raycasting.cu:15-67): front-to-back emission/absorption
// compositing with a head-light Phong term, fixed step stepSize/2, early exit at opacity > 0.95.
//
// One thread per owned pixel, 16x16-pixel blocks of four 8x8 waves.  The RGBA transfer function and
// the `empty` macro-cell bitmask (svr_accel.hip) live in LDS.  A sample whose trilinear cell lies in
// an empty macro-cell has opacity exactly 0, so its contribution to (L.rgb, L.a) is exactly +0 and
// the sample -- its intensity tap, six gradient taps, TF lookup and shading -- is skipped; only the
// float accumulation t += h is kept, so the sample positions stay those of the reference.  In a
// non-empty macro-cell the intensity tap and the opacity lookup decide the same way (opacity == 0:
// no gradient, no shading).  Each lane first advances to its next contributing sample in a cheap
// loop, then the wave shades together.  (Transfer-function colours are assumed finite.)
#include "svr_walk.hpp"

namespace svr {

struct LdsRaycast {
    float4 rgba[SVR_TF_MAX + SVR_TF_PAD];      // entry e = texel clamp(e-1)
    uint32_t mask[1];
    uint32_t emask[MASK_WORDS_MAX];
};

template <int LAYOUT, bool COUNT, bool SKIP>
__global__ __launch_bounds__(256) void k_raycast(const DevScene s, const DevWork w, float stepSize)
{
    __shared__ LdsRaycast L;
    {
        const int n = s.tf_n;
        const float4* g = reinterpret_cast<const float4*>(s.tf);
        for (int e = threadIdx.x; e < n + SVR_TF_PAD; e += 256) L.rgba[e] = g[min(max(e - 1, 0), n - 1)];
        if (SKIP) {
            const uint4* src = reinterpret_cast<const uint4*>(s.empty_mask + s.mask_words);
            uint4* dst = reinterpret_cast<uint4*>(L.emask);
            for (uint32_t q = threadIdx.x; q < (s.mask_words + 3u) / 4u; q += 256u) dst[q] = src[q];
        }
        __syncthreads();
    }
    uint32_t wv = w.x1 - w.x0;
    uint32_t tiles16_x = (wv + 15u) >> 4;
    uint32_t bty = blockIdx.x / tiles16_x, btx = blockIdx.x - bty * tiles16_x;
    uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    uint32_t px = (btx << 4) + ((wave & 1u) << 3) + (lane & 7u);
    uint32_t r = (bty << 4) + ((wave >> 1) << 3) + (lane >> 3);
    uint32_t steps = 0, shaded = 0, fetched = 0;
    if (px < wv && r < w.n_rows) {
        uint32_t x = w.x0 + px, y = owned_row_to_y(w, r);
        v3 orig, dir;
        camera_ray_pinhole(s, x, y, orig, dir);
        const v3 cam = V3(s.cam_pos[0], s.cam_pos[1], s.cam_pos[2]);
        float Lr = 0.f, Lg = 0.f, Lb = 0.f, La = 0.f;
        float tNear, tFar;
        if (volume_intersect(s, orig, dir, tNear, tFar)) {
            const float h = stepSize * 0.5f;
            float t = tNear;
            for (;;) {
                // advance to the next sample that can contribute: opacity > 0 (or a non-finite head-light
                // direction, which the reference turns into NaNs)
                bool live = false;
                v3 p;
                float co[4];
                while (t <= tFar) {
                    steps++;
                    p = orig + dir * t;
                    Cell c = cell_of(s, p);
                    v3 toCam = cam - p;
                    bool finite = dot(toCam, toCam) >= 1e-30f;
                    if (SKIP && finite && cell_is_empty<false>(L, s, c)) { t += h; continue; }
                    fetched++;
                    float intensity = tex_fetch<LAYOUT>(s, c) * s.densityScale;
                    int e; float a;
                    lds_tf_coord(s, intensity, e, a);
                    float4 t0 = L.rgba[e], t1 = L.rgba[e + 1];
                    co[3] = lerpf(t0.w, t1.w, a);
                    if (SKIP && finite && co[3] == 0.f) { t += h; continue; }
                    co[0] = lerpf(t0.x, t1.x, a); co[1] = lerpf(t0.y, t1.y, a); co[2] = lerpf(t0.z, t1.z, a);
                    live = true;
                    break;
                }
                if (!live) break;
                shaded++;
                // cudaVolume::Gradient_CentralDiff, core/cuda_volume.h:54-61
                float xd = intensity_at<LAYOUT>(s, V3(p.x + s.spacing[0], p.y + 0.f, p.z + 0.f)) -
                           intensity_at<LAYOUT>(s, V3(p.x - s.spacing[0], p.y - 0.f, p.z - 0.f));
                float yd = intensity_at<LAYOUT>(s, V3(p.x + 0.f, p.y + s.spacing[1], p.z + 0.f)) -
                           intensity_at<LAYOUT>(s, V3(p.x - 0.f, p.y - s.spacing[1], p.z - 0.f));
                float zd = intensity_at<LAYOUT>(s, V3(p.x + 0.f, p.y + 0.f, p.z + s.spacing[2])) -
                           intensity_at<LAYOUT>(s, V3(p.x - 0.f, p.y - 0.f, p.z - s.spacing[2]));
                v3 gradient = V3((xd * 0.5f) * s.invSpacing[0], (yd * 0.5f) * s.invSpacing[1], (zd * 0.5f) * s.invSpacing[2]);
                float gm = __builtin_sqrtf(dot(gradient, gradient));
                float cosTerm = 1.f, specularTerm = 0.f;
                if ((double)gm > 1e-3) {
                    v3 normal = normalize(gradient);
                    v3 lightDir = normalize(cam - p);
                    cosTerm = __builtin_fabsf(dot(normal, lightDir));
                    specularTerm = powf_(cosTerm, 30.f);
                }
                co[0] = co[0] * co[3] * cosTerm * 0.8f + co[3] * specularTerm * 0.2f;
                co[1] = co[1] * co[3] * cosTerm * 0.8f + co[3] * specularTerm * 0.2f;
                co[2] = co[2] * co[3] * cosTerm * 0.8f + co[3] * specularTerm * 0.2f;
                float wgt = 1.f - La;
                Lr += wgt * co[0]; Lg += wgt * co[1]; Lb += wgt * co[2]; La += wgt * co[3];
                if (La > 0.95f) break;
                t += h;
            }
        }
        Lr = fmin_(Lr, 1.f); Lg = fmin_(Lg, 1.f); Lb = fmin_(Lb, 1.f);
        uint32_t rgba = to_u8(Lr * 255) | (to_u8(Lg * 255) << 8) | (to_u8(Lb * 255) << 16) | (to_u8(255 * La) << 24);
        reinterpret_cast<uint32_t*>(w.img)[(size_t)y * s.imageW + x] = rgba;
    }
    if (COUNT) {
        unsigned long long st = wave_sum(steps), ex = wave_sum((unsigned long long)fetched + 6ull * shaded);
        if (lane == 0) {
            atomicAdd(&w.counters[CNT_RAYCAST], st);
            atomicAdd(&w.counters[CNT_VOL_TAPS], st * 7ull);
            atomicAdd(&w.counters[CNT_TAPS_EXEC], ex);
        }
    }
}

template <int LAYOUT>
static void launch_t(const DevScene& s, const DevWork& w, float stepSize, bool count, uint32_t blocks, hipStream_t st)
{
    const bool skip = s.empty_mask != nullptr;
    if (count) {
        if (skip) hipLaunchKernelGGL((k_raycast<LAYOUT, true, true>), dim3(blocks), dim3(256), 0, st, s, w, stepSize);
        else hipLaunchKernelGGL((k_raycast<LAYOUT, true, false>), dim3(blocks), dim3(256), 0, st, s, w, stepSize);
    } else {
        if (skip) hipLaunchKernelGGL((k_raycast<LAYOUT, false, true>), dim3(blocks), dim3(256), 0, st, s, w, stepSize);
        else hipLaunchKernelGGL((k_raycast<LAYOUT, false, false>), dim3(blocks), dim3(256), 0, st, s, w, stepSize);
    }
}

hipError_t launch_raycast(const DevScene& s, const DevWork& w, float stepSize, bool count, hipStream_t st)
{
    uint32_t wv = w.x1 - w.x0;
    if (wv == 0 || w.n_rows == 0) return hipSuccess;
    uint32_t blocks = ((wv + 15u) >> 4) * ((w.n_rows + 15u) >> 4);
    if (s.layout == LAYOUT_LINEAR) launch_t<LAYOUT_LINEAR>(s, w, stepSize, count, blocks, st);
    else launch_t<LAYOUT_BRICK>(s, w, stepSize, count, blocks, st);
    return hipGetLastError();
}

} // namespace svr
