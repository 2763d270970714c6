// svr_kernels.hip -- gfx950 kernels of the SunVolumeRender-compatible render path.
//
//  k_pathtrace_pixel       one thread per pixel, the shape of the reference's kernel_pathtracer
//                          (pathtracer.cu:200-280).  Baseline / debugging kernel.
//  k_pathtrace_uloop  MI355X-first kernel: persistent waves, every lane is a small state
//                          machine that owns one path at a time; all volume fetches of all
//                          stages (primary Woodcock walk, gradient taps, shadow Woodcock walk)
//                          go through ONE shared tap site per scheduler iteration, lanes that
//                          finish a path are regenerated from a global ticket with
//                          ballot + mbcnt prefix sums, and the transfer-function LUT lives in
//                          LDS.  A path's arithmetic does not depend on scheduling, so both
//                          kernels give bit-identical radiance.
//  k_tonemap               hdr_to_ldr (pathtracer.cu:282-290)
//  (k_raycast lives in svr_raycast.hip)
//  k_repack_*              [z][y][x] u16 -> padded LINEAR / BRICK software-texture layouts
#include "svr_kernel_common.hpp"

namespace svr {

// ------------------------------------------------------------------------------------------
// Baseline: straight transcription of one path (pathtracer.cu:205-277)
// ------------------------------------------------------------------------------------------
template <int LAYOUT, bool COUNT>
SVR_DEV float sample_distance(const DevScene& s, const LdsTF& tf, v3 orig, v3 dir, Rng& rng,
                              float& tMin, float& tMax, Cnt& c)
{
    // woodcock_tracking.h:20-51
    float tNear, tFar;
    if (volume_intersect(s, orig, dir, tNear, tFar)) {
        tMin = tNear < 0.f ? (float)1e-6 : tNear;
        tMax = tFar;
        float t = tMin;
        for (uint32_t guard = 0;; ++guard) {
            if (COUNT) c.iters++;
            t += -logf_unit(1.f - rng_uniform(rng)) * s.invSigmaMaxSI;
            if (t > tMax || guard >= SVR_WALK_GUARD) return -SVR_FLT_MAX;
            v3 p = orig + dir * t;
            if (COUNT) { c.taps++; c.exec++; }
            float intensity = volume_intensity<LAYOUT>(s, p);
            float sigma_t = lds_tf_alpha(tf, s, intensity);
            if (rng_uniform(rng) < sigma_t * s.invSigmaMax) break;
        }
        return t;
    }
    return -SVR_FLT_MAX;
}

template <int LAYOUT, bool COUNT>
SVR_DEV v3 trace_path(const DevScene& s, const LdsTF& tf, uint32_t x, uint32_t y,
                      uint32_t traceDepth, uint32_t hashed, Cnt& c)
{
    uint32_t offset = y * s.imageW + x;
    Rng rng;
    rng_init(rng, hashed + offset);
    if (COUNT) c.paths++;
    v3 L = V3(0.f, 0.f, 0.f), T = V3(1.f, 1.f, 1.f);
    v3 orig, dir;
    camera_ray(s, x, y, rng, orig, dir);
    float ls_t;
    int ls_id = nearest_light(s, orig, dir, ls_t);
    for (uint32_t k = 0; k < traceDepth; ++k) {
        float tMin = (float)1e-6, tMax = SVR_FLT_MAX;
        float t = sample_distance<LAYOUT, COUNT>(s, tf, orig, dir, rng, tMin, tMax, c);
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
        Shade vs;
        if (COUNT) { c.scatter++; c.taps += 7; c.exec += 7; }
        vs.wo = -dir;
        vs.pt = orig + dir * t;
        float intensity = volume_intensity<LAYOUT>(s, vs.pt);
        lds_tf_rgba(tf, s, intensity, vs.color);
        vs.gradient = volume_gradient<LAYOUT>(s, vs.pt);
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
                // transmittance.h:10-17
                float sMin = (float)1e-6, sMax = SVR_FLT_MAX;
                if (COUNT) c.shadow++;
                float ts = sample_distance<LAYOUT, COUNT>(s, tf, vs.pt, wiL, rng, sMin, sMax, c);
                float Tr = ((ts > sMin) && (ts < sMax)) ? 0.f : 1.f;
                float kf = Tr * (float)s.num_lights;
                Ld = ((bsdf_eval(vs, wiL) * kf) * Li) / pdfL;
            }
        }
        L = L + T * Ld;
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

#ifndef SVR_WAVES_PER_EU_PIXEL
#define SVR_WAVES_PER_EU_PIXEL 4
#endif
template <int LAYOUT, bool COUNT>
__global__ __launch_bounds__(256, SVR_WAVES_PER_EU_PIXEL) void k_pathtrace_pixel(const DevScene s, const DevWork w)
{
    __shared__ LdsTF tf;
    lds_tf_load(tf, s);
    Cnt c = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    // one (16x16 pixel tile, frame slot) per block, one 8x8 sub-tile per wave
    uint32_t wv = w.x1 - w.x0;
    uint32_t tiles16_x = (wv + 15u) >> 4;
    uint32_t bt = blockIdx.x / w.nframes;
    uint32_t slot = blockIdx.x - bt * w.nframes;
    uint32_t bty = bt / tiles16_x, btx = bt - bty * tiles16_x;
    uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    uint32_t px = (btx << 4) + ((wave & 1u) << 3) + (lane & 7u);
    uint32_t r = (bty << 4) + ((wave >> 1) << 3) + (lane >> 3);
    if (px < wv && r < w.n_rows) {
        uint32_t x = w.x0 + px, y = owned_row_to_y(w, r);
        v3 L = trace_path<LAYOUT, COUNT>(s, tf, x, y, w.traceDepth, wang_hash(w.frame0 + slot), c);
        float* o = w.lbuf + (size_t)slot * w.slot_stride + 3 * ((size_t)y * s.imageW + x);
        o[0] = L.x; o[1] = L.y; o[2] = L.z;
    }
    if (COUNT) cnt_flush(w, c);
}

// ------------------------------------------------------------------------------------------
// Persistent regenerating kernel
// ------------------------------------------------------------------------------------------
enum : uint32_t { S_IDLE = 0, S_START, S_WALK, S_GRAD, S_SHADE, S_SCATTER, S_FINISH, S_DONE };

#ifndef SVR_WAVES_PER_EU
#define SVR_WAVES_PER_EU 4
#endif
template <int LAYOUT, bool COUNT>
__global__ __launch_bounds__(256, SVR_WAVES_PER_EU) void k_pathtrace_uloop(const DevScene s, const DevWork w)
{
    __shared__ LdsTF tf;
    lds_tf_load(tf, s);

    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wv = w.x1 - w.x0;
    const uint32_t n_tiles = ((wv + 7u) >> 3) * ((w.n_rows + 7u) >> 3);
    const uint32_t total_items = (n_tiles * w.nframes) << 6;

    Cnt c = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

    // wave-uniform work range
    uint32_t it_next = 0, it_end = 0;
    bool exhausted = false;

    // per-lane path state
    uint32_t state = S_IDLE;
    Rng rng = {0, 0, 0, 0, 0, 0};
    v3 o = V3(0, 0, 0), d = V3(0, 0, 1);
    float t = 0.f, tMin = 0.f, tMax = 0.f;
    bool shadow = false;
    uint32_t px = 0, py = 0, fslot = 0, k = 0;
    v3 T = V3(1, 1, 1), L = V3(0, 0, 0);
    float ls_t = 0.f; int ls_id = -1;
    Shade vs;
    vs.pt = V3(0, 0, 0); vs.wo = V3(0, 0, 1); vs.gradient = V3(0, 0, 0);
    vs.color[0] = vs.color[1] = vs.color[2] = vs.color[3] = 0.f; vs.Pbrdf = 0.f; vs.st = 0;
    uint32_t gi = 0; float gprev = 0.f;
    uint32_t guard = 0;
    // pending next-event estimate
    bool nee_valid = false; v3 nee_bsdf = V3(0, 0, 0); float nee_pdf = 1.f; int nee_light = 0; float Tr = 1.f;

    for (;;) {
        if (COUNT) c.loops += (lane == 0);

        // ---- path finished: hand its radiance to the resolve pass (running_estimate, pathtracer.cu:279) ----
        if (state == S_FINISH) {
            float* o_ = w.lbuf + (size_t)fslot * w.slot_stride + 3 * ((size_t)py * s.imageW + px);
            o_[0] = L.x; o_[1] = L.y; o_[2] = L.z;
            state = S_IDLE;
        }

        // ---- lane regeneration: hand tickets to idle lanes (ballot + mbcnt prefix sum) ----
        {
            unsigned long long m_idle = __ballot(state == S_IDLE);
            unsigned long long m_busy = __ballot(state != S_IDLE && state != S_DONE);
            uint32_t n_idle = (uint32_t)__popcll(m_idle);
            if (m_idle != 0ull && (n_idle >= w.refill_min_idle || m_busy == 0ull)) {
                if (!exhausted && it_next == it_end) {
                    uint32_t base = 0;
                    if (lane == 0) base = atomicAdd(w.ticket, 64u);
                    base = __builtin_amdgcn_readfirstlane(base);
                    if (base >= total_items) exhausted = true;
                    else { it_next = base; it_end = base + 64u; }
                }
                if (exhausted) {
                    if (state == S_IDLE) state = S_DONE;
                } else {
                    uint32_t avail = it_end - it_next;
                    uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m_idle >> 32),
                                    __builtin_amdgcn_mbcnt_lo((uint32_t)m_idle, 0u));
                    if (state == S_IDLE && rank < avail) {
                        uint32_t x, y, sl;
                        if (item_to_pixel(w, it_next + rank, x, y, sl)) {
                            px = x; py = y; fslot = sl;
                            state = S_START;
                        }
                    }
                    it_next += min(n_idle, avail);
                }
            }
        }

        // ---- start a path: pathtracer.cu:205-215 ----
        bool pend_miss = false;       // primary walk ended without a collision this iteration
        if (state == S_START) {
            uint32_t offset = py * s.imageW + px;
            rng_init(rng, wang_hash(w.frame0 + fslot) + offset);
            if (COUNT) c.paths++;
            L = V3(0.f, 0.f, 0.f);
            T = V3(1.f, 1.f, 1.f);
            camera_ray(s, px, py, rng, o, d);
            ls_id = nearest_light(s, o, d, ls_t);
            k = 0;
            shadow = false;
            float tNear, tFar;
            if (volume_intersect(s, o, d, tNear, tFar)) {
                tMin = tNear < 0.f ? (float)1e-6 : tNear;
                tMax = tFar;
                t = tMin;
                guard = 0;
                state = S_WALK;
            } else {
                pend_miss = true;
                state = S_WALK;       // resolved below, no tap issued
            }
        }

        // ---- after the shadow walk: finish estimate_direct_light, then sample the BSDF ----
        if (state == S_SCATTER) {
            v3 Ld = V3(0.f, 0.f, 0.f);
            if (nee_valid) {
                const DevLight& l = s.lights[nee_light];
                float kf = Tr * (float)s.num_lights;
                Ld = ((nee_bsdf * kf) * V3(l.radiance[0], l.radiance[1], l.radiance[2])) / nee_pdf;
            }
            L = L + T * Ld;
            if (k + 1u < w.traceDepth) {
                v3 wi; float pdf = 0.f;
                v3 f = bsdf_sample(vs, wi, pdf, rng);
                float cosTerm = __builtin_fabsf(dot(normalize(vs.gradient), wi));
                if (fmax_(f.x, fmax_(f.y, f.z)) > 0.f && pdf > 0.f) {
                    if (vs.st == 0) T = T * (f / (pdf * (1.f - vs.Pbrdf)));
                    else T = T * ((f * cosTerm) / (pdf * vs.Pbrdf));
                }
                o = vs.pt;
                d = wi;
                bool term = false;
                if (k >= 3u) term = russian_roulette(T, rng);
                k++;
                if (term) state = S_FINISH;
                else {
                    shadow = false;
                    float tNear, tFar;
                    if (volume_intersect(s, o, d, tNear, tFar)) {
                        tMin = tNear < 0.f ? (float)1e-6 : tNear;
                        tMax = tFar;
                        t = tMin;
                        guard = 0;
                    } else pend_miss = true;
                    state = S_WALK;
                }
            } else {
                // last bounce: sample_bsdf / roulette (pathtracer.cu:259-276) cannot reach L any more
                state = S_FINISH;
            }
        }

        // ---- gradient complete: shading decision + next-event estimation set-up ----
        if (state == S_SHADE) {
            float gradMag = __builtin_sqrtf(dot(vs.gradient, vs.gradient));
            vs.Pbrdf = vs.color[3] * (1.f - expf_(s.pbrdf_c * gradMag * 65535.f * s.invMaxMagnitude));
            vs.st = (rng_uniform(rng) < vs.Pbrdf) ? 1 : 0;
            nee_valid = false;
            Tr = 1.f;
            state = S_SCATTER;
            if (s.num_lights != 0) {
                int lightId = (int)((float)s.num_lights * rng_uniform(rng));
                lightId = lightId < (int)s.num_lights ? lightId : (int)s.num_lights - 1;
                v3 wiL, Li; float pdfL;
                if (sample_light(s.lights[lightId], vs.pt, rng, wiL, pdfL, Li)) {
                    nee_valid = true;
                    nee_light = lightId;
                    nee_pdf = pdfL;
                    nee_bsdf = bsdf_eval(vs, wiL);
                    if (COUNT) c.shadow++;
                    // transmittance.h:10-17: ray(start, normalize(end-start)), tMin=1e-6, tMax=FLT_MAX
                    o = vs.pt;
                    d = wiL;
                    tMin = (float)1e-6;
                    tMax = SVR_FLT_MAX;
                    float tNear, tFar;
                    if (volume_intersect(s, o, d, tNear, tFar)) {
                        tMin = tNear < 0.f ? (float)1e-6 : tNear;
                        tMax = tFar;
                        t = tMin;
                        guard = 0;
                        shadow = true;
                        state = S_WALK;
                    }
                    // else: sample_distance returns -FLT_MAX -> Tr = 1 -> S_SCATTER next iteration
                }
            }
        }

        // ---- the tap site: one volume fetch per lane per iteration, whatever the stage ----
        bool tapping = false;
        v3 p = V3(0.f, 0.f, 0.f);
        if (state == S_WALK && !pend_miss) {
            // woodcock_tracking.h:34-38
            if (COUNT) c.iters++;
            t += -logf_unit(1.f - rng_uniform(rng)) * s.invSigmaMaxSI;
            if (t > tMax || ++guard > SVR_WALK_GUARD) {
                if (shadow) { Tr = 1.f; state = S_SCATTER; }     // t = -FLT_MAX fails (t > tMin)
                else pend_miss = true;
            } else {
                p = o + d * t;
                tapping = true;
            }
        } else if (state == S_GRAD) {
            // cuda_volume.h:56-58, taps in the reference's order: +x -x +y -y +z -z
            float sx = (gi < 2u) ? s.spacing[0] : 0.f;
            float sy = (gi >= 2u && gi < 4u) ? s.spacing[1] : 0.f;
            float sz = (gi >= 4u) ? s.spacing[2] : 0.f;
            if (gi & 1u) p = V3(vs.pt.x - sx, vs.pt.y - sy, vs.pt.z - sz);
            else p = V3(vs.pt.x + sx, vs.pt.y + sy, vs.pt.z + sz);
            tapping = true;
        }

        bool pend_hit = false;
        float val = 0.f;
        if (__ballot(tapping) != 0ull) {
            if (tapping) {
                if (COUNT) { c.taps++; c.exec++; }
                val = volume_intensity<LAYOUT>(s, p);
                if (state == S_WALK) {
                    // woodcock_tracking.h:40-44
                    float sigma_t = lds_tf_alpha(tf, s, val);
                    if (rng_uniform(rng) < sigma_t * s.invSigmaMax) {
                        if (shadow) {
                            Tr = ((t > tMin) && (t < tMax)) ? 0.f : 1.f;
                            state = S_SCATTER;
                        } else pend_hit = true;
                    }
                } else {
                    if (gi & 1u) {
                        float df = gprev - val;
                        float g = (df * 0.5f) * ((gi == 1u) ? s.invSpacing[0] : (gi == 3u) ? s.invSpacing[1] : s.invSpacing[2]);
                        if (gi == 1u) vs.gradient.x = g;
                        else if (gi == 3u) vs.gradient.y = g;
                        else vs.gradient.z = g;
                    } else gprev = val;
                    gi++;
                    if (gi == 6u) state = S_SHADE;
                }
            }
        }

        // ---- primary walk ended: pathtracer.cu:220-244 ----
        if (pend_miss || pend_hit) {
            float tt = pend_hit ? t : -SVR_FLT_MAX;
            bool done = false;
            if (k == 0u && ls_id >= 0) {
                tt = tt < 0.f ? SVR_FLT_MAX : tt;
                if (ls_t < tt) {
                    const DevLight& l = s.lights[ls_id];
                    float cosTerm = dot(V3(l.normal[0], l.normal[1], l.normal[2]), -d);
                    L = L + (T * V3(l.radiance[0], l.radiance[1], l.radiance[2])) * (cosTerm <= 0.f ? 0.f : 1.f);
                    done = true;
                }
            }
            if (!done && tt < 0.f) {
                if (s.env_on_escape) L = L + T * env_radiance(s, d);
                done = true;
            }
            if (done) state = S_FINISH;
            else {
                // VolumeSample: the collision point is the last Woodcock tap, so its intensity is `val`
                if (COUNT) { c.scatter++; c.taps++; }
                vs.wo = -d;
                vs.pt = p;
                lds_tf_rgba(tf, s, val, vs.color);
                gi = 0;
                state = S_GRAD;
            }
        }

        if (__ballot(state != S_DONE) == 0ull) break;
    }
    if (COUNT) cnt_flush(w, c);
}

// ------------------------------------------------------------------------------------------
// Resolve: clear_hdr_buffer (iff frameNo == 0, pathtracer.cu:86-94,297-300) + running_estimate
// (pathtracer.cu:81-84,279) for the frames of the group IN FRAME ORDER + hdr_to_ldr
// (pathtracer.cu:282-290).  One thread per owned pixel, row-contiguous: coalesced reads of the
// scratch slots and coalesced framebuffer writes.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_resolve(const DevScene s, const DevWork w)
{
    uint32_t wv = w.x1 - w.x0;
    uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= w.n_items) return;
    uint32_t r = i / wv, px = i - r * wv;
    uint32_t x = w.x0 + px, y = owned_row_to_y(w, r);
    size_t off = (size_t)y * s.imageW + x;
    float* h = w.hdr + 3 * off;
    v3 acc = (w.frame0 == 0u) ? V3(0.f, 0.f, 0.f) : V3(h[0], h[1], h[2]);
    for (uint32_t f = 0; f < w.nframes; ++f) {
        const float* l = w.lbuf + (size_t)f * w.slot_stride + 3 * off;
        v3 L = V3(l[0], l[1], l[2]);
        if (w.nan_guard) {                                  // SVR_OPT_NAN_GUARD: a non-finite sample is replaced by the running mean (per channel)
            L.x = (f2u(L.x) & 0x7f800000u) == 0x7f800000u ? acc.x : L.x;
            L.y = (f2u(L.y) & 0x7f800000u) == 0x7f800000u ? acc.y : L.y;
            L.z = (f2u(L.z) & 0x7f800000u) == 0x7f800000u ? acc.z : L.z;
        }
        float n1 = (float)(w.frame0 + f) + 1.f;
        acc = acc + (L - acc) / n1;
    }
    h[0] = acc.x; h[1] = acc.y; h[2] = acc.z;
    if (w.img) reinterpret_cast<uint32_t*>(w.img)[off] = tonemap_pixel(acc, s.exposure);
}

// ------------------------------------------------------------------------------------------
// hdr_to_ldr, pathtracer.cu:282-290 -- one thread per owned pixel, row-contiguous (coalesced)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_tonemap(const DevScene s, const DevWork w)
{
    uint32_t wv = w.x1 - w.x0;
    uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= w.n_items) return;
    uint32_t r = i / wv, px = i - r * wv;
    uint32_t x = w.x0 + px, y = owned_row_to_y(w, r);
    size_t off = (size_t)y * s.imageW + x;
    const float* h = w.hdr + 3 * off;
    uint32_t rgba = tonemap_pixel(V3(h[0], h[1], h[2]), s.exposure);
    reinterpret_cast<uint32_t*>(w.img)[off] = rgba;
}

// ------------------------------------------------------------------------------------------
// volume repack: dst must be zero-filled (apron) before the launch
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_repack(const uint16_t* __restrict__ src, uint16_t* __restrict__ dst,
                                                int nx, int ny, int nz, int layout, int sy, int sz, int bnx, int bny)
{
    size_t n = (size_t)nx * ny * nz;
    for (size_t e = (size_t)blockIdx.x * 256u + threadIdx.x; e < n; e += (size_t)gridDim.x * 256u) {
        int x = (int)(e % (size_t)nx);
        size_t rest = e / (size_t)nx;
        int y = (int)(rest % (size_t)ny);
        int z = (int)(rest / (size_t)ny);
        int i = x + VOL_PAD, j = y + VOL_PAD, k = z + VOL_PAD;
        size_t o;
        if (layout == LAYOUT_LINEAR) o = ((size_t)k * sz + (size_t)j * sy) + (size_t)i;
        else {
            size_t X = ((size_t)(i >> 3) << 7) + (size_t)(i & 7);
            size_t Y = (size_t)(j >> 2) * ((size_t)bnx << 7) + (size_t)((j & 3) << 3);
            size_t Z = (size_t)(k >> 2) * (((size_t)bny * bnx) << 7) + (size_t)((k & 3) << 5);
            o = X + Y + Z;
        }
        dst[o] = src[e];
    }
}

// PAIR layout: element (i, j, k) of the padded brick grid = voxel x | voxel (x + 1) << 16 (x = i - VOL_PAD; voxels
// outside the volume are border texels = 0); dst (32-bit elements) must be zero-filled before the launch
__global__ __launch_bounds__(256) void k_repack_pair(const uint16_t* __restrict__ src, uint32_t* __restrict__ dst, int nx, int ny, int nz, int bnx, int bny)
{
    const size_t ex = (size_t)nx + 1;                      // x = -1 .. nx - 1 have a non-zero half
    const size_t n = ex * ny * nz;
    for (size_t e = (size_t)blockIdx.x * 256u + threadIdx.x; e < n; e += (size_t)gridDim.x * 256u) {
        const int x = (int)(e % ex) - 1;
        const size_t rest = e / ex;
        const int y = (int)(rest % (size_t)ny), z = (int)(rest / (size_t)ny);
        const size_t row = ((size_t)z * ny + y) * nx;
        const uint32_t lo = x >= 0 ? src[row + x] : 0u, hi = x + 1 < nx ? src[row + x + 1] : 0u;
        const int i = x + VOL_PAD, j = y + VOL_PAD, k = z + VOL_PAD;
        const size_t X = ((size_t)(i >> 3) << 7) + (size_t)(i & 7);
        const size_t Y = (size_t)(j >> 2) * ((size_t)bnx << 7) + (size_t)((j & 3) << 3);
        const size_t Z = (size_t)(k >> 2) * (((size_t)bny * bnx) << 7) + (size_t)((k & 3) << 5);
        dst[X + Y + Z] = lo | (hi << 16);
    }
}

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------
template <int LAYOUT, bool COUNT>
static hipError_t launch_pathtrace_t(const DevScene& s, const DevWork& w, const LaunchCfg& cfg, hipStream_t st)
{
    uint32_t wv = w.x1 - w.x0;
    if (wv == 0 || w.n_rows == 0) return hipSuccess;
    if (cfg.kernel == KERNEL_PIXEL) {
        uint32_t blocks = ((wv + 15u) >> 4) * ((w.n_rows + 15u) >> 4) * w.nframes;
        hipLaunchKernelGGL((k_pathtrace_pixel<LAYOUT, COUNT>), dim3(blocks), dim3(256), 0, st, s, w);
    } else {
        uint32_t n_tiles = ((wv + 7u) >> 3) * ((w.n_rows + 7u) >> 3) * w.nframes;
        uint32_t max_blocks = (uint32_t)(cfg.num_cus * cfg.blocks_per_cu);
        uint32_t need = (n_tiles + 3u) / 4u;                    // 4 waves per block, >= 1 tile per wave
        uint32_t blocks = need < max_blocks ? need : max_blocks;
        if (blocks == 0) blocks = 1;
        hipError_t e = hipMemsetAsync(w.ticket, 0, sizeof(uint32_t), st);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((k_pathtrace_uloop<LAYOUT, COUNT>), dim3(blocks), dim3(256), 0, st, s, w);
    }
    return hipGetLastError();
}

hipError_t launch_pathtrace(const DevScene& s, const DevWork& w, const LaunchCfg& cfg, hipStream_t st)
{
    if (s.layout == LAYOUT_CELL)
        return cfg.count ? launch_pathtrace_t<LAYOUT_CELL, true>(s, w, cfg, st) : launch_pathtrace_t<LAYOUT_CELL, false>(s, w, cfg, st);
    if (s.layout == LAYOUT_PAIR)
        return cfg.count ? launch_pathtrace_t<LAYOUT_PAIR, true>(s, w, cfg, st) : launch_pathtrace_t<LAYOUT_PAIR, false>(s, w, cfg, st);
    if (s.layout == LAYOUT_LINEAR)
        return cfg.count ? launch_pathtrace_t<LAYOUT_LINEAR, true>(s, w, cfg, st) : launch_pathtrace_t<LAYOUT_LINEAR, false>(s, w, cfg, st);
    return cfg.count ? launch_pathtrace_t<LAYOUT_BRICK, true>(s, w, cfg, st) : launch_pathtrace_t<LAYOUT_BRICK, false>(s, w, cfg, st);
}

// The resolve of ONE frame over the WHOLE image -- what a render_pathtracer call does when its frame was traced ahead (svr_api.hip) -- beside
// a persistent trace kernel that holds every CU: the queue builds of the tile kernel leave 32 VGPRs per SIMD free, so a resolve wave is
// resident only if it needs <= 32 registers, and then there is ONE of it per SIMD (k_resolve, 18 registers, one pixel per lane: 12 + 12
// bytes in flight per lane, 150 us per frame measured inside a trace launch).  The running mean does not care about channels, so it runs
// over the frame as a flat float array, 16 bytes per lane and load (k_mean_flat: <= 16 registers, two waves per SIMD), and the tone map
// follows as k_tonemap (8 registers, four waves per SIMD).  Same float operations per value as k_resolve.
__global__ __launch_bounds__(256) void k_mean_flat(float4* __restrict__ hdr, const float4* __restrict__ L, uint32_t n4, uint32_t frame0, uint32_t nan_guard)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n4) return;
    const float4 l = L[i];
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (frame0 != 0u) a = hdr[i];
    float lx = l.x, ly = l.y, lz = l.z, lw = l.w;
    if (nan_guard) {
        lx = (f2u(lx) & 0x7f800000u) == 0x7f800000u ? a.x : lx;
        ly = (f2u(ly) & 0x7f800000u) == 0x7f800000u ? a.y : ly;
        lz = (f2u(lz) & 0x7f800000u) == 0x7f800000u ? a.z : lz;
        lw = (f2u(lw) & 0x7f800000u) == 0x7f800000u ? a.w : lw;
    }
    const float n1 = (float)frame0 + 1.f;
    a.x = a.x + (lx - a.x) / n1;
    a.y = a.y + (ly - a.y) / n1;
    a.z = a.z + (lz - a.z) / n1;
    a.w = a.w + (lw - a.w) / n1;
    hdr[i] = a;
}

hipError_t launch_resolve(const DevScene& s, const DevWork& w, hipStream_t st)
{
    if (w.n_items == 0) return hipSuccess;
    const uint64_t n = (uint64_t)3 * s.imageW * s.imageH;
    if (w.nframes == 1u && w.n_items == s.imageW * s.imageH && (n & 3u) == 0u && ((uintptr_t)w.hdr & 15u) == 0u && ((uintptr_t)w.lbuf & 15u) == 0u) {
        const uint32_t n4 = (uint32_t)(n >> 2);
        hipLaunchKernelGGL(k_mean_flat, dim3((n4 + 255u) / 256u), dim3(256), 0, st, reinterpret_cast<float4*>(w.hdr), reinterpret_cast<const float4*>(w.lbuf), n4, w.frame0, w.nan_guard);
        if (w.img) hipLaunchKernelGGL(k_tonemap, dim3((w.n_items + 255u) / 256u), dim3(256), 0, st, s, w);
        return hipGetLastError();
    }
    uint32_t blocks = (w.n_items + 255u) / 256u;
    hipLaunchKernelGGL(k_resolve, dim3(blocks), dim3(256), 0, st, s, w);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Sampling table of an environment map (SVR_OPT_ENV_NEE, svr_trace_env.hip): weight of texel (i, j) = the largest luminance among the texel
// and its 8 wrap-neighbours (the bilinear lookup at any point of the texel's cell reads only those) x sin(theta of the row) + a floor of
// 1e-3 of the mean, so that the density is positive wherever the lookup can be; cdf = h rows of w + 1 prefix sums, then h + 1 prefix sums
// of the row totals.  Built once per map.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_env_weights(const float4* __restrict__ env, int w, int h, float* __restrict__ wgt, float* __restrict__ mean_acc)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    float v = 0.f;
    if (e < w * h) {
        const int i = e % w, j = e / w;
        float m = 0.f;
        for (int dj = -1; dj <= 1; ++dj)
            for (int di = -1; di <= 1; ++di) {
                const int ii = (i + di + w) % w, jj = (j + dj + h) % h;
                const float4 t = env[jj * w + ii];
                const float lum = 0.2126f * t.x + 0.7152f * t.y + 0.0722f * t.z;
                m = fmaxf(m, lum > 0.f ? lum : 0.f);              // (a NaN or negative texel counts as black)
            }
        v = m * __builtin_sinf(3.14159265358979f * ((float)j + 0.5f) / (float)h);
        wgt[e] = v;
    }
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(mean_acc, v);
}
__global__ __launch_bounds__(64) void k_env_rows(const float* __restrict__ wgt, int w, int h, const float* __restrict__ mean_acc, float* __restrict__ cdf)
{
    const int j = blockIdx.x * 64 + threadIdx.x;
    if (j >= h) return;
    const float fl = fmaxf(1e-3f * (*mean_acc) / ((float)w * (float)h), 1e-30f);
    float* row = cdf + (size_t)j * (size_t)(w + 1);
    float acc = 0.f;
    row[0] = 0.f;
    for (int i = 0; i < w; ++i) { acc += wgt[(size_t)j * w + i] + fl; row[i + 1] = acc; }
}
__global__ void k_env_marginal(int w, int h, float* __restrict__ cdf)
{
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    float* marg = cdf + (size_t)h * (size_t)(w + 1);
    float acc = 0.f;
    marg[0] = 0.f;
    for (int j = 0; j < h; ++j) { acc += cdf[(size_t)j * (size_t)(w + 1) + w]; marg[j + 1] = acc; }
}
hipError_t launch_env_cdf(const float* env_rgba, int w, int h, float* cdf, float* tmp, hipStream_t st)
{
    // tmp: w * h + 1 floats (weights, then the sum)
    hipError_t e = hipMemsetAsync(tmp + (size_t)w * h, 0, sizeof(float), st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_env_weights, dim3((w * h + 255) / 256), dim3(256), 0, st, reinterpret_cast<const float4*>(env_rgba), w, h, tmp, tmp + (size_t)w * h);
    hipLaunchKernelGGL(k_env_rows, dim3((h + 63) / 64), dim3(64), 0, st, tmp, w, h, tmp + (size_t)w * h, cdf);
    hipLaunchKernelGGL(k_env_marginal, dim3(1), dim3(1), 0, st, w, h, cdf);
    return hipGetLastError();
}

hipError_t launch_tonemap(const DevScene& s, const DevWork& w, hipStream_t st)
{
    if (w.n_items == 0) return hipSuccess;
    uint32_t blocks = (w.n_items + 255u) / 256u;
    hipLaunchKernelGGL(k_tonemap, dim3(blocks), dim3(256), 0, st, s, w);
    return hipGetLastError();
}

// CELL layout: element (i, j, k) of the padded brick grid = the 8 voxels of the trilinear cell whose lowest corner is voxel
// (i, j, k) - VOL_PAD, as four PAIR words (x | x + 1 << 16) for (y, z), (y + 1, z), (y, z + 1), (y + 1, z + 1); voxels outside the
// volume are border texels = 0.  Every element of the grid is written (no zero fill needed).
__global__ __launch_bounds__(256) void k_repack_cell(const uint16_t* __restrict__ src, uint4* __restrict__ dst, int nx, int ny, int nz, int bnx, int bny, int bnz)
{
    const size_t n = (size_t)bnx * bny * bnz * 128u;
    for (size_t e = (size_t)blockIdx.x * 256u + threadIdx.x; e < n; e += (size_t)gridDim.x * 256u) {
        const size_t brick = e >> 7;
        const uint32_t in = (uint32_t)(e & 127u);
        const int bi = (int)(brick % (size_t)bnx), bj = (int)((brick / (size_t)bnx) % (size_t)bny), bk = (int)(brick / ((size_t)bnx * bny));
        const int x = (bi << 3) + (int)(in & 7u) - VOL_PAD, y = (bj << 2) + (int)((in >> 3) & 3u) - VOL_PAD, z = (bk << 2) + (int)(in >> 5) - VOL_PAD;
        auto vox = [&](int xx, int yy, int zz) -> uint32_t {
            return (xx >= 0 && yy >= 0 && zz >= 0 && xx < nx && yy < ny && zz < nz) ? (uint32_t)src[((size_t)zz * ny + yy) * nx + xx] : 0u;
        };
        uint4 c;
        c.x = vox(x, y, z) | (vox(x + 1, y, z) << 16);
        c.y = vox(x, y + 1, z) | (vox(x + 1, y + 1, z) << 16);
        c.z = vox(x, y, z + 1) | (vox(x + 1, y, z + 1) << 16);
        c.w = vox(x, y + 1, z + 1) | (vox(x + 1, y + 1, z + 1) << 16);
        dst[e] = c;
    }
}

// frame assembly of row-sharded renders (svr_assemble_frame): the p-th owned row of `rank` is row (q * world + rank) * strip + p % strip, q = p / strip
__global__ __launch_bounds__(256) void k_strips(float* __restrict__ packed, float* __restrict__ frame, uint32_t row_floats, uint32_t n_rows,
                                                uint32_t strip_rows, uint32_t rank, uint32_t world, int to_packed)
{
    const size_t n = (size_t)n_rows * row_floats;
    for (size_t e = (size_t)blockIdx.x * 256u + threadIdx.x; e < n; e += (size_t)gridDim.x * 256u) {
        const uint32_t p = (uint32_t)(e / row_floats), col = (uint32_t)(e - (size_t)p * row_floats);
        const uint32_t q = p / strip_rows;
        const uint32_t y = world <= 1u ? p : (q * world + rank) * strip_rows + (p - q * strip_rows);
        const size_t f = (size_t)y * row_floats + col;
        if (to_packed) packed[e] = frame[f]; else frame[f] = packed[e];
    }
}

hipError_t launch_strips(float* packed, float* frame, uint32_t row_floats, uint32_t n_rows, uint32_t strip_rows, uint32_t rank, uint32_t world,
                         int to_packed, hipStream_t st)
{
    if (n_rows == 0 || row_floats == 0) return hipSuccess;
    const size_t n = (size_t)n_rows * row_floats;
    const uint32_t blocks = (uint32_t)((n + 255u) / 256u < 4096u ? (n + 255u) / 256u : 4096u);
    hipLaunchKernelGGL(k_strips, dim3(blocks), dim3(256), 0, st, packed, frame, row_floats, n_rows, strip_rows ? strip_rows : 1u, rank, world, to_packed);
    return hipGetLastError();
}

hipError_t launch_repack(const uint16_t* src, uint16_t* dst, int nx, int ny, int nz, int layout,
                         int sy, int sz, int bnx, int bny, hipStream_t st)
{
    if (layout == LAYOUT_CELL) {
        const int bnz = (nz + 2 * VOL_PAD + BRICK_Z - 1) / BRICK_Z;
        hipLaunchKernelGGL(k_repack_cell, dim3(8192), dim3(256), 0, st, src, reinterpret_cast<uint4*>(dst), nx, ny, nz, bnx, bny, bnz);
        return hipGetLastError();
    }
    if (layout == LAYOUT_PAIR) {
        hipLaunchKernelGGL(k_repack_pair, dim3(4096), dim3(256), 0, st, src, reinterpret_cast<uint32_t*>(dst), nx, ny, nz, bnx, bny);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(k_repack, dim3(4096), dim3(256), 0, st, src, dst, nx, ny, nz, layout, sy, sz, bnx, bny);
    return hipGetLastError();
}

} // namespace svr
