// svr_trace_env.hip -- OPT-IN importance sampling of the environment map (SVR_OPT_ENV_NEE, default off): SURVEY 8(f) row N4's second half.
//
// The reference has the lat-long lookup (core/lights/cuda_environment_light.h:58-72) and even that is disabled (pathtracer.cu:233);
// SVR_OPT_ENV_ON_ESCAPE enables it: a path that leaves the volume after bounce k adds T x env(dir).  So the environment lights the
// medium only through directions the BSDF / phase sampling happens to pick -- hopeless for a small bright sun.  Here every scatter event
// that is followed by a bounce (k = 0 .. traceDepth - 2) also draws ONE direction from the map's luminance (a piecewise-constant density
// over the texels, built on the GPU when the map is created: svr_kernels.hip k_env_cdf), walks a shadow ray along it, and the two ways of
// reaching the environment are combined with the balance heuristic (Veach): the sampled direction's estimate weighs p_env / (p_env + q),
// the escape term of the next bounce q / (q + p_env), q = the density the reference's own sampling code gives that direction.
//
// "The same image in expectation" is taken literally.  The escape estimator's contribution for a bounce direction w is T x W(w) x env(w)
// with W what pathtracer.cu:258-276 does to the throughput -- including its quirks: a lobe's f / pdf (pdf as the reference REPORTS it:
// sample_beckmann, core/bsdf/microfacet.h:95-111, draws cos(theta_h) = 1 / (1 - alpha^2 ln(1 - xi)), which is not the distribution its
// pdf describes), T left unchanged when f <= 0 or pdf <= 0, and Russian roulette's min(1, illum) / illum from the fourth bounce on.  The
// env sample estimates the same integral, sum over lobes of P(lobe) x q_lobe(w) x W_lobe(w) x env(w) x Tr(w), with q_lobe the TRUE density
// of the reference's sampling code (derived in bsdf_qw below).  Contract: converged images agree with the escape-only estimator
// (tests/test_env_nee_gpu.py: per-channel means of frame and quadrants within 4 standard errors at 4 096 spp on a scene lit by a small
// bright sun); inside the mode a frame is a pure function of (scene, pixel, frame).  Not bit-identical to the default mode: the extra
// draws shift the path's random stream.
//
// One kernel, straight-line paths, the tile kernel's task order; radiance goes to the scratch slots (k_resolve folds them).
#include "svr_walk.hpp"
#include "svr_lanes.hpp"
#include "svr_tile_tasks.hpp"

namespace svr {

#define SVR_SHADOW_REMARCH_ENV true      // (shadow walks re-march when they leave the occupied region, as in svr_trace_tile.hip)
constexpr float ENV_TWO_PI = 6.28318530717958647692f;
constexpr float ENV_PI = 3.14159265358979323846f;

// density (per solid angle) of the map's sampler at direction dir; 0 at the poles' last 1e-4 (the Jacobian 1 / sin(theta) explodes there:
// those directions are left to the escape term alone, consistently on both sides of the heuristic)
SVR_DEV float env_pdf(const DevScene& s, v3 dir)
{
    const float sinTheta = __builtin_sqrtf(fmax_(0.f, 1.f - dir.y * dir.y));
    if (!(sinTheta > 1e-4f)) return 0.f;
    // the texel the reference's lookup is centred in: cuda_environment_light.h:58-66 + wrap addressing
    float theta = acosf_(fmin_(fmax_(dir.y, -1.f), 1.f));
    float phi = atan2f_(dir.x, dir.z);
    phi = phi < 0.f ? phi + ENV_TWO_PI : phi;
    float u = phi * (0.5f / ENV_PI) + s.env_offset[0], v = theta * (1.f / ENV_PI) + s.env_offset[1];
    u = u - __builtin_floorf(u); v = v - __builtin_floorf(v);
    const int W = s.env_w, H = s.env_h;
    const int i = min(max((int)(u * (float)W), 0), W - 1), j = min(max((int)(v * (float)H), 0), H - 1);
    const float* row = s.env_cdf + (size_t)j * (size_t)(W + 1);
    const float total = s.env_cdf[(size_t)H * (size_t)(W + 1) + (size_t)H];
    const float wgt = row[i + 1] - row[i];
    return (wgt / total) * ((float)W * (float)H) / (2.f * ENV_PI * ENV_PI * sinTheta);
}

// one direction from the map: u1 picks the row (marginal), u2 the column (conditional), continuous inside the texel
SVR_DEV v3 env_sample(const DevScene& s, float u1, float u2)
{
    const int W = s.env_w, H = s.env_h;
    const float* marg = s.env_cdf + (size_t)H * (size_t)(W + 1);
    const float x1 = u1 * marg[H];
    int lo = 0, hi = H;                                         // largest j with marg[j] <= x1 (j < H)
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (marg[mid] <= x1) lo = mid; else hi = mid; }
    const int j = lo;
    const float dv = fmin_(fmax_((x1 - marg[j]) / (marg[j + 1] - marg[j]), 0.f), 0.99999f);
    const float* row = s.env_cdf + (size_t)j * (size_t)(W + 1);
    const float x2 = u2 * row[W];
    lo = 0; hi = W;
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (row[mid] <= x2) lo = mid; else hi = mid; }
    const int i = lo;
    const float du = fmin_(fmax_((x2 - row[i]) / (row[i + 1] - row[i]), 0.f), 0.99999f);
    float u = ((float)i + du) / (float)W - s.env_offset[0], v = ((float)j + dv) / (float)H - s.env_offset[1];
    v = v - __builtin_floorf(v);
    float sp, cp, st, ct;
    sincosf_(u * ENV_TWO_PI, &sp, &cp);
    sincosf_(v * ENV_PI, &st, &ct);
    return V3(st * sp, ct, st * cp);                            // theta = acos(dir.y), phi = atan2(dir.x, dir.z)
}

// terminate_with_raussian_roulette (pathtracer.cu:96-103) in expectation: survival probability x the 1 / illum it rescales by
SVR_DEV float rr_factor(v3 T, uint32_t k)
{
    if (k < 3u) return 1.f;
    const float illum = 0.2126f * T.x + 0.7152f * T.y + 0.0722f * T.z;
    if (!(illum > 0.f)) return 0.f;
    return illum >= 1.f ? 1.f / illum : 1.f;
}

// What sample_bsdf + the throughput update + roulette of bounce k (pathtracer.cu:133-169, 258-276) do with direction w, in expectation:
// returns sum over lobes of P(lobe) x q_lobe(w) x W_lobe(w) x rr, and q = sum P(lobe) x q_lobe(w), the density with which they pick w.
SVR_DEV v3 bsdf_qw(const Shade& vs, v3 w, v3 T, uint32_t k, float& q)
{
    const v3 color = V3(vs.color[0], vs.color[1], vs.color[2]);
    const v3 one = V3(1.f, 1.f, 1.f);
    if (vs.st == 0) {
        // hg_phase_sample_f with g = 0: uniform over the sphere, f = color x HG_ISO, pdf = HG_ISO; T *= f / (pdf x (1 - Pbrdf))
        q = 1.f / (4.f * ENV_PI);
        const bool upd = fmax_(color.x, fmax_(color.y, color.z)) > 0.f;
        const v3 Wl = upd ? color / (1.f - vs.Pbrdf) : one;
        return Wl * (q * rr_factor(T * Wl, k));
    }
    v3 n = normalize(vs.gradient);
    float cosT = dot(vs.wo, n);
    if (cosT < 0.f) { cosT = -cosT; n = -n; }
    const float ks = schlick_fresnel(1.f, SVR_IOR, cosT), kd = 1.f - ks, p = 0.25f + 0.5f * ks;
    const float cd = dot(n, w), cosTerm = __builtin_fabsf(cd);
    // Lambert lobe (lambert.h:20-24, sampling.h:47-56): cosine-weighted about n; pdf = |n.w| / pi is its true density on the upper hemisphere
    const float qd = fmax_(cd, 0.f) / ENV_PI;
    v3 Wd = one;
    {
        const v3 fd = (color * (1.f / ENV_PI)) * kd / (1.f - p);
        const float pdfd = cosTerm / ENV_PI;
        if (fmax_(fd.x, fmax_(fd.y, fd.z)) > 0.f && pdfd > 0.f) Wd = (fd * cosTerm) / (pdfd * vs.Pbrdf);
    }
    // microfacet lobe (microfacet.h:70-79, 95-111): half vector about n with cos(theta_h) = c = 1 / (1 - alpha^2 ln u), u uniform:
    // P(C <= c) = exp((c - 1) / (c alpha^2)), so its density per solid angle is exp((c - 1) / (c alpha^2)) / (2 pi c^2 alpha^2); flipped to
    // wo's side; w = reflect(-wo, wh) has density q_h / (4 wo.wh)
    const v3 wh = normalize(w + vs.wo);
    const float owh = dot(vs.wo, wh), c = __builtin_fabsf(dot(n, wh));
    const float a2 = SVR_ALPHA * SVR_ALPHA;
    float qs = 0.f;
    if (owh > 1e-6f && c > 1e-4f) qs = expf_((c - 1.f) / (c * a2)) / (2.f * ENV_PI * c * c * a2) / (4.f * owh);
    float Ws = 1.f;
    {
        const float fs = microfacet_brdf_f(w, vs.wo, n, SVR_IOR, SVR_ALPHA) * ks / p;
        const float pdfs = beckmann_distribution(n, wh, SVR_ALPHA) / (4.f * __builtin_fabsf(owh));
        if (fs > 0.f && pdfs > 0.f) Ws = (fs * cosTerm) / (pdfs * vs.Pbrdf);
    }
    q = p * qs + (1.f - p) * qd;
    return one * (p * qs * Ws * rr_factor(T * Ws, k)) + Wd * ((1.f - p) * qd * rr_factor(T * Wd, k));
}

// kernel_pathtracer's body (pathtracer.cu:205-277) with the environment term on escape and the env sample per scatter event
template <int LAYOUT, bool COUNT, bool SKIP, typename LDS>
SVR_DEV v3 trace_path_env(const DevScene& s, const LDS& L_, uint32_t x, uint32_t y, uint32_t traceDepth, uint32_t hashed, Cnt& c)
{
    Rng rng;
    rng_init(rng, hashed + (y * s.imageW + x));
    if (COUNT) c.paths++;
    v3 L = V3(0.f, 0.f, 0.f), T = V3(1.f, 1.f, 1.f);
    v3 orig, dir;
    camera_ray(s, x, y, rng, orig, dir);
    float ls_t;
    const int ls_id = nearest_light(s, orig, dir, ls_t);
    float esc_w = 1.f;                                    // weight of the escape term: 1 for the camera ray, the balance heuristic after a bounce
    for (uint32_t k = 0; k < traceDepth; ++k) {
        float tMin = (float)1e-6, tMax = SVR_FLT_MAX, val = 0.f;
        float t = walk<LAYOUT, COUNT, SKIP, false>(s, L_, orig, dir, rng, tMin, tMax, val, false, c);
        if (k == 0 && ls_id >= 0) {
            t = t < 0.f ? SVR_FLT_MAX : t;
            if (ls_t < t) {
                const DevLight& l = s.lights[ls_id];
                const float cosTerm = dot(V3(l.normal[0], l.normal[1], l.normal[2]), -dir);
                L = L + (T * V3(l.radiance[0], l.radiance[1], l.radiance[2])) * (cosTerm <= 0.f ? 0.f : 1.f);
                break;
            }
        }
        if (t < 0.f) {
            L = L + (T * env_radiance(s, dir)) * esc_w;
            break;
        }
        Shade vs;
        vs.wo = -dir;
        vs.pt = orig + dir * t;
        Nee ne;
        shade_event<LAYOUT, COUNT>(s, vs, val, rng, ne, c);
        if (ne.have) {
            float sMin = (float)1e-6, sMax = SVR_FLT_MAX, sval = 0.f;
            const float ts = walk<LAYOUT, COUNT, SKIP, SVR_SHADOW_REMARCH_ENV>(s, L_, vs.pt, ne.wi, rng, sMin, sMax, sval, true, c);
            const float Tr = ((ts > sMin) && (ts < sMax)) ? 0.f : 1.f;
            const float kf = Tr * (float)s.num_lights;
            const DevLight& l = s.lights[ne.light];
            L = L + T * (((ne.B * kf) * V3(l.radiance[0], l.radiance[1], l.radiance[2])) / ne.pdf);
        }
        if (k + 1u >= traceDepth) break;
        // ---- the env sample of this event, paired with the escape term of bounce k + 1 ----
        {
            const float u1 = rng_uniform(rng), u2 = rng_uniform(rng);
            const v3 we = env_sample(s, u1, u2);
            const float pe = env_pdf(s, we);
            if (pe > 0.f) {
                float q;
                const v3 QW = bsdf_qw(vs, we, T, k, q);
                const v3 Le = env_radiance(s, we);
                const v3 F = QW * Le;
                if (fmax_(F.x, fmax_(F.y, F.z)) > 0.f) {
                    float sMin = (float)1e-6, sMax = SVR_FLT_MAX, sval = 0.f;
                    const float ts = walk<LAYOUT, COUNT, SKIP, SVR_SHADOW_REMARCH_ENV>(s, L_, vs.pt, we, rng, sMin, sMax, sval, true, c);
                    const float Tr = ((ts > sMin) && (ts < sMax)) ? 0.f : 1.f;
                    L = L + (T * F) * (Tr / (pe + q));
                }
            }
        }
        v3 wi; float pdf = 0.f;
        const v3 f = bsdf_sample(vs, wi, pdf, rng);
        const float cosTerm = __builtin_fabsf(dot(normalize(vs.gradient), wi));
        {
            // weight of what this direction will find if it escapes (the throughput T is still the one the env sample used)
            float q;
            (void)bsdf_qw(vs, wi, T, k, q);
            const float pe = env_pdf(s, wi);
            esc_w = (q + pe) > 0.f ? q / (q + pe) : 1.f;
        }
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

constexpr uint32_t ENV_THREADS = 1024;
static_assert(TILE_WAVES == ENV_THREADS / 64, "16 waves per block share the LDS image");

template <int LAYOUT, bool COUNT, bool SKIP>
__global__ __launch_bounds__(ENV_THREADS, 4) void k_trace_env(const DevScene s, const DevWork w)
{
    using LDS = typename std::conditional<SKIP, LdsTileCull, LdsTileNoMask>::type;
    __shared__ LDS lds;
    lds_tile_load(lds, s, SKIP);
    const uint32_t lane = threadIdx.x & 63u;
    const TaskShape ts = task_shape(w);
    const uint32_t fl2 = ts.fl2, P2 = ts.P2, tw2 = ts.tw2, th2 = ts.th2, wv = ts.wv;
    const uint32_t n_tasks = ts.tiles_x * ts.tiles_y * ts.fgroups;
    const uint32_t shard0 = blockIdx.x % TICKET_SHARDS;
    Cnt c = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
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
            if (px < wv && r < w.n_rows && slot < w.nframes) {
                const uint32_t x = w.x0 + px, y = owned_row_to_y(w, r);
                const v3 L = trace_path_env<LAYOUT, COUNT, SKIP>(s, lds, x, y, w.traceDepth, wang_hash(w.frame0 + slot), c);
                float* o = w.lbuf + (size_t)slot * w.slot_stride + 3 * ((size_t)y * s.imageW + x);
                o[0] = L.x; o[1] = L.y; o[2] = L.z;
            }
        }
    }
    if (COUNT) cnt_flush(w, c);
}

template <int LAYOUT, bool COUNT>
static hipError_t launch_env_t(const DevScene& s, const DevWork& w, const LaunchCfg& cfg, hipStream_t st)
{
    const uint32_t wv = w.x1 - w.x0;
    if (wv == 0 || w.n_rows == 0) return hipSuccess;
    if (s.env == nullptr || s.env_cdf == nullptr || w.lbuf == nullptr) return hipErrorInvalidValue;
    uint32_t fl2 = 0;
    while (fl2 < 6u && (2u << fl2) <= w.nframes) ++fl2;
    const uint32_t P2 = 6u - fl2, tw2 = (P2 + 1u) >> 1, th2 = P2 >> 1;
    const uint32_t fgroups = (w.nframes + (1u << fl2) - 1u) >> fl2;
    const uint32_t n_tasks = ((wv + (1u << tw2) - 1u) >> tw2) * ((w.n_rows + (1u << th2) - 1u) >> th2) * fgroups;
    constexpr uint32_t WPB = ENV_THREADS / 64;
    const uint32_t max_blocks = (uint32_t)(cfg.num_cus * cfg.blocks_per_cu) * 4u / WPB;
    uint32_t blocks = (n_tasks + WPB - 1u) / WPB;
    if (blocks > max_blocks) blocks = max_blocks;
    if (blocks == 0) blocks = 1;
    DevWork w2 = w;
    w2.unit = 1u;
    w2.frames_log2 = fl2;
    hipError_t e = hipMemsetAsync(w.ticket, 0, sizeof(uint32_t) * TICKET_SHARDS * TICKET_STRIDE, st);
    if (e != hipSuccess) return e;
    if (s.empty_mask != nullptr) hipLaunchKernelGGL((k_trace_env<LAYOUT, COUNT, true>), dim3(blocks), dim3(ENV_THREADS), 0, st, s, w2);
    else hipLaunchKernelGGL((k_trace_env<LAYOUT, COUNT, false>), dim3(blocks), dim3(ENV_THREADS), 0, st, s, w2);
    return hipGetLastError();
}

hipError_t launch_trace_env(const DevScene& s, const DevWork& w, const LaunchCfg& cfg, hipStream_t st)
{
    if (s.layout == LAYOUT_CELL) return cfg.count ? launch_env_t<LAYOUT_CELL, true>(s, w, cfg, st) : launch_env_t<LAYOUT_CELL, false>(s, w, cfg, st);
    if (s.layout == LAYOUT_PAIR) return cfg.count ? launch_env_t<LAYOUT_PAIR, true>(s, w, cfg, st) : launch_env_t<LAYOUT_PAIR, false>(s, w, cfg, st);
    if (s.layout == LAYOUT_BRICK) return cfg.count ? launch_env_t<LAYOUT_BRICK, true>(s, w, cfg, st) : launch_env_t<LAYOUT_BRICK, false>(s, w, cfg, st);
    return cfg.count ? launch_env_t<LAYOUT_LINEAR, true>(s, w, cfg, st) : launch_env_t<LAYOUT_LINEAR, false>(s, w, cfg, st);
}

} // namespace svr
