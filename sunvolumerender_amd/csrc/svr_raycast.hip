// svr_raycast.hip -- kernel_raycasting (raycasting.cu:15-67): front-to-back emission/absorption
// compositing with a head-light Phong term, fixed step stepSize/2, early exit at opacity > 0.95.
//
// The reference runs one thread per pixel; a frame then takes as long as its longest ray (~1200 shaded
// samples of ~1200 instructions each through a translucent 512^3 volume: 5 ms), with the chip almost
// empty behind it.  Here S = 2^SL2 adjacent lanes share one ray: lane j evaluates the samples
// n = j, j+S, j+2S ... (the sample parameters are the reference's float chain t += h, replayed by every
// lane), and the S colours of a chunk are composited in order by all S lanes redundantly, so the
// accumulation L += (1 - L.a) * c is the reference's, sample by sample.  Samples past the early exit are
// evaluated speculatively and dropped.
//
// Empty space: a sample whose trilinear cell lies in an `empty` macro-cell (svr_accel.hip, bitmask in
// LDS), or whose opacity lookup returns exactly 0, contributes exactly +0 to (L.rgb, L.a), so its six
// gradient taps and the shading are skipped (the first kind also skips the intensity tap).  Each ray
// first advances chunk by chunk through such samples in a cheap loop; the wave shades when every ray
// of the wave has a contributing sample pending (or has finished).  Transfer-function colours are
// assumed finite (0 * inf would differ).
//
// Long empty stretches are not stepped through at all: a sphere trace on the distance field (svr_walk.hpp) tells how far
// the ray is clear, and the sample chain is replayed in closed form up to there (chain_* below).
//
// Persistent 1024-thread blocks pull tasks (64 >> SL2 pixels each) from sharded tickets, as k_trace_tile.
#include "svr_walk.hpp"

namespace svr {

#define SVR_RC_THREADS 1024    // 16 waves share the RGBA table, the two bitmasks and the distance field (96 KB): one block per CU

struct LdsRaycast {
    float4 rgba[SVR_TF_MAX + SVR_TF_PAD];      // entry e = texel clamp(e-1)
    uint32_t dist[DIST_WORDS_MAX];             // svr_walk.hpp: distance field, deep-empty bits, empty bits
    uint32_t mask[MASK_WORDS_MAX];
    uint32_t emask[MASK_WORDS_MAX];
};


// ------------------------------------------------------------------------------------------------------------------
// The reference's sample parameters are a float accumulation chain t_{n+1} = fl(t_n + h) (raycasting.cu:63).  To skip
// samples without evaluating them the chain has to be replayed exactly.  Within one binade [2^e, 2^(e+1)) every t is
// a multiple of u = ulp, and fl(t + h) = t + delta with ONE delta for the whole binade (h = q u + r rounds to q u or
// (q + 1) u, the same way for every t, unless r is exactly u / 2 where ties-to-even alternates).  So k steps inside a
// binade are t + k * delta, computed on the integer mantissas; the step that crosses into the next binade is taken
// with a real float addition.  Chains the closed form does not cover (t < 1, a tie, delta == 0) are simply not
// skipped: the caller falls back to one chunk at a time.
// ------------------------------------------------------------------------------------------------------------------
struct ChainSeg { uint32_t A, J, kmax; float u; bool ok; };     // t = A u, delta = J u, steps k <= kmax stay in the binade

SVR_DEV ChainSeg chain_segment(float t, float h)
{
    ChainSeg g;
    const uint32_t tb = __float_as_uint(t);
    const uint32_t ex = tb >> 23;                                   // t >= 1: sign 0, exponent >= 127
    g.u = __uint_as_float((ex - 23u) << 23);
    g.A = (tb & 0x7fffffu) | 0x800000u;
    const float t1 = t + h;
    const float delta = t1 - t;                                     // exact
    const float err = h - delta;                                    // exact: the rounding error of t + h
    const bool same = (__float_as_uint(t1) >> 23) == ex;
    g.J = (uint32_t)(delta * __uint_as_float((127u + 127u + 23u - ex) << 23));        // delta / u, an integer < 2^24
    g.ok = same && delta > 0.f && __builtin_fabsf(err) != 0.5f * g.u && t >= 1.f && ex < 127u + 100u;
    g.kmax = g.ok ? (0xffffffu - g.A) / g.J : 0u;
    return g;
}

// number of chain elements t^[0] = t, t^[1], ... that are < bound (inclusive: <= bound); cap = false if the chain left
// the closed form before the answer was known (then the count is a lower bound that is still safe to skip)
SVR_DEV uint32_t chain_count(float t, float h, float bound, bool inclusive, bool& exact)
{
    uint32_t n = 0;
    exact = true;
    for (int seg = 0; seg < 6; ++seg) {
        if (!(inclusive ? t <= bound : t < bound)) return n;
        const ChainSeg g = chain_segment(t, h);
        if (!g.ok) {
            // one real step (binade crossing) -- or give up on anything the closed form does not cover
            const float t1 = t + h;
            if (!(t1 > t) || !(t >= 1.f)) { exact = false; return n; }
            n += 1u; t = t1;
            continue;
        }
        // elements t + k delta, k = 0 .. kmax, of this binade that are before the bound
        const float xs = bound * __uint_as_float((254u - (__float_as_uint(g.u) >> 23)) << 23);      // bound / u (exact scaling)
        uint32_t k_in;
        if (xs >= 16777216.f) k_in = g.kmax + 1u;                  // the bound lies beyond this binade
        else {
            const float fl = __builtin_floorf(xs);
            uint32_t X = (uint32_t)fl;                               // t is before the bound, so X >= A >= 2^23
            if (!inclusive && fl == xs) X -= 1u;                   // strict: A + k J <= ceil(x) - 1
            k_in = X >= g.A ? (X - g.A) / g.J + 1u : 0u;
            if (k_in > g.kmax + 1u) k_in = g.kmax + 1u;
        }
        n += k_in;
        if (k_in <= g.kmax) return n;                              // the bound was met inside the binade
        // all kmax + 1 elements counted: continue from the first element of the next binade
        t = (float)(g.A + g.kmax * g.J) * g.u;
        t = t + h;
    }
    exact = false;
    return n;
}

// t after n chain steps; ok = false if the closed form gave up (t is then unchanged)
SVR_DEV float chain_advance(float t, float h, uint32_t n, bool& ok)
{
    const float t_in = t;
    ok = true;
    for (int seg = 0; seg < 8 && n != 0u; ++seg) {
        const ChainSeg g = chain_segment(t, h);
        if (!g.ok) {
            const float t1 = t + h;
            if (!(t1 > t) || !(t >= 1.f)) { ok = false; return t_in; }
            t = t1; n -= 1u;
            continue;
        }
        const uint32_t k = n < g.kmax ? n : g.kmax;
        t = (float)(g.A + k * g.J) * g.u;                          // exact: an integer below 2^24 times a power of two
        n -= k;
        if (n != 0u) { t = t + h; n -= 1u; }                       // the crossing step
    }
    if (n != 0u) { ok = false; return t_in; }
    return t;
}

template <int LAYOUT, bool COUNT, bool SKIP, int SL2>
__global__ __launch_bounds__(SVR_RC_THREADS) void k_raycast(const DevScene s, const DevWork w, float stepSize)
{
    constexpr uint32_t S = 1u << SL2;                  // lanes per ray
    constexpr uint32_t P2 = 6u - SL2;                  // log2(rays per wave)
    constexpr uint32_t tw2 = P2 < 3u ? P2 : 3u, th2 = P2 - tw2;   // pixel block of a task: up to 8 wide
    constexpr uint32_t GMASK = S == 32u ? 0xffffffffu : ((1u << S) - 1u);
    static_assert(SL2 >= 0 && SL2 <= 5, "1..32 lanes per ray");

    __shared__ LdsRaycast L;
    {
        const int n = s.tf_n;
        const float4* g = reinterpret_cast<const float4*>(s.tf);
        for (int e = threadIdx.x; e < n + SVR_TF_PAD; e += SVR_RC_THREADS) L.rgba[e] = g[min(max(e - 1, 0), n - 1)];
        if (SKIP) {
            const uint4* src = reinterpret_cast<const uint4*>(s.empty_mask + DIST_WORDS_MAX + MASK_WORDS_MAX);
            uint4* dst = reinterpret_cast<uint4*>(L.emask);
            for (uint32_t q = threadIdx.x; q < (s.mask_words + 3u) / 4u; q += SVR_RC_THREADS) dst[q] = src[q];
            const uint4* src1 = reinterpret_cast<const uint4*>(s.empty_mask + DIST_WORDS_MAX);
            uint4* dst1 = reinterpret_cast<uint4*>(L.mask);
            for (uint32_t q = threadIdx.x; q < (s.mask_words + 3u) / 4u; q += SVR_RC_THREADS) dst1[q] = src1[q];
            const uint4* src0 = reinterpret_cast<const uint4*>(s.empty_mask);
            uint4* dst0 = reinterpret_cast<uint4*>(L.dist);
            for (uint32_t q = threadIdx.x; q < (s.dist_words + 3u) / 4u; q += SVR_RC_THREADS) dst0[q] = src0[q];
        }
        __syncthreads();
    }
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t j = lane & (S - 1u), ray = lane >> SL2, gshift = ray << SL2;
    const uint32_t wv = w.x1 - w.x0;
    const uint32_t tiles_x = (wv + (1u << tw2) - 1u) >> tw2;
    const uint32_t tiles_y = (w.n_rows + (1u << th2) - 1u) >> th2;
    const uint32_t n_tasks = tiles_x * tiles_y;
    const uint32_t per_shard = (n_tasks + TICKET_SHARDS - 1u) / TICKET_SHARDS;
    const uint32_t shard0 = blockIdx.x % TICKET_SHARDS;
    const v3 cam = V3(s.cam_pos[0], s.cam_pos[1], s.cam_pos[2]);
    const float h = stepSize * 0.5f;
    uint32_t n_steps = 0, n_shaded = 0, n_fetched = 0;

    for (uint32_t si = 0; si < TICKET_SHARDS; ++si) {
        const uint32_t shard = (shard0 + si) % TICKET_SHARDS;
        const uint32_t t_begin = shard * per_shard;
        const uint32_t t_count = t_begin >= n_tasks ? 0u : min(per_shard, n_tasks - t_begin);
        uint32_t* ticket = w.ticket + shard * TICKET_STRIDE;
        for (;;) {
            uint32_t u = 0;
            if (lane == 0) u = atomicAdd(ticket, 1u);
            u = __builtin_amdgcn_readfirstlane(u);
            if (u >= t_count) break;
            const uint32_t task = t_begin + u;
            const uint32_t ty = task / tiles_x, tx = task - ty * tiles_x;
            const uint32_t px = (tx << tw2) + (ray & ((1u << tw2) - 1u));
            const uint32_t r = (ty << th2) + (ray >> tw2);
            const bool act = px < wv && r < w.n_rows;
            uint32_t x = 0, y = 0;
            v3 orig = cam, dir = V3(0.f, 0.f, 1.f);
            float tNear = 0.f, tFar = -1.f;
            bool done = true;
            if (act) {
                x = w.x0 + px; y = owned_row_to_y(w, r);
                camera_ray_pinhole(s, x, y, orig, dir);
                done = !volume_intersect(s, orig, dir, tNear, tFar);
            }
            float Lr = 0.f, Lg = 0.f, Lb = 0.f, La = 0.f;
            uint32_t steps = 0;
            // parameter of this lane's sample in the current chunk: t_n, n = chunk * S + j
            float t = tNear;
#pragma unroll
            for (uint32_t i = 0; i + 1u < S; ++i) t = (i < j) ? t + h : t;
            uint32_t gv = 0, gc = 0;             // valid / contributing lanes of this ray's current chunk
            // jumps need the whole box inside the texture domain (s.ray_skip) and sample points that cannot coincide
            // with the eye (t >= 1 with |orig| < 2^20: dir * t is never absorbed)
            const bool jump_ok = SKIP && s.ray_skip && fmax_(__builtin_fabsf(orig.x), fmax_(__builtin_fabsf(orig.y), __builtin_fabsf(orig.z))) < 1048576.f;
            v3 p = orig;
            int e = 0;
            float a = 0.f, alpha = 0.f;

            for (;;) {
                // ---- advance: rays with nothing pending evaluate chunks until one contributes ----
                for (;;) {
                    const bool need = !done && gc == 0u;
                    if (__ballot(need) == 0ull) break;
                    const bool valid = need && t <= tFar;
                    bool contrib = false, deep = false;
                    if (valid) {
                        v3 q = orig + dir * t;
                        Cell c = cell_of(s, q);
                        v3 toCam = cam - q;
                        // the head-light direction must stay finite for "opacity 0 => contributes +0"
                        const bool finite = dot(toCam, toCam) >= 1e-30f;
                        if (SKIP && jump_ok) deep = finite && cell_is_empty<true>(L, s, c);
                        if (!(SKIP && finite && cell_is_empty<false>(L, s, c))) {
                            if (COUNT) n_fetched++;
                            float intensity = tex_fetch<LAYOUT>(s, c) * s.densityScale;
                            int ee; float aa;
                            lds_tf_coord(s, intensity, ee, aa);
                            float al = lerpf(L.rgba[ee].w, L.rgba[ee + 1].w, aa);
                            if (!(SKIP && finite && al == 0.f)) {
                                contrib = true;
                                p = q; e = ee; a = aa; alpha = al;
                            }
                        }
                    }
                    const uint64_t vb = __ballot(valid), cb = __ballot(contrib), db = __ballot(deep);
                    if (need) {
                        gv = (uint32_t)(vb >> gshift) & GMASK;
                        gc = (uint32_t)(cb >> gshift) & GMASK;
                        if (gc == 0u) {
                            const uint32_t nv = (uint32_t)__builtin_popcount(gv);
                            steps += nv;
                            if (nv < S) done = true;
#pragma unroll
                            for (uint32_t i = 0; i < S; ++i) t += h;
                            // the whole chunk sat in deep-empty cells: look ahead and replay the chain up to there
                            if (SKIP && !done && (((uint32_t)(db >> gshift) & GMASK) == GMASK)) {
                                const float t0 = __shfl(t, (int)gshift, 64);             // the chunk's first sample
                                if (t0 >= 1.f && t0 <= tFar) {
                                    const float t_clear = first_occupied(s, L, orig, dir, t0, tFar);
                                    const bool to_end = t_clear == u2f(SVR_INF_BITS);
                                    bool exact;
                                    const uint32_t cnt = chain_count(t, h, to_end ? tFar : t_clear, to_end, exact);
                                    uint32_t mine = (cnt + S - 1u) >> SL2;                // samples of this lane before the bound
                                    uint32_t lo = mine, sum = mine;
                                    int all_exact = exact ? 1 : 0;          // (no &&: every lane must take part in every shuffle)
#pragma unroll
                                    for (uint32_t o = 1u; o < S; o <<= 1) {
                                        lo = min(lo, (uint32_t)__shfl_xor((int)lo, (int)o, 64));
                                        sum += (uint32_t)__shfl_xor((int)sum, (int)o, 64);
                                        all_exact &= __shfl_xor(all_exact, (int)o, 64);
                                    }
                                    if (to_end && all_exact != 0) {
                                        steps += sum;                                     // every remaining sample is transparent
                                        done = true;
                                    } else if (lo != 0u) {
                                        bool ok;
                                        const float tn = chain_advance(t, h, lo << SL2, ok);
                                        int all_ok = ok ? 1 : 0;
#pragma unroll
                                        for (uint32_t o = 1u; o < S; o <<= 1) all_ok &= __shfl_xor(all_ok, (int)o, 64);
                                        if (all_ok != 0) { t = tn; steps += lo << SL2; }
                                    }
                                }
                            }
                        }
                    }
                }
                if (__ballot(!done) == 0ull) break;

                // ---- shade the pending contributing samples ----
                float co[4] = {0.f, 0.f, 0.f, 0.f};
                if (!done && ((gc >> j) & 1u)) {
                    if (COUNT) n_shaded++;
                    float4 t0 = L.rgba[e], t1 = L.rgba[e + 1];
                    co[0] = lerpf(t0.x, t1.x, a); co[1] = lerpf(t0.y, t1.y, a); co[2] = lerpf(t0.z, t1.z, a);
                    co[3] = alpha;
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
                }

                // ---- composite the chunk in sample order (every lane of the ray keeps the same L) ----
#pragma unroll
                for (uint32_t jj = 0; jj < S; ++jj) {
                    const int src = (int)(gshift + jj);
                    float c0 = __shfl(co[0], src, 64), c1 = __shfl(co[1], src, 64);
                    float c2 = __shfl(co[2], src, 64), c3 = __shfl(co[3], src, 64);
                    if (!done && ((gc >> jj) & 1u)) {
                        float wgt = 1.f - La;
                        Lr += wgt * c0; Lg += wgt * c1; Lb += wgt * c2; La += wgt * c3;
                        if (La > 0.95f) { done = true; steps += jj + 1u; }
                    }
                }
                if (!done) {
                    const uint32_t nv = (uint32_t)__builtin_popcount(gv);
                    steps += nv;
                    if (nv < S) done = true;
#pragma unroll
                    for (uint32_t i = 0; i < S; ++i) t += h;
                }
                gc = 0u;
            }

            if (act && j == 0u) {
                float cr = fmin_(Lr, 1.f), cg = fmin_(Lg, 1.f), cb = fmin_(Lb, 1.f);
                uint32_t rgba = to_u8(cr * 255) | (to_u8(cg * 255) << 8) | (to_u8(cb * 255) << 16) | (to_u8(255 * La) << 24);
                reinterpret_cast<uint32_t*>(w.img)[(size_t)y * s.imageW + x] = rgba;
                if (COUNT) n_steps += steps;
            }
        }
    }
    if (COUNT) {
        unsigned long long st = wave_sum((unsigned long long)n_steps);
        unsigned long long ex = wave_sum((unsigned long long)n_fetched + 6ull * n_shaded);
        if (lane == 0) {
            atomicAdd(&w.counters[CNT_RAYCAST], st);
            atomicAdd(&w.counters[CNT_VOL_TAPS], st * 7ull);
            atomicAdd(&w.counters[CNT_TAPS_EXEC], ex);
        }
    }
}

// unit-test hook for the chain primitives: item i = (t, h, bound, n) -> (count strict, count inclusive, exact flags, advanced t, ok)
__global__ void k_chain_selftest(const float4* __restrict__ in, float4* __restrict__ out, uint32_t n_items)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_items) return;
    const float4 q = in[i];
    bool e0, e1, ok;
    const uint32_t c0 = chain_count(q.x, q.y, q.z, false, e0);
    const uint32_t c1 = chain_count(q.x, q.y, q.z, true, e1);
    const float ta = chain_advance(q.x, q.y, (uint32_t)q.w, ok);
    out[i] = make_float4(__uint_as_float(c0), __uint_as_float(c1), ta, __uint_as_float((e0 ? 1u : 0u) | (e1 ? 2u : 0u) | (ok ? 4u : 0u)));
}

hipError_t launch_chain_selftest(const float4* in, float4* out, uint32_t n, hipStream_t st)
{
    hipLaunchKernelGGL(k_chain_selftest, dim3((n + 255u) / 256u), dim3(256), 0, st, in, out, n);
    return hipGetLastError();
}

template <int LAYOUT, bool COUNT, bool SKIP>
static void launch_s(const DevScene& s, const DevWork& w, float stepSize, int sl2, uint32_t max_blocks, hipStream_t st)
{
    const uint32_t wv = w.x1 - w.x0;
    auto go = [&](auto tag) {
        constexpr int SL2 = decltype(tag)::value;
        constexpr uint32_t P2 = 6u - SL2, tw2 = P2 < 3u ? P2 : 3u, th2 = P2 - tw2;
        uint32_t n_tasks = ((wv + (1u << tw2) - 1u) >> tw2) * ((w.n_rows + (1u << th2) - 1u) >> th2);
        uint32_t need = (n_tasks + SVR_RC_THREADS / 64 - 1u) / (SVR_RC_THREADS / 64);
        uint32_t blocks = need < max_blocks ? need : max_blocks;
        hipLaunchKernelGGL((k_raycast<LAYOUT, COUNT, SKIP, SL2>), dim3(blocks ? blocks : 1u), dim3(SVR_RC_THREADS), 0, st, s, w, stepSize);
    };
    switch (sl2) {
    case 0: go(std::integral_constant<int, 0>{}); break;
    case 1: go(std::integral_constant<int, 1>{}); break;
    case 2: go(std::integral_constant<int, 2>{}); break;
    case 4: go(std::integral_constant<int, 4>{}); break;
    case 5: go(std::integral_constant<int, 5>{}); break;
    default: go(std::integral_constant<int, 3>{}); break;
    }
}

hipError_t launch_raycast(const DevScene& s, const DevWork& w, float stepSize, bool count, int num_cus, int lanes_log2, hipStream_t st)
{
    if (w.x1 == w.x0 || w.n_rows == 0) return hipSuccess;
    hipError_t e = hipMemsetAsync(w.ticket, 0, sizeof(uint32_t) * TICKET_SHARDS * TICKET_STRIDE, st);
    if (e != hipSuccess) return e;
    const uint32_t max_blocks = (uint32_t)num_cus;                       // one 1024-thread block (16 waves, 96 KB LDS) per CU
    const bool skip = s.empty_mask != nullptr;
    if (s.layout == LAYOUT_CELL) {
        if (count) { if (skip) launch_s<LAYOUT_CELL, true, true>(s, w, stepSize, lanes_log2, max_blocks, st); else launch_s<LAYOUT_CELL, true, false>(s, w, stepSize, lanes_log2, max_blocks, st); }
        else { if (skip) launch_s<LAYOUT_CELL, false, true>(s, w, stepSize, lanes_log2, max_blocks, st); else launch_s<LAYOUT_CELL, false, false>(s, w, stepSize, lanes_log2, max_blocks, st); }
    } else if (s.layout == LAYOUT_PAIR) {
        if (count) { if (skip) launch_s<LAYOUT_PAIR, true, true>(s, w, stepSize, lanes_log2, max_blocks, st); else launch_s<LAYOUT_PAIR, true, false>(s, w, stepSize, lanes_log2, max_blocks, st); }
        else { if (skip) launch_s<LAYOUT_PAIR, false, true>(s, w, stepSize, lanes_log2, max_blocks, st); else launch_s<LAYOUT_PAIR, false, false>(s, w, stepSize, lanes_log2, max_blocks, st); }
    } else if (s.layout == LAYOUT_LINEAR) {
        if (count) { if (skip) launch_s<LAYOUT_LINEAR, true, true>(s, w, stepSize, lanes_log2, max_blocks, st); else launch_s<LAYOUT_LINEAR, true, false>(s, w, stepSize, lanes_log2, max_blocks, st); }
        else { if (skip) launch_s<LAYOUT_LINEAR, false, true>(s, w, stepSize, lanes_log2, max_blocks, st); else launch_s<LAYOUT_LINEAR, false, false>(s, w, stepSize, lanes_log2, max_blocks, st); }
    } else {
        if (count) { if (skip) launch_s<LAYOUT_BRICK, true, true>(s, w, stepSize, lanes_log2, max_blocks, st); else launch_s<LAYOUT_BRICK, true, false>(s, w, stepSize, lanes_log2, max_blocks, st); }
        else { if (skip) launch_s<LAYOUT_BRICK, false, true>(s, w, stepSize, lanes_log2, max_blocks, st); else launch_s<LAYOUT_BRICK, false, false>(s, w, stepSize, lanes_log2, max_blocks, st); }
    }
    return hipGetLastError();
}

} // namespace svr
