// svr_accel.hip -- acceleration data for bit-exact empty-space skipping (see svr_trace_tile.hip).
//
// Macro-cell m (per axis, S = 2^shift) covers the trilinear cells c with c+1 in [m*S, m*S+S-1], i.e.
// cells c in [m*S-1, m*S+S-2], whose 8-voxel footprints span voxels [m*S-1, m*S+S-1] (voxel -1 and
// voxel N are border texels = 0); the LAST macro-cell of an axis also takes cell c = N-1 (footprint up
// to the border voxel N), so the grid covers every cell a point of the texture domain [0,1]^3 can map to.
// k_minmax stores the min/max raw voxel value over that footprint.
//
// k_empty_mask marks a macro-cell empty iff the transfer-function alpha is exactly 0 for EVERY
// intensity a fetch inside it can return.  Argument: each lerp fma(t, q-p, p) with t in [0,1) rounds
// monotonically and stays within [min(p,q), max(p,q)], so the filtered raw value lies in
// [rmin, rmax]; the two scalings (x 1/65535, x densityScale) and the LUT coordinate
// floor(fma(x, n, -0.5)) are monotonic, so the LUT entries a fetch can touch are
// e(Imin) .. e(Imax)+1; if all of those alphas are 0 the interpolated alpha is fma(a, 0, 0) = 0.
#include "svr_kernel_common.hpp"

namespace svr {

// pad > 0 (the WIDE table of the fast bound look-up, k_bound8): the footprint grows by `pad` voxels on every side.
__global__ __launch_bounds__(64) void k_minmax(const uint16_t* __restrict__ src, uint16_t* __restrict__ mm,
                                               int nx, int ny, int nz, int shift, int gx, int gy, int gz, int pad)
{
    const int S = 1 << shift;
    uint32_t m = blockIdx.x;
    int mx = (int)(m % (uint32_t)gx);
    int my = (int)((m / (uint32_t)gx) % (uint32_t)gy);
    int mz = (int)(m / ((uint32_t)gx * (uint32_t)gy));
    int x0 = mx * S - 1 - pad, y0 = my * S - 1 - pad, z0 = mz * S - 1 - pad;
    // voxels per axis in the footprint (+1 on the last macro-cell: border voxel N)
    int Ex = S + 1 + 2 * pad + (mx == gx - 1), Ey = S + 1 + 2 * pad + (my == gy - 1), Ez = S + 1 + 2 * pad + (mz == gz - 1);
    int total = Ex * Ey * Ez;
    uint32_t lo = 0xffffu, hi = 0u;
    for (int e = threadIdx.x; e < total; e += 64) {
        int dx = e % Ex, dy = (e / Ex) % Ey, dz = e / (Ex * Ey);
        int x = x0 + dx, y = y0 + dy, z = z0 + dz;
        uint32_t v = 0u;                 // border texel
        if (x >= 0 && y >= 0 && z >= 0 && x < nx && y < ny && z < nz)
            v = src[((size_t)z * ny + y) * nx + x];
        lo = min(lo, v);
        hi = max(hi, v);
    }
    for (int off = 32; off > 0; off >>= 1) {
        lo = min(lo, (uint32_t)__shfl_down((int)lo, off, 64));
        hi = max(hi, (uint32_t)__shfl_down((int)hi, off, 64));
    }
    if (threadIdx.x == 0) {
        mm[2 * (size_t)m] = (uint16_t)lo;
        mm[2 * (size_t)m + 1] = (uint16_t)hi;
    }
}

__global__ __launch_bounds__(256) void k_empty_mask(const uint16_t* __restrict__ mm, uint32_t n_cells,
                                                    const uint32_t* __restrict__ zero_prefix, int tf_n,
                                                    float densityScale, uint32_t* __restrict__ mask, uint32_t* __restrict__ n_empty)
{
    uint32_t m = blockIdx.x * 256u + threadIdx.x;
    if (m >= n_cells) return;
    float rlo = (float)mm[2 * (size_t)m], rhi = (float)mm[2 * (size_t)m + 1];
    // the same two multiplies as tex_fetch / intensity_at
    float ilo = (rlo * 1.5259021896696422e-05f) * densityScale;
    float ihi = (rhi * 1.5259021896696422e-05f) * densityScale;
    if (!(ilo == ilo) || !(ihi == ihi)) return;           // NaN scale: never skip
    if (ihi < ilo) { float t = ilo; ilo = ihi; ihi = t; } // negative densityScale
    float nf = (float)tf_n;
    // lds_tf_coord
    float xl = fmin_(fmax_(fma_(ilo, nf, -0.5f), -1.f), nf);
    float xh = fmin_(fmax_(fma_(ihi, nf, -0.5f), -1.f), nf);
    int e_lo = (int)__builtin_floorf(xl) + 1;
    int e_hi = (int)__builtin_floorf(xh) + 2;             // the pair (e, e+1) of the upper end
    // zero_prefix[e] = number of entries < e of the padded alpha table that are exactly 0
    uint32_t zeros = zero_prefix[e_hi + 1] - zero_prefix[e_lo];
    const bool empty = zeros == (uint32_t)(e_hi - e_lo + 1);
    if (empty) atomicOr(&mask[m >> 5], 1u << (m & 31u));
    if (n_empty != nullptr) {
        const uint64_t b = __ballot(empty);
        if (b != 0ull && (threadIdx.x & 63u) == (uint32_t)__builtin_ctzll(__ballot(true))) atomicAdd(n_empty, (uint32_t)__popcll(b));
    }
}

// MAJORANT-BOUND FETCH CULLING (bit-exact).  The reference's accept test is  xi < sigma_t * invSigmaMax
// (woodcock_tracking.h:43) with sigma_t = the transfer function's alpha at the fetched intensity.  With A(m) the largest
// alpha ANY fetch inside macro-cell m can return (k_empty_mask's argument: the LUT entries e(Imin) .. e(Imax)+1, and a
// lerp never leaves [min, max] of its two entries), sigma_t * invSigmaMax <= A(m) * invSigmaMax =: b(m) (the same
// float multiply, monotone).  The accept draw xi does not depend on the fetch, so a walk may draw it FIRST and fetch
// only if xi < b(m): otherwise the test fails whatever the voxels hold.  Cells where alpha is not exactly 0 but small
// -- noisy air in real CT data under any smooth transfer function -- then cost a fetch in a fraction b(m) of the
// iterations instead of all of them, and every path still consumes the reference's random numbers in the
// reference's order.  b is stored as a 4-bit class per HALF-resolution macro-cell (largest class of the 2x2x2
// children; 16 KB in LDS): thresholds thr[0] = 0, thr[c] = 2^((c-15)/2) for c = 1..14, thr[15] = +inf (always fetch:
// also covers a majorant that is violated, b > 1); class = the smallest c with b <= thr[c].
__device__ inline float bound_thr(uint32_t c)
{
    if (c == 0u) return 0.f;
    if (c >= BOUND_CLASSES - 1u) return u2f(SVR_INF_BITS);
    const uint32_t k = 15u - c;                                // thr = 2^(-k/2)
    const float p = u2f((127u - ((k + 1u) >> 1)) << 23);       // 2^-ceil(k/2)
    return (k & 1u) ? p * 1.41421356237f : p;
}

// b = the largest value of  sigma_t * invSigmaMax  any fetch can produce whose raw footprint lies in [rlo, rhi]; bad: not a number
// somewhere (no bound)
__device__ inline float accept_bound(float rlo, float rhi, const float* __restrict__ tf, int tf_n, float densityScale, float invSigmaMax, bool& bad)
{
    const float nf = (float)tf_n;
    float ilo = (rlo * 1.5259021896696422e-05f) * densityScale;      // the two multiplies of tex_fetch / intensity_at
    float ihi = (rhi * 1.5259021896696422e-05f) * densityScale;
    if (!(ilo == ilo) || !(ihi == ihi)) { bad = true; return 0.f; }
    if (ihi < ilo) { float t = ilo; ilo = ihi; ihi = t; }
    const float xl = fmin_(fmax_(fma_(ilo, nf, -0.5f), -1.f), nf);   // lds_tf_coord
    const float xh = fmin_(fmax_(fma_(ihi, nf, -0.5f), -1.f), nf);
    const int e_lo = (int)__builtin_floorf(xl) + 1;
    const int e_hi = (int)__builtin_floorf(xh) + 2;
    float amax = 0.f;
    bad = false;
    for (int e = e_lo; e <= e_hi; ++e) {                              // padded table: entry e = alpha of texel clamp(e - 1)
        const int t = min(max(e - 1, 0), tf_n - 1);
        const float a = tf[4 * t + 3];
        bad = bad || !(a == a);
        amax = fmax_(amax, a);
    }
    const float b = amax * invSigmaMax;                               // the product of the accept test
    bad = bad || !(b == b);
    return b;
}

__global__ __launch_bounds__(256) void k_bound_class(const uint16_t* __restrict__ mm, int gx, int gy, int gz, int hgx, int hgy, int hgz,
                                                     const float* __restrict__ tf, int tf_n, float densityScale, float invSigmaMax,
                                                     uint32_t* __restrict__ cls, float* __restrict__ thr, uint32_t* __restrict__ census)
{
    const uint32_t hq = blockIdx.x * 256u + threadIdx.x;
    if (hq < BOUND_CLASSES) thr[hq] = bound_thr(hq);
    const uint32_t hn = (uint32_t)hgx * (uint32_t)hgy * (uint32_t)hgz;
    if (hq >= hn) return;
    const int hx = (int)(hq % (uint32_t)hgx), hy = (int)((hq / (uint32_t)hgx) % (uint32_t)hgy), hz = (int)(hq / ((uint32_t)hgx * (uint32_t)hgy));
    uint32_t c_max = 0u;
    for (int d = 0; d < 8; ++d) {
        const int x = 2 * hx + (d & 1), y = 2 * hy + ((d >> 1) & 1), z = 2 * hz + (d >> 2);
        if (x >= gx || y >= gy || z >= gz) continue;
        const size_t m = (size_t)x + (size_t)gx * ((size_t)y + (size_t)gy * (size_t)z);
        bool bad;
        const float b = accept_bound((float)mm[2 * m], (float)mm[2 * m + 1], tf, tf_n, densityScale, invSigmaMax, bad);
        uint32_t c = BOUND_CLASSES - 1u;
        if (!bad)
            for (uint32_t k = 0; k < BOUND_CLASSES; ++k)
                if (b <= bound_thr(k)) { c = k; break; }
        c_max = max(c_max, c);
    }
    atomicOr(&cls[hq >> 3], c_max << ((hq & 7u) << 2));
    // census (wave-aggregated): where would the bound test ever reject a fetch?
    const uint64_t partial = __ballot(c_max >= 1u && c_max < BOUND_CLASSES - 1u), full = __ballot(c_max == BOUND_CLASSES - 1u);
    if ((threadIdx.x & 63u) == 0u) {
        if (partial) atomicAdd(&census[0], (uint32_t)__popcll(partial));
        if (full) atomicAdd(&census[1], (uint32_t)__popcll(full));
    }
}

// FAST BOUND LOOK-UP (bit-exact; svr_lanes.hpp, iterate_rot).  Any valid upper bound of sigma_t * invSigmaMax culls correctly -- a culled draw is
// a rejection whatever the voxels hold -- so the look-up need not find the exact trilinear cell of the tap: the lane machine takes the
// half-resolution macro-cell from ONE fma per axis on the ray parameter (13 vector instructions where cell_of + cell_info take ~50), which may
// differ from the reference's float chain by a rounding error (far below one voxel: the host checks the camera distance that guarantees it), and this
// table bounds every fetch within one more voxel around each cell (mm_wide: launch_minmax with pad 1).  Walks stay inside the clipped box, the
// box inside the texture domain (the host checks that too), so the coordinate lies within [-eps, grid + 1/2 voxel + eps]: the table carries one
// more cell around the grid and the look-up adds 1 instead of clamping.  It also drops the second look-up
// (class -> threshold) and the conversion of the draw: the byte B is compared with the top 8 bits of the draw's RANDOM WORD x,
//     cull  <=>  (x >> 24) > B,     B = min(255, X(b) >> 24),  X(b) = the smallest word with rng_to_uniform(X) >= b  (2^32 if none)
// -- rng_to_uniform is monotone, so (x >> 24) > B implies x >= X(b), i.e. xi >= b.  Linear in the bound with 1/256 resolution (the classes'
// steps of sqrt 2 waste up to 41 % of a cell's fetches; this wastes 0.4 % of the iterations), B = 255: every draw fetches.
__global__ __launch_bounds__(256) void k_bound8(const uint16_t* __restrict__ mmw, int hgx, int hgy, int hgz, const float* __restrict__ tf, int tf_n, float densityScale,
                                                float invSigmaMax, uint8_t* __restrict__ bnd8)
{
    const uint32_t e = blockIdx.x * 256u + threadIdx.x;
    if (e >= BOUND8_BYTES) return;
    const uint32_t px = BOUND8_DIM, py = BOUND8_DIM, pz = BOUND8_DIM;
    if (e >= px * py * pz) { bnd8[e] = 255u; return; }
    // entry (x + 1, y + 1, z + 1) = cell (x, y, z); the entries around the grid repeat its edge cells
    const int x = min(max((int)(e % px) - 1, 0), hgx - 1), y = min(max((int)((e / px) % py) - 1, 0), hgy - 1), z = min(max((int)(e / (px * py)) - 1, 0), hgz - 1);
    const uint32_t hq = (uint32_t)x + (uint32_t)hgx * ((uint32_t)y + (uint32_t)hgy * (uint32_t)z);
    bool bad;
    const float b = accept_bound((float)mmw[2 * (size_t)hq], (float)mmw[2 * (size_t)hq + 1], tf, tf_n, densityScale, invSigmaMax, bad);
    uint32_t B = 255u;
    if (!bad) {
        uint64_t lo = 0ull, hi = 1ull << 32;                        // X(b) by bisection on the monotone conversion
        while (lo < hi) {
            const uint64_t mid = (lo + hi) >> 1;
            if (rng_to_uniform((uint32_t)mid) >= b) hi = mid; else lo = mid + 1ull;
        }
        B = min(255u, (uint32_t)(lo >> 24));
    }
    bnd8[e] = (uint8_t)B;
}

hipError_t launch_bound8(const uint16_t* mm_wide, int hgx, int hgy, int hgz, const float* tf_rgba, int tf_n, float densityScale,
                         float invSigmaMax, uint8_t* bnd8, hipStream_t st)
{
    if (hgx + 2 > (int)BOUND8_DIM || hgy + 2 > (int)BOUND8_DIM || hgz + 2 > (int)BOUND8_DIM) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_bound8, dim3(BOUND8_BYTES / 256u), dim3(256), 0, st, mm_wide, hgx, hgy, hgz, tf_rgba, tf_n, densityScale, invSigmaMax, bnd8);
    return hipGetLastError();
}

hipError_t launch_fine_mask(const uint16_t* mm, uint32_t n_cells, const uint32_t* tf_zero_prefix, int tf_n, float densityScale,
                            uint32_t* mask, uint32_t words, hipStream_t st)
{
    hipError_t e = hipMemsetAsync(mask, 0, (size_t)words * 4u, st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_empty_mask, dim3((n_cells + 255u) / 256u), dim3(256), 0, st, mm, n_cells, tf_zero_prefix, tf_n, densityScale, mask, (uint32_t*)nullptr);
    return hipGetLastError();
}

// Occupancy of the 2 x 2 x 2 fine cells (half the edge) inside every macro-cell, one byte per macro-cell: bit dx + 2 dy + 4 dz is
// set unless that fine cell is `empty` (fine level of k_empty_mask).  The local-majorant walk (svr_trace_lm.hip) spends its free
// path only in the occupied eighths of a macro-cell.  A child beyond the fine grid's end repeats the last fine cell (the grids
// clamp the same way).
__global__ __launch_bounds__(256) void k_sub8(const uint32_t* __restrict__ fine_empty, int fgx, int fgy, int fgz, int gx, int gy, int gz, uint8_t* __restrict__ sub8)
{
    const uint32_t m = blockIdx.x * 256u + threadIdx.x;
    if (m >= (uint32_t)gx * (uint32_t)gy * (uint32_t)gz) return;
    const int mx = (int)(m % (uint32_t)gx), my = (int)((m / (uint32_t)gx) % (uint32_t)gy), mz = (int)(m / ((uint32_t)gx * (uint32_t)gy));
    uint32_t bits = 0u;
    for (int d = 0; d < 8; ++d) {
        const int fx = min(2 * mx + (d & 1), fgx - 1), fy = min(2 * my + ((d >> 1) & 1), fgy - 1), fz = min(2 * mz + (d >> 2), fgz - 1);
        const uint32_t f = (uint32_t)fx + (uint32_t)fgx * ((uint32_t)fy + (uint32_t)fgy * (uint32_t)fz);
        if (!((fine_empty[f >> 5] >> (f & 31u)) & 1u)) bits |= 1u << d;
    }
    sub8[m] = (uint8_t)bits;
}

hipError_t launch_sub8(const uint32_t* fine_empty, int fgx, int fgy, int fgz, int gx, int gy, int gz, uint8_t* sub8, hipStream_t st)
{
    const uint32_t n = (uint32_t)gx * (uint32_t)gy * (uint32_t)gz;
    hipLaunchKernelGGL(k_sub8, dim3((n + 255u) / 256u), dim3(256), 0, st, fine_empty, fgx, fgy, fgz, gx, gy, gz, sub8);
    return hipGetLastError();
}

hipError_t launch_bound_class(const uint16_t* mm, int gx, int gy, int gz, const float* tf_rgba, int tf_n, float densityScale,
                              float invSigmaMax, uint32_t* accel, hipStream_t st)
{
    const int hgx = (gx + 1) / 2, hgy = (gy + 1) / 2, hgz = (gz + 1) / 2;
    const uint32_t hn = (uint32_t)hgx * (uint32_t)hgy * (uint32_t)hgz;
    hipError_t e = hipMemsetAsync(accel + ACCEL_CLASS_OFF, 0, (size_t)(DIST_WORDS_MAX + BOUND_CLASSES + 2u) * 4u, st);   // (census words 2.. belong to launch_empty_mask)
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_bound_class, dim3((hn + 255u) / 256u), dim3(256), 0, st, mm, gx, gy, gz, hgx, hgy, hgz, tf_rgba, tf_n, densityScale,
                       invSigmaMax, accel + ACCEL_CLASS_OFF, reinterpret_cast<float*>(accel + ACCEL_THR_OFF), accel + ACCEL_CENSUS_OFF);
    return hipGetLastError();
}

hipError_t launch_minmax(const uint16_t* src, uint16_t* mm, int nx, int ny, int nz, int shift,
                         int gx, int gy, int gz, hipStream_t st, int pad)
{
    uint32_t n = (uint32_t)gx * (uint32_t)gy * (uint32_t)gz;
    hipLaunchKernelGGL(k_minmax, dim3(n), dim3(64), 0, st, src, mm, nx, ny, nz, shift, gx, gy, gz, pad);
    return hipGetLastError();
}

// Distance field over the macro grid: D(c) = Chebyshev distance (in macro-cells, capped at 15) from cell c to the
// nearest macro-cell that is NOT empty; cells outside the grid count as empty (no fetch ever happens there).
// D(c) >= 1 <=> c is empty; D(c) >= 2 <=> c and its 26 neighbours are empty ("deep-empty"); in general every cell
// within Chebyshev distance D(c) - 1 of c is empty, so a ray at a point of c may advance until its largest-axis
// displacement reaches D(c) - 1 cells without meeting a fetch that could return a non-zero opacity.  A march
// through the grid in float arithmetic is off by far less than one macro-cell.
// Built separably (x, then y, then z: min over the offset of max(|offset|, previous)), then stored at HALF
// resolution (minimum over the 2x2x2 children, 4 bits per coarse cell: <= 16 KB for a 64^3 grid) -- that is what
// the kernels keep in LDS next to the `empty` bitmask.
__global__ __launch_bounds__(256) void k_dist_axis(const uint32_t* __restrict__ empty, const uint8_t* __restrict__ in,
                                                   uint8_t* __restrict__ out, int gx, int gy, int gz, int axis)
{
    uint32_t m = blockIdx.x * 256u + threadIdx.x;
    uint32_t n = (uint32_t)gx * (uint32_t)gy * (uint32_t)gz;
    if (m >= n) return;
    int c[3] = {(int)(m % (uint32_t)gx), (int)((m / (uint32_t)gx) % (uint32_t)gy), (int)(m / ((uint32_t)gx * (uint32_t)gy))};
    const int g[3] = {gx, gy, gz};
    const int stride = axis == 0 ? 1 : (axis == 1 ? gx : gx * gy);
    int best = DIST_CAP;
    for (int o = -(DIST_CAP - 1); o <= DIST_CAP - 1; ++o) {
        int p = c[axis] + o;
        if (p < 0 || p >= g[axis]) continue;                       // outside the grid: empty, infinitely far
        uint32_t q = (uint32_t)((int)m + o * stride);
        int d = axis == 0 ? (((empty[q >> 5] >> (q & 31u)) & 1u) ? DIST_CAP : 0) : (int)in[q];
        int a = o < 0 ? -o : o;
        d = d > a ? d : a;
        best = d < best ? d : best;
    }
    out[m] = (uint8_t)best;
}

__global__ __launch_bounds__(256) void k_dist_pack(const uint8_t* __restrict__ in, uint32_t* __restrict__ out, int gx, int gy, int gz,
                                                   int hgx, int hgy, int hgz, uint32_t words)
{
    uint32_t wi = blockIdx.x * 256u + threadIdx.x;
    if (wi >= words) return;
    const uint32_t hn = (uint32_t)hgx * (uint32_t)hgy * (uint32_t)hgz;
    uint32_t word = 0u;
    for (uint32_t k = 0; k < 8u; ++k) {
        uint32_t hq = wi * 8u + k;
        if (hq >= hn) break;
        int hx = (int)(hq % (uint32_t)hgx), hy = (int)((hq / (uint32_t)hgx) % (uint32_t)hgy), hz = (int)(hq / ((uint32_t)hgx * (uint32_t)hgy));
        int best = DIST_CAP;
        for (int dz = 0; dz < 2; ++dz)
            for (int dy = 0; dy < 2; ++dy)
                for (int dx = 0; dx < 2; ++dx) {
                    int x = 2 * hx + dx, y = 2 * hy + dy, z = 2 * hz + dz;
                    if (x >= gx || y >= gy || z >= gz) continue;
                    int d = in[(uint32_t)x + (uint32_t)gx * ((uint32_t)y + (uint32_t)gy * (uint32_t)z)];
                    best = d < best ? d : best;
                }
        word |= (uint32_t)best << (4u * k);
    }
    out[wi] = word;
}

// deep-empty bit of a macro-cell: distance >= 2, i.e. the cell and its 26 in-grid neighbours are empty (full resolution)
__global__ __launch_bounds__(256) void k_deep_mask(const uint8_t* __restrict__ dist, uint32_t* __restrict__ deep, uint32_t n)
{
    uint32_t m = blockIdx.x * 256u + threadIdx.x;
    if (m < n && dist[m] >= 2) atomicOr(&deep[m >> 5], 1u << (m & 31u));
}

// mask: DIST_WORDS_MAX words of packed half-resolution distances, MASK_WORDS_MAX words of deep-empty bits,
// then mask_words words of `empty` bits; tmp: 2 x n_cells bytes
hipError_t launch_empty_mask(const uint16_t* mm, int gx, int gy, int gz, const uint32_t* tf_zero_prefix, int tf_n,
                             float densityScale, uint32_t* mask, uint32_t mask_words, uint8_t* tmp, hipStream_t st)
{
    uint32_t n_cells = (uint32_t)gx * (uint32_t)gy * (uint32_t)gz;
    uint32_t* deep = mask + DIST_WORDS_MAX;
    uint32_t* empty = mask + DIST_WORDS_MAX + MASK_WORDS_MAX;
    hipError_t e = hipMemsetAsync(deep, 0, (size_t)(MASK_WORDS_MAX + mask_words) * 4u, st);
    if (e != hipSuccess) return e;
    const uint32_t blocks = (n_cells + 255u) / 256u;
    e = hipMemsetAsync(mask + ACCEL_CENSUS_OFF + 2, 0, 8, st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_empty_mask, dim3(blocks), dim3(256), 0, st, mm, n_cells, tf_zero_prefix, tf_n, densityScale, empty, mask + ACCEL_CENSUS_OFF + 2);
    uint8_t* a = tmp;
    uint8_t* b = tmp + n_cells;
    hipLaunchKernelGGL(k_dist_axis, dim3(blocks), dim3(256), 0, st, empty, (const uint8_t*)nullptr, a, gx, gy, gz, 0);
    hipLaunchKernelGGL(k_dist_axis, dim3(blocks), dim3(256), 0, st, empty, a, b, gx, gy, gz, 1);
    hipLaunchKernelGGL(k_dist_axis, dim3(blocks), dim3(256), 0, st, empty, b, a, gx, gy, gz, 2);
    const int hgx = (gx + 1) / 2, hgy = (gy + 1) / 2, hgz = (gz + 1) / 2;
    const uint32_t words = ((uint32_t)hgx * (uint32_t)hgy * (uint32_t)hgz + 7u) / 8u;
    hipLaunchKernelGGL(k_deep_mask, dim3(blocks), dim3(256), 0, st, a, deep, n_cells);
    hipLaunchKernelGGL(k_dist_pack, dim3((words + 255u) / 256u), dim3(256), 0, st, a, mask, gx, gy, gz, hgx, hgy, hgz, words);
    return hipGetLastError();
}

} // namespace svr
