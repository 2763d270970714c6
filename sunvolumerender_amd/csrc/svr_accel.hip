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

__global__ __launch_bounds__(64) void k_minmax(const uint16_t* __restrict__ src, uint16_t* __restrict__ mm,
                                               int nx, int ny, int nz, int shift, int gx, int gy, int gz)
{
    const int S = 1 << shift;
    uint32_t m = blockIdx.x;
    int mx = (int)(m % (uint32_t)gx);
    int my = (int)((m / (uint32_t)gx) % (uint32_t)gy);
    int mz = (int)(m / ((uint32_t)gx * (uint32_t)gy));
    int x0 = mx * S - 1, y0 = my * S - 1, z0 = mz * S - 1;
    // voxels per axis in the footprint (+1 on the last macro-cell: border voxel N)
    int Ex = S + 1 + (mx == gx - 1), Ey = S + 1 + (my == gy - 1), Ez = S + 1 + (mz == gz - 1);
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
                                                    float densityScale, uint32_t* __restrict__ mask)
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
    if (zeros == (uint32_t)(e_hi - e_lo + 1)) atomicOr(&mask[m >> 5], 1u << (m & 31u));
}

hipError_t launch_minmax(const uint16_t* src, uint16_t* mm, int nx, int ny, int nz, int shift,
                         int gx, int gy, int gz, hipStream_t st)
{
    uint32_t n = (uint32_t)gx * (uint32_t)gy * (uint32_t)gz;
    hipLaunchKernelGGL(k_minmax, dim3(n), dim3(64), 0, st, src, mm, nx, ny, nz, shift, gx, gy, gz);
    return hipGetLastError();
}

// deep-empty: the macro-cell and all its in-grid neighbours (3x3x3) are empty.  A ray marched through
// the macro grid in float arithmetic can be off by far less than one macro-cell, so "every visited
// cell is deep-empty" proves that every fetch along the ray lies in an empty macro-cell.
__global__ __launch_bounds__(256) void k_deep_mask(const uint32_t* __restrict__ empty, uint32_t* __restrict__ deep,
                                                   int gx, int gy, int gz)
{
    uint32_t m = blockIdx.x * 256u + threadIdx.x;
    uint32_t n = (uint32_t)gx * (uint32_t)gy * (uint32_t)gz;
    if (m >= n) return;
    int mx = (int)(m % (uint32_t)gx), my = (int)((m / (uint32_t)gx) % (uint32_t)gy), mz = (int)(m / ((uint32_t)gx * (uint32_t)gy));
    bool all = true;
    for (int dz = -1; dz <= 1; ++dz)
        for (int dy = -1; dy <= 1; ++dy)
            for (int dx = -1; dx <= 1; ++dx) {
                int x = mx + dx, y = my + dy, z = mz + dz;
                if (x < 0 || y < 0 || z < 0 || x >= gx || y >= gy || z >= gz) continue;
                uint32_t q = (uint32_t)x + (uint32_t)gx * ((uint32_t)y + (uint32_t)gy * (uint32_t)z);
                all = all && ((empty[q >> 5] >> (q & 31u)) & 1u);
            }
    if (all) atomicOr(&deep[m >> 5], 1u << (m & 31u));
}

hipError_t launch_empty_mask(const uint16_t* mm, int gx, int gy, int gz, const uint32_t* tf_zero_prefix, int tf_n,
                             float densityScale, uint32_t* mask, uint32_t mask_words, hipStream_t st)
{
    uint32_t n_cells = (uint32_t)gx * (uint32_t)gy * (uint32_t)gz;
    hipError_t e = hipMemsetAsync(mask, 0, (size_t)mask_words * 8u, st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_empty_mask, dim3((n_cells + 255u) / 256u), dim3(256), 0, st, mm, n_cells,
                       tf_zero_prefix, tf_n, densityScale, mask + mask_words);
    hipLaunchKernelGGL(k_deep_mask, dim3((n_cells + 255u) / 256u), dim3(256), 0, st, mask + mask_words, mask, gx, gy, gz);
    return hipGetLastError();
}

} // namespace svr
