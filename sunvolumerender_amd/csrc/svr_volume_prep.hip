// svr_volume_prep.hip -- VolumeReader::Read after the file is in memory (core/VolumeReader.cpp:41-76) and
// CreateDeviceVolume (174-185), on the GPU.  The reference runs four single-threaded VTK filters over the
// volume (cast, scalar range, accumulate, gradient magnitude) plus its own Rescale loop; all of it is
// streaming work over the voxels, i.e. HBM-bound here:
//
//   pass A  k_cast_range_grad<T>   elems -> short (vtkImageCast), min/max (GetScalarRange), and the largest
//                                  squared central-difference gradient (vtkImageGradientMagnitude, 3-D,
//                                  HandleBoundaries on).  Reads the elements once; neighbours come from cache.
//   pass B  k_rescale_hist         short -> u16 (VolumeReader::Rescale) + histogram (vtkImageAccumulate with
//                                  IgnoreZero, bins = max - min); needs the range of pass A.
//
// Algorithmic bytes per voxel: sizeof(T) + 2 (short copy; none for MET_SHORT) + 2 + 2.
//
// Gradient magnitude: VTK stores static_cast<short>(sqrt(sum)) per voxel and the reference takes the range
// maximum.  trunc(sqrt(.)) is monotone, so the maximum of the shorts is trunc(sqrt(max sum)) -- one sqrt on
// the host -- unless some magnitude reaches 32768, where the narrowing wraps (x86: cvttsd2si, low 16 bits);
// only then the exact per-voxel kernel k_gradmag_exact runs.
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "svr_internal.hpp"
#include "svr_io.h"

// svr_host_io.hip
extern "C" int svr_internal_mhd_load(const char* path, svr_mhd_header* h, std::vector<uint8_t>* elems);

namespace {

using svr::failf;

#define PREP_TRY(expr)                                                                                   \
    do {                                                                                                 \
        hipError_t _e = (expr);                                                                          \
        if (_e != hipSuccess) { cleanup(); return failf((int)_e, "HIP error %s at %s:%d", hipGetErrorName(_e), __FILE__, __LINE__); } \
    } while (0)

struct PrepStats {                  // device-side reduction targets
    int vmin, vmax;                 // of the short image
    unsigned long long max_sum;     // bit pattern of the largest squared gradient (non-negative doubles order like integers)
    int max_mag;                    // k_gradmag_exact: largest wrapped short magnitude
};

__device__ __forceinline__ short wrap16(long long v) { return (short)(unsigned short)((unsigned long long)v & 0xffffull); }

// float/double -> integer as x86's cvttsd2si does it (truncate; NaN and out of range give INT64_MIN)
__device__ __forceinline__ long long trunc_i64(double d)
{
    if (!(d == d) || d >= 9223372036854775808.0 || d < -9223372036854775808.0) return (long long)0x8000000000000000ull;
    return (long long)d;
}

template <typename T> __device__ __forceinline__ short cast_short(T v) { return wrap16((long long)v); }
template <> __device__ __forceinline__ short cast_short<float>(float v) { return wrap16(trunc_i64((double)v)); }
template <> __device__ __forceinline__ short cast_short<double>(double v) { return wrap16(trunc_i64(v)); }

__device__ __forceinline__ int wave_min(int v) { for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o, 64)); return v; }
__device__ __forceinline__ int wave_max(int v) { for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64)); return v; }
__device__ __forceinline__ unsigned long long wave_max(unsigned long long v)
{
    for (int o = 32; o > 0; o >>= 1) { unsigned long long u = __shfl_xor(v, o, 64); v = u > v ? u : v; }
    return v;
}

// squared gradient of voxel (x, row) as vtkImageGradientMagnitude computes it: (v[-1] - v[+1]) * 0.5/spacing per
// axis, the voxel itself standing in for a neighbour outside the volume, summed x, y, z in double
template <typename T>
__device__ __forceinline__ double grad_sum(const T* __restrict__ src, size_t i, int x, int y, int z, int nx, int ny, int nz,
                                           double r0, double r1, double r2)
{
    const size_t sy = (size_t)nx, sz = (size_t)nx * ny;
    double d, sum = 0.0;
    d = (double)cast_short(src[x > 0 ? i - 1 : i]) - (double)cast_short(src[x < nx - 1 ? i + 1 : i]); d *= r0; sum += d * d;
    d = (double)cast_short(src[y > 0 ? i - sy : i]) - (double)cast_short(src[y < ny - 1 ? i + sy : i]); d *= r1; sum += d * d;
    d = (double)cast_short(src[z > 0 ? i - sz : i]) - (double)cast_short(src[z < nz - 1 ? i + sz : i]); d *= r2; sum += d * d;
    return sum;
}

// Block-level reduction, then at most three atomics per block, and only when the block improves on what is
// already published: one address takes ~88 returning atomics per microsecond chip-wide, so a per-wave
// atomicMin/Max (32 k waves) alone would cost a millisecond.
__device__ __forceinline__ void publish(PrepStats* stats, int lo, int hi, double best)
{
    __shared__ int s_lo[4], s_hi[4];
    __shared__ unsigned long long s_best[4];
    lo = wave_min(lo); hi = wave_max(hi);
    unsigned long long b = wave_max((unsigned long long)__double_as_longlong(best));
    const unsigned w = threadIdx.x >> 6;
    if ((threadIdx.x & 63u) == 0u) { s_lo[w] = lo; s_hi[w] = hi; s_best[w] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (unsigned k = 1; k < 4; ++k) { lo = min(lo, s_lo[k]); hi = max(hi, s_hi[k]); b = s_best[k] > b ? s_best[k] : b; }
        if (lo < __atomic_load_n(&stats->vmin, __ATOMIC_RELAXED)) atomicMin(&stats->vmin, lo);
        if (hi > __atomic_load_n(&stats->vmax, __ATOMIC_RELAXED)) atomicMax(&stats->vmax, hi);
        if (b > __atomic_load_n(&stats->max_sum, __ATOMIC_RELAXED)) atomicMax(&stats->max_sum, b);
    }
}

template <typename T, bool WRITE_SHORT>
__global__ __launch_bounds__(256) void k_cast_range_grad(const T* __restrict__ src, short* __restrict__ dst, int nx, int ny, int nz,
                                                         double r0, double r1, double r2, PrepStats* stats)
{
    int lo = 32767, hi = -32768;
    double best = 0.0;
    const int rows = ny * nz;
    for (int row = blockIdx.x; row < rows; row += gridDim.x) {
        const int z = row / ny, y = row - z * ny;
        const size_t base = (size_t)row * nx;
        for (int x = threadIdx.x; x < nx; x += 256) {
            const size_t i = base + x;
            const short v = cast_short(src[i]);
            if (WRITE_SHORT) dst[i] = v;
            lo = min(lo, (int)v); hi = max(hi, (int)v);
            const double s = grad_sum(src, i, x, y, z, nx, ny, nz, r0, r1, r2);
            best = s > best ? s : best;            // NaN-free: sums of squares of finite numbers
        }
    }
    publish(stats, lo, hi, best);
}

// the rare exact path: per-voxel sqrt and narrowing, maximum of the wrapped shorts
__global__ __launch_bounds__(256) void k_gradmag_exact(const short* __restrict__ v, int nx, int ny, int nz,
                                                       double r0, double r1, double r2, PrepStats* stats)
{
    int best = -32768;
    const int rows = ny * nz;
    for (int row = blockIdx.x; row < rows; row += gridDim.x) {
        const int z = row / ny, y = row - z * ny;
        for (int x = threadIdx.x; x < nx; x += 256) {
            const double s = grad_sum(v, (size_t)row * nx + x, x, y, z, nx, ny, nz, r0, r1, r2);
            best = max(best, (int)wrap16(trunc_i64(__builtin_sqrt(s))));
        }
    }
    best = wave_max(best);
    if ((threadIdx.x & 63) == 0 && best > __atomic_load_n(&stats->max_mag, __ATOMIC_RELAXED)) atomicMax(&stats->max_mag, best);
}

constexpr int HIST_LDS_BINS = 16384;           // at most 64 KB of LDS counters per block

// VolumeReader::Rescale<short, unsigned short> + vtkImageAccumulate(origin = min, spacing 1, bins, IgnoreZero)
template <bool LDS_HIST>
__global__ __launch_bounds__(256) void k_rescale_hist(const short* __restrict__ v, unsigned short* __restrict__ out, size_t n,
                                                      float dataMin, float dataMax, unsigned int* __restrict__ hist, int bins, int vmin)
{
    extern __shared__ unsigned int lh[];             // `bins` counters when LDS_HIST (dynamic: small ranges leave room for more blocks)
    if (LDS_HIST) {
        for (int b = threadIdx.x; b < bins; b += 256) lh[b] = 0u;
        __syncthreads();
    }
    const float extent = dataMax - dataMin;
    const float dataTypeExtent = 65535.f;
    for (size_t i = (size_t)blockIdx.x * 256u + threadIdx.x; i < n; i += (size_t)gridDim.x * 256u) {
        const short s = v[i];
        const float r = ((float)s - dataMin) / extent * dataTypeExtent;
        out[i] = (r == r) ? (unsigned short)wrap16(trunc_i64((double)r)) : (unsigned short)0;     // 0/0 (constant volume) -> 0
        if (hist && s != 0) {
            const int b = (int)s - vmin;
            if (b >= 0 && b < bins) {
                if (LDS_HIST) atomicAdd(&lh[b], 1u);
                else atomicAdd(&hist[b], 1u);
            }
        }
    }
    if (LDS_HIST && hist) {
        __syncthreads();
        for (int b = threadIdx.x; b < bins; b += 256) {
            const unsigned int c = lh[b];
            if (c) atomicAdd(&hist[b], c);
        }
    }
}

// ---- 8 voxels per thread (nx % 8 == 0, 16-byte aligned buffers): one wide load per neighbour row instead of
// eight 2-byte ones.  Same arithmetic as the scalar kernels above, which remain the fallback.
template <typename T> struct alignas((8 * sizeof(T)) > 16 ? 16 : (8 * sizeof(T))) Pack8 { T v[8]; };

template <typename T>
__device__ __forceinline__ Pack8<T> load8(const T* p) { return *reinterpret_cast<const Pack8<T>*>(p); }

template <typename T, bool WRITE_SHORT>
__global__ __launch_bounds__(256) void k_cast_range_grad_v8(const T* __restrict__ src, short* __restrict__ dst, int nx, int ny, int nz,
                                                            double r0, double r1, double r2, PrepStats* stats)
{
    int lo = 32767, hi = -32768;
    double best = 0.0;
    const unsigned ppr = (unsigned)nx >> 3;                              // packs per row
    const unsigned long long n_packs = (unsigned long long)ppr * (unsigned)ny * (unsigned)nz;
    const size_t sy = (size_t)nx, sz = (size_t)nx * ny;
    for (unsigned long long p = (unsigned long long)blockIdx.x * 256u + threadIdx.x; p < n_packs; p += (unsigned long long)gridDim.x * 256u) {
        const unsigned row = (unsigned)(p / ppr), px = (unsigned)(p - (unsigned long long)row * ppr);
        const int z = (int)(row / (unsigned)ny), y = (int)(row - (unsigned)z * (unsigned)ny), x0 = (int)(px << 3);
        const size_t i0 = (size_t)row * nx + x0;
        const Pack8<T> c = load8(src + i0);
        const Pack8<T> ym = load8(src + (y > 0 ? i0 - sy : i0)), yp = load8(src + (y < ny - 1 ? i0 + sy : i0));
        const Pack8<T> zm = load8(src + (z > 0 ? i0 - sz : i0)), zp = load8(src + (z < nz - 1 ? i0 + sz : i0));
        int s[10];
#pragma unroll
        for (int k = 0; k < 8; ++k) s[k + 1] = cast_short(c.v[k]);
        s[0] = x0 > 0 ? (int)cast_short(src[i0 - 1]) : s[1];
        s[9] = x0 + 8 < nx ? (int)cast_short(src[i0 + 8]) : s[8];
        if (WRITE_SHORT) {
            Pack8<short> o;
#pragma unroll
            for (int k = 0; k < 8; ++k) o.v[k] = (short)s[k + 1];
            *reinterpret_cast<Pack8<short>*>(dst + i0) = o;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            lo = min(lo, s[k + 1]); hi = max(hi, s[k + 1]);
            // differences of shorts are exact in int and in double: (double)a - (double)b == (double)(a - b)
            double d, sum = 0.0;
            d = (double)(s[k] - s[k + 2]); d *= r0; sum += d * d;
            d = (double)((int)cast_short(ym.v[k]) - (int)cast_short(yp.v[k])); d *= r1; sum += d * d;
            d = (double)((int)cast_short(zm.v[k]) - (int)cast_short(zp.v[k])); d *= r2; sum += d * d;
            best = sum > best ? sum : best;
        }
    }
    publish(stats, lo, hi, best);
}

template <bool LDS_HIST>
__global__ __launch_bounds__(256) void k_rescale_hist_v8(const short* __restrict__ v, unsigned short* __restrict__ out, size_t n_packs,
                                                         float dataMin, float dataMax, unsigned int* __restrict__ hist, int bins, int vmin)
{
    extern __shared__ unsigned int lh[];             // `bins` counters when LDS_HIST (dynamic: small ranges leave room for more blocks)
    if (LDS_HIST) {
        for (int b = threadIdx.x; b < bins; b += 256) lh[b] = 0u;
        __syncthreads();
    }
    const float extent = dataMax - dataMin;
    const float dataTypeExtent = 65535.f;
    const unsigned lane = threadIdx.x & 63u;
    for (size_t p = (size_t)blockIdx.x * 256u + threadIdx.x; p < n_packs; p += (size_t)gridDim.x * 256u) {
        const Pack8<short> in = load8(v + (p << 3));
        Pack8<unsigned short> o;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const short s = in.v[k];
            const float r = ((float)s - dataMin) / extent * dataTypeExtent;
            o.v[k] = (r == r) ? (unsigned short)wrap16(trunc_i64((double)r)) : (unsigned short)0;
        }
        *reinterpret_cast<Pack8<unsigned short>*>(out + (p << 3)) = o;
        if (hist) {
            // air: a whole wave of 8-voxel packs lands in one bin -> one add of 512 instead of 512 conflicting ones
            const int first = in.v[0];
            bool same = true;
#pragma unroll
            for (int k = 1; k < 8; ++k) same = same && in.v[k] == first;
            const int b_first = first - vmin;
            const bool counts_first = first != 0 && b_first >= 0 && b_first < bins;
            const unsigned long long act = __ballot(true);
            const int b0 = __builtin_amdgcn_readfirstlane(b_first);
            if (__ballot(same && counts_first && b_first == b0) == act) {
                if (lane == (unsigned)__builtin_ctzll(act)) {
                    const unsigned add = 8u * (unsigned)__builtin_popcountll(act);
                    if (LDS_HIST) atomicAdd(&lh[b0], add);
                    else atomicAdd(&hist[b0], add);
                }
            } else {
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int s = in.v[k];
                    const int b = s - vmin;
                    if (s != 0 && b >= 0 && b < bins) {
                        if (LDS_HIST) atomicAdd(&lh[b], 1u);
                        else atomicAdd(&hist[b], 1u);
                    }
                }
            }
        }
    }
    if (LDS_HIST && hist) {
        __syncthreads();
        for (int b = threadIdx.x; b < bins; b += 256) {
            const unsigned int c = lh[b];
            if (c) atomicAdd(&hist[b], c);
        }
    }
}

template <typename T>
void launch_pass_a(const void* src, short* dst, bool write_short, int nx, int ny, int nz, const double r[3], PrepStats* st, bool vec8, hipStream_t s)
{
    const int rows = ny * nz;
    if (vec8) {
        const unsigned long long packs = (unsigned long long)rows * (unsigned)(nx >> 3);
        const unsigned long long want = (packs + 255) / 256;
        const int blocks = (int)(want < 256ull * 8 ? want : 256ull * 8);
        if (write_short) hipLaunchKernelGGL((k_cast_range_grad_v8<T, true>), dim3(blocks), dim3(256), 0, s, (const T*)src, dst, nx, ny, nz, r[0], r[1], r[2], st);
        else hipLaunchKernelGGL((k_cast_range_grad_v8<T, false>), dim3(blocks), dim3(256), 0, s, (const T*)src, dst, nx, ny, nz, r[0], r[1], r[2], st);
        return;
    }
    const int blocks = rows < 256 * 8 ? rows : 256 * 8;
    if (write_short) hipLaunchKernelGGL((k_cast_range_grad<T, true>), dim3(blocks), dim3(256), 0, s, (const T*)src, dst, nx, ny, nz, r[0], r[1], r[2], st);
    else hipLaunchKernelGGL((k_cast_range_grad<T, false>), dim3(blocks), dim3(256), 0, s, (const T*)src, dst, nx, ny, nz, r[0], r[1], r[2], st);
}

const int kElemSize[8] = {1, 1, 2, 2, 4, 4, 4, 8};

float g_last_ms = 0.f;
uint64_t g_last_bytes = 0;

} // namespace

extern "C" {

int svr_volume_preprocess(const void* elems, int elem_type, int nx, int ny, int nz, const double spacing[3],
                          int elems_on_device, uint16_t* out_u16_device, uint32_t* hist, uint32_t hist_capacity,
                          svr_volume_info* info)
{
    if (svr::ensure_ready()) return svr_last_error_code();
    if (!elems || !out_u16_device || !info || !spacing) return failf(-4, "svr_volume_preprocess: null argument");
    if (elem_type < 0 || elem_type > SVR_ELEM_F64) return failf(-6, "svr_volume_preprocess: unknown element type %d", elem_type);
    if (nx <= 0 || ny <= 0 || nz <= 0 || (int64_t)ny * nz > 0x7fffffff) return failf(-6, "svr_volume_preprocess: bad dimensions %d x %d x %d", nx, ny, nz);
    for (int a = 0; a < 3; ++a) if (!(spacing[a] > 0.0)) return failf(-6, "svr_volume_preprocess: spacing must be positive");
    const size_t n = (size_t)nx * ny * nz;
    const size_t esz = (size_t)kElemSize[elem_type];
    hipStream_t st = svr::current_stream();

    void* d_elems = nullptr; short* d_short = nullptr; PrepStats* d_stats = nullptr; unsigned int* d_hist = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    auto cleanup = [&]() {
        if (d_elems && !elems_on_device) hipFree(d_elems);
        if (d_short && elem_type != SVR_ELEM_I16) hipFree(d_short);
        if (d_stats) hipFree(d_stats);
        if (d_hist) hipFree(d_hist);
        if (ev0) hipEventDestroy(ev0);
        if (ev1) hipEventDestroy(ev1);
    };
    if (elems_on_device) d_elems = const_cast<void*>(elems);
    else {
        PREP_TRY(hipMalloc(&d_elems, n * esz));
        PREP_TRY(hipMemcpyAsync(d_elems, elems, n * esz, hipMemcpyHostToDevice, st));
    }
    const bool write_short = elem_type != SVR_ELEM_I16;
    if (write_short) PREP_TRY(hipMalloc((void**)&d_short, n * sizeof(short)));
    else d_short = (short*)d_elems;
    PREP_TRY(hipMalloc((void**)&d_stats, sizeof(PrepStats)));
    PrepStats init = {32767, -32768, 0ull, -32768};
    PREP_TRY(hipMemcpyAsync(d_stats, &init, sizeof init, hipMemcpyHostToDevice, st));
    PREP_TRY(hipEventCreate(&ev0));
    PREP_TRY(hipEventCreate(&ev1));

    const double r[3] = {0.5 / spacing[0], 0.5 / spacing[1], 0.5 / spacing[2]};
    const int rows = ny * nz;
    const int blocks_a = rows < 256 * 8 ? rows : 256 * 8;
    const bool vec8 = (nx % 8 == 0) && ((uintptr_t)d_elems % 16 == 0) && ((uintptr_t)d_short % 16 == 0) && ((uintptr_t)out_u16_device % 16 == 0);
    PREP_TRY(hipEventRecord(ev0, st));
    switch (elem_type) {
    case SVR_ELEM_I8:  launch_pass_a<int8_t>(d_elems, d_short, write_short, nx, ny, nz, r, d_stats, vec8, st); break;
    case SVR_ELEM_U8:  launch_pass_a<uint8_t>(d_elems, d_short, write_short, nx, ny, nz, r, d_stats, vec8, st); break;
    case SVR_ELEM_I16: launch_pass_a<int16_t>(d_elems, d_short, write_short, nx, ny, nz, r, d_stats, vec8, st); break;
    case SVR_ELEM_U16: launch_pass_a<uint16_t>(d_elems, d_short, write_short, nx, ny, nz, r, d_stats, vec8, st); break;
    case SVR_ELEM_I32: launch_pass_a<int32_t>(d_elems, d_short, write_short, nx, ny, nz, r, d_stats, vec8, st); break;
    case SVR_ELEM_U32: launch_pass_a<uint32_t>(d_elems, d_short, write_short, nx, ny, nz, r, d_stats, vec8, st); break;
    case SVR_ELEM_F32: launch_pass_a<float>(d_elems, d_short, write_short, nx, ny, nz, r, d_stats, vec8, st); break;
    default:           launch_pass_a<double>(d_elems, d_short, write_short, nx, ny, nz, r, d_stats, vec8, st); break;
    }
    PREP_TRY(hipGetLastError());
    PrepStats hs;
    PREP_TRY(hipMemcpyAsync(&hs, d_stats, sizeof hs, hipMemcpyDeviceToHost, st));
    PREP_TRY(hipStreamSynchronize(st));

    // VolumeReader.cpp:54-59: the range goes through double (GetScalarRange) and float (Rescale's parameters)
    const double range0 = hs.vmin, range1 = hs.vmax;
    const int bins_i = (int)(range1 - range0 - 1.0) + 1;
    const int bins = bins_i < 0 ? 0 : bins_i;
    const bool want_hist = hist != nullptr && hist_capacity > 0 && bins > 0;
    if (want_hist) {
        PREP_TRY(hipMalloc((void**)&d_hist, sizeof(unsigned int) * (size_t)bins));
        PREP_TRY(hipMemsetAsync(d_hist, 0, sizeof(unsigned int) * (size_t)bins, st));
    }
    {
        const size_t items = vec8 ? n / 8 : n;
        const size_t want_blocks = (items + 255) / 256;
        const int blocks_b = (int)(want_blocks < 256 * 8 ? want_blocks : 256 * 8);
        unsigned int* hp = want_hist ? d_hist : nullptr;
        const float fmin = (float)range0, fmax = (float)range1;
        if (vec8) {
            if (bins <= HIST_LDS_BINS) hipLaunchKernelGGL((k_rescale_hist_v8<true>), dim3(blocks_b), dim3(256), sizeof(unsigned int) * (size_t)bins, st, d_short, out_u16_device, items, fmin, fmax, hp, bins, hs.vmin);
            else hipLaunchKernelGGL((k_rescale_hist_v8<false>), dim3(blocks_b), dim3(256), 0, st, d_short, out_u16_device, items, fmin, fmax, hp, bins, hs.vmin);
        } else {
            if (bins <= HIST_LDS_BINS) hipLaunchKernelGGL((k_rescale_hist<true>), dim3(blocks_b), dim3(256), sizeof(unsigned int) * (size_t)bins, st, d_short, out_u16_device, n, fmin, fmax, hp, bins, hs.vmin);
            else hipLaunchKernelGGL((k_rescale_hist<false>), dim3(blocks_b), dim3(256), 0, st, d_short, out_u16_device, n, fmin, fmax, hp, bins, hs.vmin);
        }
    }
    PREP_TRY(hipGetLastError());

    double max_sum;
    memcpy(&max_sum, &hs.max_sum, sizeof max_sum);
    int max_mag;
    if (max_sum < 32768.0 * 32768.0) max_mag = (int)std::sqrt(max_sum);       // no narrowing wrap anywhere: monotone
    else {
        hipLaunchKernelGGL(k_gradmag_exact, dim3(blocks_a), dim3(256), 0, st, d_short, nx, ny, nz, r[0], r[1], r[2], d_stats);
        PREP_TRY(hipGetLastError());
        PREP_TRY(hipMemcpyAsync(&hs, d_stats, sizeof hs, hipMemcpyDeviceToHost, st));
        PREP_TRY(hipStreamSynchronize(st));
        max_mag = hs.max_mag;
    }
    PREP_TRY(hipEventRecord(ev1, st));
    if (want_hist) {
        const uint32_t m = (uint32_t)bins < hist_capacity ? (uint32_t)bins : hist_capacity;
        PREP_TRY(hipMemcpyAsync(hist, d_hist, sizeof(uint32_t) * m, hipMemcpyDeviceToHost, st));
    }
    PREP_TRY(hipStreamSynchronize(st));
    PREP_TRY(hipEventElapsedTime(&g_last_ms, ev0, ev1));
    g_last_bytes = (uint64_t)n * (esz + (write_short ? 2u : 0u) + 2u + 2u);

    info->dim[0] = nx; info->dim[1] = ny; info->dim[2] = nz;
    for (int a = 0; a < 3; ++a) info->spacing[a] = (float)spacing[a];
    info->range[0] = range0; info->range[1] = range1;
    info->maxMagnitude = (float)(double)max_mag;
    info->hist_bins = (uint32_t)bins;
    cleanup();
    return 0;
}

int svr_volume_preprocess_last_ms(float* ms, uint64_t* algorithmic_bytes)
{
    if (ms) *ms = g_last_ms;
    if (algorithmic_bytes) *algorithmic_bytes = g_last_bytes;
    return 0;
}

int svr_load_mhd(const char* path, int layout, svr_volume* volume, svr_volume_info* info, uint32_t* hist, uint32_t hist_capacity)
{
    if (svr::ensure_ready()) return svr_last_error_code();
    if (!path || !volume) return failf(-4, "svr_load_mhd: null argument");
    svr_mhd_header h;
    std::vector<uint8_t> elems;
    int rc = svr_internal_mhd_load(path, &h, &elems);
    if (rc) return rc;
    const size_t n = (size_t)h.dim[0] * h.dim[1] * h.dim[2];
    uint16_t* d_u16 = nullptr;
    hipError_t e = hipMalloc((void**)&d_u16, n * sizeof(uint16_t));
    if (e != hipSuccess) return failf((int)e, "svr_load_mhd: cannot allocate %zu bytes on the device (%s)", n * 2, hipGetErrorName(e));
    svr_volume_info local;
    rc = svr_volume_preprocess(elems.data(), h.elem_type, h.dim[0], h.dim[1], h.dim[2], h.spacing, 0, d_u16, hist, hist_capacity, &local);
    if (rc) { hipFree(d_u16); return rc; }
    std::vector<uint8_t>().swap(elems);
    uint64_t tex = svr_create_volume_texture(d_u16, h.dim[0], h.dim[1], h.dim[2], 1, layout);
    hipFree(d_u16);
    if (!tex) return svr_last_error_code();
    // VolumeReader::CreateDeviceVolume, VolumeReader.cpp:174-185 (+ cudaVolume::Set, cuda_volume.h)
    const float sx = local.spacing[0], sy = local.spacing[1], sz = local.spacing[2];
    const float ex = (float)h.dim[0] * sx, ey = (float)h.dim[1] * sy, ez = (float)h.dim[2] * sz;
    const float mx = ex - ex * 0.5f, my = ey - ey * 0.5f, mz = ez - ez * 0.5f;
    volume->bbox.vmin.x = -mx; volume->bbox.vmin.y = -my; volume->bbox.vmin.z = -mz;
    volume->bbox.vmax.x = mx; volume->bbox.vmax.y = my; volume->bbox.vmax.z = mz;
    volume->bbox.invSize.x = 1.f / (mx - -mx); volume->bbox.invSize.y = 1.f / (my - -my); volume->bbox.invSize.z = 1.f / (mz - -mz);   // cuda_bbox.h ctor
    volume->spacing.x = sx; volume->spacing.y = sy; volume->spacing.z = sz;
    volume->invSpacing.x = 1.f / sx; volume->invSpacing.y = 1.f / sy; volume->invSpacing.z = 1.f / sz;
    volume->tex = tex;
    volume->invMaxMagnitude = 1.f / local.maxMagnitude;
    if (info) *info = local;
    return 0;
}

} // extern "C"
