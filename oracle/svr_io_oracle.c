/* svr_io_oracle.c -- see svr_io_oracle.h (test infrastructure only; VTK parts parity-unpinned). */
#include "svr_io_oracle.h"

#include <math.h>
#include <string.h>

/* C leaves out-of-range float->integer conversions and narrowing to the implementation; the reference
 * ran on x86-64 (cvttss2si + truncation), which this spells out: truncate toward zero into int64
 * (saturating, NaN -> INT64_MIN like cvttsd2si), then keep the low 16 bits, two's complement. */
static int16_t wrap16(int64_t v) { return (int16_t)(uint16_t)((uint64_t)v & 0xffffu); }
static int64_t trunc_i64(double d)
{
    if (!(d == d)) return INT64_MIN;
    if (d >= 9223372036854775808.0 || d < -9223372036854775808.0) return INT64_MIN;
    return (int64_t)d;
}

void svo_cast_to_short(const void* src, int elem_type, size_t n, int16_t* dst)
{
    for (size_t i = 0; i < n; ++i) {
        int64_t v;
        switch (elem_type) {
        case SVO_ELEM_I8:  v = ((const int8_t*)src)[i]; break;
        case SVO_ELEM_U8:  v = ((const uint8_t*)src)[i]; break;
        case SVO_ELEM_I16: v = ((const int16_t*)src)[i]; break;
        case SVO_ELEM_U16: v = ((const uint16_t*)src)[i]; break;
        case SVO_ELEM_I32: v = ((const int32_t*)src)[i]; break;
        case SVO_ELEM_U32: v = ((const uint32_t*)src)[i]; break;
        case SVO_ELEM_F32: v = trunc_i64((double)((const float*)src)[i]); break;
        default:           v = trunc_i64(((const double*)src)[i]); break;
        }
        dst[i] = wrap16(v);
    }
}

void svo_scalar_range(const int16_t* v, size_t n, double range[2])
{
    int lo = 32767, hi = -32768;
    for (size_t i = 0; i < n; ++i) {
        if (v[i] < lo) lo = v[i];
        if (v[i] > hi) hi = v[i];
    }
    range[0] = lo; range[1] = hi;
}

void svo_rescale(const int16_t* src, size_t n, float dataMin, float dataMax, uint16_t* dst)
{
    float extent = dataMax - dataMin;
    float dataTypeExtent = (float)(65535 - 0);
    for (size_t i = 0; i < n; ++i) {
        float r = ((float)src[i] - dataMin) / extent * dataTypeExtent;
        /* float -> unsigned short: defined for [0, 65536); 0/0 (constant volume) pinned to 0 */
        dst[i] = (r == r) ? (uint16_t)wrap16(trunc_i64((double)r)) : 0;
    }
}

int svo_histogram(const int16_t* v, size_t n, double rmin, double rmax, uint32_t* hist, int capacity)
{
    /* SetComponentExtent(0, max - min - 1, ...): bins = max - min; origin = min; spacing 1; IgnoreZeroOn */
    int extent_hi = (int)(rmax - rmin - 1.0);
    int bins = extent_hi + 1;
    if (bins < 0) bins = 0;
    int m = bins < capacity ? bins : capacity;
    memset(hist, 0, sizeof(uint32_t) * (size_t)(m > 0 ? m : 0));
    for (size_t i = 0; i < n; ++i) {
        if (v[i] == 0) continue;
        double idx = floor(((double)v[i] - rmin) / 1.0);
        if (idx < 0.0 || idx > (double)extent_hi) continue;
        if ((int)idx < m) hist[(int)idx]++;
    }
    return bins;
}

float svo_max_gradient_magnitude(const int16_t* v, int nx, int ny, int nz, const double spacing[3])
{
    double r[3] = {0.5 / spacing[0], 0.5 / spacing[1], 0.5 / spacing[2]};
    int best = -32768;
#pragma omp parallel for reduction(max : best)
    for (int z = 0; z < nz; ++z)
        for (int y = 0; y < ny; ++y)
            for (int x = 0; x < nx; ++x) {
                size_t i = ((size_t)z * ny + y) * nx + x;
                size_t sx = 1, sy = (size_t)nx, sz = (size_t)nx * ny;
                double d, sum = 0.0;
                d = (double)v[x > 0 ? i - sx : i] - (double)v[x < nx - 1 ? i + sx : i]; d *= r[0]; sum += d * d;
                d = (double)v[y > 0 ? i - sy : i] - (double)v[y < ny - 1 ? i + sy : i]; d *= r[1]; sum += d * d;
                d = (double)v[z > 0 ? i - sz : i] - (double)v[z < nz - 1 ? i + sz : i]; d *= r[2]; sum += d * d;
                int m = wrap16(trunc_i64(sqrt(sum)));
                if (m > best) best = m;
            }
    return (float)(double)best;
}

/* ---- transfer-function tables ------------------------------------------------------------------ */
static double tf_shape(double x, double x1, double x2, double midpoint, double sharpness, int* mode)
{
    double s = (x - x1) / (x2 - x1);
    if (s < midpoint) s = 0.5 * s / midpoint;
    else s = 0.5 + 0.5 * (s - midpoint) / (1.0 - midpoint);
    if (sharpness > 0.99) { *mode = 2; return s; }     /* step at the midpoint */
    if (sharpness < 0.01) { *mode = 1; return s; }     /* piecewise linear */
    *mode = 0;
    if (s < 0.5) s = 0.5 * pow(s * 2.0, 1.0 + 10.0 * sharpness);
    else if (s > 0.5) s = 1.0 - 0.5 * pow((1.0 - s) * 2.0, 1.0 + 10.0 * sharpness);
    return s;
}

static double tf_mix(double s, int mode, double y1, double y2, double sharpness, int unit_clamp)
{
    if (mode == 2) return s < 0.5 ? y1 : y2;
    if (mode == 1) return (1.0 - s) * y1 + s * y2;
    double ss = s * s, sss = ss * s;
    double h1 = 2.0 * sss - 3.0 * ss + 1.0, h2 = -2.0 * sss + 3.0 * ss, h3 = sss - 2.0 * ss + s, h4 = sss - ss;
    double slope = y2 - y1, t = (1.0 - sharpness) * slope;
    double val = h1 * y1 + h2 * y2 + h3 * t + h4 * t;
    /* vtkPiecewiseFunction keeps the value inside [min(y1,y2), max(y1,y2)]; vtkColorTransferFunction inside [0,1] */
    double lo = unit_clamp ? 0.0 : (y1 < y2 ? y1 : y2), hi = unit_clamp ? 1.0 : (y1 > y2 ? y1 : y2);
    val = val < lo ? lo : val;
    val = val > hi ? hi : val;
    return val;
}

static void tf_table(const double* nodes, int n_nodes, int stride, int n_val, int clamping, int size, float* table)
{
    int idx = 0;
    double x1 = 0, x2 = 0, midpoint = 0, sharpness = 0;
    const double* a = 0; const double* b = 0;
    for (int i = 0; i < size; ++i) {
        float* out = table + (size_t)n_val * i;
        double x = size > 1 ? 0.0 + ((double)i / (double)(size - 1)) * (1.0 - 0.0) : 0.5 * (0.0 + 1.0);
        while (idx < n_nodes && x > nodes[(size_t)stride * idx]) {
            idx++;
            if (idx < n_nodes) {
                a = nodes + (size_t)stride * (idx - 1); b = nodes + (size_t)stride * idx;
                x1 = a[0]; x2 = b[0];
                midpoint = a[1 + n_val]; sharpness = a[2 + n_val];
                if (midpoint < 0.00001) midpoint = 0.00001;
                if (midpoint > 0.99999) midpoint = 0.99999;
            }
        }
        if (idx >= n_nodes) {
            for (int c = 0; c < n_val; ++c) out[c] = (float)(clamping && n_nodes > 0 ? nodes[(size_t)stride * (n_nodes - 1) + 1 + c] : 0.0);
        } else if (idx == 0) {
            for (int c = 0; c < n_val; ++c) out[c] = (float)(clamping ? nodes[1 + c] : 0.0);
        } else {
            int mode;
            double s = tf_shape(x, x1, x2, midpoint, sharpness, &mode);
            for (int c = 0; c < n_val; ++c) out[c] = (float)tf_mix(s, mode, a[1 + c], b[1 + c], sharpness, n_val == 3);
        }
    }
}

void svo_piecewise_table(const double* nodes, int n_nodes, int clamping, int size, float* table)
{
    tf_table(nodes, n_nodes, 4, 1, clamping, size, table);
}

void svo_color_table(const double* nodes, int n_nodes, int clamping, int size, float* table)
{
    tf_table(nodes, n_nodes, 6, 3, clamping, size, table);
}
