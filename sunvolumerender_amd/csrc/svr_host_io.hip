// svr_host_io.hip -- host-side file formats next to the render path (include/svr_io.h):
// MetaImage headers and element data, transfer-function nodes <-> table and the .tf file, Radiance .hdr
// environment maps, the TGA frame dump.  No device code here; the voxel-proportional work of a volume load is
// in svr_volume_prep.hip.
#include <zlib.h>

#include <algorithm>
#include <cerrno>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <memory>
#include <string>
#include <vector>

#include "svr_internal.hpp"
#include "svr_io.h"

namespace {

using svr::failf;

// ------------------------------------------------------------------------------------------------
// small file helpers
// ------------------------------------------------------------------------------------------------
bool read_file(const std::string& path, std::vector<uint8_t>& out, std::string& why)
{
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) { why = std::string(strerror(errno)); return false; }
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    if (n < 0) { fclose(f); why = "cannot determine the file size"; return false; }
    out.resize((size_t)n);
    size_t got = n ? fread(out.data(), 1, (size_t)n, f) : 0;
    fclose(f);
    if (got != (size_t)n) { why = "short read"; return false; }
    return true;
}

std::string trim(const std::string& s)
{
    size_t a = 0, b = s.size();
    while (a < b && (s[a] == ' ' || s[a] == '\t' || s[a] == '\r')) ++a;
    while (b > a && (s[b - 1] == ' ' || s[b - 1] == '\t' || s[b - 1] == '\r')) --b;
    return s.substr(a, b - a);
}

std::string dir_of(const std::string& path)
{
    size_t p = path.find_last_of('/');
    return p == std::string::npos ? std::string() : path.substr(0, p + 1);
}

bool truthy(const std::string& v) { return !v.empty() && (v[0] == 'T' || v[0] == 't' || v[0] == '1'); }

// ------------------------------------------------------------------------------------------------
// MetaImage
// ------------------------------------------------------------------------------------------------
struct ElemName { const char* name; int type; int size; };
const ElemName kElems[] = {
    {"MET_CHAR", SVR_ELEM_I8, 1},    {"MET_UCHAR", SVR_ELEM_U8, 1},   {"MET_SHORT", SVR_ELEM_I16, 2},
    {"MET_USHORT", SVR_ELEM_U16, 2}, {"MET_INT", SVR_ELEM_I32, 4},    {"MET_UINT", SVR_ELEM_U32, 4},
    {"MET_LONG", SVR_ELEM_I32, 4},   {"MET_ULONG", SVR_ELEM_U32, 4},  // MetaIO's MET_LONG is 4 bytes
    {"MET_FLOAT", SVR_ELEM_F32, 4},  {"MET_DOUBLE", SVR_ELEM_F64, 8},
};

int parse_mhd(const char* path, svr_mhd_header* h, std::vector<std::string>* slice_files)
{
    if (!path || !h) return failf(-4, "svr_mhd_read_header: null argument");
    std::vector<uint8_t> bytes;
    std::string why;
    if (!read_file(path, bytes, why)) return failf(-20, "%s does not exist or cannot be read (%s)", path, why.c_str());
    memset(h, 0, sizeof *h);
    h->ndims = 0; h->channels = 1; h->elem_type = -1;
    for (int a = 0; a < 3; ++a) { h->dim[a] = 1; h->spacing[a] = 1.0; }
    bool have_spacing = false, have_data = false, is_image = true;
    std::string data_file;
    size_t pos = 0;
    while (pos < bytes.size() && !have_data) {
        size_t eol = pos;
        while (eol < bytes.size() && bytes[eol] != '\n') ++eol;
        std::string line((const char*)bytes.data() + pos, eol - pos);
        pos = eol < bytes.size() ? eol + 1 : eol;
        size_t eq = line.find('=');
        if (eq == std::string::npos) { if (trim(line).empty()) continue; return failf(-21, "%s: not a MetaImage header (line without '=': \"%.60s\")", path, line.c_str()); }
        std::string key = trim(line.substr(0, eq)), val = trim(line.substr(eq + 1));
        const char* v = val.c_str();
        if (key == "ObjectType") is_image = (val == "Image");
        else if (key == "NDims") h->ndims = atoi(v);
        else if (key == "DimSize") {
            char* e = nullptr;
            for (int a = 0; a < 3; ++a) { long d = strtol(v, &e, 10); if (e == v) break; h->dim[a] = (int)d; v = e; }
        } else if (key == "ElementSpacing" || (key == "ElementSize" && !have_spacing)) {
            char* e = nullptr;
            for (int a = 0; a < 3; ++a) { double d = strtod(v, &e); if (e == v) break; h->spacing[a] = d; v = e; }
            if (key == "ElementSpacing") have_spacing = true;
        } else if (key == "ElementType") {
            for (const ElemName& en : kElems) if (val == en.name) { h->elem_type = en.type; h->elem_size = en.size; }
            if (h->elem_type < 0) return failf(-21, "%s: unsupported ElementType %s", path, v);
        } else if (key == "ElementNumberOfChannels") h->channels = atoi(v);
        else if (key == "ElementByteOrderMSB" || key == "BinaryDataByteOrderMSB") h->msb = truthy(val);
        else if (key == "CompressedData") h->compressed = truthy(val);
        else if (key == "CompressedDataSize") h->compressed_size = atoll(v);
        else if (key == "HeaderSize") h->header_size = atoll(v);
        else if (key == "BinaryData") { if (!truthy(val)) return failf(-21, "%s: ASCII element data are not supported", path); }
        else if (key == "ElementDataFile") { data_file = val; have_data = true; }
        // Offset / Position / Origin / TransformMatrix / AnatomicalOrientation / Comment ...: the reference
        // ignores them too (it recentres the volume at the origin, VolumeReader.cpp:176-180)
    }
    if (!is_image) return failf(-21, "%s: ObjectType is not Image", path);
    if (!have_data) return failf(-21, "%s: no ElementDataFile", path);
    if (h->ndims < 2 || h->ndims > 3) return failf(-21, "%s: NDims = %d (2 or 3 supported)", path, h->ndims);
    if (h->ndims == 2) h->dim[2] = 1;
    if (h->elem_type < 0) return failf(-21, "%s: no ElementType", path);
    if ((uint64_t)(h->dim[0] > 0 ? h->dim[0] : 0) * (uint64_t)(h->dim[1] > 0 ? h->dim[1] : 0) * (uint64_t)(h->dim[2] > 0 ? h->dim[2] : 0) >= (1ull << 31))
        return failf(-21, "%s: %d x %d x %d voxels exceed what one volume texture can address", path, h->dim[0], h->dim[1], h->dim[2]);
    for (int a = 0; a < 3; ++a) {
        if (h->dim[a] <= 0) return failf(-21, "%s: bad DimSize", path);
        h->spacing[a] = std::fabs(h->spacing[a]);
        if (!(h->spacing[a] > 0.0)) h->spacing[a] = 1.0;
    }
    if (data_file == "LOCAL" || data_file == "Local" || data_file == "local") {
        h->data_offset = (int64_t)pos;
        snprintf(h->data_file, sizeof h->data_file, "%s", path);
    } else if (data_file == "LIST" || data_file.compare(0, 5, "LIST ") == 0) {
        // one file per slice on the following lines
        if (!slice_files) { snprintf(h->data_file, sizeof h->data_file, "LIST"); return 0; }
        while (pos < bytes.size() && (int)slice_files->size() < h->dim[2]) {
            size_t eol = pos;
            while (eol < bytes.size() && bytes[eol] != '\n') ++eol;
            std::string f = trim(std::string((const char*)bytes.data() + pos, eol - pos));
            pos = eol < bytes.size() ? eol + 1 : eol;
            if (!f.empty()) slice_files->push_back(f[0] == '/' ? f : dir_of(path) + f);
        }
        if ((int)slice_files->size() != h->dim[2]) return failf(-21, "%s: LIST names %zu files for %d slices", path, slice_files->size(), h->dim[2]);
        snprintf(h->data_file, sizeof h->data_file, "LIST");
    } else {
        if (data_file.find('%') != std::string::npos) return failf(-21, "%s: printf-style ElementDataFile patterns are not supported", path);
        std::string full = data_file[0] == '/' ? data_file : dir_of(path) + data_file;
        if (full.size() >= sizeof h->data_file) return failf(-21, "%s: ElementDataFile path too long", path);
        snprintf(h->data_file, sizeof h->data_file, "%s", full.c_str());
    }
    return 0;
}

void byteswap(uint8_t* p, size_t n_elems, int size)
{
    if (size == 2) for (size_t i = 0; i < n_elems; ++i) std::swap(p[2 * i], p[2 * i + 1]);
    else if (size == 4) for (size_t i = 0; i < n_elems; ++i) { std::swap(p[4 * i], p[4 * i + 3]); std::swap(p[4 * i + 1], p[4 * i + 2]); }
    else if (size == 8) for (size_t i = 0; i < n_elems; ++i) for (int k = 0; k < 4; ++k) std::swap(p[8 * i + k], p[8 * i + 7 - k]);
}

int read_elements_from(const svr_mhd_header* h, const std::string& file, int64_t offset, uint8_t* dst, size_t want)
{
    std::vector<uint8_t> raw;
    std::string why;
    if (!read_file(file, raw, why)) return failf(-20, "%s cannot be read (%s)", file.c_str(), why.c_str());
    if (h->compressed) {
        size_t off = offset > 0 ? (size_t)offset : 0;
        if (off > raw.size()) return failf(-22, "%s: data offset beyond the end of the file", file.c_str());
        uLongf out_len = (uLongf)want;
        uLong in_len = (uLong)(raw.size() - off);
        if (h->compressed_size > 0 && (uLong)h->compressed_size < in_len) in_len = (uLong)h->compressed_size;
        int z = uncompress(dst, &out_len, raw.data() + off, in_len);
        if (z != Z_OK || out_len != want) return failf(-22, "%s: zlib stream does not decode to %zu bytes (zlib %d, got %lu)", file.c_str(), want, z, (unsigned long)out_len);
        return 0;
    }
    size_t off;
    if (offset < 0) {                         // HeaderSize = -1: the data are the last bytes of the file
        if (raw.size() < want) return failf(-22, "%s: %zu bytes, %zu needed", file.c_str(), raw.size(), want);
        off = raw.size() - want;
    } else off = (size_t)offset;
    if (off + want > raw.size()) return failf(-22, "%s: %zu bytes after offset %zu, %zu needed", file.c_str(), raw.size() - std::min(off, raw.size()), off, want);
    memcpy(dst, raw.data() + off, want);
    return 0;
}

// ------------------------------------------------------------------------------------------------
// transfer function: vtkPiecewiseFunction::GetTable / vtkColorTransferFunction::GetTable (RGB space)
// ------------------------------------------------------------------------------------------------
struct Segment {                    // the node pair around x
    const double* lo = nullptr; const double* hi = nullptr;
    double mid = 0.5, sharp = 0.0;
};

// value of one channel between two nodes; `s` is the position in the segment after the midpoint warp
double blend(double s, double y_lo, double y_hi, double sharp, bool colour)
{
    if (sharp > 0.99) return s < 0.5 ? y_lo : y_hi;                 // step at the midpoint
    if (sharp < 0.01) return (1.0 - s) * y_lo + s * y_hi;           // linear
    const double s2 = s * s, s3 = s2 * s;
    const double slope = y_hi - y_lo, tangent = (1.0 - sharp) * slope;
    double v = (2.0 * s3 - 3.0 * s2 + 1.0) * y_lo + (-2.0 * s3 + 3.0 * s2) * y_hi + (s3 - 2.0 * s2 + s) * tangent + (s3 - s2) * tangent;
    const double lo = colour ? 0.0 : std::min(y_lo, y_hi), hi = colour ? 1.0 : std::max(y_lo, y_hi);
    v = v < lo ? lo : v;
    v = v > hi ? hi : v;
    return v;
}

// samples `channels` values per entry at x_i = i/(size-1) from nodes of `stride` doubles:
// (x, v_0..v_{channels-1}, midpoint, sharpness)
void sample_nodes(const double* nodes, int n, int stride, int channels, int size, float* out, int out_stride, int out_off)
{
    int next = 0;                    // first node with X >= x
    Segment seg;
    for (int i = 0; i < size; ++i) {
        const double x = size > 1 ? 0.0 + ((double)i / (double)(size - 1)) * (1.0 - 0.0) : 0.5;
        while (next < n && x > nodes[(size_t)stride * next]) {
            ++next;
            if (next < n) {
                seg.lo = nodes + (size_t)stride * (next - 1);
                seg.hi = nodes + (size_t)stride * next;
                seg.mid = std::min(std::max(seg.lo[1 + channels], 0.00001), 0.99999);
                seg.sharp = seg.lo[2 + channels];
            }
        }
        float* o = out + (size_t)out_stride * i + out_off;
        if (n == 0) { for (int c = 0; c < channels; ++c) o[c] = 0.f; continue; }
        if (next >= n) { for (int c = 0; c < channels; ++c) o[c] = (float)nodes[(size_t)stride * (n - 1) + 1 + c]; continue; }   // clamping on
        if (next == 0) { for (int c = 0; c < channels; ++c) o[c] = (float)nodes[1 + c]; continue; }
        double s = (x - seg.lo[0]) / (seg.hi[0] - seg.lo[0]);
        s = s < seg.mid ? 0.5 * s / seg.mid : 0.5 + 0.5 * (s - seg.mid) / (1.0 - seg.mid);
        if (seg.sharp >= 0.01 && seg.sharp <= 0.99) {
            if (s < 0.5) s = 0.5 * std::pow(s * 2.0, 1.0 + 10.0 * seg.sharp);
            else if (s > 0.5) s = 1.0 - 0.5 * std::pow((1.0 - s) * 2.0, 1.0 + 10.0 * seg.sharp);
        }
        for (int c = 0; c < channels; ++c) o[c] = (float)blend(s, seg.lo[1 + c], seg.hi[1 + c], seg.sharp, channels == 3);
    }
}

// ------------------------------------------------------------------------------------------------
// Radiance RGBE
// ------------------------------------------------------------------------------------------------
struct Reader {
    const uint8_t* p; size_t n, at = 0;
    int get() { return at < n ? p[at++] : 0; }           // stb returns 0 past the end
    bool eof() const { return at >= n; }
    std::string line()                                   // stbi__hdr_gettoken: up to '\n', at most 1023 chars kept
    {
        std::string s;
        char c = (char)get();
        while (!eof() && c != '\n') {
            s.push_back(c);
            if (s.size() == 1023) { while (!eof() && get() != '\n') {} break; }
            c = (char)get();
        }
        return s;
    }
};

inline void rgbe_to_rgb(const uint8_t* in, float* out)
{
    if (in[3] != 0) {
        float f = (float)std::ldexp(1.0f, (int)in[3] - (128 + 8));
        out[0] = in[0] * f; out[1] = in[1] * f; out[2] = in[2] * f;
    } else out[0] = out[1] = out[2] = 0.f;
}

int decode_hdr(const char* path, const std::vector<uint8_t>& bytes, int* w_out, int* h_out, std::vector<float>* rgb)
{
    Reader r{bytes.data(), bytes.size()};
    if (r.line() != "#?RADIANCE") return failf(-23, "%s: not a Radiance HDR file", path);
    bool rle_rgbe = false;
    for (;;) {
        std::string t = r.line();
        if (t.empty()) break;
        if (t == "FORMAT=32-bit_rle_rgbe") rle_rgbe = true;
    }
    if (!rle_rgbe) return failf(-23, "%s: unsupported HDR format (FORMAT=32-bit_rle_rgbe expected)", path);
    std::string res = r.line();
    if (res.compare(0, 3, "-Y ") != 0) return failf(-23, "%s: unsupported HDR data layout", path);
    char* e = nullptr;
    const char* q = res.c_str() + 3;
    int height = (int)strtol(q, &e, 10);
    while (*e == ' ') ++e;
    if (strncmp(e, "+X ", 3) != 0) return failf(-23, "%s: unsupported HDR data layout", path);
    int width = (int)strtol(e + 3, nullptr, 10);
    if (width <= 0 || height <= 0 || (int64_t)width * height > ((int64_t)1 << 28)) return failf(-23, "%s: bad HDR size %d x %d", path, width, height);
    *w_out = width; *h_out = height;
    if (!rgb) return 0;
    rgb->assign((size_t)width * height * 3, 0.f);
    float* out = rgb->data();
    int j = 0, i = 0;
    bool flat = width < 8 || width >= 32768;
    std::vector<uint8_t> scan;
    if (!flat) {
        scan.resize((size_t)width * 4);
        for (j = 0; j < height; ++j) {
            int c1 = r.get(), c2 = r.get(), len = r.get();
            if (c1 != 2 || c2 != 2 || (len & 0x80)) {
                // not run-length encoded: these four bytes are the first pixel of a flat file
                uint8_t px[4] = {(uint8_t)c1, (uint8_t)c2, (uint8_t)len, (uint8_t)r.get()};
                rgbe_to_rgb(px, out);
                flat = true; j = 0; i = 1;         // stb re-enters its flat loop at (row 0, column 1)
                break;
            }
            len = (len << 8) | r.get();
            if (len != width) return failf(-23, "%s: corrupt HDR (invalid decoded scanline length)", path);
            for (int k = 0; k < 4; ++k) {
                int x = 0;
                while (x < width) {
                    int count = r.get();
                    if (count > 128) {
                        int value = r.get();
                        count -= 128;
                        for (int z = 0; z < count && x < width; ++z) scan[(size_t)x++ * 4 + k] = (uint8_t)value;
                    } else {
                        if (count == 0 && r.eof()) return failf(-23, "%s: truncated HDR scanline", path);
                        for (int z = 0; z < count && x < width; ++z) scan[(size_t)x++ * 4 + k] = (uint8_t)r.get();
                    }
                }
            }
            for (int x = 0; x < width; ++x) rgbe_to_rgb(&scan[(size_t)x * 4], out + ((size_t)j * width + x) * 3);
        }
    }
    if (flat) {
        for (; j < height; ++j, i = 0)
            for (; i < width; ++i) {
                uint8_t px[4] = {(uint8_t)r.get(), (uint8_t)r.get(), (uint8_t)r.get(), (uint8_t)r.get()};
                rgbe_to_rgb(px, out + ((size_t)j * width + i) * 3);
            }
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------
// TGA (type 10: run-length encoded true colour, 32 bpp, 8 alpha bits, bottom-up, BGRA)
// ------------------------------------------------------------------------------------------------
size_t tga_encode(int w, int h, const uint8_t* rgba, uint8_t* dst)
{
    size_t n = 0;
    auto put = [&](uint8_t b) { if (dst) dst[n] = b; ++n; };
    auto put16 = [&](int v) { put((uint8_t)(v & 0xff)); put((uint8_t)((v >> 8) & 0xff)); };
    auto pixel = [&](const uint8_t* p) { put(p[2]); put(p[1]); put(p[0]); put(p[3]); };
    auto same = [](const uint8_t* a, const uint8_t* b) { return memcmp(a, b, 4) == 0; };
    put(0); put(0); put(10);                   // no id, no colour map, RLE true colour
    put16(0); put16(0); put(0);                // colour-map spec
    put16(0); put16(0); put16(w); put16(h);
    put(32); put(8);
    for (int row = h - 1; row >= 0; --row) {
        const uint8_t* line = rgba + (size_t)row * w * 4;
        int x = 0;
        while (x < w) {
            const uint8_t* first = line + (size_t)x * 4;
            int len = 1;
            bool run = false;
            if (x < w - 1) {
                len = 2;
                run = same(first, first + 4);
                if (run) {
                    for (int k = x + 2; k < w && len < 128 && same(first, line + (size_t)k * 4); ++k) ++len;
                } else {
                    // a raw packet ends before the next pair of equal pixels
                    const uint8_t* prev = first;
                    for (int k = x + 2; k < w && len < 128; ++k) {
                        if (!same(prev, line + (size_t)k * 4)) { prev += 4; ++len; }
                        else { --len; break; }
                    }
                }
            }
            if (run) { put((uint8_t)(len - 129)); pixel(first); }
            else { put((uint8_t)(len - 1)); for (int k = 0; k < len; ++k) pixel(first + (size_t)k * 4); }
            x += len;
        }
    }
    return n;
}

} // namespace

extern "C" {

int svr_mhd_read_header(const char* path, svr_mhd_header* out) { return parse_mhd(path, out, nullptr); }

int svr_mhd_read_elements(const svr_mhd_header* h, void* dst, size_t dst_bytes)
{
    if (!h || !dst) return failf(-4, "svr_mhd_read_elements: null argument");
    if (h->channels != 1) return failf(-21, "ElementNumberOfChannels = %d is not supported (the reference renders scalar volumes)", h->channels);
    const size_t n = (size_t)h->dim[0] * h->dim[1] * h->dim[2];
    const size_t want = n * (size_t)h->elem_size;
    if (dst_bytes < want) return failf(-6, "svr_mhd_read_elements: buffer of %zu bytes, %zu needed", dst_bytes, want);
    uint8_t* d = (uint8_t*)dst;
    if (strcmp(h->data_file, "LIST") == 0) return failf(-21, "LIST element files: use svr_load_mhd or read the header file again");
    // LOCAL: the data follow the header line; separate file: HeaderSize bytes are skipped, -1 = the data are
    // the last bytes of the file
    const int64_t offset = h->data_offset > 0 ? h->data_offset : h->header_size;
    int rc = read_elements_from(h, h->data_file, offset, d, want);
    if (rc) return rc;
    if (h->msb && h->elem_size > 1) byteswap(d, n, h->elem_size);
    return 0;
}

// internal: header + elements in one go, LIST included (used by svr_load_mhd)
int svr_internal_mhd_load(const char* path, svr_mhd_header* h, std::vector<uint8_t>* elems)
{
    std::vector<std::string> slices;
    int rc = parse_mhd(path, h, &slices);
    if (rc) return rc;
    if (h->channels != 1) return failf(-21, "%s: ElementNumberOfChannels = %d is not supported (the reference renders scalar volumes)", path, h->channels);
    const size_t n = (size_t)h->dim[0] * h->dim[1] * h->dim[2];
    elems->resize(n * (size_t)h->elem_size);
    if (!slices.empty()) {
        const size_t per = (size_t)h->dim[0] * h->dim[1] * (size_t)h->elem_size;
        for (size_t z = 0; z < slices.size(); ++z) {
            rc = read_elements_from(h, slices[z], h->header_size, elems->data() + z * per, per);
            if (rc) return rc;
        }
        if (h->msb && h->elem_size > 1) byteswap(elems->data(), n, h->elem_size);
        return 0;
    }
    return svr_mhd_read_elements(h, elems->data(), elems->size());
}

int svr_tf_build_table(const double* opacity_nodes, int n_opacity, const double* color_nodes, int n_color,
                       int table_size, float* table_rgba, float* max_opacity)
{
    if (!table_rgba || table_size <= 0 || n_opacity < 0 || n_color < 0 || (n_opacity && !opacity_nodes) || (n_color && !color_nodes))
        return failf(-4, "svr_tf_build_table: bad arguments");
    for (int i = 1; i < n_opacity; ++i)
        if (opacity_nodes[4 * i] < opacity_nodes[4 * (i - 1)]) return failf(-6, "svr_tf_build_table: opacity nodes must be sorted by x");
    for (int i = 1; i < n_color; ++i)
        if (color_nodes[6 * i] < color_nodes[6 * (i - 1)]) return failf(-6, "svr_tf_build_table: colour nodes must be sorted by x");
    sample_nodes(color_nodes, n_color, 6, 3, table_size, table_rgba, 4, 0);
    sample_nodes(opacity_nodes, n_opacity, 4, 1, table_size, table_rgba, 4, 3);
    float mo = 0.f;                                        // TransferFunction::maxOpacity starts at 0 (transferfunction.h)
    for (int i = 0; i < table_size; ++i) mo = fmaxf(mo, table_rgba[4 * (size_t)i + 3]);
    if (max_opacity) *max_opacity = mo;
    return 0;
}

int svr_tf_save(const char* path, const double* opacity_nodes, int n_opacity, const double* color_nodes, int n_color)
{
    if (!path || n_opacity < 0 || n_color < 0) return failf(-4, "svr_tf_save: bad arguments");
    FILE* f = fopen(path, "wb");
    if (!f) return failf(-20, "unable to open %s for writing (%s)", path, strerror(errno));
    int32_t n = n_opacity, m = n_color;
    bool ok = fwrite(&n, 4, 1, f) == 1;
    ok = ok && (n == 0 || fwrite(opacity_nodes, sizeof(double) * 4, (size_t)n, f) == (size_t)n);
    ok = ok && fwrite(&m, 4, 1, f) == 1;
    ok = ok && (m == 0 || fwrite(color_nodes, sizeof(double) * 6, (size_t)m, f) == (size_t)m);
    ok = (fclose(f) == 0) && ok;
    return ok ? 0 : failf(-20, "short write to %s", path);
}

int svr_tf_load(const char* path, double* opacity_nodes, int* n_opacity, double* color_nodes, int* n_color)
{
    if (!path || !n_opacity || !n_color) return failf(-4, "svr_tf_load: null argument");
    std::vector<uint8_t> b;
    std::string why;
    if (!read_file(path, b, why)) return failf(-20, "unable to open %s (%s)", path, why.c_str());
    size_t at = 0;
    auto take = [&](void* dst, size_t n) { if (at + n > b.size()) return false; if (dst) memcpy(dst, b.data() + at, n); at += n; return true; };
    int32_t n = 0, m = 0;
    if (!take(&n, 4) || n < 0 || (size_t)n * 32 > b.size()) return failf(-24, "%s: not a .tf file", path);
    if (opacity_nodes && n > *n_opacity) return failf(-6, "%s: %d opacity nodes, room for %d", path, n, *n_opacity);
    if (!take(opacity_nodes, (size_t)n * 32)) return failf(-24, "%s: truncated .tf file", path);
    if (!take(&m, 4) || m < 0 || (size_t)m * 48 > b.size()) return failf(-24, "%s: truncated .tf file", path);
    if (color_nodes && m > *n_color) return failf(-6, "%s: %d colour nodes, room for %d", path, m, *n_color);
    if (!take(color_nodes, (size_t)m * 48)) return failf(-24, "%s: truncated .tf file", path);
    *n_opacity = n; *n_color = m;
    return 0;
}

int svr_hdr_load(const char* path, int* w, int* h, float* rgba, size_t rgba_floats)
{
    if (!path || !w || !h) return failf(-4, "svr_hdr_load: null argument");
    std::vector<uint8_t> bytes;
    std::string why;
    if (!read_file(path, bytes, why)) return failf(-20, "Unable to load environment map: %s (%s)", path, why.c_str());
    if (!rgba) return decode_hdr(path, bytes, w, h, nullptr);
    std::vector<float> rgb;
    int rc = decode_hdr(path, bytes, w, h, &rgb);
    if (rc) return rc;
    const size_t count = (size_t)*w * *h;
    if (rgba_floats < count * 4) return failf(-6, "svr_hdr_load: buffer of %zu floats, %zu needed", rgba_floats, count * 4);
    for (size_t i = 0; i < count; ++i) {                   // lights.cpp:45-53
        rgba[4 * i] = rgb[3 * i]; rgba[4 * i + 1] = rgb[3 * i + 1]; rgba[4 * i + 2] = rgb[3 * i + 2]; rgba[4 * i + 3] = 0.f;
    }
    return 0;
}

int svr_load_env_map(const char* path, svr_environment_light* env)
{
    if (!env) return failf(-4, "svr_load_env_map: null argument");
    int w = 0, h = 0;
    int rc = svr_hdr_load(path, &w, &h, nullptr, 0);
    if (rc) return rc;
    std::vector<float> rgba((size_t)w * h * 4);
    rc = svr_hdr_load(path, &w, &h, rgba.data(), rgba.size());
    if (rc) return rc;
    uint64_t tex = svr_create_env_texture(rgba.data(), w, h, 0);
    if (!tex) return svr_last_error_code();
    // cudaEnvironmentLight::Set(tex) (lights.cpp:74, cuda_environment_light.h:18-23) also resets intensity and offset
    env->tex = tex;
    env->intensity = 1.f;
    env->offset.x = 0.f; env->offset.y = 0.f;
    return 0;
}

int svr_tga_encode(int w, int h, const uint8_t* rgba, uint8_t* dst, size_t capacity, size_t* size)
{
    if (w < 0 || h < 0 || w > 65535 || h > 65535 || !size) return failf(-4, "svr_tga_encode: bad arguments");
    if (!dst) { *size = 18 + (size_t)w * h * 5; return 0; }      // every pixel its own packet
    if (!rgba && w * h) return failf(-4, "svr_tga_encode: null image");
    size_t need = tga_encode(w, h, rgba, nullptr);
    if (need > capacity) return failf(-6, "svr_tga_encode: %zu bytes needed, %zu given", need, capacity);
    *size = tga_encode(w, h, rgba, dst);
    return 0;
}

int svr_tga_write(const char* path, int w, int h, const uint8_t* rgba)
{
    if (!path || w < 0 || h < 0 || w > 65535 || h > 65535 || (!rgba && w * h)) return failf(-4, "svr_tga_write: bad arguments");
    std::vector<uint8_t> buf(tga_encode(w, h, rgba, nullptr));
    tga_encode(w, h, rgba, buf.data());
    FILE* f = fopen(path, "wb");
    if (!f) return failf(-20, "unable to open %s for writing (%s)", path, strerror(errno));
    bool ok = fwrite(buf.data(), 1, buf.size(), f) == buf.size();
    ok = (fclose(f) == 0) && ok;
    return ok ? 0 : failf(-20, "short write to %s", path);
}

} // extern "C"
