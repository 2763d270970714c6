/* svr_io.h -- C ABI of the host-side rows next to the render path:
 * what the reference's VolumeReader, TransferFunction, Lights and Canvas do around the seven render entry
 * points of svr_abi.h.  Same library (libsvr_hip.so), plain pointers and sizes.
 *
 *   N1  frame dump              stbi_write_tga("0.tga", W, H, 4, data)        gui/canvas.cpp:97-104
 *   N2  volume load             VolumeReader::Read + CreateDeviceVolume       core/VolumeReader.cpp:13-77, 124-185
 *   N3  transfer function       TransferFunction ctor / Save / Load           gui/transferfunction.cpp:3-126
 *   N4  environment map         Lights::SetEnvironmentLight(filename)         core/lights/lights.cpp:31-75
 *
 * File parsing is host code; everything proportional to the voxel count (cast, range, rescale, histogram,
 * gradient magnitude, brick repack) runs on the GPU.  All functions return 0 on success, or a negative
 * code with svr_last_error() set; under the default error mode (svr_set_error_mode(1)) a failure prints the
 * message and exits, as the reference does on a bad file (VolumeReader.cpp:20-39, lights.cpp:36-40).
 */
#ifndef SVR_IO_H
#define SVR_IO_H

#include <stddef.h>
#include <stdint.h>

#include "svr_abi.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- N2: MetaImage (.mhd / .mha) ------------------------------------------------------------- */
enum svr_elem_type {
    SVR_ELEM_I8 = 0, SVR_ELEM_U8, SVR_ELEM_I16, SVR_ELEM_U16, SVR_ELEM_I32, SVR_ELEM_U32, SVR_ELEM_F32, SVR_ELEM_F64
};

typedef struct svr_mhd_header {
    int32_t  ndims;                 /* 2 or 3 (a 2-D image is one slice) */
    int32_t  dim[3];
    double   spacing[3];            /* ElementSpacing (or ElementSize), 1 if absent */
    int32_t  elem_type;             /* enum svr_elem_type */
    int32_t  elem_size;             /* bytes */
    int32_t  channels;              /* ElementNumberOfChannels; only 1 is supported by the loader */
    int32_t  msb;                   /* ElementByteOrderMSB / BinaryDataByteOrderMSB */
    int32_t  compressed;            /* CompressedData = True (zlib) */
    int64_t  compressed_size;       /* CompressedDataSize, 0 if absent */
    int64_t  header_size;           /* HeaderSize; -1 = data are the last bytes of the file */
    int64_t  data_offset;           /* ElementDataFile = LOCAL: offset of the first data byte in the header file */
    char     data_file[1024];       /* resolved path of the element data (the header file itself for LOCAL; "LIST" for per-slice files) */
} svr_mhd_header;

/* what VolumeReader keeps after Read() (VolumeReader.h:44-50) */
typedef struct svr_volume_info {
    int32_t  dim[3];
    float    spacing[3];            /* float(dataSpacing), VolumeReader.cpp:52 */
    double   range[2];              /* scalar range of the short image, VolumeReader.cpp:54 */
    float    maxMagnitude;          /* VolumeReader.cpp:70-76 */
    uint32_t hist_bins;             /* range[1] - range[0], VolumeReader.cpp:59 */
} svr_volume_info;

int svr_mhd_read_header(const char* path, svr_mhd_header* out);
/* element data in the file's type, decompressed and byte-swapped to host order; dst holds
 * dim[0]*dim[1]*dim[2]*elem_size bytes (host memory) */
int svr_mhd_read_elements(const svr_mhd_header* header, void* dst, size_t dst_bytes);

/* VolumeReader::Read after the file is in memory (VolumeReader.cpp:41-76), on the GPU:
 * cast to short -> scalar range -> rescale to full-range u16 -> histogram -> max gradient magnitude.
 *   elems            nx*ny*nz elements of elem_type, x fastest; host pointer, or device pointer if elems_on_device
 *   out_u16_device   device buffer of nx*ny*nz uint16 (svr_malloc), plain [z][y][x]
 *   hist             host buffer for min(hist_bins, hist_capacity) counts, or NULL
 */
int svr_volume_preprocess(const void* elems, int elem_type, int nx, int ny, int nz, const double spacing[3],
                          int elems_on_device, uint16_t* out_u16_device,
                          uint32_t* hist, uint32_t hist_capacity, svr_volume_info* info);

/* VolumeReader::Read(filename) + CreateDeviceVolume(volume) (VolumeReader.cpp:13-77, 174-185): fills
 * volume->bbox (centred at the origin, size dim*spacing), spacing / invSpacing, tex (SVR_LAYOUT_*),
 * invMaxMagnitude = 1 / maxMagnitude; the other fields of *volume are left alone, as the reference's
 * cudaVolume::Set does.  The caller owns volume->tex (svr_destroy_texture). */
int svr_load_mhd(const char* path, int layout, svr_volume* volume, svr_volume_info* info,
                 uint32_t* hist, uint32_t hist_capacity);

/* kernel time of the last svr_volume_preprocess / svr_load_mhd in milliseconds (HIP events around the
 * preprocessing kernels, upload excluded), and the bytes those kernels have to move at least */
int svr_volume_preprocess_last_ms(float* ms, uint64_t* algorithmic_bytes);

/* ---- N3: transfer function --------------------------------------------------------------------- */
/* vtkPiecewiseFunction nodes are (x, y, midpoint, sharpness), vtkColorTransferFunction nodes
 * (x, r, g, b, midpoint, sharpness), as GetNodeValue returns them (transferfunction.cpp:74-87).
 * Builds the TABLE_SIZE x RGBA table of the TransferFunction constructor (transferfunction.cpp:17-28):
 * opacityTF->GetTable(0,1,n) / colorTF->GetTable(0,1,n) (RGB space, linear scale, clamping on),
 * interleaved; *max_opacity = max of the opacity column (0 if n_opacity == 0). */
int svr_tf_build_table(const double* opacity_nodes, int n_opacity, const double* color_nodes, int n_color,
                       int table_size, float* table_rgba, float* max_opacity);
/* binary .tf: int n; n x 4 f64; int m; m x 6 f64 (transferfunction.cpp:55-126).  svr_tf_load: pass
 * capacities in *n_opacity / *n_color (nodes), receives the counts. */
int svr_tf_save(const char* path, const double* opacity_nodes, int n_opacity, const double* color_nodes, int n_color);
int svr_tf_load(const char* path, double* opacity_nodes, int* n_opacity, double* color_nodes, int* n_color);

/* ---- N4: Radiance .hdr ------------------------------------------------------------------------- */
/* stbi_loadf(filename, &w, &h, &n, 0) for .hdr files (stb_image.h v2.12, 6072-6239), then the float4
 * expansion of lights.cpp:45-53 (w = 0).  Two-call protocol: with rgba == NULL only *w, *h are returned. */
int svr_hdr_load(const char* path, int* w, int* h, float* rgba, size_t rgba_floats);
/* Lights::SetEnvironmentLight(filename): load + svr_create_env_texture + cudaEnvironmentLight::Set(tex), which
 * also resets intensity to 1 and offset to 0 (lights.cpp:31-75, cuda_environment_light.h:18-23) */
int svr_load_env_map(const char* path, svr_environment_light* env);

/* ---- N1: frame dump ---------------------------------------------------------------------------- */
/* stbi_write_tga(filename, w, h, 4, data) with the writer's default RLE (stb_image_write.h v1.02, 387-452);
 * rgba is host memory, top row first.  svr_tga_encode writes into dst (capacity bytes) and returns the
 * encoded size in *size (call with dst == NULL to get an upper bound). */
int svr_tga_write(const char* path, int w, int h, const uint8_t* rgba);
int svr_tga_encode(int w, int h, const uint8_t* rgba, uint8_t* dst, size_t capacity, size_t* size);

#ifdef __cplusplus
}
#endif
#endif /* SVR_IO_H */
