/* svr_io_oracle.h -- CPU restatement of the reference's host-side data preparation (SURVEY.md section 8(f),
 * rows N1-N4): volume preprocessing after vtkMetaImageReader, the transfer-function table, the TGA frame dump.
 *
 * TEST INFRASTRUCTURE ONLY, like svr_oracle.h: tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * may use it; nothing under sunvolumerender_amd/ or include/ does.
 *
 * PARITY UNPINNED for the VTK parts: VTK (5.x, the API generation core/VolumeReader.cpp:41-73 uses:
 * SetInput / GetOutputPort mix) is not in /root/reference and not in this image, so vtkImageCast,
 * vtkImageAccumulate, vtkImageGradientMagnitude, vtkPiecewiseFunction::GetTable and
 * vtkColorTransferFunction::GetTable are restated from their published algorithms; the call sites that
 * configure them are cited per function.  The stb parts (TGA writer, .hdr loader) ARE pinned: the reference
 * vendors stb_image.h v2.12 / stb_image_write.h v1.02 under utils/, and oracle/Makefile builds them, where
 * they lie, into oracle/_ref/libstb_ref.so for the tests to compare against.
 */
#ifndef SVR_IO_ORACLE_H
#define SVR_IO_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* element types of the raw file (same numbering as include/svr_io.h) */
enum { SVO_ELEM_I8 = 0, SVO_ELEM_U8, SVO_ELEM_I16, SVO_ELEM_U16, SVO_ELEM_I32, SVO_ELEM_U32, SVO_ELEM_F32, SVO_ELEM_F64 };

/* vtkImageCast -> short, ClampOverflow off (VolumeReader.cpp:41-45) */
void svo_cast_to_short(const void* src, int elem_type, size_t n, int16_t* dst);
/* vtkImageData::GetScalarRange (VolumeReader.cpp:54) */
void svo_scalar_range(const int16_t* v, size_t n, double range[2]);
/* VolumeReader::Rescale<short, unsigned short> (VolumeReader.cpp:55, 124-136) */
void svo_rescale(const int16_t* src, size_t n, float dataMin, float dataMax, uint16_t* dst);
/* vtkImageAccumulate as configured at VolumeReader.cpp:57-63; returns the number of bins */
int svo_histogram(const int16_t* v, size_t n, double rmin, double rmax, uint32_t* hist, int capacity);
/* vtkImageGradientMagnitude, 3-D, HandleBoundaries on, output type short; its range maximum
 * (VolumeReader.cpp:70-76) */
float svo_max_gradient_magnitude(const int16_t* v, int nx, int ny, int nz, const double spacing[3]);

/* vtkPiecewiseFunction::GetTable(0, 1, n, float*) (transferfunction.cpp:17): nodes = (x, y, midpoint, sharpness) */
void svo_piecewise_table(const double* nodes, int n_nodes, int clamping, int size, float* table);
/* vtkColorTransferFunction::GetTable(0, 1, n, float*) in RGB space, linear scale (transferfunction.cpp:18):
 * nodes = (x, r, g, b, midpoint, sharpness) */
void svo_color_table(const double* nodes, int n_nodes, int clamping, int size, float* table);

#ifdef __cplusplus
}
#endif
#endif
