// ref_wanghash.cpp -- builds the reference's own wangHash (pathtracer.cu:70-79), the per-frame seed hash of
// render_pathtracer (pathtracer.cu:302).  pathtracer.cu as a whole needs GLM and curand_kernel.h (absent here, and not to be
// faked), but this function needs nothing beyond <cuda_runtime.h> (__host__ __device__; the genuine header ships inside the
// triton wheel) and <stdint.h>.  The Makefile cuts exactly those lines out of the file WHERE IT LIES into a temporary
// include (REF_WANGHASH_INC, deleted after the compile; nothing of the reference is kept in the repo or in oracle/_ref/).
// Test infrastructure only.
#include <stdint.h>
#include <cuda_runtime.h>
#include REF_WANGHASH_INC

extern "C" uint32_t ref_wang_hash(uint32_t a) { return wangHash(a); }
