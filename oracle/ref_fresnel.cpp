// ref_fresnel.cpp -- builds the one function of the reference's render path that compiles here without stand-ins:
// schlick_fresnel (core/bsdf/fresnel.h:10-15), whose header needs nothing but <cuda_runtime.h>.  A genuine
// cuda_runtime.h is in this image (shipped inside the triton wheel: triton/backends/nvidia/include); every other
// header of the path also needs GLM and/or curand_kernel.h, which are absent (and may not be faked).
// The reference header is included from where it lies under /root/reference; output goes to oracle/_ref/.
// Test infrastructure only.  Built with -ffp-contract=off like the oracle (the numeric contract of DESIGN.md 3).
#include "core/bsdf/fresnel.h"

extern "C" float ref_schlick_fresnel(float ni, float no, float cosin) { return schlick_fresnel(ni, no, cosin); }
