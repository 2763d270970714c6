// svr_trace_tile_fast.hip -- the OPT-IN fast-math build of the tile kernel (SVR_OPT_FAST_MATH, default off).
// The same source as svr_trace_tile.hip, compiled a second time into namespace svr_fast with approximate hardware
// transcendentals (svr_math.hpp, SVR_FAST_MATH) and the compiler's fast-math flags (sunvolumerender_amd/_build.py) --
// the counterpart of the reference's nvcc -use_fast_math (CMakeLists.txt:9-10).  Not bit-identical to the oracle:
// validated on converged images (tests/test_fast_math_gpu.py).
#define SVR_FAST_MATH 1
#define svr svr_fast
#include "svr_trace_tile.hip"
#undef svr

namespace svr_fast {
// type-erased entry for svr_api.hip (its DevScene / DevWork / LaunchCfg are the layout-identical types of namespace svr)
hipError_t launch_trace_tile_raw(const void* scene, const void* work, const void* cfg, hipStream_t st)
{
    return launch_trace_tile(*static_cast<const DevScene*>(scene), *static_cast<const DevWork*>(work), *static_cast<const LaunchCfg*>(cfg), st);
}
}
