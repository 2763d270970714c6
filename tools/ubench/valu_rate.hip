// VALU issue-rate microbenchmark for gfx950: wave64 instructions per cycle per SIMD for the instruction mixes
// the Woodcock loop uses (int xor/shift/add, f32 fma, cndmask), at 1..8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorName(e), __LINE__); exit(1);} } while (0)

template <int MODE>
__global__ void k(unsigned* out, int iters)
{
    unsigned a0 = threadIdx.x, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3, a4 = a0 * 11 + 4, a5 = a0 * 13 + 5, a6 = a0 * 17 + 6, a7 = a0 * 19 + 7;
    float f0 = a0 * 1e-3f, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3, f4 = f0 + 4, f5 = f0 + 5, f6 = f0 + 6, f7 = f0 + 7;
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {          // 8 independent int chains, 2 ops each
#define STEP(a) a = (a ^ (a >> 2)) + 0x9e3779b9u;
            STEP(a0) STEP(a1) STEP(a2) STEP(a3) STEP(a4) STEP(a5) STEP(a6) STEP(a7)
#undef STEP
        } else if (MODE == 1) {   // 8 independent fma chains
#define STEP(f) f = __builtin_fmaf(f, 1.0001f, 0.5f);
            STEP(f0) STEP(f1) STEP(f2) STEP(f3) STEP(f4) STEP(f5) STEP(f6) STEP(f7)
#undef STEP
        } else if (MODE == 2) {   // ONE dependent fma chain (latency-bound per wave)
            f0 = __builtin_fmaf(f0, 1.0001f, 0.5f); f0 = __builtin_fmaf(f0, 1.0001f, 0.5f); f0 = __builtin_fmaf(f0, 1.0001f, 0.5f); f0 = __builtin_fmaf(f0, 1.0001f, 0.5f);
            f0 = __builtin_fmaf(f0, 1.0001f, 0.5f); f0 = __builtin_fmaf(f0, 1.0001f, 0.5f); f0 = __builtin_fmaf(f0, 1.0001f, 0.5f); f0 = __builtin_fmaf(f0, 1.0001f, 0.5f);
        } else {                  // one dependent int chain
            a0 = (a0 ^ (a0 >> 2)) + 0x9e3779b9u; a0 = (a0 ^ (a0 >> 2)) + 0x9e3779b9u; a0 = (a0 ^ (a0 >> 2)) + 0x9e3779b9u; a0 = (a0 ^ (a0 >> 2)) + 0x9e3779b9u;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (unsigned)(f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7);
}

template <int MODE>
void run(const char* name, int inst_per_iter, unsigned* d)
{
    const int iters = 20000;
    for (int wps = 1; wps <= 8; wps *= 2) {
        int blocks = 256 * wps;            // 256-thread blocks = 4 waves = 1 wave per SIMD per block
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 100);
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        double wave_instr = (double)blocks * 4 * iters * inst_per_iter;
        double per_simd_per_s = wave_instr / 1024.0 / (ms * 1e-3);
        printf("%-28s waves/SIMD=%d  %8.3f ms  %.3f G wave-instr/s/SIMD  (= %.2f cycles/instr at 2.4 GHz)\n", name, wps, ms, per_simd_per_s / 1e9, 2.4e9 / per_simd_per_s);
    }
}

int main()
{
    unsigned* d; CHECK(hipMalloc(&d, 256 * 8 * 256 * 4));
    run<0>("int (xor,shift,add) x8 indep", 24, d);
    run<1>("fma x8 indep", 8, d);
    run<2>("fma x8 dependent", 8, d);
    run<3>("int x12 dependent", 12, d);
    return 0;
}
