// Random 16-byte gathers from a large table: what the memory system of gfx950 gives the software trilinear sampler of the CELL
// layout (one global_load_dwordx4 per fetch, csrc/svr_walk.hpp tex_fetch<LAYOUT_CELL>) -- gathers per second and, under
// `rocprofv3 --pmc`, what FETCH_SIZE / TCC_EA0_RDREQ* / TCC_BUBBLE / TCC_HIT / TCC_MISS report per KNOWN request count
// (MI355X_MICROARCH.md, HBM: "other access widths are uncalibrated: calibrate on a known byte count in your own access pattern").
//
// usage: gather16 <table_MiB> <ilp 1|2|4|8> <gathers_per_lane> [blocks_per_cu=1] [pattern 0|1|2]
//   1024-thread blocks (16 waves per CU at blocks_per_cu = 1: the occupancy of the trace kernels)
//   pattern 0: every lane its own uniformly random 16-byte element (the walk of a fog-like medium)
//   pattern 1: the 64 lanes of a wave in ONE random 2-KB brick (8 x 4 x 4 elements: a coherent wave)
//   pattern 2: dependent chain -- the next index comes from the loaded value (latency of one gather)
//   pattern 3: 8 adjacent lanes read the 8 elements of ONE random 128-byte line (8 whole lines per wave instruction): if a 16-byte gather
//              that misses moved a whole line, lines per second would be the same as in pattern 0
//   pattern 4: 4 adjacent lanes read one random 64-byte half line (16 half lines per wave instruction)
// prints: requests, ms, G gathers/s, requested GB/s (16 B each)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorName(e), __LINE__); exit(1);} } while (0)

__device__ inline uint32_t wang(uint32_t a)
{
    a = (a ^ 61u) ^ (a >> 16); a = a + (a << 3); a = a ^ (a >> 4); a = a * 0x27d4eb2du; a = a ^ (a >> 15);
    return a;
}

__global__ void k_fill(uint4* t, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t h = wang((uint32_t)i);
        t[i] = make_uint4(h, wang(h), (uint32_t)i, (uint32_t)(i >> 32));
    }
}

template <int ILP, int PATTERN>
__global__ __launch_bounds__(1024) void k_gather16(const uint4* __restrict__ t, uint32_t n_elems, uint32_t iters, uint32_t* out)
{
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t x = wang(gid * 2654435761u + 12345u);
    uint32_t acc = 0;
    const uint32_t wave = gid >> 6, lane = gid & 63u;
    for (uint32_t i = 0; i < iters; i += ILP) {
        uint4 v[ILP];
#pragma unroll
        for (int j = 0; j < ILP; ++j) {
            uint32_t idx;
            if (PATTERN == 1) {
                const uint32_t b = wang(wave * 0x9e3779b9u + (i + j) * 0x85ebca6bu) % (n_elems >> 7);
                idx = (b << 7) + ((lane * 37u + i + j) & 127u);
            } else if (PATTERN == 3 || PATTERN == 4) {
                constexpr uint32_t G = PATTERN == 3 ? 8u : 4u;                       // lanes per group = elements per piece
                const uint32_t grp = gid / G;
                const uint32_t h = wang(grp * 0x9e3779b9u + (i + j) * 0x85ebca6bu + 77u);
                idx = (uint32_t)(((uint64_t)h * (n_elems / G)) >> 32) * G + (lane & (G - 1u));
            } else {
                x = x * 1664525u + 1013904223u;
                idx = (uint32_t)(((uint64_t)wang(x) * n_elems) >> 32);
            }
            v[j] = t[idx];
        }
#pragma unroll
        for (int j = 0; j < ILP; ++j) {
            acc ^= v[j].x + v[j].y + v[j].z + v[j].w;
            if (PATTERN == 2) x ^= v[j].x;                  // the next index depends on this load
        }
    }
    if (acc == 0x12345678u) out[gid & 1023u] = acc;       // (keeps the loads alive)
}

template <int ILP>
static void launch(int pattern, int blocks, const uint4* t, uint32_t n, uint32_t iters, uint32_t* out)
{
    if (pattern == 0) hipLaunchKernelGGL((k_gather16<ILP, 0>), dim3(blocks), dim3(1024), 0, 0, t, n, iters, out);
    else if (pattern == 1) hipLaunchKernelGGL((k_gather16<ILP, 1>), dim3(blocks), dim3(1024), 0, 0, t, n, iters, out);
    else if (pattern == 2) hipLaunchKernelGGL((k_gather16<ILP, 2>), dim3(blocks), dim3(1024), 0, 0, t, n, iters, out);
    else if (pattern == 3) hipLaunchKernelGGL((k_gather16<ILP, 3>), dim3(blocks), dim3(1024), 0, 0, t, n, iters, out);
    else hipLaunchKernelGGL((k_gather16<ILP, 4>), dim3(blocks), dim3(1024), 0, 0, t, n, iters, out);
}

int main(int argc, char** argv)
{
    const size_t mib = argc > 1 ? strtoull(argv[1], 0, 10) : 2200;
    const int ilp = argc > 2 ? atoi(argv[2]) : 1;
    const uint32_t iters = argc > 3 ? (uint32_t)atoi(argv[3]) : 512;
    const int bpc = argc > 4 ? atoi(argv[4]) : 1;
    const int pattern = argc > 5 ? atoi(argv[5]) : 0;
    const size_t n = (mib << 20) / 16;
    if (n >= (1ull << 32)) { printf("table too large for 32-bit element indices\n"); return 1; }
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int blocks = prop.multiProcessorCount * bpc;
    uint4* t; uint32_t* out;
    CHECK(hipMalloc(&t, n * 16)); CHECK(hipMalloc(&out, 4096));
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, t, n);
    CHECK(hipDeviceSynchronize());
    auto go = [&](uint32_t it) {
        switch (ilp) {
        case 1: launch<1>(pattern, blocks, t, (uint32_t)n, it, out); break;
        case 2: launch<2>(pattern, blocks, t, (uint32_t)n, it, out); break;
        case 4: launch<4>(pattern, blocks, t, (uint32_t)n, it, out); break;
        default: launch<8>(pattern, blocks, t, (uint32_t)n, it, out); break;
        }
    };
    go(8);                                                   // warm-up
    CHECK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipEventRecord(e0));
        go(iters);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
    }
    const double req = (double)blocks * 1024.0 * iters;
    printf("table %zu MiB  ilp %d  pattern %d  blocks %d x 1024  requests/launch %.0f  %.3f ms  %.2f G gathers/s  %.1f GB/s requested (16 B each)\n",
           mib, ilp, pattern, blocks, req, best, req / (best * 1e-3) / 1e9, req * 16.0 / (best * 1e-3) / 1e9);
    return 0;
}
