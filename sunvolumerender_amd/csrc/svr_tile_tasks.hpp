// svr_tile_tasks.hpp -- what the persistent tile kernels (svr_trace_tile.hip, svr_trace_lm.hip) share: the order in which a
// launch's wave-tasks are handed out, and the running mean of a pixel's frames inside the kernel (fold_pending).
#pragma once
#include "svr_kernel_common.hpp"

namespace svr {

// task index of the centre-out order -> tile row, tile column, frame group (see k_trace_tile)
struct TaskShape { uint32_t tiles_x, tiles_y, fgroups, row_tasks, c_row, tw2, th2, P2, fl2, wv; };
SVR_DEV TaskShape task_shape(const DevWork& w)
{
    TaskShape ts;
    ts.wv = w.x1 - w.x0;
    ts.fl2 = w.frames_log2;                                   // 0..6
    ts.P2 = 6u - ts.fl2;                                      // log2(pixels per wave)
    ts.tw2 = (ts.P2 + 1u) >> 1; ts.th2 = ts.P2 >> 1;          // pixel block 8x8, 8x4, 4x4, 4x2, 2x2, 2x1, 1x1
    ts.tiles_x = (ts.wv + (1u << ts.tw2) - 1u) >> ts.tw2;
    ts.tiles_y = (w.n_rows + (1u << ts.th2) - 1u) >> ts.th2;
    ts.fgroups = (w.nframes + (1u << ts.fl2) - 1u) >> ts.fl2;
    ts.row_tasks = ts.tiles_x * ts.fgroups;                   // tasks of one tile row
    ts.c_row = ts.tiles_y >> 1;
    return ts;
}
SVR_DEV void task_decode(const TaskShape& ts, uint32_t k, uint32_t& tx, uint32_t& ty, uint32_t& fg)
{
    const uint32_t rr = k / ts.row_tasks, in_row = k - rr * ts.row_tasks;
    const uint32_t off = (rr + 1u) >> 1;
    ty = (rr & 1u) ? ts.c_row - off : ts.c_row + off;
    tx = in_row / ts.fgroups;
    fg = in_row - tx * ts.fgroups;
}

// running_estimate (pathtracer.cu:81-84,279) for the pending tasks of this wave: the 1 << fl2 lanes of a task that
// hold one pixel are that pixel's frames frame0 .. frame0 + nframes - 1 IN ORDER, so one lane per (pixel, channel)
// replays the reference's sequence acc += (L - acc) / (n + 1) frame by frame -- the same float operations in the same
// order as nframes calls of the reference -- and the accumulator is read and written once per launch (12 B per
// pixel) instead of once per frame.  clear_hdr_buffer (pathtracer.cu:86-94) is the frame0 == 0 case.
// rows: [task * 3 + channel] rows of `row` floats (this wave's), tasks: their task numbers
SVR_DEV void fold_pending(const DevScene& s, const DevWork& w, const float* rows, uint32_t row, const uint32_t* tasks, uint32_t npend)
{
    const TaskShape ts = task_shape(w);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t npx = 1u << ts.P2;
    const uint32_t items = npend * npx * 3u;
    const uint32_t nfr = min(w.nframes, 1u << ts.fl2);
    for (uint32_t i = lane; i < items; i += 64u) {
        const uint32_t pi = (i * 0xAAABu) >> 17;           // i / 3 for i < 2^15
        const uint32_t ch = i - 3u * pi;
        const uint32_t q = pi >> ts.P2, pl = pi & (npx - 1u);
        uint32_t tx, ty, fg;
        task_decode(ts, tasks[q], tx, ty, fg);
        const uint32_t px = (tx << ts.tw2) + (pl & ((1u << ts.tw2) - 1u));
        const uint32_t r = (ty << ts.th2) + (pl >> ts.tw2);
        if (px >= ts.wv || r >= w.n_rows) continue;
        const uint32_t x = w.x0 + px, y = owned_row_to_y(w, r);
        float* h = w.hdr + 3 * ((size_t)y * s.imageW + x) + ch;
        float acc = (w.frame0 == 0u) ? 0.f : *h;
        const float* rp = rows + (size_t)(q * 3u + ch) * row + pl;
        if (w.nan_guard) {
            // SVR_OPT_NAN_GUARD: a non-finite sample (the reference's 0/0, pathtracer.cu:106-131) is replaced by the running mean
            for (uint32_t f = 0; f < nfr; ++f) {
                float Lf = rp[f << ts.P2];
                Lf = (f2u(Lf) & 0x7f800000u) == 0x7f800000u ? acc : Lf;
                const float n1 = (float)(w.frame0 + f) + 1.f;
                acc = acc + (Lf - acc) / n1;
            }
        } else {
            for (uint32_t f = 0; f < nfr; ++f) {
                const float Lf = rp[f << ts.P2];
                const float n1 = (float)(w.frame0 + f) + 1.f;
                acc = acc + (Lf - acc) / n1;
            }
        }
        *h = acc;
    }
}

// Launches that do NOT fold (frames traced ahead of the calls that ask for them, svr_api.hip: k_resolve / k_mean_flat fold one scratch slot
// per call): the radiance of the path with id = (pending task << 6 | lane) of this wave goes straight to lbuf[frame slot][pixel].
// tasks: the wave's pending task numbers (LDS).
SVR_DEV void direct_put(const DevScene& s, const DevWork& w, const uint32_t* tasks, uint32_t id, v3 L)
{
    const TaskShape ts = task_shape(w);
    uint32_t tx, ty, fg;
    task_decode(ts, tasks[id >> 6], tx, ty, fg);
    const uint32_t ln = id & 63u, pl = ln & ((1u << ts.P2) - 1u), fs = ln >> ts.P2;
    const uint32_t px = (tx << ts.tw2) + (pl & ((1u << ts.tw2) - 1u));
    const uint32_t r = (ty << ts.th2) + (pl >> ts.tw2);
    const uint32_t x = w.x0 + px, y = owned_row_to_y(w, r), slot = (fg << ts.fl2) + fs;
    float* o = w.lbuf + (size_t)slot * w.slot_stride + 3 * ((size_t)y * s.imageW + x);
    o[0] = L.x; o[1] = L.y; o[2] = L.z;
}

} // namespace svr
