#!/usr/bin/env python3
"""bench.py -- Msamples/s (paths x spp per second) of the path-tracing hot path on MI355X.

Workload (BASELINE.json metric: "Msamples/sec at 1024^2 on 512^3 volume"): config c3 = 512^3 CT-like
volume, 1024^2 image, 3 area lights + environment map, GUI-default transfer function, trace depth 1
(the reference's default, gui/canvas.cpp:17).  A *step* is one progressive-render pass over the whole
frame: `--spp-per-step` samples for every pixel (default 32 = one frame group = one launch of the trace
kernel through svr_render_pathtracer_frames, bit-identical to 32 render_pathtracer calls; the default
8 steps are config c3's 256 spp; `--spp-per-step 1` is the reference's one-call-per-frame protocol).  With N GPUs the frame is sharded into interleaved 32-row strips
(global per-pixel seeds, so the assembled image is bit-identical to one GPU) and the HDR accumulation
buffers are summed onto rank 0 with one RCCL reduce per output, inside the timed region.

One JSON line on rank 0; see the task contract for the field meanings.  `roofline` prices the
path-tracing kernel by algorithmic bytes (16 B per volume tap + 24 B HDR read-modify-write per pixel
per launch) over its HIP-event duration; `cpu_baseline` times the CPU oracle (a plain-C
port of the reference's arithmetic, OpenMP over rows) on a bounded sample of the same workload.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--scene", default="c3")
    ap.add_argument("--trace-depth", type=int, default=1)
    ap.add_argument("--spp-per-step", type=int, default=32)
    ap.add_argument("--kernel", type=int, default=0, help="0 auto(=2), 1 block-per-tile baseline, 2 persistent tile kernel, 3 lane state machine")
    ap.add_argument("--layout", type=int, default=0, help="0 auto, 1 linear, 2 brick")
    ap.add_argument("--blocks-per-cu", type=int, default=0)
    ap.add_argument("--strip-rows", type=int, default=16, help="rows per interleaved strip under row sharding (16: max/mean rank time 1.02 at 8 ranks on c3, 32: 1.08)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU-baseline budget (0 = skip)")
    ap.add_argument("--no-count", action="store_true", help="skip the tap-counting pass (roofline.achieved = null)")
    return ap.parse_args()


def cpu_baseline(scene, trace_depth, budget_s):
    """Oracle (oracle/, a plain-C port of the reference's arithmetic, OpenMP) timed on the host cores over a
    bounded, image-representative sample of the same workload: 32-row bands every 128 rows, as many
    progressive frames as fit the budget.  Threads: the box's CPU share for one GPU (16) unless
    SVR_CPU_THREADS says otherwise."""
    from oracle import binding

    o = binding.OracleScene(scene)
    threads = int(os.environ.get("SVR_CPU_THREADS", "0")) or min(16, o.lib.svo_max_threads())
    W, H = scene.width, scene.height
    hdr = o.new_hdr()
    bands = [(0, y, W, min(H, y + 32)) for y in range(48 if H > 96 else 0, H, 128)]
    band_px = sum((y1 - y0) * (x1 - x0) for x0, y0, x1, y1 in bands)
    t0 = time.perf_counter()
    frames = 0
    while frames < 256:
        for w in bands:
            o.render_pathtracer(hdr, frames, trace_depth=trace_depth, window=w, count=False, nthreads=threads)
        frames += 1
        if time.perf_counter() - t0 > budget_s * 0.6:
            break
    dt = time.perf_counter() - t0
    pt_rate = band_px * frames / dt / 1e6
    # the ray caster (raycasting.cu arithmetic, the reference's other render mode): 8-row bands
    t1 = time.perf_counter()
    rc_px = 0
    for y in range(52 if H > 96 else 0, H, 128):
        o.render_raycasting(window=(0, y, W, min(H, y + 8)), count=False, nthreads=threads)
        rc_px += W * (min(H, y + 8) - y)
        if time.perf_counter() - t1 > budget_s * 0.4:
            break
    rc_dt = time.perf_counter() - t1
    return {
        "value": round(pt_rate, 4),
        "unit": "Msamples/s",
        "cores": threads,
        "kind": "port",
        "sample": f"oracle path tracer, {frames} progressive frame(s) of {len(bands)} bands of 32 rows "
                  f"(every 128 rows) of the {W}x{H} frame = {band_px * frames} paths in {dt:.1f} s",
        "raycasting_mpix_s": round(rc_px / rc_dt / 1e6, 5),
        "raycasting_sample": f"oracle ray caster, {rc_px} pixels (8-row bands every 128 rows) in {rc_dt:.1f} s",
        "host_cores": os.cpu_count(),
        "cpu_model": _cpu_model(),
    }


def _cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        print(f"warning: WORLD_SIZE={world} != --gpus {args.gpus}", file=sys.stderr)

    import numpy as np
    import torch
    import torch.distributed as dist

    from sunvolumerender_amd import abi, host, scenes

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (torch.cuda.is_available() is False)")
    # SVR_DIST_BACKEND=gloo lets several ranks share one GPU for rehearsals; the driver's runs use nccl (= RCCL)
    backend = os.environ.get("SVR_DIST_BACKEND", "nccl")
    local_rank = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    t_setup = time.perf_counter()
    scene = scenes.make_scene(args.scene, trace_depth=args.trace_depth)
    W, H = scene.width, scene.height
    dev = host.Device(local_rank, fatal_errors=False)
    stream = torch.cuda.current_stream()
    dev.check(dev.lib.svr_set_stream(C.c_void_p(stream.cuda_stream)))
    hdr = torch.zeros(H * W * 3, dtype=torch.float32, device="cuda")
    img = torch.zeros(H * W, dtype=torch.int32, device="cuda")
    canvas = host.Canvas(dev, W, H, img_ptr=img.data_ptr(), hdr_ptr=hdr.data_ptr())
    scenes.apply_to_canvas(scene, canvas, args.layout)
    dev.set_option(abi.OPT_KERNEL, args.kernel)
    if args.blocks_per_cu:
        dev.set_option(abi.OPT_BLOCKS_PER_CU, args.blocks_per_cu)
    if world > 1:
        dev.check(dev.lib.svr_set_row_shard(args.strip_rows, rank, world))
    S = max(1, args.spp_per_step)
    t_setup = time.perf_counter() - t_setup

    def step():
        if S == 1:
            canvas.paint()
        else:
            canvas.paint_frames(S)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- tap-counting pass (untimed): exact algorithmic bytes of the frames about to be timed ----
    counters = None
    if not args.no_count:
        dev.set_option(abi.OPT_COUNT, 1)
        dev.reset_counters()
        canvas.ReStartRender()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        counters = dev.counters()
        dev.set_option(abi.OPT_COUNT, 0)

    # ---- warm-up ----
    canvas.ReStartRender()
    for _ in range(args.warmup):
        step()
    # ---- timed region: exactly K steps (frames 0 .. K*S-1 of a fresh progressive render) ----
    canvas.ReStartRender()
    dev.set_option(abi.OPT_TIMING, 1)
    dev.check(dev.lib.svr_reset_kernel_time())
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    if world > 1:
        # one collective per output: strips are disjoint and every rank's buffer is zero outside its own
        if backend == "nccl":
            dist.reduce(hdr, dst=0, op=dist.ReduceOp.SUM)
        else:
            h = hdr.cpu()
            dist.reduce(h, dst=0, op=dist.ReduceOp.SUM)
            hdr.copy_(h)
    barrier()
    elapsed = time.perf_counter() - t0
    dev.set_option(abi.OPT_TIMING, 0)
    k_ms, k_n = dev.kernel_time()

    el = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())

    # local counters -> whole-job (every rank counts its own strips)
    cnt = None
    if counters is not None:
        keys = ["paths", "vol_taps", "vol_taps_executed", "woodcock_iters", "scatter_events", "shadow_walks"]
        tc = torch.tensor([counters[k] for k in keys], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        local = {k: counters[k] for k in keys}
        if world > 1:
            dist.all_reduce(tc, op=dist.ReduceOp.SUM)
        cnt = {k: int(v) for k, v in zip(keys, tc.tolist())}
        cnt["local"] = local

    if rank == 0:
        samples = float(W) * H * S * args.steps
        value = samples / elapsed / 1e6
        roof = {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None}
        if cnt is not None and k_n > 0:
            loc = cnt["local"]
            n_launch = k_n
            owned_px = loc["paths"] / (S * args.steps)
            # 16 B per volume tap of the algorithm + 24 B HDR read-modify-write per pixel per launch
            bytes_per_launch = (16.0 * loc["vol_taps"] + 24.0 * owned_px * args.steps) / n_launch
            avg_ms = k_ms / k_n
            ach = bytes_per_launch / (avg_ms * 1e-3) / 1e9
            agg = bytes_per_launch * n_launch / elapsed / 1e9
            roof.update({"achieved": round(ach, 2), "frac": round(ach / HBM_PEAK_GBS, 5),
                         "achieved_aggregate": round(agg, 2), "frac_aggregate": round(agg / HBM_PEAK_GBS, 5),
                         "note": "achieved = algorithmic bytes of the reference algorithm per launch (16 B per volume "
                                 "tap it would fetch + 24 B HDR per pixel-frame) / mean HIP-event duration of one launch. "
                                 "The kernel does not fetch the taps of provably transparent macro-cells (bit-exact "
                                 "empty-space skipping), so frac can exceed 1: `traffic` is what HBM really moved. The "
                                 "kernel is VALU-issue bound (profiles/r01_bench_default_prof_summary.txt), not HBM bound",
                         "executed_taps_per_path": round(loc["vol_taps_executed"] / max(1, loc["paths"]), 3),
                         "kernel": {0: "k_trace_tile", 1: "k_pathtrace_pixel", 2: "k_trace_tile", 3: "k_pathtrace_uloop"}[args.kernel],
                         "kernel_avg_ms": round(avg_ms, 4), "kernel_launches": k_n,
                         "algorithmic_bytes_per_launch": int(bytes_per_launch),
                         "vol_taps_per_path": round(loc["vol_taps"] / max(1, loc["paths"]), 3)})
        # HBM traffic per launch comes from separate rocprofv3 --pmc passes (committed under profiles/); it cannot be
        # measured from inside this process
        tfile = ROOT / "profiles" / f"r01_traffic_k_trace_tile_{args.scene}_s{S}.json"
        if args.kernel in (0, 2) and world == 1 and tfile.exists():
            try:
                trec = json.loads(tfile.read_text())
                roof["traffic"] = trec["traffic_bytes_per_launch"]
                if trec.get("valu_insts_per_launch") and roof.get("kernel_avg_ms"):
                    # the bound this kernel really runs against: vector-instruction issue.  A wave64 VALU instruction
                    # takes 2 cycles of a SIMD-32 (MI355X_MICROARCH.md), 256 CUs x 4 SIMDs at 2.4 GHz
                    peak = 256 * 4 * 2.4e9 / 2.0
                    ach = trec["valu_insts_per_launch"] / (roof["kernel_avg_ms"] * 1e-3)
                    roof["valu_issue"] = {"achieved": round(ach / 1e9, 1), "peak": round(peak / 1e9, 1), "unit": "G wave64 instr/s",
                                          "frac": round(ach / peak, 4), "lane_utilisation": trec.get("valu_lane_utilisation"),
                                          "insts_per_launch": trec["valu_insts_per_launch"],
                                          "note": "instruction count from the committed rocprofv3 --pmc SQ_INSTS_VALU pass"}
                roof["traffic_source"] = str(tfile.relative_to(ROOT))
                if roof.get("kernel_avg_ms"):
                    # the same fraction on the bytes HBM really moved ("quote the fraction both ways")
                    roof["frac_traffic"] = round(roof["traffic"] / (roof["kernel_avg_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)
            except Exception:
                pass
        out = {
            "metric": "Msamples/sec (paths x spp) at 1024^2 on 512^3 volume",
            "value": round(value, 3),
            "unit": "Msamples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{args.scene}: {scene.dim[0]}^3 u16 CT-like volume, {W}x{H} image, "
                                   f"{len(scene.lights)} area lights, env map {'on' if scene.env_on_escape else 'off'}, "
                                   f"trace depth {args.trace_depth}, {S} spp per step",
                       "spp_per_step": S, "trace_depth": args.trace_depth,
                       "kernel": {0: "tile", 1: "pixel", 2: "tile", 3: "uloop"}[args.kernel],
                       "layout": {0: "auto(brick)", 1: "linear", 2: "brick"}[args.layout],
                       "parallelism": f"row-strip tiles x{world}" if world > 1 else "single GPU",
                       "device": dev.info(), "setup_s": round(t_setup, 1)},
            "roofline": roof,
        }
        if cnt is not None:
            out["counters"] = {k: v for k, v in cnt.items() if k != "local"}
        if args.cpu_seconds > 0:
            out["cpu_baseline"] = cpu_baseline(scene, args.trace_depth, args.cpu_seconds)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)

    canvas.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
