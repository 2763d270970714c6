#!/usr/bin/env python3
"""bench.py -- Msamples/s (paths x spp per second) of the path-tracing hot path on MI355X.

Workload (BASELINE.json metric: "Msamples/sec at 1024^2 on 512^3 volume"): config c3 = 512^3 CT-like volume, 1024^2
image, 3 area lights + environment map, GUI-default transfer function, trace depth 1 (the reference's default,
gui/canvas.cpp:17).  A *step* is one complete progressive render of the configuration: `--spp-per-step` samples for
every pixel (default 256 = c3's spp), issued as ONE svr_render_pathtracer_frames call = 4 trace launches of 64 frames,
bit-identical to 256 render_pathtracer calls (`--spp-per-step 1` is the reference's one-call-per-frame protocol).

`python bench.py --gpus N` with no WORLD_SIZE in the environment starts the N ranks itself (torch.distributed.run,
before this process touches the GPU); under an external launcher (RANK / WORLD_SIZE set) it is one of the ranks.  With
N > 1 the frame is sharded into interleaved 16-row strips (global per-pixel seeds: the assembled image is bit-identical
to one GPU's); EVERY step ends like a one-GPU step, in a tone-mapped image of the whole frame: the ranks' strips are
gathered onto rank 0 over RCCL (one collective per output = per step, sunvolumerender_amd.dist.FrameAssembler) and rank 0
tone-maps the assembled frame (svr_hdr_to_ldr_frame), all inside the timed region; the ranks skip the tone map of their own
strips (SVR_OPT_SKIP_TONEMAP), whose result nobody would read.

One JSON line on rank 0 (the task contract's fields) plus
  roofline      what bounds the dominant kernel (k_trace_tile).  The counters say vector-instruction issue, not HBM
                (profiles/): `bound` = "valu_issue", `achieved` = wave64 VALU instructions per second (instructions
                per launch from the committed rocprofv3 --pmc SQ_INSTS_VALU pass of the same workload / the HIP-event
                launch duration measured live), `peak` = 256 CUs x 4 SIMDs x 2.4 GHz / 2 cycles per wave64
                instruction (MI355X_MICROARCH.md).  The SURVEY 8(d) byte figure (16 B per volume tap of the REFERENCE
                algorithm + 24 B per pixel per launch) is kept as `algorithmic_*`; on the headline scene 95 % of those
                taps are proven transparent and never fetched, so `algorithmic_demand_frac` can exceed 1 and is not a
                fraction of anything -- `hbm_traffic_bytes` is what HBM really moved.
  workloads     the same measurement on the scenes where nothing can be skipped and algorithmic bytes = executed
                bytes: c3 with noisy, non-zero air (`c3n`), and c3 with empty-space skipping switched off; the headline
                scene in the opt-in fast-math mode, and at trace depth 2 and 4.
  summary       Msamples/s of every secondary workload, early in the line (details under `workloads`): noisy air, skipping
                off, depth 2 / 4, the opt-in fast-math mode, and the OPT-IN local-majorant mode (SVR_OPT_LOCAL_MAJORANT,
                "Woodcock max-density acceleration": not bit-identical, converged images agree) on c3, c3n and c5.
  cpu_baseline  the CPU oracle (a plain-C port of the reference's arithmetic, OpenMP) on a bounded sample: on the box's CPU
                share of one GPU (16 threads) and on all host cores (`value_all`, `cores_all`).
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0                       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
VALU_PEAK_GINST_S = 256 * 4 * 2.4 / 2.0     # wave64 VALU instructions per ns: 2 cycles each on a SIMD-32 -> 1228.8 G/s
GATHER_REQ_PEAK_G_S = 54.4                  # measured: random 16-byte gathers from a 2.2 GiB table, tools/ubench/gather16.hip (profiles/r04_fetch_size_calibration.txt)
METRIC = "Msamples/sec (paths x spp) at 1024^2 on 512^3 volume"


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--scene", default="c3")
    ap.add_argument("--trace-depth", type=int, default=1)
    ap.add_argument("--spp-per-step", type=int, default=256)
    ap.add_argument("--kernel", type=int, default=0, help="0 auto(=2), 1 block-per-tile baseline, 2 persistent tile kernel, 3 lane state machine")
    ap.add_argument("--layout", type=int, default=0, help="0 auto, 1 linear, 2 brick, 3 pair, 4 cell")
    ap.add_argument("--blocks-per-cu", type=int, default=0)
    ap.add_argument("--strip-rows", type=int, default=16, help="rows per interleaved strip under row sharding (16: max/mean rank time 1.02 at 8 ranks on c3, 32: 1.08)")
    ap.add_argument("--assemble", default="gather", choices=["gather", "reduce"], help="collective that assembles the frame on rank 0")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU-baseline budget (0 = skip)")
    ap.add_argument("--no-count", action="store_true", help="skip the tap-counting pass (algorithmic_* = null)")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary workloads (noisy air, skipping off, fast math, depth 2 / 4)")
    ap.add_argument("--extra-steps", type=int, default=2)
    ap.add_argument("--empty-skip", type=int, default=1)
    ap.add_argument("--fast-math", type=int, default=0)
    ap.add_argument("--set", action="append", default=[], metavar="OPTION=VALUE",
                    help="library option by name (abi.OPT_<NAME>), e.g. --set queue=0 --set park_end=8; experiments only")
    return ap.parse_args()


def launch_ranks(n: int) -> int:
    """Start the n ranks as children of this (GPU-free) process and pass rank 0's line through."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(Path(__file__).resolve()), *sys.argv[1:]]
    return subprocess.call(cmd, env=env)


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
    # every host core (north star: "timed on the host cores (core count stated)"): whole frames, so that each of the ~10^2 threads
    # has rows to work on (the 32-row bands above would leave most of them idle), a few seconds
    all_threads = int(o.lib.svo_max_threads())
    value_all = None
    if all_threads > threads and budget_s >= 5:
        hdr2 = o.new_hdr()
        o.render_pathtracer(hdr2, 0, trace_depth=trace_depth, count=False, nthreads=all_threads)      # (thread start-up)
        t2 = time.perf_counter()
        fr2 = 0
        while fr2 < 256:
            o.render_pathtracer(hdr2, fr2, trace_depth=trace_depth, count=False, nthreads=all_threads)
            fr2 += 1
            if time.perf_counter() - t2 > min(5.0, budget_s * 0.3):
                break
        value_all = round(W * H * fr2 / (time.perf_counter() - t2) / 1e6, 4)
    return {
        "value": round(pt_rate, 4),
        "unit": "Msamples/s",
        "cores": threads,
        "value_all": value_all,
        "cores_all": all_threads if value_all is not None else None,
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


def kernel_source_hash() -> str:
    """Hash of the kernel sources (comments and whitespace ignored): a committed PMC record made from other code is flagged as stale."""
    from sunvolumerender_amd._build import kernel_source_hash as h

    return h()


PMC_ROUND = "r04"


def pmc_record(tag: str, default_shape: bool = True):
    """profiles/r04_pmc_<tag>.json: per-launch averages of the rocprofv3 --pmc passes of this workload (tools/pmc_json.py).
    The records were taken at the default launch shape (64 frames per launch, automatic layout / queue / block count, no
    experiment option): a run with another shape gets no record, so its roofline fraction is null rather than wrong."""
    f = ROOT / "profiles" / f"{PMC_ROUND}_pmc_{tag}.json"
    if not default_shape or not f.exists():
        return None
    try:
        rec = json.loads(f.read_text())
        rec["_file"] = str(f.relative_to(ROOT))
        return rec
    except Exception:
        return None


class Workload:
    """One scene on one canvas: counting pass, warm-up, timed steps."""

    def __init__(self, dev, torch, name, trace_depth, layout):
        from sunvolumerender_amd import host, scenes

        self.dev, self.torch, self.name = dev, torch, name
        self.scene = scenes.make_scene(name, trace_depth=trace_depth)
        W, H = self.scene.width, self.scene.height
        self.hdr = torch.zeros(H * W * 3, dtype=torch.float32, device="cuda")
        self.img = torch.zeros(H * W, dtype=torch.int32, device="cuda")
        self.canvas = host.Canvas(dev, W, H, img_ptr=self.img.data_ptr(), hdr_ptr=self.hdr.data_ptr())
        scenes.apply_to_canvas(self.scene, self.canvas, layout)

    def step(self, S):
        if S == 1:
            self.canvas.paint()
        else:
            self.canvas.paint_frames(S)

    def count(self, steps, S):
        from sunvolumerender_amd import abi

        self.dev.set_option(abi.OPT_COUNT, 1)
        self.dev.reset_counters()
        self.canvas.ReStartRender()
        for _ in range(steps):
            self.step(S)
        self.torch.cuda.synchronize()
        c = self.dev.counters()
        self.dev.set_option(abi.OPT_COUNT, 0)
        return c

    def close(self):
        self.canvas.close()


def roofline_block(counters, S, steps, k_ms, k_n, pmc):
    """Flat roofline record for one timed workload of this rank (see the module docstring)."""
    roof = {"bound": "valu_issue", "achieved": None, "peak": round(VALU_PEAK_GINST_S, 1), "unit": "G wave64 VALU instr/s",
            "frac": None, "traffic": None, "kernel": "k_trace_tile", "kernel_avg_ms": None, "kernel_launches": k_n}
    if k_n <= 0:
        return roof
    avg_ms = k_ms / k_n
    roof["kernel_avg_ms"] = round(avg_ms, 4)
    if counters is not None:
        owned_px = counters["paths"] / (S * steps)
        # 16 B per volume tap of the algorithm + 24 B HDR read-modify-write per pixel per launch
        alg = (16.0 * counters["vol_taps"] + 24.0 * owned_px * k_n) / k_n
        exe = (16.0 * counters["vol_taps_executed"] + 24.0 * owned_px * k_n) / k_n
        roof.update({
            "algorithmic_bytes_per_launch": int(alg),
            "algorithmic_demand_gbs": round(alg / (avg_ms * 1e-3) / 1e9, 1),
            "algorithmic_demand_frac": round(alg / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "executed_bytes_per_launch": int(exe),
            "executed_demand_frac": round(exe / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "vol_taps_per_path": round(counters["vol_taps"] / max(1, counters["paths"]), 3),
            "executed_taps_per_path": round(counters["vol_taps_executed"] / max(1, counters["paths"]), 3),
        })
    if pmc is not None:
        stale = pmc.get("kernel_source_hash") not in (None, kernel_source_hash())
        roof.update({"pmc_source": pmc["_file"], "pmc_stale": bool(stale)})
        if pmc.get("valu_insts_per_launch"):
            ach = pmc["valu_insts_per_launch"] / (avg_ms * 1e-3) / 1e9
            roof.update({"achieved": round(ach, 1), "frac": round(ach / VALU_PEAK_GINST_S, 4),
                         "valu_insts_per_launch": int(pmc["valu_insts_per_launch"]),
                         "valu_peak_ginst_s": round(VALU_PEAK_GINST_S, 1),
                         "lane_utilisation": pmc.get("valu_lane_utilisation")})
        if pmc.get("read_requests_per_launch"):
            # the ceiling the software sampler's 16-byte gathers actually meet (profiles/r04_fetch_size_calibration.txt): ~54 G memory-side line
            # requests per second, whatever part of the 128-byte line is used and whether the table lives in HBM or the Infinity Cache
            rq = (pmc["read_requests_per_launch"] + pmc.get("WRITE_SIZE_KB_per_launch", 0.0) * 1024.0 / 128.0) / (avg_ms * 1e-3) / 1e9
            roof.update({"mem_line_requests_g_per_s": round(rq, 2), "gather_request_peak_g_per_s": GATHER_REQ_PEAK_G_S,
                         "gather_request_frac": round(rq / GATHER_REQ_PEAK_G_S, 4)})
        if pmc.get("traffic_bytes_per_launch"):
            t = pmc["traffic_bytes_per_launch"]
            roof.update({"traffic": int(t), "hbm_traffic_bytes": int(t), "hbm_peak_gbs": HBM_PEAK_GBS,
                         "hbm_frac_traffic": round(t / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)})
    return roof


ROOF_HEAD = ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_avg_ms", "lane_utilisation", "hbm_frac_traffic",
             "gather_request_frac", "executed_demand_frac", "pmc_source", "pmc_stale")


def compact_first(out: dict) -> dict:
    """The same record with what a reader needs FIRST: the contract's fields, the roofline's headline keys, the CPU baseline and
    one number per secondary workload (`summary`, Msamples/s, with the roofline fraction where a PMC record exists) -- the
    first ~2000 characters of the line are self-sufficient; the long blocks follow."""
    roof = out.get("roofline") or {}
    head = {k: out[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                                "dtype", "data") if k in out}
    cfg = dict(out.get("config") or {})
    head["config"] = {k: cfg[k] for k in ("workload", "math", "parallelism") if k in cfg}
    head["roofline"] = {k: roof[k] for k in ROOF_HEAD if k in roof}
    cb = out.get("cpu_baseline")
    head["cpu_baseline"] = None if cb is None else {k: cb[k] for k in ("value", "unit", "cores", "value_all", "cores_all", "kind", "sample") if k in cb}
    if out.get("workloads"):
        head["summary"] = {t: ([w["value"], w["roofline"].get("frac")] if w["roofline"].get("frac") is not None else w["value"])
                           for t, w in out["workloads"].items()}
        head["summary_unit"] = "Msamples/s (and roofline.frac of that workload's kernel where a PMC record exists)"
    head["config_detail"] = {k: v for k, v in cfg.items() if k not in head["config"]}
    head["roofline_detail"] = {k: v for k, v in roof.items() if k not in head["roofline"]}
    if cb is not None:
        head["cpu_baseline_detail"] = {k: v for k, v in cb.items() if k not in head["cpu_baseline"]}
    for k in ("counters", "workloads"):
        if k in out:
            head[k] = out[k]
    return head


def main():
    args = parse_args()
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        # no launcher around us: become the launcher.  Nothing in this process has touched the GPU yet.
        sys.exit(launch_ranks(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    world = int(env_world or "1")
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: WORLD_SIZE={world} but --gpus {args.gpus}; start it as `python bench.py --gpus N` "
                         f"or under torch.distributed.run with --nproc-per-node equal to --gpus")

    import numpy as np  # noqa: F401
    import torch
    import torch.distributed as tdist

    from sunvolumerender_amd import abi, dist, host

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (torch.cuda.is_available() is False)")
    # SVR_DIST_BACKEND=gloo lets several ranks share one GPU for rehearsals; the driver's runs use nccl (= RCCL)
    backend = os.environ.get("SVR_DIST_BACKEND", "nccl")
    local_rank = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            tdist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            tdist.init_process_group(backend, rank=rank, world_size=world)

    t_setup = time.perf_counter()
    dev = host.Device(local_rank, fatal_errors=False)
    stream = torch.cuda.current_stream()
    dev.check(dev.lib.svr_set_stream(C.c_void_p(stream.cuda_stream)))
    wl = Workload(dev, torch, args.scene, args.trace_depth, args.layout)
    scene, canvas = wl.scene, wl.canvas
    W, H = scene.width, scene.height
    dev.set_option(abi.OPT_KERNEL, args.kernel)
    dev.set_option(abi.OPT_EMPTY_SKIP, args.empty_skip)
    if args.fast_math:
        dev.set_option(abi.OPT_FAST_MATH, 1)
    if args.blocks_per_cu:
        dev.set_option(abi.OPT_BLOCKS_PER_CU, args.blocks_per_cu)
    for kv in args.set:
        name, value = kv.split("=")
        dev.set_option(getattr(abi, "OPT_" + name.upper()), int(value))
    # the committed PMC records describe the default launch shape only (see pmc_record)
    default_shape = not args.set and args.layout == 0 and args.blocks_per_cu == 0 and args.spp_per_step % 64 == 0
    asm = None
    if world > 1:
        dev.set_option(abi.OPT_SKIP_TONEMAP, 1)      # a rank's own strips are never shown: the assembled frame is tone-mapped on rank 0
        dist.shard(dev, args.strip_rows, rank, world)
        asm = dist.FrameAssembler(H, W, args.strip_rows, rank, world, dst=0, mode=args.assemble)
        frame_img = torch.zeros(H * W, dtype=torch.int32, device="cuda") if rank == 0 else None
    S = max(1, args.spp_per_step)
    t_setup = time.perf_counter() - t_setup

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            tdist.barrier()
        torch.cuda.synchronize()

    def assemble_and_tonemap(hdr):
        """The output of a sharded render: strips -> rank 0 (one collective), full-frame tone map on rank 0."""
        src = hdr if backend == "nccl" else hdr.cpu()
        frame = asm.assemble(src)
        if rank == 0:
            if frame.device.type != "cuda":
                frame = frame.cuda()
            dev.check(dev.lib.svr_hdr_to_ldr_frame(C.c_void_p(frame_img.data_ptr()), C.c_void_p(frame.data_ptr()), W, H))
        return frame

    # ---- tap-counting pass (untimed): exact algorithmic bytes of the frames about to be timed ----
    counters = None if args.no_count else wl.count(args.steps, S)

    # ---- warm-up (includes one assembly, so that its staging buffers and the RCCL channels exist) ----
    def full_step():
        """One step = S more samples for every pixel AND the output a one-GPU step ends in: a tone-mapped image of the whole
        frame (on one GPU the library's own tone map behind the last launch; on N GPUs gather + tone map on rank 0)."""
        wl.step(S)
        if world > 1:
            assemble_and_tonemap(wl.hdr)

    canvas.ReStartRender()
    for _ in range(max(args.warmup, 1 if world > 1 else 0)):     # (N > 1: at least one, so that the staging buffers and the RCCL channels exist)
        full_step()
    # ---- timed region: exactly K steps (frames 0 .. K*S-1 of a fresh progressive render) ----
    canvas.ReStartRender()
    dev.set_option(abi.OPT_TIMING, 1)
    dev.check(dev.lib.svr_reset_kernel_time())
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        full_step()
    barrier()
    elapsed = time.perf_counter() - t0
    dev.set_option(abi.OPT_TIMING, 0)
    k_ms, k_n = dev.kernel_time()

    el = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
    if world > 1:
        tdist.all_reduce(el, op=tdist.ReduceOp.MAX)
    elapsed = float(el.item())

    # local counters -> whole-job (every rank counts its own strips)
    cnt = None
    keys = ["paths", "vol_taps", "vol_taps_executed", "woodcock_iters", "scatter_events", "shadow_walks"]
    if counters is not None:
        tc = torch.tensor([counters[k] for k in keys], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        if world > 1:
            tdist.all_reduce(tc, op=tdist.ReduceOp.SUM)
        cnt = {k: int(v) for k, v in zip(keys, tc.tolist())}

    if rank == 0:
        samples = float(W) * H * S * args.steps
        value = samples / elapsed / 1e6
        variant = f"d{args.trace_depth}" + ("" if args.empty_skip else "_noskip") + ("_fast" if args.fast_math else "")
        pmc = pmc_record(f"{args.scene}_{variant}", default_shape) if (args.kernel in (0, 2) and world == 1) else None
        roof = roofline_block(counters, S, args.steps, k_ms, k_n, pmc)
        out = {
            "metric": METRIC,
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
                       "layout": {0: "auto", 1: "linear", 2: "brick", 3: "pair", 4: "cell"}[args.layout],
                       "math": "fast (v_log/v_exp/v_rcp, opt-in)" if args.fast_math else "bit-exact contract",
                       "parallelism": f"row-strip tiles x{world}, {args.strip_rows}-row strips; every step ends in {args.assemble} onto rank 0 + "
                                      f"full-frame tone map (inside the timed region)" if world > 1 else "single GPU",
                       "device": dev.info(), "setup_s": round(t_setup, 1)},
            "roofline": roof,
        }
        if asm is not None:
            out["config"]["assemble_bytes_per_peer"] = asm.bytes_sent_per_rank()
        if cnt is not None:
            out["counters"] = cnt

    # ---- secondary workloads (one GPU only) ----
    if world == 1 and not args.no_extra and args.scene == "c3" and args.kernel in (0, 2):
        extra = {}
        # the reference's own host protocol (gui/canvas.cpp:96-116): ONE render_pathtracer call + ONE device synchronisation per frame, frameNo++ --
        # what an unmodified host does.  Frames are traced ahead in batches (1, 2, 4 ... 64) on the library's streams and a call folds its frame;
        # the ramp (frames 0..63) and the steady state (frames 64..1087) are timed separately, the steady state is the figure of `summary`
        canvas.ReStartRender()
        for _ in range(200):
            canvas.paint(sync=True)
        canvas.ReStartRender()
        torch.cuda.synchronize()
        tp0 = time.perf_counter()
        for _ in range(64):
            canvas.paint(sync=True)
        torch.cuda.synchronize()
        tp1 = time.perf_counter()
        for _ in range(1024):
            canvas.paint(sync=True)
        torch.cuda.synchronize()
        tp2 = time.perf_counter()
        extra["c3_per_frame_calls"] = {"value": round(float(W) * H * 1024 / (tp2 - tp1) / 1e6, 3), "unit": "Msamples/s", "steps": 1024, "ms_per_step": round((tp2 - tp1) / 1024 * 1e3, 4),
                                       "trace_depth": args.trace_depth, "ramp_frames_0_63": round(float(W) * H * 64 / (tp1 - tp0) / 1e6, 3),
                                       "what": "the headline scene under the reference's host protocol: one render_pathtracer call + one svr_device_synchronize per frame "
                                               "(gui/canvas.cpp:96-116), frames 64..1087 of a progressive render (steady state of frame-ahead tracing; `ramp_frames_0_63` = the first 64 frames)",
                                       "roofline": {"frac": None}}
        # (the C ABI holds ONE scene per process, like the reference's __constant__ globals: the headline canvas goes
        # first, every further canvas replaces the scene and nothing is rendered on an earlier one afterwards)
        LM_WHAT = ("in the OPT-IN local-majorant mode (SVR_OPT_LOCAL_MAJORANT, 'Woodcock max-density acceleration': delta tracking against "
                   "per-macro-cell majorants; not bit-identical, converged images agree within Monte-Carlo noise)")
        d0 = args.trace_depth
        plan = [dict(tag="c3_skip_off", scene="c3", skip=0, what="the headline scene with SVR_OPT_EMPTY_SKIP = 0: every tap of the reference algorithm is fetched"),
                dict(tag="c3_fast_math", scene="c3", fast=1, what="the headline scene in the OPT-IN fast-math mode (SVR_OPT_FAST_MATH: v_log / reciprocal division / "
                                                                  "contraction; not bit-identical, converged images agree within Monte-Carlo noise)"),
                dict(tag="c3_local_majorant", scene="c3", lm=1, what="the headline scene " + LM_WHAT)]
        if d0 == 1 and not args.fast_math:
            for d in (2, 4):
                plan += [dict(tag=f"c3_depth{d}", scene="c3", depth=d, what=f"the headline scene at trace depth {d} (the reference's GUI range is 1-10)"),
                         dict(tag=f"c3_depth{d}_local_majorant", scene="c3", depth=d, lm=1, what=f"the headline scene at trace depth {d} " + LM_WHAT)]
        if d0 == 1 and not args.fast_math:
            plan += [dict(tag="c3_depth3_env_nee", scene="c3", depth=3, env_nee=1, what="the headline scene at trace depth 3 in the OPT-IN env-map importance sampling mode (SVR_OPT_ENV_NEE: "
                                                                                         "one direction drawn from the map's luminance per scatter event + balance heuristic; straight-line paths)")]
        plan += [dict(tag="c3_noisy_air", scene="c3n", what="c3 with noisy non-zero air (64..191 raw LSB, like CT data rescaled to the full u16 range): no "
                                                            "macro-cell is exactly transparent"),
                 dict(tag="c3n_local_majorant", scene="c3n", lm=1, what="c3 with noisy non-zero air " + LM_WHAT)]
        if d0 == 1 and not args.fast_math:
            plan += [dict(tag="c3n_depth4", scene="c3n", depth=4, what="c3 with noisy non-zero air at trace depth 4, default mode")]
        plan += [
                 dict(tag="c5", scene="c5", what="BASELINE config 5 on one GPU: 1024^3 u16 volume (HBM-resident), 1024^2, default mode"),
                 dict(tag="c5_local_majorant", scene="c5", lm=1, what="BASELINE config 5 on one GPU (1024^3 u16, 1024^2) " + LM_WHAT)]
        w2, w2_scene = wl, args.scene
        for item in plan:
            tag, sc_name, what = item["tag"], item["scene"], item["what"]
            skip, fast, lm, depth = item.get("skip", 1), item.get("fast", 0), item.get("lm", 0), item.get("depth", d0)
            env_nee = item.get("env_nee", 0)
            if sc_name != w2_scene:
                if w2 is not wl:
                    w2.close()
                w2, w2_scene = Workload(dev, torch, sc_name, d0, args.layout), sc_name
            dev.set_option(abi.OPT_EMPTY_SKIP, skip)
            dev.set_option(abi.OPT_FAST_MATH, fast)
            dev.set_option(abi.OPT_LOCAL_MAJORANT, lm)
            dev.set_option(abi.OPT_ENV_NEE, env_nee)
            w2.canvas.SetScatterTimes(depth)
            n = max(1, args.extra_steps)
            c2 = None if (args.no_count or fast or env_nee) else w2.count(n, S)
            w2.canvas.ReStartRender()
            w2.step(S)
            w2.canvas.ReStartRender()
            dev.set_option(abi.OPT_TIMING, 1)
            dev.check(dev.lib.svr_reset_kernel_time())
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(n):
                w2.step(S)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t1
            dev.set_option(abi.OPT_TIMING, 0)
            ms2, n2 = dev.kernel_time()
            rec = None if (fast or env_nee) else pmc_record(f"{sc_name}_d{depth}" + ("" if skip else "_noskip") + ("_lm" if lm else ""), default_shape)
            r2 = roofline_block(c2, S, n, ms2, n2, rec)
            if lm:
                r2["kernel"] = "k_trace_lm"
            extra[tag] = {"value": round(float(W) * H * S * n / dt / 1e6, 3), "unit": "Msamples/s", "steps": n,
                          "ms_per_step": round(dt / n * 1e3, 3), "trace_depth": depth, "what": what, "roofline": r2}
            dev.set_option(abi.OPT_EMPTY_SKIP, args.empty_skip)
            dev.set_option(abi.OPT_FAST_MATH, args.fast_math)
            dev.set_option(abi.OPT_LOCAL_MAJORANT, 0)
            dev.set_option(abi.OPT_ENV_NEE, 0)
            w2.canvas.SetScatterTimes(d0)
        if w2 is not wl:
            w2.close()
        out["workloads"] = extra

    if rank == 0:
        if args.cpu_seconds > 0 and world == 1:
            out["cpu_baseline"] = cpu_baseline(scene, args.trace_depth, args.cpu_seconds)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(compact_first(out)), flush=True)

    wl.close()
    if world > 1:
        tdist.barrier()
        tdist.destroy_process_group()


if __name__ == "__main__":
    main()
