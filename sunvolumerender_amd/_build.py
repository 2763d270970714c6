"""Build recipe for libsvr_hip.so (hipcc, gfx950 only).

The numeric contract is part of the flags: no floating-point
contraction, no fast-math, correctly rounded divide/sqrt.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from pathlib import Path

PKG_DIR = Path(__file__).resolve().parent
REPO_ROOT = PKG_DIR.parent
CSRC = PKG_DIR / "csrc"
LIB_DIR = PKG_DIR / "lib"
LIB_PATH = Path(os.environ["SVR_HIP_LIB"]) if os.environ.get("SVR_HIP_LIB") else LIB_DIR / "libsvr_hip.so"

HIP_SOURCES = ["svr_api.hip", "svr_kernels.hip", "svr_trace_tile.hip", "svr_wavefront.hip", "svr_accel.hip", "svr_raycast.hip", "svr_host_io.hip", "svr_volume_prep.hip", "svr_selftest.hip", "svr_trace_tile_fast.hip", "svr_trace_lm.hip", "svr_trace_split.hip", "svr_trace_env.hip"]
# svr_trace_tile_fast.hip (the opt-in fast-math build) includes svr_trace_tile.hip and is compiled WITHOUT the contract flags
FAST_SOURCES = {"svr_trace_tile_fast.hip"}
CONTRACT_FLAGS = {"-ffp-contract=off", "-fno-fast-math", "-fhip-fp32-correctly-rounded-divide-sqrt"}
# (full -ffast-math, and hardware exp / sin / cos in place of the polynomial versions, cost 50-60 spilled VGPRs in this kernel
# and made it 38 % SLOWER; the walk's logarithm + reciprocal division + contraction keep the register allocation)
FAST_FLAGS = ["-ffp-contract=fast", "-freciprocal-math", "-fno-signed-zeros"]
HIP_HEADERS = ["svr_trace_tile.hip", "svr_math.hpp", "svr_scene.hpp", "svr_device.hpp", "svr_kernels.hpp", "svr_kernel_common.hpp", "svr_walk.hpp", "svr_lanes.hpp", "svr_tile_tasks.hpp", "svr_primary.hpp"]

HIPCC_FLAGS = [
    "-O3",
    "--offload-arch=gfx950",
    "-std=c++17",
    "-fPIC",
    "-ffp-contract=off",
    "-fno-fast-math",
    "-fhip-fp32-correctly-rounded-divide-sqrt",
    "-Wall",
    "-Wno-unused-value",
    "-fno-slp-vectorize",       # SLP packs adjacent f32 operations into v_pk_* instructions: an anti-lever on gfx950 (MI355X_MICROARCH.md); +2.6 % headline, +7 % at depth 4 (same-box A/B)
    "-I" + str(REPO_ROOT / "include"),
]
# extra compile flags for experiment builds (e.g. -DSVR_TEST_HOOKS: the timing-ablation options of tools/exp.py)
HIPCC_FLAGS += os.environ.get("SVR_EXTRA_HIPCC_FLAGS", "").split()
LINK_LIBS = ["-lz", "-ldl"]  # MetaImage CompressedData (svr_host_io.hip); dlopen of the process's RCCL (svr_assemble_frame)


HOST_ONLY_SOURCES = ("svr_host_io.hip", "svr_internal.hpp")     # (svr_api.hip holds the launch policy: queue / layout / group heuristics)


def kernel_source_hash() -> str:
    """Hash of the kernel sources (device code and the launch policy of svr_api.hip: everything but the file I/O translation unit) with comments
    and whitespace removed: the PMC records under profiles/ carry it, and bench.py flags a record made from other code as stale."""
    import hashlib
    import re

    h = hashlib.sha1()
    for f in sorted(CSRC.glob("*")):
        if f.suffix in (".hip", ".hpp") and f.name not in HOST_ONLY_SOURCES:
            text = f.read_text()
            text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
            text = re.sub(r"//[^\n]*", "", text)
            h.update(re.sub(r"\s+", "", text).encode())
    h.update(" ".join(f for f in HIPCC_FLAGS if not f.startswith("-I")).encode())      # the code generation flags are part of the code
    return h.hexdigest()[:12]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found (set HIPCC=/path/to/hipcc)")


def _stale(target: Path, deps) -> bool:
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any(Path(d).stat().st_mtime > t for d in deps)


def build_hip(force: bool = False, verbose: bool = False) -> Path:
    """Compile the HIP kernels + C-ABI into sunvolumerender_amd/lib/libsvr_hip.so: one object per source
    (compiled in parallel, only when stale), then one link."""
    from concurrent.futures import ThreadPoolExecutor

    LIB_DIR.mkdir(exist_ok=True)
    obj_dir = LIB_DIR / ("obj" if LIB_PATH.name == "libsvr_hip.so" else "obj_" + LIB_PATH.stem)   # variant builds (SVR_HIP_LIB) keep their own objects
    obj_dir.mkdir(exist_ok=True)
    common = [CSRC / f for f in HIP_HEADERS] + [REPO_ROOT / "include" / "svr_abi.h", REPO_ROOT / "include" / "svr_io.h",
                                                CSRC / "svr_internal.hpp", Path(__file__)]
    hipcc = _hipcc()
    # the effective flags are part of an object's identity: a build with other flags (SVR_EXTRA_HIPCC_FLAGS=-DSVR_TEST_HOOKS ...)
    # must not leave its objects behind for the next plain build
    stamp = obj_dir / "flags.stamp"
    flags_now = " ".join(HIPCC_FLAGS + ["|"] + FAST_FLAGS)
    if not stamp.exists() or stamp.read_text() != flags_now:
        force = True
        stamp.write_text(flags_now)

    def compile_one(name: str):
        src, obj = CSRC / name, obj_dir / (Path(name).stem + ".o")
        if not force and not _stale(obj, [src, *common]):
            return obj, False
        flags = HIPCC_FLAGS if name not in FAST_SOURCES else [f for f in HIPCC_FLAGS if f not in CONTRACT_FLAGS] + FAST_FLAGS
        cmd = [hipcc, *flags, "-c", str(src), "-o", str(obj)]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError(f"hipcc failed on {name} ({res.returncode}):\n{res.stdout}\n{res.stderr}")
        return obj, True

    with ThreadPoolExecutor(max_workers=min(6, os.cpu_count() or 1)) as pool:
        results = list(pool.map(compile_one, HIP_SOURCES))
    objs = [o for o, _ in results]
    if force or any(changed for _, changed in results) or _stale(LIB_PATH, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *[str(o) for o in objs], *LINK_LIBS, "-o", str(LIB_PATH)]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError(f"link failed ({res.returncode}):\n{res.stdout}\n{res.stderr}")
    return LIB_PATH


if __name__ == "__main__":
    print(build_hip(force="--force" in sys.argv, verbose=True))
