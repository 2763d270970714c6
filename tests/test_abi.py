"""The C-ABI library builds, loads, and exports every symbol include/svr_abi.h and include/svr_io.h declare; the POD layouts
match the reference's classes byte for byte.  No compute calls (no GPU here)."""
import ctypes as C
import re
from pathlib import Path

import pytest

from sunvolumerender_amd import abi

ROOT = Path(__file__).resolve().parents[1]
HEADER = (ROOT / "include" / "svr_abi.h").read_text() + "\n" + (ROOT / "include" / "svr_io.h").read_text()


def declared_functions():
    body = re.sub(r"/\*.*?\*/", "", HEADER, flags=re.S)
    body = re.sub(r"//[^\n]*", "", body)
    names = re.findall(r"^[ \t]*(?:const\s+)?[A-Za-z_][A-Za-z_0-9\s\*]*?\b([a-z_][a-z_0-9]*)\s*\([^;{]*\)\s*;", body, flags=re.M)
    return sorted(set(n for n in names if not n.startswith("sizeof")))


def test_library_present_and_exports_every_declared_symbol():
    assert abi.library_path().exists(), "libsvr_hip.so must be built (python -m sunvolumerender_amd._build)"
    lib = C.CDLL(str(abi.library_path()))
    decl = declared_functions()
    assert len(decl) >= 35
    for name in decl:
        assert hasattr(lib, name), f"{name} declared in include/*.h but not exported"
        assert name in abi.PROTOTYPES, f"{name} has no ctypes prototype"
    for name in abi.PROTOTYPES:
        assert name in decl, f"{name} bound in abi.py but not declared in include/*.h"


def test_reference_entry_points_have_reference_names():
    for n in ["render_pathtracer", "setup_volume", "setup_transferfunction", "setup_camera", "setup_env_lights",
              "setup_area_lights", "render_raycasting"]:
        assert n in abi.PROTOTYPES      # pathtracer.h:17-24, raycasting.h:8


def test_pod_layouts_match_reference():
    # sizeof / offsetof of the reference classes
    V = abi.cudaVolume
    assert C.sizeof(V) == 112 and V.tex.offset == 40 and V.densityScale.offset == 48 and V.invMaxMagnitude.offset == 52
    assert V.gradientFactor.offset == 56 and V.spacing.offset == 60 and V.invSpacing.offset == 72
    assert V.x_clip.offset == 84 and V.y_clip.offset == 92 and V.z_clip.offset == 100
    T = abi.cudaTransferFunction
    assert C.sizeof(T) == 16 and T.tex.offset == 0 and T.maxOpacity.offset == 8
    K = abi.cudaCamera
    assert C.sizeof(K) == 76 and [getattr(K, f).offset for f in ("imageW", "imageH", "exposure", "apeture", "focalLength",
                                                                 "aspectRatio", "tanFovxOverTwo", "pos", "u", "v", "w")] == \
        [0, 4, 8, 12, 16, 20, 24, 28, 40, 52, 64]
    D = abi.cudaDisk
    assert C.sizeof(D) == 28 and D.radius.offset == 0 and D.center.offset == 4 and D.normal.offset == 16
    L = abi.cudaAreaLight
    assert C.sizeof(L) == 44 and L.disk.offset == 0 and L.color.offset == 28 and L.intensity.offset == 40
    E = abi.cudaEnvironmentLight
    assert C.sizeof(E) == 32 and E.tex.offset == 0 and E.defaultRadiance.offset == 8 and E.intensity.offset == 20 and E.offset.offset == 24
    R = abi.RenderParams
    assert C.sizeof(R) == 16 and R.traceDepth.offset == 0 and R.frameNo.offset == 4 and R.hdrBuffer.offset == 8
    B = abi.cudaBBox
    assert C.sizeof(B) == 36 and B.vmin.offset == 0 and B.vmax.offset == 12 and B.invSize.offset == 24


def test_oracle_structs_agree_with_abi(oracle):
    assert oracle.svo_sizeof(1) == C.sizeof(abi.cudaVolume)
    assert oracle.svo_sizeof(2) == C.sizeof(abi.cudaTransferFunction)
    assert oracle.svo_sizeof(3) == C.sizeof(abi.cudaCamera)
    assert oracle.svo_sizeof(4) == C.sizeof(abi.cudaAreaLight)
    assert oracle.svo_sizeof(5) == C.sizeof(abi.cudaEnvironmentLight)


def test_product_does_not_reference_the_oracle():
    """The shipped package must not import, link or call anything under oracle/."""
    pkg = ROOT / "sunvolumerender_amd"
    for p in list(pkg.rglob("*.py")) + list(pkg.rglob("*.hip")) + list(pkg.rglob("*.hpp")) + [ROOT / "include" / "svr_abi.h"]:
        txt = p.read_text()
        assert "svr_oracle" not in txt and "from oracle" not in txt and "import oracle" not in txt and "svo_" not in txt, p
    import subprocess
    out = subprocess.run(["ldd", str(abi.library_path())], capture_output=True, text=True).stdout
    assert "oracle" not in out


@pytest.mark.parametrize("example", ["headless_canvas", "render_mhd"])
def test_cpp_host_headers_compile_and_link(tmp_path, example):
    """include/sunvolumerender/{host_api,canvas}.hpp + the examples build with plain g++ against the C ABI
    (compile + link only; running them needs a GPU: tests/test_io_gpu.py)."""
    import shutil
    import subprocess

    if shutil.which("g++") is None:
        pytest.skip("no g++")
    exe = tmp_path / example
    cmd = ["g++", "-std=c++14", "-O1", "-Wall", "-Werror", f"-I{ROOT / 'include'}", str(ROOT / "examples" / f"{example}.cpp"),
           "-o", str(exe), f"-L{abi.library_path().parent}", "-lsvr_hip"]
    res = subprocess.run(cmd, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    assert exe.exists()


def test_cpp_rccl_assembly_example_compiles(tmp_path):
    """examples/assemble_rccl.cpp -- a C++ host doing what bench.py --gpus N does (row shards, svr_assemble_frame with its own
    ncclComm_t, tone map of the assembled frame) -- compiles and links against rccl.h / librccl.so and libsvr_hip.so.
    (Running it needs >= 2 GPUs.)"""
    import shutil
    import subprocess

    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not Path(hipcc).exists() or not Path("/opt/rocm/include/rccl/rccl.h").exists():
        pytest.skip("no hipcc / rccl.h")
    exe = tmp_path / "assemble_rccl"
    cmd = [hipcc, "-std=c++14", "-O1", "-Wall", "-Werror", "-Wno-unused-result", f"-I{ROOT / 'include'}", str(ROOT / "examples" / "assemble_rccl.cpp"),
           "-o", str(exe), f"-L{abi.library_path().parent}", "-lsvr_hip", "-L/opt/rocm/lib", "-lrccl"]
    res = subprocess.run(cmd, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    assert exe.exists()
