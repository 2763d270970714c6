#!/usr/bin/env python3
"""A variant of libsvr_hip.so that differs from the default build in ONE translation unit compiled with extra flags (same-box A/B of a
compile-time knob without rebuilding the rest): tools/variant_lib.py <name> <source.hip> <flag>...  ->  sunvolumerender_amd/lib/libsvr_hip_<name>.so
Use it with SVR_HIP_LIB=<path> (tools/sweep.py, bench.py ...)."""
import subprocess
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from sunvolumerender_amd import _build  # noqa: E402

name, src, flags = sys.argv[1], sys.argv[2], sys.argv[3:]
_build.build_hip()
obj_dir = _build.LIB_DIR / "obj"
vobj = _build.LIB_DIR / f"variant_{name}_{Path(src).stem}.o"
base = _build.HIPCC_FLAGS if src not in _build.FAST_SOURCES else [f for f in _build.HIPCC_FLAGS if f not in _build.CONTRACT_FLAGS] + _build.FAST_FLAGS
subprocess.run([_build._hipcc(), *base, *flags, "-c", str(_build.CSRC / src), "-o", str(vobj)], check=True)
objs = [str(vobj) if Path(s).stem == Path(src).stem else str(obj_dir / (Path(s).stem + ".o")) for s in _build.HIP_SOURCES]
out = _build.LIB_DIR / f"libsvr_hip_{name}.so"
subprocess.run([_build._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *objs, *_build.LINK_LIBS, "-o", str(out)], check=True)
print(out)
