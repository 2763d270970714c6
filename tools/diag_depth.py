"""Diagnostic: depth-2 window vs oracle on a big scene under option variations."""
import sys, os
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import numpy as np
from oracle import binding
from sunvolumerender_amd import abi, host, scenes
from tests.test_configs_gpu import Rig

name = sys.argv[1] if len(sys.argv) > 1 else "c3"
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 2
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 4
dev = host.Device(0, fatal_errors=False)
r = Rig(dev, name)
r.canvas.SetScatterTimes(depth)
w = (480, 500, 544, 516)
o = binding.OracleScene(r.sc)
ref = o.new_hdr()
for f in range(frames):
    o.render_pathtracer(ref, f, trace_depth=depth, window=w, count=False, nthreads=16)
x0, y0, x1, y1 = w
R = ref[y0:y1, x0:x1]
for what, kw in (("default", {}), ("count", dict(count=True)), ("skip0", dict(skip=0)), ("rayskip0", dict(rayskip=0)), ("pixel", dict(kernel=abi.KERNEL_PIXEL)),
                 ("window", dict(window=w)), ("seq", dict(batch=False)), ("fl0", dict(fl2=0))):
    fl2 = kw.pop("fl2", None)
    if fl2 is not None:
        dev.set_option(abi.OPT_FRAMES_PER_WAVE_LOG2, fl2)
    hdr, _, _ = r.run(frames, **kw)
    dev.set_option(abi.OPT_FRAMES_PER_WAVE_LOG2, -1)
    H = hdr[y0:y1, x0:x1]
    nd = int((H.view(np.uint32) != R.view(np.uint32)).sum())
    print(f"{name} d{depth} f{frames} {what:10s}: {nd} of {H.size} floats differ; mean hip {H.mean():.5f} ref {R.mean():.5f}", flush=True)
r.close()
