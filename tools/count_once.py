import sys
sys.path.insert(0,'.')
from sunvolumerender_amd import abi, host, scenes
import os
sc = scenes.make_scene("c3", trace_depth=int(os.environ.get("DEPTH", "1")))
dev = host.Device(0)
c = host.Canvas(dev, sc.width, sc.height)
scenes.apply_to_canvas(sc, c, 0)
dev.set_option(abi.OPT_COUNT, 1); dev.reset_counters()
c.paint(); dev.synchronize()
print(dev.counters())
c.close()
