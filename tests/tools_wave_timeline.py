"""Diagnostic (profiles/r4f_*): where the fixed cost of a launch goes.  Instrumented kernels record, per wavefront, when it started, when it found the work
queue empty and when it ended (device's constant 100 MHz clock).  usage: python tests/tools_wave_timeline.py [cfg] [size | WxH] [spp] [tiles 1/N]"""
import os
import sys
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__))]
import numpy as np  # noqa: E402
import scenes  # noqa: E402
from volren_amd.shard import TileShard  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
size = sys.argv[2] if len(sys.argv) > 2 else "1024"
w, h = (int(v) for v in size.split("x")) if "x" in size else (int(size), int(size))
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 128
world = int(sys.argv[4]) if len(sys.argv) > 4 else 1
r = scenes.hip_scene(cfg, w, h)
r.launch_target_ms = 0
if world > 1:
    r.set_tiles(TileShard(w, h, world, 3 % world).mine)
for order in (0, 1):
    r.order_tiles = 2 * order
    r.reset(); r.render(spp)
    r.sched_stats(True)
    r.reset(); r.render(spp)
    ms = r.last_pathtrace_ms()
    t = r.wave_timeline() * 1e3                      # ms
    r.sched_stats(False)
    end = t[:, 2].max()
    q = np.percentile
    print("%s %dx%d x %d spp, 1/%d of the tiles, order_tiles %d: kernel %.2f ms (instrumented), %d wavefronts" % (cfg, w, h, spp, world, order, ms, len(t)))
    print("   start:        last wavefront begins at %.3f ms" % t[:, 0].max())
    print("   queue empty:  first %.2f  median %.2f  last %.2f ms" % (t[:, 1].min(), q(t[:, 1], 50), t[:, 1].max()))
    print("   end:          first %.2f  10%% %.2f  median %.2f  90%% %.2f  99%% %.2f  last %.2f ms" % (t[:, 2].min(), q(t[:, 2], 10), q(t[:, 2], 50), q(t[:, 2], 90), q(t[:, 2], 99), end))
    print("   drain of a wavefront (end - queue empty): median %.2f  90%% %.2f  99%% %.2f  max %.2f ms;  idle wavefront-time before the kernel ends: %.1f %% of the launch" % (
        q(t[:, 2] - t[:, 1], 50), q(t[:, 2] - t[:, 1], 90), q(t[:, 2] - t[:, 1], 99), (t[:, 2] - t[:, 1]).max(), 100.0 * (end - t[:, 2]).mean() / end), flush=True)
