"""Diagnostic driver for rocprofv3: one warm-up + one measured render.  usage: tools_profile_run.py [cfg] [size | WxH] [spp] [thresholds]"""
import os
import sys
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__))]
import scenes  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
size = sys.argv[2] if len(sys.argv) > 2 else "1024"
w, h = (int(v) for v in size.split("x")) if "x" in size else (int(size), int(size))
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 32
r = scenes.hip_scene(cfg, w, h)
r.launch_target_ms = 0                      # one launch per render: the counters of a pass belong to two equal dispatches
if len(sys.argv) > 4:
    t = [int(x) for x in sys.argv[4].split(",")]
    r.set_sched(t + [0] * (8 - len(t)))
r.render(spp)
r.reset()
r.render(spp)
print("kernel ms", r.last_kernel_ms(), "Msamples/s", w * h * spp / r.last_kernel_ms() / 1e3, "launches", r.last_launches)
