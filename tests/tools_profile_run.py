"""Diagnostic driver for rocprofv3: one warm-up + one measured render.  usage: tools_profile_run.py [cfg] [size] [spp]"""
import os
import sys
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__))]
import scenes  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
size = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 32
r = scenes.hip_scene(cfg, size, size)
if len(sys.argv) > 4:
    t = [int(x) for x in sys.argv[4].split(",")]
    r.set_sched(t + [0] * (8 - len(t)))
r.render(spp)
r.reset()
r.render(spp)
print("kernel ms", r.last_kernel_ms(), "Msamples/s", size * size * spp / r.last_kernel_ms() / 1e3)
