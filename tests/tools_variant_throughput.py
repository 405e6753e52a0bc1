"""Diagnostic: throughput of every compiled instance of the path-tracing kernel on scenes that select it -- what the rarely used variants (and their
SGPR spills: profiles/r6_kernel_resources.txt) cost next to the main ones.  usage: python tests/tools_variant_throughput.py > profiles/r6_variant_throughput.txt"""
import os
import sys
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__))]
import numpy as np  # noqa: E402
import scenes  # noqa: E402

LUT = np.array([[0, 0, 0, 0], [0.2, 0.4, 0.9, 0.3], [0.9, 0.6, 0.2, 0.7], [1, 1, 1, 1]], np.float32)


def run(label, cfg, w, h, spp, integrator=0, lut=False):
    r = scenes.hip_scene(cfg, w, h)
    r.launch_target_ms = 0
    r.integrator = integrator
    if lut:
        r.set_transferfunc(LUT)
    r.render(spp)
    r.reset()
    r.render(spp)
    ms = r.last_pathtrace_ms()
    print("%-74s %-10s %4d x %4d x %4d spp  %8.2f ms  %8.1f Msamples/s" % (label, cfg, w, h, spp, ms, w * h * spp / ms / 1e3), flush=True)


print("# kernel instance (TraceCfg<tf, global, emission, dense>) selected by the scene; one MI355X; pathtrace kernel alone (HIP events)")
run("variant 0 <tf=0, 0, 0, 0>  brick grid (c2)", "c2", 1024, 1024, 256)
run("variant 0 <tf=1, 0, 0, 0>  brick grid + LUT (c3)", "c3", 1024, 1024, 256)
run("variant 1 <tf=0, 0, 0, 1>  dense fp16 grid (c4)", "c4:512", 1024, 1024, 64)
run("variant 1 <tf=1, 0, 0, 1>  dense fp16 grid + LUT", "c4:512", 1024, 1024, 64, lut=True)
run("variant 4 <tf=0, 0, 1, 0, blocked majorants>  brick + emission grid (c5cloud)", "c5cloud", 2048, 2048, 32)
run("variant 2 <tf=0, 0, 1, 0>  brick + emission grid (c5full)", "c5full", 2048, 2048, 32)
run("variant 4 <tf=1, 0, 1, 0, blocked majorants>  brick + emission grid + LUT (26 SGPR spills)", "c5cloud", 2048, 2048, 32, lut=True)
run("variant 3 <tf=0, 2, 2, 2>  global-majorant trackers, brick grid (run-time variant)", "c2", 1024, 1024, 64, integrator=1)
run("variant 3 <tf=1, 2, 2, 2>  global-majorant trackers + LUT (run-time variant, 17 SGPR spills)", "c3", 1024, 1024, 64, integrator=1)
run("variant 3 <tf=0, 2, 2, 2>  global-majorant trackers, emission grids (run-time variant)", "c5cloud", 2048, 2048, 8, integrator=1)
