"""Diagnostic (round 5): run-time scheduler thresholds re-swept on the kernels with VR_HOT_PAIRS copies of the hot pair per iteration.
thr = NEW, cap, hungry (low-water mark of live paths), COLLIDE (0 = per kernel: 24, 32 with a LUT / emission grid), NEE, POSTNEE, ESCAPE.
usage: python tests/tools_sweep5.py [cfg ...] > profiles/r5e_threshold_sweep_hot_pairs.txt"""
import os
import sys
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__))]
import scenes  # noqa: E402

sets = ["64,0,56,0,60,60,64", "64,0,48,0,60,60,64", "64,0,40,0,60,60,64", "64,0,60,0,60,60,64", "64,0,56,0,64,64,64", "64,0,56,0,56,56,64", "64,0,56,0,48,48,64", "64,0,56,0,40,40,64",
        "64,0,56,16,60,60,64", "64,0,56,24,60,60,64", "64,0,56,32,60,60,64", "64,0,56,40,60,60,64", "64,0,56,48,60,60,64", "56,0,56,0,60,60,56", "48,0,48,0,48,48,48", "64,0,56,0,60,60,48",
        "64,0,48,32,56,56,64", "64,0,48,40,56,56,64"]
ALL = (("c2", 1024, 128), ("c3", 1024, 128), ("c4:512", 1024, 32), ("c5cloud", 2048, 16), ("c5full", 2048, 32))
for cfg, size, spp in [c for c in ALL if len(sys.argv) < 2 or c[0] in sys.argv[1:]]:
    r = scenes.hip_scene(cfg, size, size)
    r.launch_target_ms = 0
    r.render(spp)
    for t in sets:
        thr = [int(x) for x in t.split(",")]
        r.set_sched(thr + [0])
        best = 1e9
        for k in range(2):
            r.reset(); r.render(spp); best = min(best, r.last_pathtrace_ms())
        print("%s thr %s  ms %.2f  Msamples/s %.0f" % (cfg, t, best, size * size * spp / best / 1e3), flush=True)
    del r
