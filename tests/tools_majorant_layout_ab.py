"""Diagnostic (round 5): the two layouts of the majorant table's levels 0-1 (linear / 4x4x4-cell blocks) on the two 1024^3 sparse + emission grids, same
renderer, alternating, and what commit() chose by itself.  usage: python tests/tools_majorant_layout_ab.py [spp] > profiles/r5_majorant_layout_per_grid.txt"""
import os
import sys
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__))]
import scenes  # noqa: E402

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
print("# majorant_layout A/B, 2048 x 2048 x %d spp, path-tracing kernel alone (HIP events), one MI355X; layout 0 = linear, 1 = levels 0-1 in 4x4x4-cell blocks" % spp)
for cfg in ("c5cloud", "c5full"):
    r = scenes.hip_scene(cfg, 2048, 2048)
    r.launch_target_ms = 0
    chosen = r.majorant_blocked
    r.render(8)
    res = {0: [], 1: []}
    for rep in range(3):
        for layout in (0, 1):
            r.majorant_layout = layout
            r.reset(); r.render(spp)
            ms = r.last_pathtrace_ms()
            res[layout].append(2048 * 2048 * spp / ms / 1e3)
    best = {k: max(v) for k, v in res.items()}
    print("%-8s commit() chose layout %d | linear %s -> best %.1f Msamples/s | blocked %s -> best %.1f Msamples/s | blocked / linear = %.3f" % (
        cfg, chosen, ["%.1f" % x for x in res[0]], best[0], ["%.1f" % x for x in res[1]], best[1], best[1] / best[0]), flush=True)
    del r
