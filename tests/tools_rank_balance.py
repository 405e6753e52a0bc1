"""Diagnostic (one GPU stands in for N): renders EVERY rank's tile set of an N-way shard, one after the other, and reports
    predicted strong scaling = T(full frame) / max_r T(rank r's tiles),      balance = max_r T / mean_r T
for the tile deals of volren_amd/shard.py.  The all_gather (4-8 MiB per rank over xGMI) is not in it; the fixed cost of a launch is.
usage: tools_rank_balance.py [cfg] [width] [height] [spp] [schemes=diagonal,hashed]"""
import os
import sys
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__))]
import scenes  # noqa: E402
from volren_amd.shard import tile_owner_lists  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
W = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
H = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
SPP = int(sys.argv[4]) if len(sys.argv) > 4 else 1024
schemes = (sys.argv[5] if len(sys.argv) > 5 else "diagonal,hashed").split(",")
r = scenes.hip_scene(cfg, W, H)
r.launch_target_ms = 0                      # no probe launch when the tile set changes: a rank's time is its launches' time
r.render(min(SPP, 32))


def timed(tiles):
    r.set_tiles(tiles)
    r.reset()
    r.render(SPP)
    return r.last_kernel_ms()


t_full = min(timed([]), timed([]))
print("%s %dx%d %d spp: full frame %.2f ms (%.1f Msamples/s)" % (cfg, W, H, SPP, t_full, W * H * SPP / t_full / 1e3))
for scheme in schemes:
    for n in (2, 4, 8):
        lists = tile_owner_lists(W, H, n, scheme)
        t = [timed(tl) for tl in lists]
        worst, mean = max(t), sum(t) / n
        print("  %-8s N=%d: rank times %s ms   max/mean %.3f   predicted scaling %.2fx of %d (sum of shares / full = %.3f)" % (
            scheme, n, " ".join("%.2f" % x for x in t), worst / mean, t_full / worst, n, sum(t) / t_full))
