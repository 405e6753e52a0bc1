"""Diagnostic (round 6, verdict r5 #4): would splitting ONE frame's samples into two sub-launches on two streams hide the drain of the first behind the second?
Timing what-if with the machinery bench.py's value_pipelined uses -- two renderers on two streams; the GPU cannot tell a second frame from the second half of the
first: same kernels, same tiles, same workspace sizes -- on a rank's share of the tiles (N = 8 diagonal deal, rank 0) and on the full frame:
    one launch of S samples                     vs.     S/2 on stream A and S/2 on stream B, both enqueued at once (B's workgroups move in as A's drain)
                                                vs.     3S/4 + S/4 (a short second launch: its own drain starts from fewer paths)
usage: tools_split_whatif.py [cfg] [size] [spp]"""
import os
import sys
import time
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__))]
import torch  # noqa: E402
import scenes  # noqa: E402
from volren_amd.shard import tile_owner_lists  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
size = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
rs, streams = [], []
for k in range(2):
    r = scenes.hip_scene(cfg, size, size)
    r.launch_target_ms = 0
    s = torch.cuda.Stream()
    r.set_stream(s.cuda_stream)
    rs.append(r)
    streams.append(s)


def run(parts):
    """parts: samples for renderer 0 [, renderer 1]; returns wall ms from the first enqueue to both streams idle (best of 3)"""
    best = 1e30
    for _ in range(3):
        for r in rs:
            r.synchronize()
        t0 = time.perf_counter()
        for r, n in zip(rs, parts):
            r.reset()
            r.render(n, sync=False)
        for r, n in zip(rs, parts):
            r.synchronize()
        best = min(best, (time.perf_counter() - t0) * 1e3)
    return best


for label, tiles in (("full frame", []), ("rank 0 of 8 (diagonal deal)", tile_owner_lists(size, size, 8, "diagonal")[0])):
    for r in rs:
        r.set_tiles(tiles)
    run([spp]); run([spp // 2, spp // 2])                                    # pools and workspaces allocated
    one = run([spp])
    half = run([spp // 2, spp // 2])
    q = run([3 * spp // 4, spp // 4])
    e = run([7 * spp // 8, spp // 8])
    print("%s %d^2 x %d spp, %s: one launch %.2f ms | two concurrent launches: S/2 + S/2 %.2f ms (x %.3f), 3S/4 + S/4 %.2f ms (x %.3f), 7S/8 + S/8 %.2f ms (x %.3f)" % (
        cfg, size, spp, label, one, half, one / half, q, one / q, e, one / e), flush=True)
