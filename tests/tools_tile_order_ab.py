"""Diagnostic (profiles/r4f_*): the fixed cost of a launch -- full frames and a rank's eighth of the tiles, raster tile order against costliest-first
(order_tiles), on whichever library VOLREN_AMD_LIB names (READY as a stack: build/exp_lifo, -DVR_READY_FIFO=0; as a queue: the default).
usage: python tests/tools_tile_order_ab.py [label]"""
import os
import sys
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__))]
import numpy as np  # noqa: E402
import scenes  # noqa: E402
from volren_amd.shard import TileShard  # noqa: E402

label = sys.argv[1] if len(sys.argv) > 1 else "default"
for cfg, w, h, spp in (("c2", 1024, 1024, 1024), ("c3", 1024, 1024, 1024), ("c4:512", 1024, 1024, 256), ("c4:512", 1920, 1080, 128), ("c5full", 2048, 2048, 128), ("c5cloud", 2048, 2048, 32)):
    r = scenes.hip_scene(cfg, w, h)
    r.launch_target_ms = 0
    ref = None
    for world in (1, 8):
        sh = TileShard(w, h, world, 3 % world)
        r.set_tiles(sh.mine if world > 1 else [])
        out = {}
        for mode in (0, 1, 0, 1):
            r.order_tiles = 2 * mode
            r.reset()
            r.render(spp)
            out.setdefault(mode, []).append(r.last_pathtrace_ms())
            if world == 1:
                fb = r.framebuffer()
                if ref is None:
                    ref = fb.copy()
                assert np.array_equal(fb.view(np.uint32), ref.view(np.uint32)), "order changed the image"
        a, b = min(out[0]), min(out[1])
        n = (len(sh.mine) * 256 if world > 1 else w * h) * spp
        print("[%s] %-8s %4dx%4d x%5d spp  1/%d of the tiles: raster %8.2f ms (%7.1f Msamples/s)  costliest first %8.2f ms (%7.1f)  %+.1f %%" % (
            label, cfg, w, h, spp, world, a, n / a / 1e3, b, n / b / 1e3, 100 * (a / b - 1)), flush=True)
