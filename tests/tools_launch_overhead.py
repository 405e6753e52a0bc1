"""Diagnostic: kernel time vs work per launch (fixed cost of a launch: ramp-up + drain of the persistent waves)."""
import os, sys
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__))]
import scenes
from volren_amd.shard import TileShard
r = scenes.hip_scene("c2", 1024, 1024)
r.render(8); r.reset()
for spp in (8, 32, 128, 512):
    r.reset(); r.render(spp); ms = r.last_kernel_ms()
    print("full frame  spp %4d: %8.2f ms  %7.1f Msamples/s" % (spp, ms, 1024 * 1024 * spp / ms / 1e3))
for world in (2, 4, 8):
    sh = TileShard(1024, 1024, world, 0)
    r.set_tiles(sh.mine)
    for spp in (1024,):
        r.reset(); r.render(spp); r.reset(); r.render(spp); ms = r.last_kernel_ms()
        print("1/%d of the tiles spp %4d: %8.2f ms  %7.1f Msamples/s (x%d = %7.1f)" % (world, spp, ms, len(sh.mine) * 256 * spp / ms / 1e3, world, world * len(sh.mine) * 256 * spp / ms / 1e3))
