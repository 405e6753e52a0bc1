"""Diagnostic: kernel time vs work per launch (fixed cost of a launch: ramp-up + drain of the persistent waves), and what
pipelining consecutive frames over two streams -- the next frame's wavefronts move in while the previous frame's pools drain
-- buys for a rank's tile share at 2 / 4 / 8 GPUs.  usage: tools_launch_overhead.py [cfg] [spp]"""
import os, sys, time
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__))]
import torch
import scenes
from volren_amd.shard import TileShard

cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
SPP = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
W = H = 1024
r = scenes.hip_scene(cfg, W, H)
r.render(8); r.reset()
for spp in (8, 32, 128, 512):
    r.reset(); r.render(spp); ms = r.last_kernel_ms()
    print("full frame  spp %4d: %8.2f ms  %7.1f Msamples/s" % (spp, ms, W * H * spp / ms / 1e3))
r2 = scenes.hip_scene(cfg, W, H)
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
rs = [r, r2]
for x, s in zip(rs, streams):
    x.set_stream(s.cuda_stream)
FRAMES = 8
for world in (1, 2, 4, 8):
    sh = TileShard(W, H, world, 0)
    for x in rs:
        x.set_tiles(sh.mine if world > 1 else [])
        x.reset(); x.render(SPP)                      # warm-up, allocates the pools
    n = (len(sh.mine) * 256 if world > 1 else W * H) * SPP
    res = {}
    for mode in ("serial", "pipelined"):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for f in range(FRAMES):
            x = rs[f % 2] if mode == "pipelined" else rs[0]
            x.reset(); x.render(SPP, sync=False)
        torch.cuda.synchronize()
        res[mode] = (time.perf_counter() - t0) / FRAMES * 1e3
    print("1/%d of the tiles, %d spp, %d frames: serial %7.2f ms/frame (%7.1f Msamples/s, x%d = %7.1f)   pipelined over two streams %7.2f ms/frame (x%d = %7.1f)  gain %.3f" % (
        world, SPP, FRAMES, res["serial"], n / res["serial"] / 1e3, world, world * n / res["serial"] / 1e3, res["pipelined"], world, world * n / res["pipelined"] / 1e3, res["serial"] / res["pipelined"]))
