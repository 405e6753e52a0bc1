import os, sys, time
sys.path[:0] = ["/root/repo", "/root/repo/tests"]
import scenes
for cfg, w, h, spp in (("c4:512", 1920, 1080, 4096), ("c5full", 2048, 2048, 4096)):
    r = scenes.hip_scene(cfg, w, h)
    r.launch_target_ms = 0
    for mb in (16384, 65536, 16384, 65536):
        r.sample_pool_mb = mb
        r.reset(); r.render(spp)           # allocates
        r.reset(); t0 = time.perf_counter(); r.render(spp); dt = time.perf_counter() - t0
        print(cfg, "pool %d MiB: %d launches, frame %.1f ms, %.1f Msamples/s" % (mb, r.last_launches, dt * 1e3, w * h * spp / dt / 1e6), flush=True)
    del r
