"""Diagnostic: the same frame rendered N times by fresh and reused renderers, in the bit-exact and the tolerance mode -- every CRC of a mode must be the same.
With `poison` the workspace and the sample pool are filled with NaN patterns before every launch (VR_TEST_POISON_WORKSPACE=2, renderer.cpp).
usage: tools_determinism.py [cfg] [size] [spp] [n] [poison]"""
import os, sys, zlib
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__))]
import numpy as np
import scenes

cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
size = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
n = int(sys.argv[4]) if len(sys.argv) > 4 else 6
if len(sys.argv) > 5 and sys.argv[5] == "poison":
    os.environ["VR_TEST_POISON_WORKSPACE"] = "2"
crcs = {0: [], 1: []}
keep = []
for k in range(n):
    r = scenes.hip_scene(cfg, size, size)          # a fresh renderer every other round, two alive at a time (as bench.py's tolerance-mode leg has them)
    keep = (keep + [r])[-2:]
    for mode in (0, 1):
        r.fast_math = mode
        r.reset(); r.render(spp)
        fb = r.framebuffer()
        crcs[mode].append(zlib.crc32(np.ascontiguousarray(fb).tobytes()))
        if len(crcs[mode]) > 1 and crcs[mode][-1] != crcs[mode][0]:
            ref = getattr(sys.modules[__name__], "first_%d" % mode)
            d = np.abs(fb[..., :3].astype(np.float64) - ref[..., :3]).sum(-1)
            ys, xs = np.nonzero(d)
            print("mode %d round %d DIFFERS: %d pixels, bbox x %d..%d y %d..%d, largest |diff| %.4g" % (mode, k, len(xs), xs.min(), xs.max(), ys.min(), ys.max(), d.max()))
        elif len(crcs[mode]) == 1:
            setattr(sys.modules[__name__], "first_%d" % mode, fb.copy())
print("%s %dx%d %d spp, %d rounds: exact CRCs %s, tolerance-mode CRCs %s" % (cfg, size, size, spp, n, sorted(set(crcs[0])), sorted(set(crcs[1]))))
print("deterministic" if len(set(crcs[0])) == 1 and len(set(crcs[1])) == 1 else "NOT deterministic")
