"""Diagnostic: tolerance-mode (fast_math) kernels vs the bit-exact default: speed and relative L2.  usage: tools_fast_math.py [cfg] [size] [spp]"""
import os, sys
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__))]
import numpy as np
import scenes

cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
size = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 256
r = scenes.hip_scene(cfg, size, size)
out = {}
for mode in (0, 1):
    r.fast_math = mode
    r.reset(); r.render(spp); r.reset(); r.render(spp)
    out[mode] = (r.last_kernel_ms(), r.framebuffer().copy())
ms0, a = out[0]; ms1, b = out[1]
rl2 = scenes.rel_l2(b[..., :3], a[..., :3])
d = np.abs(b[..., :3].astype(np.float64) - a[..., :3]).max(-1) / (np.abs(a[..., :3]).max(-1) + 1e-6)
print("%s %dx%d %d spp: exact %.2f ms (%.0f Msamples/s)  fast %.2f ms (%.0f Msamples/s)  speed-up %.3f  rel L2 %.3e  pixels > 1e-3: %.4f  mean ratio %.6f" % (
    cfg, size, size, spp, ms0, size * size * spp / ms0 / 1e3, ms1, size * size * spp / ms1 / 1e3, ms0 / ms1, rl2, (d > 1e-3).mean(), b[..., :3].mean() / a[..., :3].mean()))
e2 = ((b[..., :3].astype(np.float64) - a[..., :3]) ** 2).sum(-1)
top = np.argsort(e2.ravel())[::-1][:5]
tot = e2.sum()
print("largest pixel contributions to the squared difference:", ", ".join("(%d,%d): %.1f %% exact %s fast %s" % (i % size, i // size, 100 * e2.ravel()[i] / tot, np.round(a.reshape(-1, 4)[i, :3], 3), np.round(b.reshape(-1, 4)[i, :3], 3)) for i in top))
print("rel L2 without the 5 largest pixels: %.3e" % np.sqrt((tot - e2.ravel()[top].sum()) / (a[..., :3].astype(np.float64) ** 2).sum()))
