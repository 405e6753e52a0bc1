"""Diagnostic (what-if): the c4 scene on a 512^3 dense fp16 field that is PERIODIC with period p voxels (the first p^3 block of the
c4 field, tiled).  A library built with -DVR_WHATIF_WRAP=p reads every voxel tap from the first period (same values, same image),
so its taps have a working set of 2 p^3 bytes: p = 128 -> 4 MiB (one XCD's L2), p = 16 -> 8 KiB (L1).  Against the default library
on the same field this is what voxel taps cost beyond the cache level they would hit if paths were perfectly regrouped by
position -- an upper bound for any regrouping scheme, before its exchange costs.  usage: tools_whatif_wrap.py <period> [size] [spp]"""
import os
import sys
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__))]
import numpy as np  # noqa: E402
import scenes  # noqa: E402
import volren_amd  # noqa: E402

p = int(sys.argv[1])
size = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 64
n = 512
f = scenes.synthetic_dense_fp16(n)
o = (n - p) // 2                                   # a block from the middle of the cloud
f = np.ascontiguousarray(np.tile(f[o:o + p, o:o + p, o:o + p], (n // p,) * 3))
scenes._DENSE_CACHE[n] = f
r = scenes.configure_dense(volren_amd.Renderer(size, size), False, n)
r.render(spp)
r.reset()
r.render(spp)
fb = r.framebuffer()
print("period", p, "kernel ms", r.last_kernel_ms(), "Msamples/s", size * size * spp / r.last_kernel_ms() / 1e3,
      "checksum", int(fb.view(np.uint32).astype(np.uint64).sum()), "non-zero voxels %.3f" % float((f > 0).mean()))
