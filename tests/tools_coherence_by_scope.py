"""Diagnostic (CPU, test infrastructure): how coherent could the gathers of the path tracer be made by regrouping paths?

The product's lane code, compiled for the host (tests/hostkernel), traces random pixel-samples of a scene and logs, per lane-step, which 128-byte
line of the majorant table a marching path reads next and which voxel block a path at a tentative collision stands in.  A population of S paths
"in flight" = S log entries drawn at random (each path at a random point of its life).  For every scope S -- one wavefront's pool, one CU, one
XCD, the whole GPU, and a hypothetical HBM-resident pool -- the S accesses are sorted by line (the best any regrouping could do) and cut into
wave-sized groups of 64: reported is the mean number of DISTINCT lines per group (64 = no sharing at all; the TCP's cost follows this number).

usage: tools_coherence_by_scope.py [cfg=c4:512] [samples per pixel=2]      (c4:512 needs ~2 min to encode the 512^3 grid for the host)"""
import ctypes as C
import os
import sys
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__))]
import numpy as np  # noqa: E402
import scenes  # noqa: E402
import hk_binding  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "c4:512"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 2
W = H = 256
o = scenes.oracle_scene(cfg, W, H)
L = hk_binding.lib()
L.hk_trace_count.restype = C.c_ulonglong
cap = 64 << 20
buf = np.zeros(cap, np.uint32)
L.hk_set_trace(buf.ctypes.data_as(C.c_void_p), C.c_ulonglong(cap))
fb, steps = hk_binding.render(o, spp)
n = int(L.hk_trace_count())
L.hk_set_trace(None, C.c_ulonglong(0))
ev = buf[:n]
kinds = {"majorant line of a marching path's next DDA step": ev[(ev >> 30) == 0] & 0x3FFFFFFF,
         "voxel block of a path at a tentative collision": ev[(ev >> 30) == 1] & 0x3FFFFFFF}
print("%s, %dx%d x %d spp on the host: %d lane-steps logged (%.1f per sample)" % (cfg, W, H, spp, n, n / (W * H * spp)))
rs = np.random.RandomState(1)
SCOPES = (("one wavefront's pool", 192), ("one CU (16 wavefronts)", 2816), ("one XCD (32 CUs)", 89600), ("the GPU (256 CUs)", 716800), ("a 16 M-path pool in HBM", 1 << 24))
for what, a in kinds.items():
    print("%s: %d logged, %d distinct lines touched in all" % (what, a.size, np.unique(a).size))
    g = a[rs.randint(0, a.size, 64 * 4096)].reshape(-1, 64)
    g.sort(axis=1)
    print("   today (64 paths that happen to share a wavefront)          : %5.1f distinct lines per 64-lane load" % float(((np.diff(g, axis=1) != 0).sum(1) + 1).mean()))
    for name, S in SCOPES:
        reps = max(1, min(64, (1 << 22) // S))
        tot = 0.0
        for _ in range(reps):
            pick = a[rs.randint(0, a.size, S)]
            pick.sort()
            m = (S // 64) * 64
            g = pick[:m].reshape(-1, 64)
            tot += float(((np.diff(g, axis=1) != 0).sum(1) + 1).mean())
        print("   scope %-28s S = %8d : %5.1f distinct lines per 64-lane load after perfect regrouping (64 = none shared)" % (name, S, tot / reps))
# march steps by majorant level (which levels an LDS-resident copy would have to hold)
ms = o.density.n_bricks
import math
k = sum(max(3, math.ceil(math.log2(max(1, n)))) for n in ms)
offs = [0, 1 << k, (9 << k) >> 3, (73 << k) >> 6, ((73 << k) >> 6) + (1 << (k - 9))]
a = kinds["majorant line of a marching path's next DDA step"]
for lv in range(4):
    sel = (a >= offs[lv] // 64) & (a < max(offs[lv + 1] // 64, offs[lv] // 64 + 1))
    print("   majorant level %d: %5.1f %% of the march steps, table cells %d..%d (%d KiB as fp16)" % (lv, 100.0 * sel.mean(), offs[lv], offs[lv + 1], (offs[lv + 1] - offs[lv]) * 2 // 1024))
