"""Diagnostic (round 6): the rarely used kernel instances -- transfer function + emission grid (variants 2 / 4, tf), the run-time variant with a transfer
function -- under the library named by VOLREN_AMD_LIB.  usage: [VOLREN_AMD_LIB=build/exp_<name>/libvolren_amd.so] python tests/tools_rare_variants_ab.py <label>"""
import os
import sys
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__))]
import zlib
import numpy as np  # noqa: E402
import scenes  # noqa: E402

LUT = np.array([[0, 0, 0, 0], [0.2, 0.4, 0.9, 0.3], [0.9, 0.6, 0.2, 0.7], [1, 1, 1, 1]], np.float32)
label = sys.argv[1] if len(sys.argv) > 1 else "default"
for name, cfg, size, spp, integrator in (("tf + emission, blocked majorants (variant 4)", "c5cloud", 2048, 16, 0), ("tf + emission (variant 2)", "c5full", 2048, 32, 0),
                                         ("run-time variant + tf", "c3", 1024, 32, 1)):
    r = scenes.hip_scene(cfg, size, size)
    r.launch_target_ms = 0
    r.integrator = integrator
    if cfg != "c3":
        r.set_transferfunc(LUT)
    r.render(spp)
    ms = []
    for _ in range(2):
        r.reset()
        r.render(spp)
        ms.append(r.last_pathtrace_ms())
    crc = zlib.crc32(np.ascontiguousarray(r.framebuffer()).tobytes())
    print("== %-8s %-46s %-8s %4d^2 x %3d spp  %8.2f ms  %8.1f Msamples/s  crc %08x" % (label, name, cfg, size, spp, min(ms), size * size * spp / min(ms) / 1e3, crc), flush=True)
    del r
