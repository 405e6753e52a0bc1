"""Diagnostic (CPU, test infrastructure; round 6, verdict r5 #3): which class of memory access leaves one XCD's L2 -- density taps, emission taps, majorant levels 0-1 /
2-3, the environment's warp table, its texels, the cold path state, the sample pool?

The product's lane code compiled for the host WITH its access hooks (tests/hostkernel/host_kernel.cpp hk_l2_breakdown, -DVR_HOST_TRACE; vr_trace.h VR_TRACE) runs a
population of paths as large as the one an XCD keeps in flight (1024 wavefronts x 188 pool slots) over a band of the frame -- an XCD's segment of the work queue is a band
of tiles -- every path advancing one state transition per round, and presents every access the DEVICE would make, at its device-layout address (paired atlas, blocked or
linear majorant table, pair-blocked warp table, 64-byte cold slots), to a model of that XCD's L2 (4 MiB, 16-way, 128-byte lines, LRU).  Reported per class and per
sample: accesses, accesses to a line other than the path's previous one of that class, L2 misses (= 128-byte lines read through the fabric), and the share of the
misses; the measured totals of the same configuration (rocprofv3, profiles/r5_pmc_summary.json) stand beside them.
Not modelled: the L1s (a CU's 16 wavefronts share 32 KiB: the "new line" column is what they cannot merge), the 256 MiB Infinity Cache behind the L2, the scheduler's
batching (on the device a path waits for its event batch), the other seven XCDs (independent L2s, other bands).

The frame is simulated as the device works through it: eight bands of tile rows, one per XCD, each with its own L2 and population; the table is the sum over the
bands, the per-band totals follow (the bands through the cloud carry most of the work).

usage: tools_l2_breakdown.py [cfg=c5cloud] [frame=2048] [spp=2] [bands=8] [population=192512] [l2 MiB=4]"""
import ctypes as C
import json
import os
import subprocess
import sys
import time
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__))]
import numpy as np  # noqa: E402
import scenes  # noqa: E402
import hk_binding  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "c5cloud"
frame = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 2
bands = int(sys.argv[4]) if len(sys.argv) > 4 else 8
rows = frame // bands
population = int(sys.argv[5]) if len(sys.argv) > 5 else 1024 * 188
l2_mib = float(sys.argv[6]) if len(sys.argv) > 6 else 4.0

here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "hostkernel", "libhostkernel_trace.so")
src = os.path.join(here, "hostkernel", "host_kernel.cpp")
deps = [src] + [os.path.join(os.path.dirname(here), "volren_amd", "csrc", f) for f in ("vr_trace.h", "vr_math.h", "vr_scene.h")]
if not (os.path.exists(so) and all(os.path.getmtime(d) <= os.path.getmtime(so) for d in deps)):
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math", "-mfma", "-mavx2", "-Wno-unknown-pragmas", "-DVR_HOST_TRACE=1", "-o", so, src])
L = C.CDLL(so)
L.hk_l2_breakdown.restype = C.c_longlong

o = scenes.oracle_scene(cfg, frame, frame)
has_emission = o.emission is not None
# the layouts the product picks for this scene (RendererHIP::commit): a paired atlas when both grids are brick grids of one layout; the majorant table's levels 0-1
# in 4x4x4-cell blocks above 2^18 active bricks
paired = bool(has_emission and tuple(o.density.n_bricks) == tuple(o.emission.n_bricks) and getattr(o.density, "dense", None) is None)
n_active = int((np.asarray(o.density.range) >> 16 != (np.asarray(o.density.range) & 0xFFFF)).sum()) if hasattr(o.density, "range") else 0
blocked = paired and n_active > (1 << 18)
if blocked:
    os.environ["VR_HOST_MAJ_BLOCKED"] = "1"
p = o.params()
dd = hk_binding.grid_desc(o.density)
ed = hk_binding.grid_desc(o.emission) if has_emission else None
env, lut = o.env_tex, o.lut
out = (C.c_ulonglong * (32 * bands))()
y0 = 0
t0 = time.time()
n = L.hk_l2_breakdown(C.byref(p), C.byref(dd), C.byref(ed) if ed is not None else None, lut.ctypes.data_as(C.c_void_p) if lut is not None else None,
                      env.ctypes.data_as(C.c_void_p), env.shape[1], env.shape[0], o.impmap.ctypes.data_as(C.c_void_p), 512,
                      0, y0, frame, y0 + rows, spp, population, C.c_longlong(int(l2_mib * (1 << 20))), 16, int(paired), 1, out, bands)
dt = time.time() - t0
names = ["majorant table, levels 0-1", "majorant table, levels 2-3", "density tap", "emission tap", "environment warp table", "environment texels", "cold path state, reads",
         "cold path state, writes", "sample pool, writes"]
raw = np.array([[[out[32 * b + 3 * c + k] for k in range(3)] for c in range(9)] for b in range(bands)], np.float64)
acc = raw.sum(0) / max(n, 1)
print("# %s at %d^2 in %d bands of %d rows (one per XCD), %d spp: %d samples, %d concurrent paths and a %.0f MiB 16-way L2 per band; layouts: %s atlas, majorant levels 0-1 %s (%d active bricks); %.0f s on the host" % (
    cfg, frame, bands, rows, spp, n, population, l2_mib, "paired" if paired else "own", "in 4x4x4-cell blocks" if blocked else "linear", n_active, dt))
print("# per sample:                      accesses   to a new line   L2 misses   share of the read misses   fabric bytes (128 B per read miss; writes: 32 B per dirtied sector)")
rd = [0, 1, 2, 3, 4, 5, 6]
tot_miss = acc[rd, 2].sum()
for c, nm in enumerate(names):
    is_w = c >= 7
    print("  %-30s %9.2f %13.2f %11.2f %12s %20.0f" % (nm, acc[c, 0], acc[c, 1], acc[c, 2], "-" if is_w else "%.1f %%" % (100 * acc[c, 2] / max(tot_miss, 1e-9)), acc[c, 0] * 32 if is_w else acc[c, 2] * 128))
print("  %-30s %9.2f %13.2f %11.2f %12s %20.0f   (reads only)" % ("all reads", acc[rd, 0].sum(), acc[rd, 1].sum(), tot_miss, "100 %", tot_miss * 128))
print("  dirty lines written back by the model: %.2f per sample" % (sum(out[32 * b + 27] for b in range(bands)) / max(n, 1)))
print("# per band (XCD): share of the frame's read misses / of its accesses:", "  ".join("%d: %.0f %% / %.0f %%" % (b, 100 * raw[b, :7, 2].sum() / max(raw[:, :7, 2].sum(), 1), 100 * raw[b, :7, 0].sum() / max(raw[:, :7, 0].sum(), 1)) for b in range(bands)))
try:
    pm = json.load(open(os.path.join(os.path.dirname(here), "profiles", "r5_pmc_summary.json")))
    key = {"c5cloud": "c5cloud", "c4:512": "c4", "c2": "c2", "c5full": "c5full"}.get(cfg)
    if key:
        m = pm[key]
        print("# measured on the GPU, whole frame (rocprofv3, profiles/r5_pmc_summary.json): %.1f L1 accesses, %.1f requests to the L2 per sample, L2 hit rate %.3f -> %.1f L2 misses per sample; fabric reads %.0f B, writes %.0f B per sample" % (
            m["per_sample"]["l1_accesses"], m["per_sample"]["l1_misses_to_l2"], m["l2_hit_rate"], m["per_sample"]["l1_misses_to_l2"] * (1 - m["l2_hit_rate"]), m["fetch_bytes_per_sample"], m["write_bytes_per_sample"]))
except Exception as e:      # noqa: BLE001
    print("# (no measured totals:", e, ")")
