"""ctypes binding of tests/hostkernel/libhostkernel.so: the product's device code (vr_trace.h) built for the host.
TEST HARNESS ONLY -- lets the CPU-only suite check the lane state machine against the oracle."""
import ctypes as C
import os
import subprocess

import numpy as np

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hostkernel")
_SO = os.path.join(_DIR, "libhostkernel.so")
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build(sanitize=False, fast_tap=False):
    """fast_tap: the device's decision of the stochastic-filter tests (VR_TAP_FAST, vr_trace.h) instead of the reference's loop."""
    so = _SO if not sanitize else os.path.join(_DIR, "libhostkernel_san.so")
    if fast_tap:
        so = os.path.join(_DIR, "libhostkernel_fasttap.so")
    src = os.path.join(_DIR, "host_kernel.cpp")
    deps = [src] + [os.path.join(_ROOT, "volren_amd", "csrc", f) for f in ("vr_trace.h", "vr_math.h", "vr_scene.h")]
    if os.path.exists(so) and all(os.path.getmtime(d) <= os.path.getmtime(so) for d in deps):
        return so
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math", "-mfma", "-mavx2",
           "-Wno-unknown-pragmas", "-o", so, src]
    if sanitize:
        cmd[1:1] = ["-fsanitize=undefined", "-fno-sanitize-recover=undefined", "-g"]
    if fast_tap:
        cmd[1:1] = ["-DVR_TAP_FAST=1"]
    subprocess.check_call(cmd)
    return so


def build_tricubic_band_tool():
    """tests/tools_tricubic_band.cpp: exhaustive / strided check of the fast filter tests against the reference's."""
    exe = os.path.join(_DIR, "tricubic_band")
    src = os.path.join(os.path.dirname(_DIR), "tools_tricubic_band.cpp")
    deps = [src] + [os.path.join(_ROOT, "volren_amd", "csrc", f) for f in ("vr_trace.h", "vr_math.h", "vr_scene.h")]
    if not (os.path.exists(exe) and all(os.path.getmtime(d) <= os.path.getmtime(exe) for d in deps)):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-mfma", "-mavx2", "-fopenmp",
                               "-Wno-unknown-pragmas", "-o", exe, src])
    return exe


class GridDesc(C.Structure):
    _fields_ = [("nb", C.c_uint32 * 3), ("atlas_dim", C.c_uint32 * 3), ("n_mips", C.c_int32),
                ("indirection", C.c_void_p), ("range", C.c_void_p), ("atlas", C.c_void_p), ("mips", C.c_void_p * 3),
                ("dense", C.c_void_p), ("dim", C.c_uint32 * 3)]


def grid_desc(g):
    d = GridDesc()
    d.nb[:] = g.n_bricks
    d.atlas_dim[:] = g.atlas_dim
    d.n_mips = len(g.mips)
    d.indirection = g.indirection.ctypes.data
    d.range = g.range.ctypes.data
    d.atlas = g.atlas.ctypes.data
    for i, (_, a) in enumerate(g.mips):
        d.mips[i] = a.ctypes.data
    if getattr(g, "dense", None) is not None:
        d.dense = g.dense.ctypes.data
        d.dim[:] = g.extent
    return d


_libs = {}


def lib(fast_tap=False):
    if fast_tap not in _libs:
        L = C.CDLL(build(fast_tap=fast_tap))
        L.hk_render.restype = C.c_longlong
        L.hk_math.restype = C.c_float
        L.hk_math.argtypes = [C.c_int, C.c_float, C.c_float]
        _libs[fast_tap] = L
    return _libs[fast_tap]


def render(orc_renderer, spp, rect=None, fb=None, first_sample=1, fast_tap=False):
    """Run the host-compiled product kernel on the scene held by an oracle.binding.OracleRenderer."""
    L = lib(fast_tap)
    p = orc_renderer.params()
    assert C.sizeof(p) == L.hk_uniforms_size(), (C.sizeof(p), L.hk_uniforms_size())
    dd = grid_desc(orc_renderer.density)
    ed = grid_desc(orc_renderer.emission) if orc_renderer.emission is not None else None
    w, h = orc_renderer.w, orc_renderer.h
    if fb is None:
        fb = np.zeros((h, w, 4), np.float32)
    x0, y0, x1, y1 = rect if rect is not None else (0, 0, w, h)
    env = orc_renderer.env_tex
    lut = orc_renderer.lut
    steps = L.hk_render(C.byref(p), C.byref(dd), C.byref(ed) if ed is not None else None,
                        lut.ctypes.data_as(C.c_void_p) if lut is not None else None,
                        env.ctypes.data_as(C.c_void_p), env.shape[1], env.shape[0],
                        orc_renderer.impmap.ctypes.data_as(C.c_void_p), 512,
                        fb.ctypes.data_as(C.c_void_p), x0, y0, x1, y1, first_sample, spp)
    assert steps >= 0
    return fb, steps
