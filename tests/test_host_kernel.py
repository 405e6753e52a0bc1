"""CPU: the product's device code (volren_amd/csrc/vr_trace.h + vr_math.h -- the headers the HIP kernel is built
from) compiled for the host by tests/hostkernel, checked BIT FOR BIT against the oracle.  This is what lets the
GPU-less container catch a wrong state transition, a changed RNG draw order or a re-associated expression before the
code ever reaches an MI355X.  The harness is test-only; the product has no CPU path."""
import numpy as np
import pytest

import hk_binding as hk
import scenes
from oracle import binding as ob


def _same(a, b):
    return np.array_equal(np.ascontiguousarray(a, np.float32).view(np.uint32), np.ascontiguousarray(b, np.float32).view(np.uint32))


@pytest.mark.parametrize("name,w,h,spp", [("c1", 64, 64, 16), ("c2", 40, 40, 8), ("c3", 48, 48, 8), ("readme", 40, 40, 8)])
def test_state_machine_matches_oracle(name, w, h, spp):
    r = scenes.oracle_scene(name, w, h)
    want = r.render(spp).copy()
    got, steps = hk.render(r, spp)
    assert steps > 0
    assert _same(got, want), "relative L2 %.3e" % scenes.rel_l2(got[..., :3], want[..., :3])


@pytest.mark.parametrize("name,w,h,spp", [("c1", 64, 64, 16), ("c2", 48, 48, 16)])
def test_state_machine_with_the_device_filter_tests(name, w, h, spp):
    """The same with the DEVICE's decision of the nine stochastic-filter tests (VR_TAP_FAST: Horner weights, cross-multiplied,
    guard band, exact fallback) switched on in the host build -- the GPU kernels' collision code, run on the CPU."""
    r = scenes.oracle_scene(name, w, h)
    want = r.render(spp).copy()
    got, steps = hk.render(r, spp, fast_tap=True)
    assert steps > 0
    assert _same(got, want), "relative L2 %.3e" % scenes.rel_l2(got[..., :3], want[..., :3])


def test_tricubic_fast_path_agrees_with_the_reference():
    """tests/tools_tricubic_band.cpp: for every 61st float t in [0, 1] (17 million; `tricubic_band 1` runs all 1.07e9, recorded in
    profiles/r2q_tricubic_band_exhaustive.txt) and all 2^24 values of a draw, a filter test decided by the fast path is decided the same
    way by the reference's r < w / s; non-finite coordinates end in the guard band."""
    import subprocess
    exe = hk.build_tricubic_band_tool()
    out = subprocess.run([exe, "61"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "violations 0" in out.stdout


@pytest.mark.parametrize("name", ["c1", "c3"])
def test_global_majorant_tracking_variant(name):
    """common.glsl:333-394 (delta / ratio tracking against the global majorant; the reference compiles it out with USE_DDA):
    selectable as integrator = 1, and an unbiased cross-check of the DDA trackers."""
    r = scenes.oracle_scene(name, 40, 40)
    r.integrator = 1
    want = r.render(8).copy()
    got, _ = hk.render(r, 8)
    assert _same(got, want)
    dda = scenes.oracle_scene(name, 40, 40).render(8)
    assert abs(float(want[..., :3].mean()) - float(dda[..., :3].mean())) < 0.1 * float(dda[..., :3].mean()) + 1e-4


def test_direct_volume_rendering_integrator():
    """common.glsl:571-591 (64-step emission-absorption ray marcher through the LUT; dead code in the reference):
    integrator = 2."""
    r = scenes.oracle_scene("c3", 40, 40)
    r.integrator = 2
    r.show_environment = True
    want = r.render(4).copy()
    got, _ = hk.render(r, 4)
    assert want[..., 3].max() > 0.5 and want[..., 3].min() == 0.0
    assert _same(got, want)


@pytest.mark.parametrize("name", ["c2", "c3"])
def test_raymarch_trackers_integrator(name):
    """common.glsl:506-566 (transmittance_raymarch / sample_volume_raymarch inside trace_path; dead code in the reference):
    integrator = 3, the device's raymarch_path_sample compiled for the host against the oracle."""
    r = scenes.oracle_scene(name, 32, 32)
    r.integrator = 3
    want = r.render(4).copy()
    got, _ = hk.render(r, 4)
    assert want[..., :3].max() > 0
    assert _same(got, want)


def test_sample_chunks_and_progressive_accumulation():
    """40 spp spans two 32-sample chunks of a wave's item pool; rendering 3 + 5 more samples continues the running mean."""
    r = scenes.oracle_scene("c1", 24, 24)
    want = r.render(40).copy()
    got, _ = hk.render(r, 40)
    assert _same(got, want)
    r2 = scenes.oracle_scene("c1", 24, 24)
    want2 = r2.render(8).copy()
    fb, _ = hk.render(r2, 3)
    fb, _ = hk.render(r2, 5, fb=fb, first_sample=4)
    assert _same(fb, want2)


def test_ragged_frame():
    r = scenes.oracle_scene("c1", 37, 21)
    want = r.render(4).copy()
    got, _ = hk.render(r, 4)
    assert _same(got, want)


def test_emission_grid():
    import encoder_ref
    n = 40
    dens = scenes.synthetic_density(n)
    temp = np.clip(dens * 0.2 + 0.1 * scenes.synthetic_density(n, seed=99), 0, None).astype(np.float32)
    gd, gt = encoder_ref.encode(dens), encoder_ref.encode(temp)
    o = ob.OracleRenderer(40, 40)
    o.load_envmap(scenes.HDR)
    o.set_volume(gd, emission=gt, majorant_emission=gt.min_maj[1])
    o.cam_fov = 40.0
    o.bounces = 8
    o.albedo = (0.7, 0.8, 0.9)
    o.emission_scale = 50.0
    want = o.render(8).copy()
    got, _ = hk.render(o, 8)
    assert want[..., :3].max() > 0
    assert _same(got, want)


def test_blocked_majorant_layout(monkeypatch):
    """Round 5: the majorant table's levels 0-1 in 4x4x4-cell blocks (vr_scene.h majorant_cell_index), a per-grid layout choice of the product.  The lane code
    compiled for the host reads the layout flag at run time; the frame must not depend on it (the table is a permutation of the same cells)."""
    monkeypatch.setenv("VR_HOST_MAJ_BLOCKED", "1")
    o = scenes.oracle_scene("c1", 40, 32)
    want = o.render(4).copy()
    got, _ = hk.render(o, 4)
    assert _same(got, want)
    test_emission_grid()


def test_dense_fp16_grid():
    """Dense fp16 voxels + macro-cell majorants (no brick indirection): product device code vs oracle."""
    import encoder_ref
    dens = scenes.synthetic_density(44)[:40, :36, :44].copy()
    o = ob.OracleRenderer(32, 32)
    o.load_envmap(scenes.HDR)
    o.set_volume(encoder_ref.encode_dense_fp16(dens))
    o.cam_fov = 40.0
    o.bounces = 8
    want = o.render(8).copy()
    got, _ = hk.render(o, 8)
    assert want[..., 3].max() > 0
    assert _same(got, want)


def test_degenerate_inputs_do_not_diverge():
    """Zero-majorant volume (all-empty grid) and a camera inside the volume: both sides must agree, no hangs."""
    import encoder_ref
    g = encoder_ref.encode(np.zeros((16, 16, 16), np.float32))
    o = ob.OracleRenderer(16, 16)
    o.load_envmap(scenes.HDR)
    o.set_volume(g)
    o.cam_fov = 60.0
    want = o.render(2).copy()
    got, _ = hk.render(o, 2)
    assert _same(got, want)
    r = scenes.oracle_scene("c1", 16, 16)
    r.cam_pos = (0.0, -0.3, 0.0)
    r.cam_dir = (0.0, 1.0, 0.0)
    r.cam_up = (0.0, 0.0, 1.0)
    want = r.render(2).copy()
    got, _ = hk.render(r, 2)
    assert _same(got, want)


def test_math_matches_oracle_bitwise():
    L, H = ob.lib(), hk.lib()
    rs = np.random.RandomState(11)
    n = 4000
    cases = {
        0: (np.concatenate([1.0 - rs.randint(0, 1 << 24, n) / np.float32(1 << 24), rs.uniform(1e-30, 100, 500), [0.0, -1.0, np.inf, 1e-42]]), None),
        1: (rs.uniform(-7, 7, n), None), 2: (rs.uniform(-7, 7, n), None), 3: (rs.uniform(0.01, 1.5, n), None),
        4: (rs.uniform(-1.01, 1.01, n), None), 5: (rs.uniform(-2, 2, n), rs.uniform(-2, 2, n)),
        6: (rs.uniform(-20, 20, n), None), 7: (rs.uniform(0, 4, n), rs.uniform(0.2, 3, n)), 8: (rs.uniform(-1, 1, n), None),
    }
    for fn, (a, b) in cases.items():
        a = np.asarray(a, np.float32)
        b = np.asarray(b, np.float32) if b is not None else np.zeros_like(a)
        for x, y in zip(a, b):
            u, v = L.orc_math(fn, float(x), float(y)), H.hk_math(fn, float(x), float(y))
            assert (u != u and v != v) or np.float32(u).view(np.uint32) == np.float32(v).view(np.uint32), (fn, x, y, u, v)


def test_math_accuracy_against_libm():
    """The deterministic functions are within a few ulp of the correctly rounded result on the path tracer's domains."""
    L = ob.lib()
    rs = np.random.RandomState(2)

    def ulps(fn, xs, ref, ys=None):
        got = np.array([L.orc_math(fn, float(x), float(ys[i]) if ys is not None else 0.0) for i, x in enumerate(xs)], np.float64)
        ref = np.asarray(ref, np.float64)
        return np.abs(got - ref) / np.maximum(np.spacing(np.abs(ref).astype(np.float32)).astype(np.float64), 1e-45)
    x = (1.0 - rs.randint(1, 1 << 24, 3000) / np.float32(1 << 24)).astype(np.float32)
    assert ulps(0, x, np.log(x.astype(np.float64))).max() < 3
    x = rs.uniform(-6.3, 6.3, 3000).astype(np.float32)
    assert (np.abs(np.array([L.orc_math(1, float(v), 0) for v in x]) - np.sin(x.astype(np.float64)))).max() < 3e-7
    assert (np.abs(np.array([L.orc_math(2, float(v), 0) for v in x]) - np.cos(x.astype(np.float64)))).max() < 3e-7
    x = rs.uniform(-1, 1, 3000).astype(np.float32)
    assert (np.abs(np.array([L.orc_math(4, float(v), 0) for v in x]) - np.arccos(x.astype(np.float64)))).max() < 1e-6
    y, x2 = rs.uniform(-2, 2, 3000).astype(np.float32), rs.uniform(-2, 2, 3000).astype(np.float32)
    assert (np.abs(np.array([L.orc_math(5, float(a), float(b)) for a, b in zip(y, x2)]) - np.arctan2(y.astype(np.float64), x2.astype(np.float64)))).max() < 1e-6
    x = rs.uniform(-10, 10, 3000).astype(np.float32)
    assert ulps(6, x, np.exp(x.astype(np.float64))).max() < 3


def test_harness_under_ubsan():
    """Same code under UndefinedBehaviorSanitizer (GPU sanitizers are not available on the pool: CPU build only)."""
    import ctypes as C
    import subprocess
    import sys
    so = hk.build(sanitize=True)
    code = (
        "import sys; sys.path[:0]=[%r,%r]\n"
        "import ctypes as C, numpy as np, scenes, hk_binding as hk\n"
        "L = C.CDLL(%r); L.hk_render.restype = C.c_longlong; hk._libs[False] = L\n"
        "r = scenes.oracle_scene('c3', 24, 24); want = r.render(4).copy(); got,_ = hk.render(r, 4)\n"
        "assert np.array_equal(got.view(np.uint32), want.view(np.uint32)); print('ok')\n"
    ) % (scenes.ROOT, scenes.ROOT + "/tests", so)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-2000:]


def test_transfer_function_that_is_opaque_at_zero_density():
    """A LUT whose alpha at density 0 is not 0: the reference's majorant of a cell without data -- beyond a level's real extent, outside the grid -- is then
    vol_majorant * tf_lookup(0).a, not 0 (common.glsl:278-281, 425: the out-of-range texelFetch returns 0 and goes through the transfer function).  The
    majorant table holds that value in those cells and in its last one, which a DDA step outside the padded box reads (vr_scene.h majorant_table_cells)."""
    rs = np.random.RandomState(5)
    lut = rs.uniform(0, 1, (16, 4)).astype(np.float32)
    lut[:, 3] = np.sort(lut[:, 3])[::-1]                  # most opaque where the density is lowest
    r = scenes.oracle_scene("c2", 40, 40)
    r.set_transferfunc(lut)
    r.tf_window_left, r.tf_window_width = -0.2, 0.9
    want = r.render(6).copy()
    got, steps = hk.render(r, 6)
    assert steps > 0
    assert _same(got, want), "relative L2 %.3e" % scenes.rel_l2(got[..., :3], want[..., :3])


def _far_camera(r):
    pos = np.array([3.0e4, 0.5e4, 2.0e4], np.float32)
    r.cam_pos = tuple(pos.tolist())
    r.cam_dir = tuple((-pos / np.float32(np.linalg.norm(pos))).astype(np.float32).tolist())
    r.cam_fov = 0.003
    return r


def test_segments_that_are_not_clean():
    """vr_trace.h seg_clean: a camera 36 000 volume widths away starts its segments at |ipos| > 2^20 voxels -- not "clean", so they take the general forms of the
    DDA step / inside test / tap (float compares, NaN guard), while the scatter and shadow segments that begin inside the volume take the clean ones.  Both against
    the oracle, bit for bit; the volume is hit (alpha > 0) so the segments do march."""
    r = _far_camera(scenes.oracle_scene("c2", 40, 40))
    ipos = np.array(r.params().vol_density_inv_transform, np.float32).reshape(4, 4).T @ np.array([*r.cam_pos, 1.0], np.float32)
    assert np.abs(ipos[:3]).max() > 2.0 ** 20
    want = r.render(6).copy()
    assert want[..., 3].max() == 1.0 and want[..., 3].mean() > 0.01
    for fast_tap in (False, True):
        got, steps = hk.render(r, 6, fast_tap=fast_tap)
        assert steps > 0
        assert _same(got, want), "relative L2 %.3e" % scenes.rel_l2(got[..., :3], want[..., :3])
