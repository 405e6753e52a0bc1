"""Worker of tests/test_glsl_pin.py: evaluates the CPU oracle (the build selected by VOLREN_ORACLE_SO, default the
standard one) on every scene and probe input stored in tests/golden/glsl_golden.npz and prints, as JSON, how it compares
with the stored outputs of the reference's GLSL kernels.  Separate process because the oracle binding loads one build."""
import ctypes as C
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.dirname(HERE), HERE]
import scenes  # noqa: E402
from oracle import binding as ob  # noqa: E402

G = np.load(os.path.join(HERE, "golden", "glsl_golden.npz"))
META = json.load(open(os.path.join(HERE, "golden", "glsl_golden.json")))
W, H, SPP = META["width"], META["height"], META["spp"]


def image_metrics(ref, img):
    d = np.abs(img.astype(np.float64) - ref)
    rel = d[..., :3].max(-1) / (np.abs(ref[..., :3]).max(-1) + 1e-6)
    return dict(rel_l2=float(np.linalg.norm(img[..., :3].astype(np.float64) - ref[..., :3]) / np.linalg.norm(ref[..., :3])),
                within_1e5=float((rel <= 1e-5).mean()), within_1e3=float((rel <= 1e-3).mean()),
                alpha_max_diff=float(d[..., 3].max()), mean_ratio=float(img[..., :3].mean() / ref[..., :3].mean()))


def emission_scene(w, h):
    import encoder_ref
    dens = scenes.synthetic_density(40)
    temp = np.clip(dens * 0.2 + 0.1 * scenes.synthetic_density(40, seed=99), 0, None).astype(np.float32)
    at = encoder_ref.encode_arrays(temp)
    o = ob.OracleRenderer(w, h)
    o.load_envmap(scenes.HDR)
    o.set_volume(encoder_ref.encode(dens), emission=encoder_ref.encode(temp), majorant_emission=at["min_maj"][1])
    o.cam_fov, o.bounces, o.albedo, o.emission_scale = 40.0, 8, (0.7, 0.8, 0.9), 50.0
    return o


def ulps(a, b):
    b32 = np.asarray(b, np.float32)
    return np.abs(np.asarray(a, np.float64) - b32.astype(np.float64)) / np.spacing(np.abs(b32)).astype(np.float64)


def main():
    L = ob.lib()
    res = {"images": {}, "probes": {}}
    for name, m in META["images"].items():
        if name == "emission_spec":
            o = emission_scene(W, H)
        else:
            o = scenes.oracle_scene(m["config"], W, H)
            if m["white_env"]:
                o.set_envmap(np.ones((1, 1, 3), np.float32))
            o.integrator = m.get("integrator", 0)
        res["images"][name] = image_metrics(G["img_" + name], o.render(SPP))
    o = scenes.oracle_scene("c2", W, H)
    o.set_envmap(np.ones((1, 1, 3), np.float32))
    res["images"]["c2_white_driver_atlas_rgtc1"] = image_metrics(G["img_c2_white_driver_atlas_rgtc1"], o.render(SPP))

    tm = {}
    for key, (e, gm) in (("tonemap_e3_g2", (3.0, 2.0)), ("tonemap_e5_g22", (5.0, 2.2))):
        t = G["img_c2_hdr_spec"].copy()
        L.orc_tonemap(t.ctypes.data_as(C.POINTER(C.c_float)), W, H, C.c_float(e), C.c_float(gm))
        tm[key] = float(np.abs(t - G[key]).max())
    res["tonemap_max_abs"] = tm
    o = scenes.oracle_scene("c2", W, H)
    p, s = o.params(), o.scene()
    lv = ob.impmap_levels(o.impmap)
    imp = {}
    for k in range(3, 10):
        imp["level%d_max_rel" % k] = float((np.abs(lv[k] - G["impmap_level%d" % k]) / np.abs(G["impmap_level%d" % k])).max())
    imp["level0_rowsum_max_rel"] = float((np.abs(lv[0].astype(np.float64).sum(1) - G["impmap_level0_rowsum"]) / G["impmap_level0_rowsum"]).max())
    imp["level0_colsum_max_rel"] = float((np.abs(lv[0].astype(np.float64).sum(0) - G["impmap_level0_colsum"]) / G["impmap_level0_colsum"]).max())
    imp["level0_patch_max_rel"] = float((np.abs(lv[0][200:232, 300:332] - G["impmap_level0_patch"]) / G["impmap_level0_patch"]).max())
    res["impmap"] = imp

    P = res["probes"]
    L.orc_rng.restype = C.c_float
    L.orc_lookup_density_brick.restype = C.c_float
    L.orc_lookup_majorant_raw.restype = C.c_float
    L.orc_phase_hg.restype = C.c_float
    L.orc_phase_hg.argtypes = [C.c_float, C.c_float]
    L.orc_sample_phase_hg.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_void_p]
    L.orc_sample_environment.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_void_p, C.c_void_p]
    L.orc_transmittance.restype = C.c_float
    L.orc_transmittance.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.orc_math.restype = C.c_float
    L.orc_math.argtypes = [C.c_int, C.c_float, C.c_float]
    # 0: tea + rng
    a, b = G["probe0_in"], G["probe0_out"]
    au, bu = a.view(np.uint32), b.view(np.uint32)
    bad = 0
    for i in range(a.shape[0]):
        st = C.c_uint32(L.orc_tea(int(au[i, 0]), int(au[i, 1]), 32))
        ok = bu[i, 0] == st.value
        d = np.float32([L.orc_rng(C.byref(st)) for _ in range(3)])
        bad += not (ok and np.array_equal(d, b[i, 1:4]) and bu[i, 4] == st.value)
    P["tea_rng_mismatches"] = int(bad)
    # 1: brick fetches
    a, b = G["probe1_in"], G["probe1_out"]
    bad, dens_ulp = 0, 0.0
    for i in range(a.shape[0]):
        x, y, z = [int(np.floor(v)) for v in a[i, :3]]
        d = np.float32(L.orc_lookup_density_brick(C.byref(o.density.c), x, y, z))
        v = [np.float32(p.vol_density_scale) * np.float32(L.orc_lookup_majorant_raw(C.byref(o.density.c), x, y, z, k)) for k in range(4)]
        bad += not all(np.float32(v[k]) == b[i, 1 + k] for k in range(4))
        dens_ulp = max(dens_ulp, float(ulps(b[i, 0], d)) if d != b[i, 0] else 0.0)
    P["majorant_fetch_mismatches"] = int(bad)
    P["density_fetch_max_ulp"] = dens_ulp
    # 2: sample_environment
    a, b = G["probe2_in"], G["probe2_out"]
    wi = np.zeros(3, np.float32); lp = np.zeros(4, np.float32)
    dw, dl, dp = [], [], []
    for i in range(a.shape[0]):
        L.orc_sample_environment(C.byref(p), C.byref(s), float(a[i, 0]), float(a[i, 1]), wi.ctypes.data, lp.ctypes.data)
        dw.append(np.abs(wi - b[i, 4:7]).max())
        dl.append(np.abs(lp[:3] - b[i, :3]).max() / max(np.abs(b[i, :3]).max(), 1e-9))
        dp.append(abs(lp[3] - b[i, 3]) / max(abs(b[i, 3]), 1e-12))
    P["sample_environment"] = dict(w_i_max_abs=float(max(dw)), Le_max_rel=float(max(dl)), Le_median_rel=float(np.median(dl)), pdf_max_rel=float(max(dp)))
    # 3: phase
    a, b = G["probe3_in"], G["probe3_out"]
    e1, e2 = [], []
    o3 = np.zeros(3, np.float32)
    for i in range(a.shape[0]):
        e1.append(abs(L.orc_phase_hg(float(a[i, 3]), float(a[i, 4])) - b[i, 0]) / abs(b[i, 0]))
        dd = np.ascontiguousarray(a[i, :3])
        L.orc_sample_phase_hg(dd.ctypes.data, float(a[i, 4]), float(a[i, 5]), float(a[i, 6]), o3.ctypes.data)
        e2.append(np.abs(o3 - b[i, 4:7]).max())
    P["phase"] = dict(phase_hg_max_rel=float(max(e1)), sample_max_abs=float(max(e2)))
    # 6: transmittanceDDA with its RNG stream
    a, b = G["probe6_in"], G["probe6_out"]
    same_rng, exact, close = 0, 0, 0
    for i in range(a.shape[0]):
        sd = C.c_uint32(int(a.view(np.uint32)[i, 3]))
        pos = np.ascontiguousarray(a[i, :3]); dr = np.ascontiguousarray(a[i, 4:7])
        t = np.float32(L.orc_transmittance(C.byref(p), C.byref(s), pos.ctypes.data, dr.ctypes.data, C.byref(sd)))
        r = sd.value == b.view(np.uint32)[i, 1]
        same_rng += r; exact += (r and t == b[i, 0]); close += (r and abs(t - b[i, 0]) <= 1e-5 * max(abs(b[i, 0]), 1e-3))
    n = a.shape[0]
    P["transmittanceDDA"] = dict(same_rng_state=same_rng / n, identical=exact / n, within_1e5=close / n)
    # 4: view_dir (camera ray)
    a, b = G["probe4_in"], G["probe4_out"]
    L.orc_view_dir.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_void_p]
    o3 = np.zeros(3, np.float32)
    e = []
    for i in range(a.shape[0]):
        L.orc_view_dir(C.byref(p), int(a[i, 0]), int(a[i, 1]), int(a[i, 2]), int(a[i, 3]), float(a[i, 4]), float(a[i, 5]), o3.ctypes.data)
        e.append(np.abs(o3 - b[i, :3]).max())
    P["view_dir_max_abs"] = float(max(e))
    # 5: intersect_box
    a, b = G["probe5_in"], G["probe5_out"]
    L.orc_intersect_box.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    nf = np.zeros(2, np.float32)
    flag_bad, e = 0, []
    for i in range(a.shape[0]):
        pos = np.ascontiguousarray(a[i, :3]); dr = np.ascontiguousarray(a[i, 4:7])
        hit = L.orc_intersect_box(C.byref(p), pos.ctypes.data, dr.ctypes.data, nf.ctypes.data)
        flag_bad += int(bool(hit) != bool(b[i, 0] > 0.5))
        if hit and b[i, 0] > 0.5:
            e.append(float(np.max(np.abs(nf - b[i, 1:3]) / np.maximum(np.abs(b[i, 1:3]), 1e-3))))
    P["intersect_box"] = dict(flag_mismatches=flag_bad, near_far_max_rel=float(max(e)), hits=len(e))
    # 9: sample_volumeDDA (one camera segment) with its RNG stream
    a, b = G["probe9_in"], G["probe9_out"]
    L.orc_sample_volume.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    o4 = np.zeros(4, np.float32)
    same_rng = same_flag = exact = close = 0
    for i in range(a.shape[0]):
        sd = C.c_uint32(int(a.view(np.uint32)[i, 3]))
        pos = np.ascontiguousarray(a[i, :3]); dr = np.ascontiguousarray(a[i, 4:7])
        real = L.orc_sample_volume(C.byref(p), C.byref(s), pos.ctypes.data, dr.ctypes.data, C.byref(sd), o4.ctypes.data)
        r = sd.value == b.view(np.uint32)[i, 2]
        f = bool(real) == bool(b[i, 0] > 0.5)
        same_rng += r; same_flag += f
        if r and f:
            tt = o4[0] == b[i, 1] or not real                    # t is only meaningful after a real collision
            exact += bool(tt and np.array_equal(o4[1:], b[i, 4:7]))
            close += bool((not real or abs(o4[0] - b[i, 1]) <= 1e-5 * max(abs(b[i, 1]), 1e-3)) and np.abs(o4[1:] - b[i, 4:7]).max() <= 1e-6)
    n = a.shape[0]
    P["sample_volumeDDA"] = dict(same_rng_state=same_rng / n, same_flag=same_flag / n, identical=exact / n, within_1e5=close / n)
    # 7: the driver's built-ins against the specification
    a, b = G["probe7_in"], G["probe7_out"]
    spec = [("log", 0, 0, None, 0), ("sin", 1, 1, None, 1), ("cos", 2, 1, None, 2), ("acos", 4, 2, None, 3), ("atan2", 5, 3, 4, 4), ("exp", 6, 5, None, 5), ("pow", 7, 6, 7, 6)]
    mm = {}
    for nm, fn, ia, ib, io in spec:
        mine = np.float32([L.orc_math(fn, float(a[i, ia]), float(a[i, ib]) if ib is not None else 0.0) for i in range(a.shape[0])])
        mm[nm] = dict(max_ulp=float(ulps(b[:, io], mine).max()), max_rel=float((np.abs(b[:, io] - mine) / np.abs(mine).clip(1e-30)).max()))
    P["driver_builtins_vs_spec"] = mm
    # 8: bilinear environment fetch
    a, b = G["probe8_in"], G["probe8_out"]
    rgb = np.zeros(3, np.float32)
    L.orc_env_texture.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_void_p]
    e = []
    for i in range(a.shape[0]):
        L.orc_env_texture(o.env_tex.ctypes.data, o.env_tex.shape[1], o.env_tex.shape[0], float(a[i, 0]), float(a[i, 1]), rgb.ctypes.data)
        e.append(np.abs(rgb - b[i, :3]).max() / max(np.abs(b[i, :3]).max(), 1e-9))
    P["env_texture"] = dict(max_rel=float(max(e)), median_rel=float(np.median(e)))
    # round-2 goldens: the same images at 1024 spp (north star: 1e-3 relative L2), and the ray-marching trackers (integrator 3)
    G2 = np.load(os.path.join(HERE, "golden", "glsl_golden_r2.npz"))
    M2 = json.load(open(os.path.join(HERE, "golden", "glsl_golden_r2.json")))
    res["r2"] = {}
    for name, m in M2["images"].items():
        if m["config"] == "emission":
            o = emission_scene(W, H)
        else:
            o = scenes.oracle_scene(m["config"], W, H)
            if m["white_env"]:
                o.set_envmap(np.ones((1, 1, 3), np.float32))
        o.integrator = m.get("integrator", 0)
        res["r2"][name] = image_metrics(G2[name], o.render(m["spp"]))
    # round-3 goldens: the scenes pinned so far only with the specification's log / acos / atan spliced in, rendered from the reference's
    # UNMODIFIED kernel text (the driver's built-ins), 8 and 1024 spp (make_golden_glsl.py --r3)
    G3 = np.load(os.path.join(HERE, "golden", "glsl_golden_r3.npz"))
    M3 = json.load(open(os.path.join(HERE, "golden", "glsl_golden_r3.json")))
    res["r3"] = {}
    for name, m in M3["images"].items():
        o = emission_scene(W, H) if m["config"] == "emission" else scenes.oracle_scene(m["config"], W, H)
        res["r3"][name] = image_metrics(G3[name], o.render(m["spp"]))
    # round-5 goldens: three more views rendered by the reference's kernels with uniform values derived INDEPENDENTLY of this oracle
    # (tests/golden/host_rows.py, make_golden_glsl.py --r5): the oracle's own host rows (unit cube, AABB + crop, lookAt, env rotation, density scale)
    # must lead to the same frames, and its uniform values must be the hand-derived ones
    G5 = np.load(os.path.join(HERE, "golden", "glsl_golden_r5.npz"))
    M5 = json.load(open(os.path.join(HERE, "golden", "glsl_golden_r5.json")))
    res["r5"] = {}
    for name in M5["scenes"]:
        o = scenes.configure_r5(ob.OracleRenderer(W, H), name, True)
        p5 = o.params()
        worst = 0.0
        for key in G5.files:
            if key.startswith("u_%s_" % name):
                field = key[len("u_%s_" % name):]
                mine = np.asarray(getattr(p5, field), np.float64).reshape(-1)
                want = G5[key].astype(np.float64).reshape(-1)
                worst = max(worst, float((np.abs(mine - want) / np.maximum(np.abs(want), 1.0)).max()))      # absolute below 1, relative above
        res["r5"][name] = dict(uniform_max_rel=worst, img=image_metrics(G5["img_" + name], o.render(SPP)))
        o = scenes.configure_r5(ob.OracleRenderer(W, H), name, True)
        res["r5"][name]["hi"] = image_metrics(G5["hi_" + name], o.render(M5["hi_spp"]))
    print(json.dumps(res))


if __name__ == "__main__":
    main()
