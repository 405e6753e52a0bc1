"""Generates tests/golden/glsl_golden.npz: outputs of the REFERENCE'S OWN GLSL kernels, executed in the build container.

    python tests/golden/make_golden_glsl.py          (build container only: needs /root/reference and Mesa's swrast_dri.so)

The reference's hot path is GLSL (shader/pathtracer_brick*.glsl, common.glsl, env_setup.glsl, tonemap.glsl).  Its host
program cannot be built here (cppgl, voldata, GLFW, imgui are not vendored), but Mesa's software rasteriser can run the
kernels: oracle/glref (glref.c + binding.py) opens an OpenGL 4.5 context on llvmpipe, compiles the shader files where
they lie under /root/reference/shader, sets up textures and uniforms the way src/renderer.cpp / src/environment.cpp do,
dispatches once per sample and reads the image back.  This script stores those outputs -- data, not source -- as the
golden vectors that tests/test_glsl_pin.py checks the CPU oracle against (no GL, no /root/reference needed at test time).

Two ways the kernels are run, both stored:
  "driver"  -- untouched text (apart from one NVIDIA-only `bvec || bvec` expression, see binding._portable), the driver's
               built-in log/acos/atan (llvmpipe: 86 ulp, 2e-4, 2e-5 relative -- GLSL leaves their precision open);
  "spec"    -- the arithmetic specification's log/acos/atan spliced in through #define (oracle/glref/spec_math.glsl).
One deliberate deviation from the reference's GL calls: the brick atlas is uploaded as GL_R8 instead of the generic
GL_COMPRESSED_RED, for which Mesa picks lossy RGTC1 (see glref.c); `atlas_rgtc1_*` records what that choice costs.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]

import scenes  # noqa: E402
from oracle.glref import binding as gb  # noqa: E402

W, H, SPP = 64, 48, 8
IMAGES = {      # name -> (config, constant white environment?, spec math?)
    "c2_white_driver": ("c2", True, False),
    "c2_hdr_driver": ("c2", False, False),
    "c2_hdr_spec": ("c2", False, True),
    "readme_hdr_spec": ("readme", False, True),
    "c3_tf_spec": ("c3", False, True),
    "c1_hdr_spec": ("c1", False, True),
}


def emission_scene(w, h):
    """Small synthetic density + temperature grids (numpy reference encoder): the emission path of common.glsl:314-328."""
    import encoder_ref
    from oracle import binding as ob
    dens = scenes.synthetic_density(40)
    temp = np.clip(dens * 0.2 + 0.1 * scenes.synthetic_density(40, seed=99), 0, None).astype(np.float32)
    at = encoder_ref.encode_arrays(temp)
    o = ob.OracleRenderer(w, h)
    o.load_envmap(scenes.HDR)
    o.set_volume(encoder_ref.encode(dens), emission=encoder_ref.encode(temp), majorant_emission=at["min_maj"][1])
    o.cam_fov, o.bounces, o.albedo, o.emission_scale = 40.0, 8, (0.7, 0.8, 0.9), 50.0
    return o


def main():
    out = {}
    meta = {"width": W, "height": H, "spp": SPP, "gl": None, "images": {}, "probes": {}}
    for name, (cfg, white, spec) in IMAGES.items():
        o = scenes.oracle_scene(cfg, W, H)
        if white:
            o.set_envmap(np.ones((1, 1, 3), np.float32))
        g = gb.GLSLReference(o, spec_math=spec)
        meta["gl"] = g.info
        out["img_" + name] = g.render(SPP)
        meta["images"][name] = dict(config=cfg, white_env=white, spec_math=spec)
        print(name, "done", flush=True)
    o = emission_scene(W, H)
    out["img_emission_spec"] = gb.GLSLReference(o, spec_math=True).render(SPP)
    meta["images"]["emission_spec"] = dict(config="synthetic 40^3 density + temperature (tests/test_gpu_parity.py emission scene)", white_env=False, spec_math=True)
    # kernels the reference contains but does not build (binding._variant_program): global-majorant trackers = trace_path without
    # USE_DDA (the product's integrator 1), direct_volume_rendering (integrator 2)
    for name, cfg, variant in (("c2_global_spec", "c2", "global"), ("c3_global_spec", "c3", "global"), ("c3_dvr_spec", "c3", "dvr")):
        o = scenes.oracle_scene(cfg, W, H)
        out["img_" + name] = gb.GLSLReference(o, spec_math=True).render(SPP, variant=variant)
        meta["images"][name] = dict(config=cfg, white_env=False, spec_math=True, integrator=1 if variant == "global" else 2)
    # what Mesa's choice for GL_COMPRESSED_RED (RGTC1) does to the image
    o = scenes.oracle_scene("c2", W, H)
    o.set_envmap(np.ones((1, 1, 3), np.float32))
    out["img_c2_white_driver_atlas_rgtc1"] = gb.GLSLReference(o, literal_compressed_atlas=True).render(SPP)

    # tonemap.glsl on the c2 render (README settings exposure 3, gamma 2; defaults 5 / 2.2)
    g2 = gb.GLSLReference(scenes.oracle_scene("c2", W, H))
    out["tonemap_e3_g2"] = g2.tonemap(out["img_c2_hdr_spec"], 3.0, 2.0)
    out["tonemap_e5_g22"] = g2.tonemap(out["img_c2_hdr_spec"], 5.0, 2.2)
    # importance map of env_setup.glsl + glGenerateMipmap (environment.cpp:11-37): levels 3..9 whole, level 0 as row/column sums
    o = scenes.oracle_scene("c2", W, H)
    g = gb.GLSLReference(o)
    lv = g.impmap_levels()
    for k in range(3, 10):
        out["impmap_level%d" % k] = lv[k]
    out["impmap_level0_rowsum"] = lv[0].astype(np.float64).sum(1)
    out["impmap_level0_colsum"] = lv[0].astype(np.float64).sum(0)
    out["impmap_level0_patch"] = lv[0][200:232, 300:332].copy()

    # function-level probes (oracle/glref/probe.glsl calls into the reference's common.glsl)
    rs = np.random.RandomState(20260101)
    N = 256
    u32 = lambda n: rs.randint(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32)
    unit = lambda n: (lambda d: d / np.linalg.norm(d, axis=1)[:, None])(rs.normal(size=(n, 3)))
    nbv = np.array(o.density.n_bricks) * 8
    probes = {}
    a = np.zeros((N, 8), np.float32); a.view(np.uint32)[:, 0] = u32(N); a.view(np.uint32)[:, 1] = u32(N); probes[0] = a
    a = np.zeros((N, 8), np.float32); a[:, :3] = rs.uniform(0, 1, (N, 3)) * nbv; probes[1] = a                  # inside the grid (GL: outside is undefined)
    a = np.zeros((N, 8), np.float32); a[:, :2] = rs.uniform(0, 1, (N, 2)); probes[2] = a
    a = np.zeros((N, 8), np.float32); a[:, :3] = unit(N); a[:, 3] = rs.uniform(-1, 1, N); a[:, 4] = rs.choice([0.0, 0.3, -0.5, 0.8], N); a[:, 5:7] = rs.uniform(0, 1, (N, 2)); probes[3] = a
    a = np.zeros((N, 8), np.float32); a[:, 0] = rs.randint(0, 1024, N); a[:, 1] = rs.randint(0, 768, N); a[:, 2] = 1024; a[:, 3] = 768; a[:, 4:6] = rs.uniform(0, 1, (N, 2)); probes[4] = a
    a = np.zeros((N, 8), np.float32); a[:, :3] = rs.uniform(-1.5, 1.5, (N, 3)); a[:, 4:7] = unit(N); probes[5] = a
    a = np.zeros((N, 8), np.float32); a[:, :3] = rs.uniform(-0.3, 0.3, (N, 3)); a.view(np.uint32)[:, 3] = u32(N); a[:, 4:7] = unit(N); probes[6] = a
    a = np.zeros((N, 8), np.float32); a[:, :3] = rs.uniform(-0.3, 0.3, (N, 3)); a.view(np.uint32)[:, 3] = u32(N); a[:, 4:7] = unit(N); probes[9] = a
    a = np.zeros((N, 8), np.float32)
    a[:, 0] = rs.uniform(1e-7, 1, N); a[:, 1] = rs.uniform(-7, 7, N); a[:, 2] = rs.uniform(-1, 1, N); a[:, 3] = rs.uniform(-2, 2, N)
    a[:, 4] = rs.uniform(-2, 2, N); a[:, 5] = rs.uniform(-10, 3, N); a[:, 6] = rs.uniform(0.01, 4, N); a[:, 7] = rs.uniform(0.1, 2.5, N)
    probes[7] = a
    a = np.zeros((N, 8), np.float32); a[:, :2] = rs.uniform(0, 1, (N, 2)); a[:, 4:7] = unit(N); probes[8] = a
    gs = gb.GLSLReference(o, spec_math=False)
    for mode, inp in probes.items():
        out["probe%d_in" % mode] = inp
        out["probe%d_out" % mode] = gs.probe(mode, inp)
    meta["probes"] = {"0": "tea + 3 LCG draws", "1": "lookup_density_brick, lookup_majorant mips 0..3", "2": "sample_environment", "3": "phase_henyey_greenstein, sample_phase_henyey_greenstein",
                      "4": "view_dir", "5": "intersect_box", "6": "transmittanceDDA", "9": "sample_volumeDDA", "7": "built-ins log sin cos acos atan exp pow sqrt as llvmpipe evaluates them",
                      "8": "texture(env_envmap), lookup_environment, pdf_environment", "scene": "c2 (smoke.brick + table_mountain hdr), 64x48"}
    np.savez_compressed(os.path.join(HERE, "glsl_golden.npz"), **out)
    json.dump(meta, open(os.path.join(HERE, "glsl_golden.json"), "w"), indent=1)
    print("wrote glsl_golden.npz (%d arrays), %s" % (len(out), meta["gl"]))


HI_SPP = 1024


def main_r2():
    """Round-2 additions, written to glsl_golden_r2.npz (glsl_golden.npz stays byte-identical):
      hi_<name>        the images of IMAGES / the emission scene at HI_SPP samples per pixel -- the sample count at which the
                       north star's "within 1e-3 relative L2 of the GLSL reference" is meaningful (at 8 spp one flipped
                       stochastic decision moves the norm by more than that);
      rm_<cfg>_spec    trace_path with the ray-marching trackers of common.glsl:506-566 (binding._variant_program "raymarch":
                       code the reference contains but calls from no kernel; the product's integrator 3), 8 spp."""
    out = {}
    meta = {"width": W, "height": H, "hi_spp": HI_SPP, "spp": SPP, "images": {}}
    for name, (cfg, white, spec) in IMAGES.items():
        o = scenes.oracle_scene(cfg, W, H)
        if white:
            o.set_envmap(np.ones((1, 1, 3), np.float32))
        out["hi_" + name] = gb.GLSLReference(o, spec_math=spec).render(HI_SPP)
        meta["images"]["hi_" + name] = dict(config=cfg, white_env=white, spec_math=spec, spp=HI_SPP)
        print("hi", name, "done", flush=True)
    out["hi_emission_spec"] = gb.GLSLReference(emission_scene(W, H), spec_math=True).render(HI_SPP)
    meta["images"]["hi_emission_spec"] = dict(config="emission", white_env=False, spec_math=True, spp=HI_SPP)
    for cfg in ("c2", "c3"):
        o = scenes.oracle_scene(cfg, W, H)
        out["rm_%s_spec" % cfg] = gb.GLSLReference(o, spec_math=True).render(SPP, variant="raymarch")
        meta["images"]["rm_%s_spec" % cfg] = dict(config=cfg, white_env=False, spec_math=True, spp=SPP, integrator=3)
    np.savez_compressed(os.path.join(HERE, "glsl_golden_r2.npz"), **out)
    json.dump(meta, open(os.path.join(HERE, "glsl_golden_r2.json"), "w"), indent=1)
    print("wrote glsl_golden_r2.npz (%d arrays)" % len(out))


R3_IMAGES = {      # name -> config: the reference's kernel text AS IT STANDS (no spec_math splice), the driver's log / acos / atan
    "c3_tf_driver": "c3",            # shader/pathtracer_brick_tf.glsl
    "c1_hdr_driver": "c1",
    "readme_hdr_driver": "readme",
    "emission_driver": None,          # emission_scene(): common.glsl:314-328
}


def main_r3():
    """Round-3 additions, written to glsl_golden_r3.npz (the earlier files stay byte-identical): the scenes that rounds 1-2 pinned
    only with the specification's log/acos/atan spliced into the reference's text, now rendered from the UNMODIFIED kernel text
    (only binding._portable's one NVIDIA-only `bvec || bvec` rewrite, without which Mesa does not compile it), at 8 and 1024 spp."""
    out = {}
    meta = {"width": W, "height": H, "spp": SPP, "hi_spp": HI_SPP, "images": {}}
    for name, cfg in R3_IMAGES.items():
        for tag, spp in (("img_", SPP), ("hi_", HI_SPP)):
            o = emission_scene(W, H) if cfg is None else scenes.oracle_scene(cfg, W, H)
            g = gb.GLSLReference(o, spec_math=False)
            meta["gl"] = g.info
            out[tag + name] = g.render(spp)
            meta["images"][tag + name] = dict(config=cfg or "emission", white_env=False, spec_math=False, spp=spp)
            print(tag + name, "done", flush=True)
    np.savez_compressed(os.path.join(HERE, "glsl_golden_r3.npz"), **out)
    json.dump(meta, open(os.path.join(HERE, "glsl_golden_r3.json"), "w"), indent=1)
    print("wrote glsl_golden_r3.npz (%d arrays)" % len(out))


def r5_scene(name, w=W, h=H):
    """config c2 with the fields of host_rows.R5_SCENES[name] set on an OracleRenderer (arrays for the GL objects; its params() is NOT what the GLSL gets)."""
    from oracle import binding as ob
    return scenes.configure_r5(ob.OracleRenderer(w, h), name, True)


def r5_hand_derived(o, name):
    """the uniforms of scene `name` from tests/golden/host_rows.py (numpy, glm's documented formulas): dict name -> value"""
    import host_rows as hr
    s = hr.R5_SCENES[name]
    g = o.density
    extent = getattr(g, "extent", None) or tuple(int(n) * 8 for n in g.n_bricks)
    return hr.derive(np.asarray(g.transform, np.float32).reshape(16), extent, g.min_maj[0], g.min_maj[1], s["cam_pos"], hr.scene_dir(s), s["cam_up"],
                     s.get("density"), s.get("env_rot"), s.get("vol_crop_min", (0, 0, 0)), s.get("vol_crop_max", (1, 1, 1)))


def main_r5():
    """Round-5 additions, written to glsl_golden_r5.npz: a second pin for the HOST rows.  Three further views of smoke.brick -- another
    camera (position, target, tilted up vector, field of view), a rotated and scaled environment, a cropped volume at another density scale -- rendered by the
    reference's kernels on llvmpipe with uniform values that do NOT come from the oracle: cam_transform, vol_bb_*, vol_density_transform and its inverse,
    vol_minorant / majorant / inv_majorant, vol_density_scale, env_transform and its inverse are derived in tests/golden/host_rows.py from glm's documented
    formulas and src/renderer.cpp:88-131,227-242, src/main.cpp:379-382,417-428.  The tests set the same scenes up through the product's C ABI setters
    (and the oracle's fields) and compare the frames; the derived values themselves are stored too (u_<scene>_<uniform>)."""
    sys.path.insert(0, HERE)
    import host_rows as hr
    out = {}
    meta = {"width": W, "height": H, "spp": SPP, "hi_spp": HI_SPP, "spec_math": True, "scenes": {k: {kk: (list(vv) if isinstance(vv, tuple) else vv) for kk, vv in v.items()} for k, v in hr.R5_SCENES.items()}}
    for name in hr.R5_SCENES:
        o = r5_scene(name)
        d = r5_hand_derived(o, name)
        p = o.params()
        for k, v in d.items():
            out["u_%s_%s" % (name, k)] = np.asarray(v, np.float32).reshape(-1)
            if hasattr(getattr(p, k), "__len__"):
                getattr(p, k)[:] = [float(x) for x in np.asarray(v, np.float32).reshape(-1)]
            else:
                setattr(p, k, float(v))
        g = gb.GLSLReference(o, spec_math=True)
        meta["gl"] = g.info
        out["img_" + name] = g.render(SPP, params=p)
        out["hi_" + name] = g.render(HI_SPP, params=p)
        print(name, "done", flush=True)
    np.savez_compressed(os.path.join(HERE, "glsl_golden_r5.npz"), **out)
    json.dump(meta, open(os.path.join(HERE, "glsl_golden_r5.json"), "w"), indent=1)
    print("wrote glsl_golden_r5.npz (%d arrays)" % len(out))


if __name__ == "__main__":
    if "--r5" in sys.argv:
        main_r5()
    elif "--r3" in sys.argv:
        main_r3()
    elif "--r2" in sys.argv:
        main_r2()
    else:
        main()
