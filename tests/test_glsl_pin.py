"""CPU: pins the oracle against outputs of the REFERENCE'S OWN GLSL KERNELS.

tests/golden/glsl_golden.npz was produced in the build container by tests/golden/make_golden_glsl.py, which ran
shader/pathtracer_brick.glsl, pathtracer_brick_tf.glsl, env_setup.glsl and individual functions of common.glsl -- read
from /root/reference, compiled by Mesa -- on llvmpipe (oracle/glref).  Nothing here needs GL or /root/reference.

Two builds of the oracle are checked (tests/glsl_pin_worker.py, one process each):
  * the UNFUSED variant (oracle/_ref/liboracle_unfused.so, -DORC_UNFUSED: multiply-add never fused, llvmpipe's
    convention): agrees with the reference's kernels to ~1e-7 relative L2 -- the oracle's LOGIC is the reference's;
  * the standard oracle (fma where the specification says so -- the convention the HIP product shares): same images up to
    the handful of pixel-samples where a last-bit difference flips a stochastic decision.
GLSL leaves the precision of log/acos/atan to the driver; images tagged "spec" were rendered with the specification's
versions spliced into the reference's text, "driver" ones with llvmpipe's built-ins (whose deviation is recorded too).
"""
import json
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
UNFUSED = os.path.join(ROOT, "oracle", "_ref", "liboracle_unfused.so")


def _run(variant):
    env = dict(os.environ)
    if variant == "unfused":
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "unfused"])
        env["VOLREN_ORACLE_SO"] = UNFUSED
    else:
        env.pop("VOLREN_ORACLE_SO", None)
    out = subprocess.check_output([sys.executable, os.path.join(HERE, "glsl_pin_worker.py")], env=env)
    return json.loads(out.decode().strip().split("\n")[-1])


@pytest.fixture(scope="module")
def unfused():
    return _run("unfused")


@pytest.fixture(scope="module")
def standard():
    return _run("standard")


def test_unfused_oracle_reproduces_reference_kernels(unfused):
    im = unfused["images"]
    # driver built-ins, constant environment (no acos/atan on the path): every pixel of a 100-bounce render agrees
    assert im["c2_white_driver"]["rel_l2"] < 1e-6 and im["c2_white_driver"]["within_1e5"] == 1.0, im["c2_white_driver"]
    for name in ("c2_hdr_spec", "c1_hdr_spec", "readme_hdr_spec"):
        assert im[name]["rel_l2"] < 5e-6 and im[name]["within_1e5"] > 0.999, (name, im[name])
    assert im["c3_tf_spec"]["rel_l2"] < 5e-5 and im["c3_tf_spec"]["within_1e3"] > 0.999, im["c3_tf_spec"]       # transfer-function kernel
    assert im["emission_spec"]["rel_l2"] < 5e-4 and im["emission_spec"]["within_1e5"] > 0.998, im["emission_spec"]
    # code the reference contains but does not build: trace_path without USE_DDA (integrator 1), direct_volume_rendering (integrator 2; .w is ours)
    assert im["c2_global_spec"]["rel_l2"] < 5e-5 and im["c3_global_spec"]["rel_l2"] < 2e-4 and im["c3_dvr_spec"]["rel_l2"] < 2e-5, im
    # with the driver's own acos/atan (2e-4 / 2e-5 relative on llvmpipe) the environment lookups move by that much
    assert im["c2_hdr_driver"]["rel_l2"] < 2e-3 and im["c2_hdr_driver"]["within_1e3"] > 0.9, im["c2_hdr_driver"]
    # the north star's bar (1e-3 relative L2 against the GLSL reference) with room to spare
    assert max(im[n]["rel_l2"] for n in ("c2_white_driver", "c2_hdr_spec", "c1_hdr_spec", "readme_hdr_spec", "c3_tf_spec", "emission_spec")) < 1e-3


def test_standard_oracle_matches_up_to_stochastic_flips(standard):
    im = standard["images"]
    for name in ("c2_white_driver", "c2_hdr_spec", "c1_hdr_spec", "readme_hdr_spec", "emission_spec"):
        assert im[name]["within_1e5"] > 0.995, (name, im[name])              # >= 99.5 % of the pixels identical to 1e-5
        assert im[name]["rel_l2"] < 5e-2 and abs(im[name]["mean_ratio"] - 1.0) < 1e-3, (name, im[name])
    assert im["c3_tf_spec"]["within_1e3"] > 0.999 and im["c3_tf_spec"]["rel_l2"] < 1e-3, im["c3_tf_spec"]
    for name in ("c2_global_spec", "c3_global_spec", "c3_dvr_spec"):
        assert im[name]["within_1e3"] > 0.995 and im[name]["rel_l2"] < 5e-2, (name, im[name])


def test_functions_of_common_glsl(standard):
    p = standard["probes"]
    assert p["tea_rng_mismatches"] == 0                                       # tea(), rng(): bit for bit
    assert p["majorant_fetch_mismatches"] == 0                                # lookup_majorant on mips 0..3: bit for bit
    assert p["density_fetch_max_ulp"] <= 1.0                                  # unorm8 -> float is c/255 in the GL spec; llvmpipe is 1 ulp off on some codes
    t = p["transmittanceDDA"]
    assert t["same_rng_state"] == 1.0 and t["identical"] == 1.0, t            # a whole shadow segment: value and RNG end state
    assert p["phase"]["phase_hg_max_rel"] == 0.0 and p["phase"]["sample_max_abs"] < 3e-7, p["phase"]
    e = p["sample_environment"]
    assert e["pdf_max_rel"] < 1e-6 and e["w_i_max_abs"] < 2e-6 and e["Le_max_rel"] < 1e-4, e
    assert p["env_texture"]["max_rel"] < 1e-6, p["env_texture"]               # GL_LINEAR fetch of the RGB32F environment map
    for k, v in standard["impmap"].items():
        assert v < 1e-6, (k, v)                                               # env_setup.glsl + glGenerateMipmap
    for k, v in standard["tonemap_max_abs"].items():
        assert v < 2e-6, (k, v)                                               # tonemap.glsl (pow is the driver's: a few ulp)


def test_more_functions_of_common_glsl(standard, unfused):
    """Probes 4, 5 and 9 of the golden file: view_dir, intersect_box and a complete sample_volumeDDA camera segment."""
    for res in (standard, unfused):
        p = res["probes"]
        assert p["view_dir_max_abs"] < 3e-7, p["view_dir_max_abs"]
        b = p["intersect_box"]
        assert b["flag_mismatches"] == 0 and b["hits"] >= 5 and b["near_far_max_rel"] < 2e-6, b
    d = standard["probes"]["sample_volumeDDA"]
    # decisions and RNG end state of every segment identical; t / throughput to 1e-5 (the probes run with the driver's log, 86 ulp)
    assert d["same_flag"] == 1.0 and d["same_rng_state"] == 1.0 and d["within_1e5"] >= 0.99, d
    assert unfused["probes"]["sample_volumeDDA"]["same_rng_state"] >= 0.99, unfused["probes"]["sample_volumeDDA"]


def test_north_star_tolerance_at_1024_spp(standard, unfused):
    """"Output within 1e-3 relative L2 of the GLSL reference", at 1024 samples per pixel (glsl_golden_r2.npz): the standard
    oracle -- which the HIP kernels reproduce bit for bit -- and the unfused build."""
    for name in ("hi_c2_white_driver", "hi_c2_hdr_spec", "hi_c3_tf_spec", "hi_readme_hdr_spec", "hi_c1_hdr_spec", "hi_emission_spec"):
        assert standard["r2"][name]["rel_l2"] <= 1e-3, (name, standard["r2"][name])
        assert unfused["r2"][name]["rel_l2"] <= 1e-3, (name, unfused["r2"][name])
    assert standard["r2"]["hi_c2_hdr_driver"]["rel_l2"] <= 2e-3, standard["r2"]["hi_c2_hdr_driver"]      # the driver's own acos/atan: 2e-4 relative


def test_unmodified_kernel_text(standard, unfused):
    """Round 3 (glsl_golden_r3.npz): the transfer-function kernel (pathtracer_brick_tf.glsl), config c1, the README scene and the
    emission path rendered from the reference's kernel text AS IT STANDS -- the driver's log / acos / atan instead of the
    specification's spliced in.  llvmpipe's acos / atan are 2e-4 / 2e-5 relative, which is what the environment lookups of the three
    HDR scenes then differ by (measured 3.4e-4 ... 4.4e-4, at 8 and at 1024 spp alike: a systematic offset, not noise); the
    transfer-function scene (no environment lookups: show_environment = 0) agrees to 7e-6.  At 1024 spp every image is inside the
    north star's 1e-3, for the build without multiply-add fusion and for the standard oracle, which the HIP kernels equal bit for bit."""
    for res in (standard, unfused):
        for name in ("hi_c3_tf_driver", "hi_c1_hdr_driver", "hi_readme_hdr_driver", "hi_emission_driver"):
            assert res["r3"][name]["rel_l2"] <= 1e-3, (name, res["r3"][name])
    u = unfused["r3"]
    assert u["img_c3_tf_driver"]["rel_l2"] < 5e-5 and u["img_c3_tf_driver"]["within_1e5"] > 0.98, u["img_c3_tf_driver"]
    for name in ("img_c1_hdr_driver", "img_readme_hdr_driver", "img_emission_driver"):
        assert u[name]["rel_l2"] < 1e-3 and u[name]["within_1e3"] > 0.9 and abs(u[name]["mean_ratio"] - 1.0) < 5e-4, (name, u[name])
    s = standard["r3"]
    for name in ("img_c3_tf_driver", "img_c1_hdr_driver", "img_readme_hdr_driver", "img_emission_driver"):
        assert s[name]["within_1e3"] > 0.9 and s[name]["rel_l2"] < 5e-2 and abs(s[name]["mean_ratio"] - 1.0) < 1e-3, (name, s[name])      # 8 spp: a flipped path is a pixel


def test_host_rows_second_pin(standard, unfused):
    """Round 5 (glsl_golden_r5.npz; verdict r4 #5): SURVEY 8 rows a17 / a18 -- the host marshalling -- rested on one JPEG.  Three more views (another camera
    with a tilted up vector and another field of view; an environment rotated by 135 degrees at strength 2; a cropped volume at another density scale) were
    rendered by the reference's kernels on llvmpipe with uniforms derived in tests/golden/host_rows.py from glm's documented formulas, NOT by this oracle.
    The oracle's own uniform values equal the hand-derived ones to one rounding, and its frames -- set up through its fields only -- are the reference's."""
    for name in ("cam_b", "env_rot", "crop"):
        for res in (standard, unfused):
            r = res["r5"][name]
            assert r["uniform_max_rel"] < 3e-7, (name, r["uniform_max_rel"])          # lookAt / inverse: the last bit may differ, nothing else
            assert r["hi"]["rel_l2"] <= 1e-3, (name, r["hi"])                        # the north star's bar at 1024 spp
        u, s = unfused["r5"][name]["img"], standard["r5"][name]["img"]
        assert u["within_1e5"] > 0.99 and u["rel_l2"] < 2e-2 and abs(u["mean_ratio"] - 1.0) < 1e-3, (name, u)      # 8 spp: a flipped path is a pixel
        assert s["within_1e5"] > 0.99 and s["rel_l2"] < 5e-2 and abs(s["mean_ratio"] - 1.0) < 1e-3, (name, s)
    # the three views really differ from config c2's frame (a pin that any frame passes pins nothing)
    import numpy as np
    g5 = np.load(os.path.join(HERE, "golden", "glsl_golden_r5.npz"))
    g2 = np.load(os.path.join(HERE, "golden", "glsl_golden_r2.npz"))
    for name in ("cam_b", "env_rot", "crop"):
        a, b = g5["hi_" + name][..., :3].astype(np.float64), g2["hi_c2_hdr_spec"][..., :3].astype(np.float64)
        assert np.linalg.norm(a - b) / np.linalg.norm(b) > 0.05, name


def test_raymarch_trackers_reproduce_reference_text(standard, unfused):
    """trace_path with sample_volume_raymarch / transmittance_raymarch (common.glsl:506-566, integrator 3) against the
    reference's text run on llvmpipe."""
    for name in ("rm_c2_spec", "rm_c3_spec"):
        assert unfused["r2"][name]["rel_l2"] < 5e-5 and unfused["r2"][name]["within_1e5"] > 0.995, (name, unfused["r2"][name])
        assert standard["r2"][name]["within_1e5"] > 0.99 and standard["r2"][name]["rel_l2"] < 5e-2, (name, standard["r2"][name])


def test_recorded_precision_of_driver_builtins(standard):
    """What "the reference" is, in the last digits, depends on the GL driver: llvmpipe's sin/cos equal the specification's
    (both Cephes), its log/acos/atan are short polynomials.  Recorded so that the tolerances above can be read."""
    b = standard["probes"]["driver_builtins_vs_spec"]
    assert b["sin"]["max_ulp"] == 0.0 and b["cos"]["max_ulp"] == 0.0
    assert b["exp"]["max_ulp"] <= 16 and b["pow"]["max_ulp"] <= 16
    assert 1e-7 < b["log"]["max_rel"] < 1e-3 and 1e-6 < b["acos"]["max_rel"] < 1e-3 and 1e-7 < b["atan2"]["max_rel"] < 1e-4


def test_generic_compressed_atlas_is_a_driver_choice(standard):
    """renderer.cpp:200 asks for GL_COMPRESSED_RED; Mesa stores RGTC1 (lossy), a driver without 3D RGTC stores R8.  The
    oracle, like the product, implements the lossless outcome; the RGTC1 render differs visibly from it."""
    r = standard["images"]["c2_white_driver_atlas_rgtc1"]
    assert r["rel_l2"] > 5e-3 and abs(r["mean_ratio"] - 1.0) < 5e-3, r
