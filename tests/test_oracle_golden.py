"""CPU: pins the oracle against everything the reference offers for this path:
known-answers on its data files, RNG known-answers, the LUT CDF fix-up, and its only output artefact
(imgs/example.jpg, committed box-downsampled as tests/golden/example_64.npy); plus analytic checks."""
import ctypes as C
import json
import os
import zlib

import numpy as np
import pytest

import scenes
from oracle import binding as ob

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
KA = json.load(open(os.path.join(GOLD, "known_answers.json")))


def test_rng_known_answers():
    L = ob.lib()
    for v0, v1, want in KA["rng"]["tea"]:
        assert L.orc_tea(v0, v1, 32) == want
    s = C.c_uint32(L.orc_tea(42, 1, 32))
    got = [L.orc_rng(C.byref(s)) for _ in range(4)]
    assert np.allclose(got, KA["rng"]["stream_from_tea_42_1"], atol=1e-10, rtol=0)
    assert s.value == KA["rng"]["end_state"]


def test_brick_loader_known_answers():
    g = ob.Grid.from_file(scenes.SMOKE)
    b = KA["brick"]
    assert os.path.getsize(scenes.SMOKE) == b["file_bytes"]
    assert list(g.n_bricks) == b["n_bricks"] and list(g.atlas_dim) == b["atlas_dim"]
    assert g.brick_counter == b["brick_counter"]
    assert np.allclose(g.min_maj, b["min_maj"]) and np.allclose(g.transform, b["transform"])
    assert [list(d) for d, _ in g.mips] == b["mip_dims"]
    assert int(g.atlas.astype(np.uint64).sum()) == b["atlas_byte_sum"]
    assert zlib.crc32(g.atlas.tobytes()) == b["atlas_crc32"]
    d = g.decode_dense()
    assert abs(float(d.astype(np.float64).sum()) - b["decoded_sum"]) < 1e-2
    assert float(d.max()) == pytest.approx(b["decoded_max"])
    assert abs(float((d != 0).mean()) - b["decoded_nonzero_fraction"]) < 1e-3
    # max == majorant and the mips are exact (min of mins, max of maxes) over 2x2x2 children
    assert float(d.max()) == pytest.approx(g.min_maj[1])
    L = ob.lib()
    rs = np.random.RandomState(1)
    for _ in range(500):
        x, y, z = int(rs.randint(0, 128)), int(rs.randint(0, 256)), int(rs.randint(0, 128))
        assert L.orc_lookup_density_brick(C.byref(g.c), x, y, z) == d[z, y, x]
        for mip in range(1, 4):
            m = L.orc_lookup_majorant_raw(C.byref(g.c), x, y, z, mip)
            s = 8 << mip
            x0, y0, z0 = x // s * s, y // s * s, z // s * s
            kids = [L.orc_lookup_majorant_raw(C.byref(g.c), x0 + dx, y0 + dy, z0 + dz, 0)
                    for dx in range(0, s, 8) for dy in range(0, s, 8) for dz in range(0, s, 8)]
            assert m == max(kids)
    # out-of-range fetches read 0
    assert L.orc_lookup_density_brick(C.byref(g.c), -1, 0, 0) == 0.0
    assert L.orc_lookup_density_brick(C.byref(g.c), 128, 0, 0) == 0.0


def test_hdr_loader_and_importance_pyramid():
    img = ob.load_hdr(scenes.HDR)
    h = KA["hdr"]
    assert list(img.shape[:2]) == h["shape"]
    assert np.allclose(img.mean((0, 1)), h["mean_rgb_f32"], rtol=1e-6)
    assert float(img.max()) == h["max"]
    assert list(np.unravel_index(img.max(2).argmax(), img.shape[:2])) == h["argmax_row_col"]
    lum = img @ np.array([0.212671, 0.715160, 0.072169], np.float32)
    assert abs(lum[:128].astype(np.float64).mean() - h["top_quarter_luma"]) < 1e-4
    assert abs(lum[-128:].astype(np.float64).mean() - h["bottom_quarter_luma"]) < 1e-4
    pyr = ob.build_impmap(np.ascontiguousarray(img[::-1]))
    lv = ob.impmap_levels(pyr)
    assert len(lv) == 10 and lv[0].shape == (512, 512) and lv[-1].shape == (1, 1)
    assert float(lv[-1][0, 0]) == pytest.approx(h["impmap_avg_w"], rel=1e-6)
    for a, b in zip(lv[:-1], lv[1:]):      # 2x2 box mips
        ref = ((a[0::2, 0::2] + a[0::2, 1::2]) + (a[1::2, 0::2] + a[1::2, 1::2])) * np.float32(0.25)
        assert np.array_equal(ref, b)
    # the sun is in the upper half of the image -> v > 0.5 in texture space
    assert np.unravel_index(lv[0].argmax(), lv[0].shape)[0] > 256


def test_lut_cdf_fixup():
    lut = ob.load_lut(scenes.LUT)
    assert lut.shape == (8, 4)
    fixed, ran = ob.lut_fixup(lut)
    assert ran
    assert np.allclose(fixed[:, 3], KA["lut_cdf_alpha"], atol=1e-7)
    assert np.array_equal(fixed[:, :3], lut[:, :3])
    mono = np.array([[0, 0, 0, 0], [1, 1, 1, 0.5], [1, 1, 1, 0.5], [1, 1, 1, 1]], np.float32)
    same, ran2 = ob.lut_fixup(mono)
    assert not ran2 and np.array_equal(same, mono)
    zeros, _ = ob.lut_fixup(np.array([[0, 0, 0, 0.0], [0, 0, 0, -0.0], [0, 0, 0, 0.0]], np.float32))
    assert np.array_equal(zeros, zeros)           # no NaN


def test_reference_example_image():
    """README.md:72-73 rendered by the oracle at 64x64 vs the reference's own imgs/example.jpg (box-downsampled)."""
    ref = np.load(os.path.join(GOLD, "example_64.npy")).astype(np.float32)
    r = scenes.oracle_scene("readme", 64, 64)
    r.render(48)
    tm = r.tonemapped()[::-1, :, :3]
    img8 = np.floor(np.clip(tm, 0, 1) * 255 + 0.5)
    mse = float(((img8 - ref) ** 2).mean())
    psnr = 10 * np.log10(255.0 ** 2 / mse)
    assert psnr > 33.0, psnr                      # measured 37.8 dB at 64 spp; a y-flip / wrong env rotation gives < 20 dB
    assert np.abs(img8.reshape(-1, 3).mean(0) - ref.reshape(-1, 3).mean(0)).max() < 2.0
    flipped = float(((img8[::-1] - ref) ** 2).mean())
    assert flipped > 4 * mse


def test_counters_match_survey():
    """SURVEY.md 8d measured N_dda 11.12+3.99, N_coll 5.24+2.71, N_nee 0.62 on the README parameters."""
    r = scenes.oracle_scene("readme", 64, 64)
    r.render(8)
    c = r.counters.as_dict()
    n = c["samples"]
    assert n == 64 * 64 * 8
    assert abs(c["n_dda_sv"] / n - 11.12) < 0.4 and abs(c["n_dda_tr"] / n - 3.99) < 0.3
    assert abs(c["n_coll_sv"] / n - 5.24) < 0.3 and abs(c["n_coll_tr"] / n - 2.71) < 0.3
    assert abs(c["n_nee"] / n - 0.62) < 0.05
    assert abs(c["n_primary_miss"] / n - 0.31) < 0.02


def _homogeneous(sigma, n=16):
    import encoder_ref
    return encoder_ref.encode(np.full((n, n, n), sigma, np.float32))


def test_transmittance_homogeneous_box():
    """Delta tracking through a constant medium: E[Tr] = exp(-sigma * d) (both trackers)."""
    g = _homogeneous(2.0)
    for integrator in (0, 1):
        r = ob.OracleRenderer(8, 8)
        r.set_volume(g)
        r.integrator = integrator
        r.density_scale = 3.0                    # world density (index extent 64 -> unit cube, --density sets)
        p, s = r.params(), r.scene()
        pos = np.array([-0.375, -0.375, -2.0], np.float32)   # through the middle of the filled 16^3 corner block
        d = np.array([0.0, 0.0, 1.0], np.float32)
        acc = 0.0
        n = 4000
        for i in range(n):
            seed = C.c_uint32(ob.lib().orc_tea(i, 7, 32))
            acc += ob.lib().orc_transmittance(C.byref(p), C.byref(s), ob.fptr(pos), ob.fptr(d), C.byref(seed))
        # path length through the unit-ish cube: bb from -0.125 to 0.125 for a 16^3 data block in a 64^3 brick grid? use bb
        length = p.vol_bb_max[2] - p.vol_bb_min[2]
        sigma_world = 2.0 * 3.0
        expect = np.exp(-sigma_world * length * (16 / 64))     # only 16 of the 64 index cells along z are filled
        assert abs(acc / n - expect) < 0.03, (integrator, acc / n, expect)


def test_phase_function_moments():
    L = ob.lib()
    rs = np.random.RandomState(3)
    d = np.array([0.3, -0.5, 0.81], np.float32)
    d /= np.linalg.norm(d)
    for g in (0.0, 0.3, -0.6):
        cos = []
        out = np.zeros(3, np.float32)
        for _ in range(4000):
            L.orc_sample_phase_hg(ob.fptr(d), g, float(rs.rand()), float(rs.rand()), ob.fptr(out))
            assert abs(np.linalg.norm(out) - 1) < 1e-5
            cos.append(float(out @ d))
        # the reference's convention: pdf is evaluated with cos_t = dot(-dir, w) (common.glsl:618,640), sampler mean cosine = -g... check sign empirically stable
        assert abs(abs(np.mean(cos)) - abs(g)) < 0.03
    # pdf integrates to 1 over the sphere
    mu = np.linspace(-1, 1, 20001)
    for g in (0.0, 0.3, 0.8):
        pdf = np.array([L.orc_phase_hg(float(m), g) for m in mu])
        assert abs(float(np.sum((pdf[1:] + pdf[:-1]) * 0.5 * np.diff(mu))) * 2 * np.pi - 1) < 1e-3


def test_environment_sampler_matches_importance():
    """sample_environment picks texel (x, y) of the 512^2 map with probability importance / sum."""
    r = scenes.oracle_scene("c1", 8, 8)
    p, s = r.params(), r.scene()
    lv0 = ob.impmap_levels(r.impmap)[0]
    rs = np.random.RandomState(5)
    n = 20000
    hist = np.zeros((8, 8))
    w_i = np.zeros(3, np.float32)
    le = np.zeros(4, np.float32)
    pdfs = []
    for _ in range(n):
        ob.lib().orc_sample_environment(C.byref(p), C.byref(s), float(rs.rand()), float(rs.rand()), ob.fptr(w_i), ob.fptr(le))
        assert abs(np.linalg.norm(w_i) - 1) < 1e-4
        th = np.arccos(np.clip(w_i[1], -1, 1))
        ph = np.arctan2(w_i[2], w_i[0])
        u, v = ph / (2 * np.pi) + 0.5, 1 - th / np.pi
        hist[min(7, int(v * 8)), min(7, int(u * 8))] += 1
        pdfs.append(le[3])
    want = lv0.reshape(8, 64, 8, 64).sum((1, 3))
    want = want / want.sum()
    assert np.abs(hist / n - want).max() < 0.02
    assert min(pdfs) > 0


def test_white_furnace():
    """albedo 1 + constant white environment: every path carries radiance 1 (MIS weights sum to one)."""
    g = _homogeneous(1.5)
    r = ob.OracleRenderer(16, 16)
    r.set_volume(g)
    r.set_envmap(np.ones((4, 8, 3), np.float32))
    r.albedo = (1.0, 1.0, 1.0)
    r.bounces = 1000
    r.cam_fov = 30.0
    r.density_scale = 20.0
    fb = r.render(64)
    assert abs(fb[..., :3].mean() - 1.0) < 0.03


def test_cloud_generator_equals_the_reference_encoder():
    """scenes.cloud_brick_arrays builds BASELINE configs[4]'s grid ('c5cloud') in brick form, block by block, with a vectorised restatement of
    the encoder's rules; on a grid small enough to hold densely it must equal encoder_ref.encode_arrays of the same voxels word for word --
    across block borders (128^3 = 8 blocks with halos, some of them skipped as empty) and for both grids."""
    import encoder_ref
    import scenes
    n = 128
    ad, at = scenes.cloud_brick_arrays(n)
    dd, dt = scenes.cloud_dense(n)
    assert dd.max() == 5.0 and (dt > 0).sum() <= (dd > 0).sum() and dt[dd == 0].max() == 0.0        # the temperature lives where the density does
    for a, d in ((ad, dd), (at, dt)):
        e = encoder_ref.encode_arrays(d)
        assert a["brick_counter"] == e["brick_counter"] > 0 and tuple(a["atlas_dim"]) == tuple(e["atlas_dim"]) and a["min_maj"] == e["min_maj"]
        for k in ("indirection", "rng", "atlas"):
            assert np.array_equal(np.asarray(a[k]), np.asarray(e[k])), k
        for (da, wa), (de, we) in zip(a["mips"], e["mips"]):
            assert tuple(da) == tuple(de) and np.array_equal(wa, we)
    # the occupancy falls towards the 1024^3 grid's 16.8 % as the surface-to-volume ratio does (asserted at full size by the GPU suite)
    a256, _ = scenes.cloud_brick_arrays(256)
    assert 0.15 < a256["brick_counter"] / 32 ** 3 < 0.35
