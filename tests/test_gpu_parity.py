"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.

Bar: the renderer's arithmetic is specified operation by operation (vr_math.h / oracle_math.h), so the HIP kernel
must reproduce the oracle BIT FOR BIT; the north-star tolerance (relative L2 <= 1e-3) is asserted as well so that a
failure report shows how far off a run is."""
import os

import numpy as np
import pytest

import scenes

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def _assert_same(hip, orc, what):
    rl2 = scenes.rel_l2(hip[..., :3], orc[..., :3])
    nbad = int((_bits(hip) != _bits(orc)).any(-1).sum())
    assert rl2 <= 1e-3, "%s: relative L2 %.3e > 1e-3 (%d pixels differ)" % (what, rl2, nbad)
    assert nbad == 0, "%s: %d pixels differ bitwise (relative L2 %.3e)" % (what, nbad, rl2)


def test_device_math_bit_exact():
    import volren_amd
    from oracle import binding as ob
    L = ob.lib()
    rs = np.random.RandomState(7)
    n = 20000
    cases = {
        0: (np.concatenate([1.0 - rs.randint(0, 1 << 24, n) / np.float32(1 << 24), rs.uniform(1e-30, 100, 2000)]).astype(np.float32), None),
        1: (rs.uniform(-7, 7, n).astype(np.float32), None),
        2: (rs.uniform(-7, 7, n).astype(np.float32), None),
        3: (rs.uniform(0.01, 1.5, n).astype(np.float32), None),
        4: (rs.uniform(-1.01, 1.01, n).astype(np.float32), None),
        5: (rs.uniform(-2, 2, n).astype(np.float32), rs.uniform(-2, 2, n).astype(np.float32)),
        6: (rs.uniform(-20, 20, n).astype(np.float32), None),
        7: (rs.uniform(0, 4, n).astype(np.float32), rs.uniform(0.2, 3, n).astype(np.float32)),
        8: (rs.uniform(-1, 1, n).astype(np.float32), None),
    }
    for fn, (a, b) in cases.items():
        b = b if b is not None else np.zeros_like(a)
        dev = volren_amd.math_probe(fn, a, b)
        ref = np.array([L.orc_math(fn, float(x), float(y)) for x, y in zip(a, b)], np.float32)
        bad = _bits(dev) != _bits(ref)
        bad &= ~(np.isnan(dev) & np.isnan(ref))
        assert not bad.any(), "math fn %d differs at %d inputs, e.g. %r -> dev %r ref %r" % (
            fn, bad.sum(), a[bad][:3], dev[bad][:3], ref[bad][:3])
    # IEEE divide / sqrt / fma / no-contraction on the device
    a = rs.uniform(-100, 100, n).astype(np.float32)
    b = rs.uniform(0.001, 100, n).astype(np.float32)
    assert np.array_equal(_bits(volren_amd.math_probe(9, a, b)), _bits(a / b))
    assert np.array_equal(_bits(volren_amd.math_probe(10, np.abs(a), b)), _bits(np.sqrt(np.abs(a))))
    assert np.array_equal(_bits(volren_amd.math_probe(14, a, b)), _bits((a * b).astype(np.float32) + a))
    halves = np.arange(65536, dtype=np.uint32)                                 # every binary16 value through the device's conversion
    dev = volren_amd.math_probe(15, halves.view(np.float32), halves.view(np.float32))
    ref = halves.astype(np.uint16).view(np.float16).astype(np.float32)
    same = (_bits(dev) == _bits(ref)) | (np.isnan(dev) & np.isnan(ref))
    assert same.all(), "half2float differs at %d codes" % (~same).sum()
    u8 = np.arange(256, dtype=np.float32)
    assert np.array_equal(_bits(volren_amd.math_probe(12, u8, u8)), _bits(u8 / np.float32(255)))
    # rcp_exact / rcp3_exact (v_rcp_f32 + Newton step + fix-up inside the verified exponent range, IEEE division outside it) == 1 / x:
    # random bit patterns of every class, every exponent's extreme mantissas, zeros, denormals, infinities, NaN.  The full 2^32 sweep is
    # tests/tools_rcp_exact.hip (profiles/r3f_*).
    bits = np.concatenate([rs.randint(0, 1 << 32, 200000, dtype=np.uint64).astype(np.uint32),
                           (np.arange(256, dtype=np.uint32)[:, None] << 23 | np.array([0, 1, 0x400000, 0x7FFFFE, 0x7FFFFF], np.uint32)[None, :]).reshape(-1),
                           (np.arange(256, dtype=np.uint32)[:, None] << 23 | np.array([0, 1, 0x7FFFFF], np.uint32)[None, :]).reshape(-1) | np.uint32(0x80000000)])
    x = bits.view(np.float32)
    with np.errstate(all="ignore"):
        want = (np.float32(1.0) / x).astype(np.float32)
    for fn, args in ((16, (x, x)), (17, (np.full_like(x, 3.0), x)), (17, (np.full_like(x, 1e-42), x))):
        dev = volren_amd.math_probe(fn, *args)
        same = (_bits(dev) == _bits(want)) | (np.isnan(dev) & np.isnan(want))
        assert same.all(), "rcp (probe %d) differs at %d inputs, e.g. %r" % (fn, (~same).sum(), x[~same][:4])


def test_impmap_matches_oracle():
    o = scenes.oracle_scene("c1", 16, 16)
    r = scenes.hip_scene("c1", 16, 16)
    dev = r.impmap()
    assert dev.shape == o.impmap.shape
    assert np.array_equal(_bits(dev), _bits(o.impmap))


@pytest.mark.parametrize("name", ["c1", "c3", "readme"])
def test_uniforms_match_oracle(name):
    import ctypes as C
    o = scenes.oracle_scene(name, 64, 48)
    r = scenes.hip_scene(name, 64, 48)
    got = r.uniforms_bytes()
    p = o.params()
    want = bytes((C.c_uint8 * C.sizeof(p)).from_buffer_copy(p))
    assert len(got) == len(want)
    assert got == want


@pytest.mark.parametrize("name,w,h,spp", [("c1", 256, 256, 16), ("c3", 96, 96, 8), ("readme", 96, 96, 8), ("c2", 64, 64, 32)])
def test_render_matches_oracle(name, w, h, spp):
    o = scenes.oracle_scene(name, w, h)
    r = scenes.hip_scene(name, w, h)
    r.render(spp)
    _assert_same(r.framebuffer(), o.render(spp), "%s %dx%d %dspp" % (name, w, h, spp))


@pytest.mark.parametrize("name", ["c1", "c3"])
def test_global_majorant_tracking_variant(name):
    o = scenes.oracle_scene(name, 64, 64)
    o.integrator = 1
    r = scenes.hip_scene(name, 64, 64)
    r.integrator = 1
    r.render(8)
    _assert_same(r.framebuffer(), o.render(8), "global-majorant tracking " + name)


def test_direct_volume_rendering_integrator():
    o = scenes.oracle_scene("c3", 80, 64)
    o.integrator = 2
    r = scenes.hip_scene("c3", 80, 64)
    r.integrator = 2
    r.render(6)
    _assert_same(r.framebuffer(), o.render(6), "direct volume rendering")
    n = scenes.hip_scene("c1", 16, 16)          # no transfer function bound: refused, not silently path traced
    n.integrator = 2
    with pytest.raises(Exception):
        n.render(1)


def test_volume_animation_folder(tmp_path):
    """load_volume on a directory = animation frames in alphanumerical order (main.cpp:40-42); grid_frame_counter selects
    the frame (main.cpp:530-532).  Frames are written with the product's own dense->.brick encoder."""
    import volren_amd
    from oracle import binding as ob
    lib = volren_amd.load()
    frames = [scenes.synthetic_density(40, seed=s) for s in (5, 6)]
    for i, f in enumerate(frames):
        assert lib.vr_write_brick_from_dense(f.ctypes.data, 40, 40, 40, None, str(tmp_path / ("f%03d.brick" % i)).encode()) == 0
    r = volren_amd.Renderer(48, 48)
    r.load_envmap(scenes.HDR)
    r.load_volume(tmp_path)
    assert r.n_grid_frames == 2
    r.cam_fov, r.bounces = 40.0, 6
    for i in range(2):
        r.grid_frame_counter = i
        r.reset()
        r.render(4)
        o = ob.OracleRenderer(48, 48)
        o.load_envmap(scenes.HDR)
        o.load_volume(str(tmp_path / ("f%03d.brick" % i)))
        o.cam_fov, o.bounces = 40.0, 6
        _assert_same(r.framebuffer(), o.render(4), "animation frame %d" % i)


FLAG_VARIANTS = {
    # name: (base config, {field: value}) -- the renderer flags of main.cpp:360-435, one deviation at a time
    "env_hide": ("c1", dict(show_environment=False)),
    "albedo_zero": ("c1", dict(albedo=(0.0, 0.0, 0.0))),
    "albedo_rgb_one_bounce": ("c1", dict(albedo=(1.0, 0.5, 0.25), bounces=1)),
    "phase_backward": ("c2", dict(phase=-0.7)),
    "dense_medium": ("c1", dict(density_scale=2000.0, bounces=32)),
    "thin_medium": ("c1", dict(density_scale=3.0)),
    "crop": ("c1", dict(vol_clip_min=(0.2, 0.1, 0.0), vol_clip_max=(0.9, 0.6, 1.0))),
    "camera_inside_wide_fov": ("c1", dict(cam_pos=(0.0, -0.2, 0.05), cam_dir=(0.3, 1.0, 0.1), cam_fov=95.0)),
    "camera_top_down": ("c1", dict(cam_pos=(0.0, 2.0, 0.0), cam_dir=(0.0, -1.0, 0.001), cam_up=(0.0, 0.0, 1.0))),
    # 36 000 volume widths away: the camera segments start at |ipos| > 2^20 voxels -- not "clean" (vr_trace.h seg_clean): wavefronts that hold one run the hot
    # pair in its general form, the others in the clean form
    "camera_very_far": ("c2", dict(cam_pos=(3.0e4, 0.5e4, 2.0e4), cam_dir=(-0.8241634368896484, -0.1373605728149414, -0.5494422912597656), cam_fov=0.003)),
    "env_strong_rotated": ("c1", dict(env_strength=7.5, env_rot=123.0)),
    "seed": ("c1", dict(seed=-7)),
    "tf_window": ("c3", dict(tf_window_left=0.05, tf_window_width=0.4)),
    "tf_with_env": ("c3", dict(show_environment=True, albedo=(0.95, 0.95, 0.95))),
}


@pytest.mark.parametrize("variant", sorted(FLAG_VARIANTS))
def test_renderer_flags_match_oracle(variant):
    base, fields = FLAG_VARIANTS[variant]
    o = scenes.oracle_scene(base, 56, 40)
    r = scenes.hip_scene(base, 56, 40)
    for k, v in fields.items():
        if k == "env_rot":
            o.set_env_rot(v)
            r.env_rot = v
        else:
            setattr(o, k, v)
            setattr(r, k, v)
    r.render(5)
    _assert_same(r.framebuffer(), o.render(5), variant)


def test_zero_samples_and_tiny_frames():
    r = scenes.hip_scene("c1", 1, 1)
    r.render(3)
    o = scenes.oracle_scene("c1", 1, 1)
    _assert_same(r.framebuffer(), o.render(3), "1x1")
    z = scenes.hip_scene("c1", 24, 24)
    z.sppx = 0
    z.render(0)                     # nothing to do: sample stays 0, framebuffer stays zero
    assert z.sample == 0 and not z.framebuffer().any()


def test_trace_protocol_equals_fused_render():
    """trace() x N (the reference protocol) == render(N) == render(a) + render(b); consecutive trace() calls are launched together."""
    a = scenes.hip_scene("c1", 48, 48)
    for k in range(6):
        a.trace()
        assert a.sample == k + 1 and a.pending_samples == k + 1      # recorded, not launched
    a.synchronize()
    assert a.pending_samples == 0 and a.last_launches == 1           # ONE launch for the six calls
    b = scenes.hip_scene("c1", 48, 48)
    b.render(6)
    c = scenes.hip_scene("c1", 48, 48)
    c.render(2)
    c.render(4)
    d = scenes.hip_scene("c1", 48, 48)                               # round 4's protocol: one launch per call
    d.coalesce_trace = 0
    for _ in range(6):
        d.trace()
        assert d.pending_samples == 0
    fa, fb, fc, fd = a.framebuffer(), b.framebuffer(), c.framebuffer(), d.framebuffer()
    assert a.sample == 6 and b.sample == 6 and c.sample == 6 and d.sample == 6
    assert np.array_equal(_bits(fa), _bits(fb))
    assert np.array_equal(_bits(fa), _bits(fc))
    assert np.array_equal(_bits(fa), _bits(fd))


def test_coalesced_trace_sees_every_change_between_two_calls():
    """The reference issues a dispatch per trace() (src/renderer.cpp:78-145), so a field changed between two calls applies to the later samples only.
    Coalesced calls must give the same frame as one launch per call (coalesce_trace = 0) and as the oracle driven the same way -- for changes of
    plain fields, of the camera, of the environment's strength, of the sample counter, and for a transfer function replaced between two calls."""
    w, h = 56, 40
    lut2 = np.array([[0.9, 0.2, 0.1, 0.0], [0.1, 0.8, 0.3, 0.4], [0.2, 0.3, 0.9, 1.0]], np.float32)

    def drive(r, is_oracle):
        frames = []
        if is_oracle:                                                 # the oracle's protocol object has render(n) and .fb
            r.trace = lambda: r.render(1)
            r.framebuffer = lambda: r.fb
            set_tf = r.set_transferfunc
            r.set_transferfunc = lambda lut: set_tf(lut) if lut is not None else (setattr(r, "lut", None), setattr(r, "sample", 0))
        def step(n):
            for _ in range(n):
                r.trace()
        step(3)
        r.albedo = [0.5, 0.6, 0.7]; step(2)                           # plain field
        r.cam_pos = [0.9, 0.2, 1.1]; r.cam_dir = [-0.6, -0.1, -0.75]; step(2)     # camera
        r.env_strength = 2.0; step(1)                                 # a field of the shared Environment object
        r.density_scale = float(r.density_scale) * 0.5; step(2)       # majorant table is rebuilt: the recorded samples must not see the new one
        frames.append(r.framebuffer().copy())                         # an observation flushes
        step(2)
        r.sample = 0; step(3)                                         # reset(): the running mean starts over (sample 1 overwrites)
        frames.append(r.framebuffer().copy())
        r.set_transferfunc(lut2); r.sample = 0; step(2)               # the LUT's device array is replaced between two calls
        r.tf_window_left = 0.1; r.tf_window_width = 0.7; step(2)
        r.set_transferfunc(None); step(1)
        frames.append(r.framebuffer().copy())
        return frames

    ref = drive(scenes.oracle_scene("c1", w, h), True)
    eager = scenes.hip_scene("c1", w, h)
    eager.coalesce_trace = 0
    fe = drive(eager, False)
    lazy = scenes.hip_scene("c1", w, h)
    fl = drive(lazy, False)
    for k, (x, y, z) in enumerate(zip(ref, fe, fl)):
        assert np.array_equal(_bits(y), _bits(z)), "coalesced != one launch per call at observation %d" % k
        _assert_same(z, x, "trace protocol with changes, observation %d" % k)


def test_coalesced_trace_flush_points():
    """Everything that can observe or replace what recorded samples read launches them first."""
    w, h = 48, 32
    ref4 = scenes.oracle_scene("c1", w, h).render(4)
    r = scenes.hip_scene("c1", w, h)
    for _ in range(4):
        r.trace()
    assert r.pending_samples == 4
    r.flush(); assert r.pending_samples == 0                         # vr_flush: launched, not waited for
    _assert_same(r.framebuffer(), ref4, "flush")
    for probe in ("draw", "last_kernel_ms", "framebuffer_device_ptr", "synchronize", "commit"):
        r.reset()
        for _ in range(4):
            r.trace()
        assert r.pending_samples == 4
        getattr(r, probe)()
        assert r.pending_samples == 0, probe
        _assert_same(r.framebuffer(), ref4, probe)
    # set_tiles between two trace() calls: the recorded samples cover the whole frame, the later ones the subset
    r.reset()
    r.trace(); r.trace()
    r.set_tiles([0, 1])
    assert r.pending_samples == 0
    r.trace()
    r.set_tiles([])
    o = scenes.oracle_scene("c1", w, h)
    full2 = o.render(2).copy()
    full3 = o.render(1)
    fb = r.framebuffer()
    mask = np.zeros((h, w), bool); mask[0:16, 0:32] = True
    assert np.array_equal(_bits(fb[mask]), _bits(full3[mask])) and np.array_equal(_bits(fb[~mask]), _bits(full2[~mask]))
    # a render() after recorded trace() calls continues the same frame; resize drops what was recorded for the old framebuffer
    r.reset(); r.trace(); r.render(3)
    _assert_same(r.framebuffer(), ref4, "trace + render")
    r.trace(); r.resize(32, 32); assert r.pending_samples == 0 and r.sample == 0
    # a full sub-launch goes out by itself: with a 16 MiB pool this frame holds a few hundred samples per launch
    r.resize(w, h); r.sample_pool_mb = 16
    n = 0
    while r.pending_samples == n and n < 5000:
        r.trace(); n += 1
    assert 32 <= n < 5000 and r.pending_samples == 0, n
    r.synchronize()


def test_ragged_resolution_and_tiles():
    """W,H not multiples of 16; a tile subset renders exactly those tiles and leaves the rest untouched."""
    w, h = 70, 37
    o = scenes.oracle_scene("c1", w, h)
    ref = o.render(4)
    r = scenes.hip_scene("c1", w, h)
    r.render(4)
    _assert_same(r.framebuffer(), ref, "ragged")
    tiles_x = (w + 15) // 16
    sub = [0, 3, 7, tiles_x * 2 + 1]
    s = scenes.hip_scene("c1", w, h)
    s.set_tiles(sub)
    s.render(4)
    fb = s.framebuffer()
    mask = np.zeros((h, w), bool)
    for t in sub:
        tx, ty = t % tiles_x, t // tiles_x
        mask[ty * 16:ty * 16 + 16, tx * 16:tx * 16 + 16] = True
    assert np.array_equal(_bits(fb[mask]), _bits(ref[mask]))
    assert not fb[~mask].any()
    with pytest.raises(Exception):
        s.set_tiles([10 ** 6])


def test_tile_shard_pack_gather_unpack_on_device():
    """The multi-GPU data path on one GPU: 3 'ranks' render their diagonal-interleaved tiles, pack them with the HIP pack
    kernel, the packed buffers are concatenated like all_gather does, and every rank's unpack rebuilds the full frame --
    bit-identical to the unsharded render (torch only carries the device buffers)."""
    import torch
    from volren_amd.shard import TileShard
    w, h, spp, world = 150, 90, 3, 3
    full = scenes.hip_scene("c1", w, h)
    full.render(spp)
    ref = full.framebuffer()
    packed = []
    shards = []
    for rank in range(world):
        sh = TileShard(w, h, world, rank)
        r = scenes.hip_scene("c1", w, h)
        r.set_tiles(sh.mine)
        r.render(spp)
        ids = torch.from_numpy(sh.pack_ids).cuda()
        buf = torch.zeros(sh.packed_floats, dtype=torch.float32, device="cuda")
        r.pack_tiles(ids.data_ptr(), sh.n_max, buf.data_ptr())
        r.synchronize()
        torch.cuda.synchronize()
        packed.append(buf)
        shards.append((sh, r))
    gathered = torch.cat(packed)
    for sh, r in shards:
        assert gathered.numel() == sh.gathered_floats
        ids = torch.from_numpy(sh.unpack_ids).cuda()
        r.unpack_tiles(ids.data_ptr(), world * sh.n_max, gathered.data_ptr())
        r.synchronize()
        assert np.array_equal(_bits(r.framebuffer()), _bits(ref))


def test_emission_grid_and_brick_upload():
    """Synthetic density + temperature brick grids (numpy reference encoder) handed to both sides as raw BrickGrid
    arrays: checks vr_set_volume_brick, the emission path (common.glsl:324-328,489) and a second grid layout."""
    from oracle import binding as ob
    import encoder_ref
    import volren_amd
    n = 40
    dens = scenes.synthetic_density(n)
    temp = np.clip(dens * 0.2 + 0.1 * scenes.synthetic_density(n, seed=99), 0, None).astype(np.float32)
    ad, at = encoder_ref.encode_arrays(dens), encoder_ref.encode_arrays(temp)
    r = volren_amd.Renderer(64, 64)
    r.load_envmap(scenes.HDR)
    r.set_volume_brick(ad["transform"], ad["n_bricks"], ad["min_maj"], ad["indirection"], ad["rng"], ad["atlas_dim"], ad["atlas"], ad["mips"], commit=False)
    r.set_volume_brick(at["transform"], at["n_bricks"], at["min_maj"], at["indirection"], at["rng"], at["atlas_dim"], at["atlas"], at["mips"], name="temperature", commit=True)
    o = ob.OracleRenderer(64, 64)
    o.load_envmap(scenes.HDR)
    o.set_volume(encoder_ref.encode(dens), emission=encoder_ref.encode(temp), majorant_emission=at["min_maj"][1])
    for x in (r, o):
        x.cam_fov = 40.0
        x.bounces = 8
        x.albedo = (0.7, 0.8, 0.9)
        x.emission_scale = 50.0
    r.render(8)
    hip = r.framebuffer()
    assert hip[..., :3].max() > 0
    _assert_same(hip, o.render(8), "brick upload + emission")


def test_emission_grid_with_a_different_brick_layout():
    """A temperature grid of another resolution and transform than the density grid (40^3 voxels scaled 1.8x over a 72^3 density grid: 8^3 against 16^3
    bricks): no paired atlas for this frame -- the kernel compiled for two brick grids of one layout is not the one that runs -- and emission_inv_transform *
    density_transform is a real matrix.  Same image as the oracle bit for bit; then the same pair of grids at equal layout for contrast."""
    from oracle import binding as ob
    import encoder_ref
    import volren_amd
    dens = scenes.synthetic_density(72, blobs=12)
    temp = np.clip(scenes.synthetic_density(40, seed=99) * 0.2, 0, None).astype(np.float32)
    t_temp = np.diag([1.8, 1.8, 1.8, 1.0]).astype(np.float32).reshape(16)
    ad, at = encoder_ref.encode_arrays(dens), encoder_ref.encode_arrays(temp, t_temp)
    assert tuple(ad["n_bricks"]) == (16, 16, 16) and tuple(at["n_bricks"]) == (8, 8, 8)
    r = volren_amd.Renderer(72, 56)
    r.load_envmap(scenes.HDR)
    r.set_volume_brick(ad["transform"], ad["n_bricks"], ad["min_maj"], ad["indirection"], ad["rng"], ad["atlas_dim"], ad["atlas"], ad["mips"], commit=False)
    r.set_volume_brick(at["transform"], at["n_bricks"], at["min_maj"], at["indirection"], at["rng"], at["atlas_dim"], at["atlas"], at["mips"], name="temperature", commit=True)
    o = ob.OracleRenderer(72, 56)
    o.load_envmap(scenes.HDR)
    o.set_volume(encoder_ref.encode(dens), emission=encoder_ref.encode(temp, t_temp), majorant_emission=at["min_maj"][1])
    for x in (r, o):
        x.cam_fov, x.bounces, x.albedo, x.emission_scale = 40.0, 8, (0.7, 0.8, 0.9), 50.0
    r.render(6)
    hip = r.framebuffer()
    assert hip[..., :3].max() > 0
    ref = o.render(6)
    _assert_same(hip, ref, "emission grid of another layout")
    # The run-time kernel variant swaps a lane's marching path through its LDS slot around every event batch.  Until round 4 that swap lost the majorant of a lane
    # WAITING at a tentative collision (the collide threshold of round 2), so this variant's images depended on the scheduler's thresholds -- 2 pixels of this frame
    # at a collide threshold of 32, none at 24.  Every setting must give the oracle's image.
    for thr in ([64, 0, 56, 32, 60, 60, 64, 0], [64, 0, 56, 48, 60, 60, 64, 0], [8, 0, 8, 40, 8, 8, 8, 0], [1, 66, 1, 1, 1, 1, 1, 0], [64, 0, 64, 63, 64, 64, 64, 0]):
        r.set_sched(thr)
        r.reset()
        r.render(6)
        _assert_same(r.framebuffer(), ref, "emission grid of another layout, scheduler %s" % thr)
    r.set_sched([64, 0, 56, 0, 60, 60, 64, 0])
    r.reset()
    for _ in range(6):                                   # the reference's protocol: one trace() per sample
        r.trace()
    r.synchronize()
    _assert_same(r.framebuffer(), ref, "emission grid of another layout, trace() x 6")


def _crop_bricks(a, nbc):
    """Centred sub-block nbc = (cx, cy, cz) of an encode_arrays() brick grid: indirection/range cropped (the atlas pointers stay
    valid), range mips rebuilt with the ceil(n / 2) rule (min of mins, max of maxes over the existing children)."""
    nbx, nby, nbz = a["n_bricks"]
    cx, cy, cz = nbc
    ox, oy, oz = (nbx - cx) // 2, (nby - cy) // 2, (nbz - cz) // 2          # centred: the synthetic cloud is densest there
    ind = a["indirection"].reshape(nbz, nby, nbx)[oz:oz + cz, oy:oy + cy, ox:ox + cx].copy()
    rng = a["rng"].reshape(nbz, nby, nbx)[oz:oz + cz, oy:oy + cy, ox:ox + cx].copy()
    lo = (rng & 0xFFFF).astype(np.uint16).view(np.float16).astype(np.float32)
    hi = (rng >> 16).astype(np.uint16).view(np.float16).astype(np.float32)
    mips = []
    for _ in range(3):
        z, y, x = lo.shape
        z2, y2, x2 = (z + 1) // 2, (y + 1) // 2, (x + 1) // 2
        plo = np.full((z2 * 2, y2 * 2, x2 * 2), np.inf, np.float32)
        phi = np.full((z2 * 2, y2 * 2, x2 * 2), -np.inf, np.float32)
        plo[:z, :y, :x] = lo
        phi[:z, :y, :x] = hi
        lo = plo.reshape(z2, 2, y2, 2, x2, 2).min((1, 3, 5))
        hi = phi.reshape(z2, 2, y2, 2, x2, 2).max((1, 3, 5))
        w = lo.astype(np.float16).view(np.uint16).astype(np.uint32) | (hi.astype(np.float16).view(np.uint16).astype(np.uint32) << 16)
        mips.append(((x2, y2, z2), w.reshape(-1)))
    out = dict(a)
    out.update(n_bricks=(cx, cy, cz), indirection=ind.reshape(-1), rng=rng.reshape(-1), mips=mips)
    return out


def test_dense_fp16_density_with_emission_grid():
    """The one combination no specialised kernel serves -- a dense fp16 density grid together with a (brick) temperature grid, with and without a
    transfer function -- runs on the everything-at-run-time variant (vr_kernels.hip pathtrace_variant -> 3): bit for bit the oracle's image."""
    from oracle import binding as ob
    import encoder_ref
    import volren_amd
    n = 48
    dens = scenes.synthetic_density(n)
    temp = np.clip(dens * 0.15 + 0.05 * scenes.synthetic_density(n, seed=5), 0, None).astype(np.float32)
    r = volren_amd.Renderer(80, 64)
    r.load_envmap(scenes.HDR)
    r.set_volume_dense_f16(dens, commit=False)
    r.set_volume_dense(temp, name="temperature", commit=True)
    o = ob.OracleRenderer(80, 64)
    o.load_envmap(scenes.HDR)
    gt = encoder_ref.encode(temp)
    gt.extent = (n, n, n)
    gt.c.extent[:] = gt.extent
    o.set_volume(encoder_ref.encode_dense_fp16(dens), emission=gt, majorant_emission=float(temp.max()))
    for x in (r, o):
        x.cam_fov, x.bounces, x.albedo, x.phase, x.density_scale, x.emission_scale = 40.0, 12, (0.7, 0.8, 0.9), 0.2, 60.0, 80.0
    r.render(6)
    fb = r.framebuffer()
    assert fb[..., :3].max() > 0 and fb[..., 3].max() > 0
    ref = o.render(6).copy()
    _assert_same(fb, ref, "dense fp16 density + emission grid")
    for thr in ([64, 0, 56, 32, 60, 60, 64, 0], [8, 0, 8, 40, 8, 8, 8, 0], [64, 0, 64, 63, 64, 64, 64, 0]):      # the run-time variant under other scheduler settings
        r.set_sched(thr)
        r.reset()
        r.render(6)
        _assert_same(r.framebuffer(), ref, "dense fp16 density + emission grid, scheduler %s" % thr)
    r.set_sched([64, 0, 56, 0, 60, 60, 64, 0])
    for x in (r, o):
        x.load_transferfunc(scenes.LUT)
        x.reset() if hasattr(x, "reset") else None
    o.sample = 0
    if hasattr(o, "fb"):
        o.fb[:] = 0
    r.render(4)
    _assert_same(r.framebuffer(), o.render(4), "dense fp16 density + emission grid + transfer function")


@pytest.mark.parametrize("nbc", [(5, 3, 7), (1, 2, 1), (8, 7, 3)])
def test_odd_brick_counts(nbc):
    """Brick counts that are neither powers of two nor multiples of 8 (a .brick file may hold any): exercises the padded
    power-of-two pitches of the brick records and majorant levels (vr_scene.h) against the oracle's plain indexing."""
    from oracle import binding as ob
    import encoder_ref
    import volren_amd
    a = _crop_bricks(encoder_ref.encode_arrays(scenes.synthetic_density(64)), nbc)
    r = volren_amd.Renderer(64, 48)
    r.load_envmap(scenes.HDR)
    r.set_volume_brick(a["transform"], a["n_bricks"], a["min_maj"], a["indirection"], a["rng"], a["atlas_dim"], a["atlas"], a["mips"], commit=True)
    g = ob.Grid()
    g.set(a["transform"], a["n_bricks"], a["min_maj"], a["brick_counter"], a["indirection"], a["rng"], a["atlas_dim"], a["atlas"], a["mips"])
    o = ob.OracleRenderer(64, 48)
    o.load_envmap(scenes.HDR)
    o.set_volume(g)
    for x in (r, o):
        x.cam_fov = 40.0
        x.bounces = 6
    r.render(6)
    hip = r.framebuffer()
    assert hip[..., 3].max() > 0
    _assert_same(hip, o.render(6), "odd brick counts %s" % (nbc,))


def test_dense_fp16_grid_matches_oracle():
    """vr_set_volume_dense_f16: the grid stays dense on the device (2 B voxels + macro-cell majorant mips built by the
    product's C++ code); the oracle gets the numpy reference arrays."""
    from oracle import binding as ob
    import encoder_ref
    import volren_amd
    dens = scenes.synthetic_density(72)[:64, :56, :72].copy()
    r = volren_amd.Renderer(96, 96)
    r.load_envmap(scenes.HDR)
    r.set_volume_dense_f16(dens)
    o = ob.OracleRenderer(96, 96)
    o.load_envmap(scenes.HDR)
    o.set_volume(encoder_ref.encode_dense_fp16(dens))
    for x in (r, o):
        x.cam_fov = 40.0
        x.bounces = 16
        x.albedo = (0.8, 0.8, 0.8)
        x.phase = 0.3
    r.render(8)
    _assert_same(r.framebuffer(), o.render(8), "dense fp16")
    dense_mean = float(r.framebuffer()[..., :3].mean())
    # the same dense grid through the transfer-function kernel (8-corner trilinear fetch on the blocked layout), odd extent
    for x in (r, o):
        x.load_transferfunc(scenes.LUT)
        x.bounces = 6
        x.reset()
    if hasattr(o, "fb"):
        o.fb[:] = 0
    r.render(4)
    _assert_same(r.framebuffer(), o.render(4), "dense fp16 + transfer function")
    # the brick path on the same data (u8-quantised) agrees statistically, not bitwise
    b = volren_amd.Renderer(96, 96)
    b.load_envmap(scenes.HDR)
    b.set_volume_dense(dens)
    b.cam_fov, b.bounces, b.albedo, b.phase = 40.0, 16, (0.8, 0.8, 0.8), 0.3
    b.render(8)
    assert abs(float(b.framebuffer()[..., :3].mean()) - dense_mean) < 0.05


def test_determinism_full_size_property():
    """Full-size config, size-independent property: two independent renders are bit-identical and alpha in [0,1]."""
    a = scenes.hip_scene("c2", 1024, 1024)
    a.render(2)
    b = scenes.hip_scene("c2", 1024, 1024)
    b.render(1)
    b.render(1)
    fa, fb = a.framebuffer(), b.framebuffer()
    assert np.array_equal(_bits(fa), _bits(fb))
    assert np.isfinite(fa).all() and (fa[..., 3] >= 0).all() and (fa[..., 3] <= 1).all()
    # 31 % of camera rays miss the box at fov 40: alpha==0 pixels exist, and they carry pure env radiance
    assert 0.3 < (fa[..., 3] == 0).mean() < 0.95


def test_cli_offline_render_matches_oracle(tmp_path):
    """`volren ... --render` (parse_cmd order semantics, offline loop, tonemap.glsl, PNG naming) against the
    oracle's tonemapped frame: identical 8-bit pixels."""
    import subprocess
    from PIL import Image
    exe = scenes.ROOT + "/volren_amd/volren"
    cmd = [exe, scenes.SMOKE, scenes.HDR, "-w", "96", "-h", "80", "--render", "--spp", "12", "--bounces", "128", "--albedo", "0.8",
           "--phase", "0.3", "--density", "100", "--env_strength", "3", "--env_rot", "270", "--exposure", "3", "--gamma", "2.0",
           "--cam_fov", "40", "--output", "some/dir/shot.png"]
    out = subprocess.run(cmd, cwd=tmp_path, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    png = tmp_path / "shot_000000.png"            # directory of --output is dropped, "_%06d" appended (main.cpp:552-554)
    assert png.exists(), out.stdout
    img = np.asarray(Image.open(png))
    assert img.shape == (80, 96, 4)
    o = scenes.oracle_scene("readme", 96, 80)
    o.render(12)
    tm = o.tonemapped()[::-1]
    want = np.floor(np.clip(tm, 0, 1) * 255.0 + 0.5).astype(np.uint8)
    assert np.array_equal(img, want)
    # --tf_left / --tf_width take their value only while a transfer function exists (src/main.cpp:397-402); without one the value stays behind as an
    # argument of its own -- here a flag: `--tf_left --env_hide` must hide the environment (a parser that always consumes a value would swallow it)
    base = [exe, scenes.SMOKE, scenes.HDR, "-w", "48", "-h", "40", "--render", "--spp", "3", "--cam_fov", "40"]
    shots = {}
    for tag, extra in (("hidden", ["--env_hide"]), ("quirk", ["--tf_left", "--env_hide"]), ("shown", [])):
        out = subprocess.run(base + extra + ["--output", tag + ".png"], cwd=tmp_path, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        shots[tag] = (tmp_path / (tag + "_000000.png")).read_bytes()
    assert shots["quirk"] == shots["hidden"] != shots["shown"]


def test_tonemap_and_display_buffer():
    o = scenes.oracle_scene("c1", 48, 32)
    o.tonemap_exposure, o.tonemap_gamma = 2.5, 2.2
    o.render(4)
    r = scenes.hip_scene("c1", 48, 32)
    r.tonemap_exposure, r.tonemap_gamma = 2.5, 2.2
    r.render(4)
    r.draw()
    assert np.array_equal(_bits(r.display()), _bits(o.tonemapped()))
    assert np.array_equal(_bits(r.framebuffer()), _bits(o.fb))          # draw() leaves the accumulation buffer untouched


def test_volpy_module_drives_the_renderer(tmp_path):
    """The reference's Python surface (src/bindings.cpp:64-209) as used by scripts/datagen_colmap.py: assign
    Volume/Environment, set fields, unit cube + commit, render(spp), fbo_data(), draw() + save_with_alpha()."""
    from PIL import Image
    import volren_amd.volpy as volpy
    renderer = volpy.Renderer(64, 48)
    renderer.init()
    renderer.volume = volpy.Volume(scenes.SMOKE)
    renderer.scale_and_move_to_unit_cube()
    renderer.commit()
    renderer.environment = volpy.Environment(scenes.HDR)
    renderer.environment.strength = 3.0
    renderer.albedo = volpy.vec3(0.8)
    renderer.bounces, renderer.seed, renderer.phase = 16, 42, 0.3
    renderer.cam_pos, renderer.cam_fov = volpy.vec3(1, 0, 1), 40.0
    renderer.cam_dir = -volpy.vec3(1, 0, 1) / np.float32(np.sqrt(2))
    renderer.render(6)
    assert renderer.sample == 6 and (renderer.resolution().x, renderer.resolution().y) == (64, 48)
    rgb = np.asarray(renderer.fbo_data())
    assert rgb.shape == (64, 48, 3) and rgb.dtype == np.float32
    o = scenes.oracle_scene("c1", 64, 48)
    o.env_strength, o.albedo, o.bounces, o.phase = 3.0, (0.8, 0.8, 0.8), 16, 0.3
    o.cam_dir = tuple((-np.array([1, 0, 1], np.float32) / np.float32(np.sqrt(2))).tolist())
    ref = o.render(6)
    assert np.array_equal(_bits(rgb.reshape(48, 64, 3)), _bits(ref[..., :3]))
    renderer.draw()
    out = tmp_path / "view_0000.jpg"
    renderer.save_with_alpha(str(out))                       # extension forced to .png (bindings.cpp:163)
    img = np.asarray(Image.open(tmp_path / "view_0000.png"))
    want = np.floor(np.clip(o.tonemapped()[::-1], 0, 1) * 255.0 + 0.5).astype(np.uint8)
    assert np.array_equal(img, want)
    assert abs(renderer.colmap_focal_length() - 48 / (2 * np.tan(np.radians(20.0)))) < 1e-3
    q = renderer.colmap_view_rot()
    assert abs(float(np.linalg.norm(np.array(q))) - 1.0) < 1e-5 and np.array(renderer.colmap_view_trans()).shape == (3,)
    assert np.array(q)[[3, 0, 1, 2]][0] == q.w                           # buffer order (x, y, z, w) like glm::quat
    # a dense in-memory volume + a LUT given as a list of vec4, like datagen_denoise.py builds them
    renderer.volume = volpy.Volume(40, 40, 40, scenes.synthetic_density(40))
    renderer.scale_and_move_to_unit_cube()
    renderer.commit()
    renderer.transferfunc = volpy.TransferFunction([volpy.vec4(0), volpy.vec4(0.2, 0.4, 0.9, 0.5), volpy.vec4(1, 1, 1, 1)])
    renderer.transferfunc.window_width = 0.5
    renderer.render(2)
    assert np.isfinite(np.asarray(renderer.fbo_data())).all()


def test_volpy_grid_frames():
    """Volume.add_grid_frame / update_grid_frame (src/bindings.cpp:89-90) through volpy: a two-frame animation built in memory, the second
    frame's density replaced afterwards and temperature grids added to both; each frame renders bit for bit like the oracle on that frame's
    grids.  (Every frame carries a temperature grid: commit() appends emission grids only for the frames that have one, src/renderer.cpp:64-74,
    so a frame without one would pick up a later frame's -- a quirk the product keeps and this test stays clear of.)"""
    import encoder_ref
    from oracle import binding as ob
    import volren_amd.volpy as volpy
    W, H, SPP = 64, 48, 4
    d0 = scenes.synthetic_density(32)
    d1 = scenes.synthetic_density(32, seed=7)
    d1b = scenes.synthetic_density(32, seed=11)
    t0 = np.clip(d0 * 0.1, 0, None).astype(np.float32)
    t1 = np.clip(d1b * 0.2, 0, None).astype(np.float32)
    renderer = volpy.Renderer(W, H)
    renderer.environment = volpy.Environment(scenes.HDR)
    vol = volpy.Volume(32, 32, 32, d0)
    vol.add_grid_frame(volpy.Volume(32, 32, 32, d1))
    renderer.volume = vol
    vol.update_grid_frame(1, d1b)                                  # after the assignment: replayed onto the renderer's volume
    vol.update_grid_frame(0, t0, "temperature")
    vol.update_grid_frame(1, t1, "temperature")
    assert vol.n_grid_frames() == 2
    renderer.scale_and_move_to_unit_cube()                          # what the oracle's set_volume does; the density scale is then SET, as main.cpp's --density does
    renderer.cam_fov, renderer.bounces, renderer.density_scale = 40.0, 6, 40.0
    renderer.commit()
    maj_e = float(max(t0.max(), t1.max()))                          # commit(): the maximum of the grids' majorants over all frames (renderer.cpp:73)
    for frame, (dens, temp) in enumerate(((d0, t0), (d1b, t1))):
        vol.grid_frame_counter = frame
        renderer.render(SPP)
        o = ob.OracleRenderer(W, H)
        o.load_envmap(scenes.HDR)
        gd, gt = encoder_ref.encode(dens), encoder_ref.encode(temp)
        for g in (gd, gt):                          # an in-memory dense grid keeps its voxel extent (32, not bricks x 8 = 64)
            g.extent = (32, 32, 32)
            g.c.extent[:] = g.extent
        o.set_volume(gd, emission=gt, majorant_emission=maj_e)
        o.cam_fov, o.bounces, o.density_scale = 40.0, 6, 40.0
        ref = o.render(SPP)
        assert np.array_equal(_bits(np.asarray(renderer.fbo_data()).reshape(H, W, 3)), _bits(ref[..., :3])), frame
    # an edit AFTER scale_and_move_to_unit_cube(): the reference mutates the renderer's voldata::Volume in place, so the unit-cube transform
    # (and the density scale that goes with it) stays -- the edit must not rebuild the volume from scratch
    for k in range(3):                                                # a per-frame update loop: each update supersedes the one before
        vol.update_grid_frame(0, scenes.synthetic_density(32, seed=20 + k))
    vol.update_grid_frame(0, d1)
    assert len([e for e in vol._edits if e[0] == "update" and e[1] == 0 and e[3] == "density"]) == 1
    assert vol.n_grid_frames() == 2
    renderer.commit()
    vol.grid_frame_counter = 0
    renderer.render(SPP)
    o = ob.OracleRenderer(W, H)
    o.load_envmap(scenes.HDR)
    gd, gt = encoder_ref.encode(d1), encoder_ref.encode(t0)
    for g in (gd, gt):
        g.extent = (32, 32, 32)
        g.c.extent[:] = g.extent
    o.set_volume(gd, emission=gt, majorant_emission=maj_e)
    o.cam_fov, o.bounces, o.density_scale = 40.0, 6, 40.0
    ref = o.render(SPP)
    assert np.array_equal(_bits(np.asarray(renderer.fbo_data()).reshape(H, W, 3)), _bits(ref[..., :3])), "edit after the unit cube"


def _volpy_look_at_volume(volpy, renderer, u_pos, u_aim):
    """Places the volpy camera on the bounding sphere of the density grid, aimed near its centre; u_pos / u_aim are points of [0, 1)^2 mapped to the
    sphere by the cylinder projection.  Uses only what a volpy script has: Volume.AABB(), vec3 arithmetic, .length(), .normalize()."""
    def on_sphere(u):
        h = 1.0 - 2.0 * float(u[0])
        rho = float(np.sqrt(max(0.0, 1.0 - h * h)))
        a = 2.0 * np.pi * float(u[1])
        return volpy.vec3(rho * float(np.cos(a)), rho * float(np.sin(a)), h)
    lo, hi = renderer.volume.AABB("density")
    mid = lo + (hi - lo) * 0.5
    rad = (hi - mid).length()
    renderer.cam_pos = mid + on_sphere(u_pos) * rad
    renderer.cam_dir = (mid + on_sphere(u_aim) * rad * 0.1 - renderer.cam_pos).normalize()
    return rad


def _oracle_for_volpy(ob, renderer, w, h, unit_cube):
    """The oracle set up from the numbers a volpy renderer holds (not from the calls that put them there)."""
    o = ob.OracleRenderer(w, h)
    o.load_envmap(scenes.HDR)
    o.load_volume(scenes.SMOKE)                                    # leaves the unit-cube transform and density_scale = the cube's size
    if not unit_cube:
        o.volume_transform = np.eye(4, dtype=np.float32).reshape(16).copy()
    return o


def test_volpy_call_protocol_of_the_data_generation_scripts(tmp_path):
    """The call protocol the reference's data-generation drivers rely on (scripts/datagen_colmap.py:46-95, scripts/datagen_denoise.py:83-121), exercised on
    volren_amd.volpy and checked frame by frame, bit for bit, against the oracle:
      A. density_scale set BEFORE scale_and_move_to_unit_cube() -> commit(): the cube's factor multiplies the caller's scale (src/renderer.cpp:240);
         AABB() then reports the unit cube; the COLMAP helpers and resolution() answer; save_with_alpha() writes a file;
      B. commit() of a file volume WITHOUT the unit cube keeps the file's transform; TransferFunction.randomize(n) + window, and transferfunc = None
         switch kernels between frames; seed / bounces changed between two render() calls of one set-up; fbo_data() is (h, w, 3), bottom row first."""
    from oracle import binding as ob
    import volren_amd.volpy as volpy
    W, H = 64, 48
    rng = np.random.RandomState(20260)

    # ---- A: unit cube, several views of one set-up ----
    renderer = volpy.Renderer(W, H)
    renderer.init()
    renderer.draw()
    for name, value in (("seed", 42), ("bounces", 16), ("volume", volpy.Volume(scenes.SMOKE)), ("albedo", volpy.vec3(0.9, 0.9, 0.9)), ("phase", 0.5),
                        ("density_scale", 0.75), ("environment", volpy.Environment(scenes.HDR)), ("show_environment", True), ("tonemapping", True)):
        setattr(renderer, name, value)
    renderer.environment.strength = 2.0
    renderer.scale_and_move_to_unit_cube()
    renderer.commit()
    lo, hi = (np.array(v) for v in renderer.volume.AABB("density"))
    assert np.allclose(lo, [-0.25, -0.5, -0.25], atol=1e-6) and np.allclose(hi, [0.25, 0.5, 0.25], atol=1e-6)
    assert (renderer.resolution().x // 2, renderer.resolution().y // 2) == (W // 2, H // 2) and renderer.colmap_focal_length() > 0
    o = _oracle_for_volpy(ob, renderer, W, H, unit_cube=True)
    o.density_scale = float(np.float32(0.75) * np.float32(o.density_scale))
    assert abs(renderer.density_scale - o.density_scale) < 1e-4 * o.density_scale
    o.env_strength, o.albedo, o.phase, o.bounces, o.seed = 2.0, (0.9, 0.9, 0.9), 0.5, 16, 42
    for view in range(2):
        _volpy_look_at_volume(volpy, renderer, rng.rand(2), rng.rand(2))
        renderer.cam_fov = 70
        renderer.render(4)
        renderer.draw()
        out = os.path.join(str(tmp_path), "view_%06d.png" % view)
        renderer.save_with_alpha(out)
        assert os.path.getsize(out) > 0
        q, t = np.array(renderer.colmap_view_rot()), np.array(renderer.colmap_view_trans())
        assert q.shape == (4,) and t.shape == (3,) and abs(np.linalg.norm(q) - 1) < 1e-5
        o.cam_pos, o.cam_dir, o.cam_fov, o.sample = tuple(np.array(renderer.cam_pos).tolist()), tuple(np.array(renderer.cam_dir).tolist()), 70, 0
        _assert_same(renderer._r.framebuffer(), o.render(4), "unit cube, view %d" % view)
    renderer.shutdown()

    # ---- B: the file's own transform, kernels switched between images, two renders per set-up ----
    renderer = volpy.Renderer(W, H)
    renderer.init()
    renderer.draw()
    for image, lut_bins in enumerate((0, 9)):
        albedo = tuple(float(v) for v in rng.rand(3))
        setup = dict(phase=float(rng.uniform(-0.9, 0.9)), density_scale=float(rng.uniform(0.01, 5.0)), show_environment=bool(rng.rand() < 0.5))
        strength, fov, bounces = float(rng.uniform(0.5, 10.0)), float(rng.uniform(25, 95)), int(rng.randint(1, 34))
        renderer.volume = volpy.Volume(scenes.SMOKE)
        renderer.commit()
        renderer.albedo = volpy.vec3(*albedo)
        for name, value in setup.items():
            setattr(renderer, name, value)
        renderer.environment = volpy.Environment(scenes.HDR)
        renderer.environment.strength = strength
        o = _oracle_for_volpy(ob, renderer, W, H, unit_cube=False)
        if lut_bins:
            renderer.transferfunc = volpy.TransferFunction()
            renderer.transferfunc.randomize(lut_bins)
            renderer.transferfunc.window_left, renderer.transferfunc.window_width = 0.1, 0.8
            o.set_transferfunc(renderer.transferfunc.lut)
            o.tf_window_left, o.tf_window_width = 0.1, 0.8
        else:
            renderer.transferfunc = None
        assert _volpy_look_at_volume(volpy, renderer, rng.rand(2), rng.rand(2)) > 50       # world units of the file, not the unit cube
        renderer.cam_fov = fov
        o.albedo, o.phase, o.density_scale, o.show_environment, o.env_strength = albedo, setup["phase"], setup["density_scale"], setup["show_environment"], strength
        o.cam_pos, o.cam_dir, o.cam_fov = tuple(np.array(renderer.cam_pos).tolist()), tuple(np.array(renderer.cam_dir).tolist()), fov
        for seed, spp in ((int(rng.randint(0, 2 ** 31 - 1)), int(rng.randint(1, 9))), (int(rng.randint(0, 2 ** 31 - 1)), 12)):
            renderer.seed, renderer.bounces = seed, bounces
            renderer.render(spp)
            rows = np.array(renderer.fbo_data())
            assert np.transpose(np.flip(rows, axis=0), [2, 1, 0]).shape == (3, renderer.resolution().y, renderer.resolution().x)
            renderer.draw()
            o.seed, o.bounces, o.sample = seed, bounces, 0
            _assert_same(renderer._r.framebuffer(), o.render(spp), "file transform, image %d, %d spp" % (image, spp))
    renderer.shutdown()


@pytest.mark.parametrize("launcher", ["volren", "module"])
def test_scripts_written_for_volpy_run_unmodified(tmp_path, launcher):
    """A script whose only renderer import is `import volpy` (tests/fixtures/script_views.py: init, draw, volume =, scale_and_move_to_unit_cube,
    commit, render, fbo_data, save_with_alpha -- the calls of scripts/datagen_colmap.py:46-95), started the way the reference starts its scripts:
    `volren script.py --render -w W -h H` (src/main.cpp:83-91), or `python -m volren_amd.run_script`.  `volpy.Renderer()` gets the -w / -h
    resolution; every view is the oracle's frame bit for bit."""
    import json
    import subprocess
    import sys
    from PIL import Image
    from oracle import binding as ob
    W, H, SPP = 72, 48, 4
    script = os.path.join(scenes.FIX, "script_views.py")
    env = dict(os.environ, VIEWS_OUT=str(tmp_path), VIEWS_N="2", VIEWS_SPP=str(SPP))
    if launcher == "volren":
        cmd = [scenes.ROOT + "/volren_amd/volren", script, "--render", "-w", str(W), "-h", str(H)]
    else:
        cmd = [sys.executable, "-m", "volren_amd.run_script", script, "-w", str(W), "-h", str(H), "--render", "extra-arg"]
        env["PYTHONPATH"] = scenes.ROOT + os.pathsep + env.get("PYTHONPATH", "")
    out = subprocess.run(cmd, cwd=tmp_path, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    meta = json.load(open(tmp_path / "views.json"))
    assert (meta["width"], meta["height"], meta["spp"]) == (W, H, SPP)
    assert meta["argv"] == ([] if launcher == "volren" else ["extra-arg"])           # the context flags are the launcher's, the rest is the script's
    o = ob.OracleRenderer(W, H)
    o.load_envmap(scenes.HDR)
    o.load_volume(scenes.SMOKE)                                                      # unit cube: density_scale = size
    o.density_scale = float(np.float32(0.5) * np.float32(o.density_scale))
    o.env_strength, o.albedo, o.phase, o.bounces, o.seed = 1.5, (0.8, 0.85, 0.9), 0.25, 12, 7
    for i, v in enumerate(meta["views"]):
        assert abs(v["density_scale"] - o.density_scale) <= 1e-6 * o.density_scale
        o.cam_pos, o.cam_dir, o.cam_fov = tuple(v["cam_pos"]), tuple(v["cam_dir"]), v["cam_fov"]
        o.sample = 0
        ref = o.render(SPP)
        got = np.load(tmp_path / ("view_%03d.npy" % i))
        assert got.shape == (W, H, 3)                                                # the reference's declared buffer shape (bindings.cpp:69-77,143)
        assert np.array_equal(_bits(got.reshape(H, W, 3)), _bits(ref[..., :3])), "view %d" % i
        png = np.asarray(Image.open(tmp_path / ("view_%03d.png" % i)))
        want = np.floor(np.clip(o.tonemapped()[::-1], 0, 1) * 255.0 + 0.5).astype(np.uint8)
        assert png.shape == (H, W, 4) and np.array_equal(png, want)


def test_dense_and_raw_volume_files(tmp_path):
    """main.cpp:44 loads any grid file voldata can read; besides .brick this build reads serialized dense grids (".dense", this
    build's container: grids.cpp) and headerless ".raw" volumes named <name>_<nx>x<ny>x<nz>_<type>.raw.  Both become a
    DenseGrid and go through the device encoder at commit(); checked against the oracle on the numpy-encoded same voxels."""
    import encoder_ref
    import volren_amd
    from oracle import binding as ob
    lib = volren_amd.load()
    d = scenes.synthetic_density(40)[:36, :40, :33].copy()                 # ragged 33 x 40 x 36
    lo, hi = np.float32(0.0), np.float32(d.max())
    u8 = np.clip(np.round((d - lo) / (hi - lo) * 255.0), 0, 255).astype(np.uint8)
    nz, ny, nx = u8.shape
    cases = {}
    path = tmp_path / "cloud.dense"
    assert lib.vr_write_dense(u8.ctypes.data, nx, ny, nz, float(lo), float(hi), None, str(path).encode()) == 0
    cases[str(path)] = (lo + (u8.astype(np.float32) / np.float32(255.0)) * (hi - lo)).astype(np.float32)
    path = tmp_path / ("cloud_%dx%dx%d_uint8.raw" % (nx, ny, nz))
    u8.tofile(path)
    cases[str(path)] = (u8.astype(np.float32) / np.float32(255.0)).astype(np.float32)
    u16 = (u8.astype(np.uint16) * 257)
    path = tmp_path / ("cloud_%dx%dx%d_uint16.raw" % (nx, ny, nz))
    u16.tofile(path)
    cases[str(path)] = (u16.astype(np.float32) / np.float32(65535.0)).astype(np.float32)
    path = tmp_path / ("cloud_%dx%dx%d_float32.raw" % (nx, ny, nz))
    d.tofile(path)
    cases[str(path)] = d
    for path, vox in cases.items():
        r = volren_amd.Renderer(48, 40)
        r.load_envmap(scenes.HDR)
        r.load_volume(path)
        o = ob.OracleRenderer(48, 40)
        o.load_envmap(scenes.HDR)
        g = encoder_ref.encode(vox)
        g.extent = (nx, ny, nz)
        g.c.extent[:] = g.extent
        o.set_volume(g)
        for x in (r, o):
            x.cam_fov, x.bounces = 40.0, 6
        r.render(4)
        fb = r.framebuffer()
        assert fb[..., 3].max() > 0
        _assert_same(fb, o.render(4), os.path.basename(path))
    bad = tmp_path / "cloud.raw"
    u8.tofile(bad)
    with pytest.raises(Exception):
        volren_amd.Renderer(16, 16).load_volume(str(bad))                  # no extent in the name: refused, not guessed


def test_gpu_dense_to_brick_encoder_equals_host_encoder():
    """the commit()-time dense -> brick conversion on the device produces the same brick records, atlas and
    range mips as the host encoder (which is bit-identical to the numpy reference encoder), and renders like the oracle."""
    import encoder_ref
    import volren_amd
    from oracle import binding as ob
    dens = scenes.synthetic_density(72)[:60, :52, :70].copy()          # ragged 70 x 52 x 60
    sums = []
    for gpu in (1, 0):
        r = volren_amd.Renderer(64, 64)
        r.gpu_encoder = gpu
        r.load_envmap(scenes.HDR)
        r.set_volume_dense(dens)
        sums.append(r.grid_checksums())
        if gpu:
            r.cam_fov, r.bounces = 40.0, 8
            r.render(6)
            fb = r.framebuffer()
    assert sums[0] == sums[1], sums
    o = ob.OracleRenderer(64, 64)
    o.load_envmap(scenes.HDR)
    g = encoder_ref.encode(dens)
    g.extent = None
    o.set_volume(g)
    # the in-memory dense grid keeps its own voxel extent for the unit cube / clip box (renderer.cpp:227-242 on the DenseGrid)
    ext = np.array([70, 52, 60], np.float32)
    size = float(ext.max())
    o.volume_transform = np.array([1 / size, 0, 0, 0, 0, 1 / size, 0, 0, 0, 0, 1 / size, 0,
                                   *(np.float32(1 / size) * (-ext * np.float32(0.5))).tolist(), 1], np.float32)
    o.density_scale = size
    g.extent = (70, 52, 60)
    g.c.extent[:] = g.extent
    o.cam_fov, o.bounces = 40.0, 8
    _assert_same(fb, o.render(6), "gpu-encoded dense grid")


@pytest.mark.parametrize("name", ["c4:64", "c5:64", "c5cloud:128"])
def test_synthetic_baseline_configs_small(name):
    """BASELINE configs[3] / [4] at a size the oracle finishes in seconds: dense fp16 grid (c4) and sparse brick grid +
    temperature grid with emission through the device encoder (c5)."""
    o = scenes.oracle_scene(name, 72, 56)
    r = scenes.hip_scene(name, 72, 56)
    r.render(4)
    fb = r.framebuffer()
    assert fb[..., :3].max() > 0
    _assert_same(fb, o.render(4), name)


def _full_resolution_properties(name, w, h, spp=2):
    """Size-independent checks at a BASELINE frame size (the oracle is too slow there): finite, alpha in [0, 1], something hit,
    the same image twice, and the tile-shard property of the multi-GPU path -- a renderer restricted to a subset of the 16x16
    tiles produces exactly the pixels the full-frame render has there (seed depends on pixel and sample only)."""
    from volren_amd.shard import TileShard
    r = scenes.hip_scene(name, w, h)
    r.render(spp)
    a = r.framebuffer().copy()
    assert np.isfinite(a).all() and a[..., 3].min() >= 0.0 and a[..., 3].max() <= 1.0 and a[..., 3].max() > 0 and a[..., :3].max() > 0
    r.reset()
    r.render(spp)
    assert np.array_equal(_bits(a), _bits(r.framebuffer())), "%s: two renders of the same frame differ" % name
    sh = TileShard(w, h, 8, 3)
    r.set_tiles(sh.mine)
    r.reset()
    r.render(spp)
    b = r.framebuffer()
    tx = (w + 15) // 16
    for t in list(sh.mine)[::max(1, len(sh.mine) // 64)]:
        y0, x0 = (t // tx) * 16, (t % tx) * 16
        assert np.array_equal(_bits(a[y0:y0 + 16, x0:x0 + 16]), _bits(b[y0:y0 + 16, x0:x0 + 16])), "%s: tile %d differs between full frame and shard" % (name, t)


def test_c3_full_size_transfer_function():
    """BASELINE configs[2] at its frame size (1024x1024, smoke.brick + lut.txt: the transfer-function kernel with the decoded float atlas and
    the LUT in LDS): the size-independent properties -- finite, alpha in range, two renders identical, a rank's tile shard reproduces the full
    frame's pixels.  (Bit-exactness against the oracle at sizes it finishes: test_render_matches_oracle[c3].)"""
    _full_resolution_properties("c3", 1024, 1024)


def test_c4_full_size_dense_512():
    """BASELINE configs[3] at its real grid size: synthetic 512^3 dense fp16 grid (256 MiB of voxels: beyond L2 and the
    Infinity Cache's comfort), README parameters, 128 bounces.  Bit for bit against the oracle ON THE SAME 512^3 GRID at a
    frame size the oracle finishes in seconds (its cost scales with pixels, not voxels), then the properties at 1920x1080."""
    o = scenes.oracle_scene("c4:512", 96, 64)
    r = scenes.hip_scene("c4:512", 96, 64)
    r.render(4)
    fb = r.framebuffer()
    assert fb[..., :3].max() > 0 and fb[..., 3].mean() > 0.2
    _assert_same(fb, o.render(4), "c4 512^3 dense fp16")
    del r, o
    _full_resolution_properties("c4:512", 1920, 1080)


def test_c5_full_size_sparse_1024():
    """BASELINE configs[4] at its real grid size: a 1024^3-voxel sparse brick grid (128^3 = 2 M bricks, 65 k allocated) plus a
    temperature grid of the same size, emission on, handed to both sides in brick form (scenes.sparse_brick_arrays_full):
    exercises the 2^21-entry brick tables, the 1 GiB brick-linear atlas addressing and the padded majorant levels that a 64^3
    miniature cannot.  Bit for bit against the oracle, then the properties at 2048x2048."""
    o = scenes.oracle_scene("c5full", 96, 64)
    r = scenes.hip_scene("c5full", 96, 64)
    r.render(4)
    fb = r.framebuffer()
    assert fb[..., :3].max() > 0 and fb[..., 3].max() > 0
    _assert_same(fb, o.render(4), "c5 1024^3 sparse + emission")
    assert r.majorant_blocked == 0                                 # 65 k active bricks: commit() keeps the linear majorant table (round 5)
    # a second view from inside the grid's corner region: large brick indices on every axis
    for x in (r, o):
        x.cam_pos = (0.45, 0.4, 0.48)
        x.cam_dir = (-0.6, -0.5, -0.62)
    r.reset()
    o.sample = 0
    r.render(4)
    _assert_same(r.framebuffer(), o.render(4), "c5 1024^3, corner view")
    del r, o
    _full_resolution_properties("c5full", 2048, 2048)


@pytest.mark.timeout(900)
def test_c5_cloud_at_the_occupancy_the_survey_names():
    """BASELINE configs[4] as SURVEY 8d words it: 1024^3 voxels, wdas_cloud-like occupancy -- 10-20 % of the 2 M bricks allocated -- one connected
    cloud, temperature grid correlated with the density (scenes.cloud_brick_arrays: 352 k bricks = 16.8 %, 99 % of them one component; the
    round-1..3 stand-in `c5full` allocates 3 % in 160 sealed blobs).  Bit for bit against the oracle from outside and from inside the cloud,
    then the properties at 2048 x 2048."""
    from scipy import ndimage
    ad, at = scenes.cloud_brick_arrays(1024)
    nb = 128
    frac = ad["brick_counter"] / nb ** 3
    assert 0.10 <= frac <= 0.20, frac
    rg = ad["rng"].reshape(nb, nb, nb)
    lab, k = ndimage.label((rg >> 16) != (rg & 0xFFFF))
    sizes = np.bincount(lab.ravel())[1:]
    assert sizes.max() >= 0.98 * sizes.sum()                       # one connected cloud (a few detached wisps)
    assert at["brick_counter"] == ad["brick_counter"] and at["min_maj"][1] > 0.5
    o = scenes.oracle_scene("c5cloud", 96, 64)
    r = scenes.hip_scene("c5cloud", 96, 64)
    r.render(4)
    fb = r.framebuffer()
    assert fb[..., :3].max() > 0 and fb[..., 3].mean() > 0.2
    _assert_same(fb, o.render(4), "c5cloud 1024^3")
    # 352 k active bricks: commit() chose the majorant table with levels 0-1 in 4x4x4-cell blocks (kernel variant 4); the linear one gives the same frame
    assert r.majorant_blocked == 1
    r.majorant_layout = 0
    assert r.majorant_blocked == 0
    r.reset(); r.render(4)
    _assert_same(r.framebuffer(), fb, "c5cloud 1024^3, linear majorant table")
    r.majorant_layout = -1
    for x in (r, o):                                               # from inside the cloud's body
        x.cam_pos = (0.05, -0.02, 0.1)
        x.cam_dir = (-0.5, 0.3, -0.81)
    r.reset()
    o.sample = 0
    r.render(4)
    _assert_same(r.framebuffer(), o.render(4), "c5cloud 1024^3, inside view")
    del r, o
    _full_resolution_properties("c5cloud", 2048, 2048)


def test_readme_command_reproduces_the_reference_example_image():
    """The reference's only output artefact, imgs/example.jpg = README.md:72-73 (`-w 1024 -h 1024 --spp 4096 --bounces 128 --albedo 0.8 --phase 0.3
    --density 100 --env_strength 3 --env_rot 270 --exposure 3 --gamma 2.0 --cam_fov 40` on data/smoke.brick + the table-mountain envmap, default
    camera of main.cpp:458-459), rendered by the HIP path at that size and sample count, tonemapped (shader/tonemap.glsl), quantised as
    Texture2D::save_ldr does, and compared with the JPEG after a 4x4 box downsample of both (tests/golden/example_256.npy).  This is the pin of the
    HOST rows (a17 / a18): field of view, inverse(mat3(lookAt)), unit-cube placement and density scaling, env_rot 270 about +y, row order -- a 1 degree
    error in any of them moves the plume's edge by pixels at this size and costs many dB.  Measured: PSNR 51.3 dB, mean RGB within 0.2 of 255
    (what is left is the JPEG's own artefacts); a vertical flip or a 90-degree environment error gives < 20 dB."""
    ref = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "example_256.npy")).astype(np.float64)
    r = scenes.hip_scene("readme", 1024, 1024)
    r.render(4096)
    r.draw()
    tm = r.display()[::-1, :, :3]                                        # PNG / JPEG row order: top row first
    img8 = np.floor(np.clip(tm, 0, 1) * 255 + 0.5).astype(np.float64)
    small = img8.reshape(256, 4, 256, 4, 3).mean((1, 3))
    mse = float(((small - ref) ** 2).mean())
    psnr = 10 * np.log10(255.0 ** 2 / mse)
    dmean = np.abs(small.reshape(-1, 3).mean(0) - ref.reshape(-1, 3).mean(0)).max()
    print("example.jpg at 256x256: PSNR %.2f dB, mean RGB difference %.2f" % (psnr, dmean))
    assert psnr >= 45.0, psnr
    assert dmean < 1.0, dmean
    assert float(((small[::-1] - ref) ** 2).mean()) > 10 * mse           # flipped: far worse
    assert float(((small[:, ::-1] - ref) ** 2).mean()) > 3 * mse          # mirrored: worse


def _hip_emission_scene(w, h):
    """The emission scene of tests/golden/make_golden_glsl.py (emission_scene) on the HIP renderer."""
    import encoder_ref
    import volren_amd
    dens = scenes.synthetic_density(40)
    temp = np.clip(dens * 0.2 + 0.1 * scenes.synthetic_density(40, seed=99), 0, None).astype(np.float32)
    ad, at = encoder_ref.encode_arrays(dens), encoder_ref.encode_arrays(temp)
    r = volren_amd.Renderer(w, h)
    r.load_envmap(scenes.HDR)
    r.set_volume_brick(ad["transform"], ad["n_bricks"], ad["min_maj"], ad["indirection"], ad["rng"], ad["atlas_dim"], ad["atlas"], ad["mips"], commit=False)
    r.set_volume_brick(at["transform"], at["n_bricks"], at["min_maj"], at["indirection"], at["rng"], at["atlas_dim"], at["atlas"], at["mips"], name="temperature", commit=True)
    r.cam_fov, r.bounces, r.albedo, r.emission_scale = 40.0, 8, (0.7, 0.8, 0.9), 50.0
    return r


def _oracle_emission_scene(w, h):
    from oracle import binding as ob
    import encoder_ref
    dens = scenes.synthetic_density(40)
    temp = np.clip(dens * 0.2 + 0.1 * scenes.synthetic_density(40, seed=99), 0, None).astype(np.float32)
    at = encoder_ref.encode_arrays(temp)
    o = ob.OracleRenderer(w, h)
    o.load_envmap(scenes.HDR)
    o.set_volume(encoder_ref.encode(dens), emission=encoder_ref.encode(temp), majorant_emission=at["min_maj"][1])
    o.cam_fov, o.bounces, o.albedo, o.emission_scale = 40.0, 8, (0.7, 0.8, 0.9), 50.0
    return o


@pytest.mark.parametrize("variant", ["lut", "lut_window_env", "global_trackers", "global_trackers_lut", "one_bounce_dark", "crop_no_env", "thin"])
def test_emission_kernel_variants(variant):
    """The emission grid together with the other switches -- a transfer function (the TF + emission kernel instance: collision colour and the lazy
    first line's radiance share a path's LDS slot), the global-majorant trackers (run-time variant), bounce cap / roulette, crop box, hidden
    environment, a thin medium in which most paths never scatter (every one of them a lazy first line): bit for bit the oracle's image."""
    w, h, spp = 64, 48, 5
    r, o = _hip_emission_scene(w, h), _oracle_emission_scene(w, h)
    fields = {
        "lut": dict(),
        "lut_window_env": dict(tf_window_left=0.05, tf_window_width=0.5, show_environment=True),
        "global_trackers": dict(integrator=1),
        "global_trackers_lut": dict(integrator=1),
        "one_bounce_dark": dict(bounces=1, albedo=(0.05, 0.05, 0.05)),
        "crop_no_env": dict(vol_clip_min=(0.1, 0.2, 0.0), vol_clip_max=(0.8, 0.9, 0.7), show_environment=False),
        "thin": dict(density_scale=4.0),
    }[variant]
    if "lut" in variant:
        r.load_transferfunc(scenes.LUT)
        o.load_transferfunc(scenes.LUT)
    for k, v in fields.items():
        setattr(r, k, v)
        setattr(o, k, v)
    r.render(spp)
    fb = r.framebuffer()
    assert np.isfinite(fb).all() and fb[..., :3].max() > 0
    _assert_same(fb, o.render(spp), "emission: " + variant)


@pytest.mark.parametrize("lut", [False, True])
def test_majorant_layout_is_a_per_grid_choice(lut):
    """Round 5 (verdict r4 #3): the majorant table's levels 0-1 linear or in 4x4x4-cell blocks -- a property of the grid, chosen at commit() for the frames the
    two-brick-grid kernel serves (compiled for both: variants 2 and 4) and overridable (vr_set_int "majorant_layout").  Same frame either way, with and
    without a transfer function (whose majorants are TF-remapped floats), after switching back and forth (the table is rebuilt), under the global-majorant
    trackers (run-time variant), and on scenes that have no second layout (the setting is ignored there)."""
    w, h, spp = 72, 56, 5
    r, o = _hip_emission_scene(w, h), _oracle_emission_scene(w, h)
    if lut:
        r.load_transferfunc(scenes.LUT); o.load_transferfunc(scenes.LUT)
    want = o.render(spp).copy()
    assert r.majorant_blocked == 0                                  # a 40^3-voxel grid: far below the threshold
    for layout in (1, 0, 1, -1):
        r.majorant_layout = layout
        assert r.majorant_blocked == (1 if layout == 1 else 0)
        r.reset(); r.render(spp)
        _assert_same(r.framebuffer(), want, "emission scene, majorant layout %d, lut %s" % (layout, lut))
    r.majorant_layout = 1
    r.integrator = 1; o.integrator = 1; o.sample = 0
    r.reset(); r.render(spp)
    _assert_same(r.framebuffer(), o.render(spp), "global trackers with majorant_layout 1")
    c = scenes.hip_scene("c3" if lut else "c2", w, h)               # one brick grid: no blocked variant, the request changes nothing
    c.majorant_layout = 1
    assert c.majorant_blocked == 0
    c.render(spp)
    _assert_same(c.framebuffer(), scenes.oracle_scene("c3" if lut else "c2", w, h).render(spp), "single grid ignores majorant_layout")
    with pytest.raises(Exception):
        r.majorant_layout = 2


def test_hip_against_unmodified_reference_kernel_text():
    """Round 3: the HIP renderer against tests/golden/glsl_golden_r3.npz -- pathtracer_brick_tf.glsl (c3), c1, the README scene and the
    emission path rendered on llvmpipe from the reference's kernel text as it stands (the driver's log / acos / atan; rounds 1-2 spliced the
    specification's in).  At 1024 spp: within the north star's 1e-3 relative L2 (measured 3.8e-4 ... 6.6e-4; 3.4e-4 ... 4.4e-4 of it is
    the systematic offset of llvmpipe's acos / atan in the environment lookups).  At 8 spp a flipped path is a whole pixel: there the
    check is >= 90 % of the pixels within 1e-3 and the image mean within 1e-3."""
    import json
    here = os.path.dirname(os.path.abspath(__file__))
    g = np.load(os.path.join(here, "golden", "glsl_golden_r3.npz"))
    meta = json.load(open(os.path.join(here, "golden", "glsl_golden_r3.json")))
    w, h = meta["width"], meta["height"]
    for name, m in meta["images"].items():
        r = _hip_emission_scene(w, h) if m["config"] == "emission" else scenes.hip_scene(m["config"], w, h)
        r.render(m["spp"])
        hip, ref = r.framebuffer(), g[name]
        rl2 = scenes.rel_l2(hip[..., :3], ref[..., :3])
        if name.startswith("hi_"):
            assert rl2 <= 1e-3, (name, rl2)
        else:
            rel = np.abs(hip.astype(np.float64) - ref)[..., :3].max(-1) / (np.abs(ref[..., :3]).max(-1) + 1e-6)
            assert (rel <= 1e-3).mean() > 0.9 and rl2 < 5e-2 and abs(hip[..., :3].mean() / ref[..., :3].mean() - 1.0) < 1e-3, (name, rl2, float((rel <= 1e-3).mean()))


def test_host_rows_second_pin_on_the_device():
    """Round 5 (verdict r4 #5): rows a17 / a18 -- camera matrix, unit cube, AABB + crop box, density scaling, environment rotation -- pinned a second time.
    tests/golden/glsl_golden_r5.npz holds three further views of smoke.brick rendered by the reference's kernels on llvmpipe with uniform values derived in
    tests/golden/host_rows.py (numpy, glm's documented formulas), not by the oracle.  Here the HIP renderer is set up through the C ABI setters ONLY
    (vr_load_volume, vr_set_float cam_pos / cam_dir / cam_up / cam_fov / env_rot / env_strength / density_scale / vol_clip_*): its uniform block must hold
    the hand-derived values, and its frames must be the reference's (1e-3 relative L2 at 1024 spp; >= 99 % of the pixels to 1e-5 at 8 spp)."""
    import ctypes as C
    import json
    from oracle import binding as ob
    here = os.path.dirname(os.path.abspath(__file__))
    g = np.load(os.path.join(here, "golden", "glsl_golden_r5.npz"))
    meta = json.load(open(os.path.join(here, "golden", "glsl_golden_r5.json")))
    w, h = meta["width"], meta["height"]
    for name in meta["scenes"]:
        import volren_amd
        r = scenes.configure_r5(volren_amd.Renderer(w, h), name, False)
        u = ob.Params.from_buffer_copy(r.uniforms_bytes())                      # (the struct layout is shared: test_uniforms_match_oracle)
        for key in g.files:
            if key.startswith("u_%s_" % name):
                mine = np.asarray(getattr(u, key[len("u_%s_" % name):]), np.float64).reshape(-1)
                want = g[key].astype(np.float64).reshape(-1)
                assert (np.abs(mine - want) <= 3e-7 * np.maximum(np.abs(want), 1.0)).all(), (key, mine, want)
        r.render(meta["spp"])
        hip, ref = r.framebuffer(), g["img_" + name]
        rel = np.abs(hip.astype(np.float64) - ref)[..., :3].max(-1) / (np.abs(ref[..., :3]).max(-1) + 1e-6)
        assert (rel <= 1e-5).mean() > 0.99 and abs(hip[..., :3].mean() / ref[..., :3].mean() - 1.0) < 1e-3, (name, float((rel <= 1e-5).mean()))
        r.reset()
        r.render(meta["hi_spp"])
        assert scenes.rel_l2(r.framebuffer()[..., :3], g["hi_" + name][..., :3]) <= 1e-3, name
        # and bit for bit the oracle set up the same way
        o = scenes.configure_r5(ob.OracleRenderer(w, h), name, True)
        _assert_same(r.framebuffer(), o.render(meta["hi_spp"]), "r5 scene " + name)


def test_hip_against_reference_glsl_golden():
    """The north star's check itself: the HIP renderer against images of the reference's GLSL kernels (rendered on Mesa llvmpipe in
    the build container, tests/golden/glsl_golden.npz), same seed, same spp.  HIP == standard oracle bit for bit, so the
    numbers are those of tests/test_glsl_pin.py::test_standard_oracle_matches_up_to_stochastic_flips: >= 99.5 % of the pixels
    agree to 1e-5; the few others are pixel-samples where a last-bit difference (llvmpipe never fuses multiply-add) flipped
    a stochastic decision, which is also what bounds the relative L2 at this low sample count."""
    import json
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "glsl_golden.npz"))
    meta = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "glsl_golden.json")))
    w, h, spp = meta["width"], meta["height"], meta["spp"]
    for name in ("c2_white_driver", "c2_hdr_spec", "readme_hdr_spec", "c3_tf_spec", "c2_global_spec"):
        m = meta["images"][name]
        r = scenes.hip_scene(m["config"], w, h)
        if m["white_env"]:
            r.set_envmap(np.ones((1, 1, 3), np.float32))
        r.integrator = m.get("integrator", 0)
        r.render(spp)
        hip = r.framebuffer()
        ref = g["img_" + name]
        d = np.abs(hip.astype(np.float64) - ref)
        rel = d[..., :3].max(-1) / (np.abs(ref[..., :3]).max(-1) + 1e-6)
        rl2 = scenes.rel_l2(hip[..., :3], ref[..., :3])
        tol = 1e-3 if name == "c3_tf_spec" or name.endswith("global_spec") else 1e-5
        assert (rel <= tol).mean() > 0.995, (name, float((rel <= tol).mean()))
        assert rl2 < 5e-2 and abs(hip[..., :3].mean() / ref[..., :3].mean() - 1.0) < 1e-3, (name, rl2)


def test_hip_within_1e3_of_reference_glsl_at_1024_spp():
    """BASELINE north_star: "output within 1e-3 relative L2 of the GLSL reference" -- asserted at a sample count where the
    statement is about the renderer and not about one flipped path: tests/golden/glsl_golden_r2.npz holds the reference's
    kernels run for 1024 dispatches per image (make_golden_glsl.py --r2).  Measured: 1e-4 ... 8e-4 (tests/test_glsl_pin.py
    asserts the same for the oracle, which the HIP kernels equal bit for bit).  The opt-in tolerance mode (fast_math) is held
    to the same bar; behind a transfer function (c3: a dark image carried by a few bright pixels) it measured 1.9e-3 at this
    frame size, so the renderer refuses it there (RendererHIP::launch) -- asserted below."""
    import json
    import volren_amd
    here = os.path.dirname(os.path.abspath(__file__))
    g = np.load(os.path.join(here, "golden", "glsl_golden_r2.npz"))
    meta = json.load(open(os.path.join(here, "golden", "glsl_golden_r2.json")))
    w, h, spp = meta["width"], meta["height"], meta["hi_spp"]
    for name in ("hi_c2_white_driver", "hi_c2_hdr_spec", "hi_c3_tf_spec", "hi_readme_hdr_spec", "hi_c1_hdr_spec"):
        m = meta["images"][name]
        r = scenes.hip_scene(m["config"], w, h)
        if m["white_env"]:
            r.set_envmap(np.ones((1, 1, 3), np.float32))
        for fast in (0, 1):
            r.fast_math = fast
            r.reset()
            if fast and name == "hi_c3_tf_spec":
                with pytest.raises(volren_amd.VolrenError, match="fast_math"):
                    r.render(spp)
                continue
            r.render(spp)
            rl2 = scenes.rel_l2(r.framebuffer()[..., :3], g[name][..., :3])
            assert rl2 <= 1e-3, (name, "fast_math" if fast else "bit-exact", rl2)


@pytest.mark.parametrize("n_entries", [2, 256, 300])
def test_transfer_function_lut_paths(n_entries):
    """The transfer-function kernels stage LUTs of up to 256 entries in LDS and read larger ones from global memory; brick
    grids are sampled through the decoded float atlas (default) or the byte atlas (`tf_float_atlas = 0`).  Every combination
    must equal the oracle bit for bit."""
    rs = np.random.RandomState(n_entries)
    lut = rs.uniform(0, 1, (n_entries, 4)).astype(np.float32)
    lut[:, 3] = np.sort(lut[:, 3])                       # monotone alpha: used as given (no CDF fix-up)
    o = scenes.oracle_scene("c2", 64, 48)
    o.set_transferfunc(lut)
    o.tf_window_left, o.tf_window_width = 0.02, 0.7
    want = o.render(6).copy()
    for float_atlas in (1, 0):
        r = scenes.hip_scene("c2", 64, 48)
        r.tf_float_atlas = float_atlas
        r.set_transferfunc(lut)
        r.tf_window_left, r.tf_window_width = 0.02, 0.7
        r.render(6)
        _assert_same(r.framebuffer(), want, "LUT with %d entries, float atlas %d" % (n_entries, float_atlas))
        r.set_transferfunc(None)                        # back to the no-TF kernel: the decoded atlas is dropped, the image is c2's
        r.reset()
        r.render(2)
        assert np.isfinite(r.framebuffer()).all()


def test_lut_file_with_blank_and_short_lines(tmp_path):
    """TransferFunction::load_from_file pushes a row per LINE (src/transferfunc.cpp:86-91), also for a blank line or one with fewer than four numbers,
    whose missing components keep the previous row's values: tf_size is the line count, in the product and in the oracle (verdict r4 weak #9)."""
    import struct
    path = tmp_path / "gaps.txt"
    path.write_text("0.1, 0.2, 0.3, 0.0\n\n0.9, 0.1\n0.3, 0.3, 0.8, 0.7\n\n0.5, 0.5, 0.5, 1.0\n")
    from oracle import binding as ob
    rows = ob.load_lut(str(path))
    assert rows.shape == (6, 4)
    assert rows[1].tolist() == rows[0].tolist()                                   # blank line: the previous row again
    assert np.allclose(rows[2], [0.9, 0.1, 0.3, 0.0]) and rows[4].tolist() == rows[3].tolist()
    o = scenes.oracle_scene("c2", 56, 40)
    o.load_transferfunc(str(path))
    r = scenes.hip_scene("c2", 56, 40)
    r.load_transferfunc(str(path))
    u = r.uniforms_bytes()
    assert any(struct.unpack_from("<I", u, off)[0] == 6 for off in range(0, len(u) - 3, 4))      # tf_size = 6 lines
    r.render(5)
    _assert_same(r.framebuffer(), o.render(5), "LUT file with blank / short lines")


def test_raymarch_integrator_matches_oracle_and_reference():
    """integrator = 3: trace_path with the 64-step ray-marching trackers (common.glsl:506-566; code the reference contains but
    calls from no kernel).  Bit for bit against the oracle; the oracle itself is pinned against the reference's text run on
    llvmpipe (tests/test_glsl_pin.py, rm_* images)."""
    for name in ("c2", "c3", "c5:64"):
        o = scenes.oracle_scene(name, 64, 48)
        r = scenes.hip_scene(name, 64, 48)
        o.integrator = 3
        r.integrator = 3
        r.render(8)
        fb = r.framebuffer()
        assert fb[..., :3].max() > 0
        _assert_same(fb, o.render(8), "raymarch integrator, " + name)


@pytest.mark.parametrize("case", ["fov0", "fov180", "cam_inside", "axis_aligned", "huge_density", "zero_albedo"])
def test_degenerate_inputs_match_oracle(case):
    """Non-finite and boundary arithmetic: rays with infinite / NaN components (fov 0 gives z = -0.5 / tan(0) = -inf, hence NaN
    directions), a camera inside the volume, a ray direction with exact zeros (1 / 0 in the DDA set-up), a density scale that
    saturates every majorant, zero albedo.  The device-only shortcuts (saturating voxel indices, NaN guard, guard-banded
    filter tests) must leave the result identical to the oracle's plain arithmetic."""
    def setup(r):
        if case == "fov0":
            r.cam_fov = 0.0
        elif case == "fov180":
            r.cam_fov = 180.0
        elif case == "cam_inside":
            r.cam_pos, r.cam_dir = (0.0, 0.1, 0.0), (0.0, 0.0, -1.0)
        elif case == "axis_aligned":
            r.cam_pos, r.cam_dir, r.cam_fov = (0.0, 0.0, 1.0), (0.0, 0.0, -1.0), 1e-3
        elif case == "huge_density":
            r.density_scale = 1e6            # ~2300 DDA steps per sample, majorants near the fp16 range times 1e6
        elif case == "zero_albedo":
            r.albedo = (0.0, 0.0, 0.0)
    for cfg in ("c1", "c3"):             # stochastic tricubic taps / trilinear + transfer function
        o = scenes.oracle_scene(cfg, 48, 40)
        r = scenes.hip_scene(cfg, 48, 40)
        if cfg == "c3":
            o.bounces = r.bounces = 6
        setup(o)
        setup(r)
        r.render(4)
        _assert_same(r.framebuffer(), o.render(4), "degenerate input: %s (%s)" % (case, cfg))


def test_watchdog_turns_a_non_terminating_input_into_an_error():
    """density_scale = 1e30 overflows every majorant to +inf; the reference's tracker then never terminates (the oracle
    does not either: `timeout 60 python -c ...` in DESIGN.md).  The kernel must not hang the GPU: its watchdog ends a wavefront
    that has neither finished a path nor pulled a work unit for ~3 s of shader clock, the next call reports it, and the renderer
    stays usable."""
    import time
    import volren_amd
    r = scenes.hip_scene("c1", 32, 32)
    good = r.density_scale
    r.density_scale = 1e30
    t0 = time.time()
    with pytest.raises(volren_amd.VolrenError, match="watchdog"):
        r.render(1)
    assert time.time() - t0 < 10.0
    r.density_scale = good
    r.reset()
    r.render(2)
    o = scenes.oracle_scene("c1", 32, 32)
    _assert_same(r.framebuffer(), o.render(2), "render after a watchdog trip")


@pytest.mark.timeout(900)
def test_watchdog_watches_progress_not_duration():
    """A legitimately long launch: thick, nearly white medium (albedo 0.999, 1000 bounces, density x 10 on the dense 128^3 grid), 512 x 512 x 256 spp
    -- two orders of magnitude more work per sample than the bench scenes.  The watchdog restarts with every path a wavefront finishes, so no
    launch is too long for it; a crop is checked bit for bit against the oracle."""
    w = h = 512

    def thick(r):
        r.albedo = (0.999,) * 3
        r.bounces = 1000
        r.density_scale = 1000.0
        return r
    r = thick(scenes.hip_scene("c4:128", w, h))
    r.launch_target_ms = 0                        # one launch for the whole frame: duration is not what the watchdog measures
    r.render(256)
    assert r.last_launches == 1
    ms = r.last_kernel_ms()
    fb = r.framebuffer()
    x0, y0, cw, ch = 248, 250, 12, 6              # through the thick of the medium
    o = thick(scenes.oracle_scene("c4:128", w, h))
    ref = o.render(256, rect=(x0, y0, x0 + cw, y0 + ch))
    assert np.array_equal(_bits(fb[y0:y0 + ch, x0:x0 + cw]), _bits(ref[y0:y0 + ch, x0:x0 + cw])), "crop differs (launch took %.0f ms)" % ms
    assert fb[y0:y0 + ch, x0:x0 + cw, 3].min() == 1.0           # every sample of the crop scattered


def test_launches_are_sized_by_the_measured_rate():
    """launch_target_ms: a render is split into sub-launches planned from the rate the renderer measured (a probe launch first when it has none and
    the request is large); the image does not depend on the split."""
    w = h = 512
    a = scenes.hip_scene("c2", w, h)
    a.launch_target_ms = 0
    a.render(320)
    assert a.last_launches == 1
    ref = a.framebuffer()
    b = scenes.hip_scene("c2", w, h)
    b.launch_target_ms = 4                        # the frame takes ~20 ms: several sub-launches, the first of them the probe
    b.render(320)
    n1 = b.last_launches
    assert n1 >= 3, n1
    assert np.array_equal(_bits(b.framebuffer()), _bits(ref))
    b.reset()
    b.render(320)                                 # the rate is known now: no probe, the same plan otherwise
    assert b.last_launches >= 2
    assert np.array_equal(_bits(b.framebuffer()), _bits(ref))
    b.launch_target_ms = 2000
    b.reset()
    b.render(320)
    assert b.last_launches == 1


def test_sample_pool_falls_back_when_memory_is_short():
    """The per-sample radiance pool is sized for 288 GB of HBM (one launch per frame); when that much cannot be allocated the
    frame is split into more launches instead of failing.  Same image either way (the running mean is applied in sample
    order).  The shortage is simulated (vr_test_alloc_cap_mb, devmem.h): the GPU may be shared, nothing is hogged."""
    import volren_amd
    lib = volren_amd.load()
    r = scenes.hip_scene("c1", 1024, 1024)
    r.render(8)                                   # allocates everything else first
    lib.vr_test_alloc_cap_mb(3072)                # a 512-spp frame wants 8 GiB of pool
    try:
        r.reset()
        r.render(512)
        launches = r.last_launches
        img = r.framebuffer().copy()
    finally:
        lib.vr_test_alloc_cap_mb(-1)
    assert launches >= 2, launches
    r2 = scenes.hip_scene("c1", 1024, 1024)
    r2.launch_target_ms = 0                       # no probe launch (test_launches_are_sized_by_the_measured_rate): the pool alone decides
    r2.render(512)
    assert r2.last_launches == 1
    _assert_same(img, r2.framebuffer(), "split launches vs one launch")


@pytest.mark.parametrize("config", ["c3", "c4:64", "c5:32"])
def test_frames_split_into_sub_launches_are_identical(config):
    """A frame whose samples do not fit the sample pool is rendered by several launches (BASELINE configs[3..4] need 8 and 16 of them at their own
    frames): the running mean is applied in sample order across launches, so the image is the single-launch image bit for bit -- on the
    transfer-function, dense-grid and emission kernels, with a sample count that leaves a ragged last launch."""
    w, h, spp = 200, 136, 100
    one = scenes.hip_scene(config, w, h)
    one.render(spp)
    assert one.last_launches == 1
    want = one.framebuffer().copy()
    r = scenes.hip_scene(config, w, h)
    r.sample_pool_mb = 16                      # the smallest pool: 208 x 144 pixels (whole 16 x 16 tiles) x 16 bytes -> 32 samples per launch: 32 + 32 + 32 + 4
    r.render(spp)
    assert r.last_launches >= 3, r.last_launches
    assert np.array_equal(_bits(r.framebuffer()), _bits(want))
    assert r.last_pathtrace_ms() > 0 and r.last_pathtrace_ms() <= r.last_kernel_ms() + 1e-3      # the kernels of ALL sub-launches, inside the frame's span


@pytest.mark.parametrize("variant", ["crop", "camera_inside", "very_dense_rgb_albedo", "thin_no_env", "ragged_frame"])
def test_dense_grid_kernel_variants(variant):
    """The dense fp16 kernel (BASELINE configs[3]'s path) under the switches the brick kernel is tested with: crop box, a camera inside the grid,
    an optically thick medium with a coloured albedo, a thin one without visible environment, a frame that is not a multiple of the tile size."""
    w, h = (61, 45) if variant == "ragged_frame" else (64, 48)
    r, o = scenes.hip_scene("c4:64", w, h), scenes.oracle_scene("c4:64", w, h)
    fields = {
        "crop": dict(vol_clip_min=(0.15, 0.0, 0.2), vol_clip_max=(0.85, 0.7, 1.0)),
        "camera_inside": dict(cam_pos=(0.05, 0.0, -0.1), cam_dir=(0.4, 0.2, 1.0), cam_fov=80.0),
        "very_dense_rgb_albedo": dict(density_scale=3000.0, albedo=(0.95, 0.6, 0.3), bounces=24),
        "thin_no_env": dict(density_scale=5.0, show_environment=False),
        "ragged_frame": dict(),
    }[variant]
    for k, v in fields.items():
        setattr(r, k, v)
        setattr(o, k, v)
    r.render(5)
    _assert_same(r.framebuffer(), o.render(5), "dense grid: " + variant)


@pytest.mark.parametrize("config", ["c2", "c3", "c4:64", "c5:32", "c2+global", "c5:32+lut", "c5:32+global"])
def test_no_path_depends_on_stale_cold_state(config, monkeypatch):
    """A new path writes no cold line before its first scatter event (FirstStash, vr_trace.h) and a path that never scatters
    none at all: whatever the workspace held before -- here NaN patterns (VR_TEST_POISON_WORKSPACE, renderer.cpp) -- must
    not reach a result.  c5: with an emission grid the collision code accumulates into the line from the first segment on."""
    monkeypatch.setenv("VR_TEST_POISON_WORKSPACE", "1")
    name, _, mod = config.partition("+")          # +global / +lut as in test_results_do_not_depend_on_the_scheduler: the other kernel variants
    r, o = scenes.hip_scene(name, 96, 64), scenes.oracle_scene(name, 96, 64)
    for x in (r, o):
        if mod == "global":
            x.integrator = 1
        if mod == "lut":
            x.load_transferfunc(scenes.LUT)
    r.render(4)                                   # the workspace is allocated (and poisoned) by the first launch
    monkeypatch.delenv("VR_TEST_POISON_WORKSPACE")
    _assert_same(r.framebuffer(), o.render(4), "poisoned workspace, " + config)


@pytest.mark.parametrize("config", ["c2", "c3", "c4:64", "c5:32", "c2+global", "c3+global", "c5:32+lut", "c5:32+global", "c4:64+lut"])
def test_results_do_not_depend_on_the_scheduler(config):
    """Which lane runs which path when -- event-batch thresholds, the number of lanes that must stand at a collision before the
    collision code runs, the size of the path pool, the samples per work unit -- never changes a result: every setting below gives
    the image the default gives, bit for bit (and that one is the oracle's, asserted by the other tests)."""
    w, h, spp = 72, 56, 6
    name, _, mod = config.partition("+")              # +global: the global-majorant trackers (run-time kernel variant), +lut: a transfer function on top
    r = scenes.hip_scene(name, w, h)
    o = scenes.oracle_scene(name, w, h)
    if mod == "global":
        r.integrator = 1
        o.integrator = 1
    if mod == "lut":
        r.load_transferfunc(scenes.LUT)
        o.load_transferfunc(scenes.LUT)
    r.render(spp)
    want = r.framebuffer().copy()
    _assert_same(want, o.render(spp), config)          # ... and that image is the oracle's
    # NEW, pool cap, hungry, collide threshold, NEE, POSTNEE, ESCAPE
    settings = [[64, 0, 56, 1, 60, 60, 64, 0], [64, 0, 56, 64, 60, 60, 64, 0], [8, 0, 8, 40, 8, 8, 8, 0], [64, 70, 56, 24, 64, 64, 64, 0],
                [1, 66, 1, 1, 1, 1, 1, 0], [64, 0, 64, 63, 64, 64, 64, 0]]
    for s in settings:
        r.set_sched(s)
        r.reset()
        r.render(spp)
        got = r.framebuffer()
        assert np.array_equal(_bits(got), _bits(want)), ("scheduler setting", s)
    r.set_sched([64, 0, 56, 0, 60, 60, 64, 0])
    r.reset()
    for _ in range(spp):                               # one launch per sample (the reference's trace() protocol): pools that never fill
        r.trace()
    r.synchronize()
    assert np.array_equal(_bits(r.framebuffer()), _bits(want)), "trace() x spp"


@pytest.mark.parametrize("config", ["c2", "c3", "c4:64", "c4:64+lut", "c5:32", "c5:32+lut", "c5:32+blocked", "c2+global", "c3+global"])
def test_frames_are_reproducible_on_every_compiled_instance(config, monkeypatch):
    """Verdict r5 #5: one of seven bench runs of round 5 reported a tolerance-mode frame 30 x further from the bit-exact one than usual, once, under rocprofv3, and
    it never came back (200 + 60 full-size frames with NaN-filled workspace and sample pool between them, plain and under rocprofv3 --kernel-trace: one CRC per
    mode, tests/tools_determinism.py, profiles/r6_determinism.txt).  What CAN be pinned is pinned here: on every compiled kernel instance -- each variant, with and
    without a transfer function, the bit-exact arithmetic and, where the mode is offered (no transfer function), the tolerance mode -- 20 frames rendered in turn
    by fresh and reused renderers, the bit-exact and the tolerance kernels alternating on the SAME workspace and pool, both NaN-filled before every launch
    (VR_TEST_POISON_WORKSPACE=2), give ONE frame per mode; the bit-exact one is the oracle's."""
    import zlib
    monkeypatch.setenv("VR_TEST_POISON_WORKSPACE", "2")
    w, h, spp = 80, 48, 5
    name, _, mod = config.partition("+")

    def fresh():
        r = scenes.hip_scene(name, w, h)
        if mod == "global":
            r.integrator = 1
        if mod == "lut":
            r.load_transferfunc(scenes.LUT)
        if mod == "blocked":
            r.majorant_layout = 1                      # variant 4: the kernel for majorant tables in 4x4x4-cell blocks
        return r

    o = scenes.oracle_scene(name, w, h)
    if mod == "global":
        o.integrator = 1
    if mod == "lut":
        o.load_transferfunc(scenes.LUT)
    want = o.render(spp)
    has_tf = mod == "lut" or name == "c3"
    modes = (0,) if has_tf else (0, 1)                 # the tolerance mode is refused behind a transfer function (include/volren_amd.h "fast_math")
    crcs = {m: set() for m in modes}
    keep = []
    for k in range(20):
        if k % 3 == 0:
            keep = (keep + [fresh()])[-2:]             # two renderers alive at a time, as bench.py's tolerance-mode leg has them
        r = keep[-1 - (k % len(keep))] if len(keep) > 1 else keep[-1]
        for m in modes:
            r.fast_math = m
            r.reset()
            r.render(spp)
            fb = r.framebuffer()
            crcs[m].add(zlib.crc32(np.ascontiguousarray(fb).tobytes()))
            if m == 0 and k in (0, 19):
                _assert_same(fb, want, "%s, frame %d" % (config, k))
    assert all(len(v) == 1 for v in crcs.values()), (config, {m: sorted(v) for m, v in crcs.items()})


def test_tile_order_inside_a_launch_never_changes_the_image():
    """order_tiles (default on): a launch works through its tiles costliest first -- by the chord of the pixel rays through the volume's box -- instead of in
    raster order; with a tile subset (a rank's share) the subset is reordered.  Same image bit for bit; and the per-wavefront timeline of an instrumented
    launch is sane (every wavefront starts, finds the queue empty, ends -- in that order)."""
    from volren_amd.shard import TileShard
    w, h, spp = 200, 136, 6
    for config in ("c2", "c5:32"):
        r = scenes.hip_scene(config, w, h)
        assert r.order_tiles == 1                       # default: tile subsets only
        r.order_tiles = 2
        r.render(spp)
        want = r.framebuffer().copy()
        r.order_tiles = 0
        r.reset(); r.render(spp)
        assert np.array_equal(_bits(r.framebuffer()), _bits(want)), config
        r.cam_pos = (0.1, 0.05, 0.12)                  # inside the box: every chord is positive; a new order is computed for the new camera
        r.cam_dir = (-0.5, -0.3, -0.8)
        r.reset(); r.render(spp)
        inside = r.framebuffer().copy()
        r.order_tiles = 2
        r.reset(); r.render(spp)
        assert np.array_equal(_bits(r.framebuffer()), _bits(inside)), config
        r.order_tiles = 1
        sh = TileShard(w, h, 4, 1)
        r.set_tiles(sh.mine)
        r.reset(); r.render(spp)
        part = r.framebuffer()
        tx = (w + 15) // 16
        for t in sh.mine:
            y0, x0 = (t // tx) * 16, (t % tx) * 16
            assert np.array_equal(_bits(part[y0:y0 + 16, x0:x0 + 16]), _bits(inside[y0:y0 + 16, x0:x0 + 16])), (config, t)
    r = scenes.hip_scene("c2", 256, 256)
    r.sched_stats(True)
    r.render(16)
    t = r.wave_timeline()
    r.sched_stats(False)
    assert len(t) > 0 and (t[:, 0] >= 0).all() and (t[:, 1] >= t[:, 0]).all() and (t[:, 2] >= t[:, 1]).all() and t[:, 2].max() < 1.0


@pytest.mark.parametrize("config", ["c2", "c4:64", "c5:32", "c2+global", "c5:32+global"])
def test_instrumented_and_tolerance_kernels_are_consistent(config):
    """The other builds of every kernel variant: the instrumented (STATS) kernels must give the plain kernels' image bit for bit, and the opt-in tolerance-mode
    kernels (fast_math) -- not the oracle's image, but A deterministic image per path -- must give the same image under every scheduler setting."""
    w, h, spp = 72, 56, 5
    name, _, mod = config.partition("+")
    r = scenes.hip_scene(name, w, h)
    if mod == "global":
        r.integrator = 1
    r.render(spp)
    plain = r.framebuffer().copy()
    r.sched_stats(True)
    r.reset(); r.render(spp)
    st = r.sched_stats(False, read=True)
    assert st["waves"] > 0 and np.array_equal(_bits(r.framebuffer()), _bits(plain)), "instrumented kernels"
    r.fast_math = 1
    r.reset(); r.render(spp)
    fast = r.framebuffer().copy()
    assert scenes.rel_l2(fast[..., :3], plain[..., :3]) < 0.05            # a handful of flipped decisions at 5 spp; the 1e-3 bound is asserted at 1024 spp elsewhere
    for thr in ([64, 0, 56, 32, 60, 60, 64, 0], [8, 0, 8, 40, 8, 8, 8, 0], [1, 66, 1, 1, 1, 1, 1, 0]):
        r.set_sched(thr)
        r.reset(); r.render(spp)
        assert np.array_equal(_bits(r.framebuffer()), _bits(fast)), ("tolerance-mode kernels, scheduler", thr)


def test_scheduler_and_launch_fuzz_against_the_oracle():
    """Seeded random combinations of what must never change an image -- scheduler thresholds, pool cap, sample-pool size (how a frame is cut into
    sub-launches), launch sizing by time, tile order, a tile subset -- on every kernel variant, each against the oracle bit for bit.  (Round 4 found a
    variant whose image depended on a threshold: test_emission_grid_with_a_different_brick_layout.)"""
    from volren_amd.shard import TileShard
    rs = np.random.RandomState(4242)
    scenes_ = ["c2", "c3", "c4:64", "c5:32", "c2+global", "c3+global", "c5:32+lut", "c5:32+global", "c4:64+lut", "c5cloud:64"]
    w, h, spp = 80, 48, 5
    for trial in range(30):
        config = scenes_[trial % len(scenes_)]
        name, _, mod = config.partition("+")
        r, o = scenes.hip_scene(name, w, h), scenes.oracle_scene(name, w, h)
        for x in (r, o):
            if mod == "global":
                x.integrator = 1
            if mod == "lut":
                x.load_transferfunc(scenes.LUT)
        cap = int(rs.choice([0, 0, 70, 100, 150]))
        new = int(rs.randint(1, 65))
        if cap:
            new = min(new, cap)
        thr = [new, cap, int(rs.randint(1, 65)), int(rs.randint(1, 65)), int(rs.randint(1, 65)), int(rs.randint(1, 65)), int(rs.randint(1, 65)), 0]
        r.set_sched(thr)
        r.sample_pool_mb = int(rs.choice([16, 16, 64, 16384]))
        r.launch_target_ms = int(rs.choice([0, 1, 2000]))
        r.order_tiles = int(rs.randint(0, 3))
        r.majorant_layout = int(rs.randint(-1, 2))                # per grid / linear / 4x4x4-cell blocks: only the two-brick-grid kernel has both, the others ignore it
        ref = o.render(spp)
        share = None
        if rs.rand() < 0.5:
            share = TileShard(w, h, int(rs.choice([2, 3, 8])), 0).mine
            r.set_tiles(share)
        k = int(rs.randint(1, spp))
        r.render(k)                                   # a frame in two calls
        r.render(spp - k)
        fb = r.framebuffer()
        what = "trial %d: %s thr %s pool %d MB target %d ms order %d majorants %d tiles %s split %d" % (trial, config, thr, r.sample_pool_mb, r.launch_target_ms, r.order_tiles, r.majorant_layout, "share" if share else "all", k)
        if share is None:
            _assert_same(fb, ref, what)
        else:
            tx = (w + 15) // 16
            for t in share:
                y0, x0 = (t // tx) * 16, (t % tx) * 16
                assert np.array_equal(_bits(fb[y0:y0 + 16, x0:x0 + 16]), _bits(ref[y0:y0 + 16, x0:x0 + 16])), what + " tile %d" % t


def test_tuning_state_is_per_renderer():
    """Scheduler thresholds and the statistics switch belong to ONE renderer (vr_set_sched / vr_sched_stats take it): a second renderer
    in the same process keeps the defaults and counts nothing while the first one runs instrumented with a 66-slot pool."""
    w, h, spp = 64, 48, 4
    a = scenes.hip_scene("c2", w, h)
    b = scenes.hip_scene("c2", w, h)
    a.set_sched([1, 66, 1, 1, 1, 1, 1, 0])
    a.sched_stats(True)
    a.render(spp)
    b.render(spp)
    sa = a.sched_stats(False, read=True)
    sb = b.sched_stats(False, read=True)
    assert sa["waves"] > 0 and sa["new"][1] == w * h * spp          # every sample went through a's instrumented kernel once
    assert sb["waves"] == 0 and sb["iterations"] == 0                # b's launch was not instrumented
    assert sa["occupancy"]["free"] <= 66                             # a ran with its own pool cap
    assert np.array_equal(_bits(a.framebuffer()), _bits(b.framebuffer()))
    import volren_amd
    for bad in ([65, 0, 56, 0, 60, 60, 64, 0], [64, 0, -1, 0, 60, 60, 64, 0], [64, 193, 56, 0, 60, 60, 64, 0], [64, 40, 56, 0, 60, 60, 64, 0], [64, 0, 56, 0, 60, 60, 256, 0]):
        with pytest.raises(volren_amd.VolrenError):              # out of range, or a NEW threshold the capped pool can never reach: refused, not truncated
            a.set_sched(bad)


def test_random_parameter_sets_match_oracle():
    """Seeded random combinations of the renderer's fields (resolution, samples, bounces, albedo, phase, density scale, environment
    strength / rotation / visibility, camera, transfer function on / off) on smoke.brick: HIP == oracle bit for bit."""
    rs = np.random.RandomState(20260)
    for i in range(10):
        base = "c3" if rs.rand() < 0.4 else "c2"
        w, h, spp = int(rs.randint(3, 70)), int(rs.randint(3, 60)), int(rs.randint(1, 8))
        fields = dict(bounces=int(rs.randint(1, 24)), albedo=tuple(float(x) for x in rs.uniform(0.0, 1.0, 3)), phase=float(rs.uniform(-0.9, 0.9)),
                      density_scale=float(10.0 ** rs.uniform(0.0, 3.0)), env_strength=float(rs.uniform(0.1, 5.0)), show_environment=bool(rs.rand() < 0.7),
                      cam_fov=float(rs.uniform(15.0, 110.0)), seed=int(rs.randint(-1000, 1000)))
        d = rs.normal(size=3)
        d /= np.linalg.norm(d)
        dist = float(rs.uniform(0.2, 2.5))
        fields["cam_pos"] = tuple(float(x) for x in (-d * dist))
        fields["cam_dir"] = tuple(float(x) for x in (d + rs.normal(scale=0.15, size=3)))
        o = scenes.oracle_scene(base, w, h)
        r = scenes.hip_scene(base, w, h)
        rot = float(rs.uniform(0.0, 360.0))
        o.set_env_rot(rot)
        r.env_rot = rot
        for k, v in fields.items():
            setattr(o, k, v)
            setattr(r, k, v)
        r.render(spp)
        _assert_same(r.framebuffer(), o.render(spp), "random set %d: %s %dx%d %d spp %s" % (i, base, w, h, spp, fields))


@pytest.mark.gpu
def test_nan_ray_parameter_on_a_clean_segment():
    """A free-flight draw of exactly 0 (one in 2^24 segments) that meets an empty first cell makes the reference's step back to the collision point 0 / 0: the ray
    parameter is NaN from there on, the collision at "NaN" is still evaluated (every fetch outside the grid, a null collision, its draws consumed) and the segment ends.
    The kernels' CLEAN form of the collision code (vr_trace.h seg_clean: no NaN guard on the density tap) relies on such a tap landing on index -1 by itself (NaN
    converts to voxel 0, every filter test compares false, 0 + 0 - 1) -- this grid's only dense brick sits at voxel 0, where a tap that did not would scatter.
    2^26 camera segments through mostly empty space: four such draws expected; the frame must equal the oracle's bit for bit."""
    import encoder_ref
    vox = np.zeros((64, 64, 64), np.float16)
    vox[0:8, 0:8, 0:8] = np.float16(5.0)
    w = h = 1024
    spp = 64
    o = scenes.oracle_scene("c1", w, h)
    o.set_volume(encoder_ref.encode_dense_fp16(vox))
    r = scenes.hip_scene("c1", w, h)
    r.set_volume_dense_f16(vox)
    for x in (o, r):
        x.bounces = 3
        x.density_scale = 50.0
    want = o.render(spp)
    r.render(spp)
    _assert_same(r.framebuffer(), want, "NaN ray parameter on a clean segment")


@pytest.mark.gpu
@pytest.mark.parametrize("size", [(1, 1), (2, 1), (3, 2), (5, 4), (16, 8), (33, 17)])
def test_small_environment_maps_in_compact_form(size):
    """Round 6: a map whose texels are all exactly RGBE numbers is fetched as one dword per texel, a row's two texels of a bilinear lookup as ONE 8-byte load when they are
    neighbours in memory (vr_trace.h env_texture) -- which they are not at the map's seam (the lookup wraps to column 0), and never in a map one texel wide.  Maps of 1 to
    33 columns, odd and even, send a large share of the lookups across the seam; a float map of the same values (one texel nudged off the RGBE grid) takes the other
    branch.  Both must give the oracle's frame bit for bit, in the scene's own kernel and in the run-time one."""
    w, h = size
    rs = np.random.RandomState(w * 100 + h)
    m = rs.randint(0, 256, (h, w, 3)).astype(np.float32)
    m[..., 0] = rs.randint(128, 256, (h, w))                       # the largest component's 8-bit mantissa has its top bit set: the texel IS its own RGBE encoding
    m[..., 1:] = np.minimum(m[..., 1:], m[..., :1])
    env = (m * np.exp2(rs.randint(-12, 2, (h, w, 1)).astype(np.float32))).astype(np.float32)
    env[0, 0] = 0.0                                                # (an exactly black texel: e = 0)
    o = scenes.oracle_scene("c2", 72, 48)
    r = scenes.hip_scene("c2", 72, 48)
    for x in (o, r):
        x.set_envmap(env)
    assert r.env_compact == 1
    want = o.render(6).copy()
    r.render(6)
    _assert_same(r.framebuffer(), want, "compact %dx%d environment" % (w, h))
    r.integrator = 1; o.integrator = 1                             # the run-time variant (global-majorant trackers): its own copy of the lookup
    for x in (o, r):
        x.reset()
    want = o.render(3).copy()
    r.render(3)
    assert r.kernel_variant == 3
    _assert_same(r.framebuffer(), want, "compact %dx%d environment, run-time variant" % (w, h))
    env2 = env.copy()
    env2[h - 1, w - 1, 1] = np.float32(1.0 / 3.0)                   # not an RGBE number: the whole map keeps its float form
    r.integrator = 0; o.integrator = 0
    for x in (o, r):
        x.set_envmap(env2)
        x.reset()
    assert r.env_compact == 0
    want = o.render(4).copy()
    r.render(4)
    _assert_same(r.framebuffer(), want, "float %dx%d environment" % (w, h))


@pytest.mark.gpu
@pytest.mark.parametrize("scene", ["c2", "c5:64"])
def test_environment_whose_warp_table_fails_the_division_check(scene):
    """The kernels compiled for one scene kind take the environment warp's quotients -- and the march's step back -- by vr_math.h div_core, the IEEE sequence
    without its rescaling guards, which is only the same thing for operands of moderate size.  An environment with thresholds below 2^-76 (a texel 1e-30 beside
    texels of 1: env_cdf_kernel's check fails) and a density scale outside [2^-16, 2^24] are therefore rendered by the run-time variant, which divides in full --
    from the grids' own atlases when the scene has an emission grid.  Either way the frame is the oracle's, bit for bit."""
    rs = np.random.RandomState(3)
    env = rs.uniform(0.2, 1.0, (8, 16, 3)).astype(np.float32)
    o = scenes.oracle_scene(scene, 56, 40)
    r = scenes.hip_scene(scene, 56, 40)
    assert r.env_div_safe == 1                                  # the bench's HDR
    o.set_envmap(env)
    r.set_envmap(env)
    assert r.env_div_safe == 1
    own_kind = {"c2": 0, "c5:64": 2}[scene]
    assert (r.kernel_variant, r.kernel_variant_reason) == (own_kind, 0)      # the scene's own kernel (vr_get_int "kernel_variant", include/volren_amd.h)
    r.render(4)
    _assert_same(r.framebuffer(), o.render(4).copy(), "random environment")
    env[2:5, 4:7] = 1e-30                                      # a flat dark patch: where it meets the bilinear ramp to its neighbours the fine levels' thresholds are ~1e-28
    env[4, 11] = 0.0                                           # (an exactly black texel is no problem: 0 and NaN thresholds are NaN / 0 either way)
    for x in (o, r):
        x.set_envmap(env)
        x.reset()
    assert r.env_div_safe == 0
    assert (r.kernel_variant, r.kernel_variant_reason) == (3, 2)             # the fallback is visible to the caller, with its reason (and said once on stderr)
    r.render(4)
    _assert_same(r.framebuffer(), o.render(4).copy(), "environment with thresholds below 2^-76")
    env = rs.uniform(0.2, 1.0, (8, 16, 3)).astype(np.float32)
    for x in (o, r):
        x.set_envmap(env)
        x.density_scale = 1e-6                                 # outside [2^-16, 2^24]
        x.reset()
    assert r.env_div_safe == 1
    assert (r.kernel_variant, r.kernel_variant_reason) == (3, 4)
    r.render(4)
    _assert_same(r.framebuffer(), o.render(4).copy(), "density scale 1e-6")
