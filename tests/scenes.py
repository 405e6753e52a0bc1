"""Scene set-ups shared by the tests, bench.py and smoke(): the BASELINE.json configs on the oracle renderer and on
the HIP renderer, configured identically."""
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIX = os.path.join(ROOT, "tests", "fixtures")
SMOKE = os.path.join(FIX, "smoke.brick")
HDR = os.path.join(FIX, "table_mountain_2_puresky_1k.hdr")
LUT = os.path.join(FIX, "lut.txt")

# name -> settings.  c1/c2/c3 are BASELINE.json configs[0..2]; "readme" is README.md:72-73 of the reference.
CONFIGS = {
    "c1": dict(bounces=4, cam_fov=40.0),
    "c2": dict(bounces=100, cam_fov=40.0),
    "c3": dict(bounces=100, cam_fov=40.0, lut=True),
    "readme": dict(bounces=128, cam_fov=40.0, albedo=0.8, phase=0.3, density=100.0, env_strength=3.0, env_rot=270.0,
                   exposure=3.0, gamma=2.0),
}


def synthetic_dense_fp16(n=512, seed=1234):
    """BASELINE configs[3] generator: deterministic smoke-like n^3 field (sum of Gaussian blobs on a coarse lattice,
    tricubically upsampled, ~25 % non-zero, max 5.0) as float16 [z][y][x]."""
    from scipy import ndimage
    coarse = max(32, n // 4)
    f = synthetic_density(coarse, seed=seed, blobs=64).astype(np.float32)
    if coarse != n:
        f = ndimage.zoom(f, n / coarse, order=1, mode="nearest", prefilter=False)[:n, :n, :n]
    f = np.clip(f, 0, None)
    return np.ascontiguousarray(f * np.float32(5.0 / max(float(f.max()), 1e-6))).astype(np.float16)


_DENSE_CACHE = {}


def configure_dense(r, is_oracle, n=512):
    """BASELINE configs[3] ('c4'): synthetic n^3 dense fp16 grid + the README.md:72-73 rendering parameters."""
    if n not in _DENSE_CACHE:
        _DENSE_CACHE[n] = synthetic_dense_fp16(n)
    vox = _DENSE_CACHE[n]
    if is_oracle:
        import encoder_ref
        r.set_volume(encoder_ref.encode_dense_fp16(vox))
    else:
        r.set_volume_dense_f16(vox)
    r.load_envmap(HDR)
    cfg = CONFIGS["readme"]
    r.bounces, r.cam_fov = 128, 40.0
    r.albedo = (cfg["albedo"],) * 3
    r.phase = cfg["phase"]
    r.density_scale = cfg["density"]
    r.env_strength = cfg["env_strength"]
    if is_oracle:
        r.set_env_rot(cfg["env_rot"])
    else:
        r.env_rot = cfg["env_rot"]
    r.tonemap_exposure, r.tonemap_gamma = cfg["exposure"], cfg["gamma"]
    return r


def synthetic_sparse_pair(n=512, seed=4321):
    """BASELINE configs[4] generator: cloud-like sparse density (~15 % of the voxels non-zero, max 5.0) and a temperature
    grid correlated with it, float32 [z][y][x]."""
    from scipy import ndimage
    coarse = max(32, n // 4)
    base = synthetic_density(coarse, seed=seed, blobs=48).astype(np.float32)
    det = synthetic_density(coarse, seed=seed + 1, blobs=96).astype(np.float32)
    dens = np.maximum(base - np.quantile(base, 0.85 - 0.25) * 0 - 0.35 * base.max(), 0)     # keep the cores of the blobs
    temp = np.maximum(dens * 0.6 + 0.2 * det * (dens > 0), 0)
    if coarse != n:
        z = n / coarse
        dens = ndimage.zoom(dens, z, order=1, mode="nearest", prefilter=False)[:n, :n, :n]
        temp = ndimage.zoom(temp, z, order=1, mode="nearest", prefilter=False)[:n, :n, :n]
    dens = np.ascontiguousarray(np.clip(dens, 0, None) * np.float32(5.0 / max(float(dens.max()), 1e-6)), np.float32)
    temp = np.ascontiguousarray(np.clip(temp, 0, None) * np.float32(1.0 / max(float(temp.max()), 1e-6)), np.float32)
    return dens, temp


def configure_sparse(r, is_oracle, n=512):
    """BASELINE configs[4] ('c5'): synthetic sparse brick grid + temperature grid, emission on (common.glsl:324-328,489)."""
    key = ("c5", n)
    if key not in _DENSE_CACHE:
        _DENSE_CACHE[key] = synthetic_sparse_pair(n)
    dens, temp = _DENSE_CACHE[key]
    if is_oracle:
        import encoder_ref
        gd, gt = encoder_ref.encode(dens), encoder_ref.encode(temp)
        for g in (gd, gt):                          # an in-memory dense grid keeps its voxel extent (see grid_to_device)
            g.extent = (n, n, n)
            g.c.extent[:] = g.extent
        r.set_volume(gd, emission=gt, majorant_emission=float(temp.max()))
    else:
        r.set_volume_dense(dens, commit=False)
        r.set_volume_dense(temp, name="temperature", commit=True)
    r.load_envmap(HDR)
    r.bounces, r.cam_fov = 128, 40.0
    r.albedo = (0.9, 0.9, 0.9)
    r.phase = 0.3
    r.density_scale = 100.0
    r.emission_scale = 100.0
    return r


def configure(r, name, is_oracle):
    if name.startswith("c5full"):
        return configure_sparse_full(r, is_oracle, int(name[7:]) if len(name) > 7 else 1024)
    if name.startswith("c5"):
        return configure_sparse(r, is_oracle, int(name[3:]) if len(name) > 3 else 512)
    if name.startswith("c4"):
        return configure_dense(r, is_oracle, int(name[3:]) if len(name) > 3 else 512)
    """Apply a config in the reference's command-line order (paths first, then overrides: main.cpp:360-435)."""
    cfg = CONFIGS[name]
    r.load_volume(SMOKE)
    r.load_envmap(HDR)
    if cfg.get("lut"):
        r.load_transferfunc(LUT)
    r.bounces = cfg["bounces"]
    r.cam_fov = cfg["cam_fov"]
    if "albedo" in cfg:
        r.albedo = (cfg["albedo"],) * 3
    if "phase" in cfg:
        r.phase = cfg["phase"]
    if "density" in cfg:
        r.density_scale = cfg["density"]
    if "env_strength" in cfg:
        r.env_strength = cfg["env_strength"]
    if "env_rot" in cfg:
        if is_oracle:
            r.set_env_rot(cfg["env_rot"])
        else:
            r.env_rot = cfg["env_rot"]
    if "exposure" in cfg:
        r.tonemap_exposure = cfg["exposure"]
        r.tonemap_gamma = cfg["gamma"]
    return r


def oracle_scene(name, w, h):
    from oracle import binding as ob
    return configure(ob.OracleRenderer(w, h), name, True)


def hip_scene(name, w, h, device=0):
    import volren_amd
    return configure(volren_amd.Renderer(w, h, device=device), name, False)


def rel_l2(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.sqrt(((a - b) ** 2).sum() / max((b ** 2).sum(), 1e-30)))


def synthetic_density(n, seed=1234, blobs=24):
    """Deterministic smoke-like dense field [z][y][x] in [0, 5], ~25 % non-zero (SURVEY 8d, C4 generator, small)."""
    rs = np.random.RandomState(seed)
    z, y, x = np.meshgrid(*(np.linspace(0, 1, n, dtype=np.float32),) * 3, indexing="ij")
    f = np.zeros((n, n, n), np.float32)
    for _ in range(blobs):
        c = rs.uniform(0.2, 0.8, 3).astype(np.float32)
        s = np.float32(rs.uniform(0.05, 0.15))
        a = np.float32(rs.uniform(0.3, 1.0))
        f += a * np.exp(-((x - c[0]) ** 2 + (y - c[1]) ** 2 + (z - c[2]) ** 2) / (2 * s * s))
    thr = np.quantile(f, 0.75)
    f = np.maximum(f - thr, 0)
    return (f * (5.0 / f.max())).astype(np.float32)


def _range_mips(lo, hi):
    """Three (min of mins, max of maxes) levels over 2x2x2 children, as encoder_ref.encode_arrays builds them."""
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
    return mips


_SPARSE_FULL = {}


def sparse_brick_arrays_full(n=1024, chunks=160, seed=777):
    """BASELINE configs[4]'s grid SIZE (n^3 voxels, (n/8)^3 bricks, a few per cent of them allocated) built directly in brick
    form -- the dense n^3 array is never materialised, so the scene takes seconds, not minutes: `chunks` cells of a 64-voxel
    lattice each hold one smooth blob that vanishes 4 voxels inside the cell border (so the +-2-voxel range dilation never
    crosses cells and every cell can be encoded on its own by the numpy reference encoder), everything else is empty.
    Returns (density, temperature) as encode_arrays()-style dicts: the exact input of vr_set_volume_brick / oracle Grid.set."""
    key = (n, chunks, seed)
    if key in _SPARSE_FULL:
        return _SPARSE_FULL[key]
    import encoder_ref
    c = 64
    cells = n // c
    nb = n // 8
    rs = np.random.RandomState(seed)
    # blobs cluster around the centre so that a camera looking at the origin sees overlapping ones
    pick = set()
    while len(pick) < chunks:
        p = tuple(np.clip(np.round(rs.normal(cells / 2 - 0.5, cells / 6, 3)), 0, cells - 1).astype(int))
        pick.add(p)
    zz, yy, xx = np.meshgrid(*(np.arange(c, dtype=np.float32) - (c - 1) / 2,) * 3, indexing="ij")
    r2 = xx * xx + yy * yy + zz * zz
    out = []
    for which in ("density", "temperature"):
        rng_w = np.zeros((nb, nb, nb), np.uint32)
        ind = np.zeros((nb, nb, nb), np.uint32)
        blocks = []
        for (cz, cy, cx) in sorted(pick):
            rr = np.random.RandomState(hash((cz, cy, cx, seed)) & 0x7FFFFFFF)
            sigma = np.float32(rr.uniform(8.0, 13.0))
            amp = np.float32(rr.uniform(1.0, 5.0))
            f = np.exp(-r2 / (2 * sigma * sigma)) - np.float32(np.exp(-28.0 * 28.0 / (2 * sigma * sigma)))
            wob = 1 + np.float32(0.25) * np.sin(xx * np.float32(rr.uniform(0.2, 0.5))) * np.cos(yy * np.float32(rr.uniform(0.2, 0.5)) + zz * np.float32(0.3))
            f = (np.maximum(f, 0) * wob * amp).astype(np.float32)
            if which == "temperature":
                f = (np.clip(f / amp, 0, 1) ** 2 * np.float32(rr.uniform(0.3, 1.0))).astype(np.float32)
            a = encoder_ref.encode_arrays(f)
            lb = c // 8
            l_ind = a["indirection"].reshape(lb, lb, lb)
            l_rng = a["rng"].reshape(lb, lb, lb)
            l_atlas = a["atlas"].reshape(a["atlas_dim"][2], a["atlas_dim"][1], a["atlas_dim"][0])
            rng_w[cz * lb:(cz + 1) * lb, cy * lb:(cy + 1) * lb, cx * lb:(cx + 1) * lb] = l_rng
            lo = (l_rng & 0xFFFF).astype(np.uint16).view(np.float16)
            hi = (l_rng >> 16).astype(np.uint16).view(np.float16)
            for bz, by, bx in zip(*np.nonzero(lo != hi)):
                v = int(l_ind[bz, by, bx])
                px, py, pz = v >> 22, (v >> 12) & 1023, (v >> 2) & 1023
                blocks.append(((cz * lb + bz, cy * lb + by, cx * lb + bx), l_atlas[pz * 8:pz * 8 + 8, py * 8:py * 8 + 8, px * 8:px * 8 + 8]))
        per_layer = nb * nb
        layers = max(1, (len(blocks) + per_layer - 1) // per_layer)
        atlas = np.zeros((layers * 8, nb * 8, nb * 8), np.uint8)
        for k, ((bz, by, bx), blk) in enumerate(blocks):
            px, py, pz = k % nb, (k // nb) % nb, k // per_layer
            ind[bz, by, bx] = (px << 22) | (py << 12) | (pz << 2)
            atlas[pz * 8:pz * 8 + 8, py * 8:py * 8 + 8, px * 8:px * 8 + 8] = blk
        lo_f = (rng_w & 0xFFFF).astype(np.uint16).view(np.float16).astype(np.float32)
        hi_f = (rng_w >> 16).astype(np.uint16).view(np.float16).astype(np.float32)
        out.append(dict(transform=np.eye(4, dtype=np.float32).reshape(16), n_bricks=(nb, nb, nb), min_maj=(float(lo_f.min()), float(hi_f.max())),
                        brick_counter=len(blocks), indirection=ind.reshape(-1), rng=rng_w.reshape(-1),
                        atlas_dim=(nb * 8, nb * 8, layers * 8), atlas=atlas.reshape(-1), mips=_range_mips(lo_f, hi_f)))
    _SPARSE_FULL[key] = tuple(out)
    return _SPARSE_FULL[key]


def configure_sparse_full(r, is_oracle, n=1024):
    """The c5 settings (configure_sparse) on the brick-form full-size grid of sparse_brick_arrays_full()."""
    ad, at = sparse_brick_arrays_full(n)
    if is_oracle:
        from oracle import binding as ob

        def grid(a):
            g = ob.Grid()
            g.set(a["transform"], a["n_bricks"], a["min_maj"], a["brick_counter"], a["indirection"], a["rng"], a["atlas_dim"], a["atlas"], a["mips"])
            return g
        r.set_volume(grid(ad), emission=grid(at), majorant_emission=at["min_maj"][1])
    else:
        r.set_volume_brick(ad["transform"], ad["n_bricks"], ad["min_maj"], ad["indirection"], ad["rng"], ad["atlas_dim"], ad["atlas"], ad["mips"], commit=False)
        r.set_volume_brick(at["transform"], at["n_bricks"], at["min_maj"], at["indirection"], at["rng"], at["atlas_dim"], at["atlas"], at["mips"], name="temperature", commit=True)
    r.load_envmap(HDR)
    r.bounces, r.cam_fov = 128, 40.0
    r.albedo = (0.9, 0.9, 0.9)
    r.phase = 0.3
    r.density_scale = 100.0
    r.emission_scale = 100.0
    return r
