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
