"""The reference's host marshalling derived a SECOND time, independently of oracle/ and of the product:
numpy float32 written from glm's documented formulas (glm::lookAt right-handed, glm::rotate = Rodrigues, glm::scale, glm::translate,
column-major matrices) and from the call sites that use them --

    cam_transform              = inverse(mat3(lookAt(pos, pos + dir, up)))                       src/renderer.cpp:94 (cppgl camera: view = lookAt)
    volume->transform          = translate(scale(mat4(1), 1/size), -bb_min - 0.5 extent)         src/renderer.cpp:227-242
    density_scale             *= size                                                             src/renderer.cpp:240
    vol_bb_min / max           = bb_min + vol_clip_* (bb_max - bb_min), bb = AABB of the grid box under volume->transform * grid->transform
                                                                                                  src/renderer.cpp:97-100
    vol_density_transform      = volume->transform * grid->transform, and its inverse            src/renderer.cpp:110-111
    vol_minorant / majorant    = grid min / max * density_scale, vol_inv_majorant = 1 / (...)    src/renderer.cpp:101-103
    env_transform              = mat3(rotate(mat4(1), radians(deg), (0, 1, 0))), and its inverse src/main.cpp:381-382, renderer.cpp:128-129

Used by tests/golden/make_golden_glsl.py --r5 (the uniform values fed to the reference's GLSL on llvmpipe come from HERE, not from
OracleRenderer.params()) and by the tests that compare the product's / the oracle's uniforms with these numbers.  Inputs that are DATA -- the
grid's own index->model matrix, its brick counts and its value range -- are read from the .brick file by the caller."""
import numpy as np

F = np.float32


def _v(x):
    return np.asarray(x, F).reshape(-1)


def normalize(v):
    v = _v(v)
    return (v / F(np.sqrt(F(np.dot(v, v))))).astype(F)


def lookat_rh(eye, center, up):
    """glm::lookAtRH: f = normalize(center - eye), s = normalize(cross(f, up)), u = cross(s, f);
    column-major result m[col][row]: m[i][0] = s[i], m[i][1] = u[i], m[i][2] = -f[i], m[3] = (-dot(s, eye), -dot(u, eye), dot(f, eye), 1)."""
    eye, center, up = _v(eye), _v(center), _v(up)
    f = normalize(center - eye)
    s = normalize(np.cross(f, up).astype(F))
    u = np.cross(s, f).astype(F)
    m = np.zeros((4, 4), F)                       # m[col, row]
    for i in range(3):
        m[i, 0], m[i, 1], m[i, 2] = s[i], u[i], -f[i]
    m[3, 0], m[3, 1], m[3, 2], m[3, 3] = -np.dot(s, eye), -np.dot(u, eye), np.dot(f, eye), 1
    return m


def rotate(m, angle_rad, axis):
    """glm::rotate(m, angle, axis): m * R with R from Rodrigues' formula (axis normalised), column-major."""
    a = F(angle_rad)
    c, s = F(np.cos(a)), F(np.sin(a))
    ax = normalize(axis)
    t = (F(1) - c) * ax
    r = np.zeros((4, 4), F)                       # r[col, row]
    r[0, 0] = c + t[0] * ax[0]; r[0, 1] = t[0] * ax[1] + s * ax[2]; r[0, 2] = t[0] * ax[2] - s * ax[1]
    r[1, 0] = t[1] * ax[0] - s * ax[2]; r[1, 1] = c + t[1] * ax[1]; r[1, 2] = t[1] * ax[2] + s * ax[0]
    r[2, 0] = t[2] * ax[0] + s * ax[1]; r[2, 1] = t[2] * ax[1] - s * ax[0]; r[2, 2] = c + t[2] * ax[2]
    r[3, 3] = 1
    return matmul(m, r)


def matmul(a, b):
    """column-major product a * b for [col, row] arrays: (a b)[c, r] = sum_k a[k, r] b[c, k]"""
    return np.einsum("kr,ck->cr", a.astype(F), b.astype(F)).astype(F)


def scale(m, v):
    out = m.astype(F).copy()
    for i in range(3):
        out[i] = m[i] * F(_v(v)[i] if np.ndim(v) else v)
    return out


def translate(m, v):
    v = _v(v)
    out = m.astype(F).copy()
    out[3] = (m[0] * v[0] + m[1] * v[1] + m[2] * v[2] + m[3]).astype(F)
    return out


def transform_point(m, p):
    p = _v(p)
    return (m[0, :3] * p[0] + m[1, :3] * p[1] + m[2, :3] * p[2] + m[3, :3]).astype(F)


def inverse(m):
    """glm::inverse up to rounding: computed in float64, rounded once"""
    return np.linalg.inv(m.astype(np.float64).T).T.astype(F)


def derive(grid_transform, index_extent, grid_min, grid_max, cam_pos, cam_dir, cam_up, density_scale_flag=None, env_rot_deg=None,
           clip_min=(0, 0, 0), clip_max=(1, 1, 1)):
    """The uniforms of src/renderer.cpp:88-131 that depend on host arithmetic, for ONE density grid loaded the way load_volume does it
    (density_scale = 1, scale_and_move_to_unit_cube) followed by `--density density_scale_flag` (which SETS the scale: src/main.cpp:367-368).
    grid_transform: 16 floats column-major (the grid's index -> model matrix), index_extent: voxels per axis."""
    gt = np.asarray(grid_transform, F).reshape(4, 4)                 # [col, row]
    ext = _v(index_extent)
    # scale_and_move_to_unit_cube
    bb0 = transform_point(gt, (0, 0, 0))
    bb1 = transform_point(gt, ext)
    extent = (bb1 - bb0).astype(F)
    size = F(max(extent))
    density_scale = F(1)
    vt = np.eye(4, dtype=F)
    if size != F(1):
        vt = translate(scale(np.eye(4, dtype=F), F(1) / size), (-bb0 - F(0.5) * extent).astype(F))
        density_scale = density_scale * size
    if density_scale_flag is not None:
        density_scale = F(density_scale_flag)
    dt = matmul(vt, gt)
    # AABB of the grid's box under the full transform (all 8 corners: the transforms here are axis-aligned, min / max of two would do)
    corners = np.array([transform_point(dt, (ext[0] * (k & 1), ext[1] * ((k >> 1) & 1), ext[2] * ((k >> 2) & 1))) for k in range(8)], F)
    wb0, wb1 = corners.min(0), corners.max(0)
    cmin, cmax = _v(clip_min), _v(clip_max)
    view = lookat_rh(cam_pos, _v(cam_pos) + _v(cam_dir), cam_up)
    cam_transform = inverse3(view[:3, :3])
    env = np.eye(3, dtype=F)
    if env_rot_deg is not None:
        env = rotate(np.eye(4, dtype=F), np.radians(F(env_rot_deg)), (0, 1, 0))[:3, :3].copy()
    maj = F(grid_max) * density_scale
    return {
        "cam_pos": _v(cam_pos), "cam_transform": cam_transform.reshape(9),
        "vol_bb_min": (wb0 + cmin * (wb1 - wb0)).astype(F), "vol_bb_max": (wb0 + cmax * (wb1 - wb0)).astype(F),
        "vol_minorant": F(grid_min) * density_scale, "vol_majorant": maj, "vol_inv_majorant": F(1) / maj,
        "vol_density_scale": density_scale,
        "vol_density_transform": dt.reshape(16), "vol_density_inv_transform": inverse(dt).reshape(16),
        "env_transform": env.reshape(9), "env_inv_transform": inverse3(env).reshape(9),
    }


def inverse3(m3):
    return np.linalg.inv(m3.astype(np.float64).T).T.astype(F)


# the three extra views of round 5 (64 x 48, 8 spp, smoke.brick + the HDR environment; fields not listed keep config c2's values)
R5_SCENES = {
    "cam_b": dict(cam_pos=(-0.8, 0.6, 1.3), cam_target=(0.1, -0.05, 0.0), cam_up=(0.1, 1.0, 0.0), cam_fov=55.0),
    "env_rot": dict(cam_pos=(1.0, 0.0, 1.0), cam_target=(0.0, 0.0, 0.0), cam_up=(0.0, 1.0, 0.0), cam_fov=40.0, env_rot=135.0, env_strength=2.0),
    "crop": dict(cam_pos=(0.7, 0.9, -1.1), cam_target=(0.0, 0.0, 0.0), cam_up=(0.0, 1.0, 0.0), cam_fov=40.0, density=60.0,
                 vol_crop_min=(0.1, 0.2, 0.0), vol_crop_max=(0.9, 0.8, 0.7)),
}


def scene_dir(s):
    """what a caller passes as the camera direction: target - position (the reference's camera normalises it in lookAt)"""
    return (_v(s["cam_target"]) - _v(s["cam_pos"])).astype(F)
