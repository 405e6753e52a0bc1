"""ctypes binding of the CPU oracle (oracle/liboracle.so).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module; the product package volren_amd never does.  Pinned against the reference's GLSL kernels run here
(tests/test_glsl_pin.py, see volren_oracle.h).

`OracleRenderer` mirrors the call protocol of the reference's RendererOpenGL
(/root/reference/src/renderer.h:16-63, src/main.cpp:37-81,360-435): load_volume /
load_envmap / load_transferfunc, public fields, render(spp) -> RGBA32F framebuffer
with row 0 at the bottom.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

c_f = C.c_float
c_i = C.c_int32
c_u = C.c_uint32
P_f = C.POINTER(c_f)
P_u32 = C.POINTER(c_u)
P_u8 = C.POINTER(C.c_uint8)


class BrickGrid(C.Structure):
    _fields_ = [("transform", c_f * 16), ("n_bricks", c_u * 3), ("min_maj", c_f * 2),
                ("brick_counter", C.c_uint64), ("indirection", P_u32), ("range", P_u32),
                ("atlas_dim", c_u * 3), ("atlas", P_u8), ("n_mips", c_u),
                ("mip_dim", (c_u * 3) * 8), ("mips", P_u32 * 8),
                ("extent", c_u * 3), ("dense", C.POINTER(C.c_uint16))]


class Params(C.Structure):
    _fields_ = [("bounces", c_i), ("seed", c_i), ("show_environment", c_i),
                ("cam_pos", c_f * 3), ("cam_fov", c_f), ("cam_transform", c_f * 9),
                ("vol_bb_min", c_f * 3), ("vol_bb_max", c_f * 3),
                ("vol_minorant", c_f), ("vol_majorant", c_f), ("vol_inv_majorant", c_f),
                ("vol_albedo", c_f * 3), ("vol_phase_g", c_f), ("vol_density_scale", c_f),
                ("vol_emission_scale", c_f), ("vol_emission_norm", c_f),
                ("vol_density_transform", c_f * 16), ("vol_density_inv_transform", c_f * 16),
                ("vol_emission_transform", c_f * 16), ("vol_emission_inv_transform", c_f * 16),
                ("tf_size", c_u), ("tf_window_left", c_f), ("tf_window_width", c_f),
                ("env_transform", c_f * 9), ("env_inv_transform", c_f * 9), ("env_strength", c_f),
                ("env_imp_inv_dim", c_f * 2), ("env_imp_base_mip", c_i), ("resolution", c_i * 2),
                ("use_tf", c_i), ("has_emission", c_i), ("integrator", c_i)]


class Scene(C.Structure):
    _fields_ = [("density", C.POINTER(BrickGrid)), ("emission", C.POINTER(BrickGrid)),
                ("tf_lut", P_f), ("envmap", P_f), ("env_w", c_i), ("env_h", c_i),
                ("impmap", P_f), ("imp_dim", c_i)]


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("samples", "n_dda_sv", "n_dda_tr", "n_coll_sv", "n_coll_tr",
                                           "n_nee", "n_esc", "n_primary_miss", "n_tf_lookup")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


def build(force=False):
    """Compile oracle/liboracle.so with the committed Makefile (building the checker is not using it)."""
    if force or not os.path.exists(_LIB_PATH) or any(
            os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_LIB_PATH)
            for f in ("volren_oracle.c", "volren_oracle.h", "oracle_math.h", "Makefile")):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        build()
    # VOLREN_ORACLE_SO: an alternative build of the same sources (the unfused variant used by make_golden_glsl.py)
    L = C.CDLL(os.environ.get("VOLREN_ORACLE_SO", _LIB_PATH))
    L.orc_load_brick.argtypes = [C.c_char_p, C.POINTER(BrickGrid)]
    L.orc_load_brick.restype = C.c_int
    L.orc_free_brick.argtypes = [C.POINTER(BrickGrid)]
    L.orc_load_hdr.argtypes = [C.c_char_p, C.POINTER(P_f), C.POINTER(c_i), C.POINTER(c_i)]
    L.orc_load_hdr.restype = C.c_int
    L.orc_load_lut.argtypes = [C.c_char_p, P_f, c_i]
    L.orc_load_lut.restype = C.c_int
    L.orc_free.argtypes = [C.c_void_p]
    L.orc_lut_fixup.argtypes = [P_f, c_i]
    L.orc_lut_fixup.restype = C.c_int
    L.orc_impmap_floats.argtypes = [c_i]
    L.orc_impmap_floats.restype = c_i
    L.orc_build_impmap.argtypes = [P_f, c_i, c_i, c_i, P_f]
    L.orc_env_texture.argtypes = [P_f, c_i, c_i, c_f, c_f, P_f]
    L.orc_unit_cube.argtypes = [C.POINTER(BrickGrid), P_f, P_f]
    L.orc_camera.argtypes = [P_f, P_f, P_f, P_f]
    L.orc_env_rotation.argtypes = [c_f, P_f]
    L.orc_volume_uniforms.argtypes = [C.POINTER(Params), C.POINTER(BrickGrid), C.POINTER(BrickGrid),
                                      P_f, c_f, P_f, P_f, c_f]
    L.orc_mat3_inverse.argtypes = [P_f, P_f]
    L.orc_mat4_inverse.argtypes = [P_f, P_f]
    L.orc_mat4_mul.argtypes = [P_f, P_f, P_f]
    L.orc_render.argtypes = [C.POINTER(Params), C.POINTER(Scene), P_f, c_i, c_i, c_i, c_i, c_i, c_i, c_i,
                             C.POINTER(Counters)]
    L.orc_trace_pixel_sample.argtypes = [C.POINTER(Params), C.POINTER(Scene), c_i, c_i, c_i, P_f]
    L.orc_tonemap.argtypes = [P_f, c_i, c_i, c_f, c_f]
    L.orc_tea.argtypes = [c_u, c_u, c_u]
    L.orc_tea.restype = c_u
    L.orc_rng.argtypes = [P_u32]
    L.orc_rng.restype = c_f
    L.orc_lookup_density_brick.argtypes = [C.POINTER(BrickGrid), c_i, c_i, c_i]
    L.orc_lookup_density_brick.restype = c_f
    L.orc_lookup_majorant_raw.argtypes = [C.POINTER(BrickGrid), c_i, c_i, c_i, c_i]
    L.orc_lookup_majorant_raw.restype = c_f
    L.orc_sample_environment.argtypes = [C.POINTER(Params), C.POINTER(Scene), c_f, c_f, P_f, P_f]
    L.orc_sample_phase_hg.argtypes = [P_f, c_f, c_f, c_f, P_f]
    L.orc_phase_hg.argtypes = [c_f, c_f]
    L.orc_phase_hg.restype = c_f
    L.orc_transmittance.argtypes = [C.POINTER(Params), C.POINTER(Scene), P_f, P_f, P_u32]
    L.orc_transmittance.restype = c_f
    L.orc_math.argtypes = [c_i, c_f, c_f]
    L.orc_math.restype = c_f
    L.orc_num_threads.restype = c_i
    _lib = L
    return L


def fptr(a):
    return a.ctypes.data_as(P_f)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


class Grid:
    """A BrickGrid whose arrays are owned by numpy (loaded from .brick or passed in)."""

    def __init__(self):
        self.c = BrickGrid()
        self._keep = []

    @classmethod
    def from_file(cls, path):
        g = cls()
        raw = BrickGrid()
        rc = lib().orc_load_brick(os.fsencode(path), C.byref(raw))
        if rc != 0:
            raise RuntimeError("Unable to read brick grid %s (code %d)" % (path, rc))
        nb = tuple(raw.n_bricks)
        n = nb[0] * nb[1] * nb[2]
        ind = np.ctypeslib.as_array(raw.indirection, (n,)).copy()
        rng = np.ctypeslib.as_array(raw.range, (n,)).copy()
        ad = tuple(raw.atlas_dim)
        atlas = np.ctypeslib.as_array(raw.atlas, (ad[0] * ad[1] * ad[2],)).copy()
        mips = []
        for i in range(raw.n_mips):
            md = tuple(raw.mip_dim[i])
            mips.append((md, np.ctypeslib.as_array(raw.mips[i], (md[0] * md[1] * md[2],)).copy()))
        g.set(np.array(list(raw.transform), np.float32), nb, tuple(raw.min_maj), int(raw.brick_counter),
              ind, rng, ad, atlas, mips)
        lib().orc_free_brick(C.byref(raw))
        return g

    def set(self, transform, n_bricks, min_maj, brick_counter, indirection, rng, atlas_dim, atlas, mips, extent=None, dense=None):
        c = self.c
        self.extent = tuple(int(v) for v in extent) if extent is not None else None
        self.dense = np.ascontiguousarray(dense, np.float16) if dense is not None else None
        c.extent[:] = self.extent if self.extent is not None else (0, 0, 0)
        c.dense = self.dense.view(np.uint16).ctypes.data_as(C.POINTER(C.c_uint16)) if self.dense is not None else None
        self.transform = _f32(transform).reshape(16)
        self.n_bricks = tuple(int(v) for v in n_bricks)
        self.min_maj = (float(min_maj[0]), float(min_maj[1]))
        self.brick_counter = int(brick_counter)
        self.indirection = np.ascontiguousarray(indirection, np.uint32)
        self.range = np.ascontiguousarray(rng, np.uint32)
        self.atlas_dim = tuple(int(v) for v in atlas_dim)
        self.atlas = np.ascontiguousarray(atlas, np.uint8)
        self.mips = [(tuple(int(v) for v in d), np.ascontiguousarray(a, np.uint32)) for d, a in mips]
        c.transform[:] = self.transform.tolist()
        c.n_bricks[:] = self.n_bricks
        c.min_maj[:] = self.min_maj
        c.brick_counter = self.brick_counter
        c.indirection = self.indirection.ctypes.data_as(P_u32)
        c.range = self.range.ctypes.data_as(P_u32)
        c.atlas_dim[:] = self.atlas_dim
        c.atlas = self.atlas.ctypes.data_as(P_u8)
        c.n_mips = len(self.mips)
        for i, (d, a) in enumerate(self.mips):
            c.mip_dim[i][:] = d
            c.mips[i] = a.ctypes.data_as(P_u32)

    @property
    def index_extent(self):
        return self.extent if getattr(self, "extent", None) else tuple(8 * v for v in self.n_bricks)

    def decode_dense(self):
        """Dense float grid [z][y][x] of the decoded voxels (common.glsl:268-275), numpy."""
        nbx, nby, nbz = self.n_bricks
        ind = self.indirection.reshape(nbz, nby, nbx)
        ptr = np.stack([ind >> 22, (ind >> 12) & 1023, (ind >> 2) & 1023], -1).astype(np.int64)
        rg = self.range.reshape(nbz, nby, nbx)
        rmin = (rg & 0xFFFF).astype(np.uint16).view(np.float16).astype(np.float32)
        rmax = (rg >> 16).astype(np.uint16).view(np.float16).astype(np.float32)
        ax, ay, az = self.atlas_dim
        atlas = self.atlas.reshape(az, ay, ax)
        out = np.zeros((nbz * 8, nby * 8, nbx * 8), np.float32)
        o = np.arange(8)
        for bz in range(nbz):
            for by in range(nby):
                for bx in range(nbx):
                    p = ptr[bz, by, bx]
                    blk = atlas[p[2] * 8:p[2] * 8 + 8, p[1] * 8:p[1] * 8 + 8, p[0] * 8:p[0] * 8 + 8]
                    un = blk.astype(np.float32) / np.float32(255.0)
                    lo, hi = rmin[bz, by, bx], rmax[bz, by, bx]
                    out[bz * 8:bz * 8 + 8, by * 8:by * 8 + 8, bx * 8:bx * 8 + 8] = lo + un * (hi - lo)
        del o
        return out


def load_hdr(path):
    """Radiance .hdr -> float32 [H][W][3], rows in file order (top first)."""
    p = P_f()
    w = c_i()
    h = c_i()
    rc = lib().orc_load_hdr(os.fsencode(path), C.byref(p), C.byref(w), C.byref(h))
    if rc != 0:
        raise RuntimeError("Unable to read envmap %s (code %d)" % (path, rc))
    a = np.ctypeslib.as_array(p, (h.value, w.value, 3)).copy()
    lib().orc_free(p)
    return a


def load_lut(path):
    buf = np.zeros((4096, 4), np.float32)
    n = lib().orc_load_lut(os.fsencode(path), fptr(buf), 4096)
    if n < 0:
        raise RuntimeError("Unable to read file: %s" % path)
    return buf[:n].copy()


def lut_fixup(lut):
    out = _f32(lut).copy()
    ran = lib().orc_lut_fixup(fptr(out), out.shape[0])
    return out, bool(ran)


def build_impmap(env_tex, dim=512):
    """env_tex: [H][W][3] in texture order (row 0 = bottom). Returns the flat pyramid."""
    env_tex = _f32(env_tex)
    h, w, _ = env_tex.shape
    out = np.zeros(lib().orc_impmap_floats(dim), np.float32)
    lib().orc_build_impmap(fptr(env_tex), w, h, dim, fptr(out))
    return out


def impmap_levels(flat, dim=512):
    lv = []
    off = 0
    d = dim
    while d >= 1:
        lv.append(flat[off:off + d * d].reshape(d, d))
        off += d * d
        d >>= 1
    return lv


class OracleRenderer:
    """Reference call protocol on top of the oracle. Field names follow src/renderer.h:30-62."""

    def __init__(self, width, height):
        self.w, self.h = int(width), int(height)
        self.sample = 0
        self.sppx = 1024
        self.seed = 42
        self.bounces = 100
        self.tonemap_exposure = 5.0
        self.tonemap_gamma = 2.2
        self.show_environment = True
        self.albedo = (0.9, 0.9, 0.9)
        self.phase = 0.0
        self.density_scale = 1.0
        self.emission_scale = 100.0
        self.vol_clip_min = (0.0, 0.0, 0.0)
        self.vol_clip_max = (1.0, 1.0, 1.0)
        self.integrator = 0
        # camera defaults: src/main.cpp:458-459; fov: always set explicitly (cppgl default unverified)
        self.cam_pos = (1.0, 0.0, 1.0)
        d = -np.array(self.cam_pos, np.float32)
        self.cam_dir = tuple((d / np.float32(np.sqrt(np.float32(d @ d)))).tolist())
        self.cam_up = (0.0, 1.0, 0.0)
        self.cam_fov = 70.0
        # scene
        self.density = None
        self.emission = None
        self.majorant_emission = 0.0
        self.volume_transform = np.eye(4, dtype=np.float32).T.reshape(16).copy()
        self.env_tex = np.ones((1, 1, 3), np.float32)       # renderer.cpp:36-38: 1x1 white
        self.impmap = build_impmap(self.env_tex)
        self.env_transform = np.eye(3, dtype=np.float32).reshape(9).copy()
        self.env_strength = 1.0
        self.lut = None          # uploaded (CDF-fixed) LUT
        self.tf_window_left = 0.0
        self.tf_window_width = 1.0
        self.fb = np.zeros((self.h, self.w, 4), np.float32)
        self.counters = Counters()

    # --- main.cpp:37-81 ---
    def load_volume(self, path):
        self.set_volume(Grid.from_file(path))

    def set_volume(self, grid, emission=None, majorant_emission=0.0):
        self.density = grid
        self.emission = emission
        self.majorant_emission = float(majorant_emission)
        self.density_scale = 1.0
        ds = c_f(self.density_scale)
        vt = np.zeros(16, np.float32)
        lib().orc_unit_cube(C.byref(grid.c), fptr(vt), C.byref(ds))
        self.volume_transform = vt
        self.density_scale = ds.value
        self.sample = 0

    def load_envmap(self, path):
        self.set_envmap(load_hdr(path))

    def set_envmap(self, img_top_first):
        self.env_tex = np.ascontiguousarray(_f32(img_top_first)[::-1])
        self.impmap = build_impmap(self.env_tex)
        self.env_transform = np.eye(3, dtype=np.float32).reshape(9).copy()
        self.env_strength = 1.0
        self.sample = 0

    def load_transferfunc(self, path):
        self.set_transferfunc(load_lut(path))
        self.show_environment = False      # main.cpp:76

    def set_transferfunc(self, lut):
        self.lut, _ = lut_fixup(lut)
        self.sample = 0

    def set_env_rot(self, deg):
        m = np.zeros(9, np.float32)
        lib().orc_env_rotation(deg, fptr(m))
        self.env_transform = m

    def reset(self):
        self.sample = 0

    def resize(self, w, h):
        self.w, self.h = int(w), int(h)
        self.fb = np.zeros((self.h, self.w, 4), np.float32)
        self.sample = 0

    # --- renderer.cpp:78-145 ---
    def params(self):
        p = Params()
        p.bounces = int(self.bounces)
        p.seed = int(self.seed)
        p.show_environment = 1 if self.show_environment else 0
        p.cam_pos[:] = self.cam_pos
        p.cam_fov = self.cam_fov
        ct = np.zeros(9, np.float32)
        lib().orc_camera(fptr(_f32(self.cam_pos)), fptr(_f32(self.cam_dir)), fptr(_f32(self.cam_up)), fptr(ct))
        p.cam_transform[:] = ct.tolist()
        p.vol_albedo[:] = self.albedo
        p.vol_phase_g = self.phase
        p.vol_emission_scale = self.emission_scale
        lib().orc_volume_uniforms(C.byref(p), C.byref(self.density.c),
                                  C.byref(self.emission.c) if self.emission is not None else None,
                                  fptr(self.volume_transform), self.density_scale,
                                  fptr(_f32(self.vol_clip_min)), fptr(_f32(self.vol_clip_max)),
                                  self.majorant_emission)
        if self.lut is not None:
            p.tf_size = self.lut.shape[0]
            p.use_tf = 1
            p.tf_window_left = self.tf_window_left        # transferfunc->set_uniforms: only with a LUT (renderer.cpp:126)
            p.tf_window_width = self.tf_window_width
        p.env_transform[:] = self.env_transform.tolist()
        inv = np.zeros(9, np.float32)
        lib().orc_mat3_inverse(fptr(self.env_transform), fptr(inv))
        p.env_inv_transform[:] = inv.tolist()
        p.env_strength = self.env_strength
        p.env_imp_inv_dim[:] = (1.0 / 512.0, 1.0 / 512.0)
        p.env_imp_base_mip = 9
        p.resolution[:] = (self.w, self.h)
        p.integrator = int(self.integrator)
        return p

    def scene(self):
        s = Scene()
        s.density = C.pointer(self.density.c)
        s.emission = C.pointer(self.emission.c) if self.emission is not None else None
        s.tf_lut = fptr(self.lut) if self.lut is not None else None
        s.envmap = fptr(self.env_tex)
        s.env_h, s.env_w = self.env_tex.shape[:2]
        s.impmap = fptr(self.impmap)
        s.imp_dim = 512
        return s

    def render(self, spp, rect=None, threads=0):
        """Runs `spp` more samples (trace() x spp, bindings.cpp:124-132 without the reset)."""
        p, s = self.params(), self.scene()
        x0, y0, x1, y1 = rect if rect is not None else (0, 0, self.w, self.h)
        lib().orc_render(C.byref(p), C.byref(s), fptr(self.fb), x0, y0, x1, y1,
                         self.sample + 1, int(spp), int(threads), C.byref(self.counters))
        self.sample += int(spp)
        return self.fb

    def trace_pixel_sample(self, x, y, sample):
        p, s = self.params(), self.scene()
        out = np.zeros(4, np.float32)
        lib().orc_trace_pixel_sample(C.byref(p), C.byref(s), x, y, sample, fptr(out))
        return out

    def tonemapped(self):
        out = self.fb.copy()
        lib().orc_tonemap(fptr(out), self.w, self.h, self.tonemap_exposure, self.tonemap_gamma)
        return out
