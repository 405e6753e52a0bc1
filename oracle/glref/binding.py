"""oracle/glref -- TEST INFRASTRUCTURE ONLY.

Drives the reference's own GLSL compute shaders (read from /root/reference/shader at run time, never copied into this
repository) on Mesa llvmpipe through oracle/_ref/libglref.so (oracle/glref/glref.c), with the GL objects set up the way
src/renderer.cpp:88-218 and src/environment.cpp:11-37 do.  Used by tests/golden/make_golden_glsl.py -- in the build
container only -- to produce the golden vectors that pin the CPU oracle against the reference's kernels themselves.
Nothing under volren_amd/ imports this, and nothing here runs on the GPU box.
"""
import ctypes as C
import os
import re
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF_SHADERS = "/root/reference/shader"
SO = os.path.join(HERE, "..", "_ref", "libglref.so")

_lib = None


def available():
    return os.path.isdir(REF_SHADERS) and os.path.exists("/usr/lib/x86_64-linux-gnu/dri/swrast_dri.so")


def build():
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    src = os.path.join(HERE, "glref.c")
    if not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-o", SO, src, "-ldl"])
    return SO


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        L.glref_error.restype = C.c_char_p
        L.glref_info.restype = C.c_char_p
        L.glref_program.argtypes = [C.c_char_p]
        L.glref_uniform.argtypes = [C.c_int, C.c_char_p, C.c_int, C.c_void_p]
        L.glref_sampler.argtypes = [C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_int]
        L.glref_tex3d.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.glref_tex3d_level.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.glref_tex2d_rgb32f.argtypes = [C.c_int, C.c_int, C.c_void_p]
        L.glref_read_tex2d.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.glref_upload_tex2d.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.glref_ssbo.argtypes = [C.c_int, C.c_void_p, C.c_long]
        L.glref_read_ssbo.argtypes = [C.c_int, C.c_void_p, C.c_long]
        if L.glref_init() != 0:
            raise RuntimeError("glref_init: " + L.glref_error().decode())
        _lib = L
    return _lib


def _ck(rc):
    if rc < 0:
        raise RuntimeError(lib().glref_error().decode())
    return rc


def shader_source(name, prelude="", spec_math=False):
    """Text of /root/reference/shader/<name> with its `#include "file"` lines expanded in place (what cppgl's Shader does)."""
    def expand(path):
        out = []
        for line in open(path).read().split("\n"):
            m = re.match(r'\s*#include\s+"([^"]+)"', line)
            out.append(expand(os.path.join(os.path.dirname(path), m.group(1))) if m else line)
        return "\n".join(out)
    text = _portable(expand(os.path.join(REF_SHADERS, name)))
    if spec_math:       # splice the specification's log/acos/atan in after the #version line (oracle/glref/spec_math.glsl)
        head, rest = text.split("\n", 1)
        text = head + "\n" + open(os.path.join(HERE, "spec_math.glsl")).read() + "\n" + rest
    return prelude + text


def _portable(text):
    """The reference was written against NVIDIA's GLSL compiler, which accepts `||` between boolean VECTORS
    (common.glsl:18-19, tonemap.glsl:27: `isnan(x) || isinf(x)` on vec3/vec4).  Standard GLSL -- and Mesa -- only allow
    scalar operands there, so that one expression is rewritten, in memory, into the component-wise OR it means."""
    return re.sub(
        r"(vec([34]) sanitize\(const vec\2 (\w+)\) \{ return mix\(\3, vec\2\(0\), )isnan\(\3\) \|\| isinf\(\3\)\)",
        lambda m: "%sbvec%s(uvec%s(isnan(%s)) | uvec%s(isinf(%s))))" % (m.group(1), m.group(2), m.group(2), m.group(3), m.group(2), m.group(3)),
        text)


def reference_include(name):
    """For harness shaders of our own that call the reference's functions: the expanded text of one reference file."""
    return shader_source(name)


_KIND = {"i": 0, "u": 1, "f": 2, "2f": 3, "3f": 4, "2i": 5, "m3": 6, "m4": 7}


def set_uniform(prog, name, kind, value):
    a = np.ascontiguousarray(value, np.int32 if kind in ("i", "u", "2i") else np.float32).reshape(-1)
    return _ck(lib().glref_uniform(prog, name.encode(), _KIND[kind], a.ctypes.data))


class GLSLReference:
    """The GL objects of one scene (an oracle.binding.OracleRenderer supplies arrays and uniform values)."""

    def __init__(self, scene, literal_compressed_atlas=False, spec_math=False):
        self.L = lib()
        self.scene = scene
        self.spec_math = spec_math          # run the kernels with the specification's log/acos/atan instead of the driver's
        self.atlas_kind = 3 if literal_compressed_atlas else 2
        self.info = self.L.glref_info().decode()
        self.env_tex = self._env(scene.env_tex)
        self.impmap = self._impmap()
        self.density = self._grid(scene.density)
        self.emission = self._grid(scene.emission) if scene.emission is not None else None
        self.lut_buf = None
        if scene.lut is not None:
            lut = np.ascontiguousarray(scene.lut, np.float32)
            self.lut_buf = _ck(self.L.glref_ssbo(4, lut.ctypes.data, lut.nbytes))        # transferfunc.cpp: SSBO binding 4
        self.programs = {}

    # -- renderer.cpp:159-218 ---------------------------------------------------------------------------------------
    def _grid(self, g):
        nbx, nby, nbz = g.n_bricks
        ax, ay, az = g.atlas_dim
        ind = _ck(self.L.glref_tex3d(0, nbx, nby, nbz, g.indirection.ctypes.data, 0))
        rng = _ck(self.L.glref_tex3d(1, nbx, nby, nbz, g.range.ctypes.data, len(g.mips)))
        for i, (d, a) in enumerate(g.mips):
            _ck(self.L.glref_tex3d_level(rng, i + 1, d[0], d[1], d[2], a.ctypes.data))
        atl = _ck(self.L.glref_tex3d(self.atlas_kind, ax, ay, az, g.atlas.ctypes.data, 0))
        return dict(indirection=ind, range=rng, atlas=atl)

    def _env(self, tex_bottom_first):
        t = np.ascontiguousarray(tex_bottom_first, np.float32)
        h, w, _ = t.shape
        return _ck(self.L.glref_tex2d_rgb32f(w, h, t.ctypes.data))

    # -- environment.cpp:11-37 ----------------------------------------------------------------------------------------
    def _impmap(self, dim=512, samples=64):
        L = self.L
        prog = _ck(L.glref_program(shader_source("env_setup.glsl").encode()))
        imp = _ck(L.glref_tex2d_empty(dim, dim, 1))
        _ck(L.glref_use(prog))
        _ck(L.glref_bind_image(0, imp, 1, 1))
        _ck(L.glref_sampler(prog, b"envmap", 0, self.env_tex, 0))
        n = int(np.sqrt(samples))
        set_uniform(prog, "output_size", "2i", (dim, dim))
        set_uniform(prog, "output_size_samples", "2i", (dim * n, dim * n))
        set_uniform(prog, "num_samples", "2i", (n, n))
        set_uniform(prog, "inv_samples", "f", np.float32(1.0) / np.float32(n * n))
        _ck(L.glref_dispatch((dim + 15) // 16, (dim + 15) // 16, 1))       # cppgl dispatch_compute(w, h): ceil(w / local_size)
        _ck(L.glref_generate_mipmap(imp))
        return imp

    def impmap_levels(self, dim=512):
        out = []
        lvl, d = 0, dim
        while d >= 1:
            a = np.zeros((d, d), np.float32)
            _ck(self.L.glref_read_tex2d(self.impmap, lvl, 1, a.ctypes.data))
            out.append(a)
            lvl += 1
            d >>= 1
        return out

    # -- renderer.cpp:78-145 --------------------------------------------------------------------------------------------
    def _program(self, name):
        if name not in self.programs:
            self.programs[name] = _ck(self.L.glref_program(shader_source(name, spec_math=self.spec_math).encode()))
        return self.programs[name]

    def bind_scene(self, prog, p):
        """Uniforms and textures in the reference's order (renderer.cpp:88-131); p = OracleRenderer.params()."""
        L = self.L
        unit = 0
        set_uniform(prog, "bounces", "i", p.bounces)
        set_uniform(prog, "seed", "i", p.seed)
        set_uniform(prog, "show_environment", "i", p.show_environment)
        set_uniform(prog, "optimization", "i", 0)
        set_uniform(prog, "cam_pos", "3f", list(p.cam_pos))
        set_uniform(prog, "cam_fov", "f", p.cam_fov)
        set_uniform(prog, "cam_transform", "m3", list(p.cam_transform))
        for n in ("vol_bb_min", "vol_bb_max", "vol_albedo"):
            set_uniform(prog, n, "3f", list(getattr(p, n)))
        for n in ("vol_minorant", "vol_majorant", "vol_inv_majorant", "vol_phase_g", "vol_density_scale", "vol_emission_scale", "vol_emission_norm"):
            set_uniform(prog, n, "f", getattr(p, n))
        set_uniform(prog, "vol_density_transform", "m4", list(p.vol_density_transform))
        set_uniform(prog, "vol_density_inv_transform", "m4", list(p.vol_density_inv_transform))
        for k in ("indirection", "range", "atlas"):
            _ck(L.glref_sampler(prog, ("vol_density_" + k).encode(), unit, self.density[k], 1))
            unit += 1
        if self.emission is not None:
            set_uniform(prog, "vol_emission_transform", "m4", list(p.vol_emission_transform))
            set_uniform(prog, "vol_emission_inv_transform", "m4", list(p.vol_emission_inv_transform))
            for k in ("indirection", "range", "atlas"):
                _ck(L.glref_sampler(prog, ("vol_emission_" + k).encode(), unit, self.emission[k], 1))
                unit += 1
        if self.scene.lut is not None:
            set_uniform(prog, "tf_size", "u", p.tf_size)
            set_uniform(prog, "tf_window_left", "f", p.tf_window_left)
            set_uniform(prog, "tf_window_width", "f", p.tf_window_width)
        set_uniform(prog, "env_transform", "m3", list(p.env_transform))
        set_uniform(prog, "env_inv_transform", "m3", list(p.env_inv_transform))
        set_uniform(prog, "env_strength", "f", p.env_strength)
        set_uniform(prog, "env_imp_inv_dim", "2f", list(p.env_imp_inv_dim))
        set_uniform(prog, "env_imp_base_mip", "i", p.env_imp_base_mip)
        _ck(L.glref_sampler(prog, b"env_envmap", unit, self.env_tex, 0))
        unit += 1
        _ck(L.glref_sampler(prog, b"env_impmap", unit, self.impmap, 0))
        return unit + 1

    def _variant_program(self, variant):
        """Kernels the reference contains but does not build: its text with one line changed, in memory.
        "global": pathtracer_brick*.glsl without `#define USE_DDA` -> trace_path uses sample_volume / transmittance with the
        global majorant (common.glsl:333-394, 606-624).  "dvr": main() calls direct_volume_rendering (common.glsl:571-591,
        not called by any kernel of the reference) instead of trace_path."""
        key = "variant:" + variant
        if key not in self.programs:
            name = "pathtracer_brick_tf.glsl" if self.scene.lut is not None else "pathtracer_brick.glsl"
            text = shader_source(name, spec_math=self.spec_math)
            if variant == "global":
                assert text.count("#define USE_DDA") == 1
                text = text.replace("#define USE_DDA", "")
            elif variant == "raymarch":
                # trace_path with the 64-step ray-marching trackers (common.glsl:506-566; no kernel of the reference calls them)
                inc = reference_include("common.glsl")
                for old, new in (("while (sample_volumeDDA(pos, dir, t, throughput, L, seed)) {", "float rm_pdf; while (sample_volume_raymarch(pos, dir, t, throughput, rm_pdf, seed)) {"),
                                 ("const float Tr = transmittanceDDA(pos, w_i, seed);", "const float Tr = transmittance_raymarch(pos, w_i, seed);")):
                    assert text.count(old) == 1, old
                    text = text.replace(old, new)
                del inc
            elif variant == "dvr":
                call = "trace_path(pos, dir, seed)"
                assert text.count(call) == 1
                text = text.replace(call, "vec4(direct_volume_rendering(pos, dir, seed), 1)")
            else:
                raise ValueError(variant)
            self.programs[key] = _ck(self.L.glref_program(text.encode()))
        return self.programs[key]

    def render(self, spp, first_sample=1, variant=None, params=None):
        """`spp` dispatches of pathtracer_brick.glsl (or _tf with a LUT) = RendererOpenGL::trace() x spp. Returns RGBA [H][W][4], row 0 = bottom.
        params: the uniform values to feed instead of the scene's own params() (round 5: values derived independently of the oracle,
        tests/golden/host_rows.py)."""
        s = self.scene
        L = self.L
        prog = self._variant_program(variant) if variant else self._program("pathtracer_brick_tf.glsl" if s.lut is not None else "pathtracer_brick.glsl")
        color = _ck(L.glref_tex2d_empty(s.w, s.h, 4))
        _ck(L.glref_use(prog))
        _ck(L.glref_bind_image(0, color, 4, 0))
        self.bind_scene(prog, params if params is not None else s.params())
        set_uniform(prog, "resolution", "2i", (s.w, s.h))
        for k in range(spp):
            set_uniform(prog, "current_sample", "i", first_sample + k)
            _ck(L.glref_dispatch((s.w + 15) // 16, (s.h + 15) // 16, 1))
        out = np.zeros((s.h, s.w, 4), np.float32)
        _ck(L.glref_read_tex2d(color, 0, 4, out.ctypes.data))
        return out


    # -- function-level probes (oracle/glref/probe.glsl) -----------------------------------------------------------------
    def probe(self, mode, inputs):
        """inputs: [n][8] float32 (bit patterns for integer arguments) -> outputs [n][8] float32."""
        L = self.L
        if "probe" not in self.programs:
            src = open(os.path.join(HERE, "probe.glsl")).read().replace("@COMMON@", reference_include("common.glsl"))
            self.programs["probe"] = _ck(L.glref_program(src.encode()))
        prog = self.programs["probe"]
        a = np.ascontiguousarray(inputs, np.float32).reshape(-1, 8)
        n = a.shape[0]
        out = np.zeros_like(a)
        _ck(L.glref_use(prog))
        self.bind_scene(prog, self.scene.params())
        bi = _ck(L.glref_ssbo(5, a.ctypes.data, a.nbytes))
        bo = _ck(L.glref_ssbo(6, out.ctypes.data, out.nbytes))
        set_uniform(prog, "mode", "i", mode)
        set_uniform(prog, "n_items", "i", n)
        _ck(L.glref_dispatch((n + 63) // 64, 1, 1))
        _ck(L.glref_read_ssbo(bo, out.ctypes.data, out.nbytes))
        del bi
        return out

    # -- tonemap.glsl (renderer.cpp tonemap(): exposure, gamma) ---------------------------------------------------------
    def tonemap(self, rgba, exposure, gamma):
        L = self.L
        a = np.ascontiguousarray(rgba, np.float32)
        h, w, _ = a.shape
        prog = self._program("tonemap.glsl")
        tex = _ck(L.glref_tex2d_empty(w, h, 4))
        _ck(L.glref_upload_tex2d(tex, w, h, a.ctypes.data))
        _ck(L.glref_use(prog))
        _ck(L.glref_bind_image(0, tex, 4, 0))
        set_uniform(prog, "exposure", "f", exposure)
        set_uniform(prog, "gamma", "f", gamma)
        set_uniform(prog, "resolution", "2i", (w, h))
        _ck(L.glref_dispatch((w + 15) // 16, (h + 15) // 16, 1))
        out = np.zeros_like(a)
        _ck(L.glref_read_tex2d(tex, 0, 4, out.ctypes.data))
        return out
