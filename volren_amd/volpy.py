"""volpy -- the reference's embedded Python module (src/bindings.cpp:64-209) on top of libvolren_amd.so.

`import volren_amd.volpy as volpy` gives scripts written against the reference (`scripts/datagen_colmap.py`,
`scripts/datagen_denoise.py`) the same classes and members: `Renderer`, `Volume`, `Environment`, `TransferFunction`,
`vec3/vec4`.  Differences, all forced by running without a GL window:
  * the resolution is a constructor argument (`Renderer(w, h)`, default 1024x1024) instead of the GL context's size
    (`-w/-h` of the embedding executable); `Renderer.resolution()` returns it,
  * the camera members are per renderer (the reference exposes the global cppgl camera as class statics),
  * `draw()` tonemaps into an off-screen buffer; `save()` / `save_with_alpha()` write that buffer (they read the window in
    the reference, so call `draw()` first exactly like the reference scripts do: datagen_colmap.py:90-94),
  * `shutdown()` does not `exit(0)`.
"""
import math
import os

import numpy as np

from .renderer import Renderer as _Renderer


def vec3(x=0.0, y=None, z=None):
    if y is None:
        y = z = x
    return np.array([x, y, z], np.float32)


def vec4(x=0.0, y=None, z=None, w=None):
    if y is None:
        y = z = w = x
    return np.array([x, y, z, w], np.float32)


class Volume:
    """voldata::Volume: `Volume(path)` (a .brick file or a folder of frames) or `Volume(w, h, d, data)` (dense float32 or
    uint8 voxels, x fastest)."""

    def __init__(self, *args):
        self.path = None
        self.dense = None
        self.grid_frame_counter = 0
        if len(args) == 1:
            self.path = os.fspath(args[0])
        elif len(args) == 4:
            w, h, d, data = args
            a = np.asarray(data)
            a = a.astype(np.float32) / np.float32(255.0) if a.dtype == np.uint8 else a.astype(np.float32)
            self.dense = np.ascontiguousarray(a).reshape(int(d), int(h), int(w))
        elif args:
            raise TypeError("Volume(), Volume(path) or Volume(w, h, d, data)")

    def __repr__(self):
        return "Volume(%s)" % (self.path if self.path else ("dense %s" % (self.dense.shape,) if self.dense is not None else "empty"))


class Environment:
    def __init__(self, path):
        self.path = os.fspath(path)
        self._owner = None
        self._strength = 1.0

    @property
    def strength(self):
        return self._strength

    @strength.setter
    def strength(self, v):
        self._strength = float(v)
        if self._owner is not None:
            self._owner._r.env_strength = self._strength


class TransferFunction:
    def __init__(self, arg=None):
        self.path = None
        self.lut = None
        self._owner = None
        self._left, self._width = 0.0, 1.0
        if arg is None:
            self.lut = np.stack([np.linspace(0, 1, 8, dtype=np.float32)] * 4, 1)   # deterministic ramp (reference: random)
        elif isinstance(arg, (str, os.PathLike)):
            self.path = os.fspath(arg)
        else:
            self.lut = np.asarray(arg, np.float32).reshape(-1, 4)

    def _apply(self, which, v):
        setattr(self, which, float(v))
        if self._owner is not None:
            setattr(self._owner._r, "tf_window_left" if which == "_left" else "tf_window_width", float(v))

    window_left = property(lambda s: s._left, lambda s, v: s._apply("_left", v))
    window_width = property(lambda s: s._width, lambda s, v: s._apply("_width", v))


_FIELDS = ("sample", "sppx", "bounces", "seed", "tonemap_exposure", "tonemap_gamma", "tonemapping", "show_environment",
           "albedo", "phase", "density_scale", "emission_scale", "vol_clip_min", "vol_clip_max",
           "cam_pos", "cam_dir", "cam_up", "cam_fov")


class Renderer:
    """RendererOpenGL as exposed by src/bindings.cpp:117-209."""

    def __init__(self, width=1024, height=1024, device=0):
        object.__setattr__(self, "_r", _Renderer(width, height, device=device))
        object.__setattr__(self, "_volume", None)
        object.__setattr__(self, "_environment", None)
        object.__setattr__(self, "_transferfunc", None)

    # -- fields ------------------------------------------------------------------------------------------------------
    def __getattr__(self, name):
        if name in _FIELDS:
            return getattr(self._r, name)
        raise AttributeError(name)

    def __setattr__(self, name, value):
        if name in _FIELDS:
            setattr(self._r, name, value)
        elif name in ("volume", "environment", "transferfunc"):
            getattr(self, "_set_" + name)(value)
        else:
            object.__setattr__(self, name, value)

    volume = property(lambda s: s._volume)
    environment = property(lambda s: s._environment)
    transferfunc = property(lambda s: s._transferfunc)

    def _set_volume(self, v):
        # assignment only stores the volume; like the reference the grids reach the device in commit()
        object.__setattr__(self, "_volume", v)

    def _set_environment(self, e):
        object.__setattr__(self, "_environment", e)
        self._r.load_envmap(e.path)
        e._owner = self
        self._r.env_strength = e.strength

    def _set_transferfunc(self, t):
        object.__setattr__(self, "_transferfunc", t)
        if t is None:
            self._r.set_transferfunc(None)
            return
        if t.path:
            show = self._r.show_environment
            self._r.load_transferfunc(t.path)
            self._r.show_environment = show          # only main.cpp's loader hides the environment, not the binding
        else:
            self._r.set_transferfunc(t.lut)
        t._owner = self
        self._r.tf_window_left, self._r.tf_window_width = t.window_left, t.window_width

    # -- methods -----------------------------------------------------------------------------------------------------
    def init(self):
        pass                                          # the HIP renderer is initialised by its constructor

    def scale_and_move_to_unit_cube(self):
        object.__setattr__(self, "_unit_cube", True)

    def commit(self):
        v = self._volume
        if v is None:
            raise RuntimeError("Renderer.commit: no volume")
        unit = bool(getattr(self, "_unit_cube", False))
        ds = self._r.density_scale
        if v.path:
            self._r.load_volume(v.path)               # = Volume(path) + density_scale=1 + unit cube + commit (main.cpp:37-62)
            if not unit:
                raise RuntimeError("volpy on HIP: call scale_and_move_to_unit_cube() before commit() for file volumes")
        else:
            self._r.set_volume_dense(v.dense, unit_cube=unit, commit=True)
        if not unit:
            self._r.density_scale = ds
        self._r.grid_frame_counter = int(v.grid_frame_counter)
        object.__setattr__(self, "_unit_cube", False)

    def trace(self):
        self._r.trace()

    def reset(self):
        self._r.reset()

    def render(self, spp):
        self._r.sample = 0                            # bindings.cpp:126
        self._r.render(int(spp))

    def draw(self):
        self._r.draw()

    def resolution(self):
        return (self._r.width, self._r.height)

    def fbo_data(self):
        """Float RGB of the accumulation buffer with the reference's declared buffer shape (w, h, 3) (bindings.cpp:69-77,143)."""
        rgb = self._r.fbo_data()
        return rgb.reshape(self._r.width, self._r.height, 3)

    def _write(self, filename, channels):
        from PIL import Image
        self._r.draw()
        img = np.floor(np.clip(self._r.display()[::-1], 0, 1) * 255.0 + 0.5).astype(np.uint8)
        Image.fromarray(img[..., :channels] if channels == 3 else img).save(filename)
        print("%s written." % filename)

    def save(self, filename="out.png"):
        self._write(filename, 3)

    def save_with_alpha(self, filename="out.png"):
        self._write(os.path.splitext(filename)[0] + ".png", 4)

    # -- camera / COLMAP helpers (bindings.cpp:186-206) ----------------------------------------------------------------
    def _view(self):
        pos, d, up = (np.asarray(getattr(self._r, k), np.float64) for k in ("cam_pos", "cam_dir", "cam_up"))
        f = d / np.linalg.norm(d)
        s = np.cross(f, up)
        s /= np.linalg.norm(s)
        u = np.cross(s, f)
        m = np.eye(4)
        m[0, :3], m[1, :3], m[2, :3] = s, u, -f
        m[:3, 3] = -m[:3, :3] @ pos
        return m

    view_matrix = property(lambda s: s._view())

    def cam_aspect(self):
        return self._r.width / self._r.height

    def colmap_view_trans(self):
        g = np.diag([1.0, -1.0, -1.0, 1.0])
        return (g @ self._view())[:3, 3].astype(np.float32)

    def colmap_view_rot(self):
        m = (np.diag([1.0, -1.0, -1.0, 1.0]) @ self._view())[:3, :3]
        w = math.sqrt(max(0.0, 1.0 + m[0, 0] + m[1, 1] + m[2, 2])) / 2.0
        x = math.copysign(math.sqrt(max(0.0, 1.0 + m[0, 0] - m[1, 1] - m[2, 2])) / 2.0, m[2, 1] - m[1, 2])
        y = math.copysign(math.sqrt(max(0.0, 1.0 - m[0, 0] + m[1, 1] - m[2, 2])) / 2.0, m[0, 2] - m[2, 0])
        z = math.copysign(math.sqrt(max(0.0, 1.0 - m[0, 0] - m[1, 1] + m[2, 2])) / 2.0, m[1, 0] - m[0, 1])
        q = np.array([w, x, y, z])
        return (q / np.linalg.norm(q)).astype(np.float32)         # (w, x, y, z)

    def colmap_focal_length(self):
        return self._r.height / (2.0 * math.tan(0.5 * math.radians(self._r.cam_fov)))

    @staticmethod
    def shutdown():
        pass
