"""volpy -- the reference's embedded Python module (src/bindings.cpp:64-417) on top of libvolren_amd.so.

`import volren_amd.volpy as volpy` gives scripts written against the reference (`scripts/datagen_colmap.py`,
`scripts/datagen_denoise.py`) the same classes, members and CALL PROTOCOL: assigning `renderer.volume` only replaces the
volume; `scale_and_move_to_unit_cube()` multiplies the density scale the caller has set (renderer.cpp:227-242); `commit()`
uploads the grids.  `vec2/vec3/vec4/quat` are small glm-like value types (`.x .y .z`, arithmetic, `.length()`,
`.normalize()`, buffer access through `np.array(v)`; also `ivec2/3/4`, `uvec2/3/4`, column-major `mat3` / `mat4` with `column(i)`, `value(i, j)`), `Volume.AABB(name)` returns two of them, `resolution()` has `.x/.y`.
Differences, all forced by running without a GL window:
  * the resolution is a constructor argument (`Renderer(w, h)`, default 1024x1024; the reference takes the GL context's,
    i.e. `-w/-h` of the embedding executable),
  * the camera members are per renderer (the reference exposes the global cppgl camera as class statics),
  * `draw()` tonemaps into an off-screen buffer; `save()` / `save_with_alpha()` write that buffer (they read the window in
    the reference, so call `draw()` first exactly like the reference scripts do: datagen_colmap.py:90-94),
  * `shutdown()` does not `exit(0)`,
  * `TransferFunction.randomize()` draws from Python's `random` (the reference from C `rand()`): seedable from the script.
"""
import math
import os
import random as _random

import numpy as np

from .renderer import Renderer as _Renderer


# ---- glm-like value types (bindings.cpp:13-62, 215-416) ------------------------------------------------------------------
class _Vec:
    """float32 vector with glm's operator set: component-wise + - * / with a vector of the same size or a scalar."""
    __slots__ = ("v",)
    _n = 0
    _dtype = np.float32

    def __init__(self, *args):
        n = self._n
        if len(args) == 0:
            self.v = np.zeros(n, self._dtype)
        elif len(args) == 1 and np.ndim(args[0]) == 0 and not isinstance(args[0], _Vec):
            self.v = np.full(n, args[0], self._dtype)
        elif len(args) == 1:
            a = np.asarray(args[0].v if isinstance(args[0], _Vec) else args[0], self._dtype).reshape(-1)
            if a.size != n:
                raise TypeError("%s from %d components" % (type(self).__name__, a.size))
            self.v = a.copy()
        elif len(args) == n:
            self.v = np.array(args, self._dtype)
        else:
            raise TypeError("%s(): %d arguments" % (type(self).__name__, len(args)))

    def _other(self, o):
        if isinstance(o, _Vec):
            if o._n != self._n:
                raise TypeError("vector sizes differ")
            return o.v
        if np.ndim(o) == 0:
            return self._dtype(o)
        a = np.asarray(o, self._dtype).reshape(-1)
        if a.size != self._n:
            raise TypeError("vector sizes differ")
        return a

    def _new(self, a):
        r = type(self).__new__(type(self))
        r.v = np.asarray(a, self._dtype)
        return r

    def __add__(self, o): return self._new(self.v + self._other(o))
    def __radd__(self, o): return self._new(self._other(o) + self.v)
    def __sub__(self, o): return self._new(self.v - self._other(o))
    def __rsub__(self, o): return self._new(self._other(o) - self.v)
    def __mul__(self, o): return self._new(self.v * self._other(o))
    def __rmul__(self, o): return self._new(self._other(o) * self.v)
    def __truediv__(self, o): return self._new(self.v / self._other(o))
    def __rtruediv__(self, o): return self._new(self._other(o) / self.v)
    def __neg__(self): return self._new(-self.v)

    def __iadd__(self, o): self.v = (self.v + self._other(o)).astype(self._dtype); return self
    def __isub__(self, o): self.v = (self.v - self._other(o)).astype(self._dtype); return self
    def __imul__(self, o): self.v = (self.v * self._other(o)).astype(self._dtype); return self
    def __itruediv__(self, o): self.v = (self.v / self._other(o)).astype(self._dtype); return self

    def length(self):
        return float(np.sqrt(np.float32((self.v * self.v).sum(dtype=np.float32))))

    def normalize(self):
        return self._new(self.v / np.float32(self.length()))

    def __array__(self, dtype=None, copy=None):
        return self.v.astype(dtype) if dtype is not None else self.v.copy()

    def __len__(self): return self._n
    def __iter__(self): return iter(self.v.tolist())
    def __getitem__(self, i): return self.v[i].item() if np.ndim(self.v[i]) == 0 else self.v[i]
    def __setitem__(self, i, x): self.v[i] = x
    def __eq__(self, o): return isinstance(o, _Vec) and o._n == self._n and bool(np.array_equal(self.v, o.v))
    def __hash__(self): return hash(self.v.tobytes())
    def __repr__(self): return "%s(%s)" % (type(self).__name__, ", ".join(("%d" if np.issubdtype(self._dtype, np.integer) else "%f") % c for c in self.v))


def _component(i):
    return property(lambda s: s.v[i].item(), lambda s, x: s.v.__setitem__(i, x))


class vec2(_Vec):
    __slots__ = ()
    _n = 2
    x, y = _component(0), _component(1)


class vec3(_Vec):
    __slots__ = ()
    _n = 3
    x, y, z = _component(0), _component(1), _component(2)


class vec4(_Vec):
    __slots__ = ()
    _n = 4
    x, y, z, w = _component(0), _component(1), _component(2), _component(3)


class ivec2(_Vec):
    __slots__ = ()
    _n = 2
    _dtype = np.int32
    x, y = _component(0), _component(1)


class ivec3(_Vec):
    __slots__ = ()
    _n = 3
    _dtype = np.int32
    x, y, z = _component(0), _component(1), _component(2)


class ivec4(_Vec):
    __slots__ = ()
    _n = 4
    _dtype = np.int32
    x, y, z, w = _component(0), _component(1), _component(2), _component(3)


class uvec2(_Vec):
    __slots__ = ()
    _n = 2
    _dtype = np.uint32
    x, y = _component(0), _component(1)


class uvec3(_Vec):
    __slots__ = ()
    _n = 3
    _dtype = np.uint32
    x, y, z = _component(0), _component(1), _component(2)


class uvec4(_Vec):
    __slots__ = ()
    _n = 4
    _dtype = np.uint32
    x, y, z, w = _component(0), _component(1), _component(2), _component(3)


class _Mat:
    """glm::mat3 / glm::mat4 as bound by bindings.cpp:348-395: column-major, `column(i)`, `value(i, j)` = m[i][j] (column i, row j),
    + - * with a matrix, * with a scalar, unary minus; `np.array(m)` has the bound buffer's shape (n, n) with row i = COLUMN i (glm memory)."""
    __slots__ = ("m",)
    _n = 0
    _vec = None

    def __init__(self, *args):
        n = self._n
        if len(args) == 0:
            self.m = np.eye(n, dtype=np.float32)                  # glm's default constructor: identity
        elif len(args) == 1 and np.ndim(args[0]) == 0:
            self.m = np.eye(n, dtype=np.float32) * np.float32(args[0])
        elif len(args) == n:
            self.m = np.stack([np.asarray(c, np.float32).reshape(n) for c in args]).copy()      # the columns
        elif len(args) == 1:
            a = np.asarray(args[0].m if isinstance(args[0], _Mat) else args[0], np.float32)
            if a.shape != (n, n):
                raise TypeError("%s from shape %s" % (type(self).__name__, a.shape))
            self.m = a.copy()
        else:
            raise TypeError("%s(): %d arguments" % (type(self).__name__, len(args)))

    def _new(self, a):
        r = type(self).__new__(type(self))
        r.m = np.asarray(a, np.float32)
        return r

    def column(self, i): return self._vec(self.m[int(i)])
    def value(self, i, j): return self.m[int(i), int(j)].item()
    def __add__(self, o): return self._new(self.m + o.m)
    def __sub__(self, o): return self._new(self.m - o.m)
    def __neg__(self): return self._new(-self.m)

    def __mul__(self, o):
        if isinstance(o, _Mat):
            return self._new((self.m.T @ o.m.T).T)                # rows of .m are columns: (A B) stored column-major
        if isinstance(o, _Vec):
            return self._vec((self.m.T @ o.v).astype(np.float32))
        return self._new(self.m * np.float32(o))

    def __rmul__(self, o): return self._new(self.m * np.float32(o))
    def __iadd__(self, o): self.m = self.m + o.m; return self
    def __isub__(self, o): self.m = self.m - o.m; return self
    def __imul__(self, o): self.m = (self * o).m; return self
    def __array__(self, dtype=None, copy=None): return self.m.astype(dtype) if dtype is not None else self.m.copy()
    def __eq__(self, o): return isinstance(o, _Mat) and o._n == self._n and bool(np.array_equal(self.m, o.m))
    def __hash__(self): return hash(self.m.tobytes())
    def __repr__(self): return "%s(%s)" % (type(self).__name__, ", ".join("(%s)" % ", ".join("%f" % c for c in col) for col in self.m))


class mat3(_Mat):
    __slots__ = ()
    _n = 3
    _vec = vec3


class mat4(_Mat):
    __slots__ = ()
    _n = 4
    _vec = vec4


class quat(_Vec):
    """glm::quat; memory order (x, y, z, w) like glm's default, so `np.array(q)[[3, 0, 1, 2]]` is (w, x, y, z)
    (datagen_colmap.py:94).  Constructor order (w, x, y, z) as in glm."""
    __slots__ = ()
    _n = 4
    x, y, z, w = _component(0), _component(1), _component(2), _component(3)

    def __init__(self, *args):
        if len(args) == 4:
            w, x, y, z = args
            _Vec.__init__(self, x, y, z, w)
        elif len(args) == 0:
            _Vec.__init__(self, 0.0, 0.0, 0.0, 1.0)
        else:
            _Vec.__init__(self, *args)


# ---- scene objects -------------------------------------------------------------------------------------------------------
class Volume:
    """voldata::Volume as bound by bindings.cpp:82-95: `Volume()`, `Volume(path)` (a .brick file or a folder of frames) or
    `Volume(w, h, d, data)` (dense float32 or uint8 voxels, x fastest)."""

    def __init__(self, *args):
        self.path = None
        self.dense = None
        self._owner = None
        self._frame = 0
        self._edits = []                 # add_grid_frame / update_grid_frame calls, replayed whenever the volume is (re)attached to a renderer
        if len(args) == 1:
            self.path = os.fspath(args[0])
        elif len(args) == 4:
            w, h, d, data = args
            a = np.asarray(data)
            a = a.astype(np.float32) / np.float32(255.0) if a.dtype == np.uint8 else a.astype(np.float32)
            self.dense = np.ascontiguousarray(a).reshape(int(d), int(h), int(w))
        elif args:
            raise TypeError("Volume(), Volume(path) or Volume(w, h, d, data)")

    def _need_owner(self, what):
        if self._owner is None:
            raise RuntimeError("Volume.%s: assign the volume to a Renderer first (the grids live in the HIP library)" % what)
        return self._owner._r

    def AABB(self, name="density"):
        """World-space bounding box (bb_min, bb_max) of the current frame's grid `name` under the volume transform."""
        lo, hi = self._need_owner("AABB").volume_aabb(name)
        return vec3(lo), vec3(hi)

    def minorant_majorant(self, name="density"):
        return self._need_owner("minorant_majorant").volume_minorant_majorant(name)

    def clear(self):
        self.path, self.dense = None, None
        self._edits = []

    @staticmethod
    def _dense_of(grid):
        """A grid argument: a dense Volume(w, h, d, data) or an array [z][y][x] (the reference's Python has no Grid class of its own)."""
        a = grid.dense if isinstance(grid, Volume) else np.asarray(grid, np.float32)
        if a is None or a.ndim != 3:
            raise TypeError("grid: a dense Volume(w, h, d, data) or a float array [z][y][x]")
        return np.ascontiguousarray(a, np.float32)

    def _record(self, edit):
        """Keep the edit for a later (re)assignment of this volume; an update supersedes earlier updates of the same (frame, name)."""
        if edit[0] == "update":
            self._edits = [e for e in self._edits if not (e[0] == "update" and e[1] == edit[1] and e[3] == edit[3])]
        self._edits.append(edit)

    def add_grid_frame(self, grid, name="density"):
        """voldata::Volume::add_grid_frame (bindings.cpp:89): a further animation frame that holds `grid` as `name`.  On a volume that a
        renderer holds the edit goes to THAT volume object in place, as in the reference (its transform -- scale_and_move_to_unit_cube --
        and frame counter stay); it takes effect at the next commit()."""
        dense = self._dense_of(grid)
        self._record(("add", None, dense, name))
        if self._owner is not None:
            if len(self._edits) == 1 and not self.path and self.dense is None:
                self._owner._attach_volume(self)          # Volume() + the first add_grid_frame: this creates the renderer's volume
            else:
                self._owner._r.volume_add_grid_frame(dense, name)
                object.__setattr__(self._owner, "_committed", False)

    def update_grid_frame(self, i, grid, name="density"):
        """voldata::Volume::update_grid_frame (bindings.cpp:90): grid `name` of frame `i` replaced (or added to that frame), in place."""
        dense = self._dense_of(grid)
        self._record(("update", int(i), dense, name))
        if self._owner is not None:
            self._owner._r.volume_update_grid_frame(int(i), dense, name)
            object.__setattr__(self._owner, "_committed", False)

    def n_grid_frames(self):
        return self._need_owner("n_grid_frames").volume_n_grid_frames()

    def load_grid(self, path):
        """Replaces the volume by the grid file `path` (takes effect at the next assignment / commit)."""
        self.path, self.dense = os.fspath(path), None
        if self._owner is not None:
            self._owner._attach_volume(self)

    @property
    def grid_frame_counter(self):
        return self._frame

    @grid_frame_counter.setter
    def grid_frame_counter(self, v):
        self._frame = int(v)
        if self._owner is not None and self._owner._committed:
            self._owner._r.grid_frame_counter = self._frame

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
    """TransferFunction(), (path) or (list of vec4) -- transferfunc.h:9-42 as bound by bindings.cpp:104-112."""

    def __init__(self, arg=None):
        self.path = None
        self.lut = None
        self._owner = None
        self._left, self._width = 0.0, 1.0
        if arg is None:
            self.randomize()                              # the reference's default constructor randomizes too (transferfunc.cpp:7-9)
        elif isinstance(arg, (str, os.PathLike)):
            self.path = os.fspath(arg)
        else:
            self.lut = np.asarray([np.asarray(v, np.float32) for v in arg], np.float32).reshape(-1, 4)

    def randomize(self, n_bins=8):
        """transferfunc.cpp:62-67: bin 0 is all zero, the others uniform random RGBA."""
        lut = np.zeros((int(n_bins), 4), np.float32)
        for i in range(1, int(n_bins)):
            lut[i] = [_random.random() for _ in range(4)]
        self.lut, self.path = lut, None
        if self._owner is not None:
            self._owner._upload_transferfunc(self)

    def _apply(self, which, v):
        setattr(self, which, float(v))
        if self._owner is not None:
            setattr(self._owner._r, "tf_window_left" if which == "_left" else "tf_window_width", float(v))

    window_left = property(lambda s: s._left, lambda s, v: s._apply("_left", v))
    window_width = property(lambda s: s._width, lambda s, v: s._apply("_width", v))


# what stands in for the reference's GL context: the resolution (Context::resolution()) and the device renderers are created on
_CONTEXT = {"width": 1024, "height": 1024, "device": 0}


def set_context(width=None, height=None, device=None):
    """The -w / -h (and --device) of the command line that started the script (src/main.cpp:311-357): the size of every `Renderer()` created
    without explicit arguments afterwards."""
    for k, v in (("width", width), ("height", height), ("device", device)):
        if v is not None:
            if int(v) < (0 if k == "device" else 1):
                raise ValueError("%s must be positive" % k)
            _CONTEXT[k] = int(v)


_SCALARS = ("sample", "sppx", "bounces", "seed", "tonemap_exposure", "tonemap_gamma", "tonemapping", "show_environment",
            "phase", "density_scale", "emission_scale", "cam_fov")
_VEC3S = ("albedo", "vol_clip_min", "vol_clip_max", "cam_pos", "cam_dir", "cam_up")


class Renderer:
    """RendererOpenGL as exposed by src/bindings.cpp:117-209."""

    def __init__(self, width=None, height=None, device=None):
        # `volpy.Renderer()` of the reference's scripts takes its size from the GL context the executable created from -w / -h
        # (src/main.cpp:311-357, src/renderer.cpp:47); here from set_context(), which volren_amd.run_script calls with the same flags
        width = _CONTEXT["width"] if width is None else width
        height = _CONTEXT["height"] if height is None else height
        device = _CONTEXT["device"] if device is None else device
        object.__setattr__(self, "_r", _Renderer(width, height, device=device))
        object.__setattr__(self, "_volume", None)
        object.__setattr__(self, "_environment", None)
        object.__setattr__(self, "_transferfunc", None)
        object.__setattr__(self, "_committed", False)

    # -- fields ------------------------------------------------------------------------------------------------------
    def __getattr__(self, name):
        if name in _SCALARS:
            return getattr(self._r, name)
        if name in _VEC3S:
            return vec3(getattr(self._r, name))
        raise AttributeError(name)

    def __setattr__(self, name, value):
        if name in _SCALARS:
            setattr(self._r, name, value)
        elif name in _VEC3S:
            setattr(self._r, name, np.asarray(vec3(value) if np.ndim(value) == 0 else value, np.float32).reshape(3))
        elif name in ("volume", "environment", "transferfunc"):
            getattr(self, "_set_" + name)(value)
        else:
            object.__setattr__(self, name, value)

    volume = property(lambda s: s._volume)
    environment = property(lambda s: s._environment)
    transferfunc = property(lambda s: s._transferfunc)

    def _attach_volume(self, v):
        # like `renderer->volume = ...` in the reference: replaces the volume object only.  density_scale, the unit-cube
        # transform and the device grids change in scale_and_move_to_unit_cube() / commit(), in the caller's order.
        edits = list(v._edits)
        if v.path:
            self._r.set_volume_path(v.path)
        elif v.dense is not None:
            self._r.set_volume_dense(v.dense, unit_cube=False, commit=False)
        elif edits and edits[0][0] == "add" and edits[0][3] == "density":
            self._r.set_volume_dense(edits.pop(0)[2], unit_cube=False, commit=False)      # Volume() + add_grid_frame(...): the first frame
        else:
            raise RuntimeError("Renderer.volume: empty Volume")
        for kind, i, dense, name in edits:
            if kind == "add":
                self._r.volume_add_grid_frame(dense, name)
            else:
                self._r.volume_update_grid_frame(i, dense, name)
        object.__setattr__(self, "_committed", False)

    def _set_volume(self, v):
        object.__setattr__(self, "_volume", v)
        if v is not None:
            v._owner = self
            self._attach_volume(v)

    def _set_environment(self, e):
        object.__setattr__(self, "_environment", e)
        self._r.load_envmap(e.path)
        e._owner = self
        self._r.env_strength = e.strength

    def _upload_transferfunc(self, t):
        if t.path:
            show = self._r.show_environment
            self._r.load_transferfunc(t.path)
            self._r.show_environment = show          # only main.cpp's loader hides the environment, not the binding
        else:
            self._r.set_transferfunc(t.lut)
        self._r.tf_window_left, self._r.tf_window_width = t.window_left, t.window_width

    def _set_transferfunc(self, t):
        object.__setattr__(self, "_transferfunc", t)
        if t is None:
            self._r.set_transferfunc(None)
            return
        t._owner = self
        self._upload_transferfunc(t)

    # -- methods -----------------------------------------------------------------------------------------------------
    def init(self):
        pass                                          # the HIP renderer is initialised by its constructor

    def scale_and_move_to_unit_cube(self):
        """renderer.cpp:227-242: volume.transform = scale(1/size) o translate(...), density_scale *= size."""
        if self._volume is None:
            raise RuntimeError("Renderer.scale_and_move_to_unit_cube: no volume")
        self._r.scale_and_move_to_unit_cube()

    def commit(self):
        if self._volume is None:
            raise RuntimeError("Renderer.commit: no volume")
        self._r.commit()
        object.__setattr__(self, "_committed", True)
        self._r.grid_frame_counter = int(self._volume.grid_frame_counter)

    def trace(self):
        self._r.trace()

    def reset(self):
        self._r.reset()

    def render(self, spp):
        self._r.sample = 0                            # bindings.cpp:126
        self._r.render(int(spp))

    def draw(self):
        if self._committed:
            self._r.draw()                            # before the first commit there is nothing to show (the reference clears the window)

    def resolution(self):
        return ivec2(self._r.width, self._r.height)

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

    # cam_near / cam_far / proj_matrix (src/bindings.cpp:190-193): fields of the cppgl camera the renderer never reads (a pinhole through
    # cam_fov, shader/common.glsl:76-80) -- kept so that a script that sets or logs them runs.  Defaults: cppgl's (0.01 / 1000, unverified:
    # cppgl is not vendored); proj_matrix = glm::perspective(radians(fov), aspect, near, far), column-major like every matrix here.
    cam_near = 0.01
    cam_far = 1000.0

    def _proj(self):
        f = 1.0 / math.tan(0.5 * math.radians(self._r.cam_fov))
        n, fa = float(self.cam_near), float(self.cam_far)
        return mat4(vec4(f / self.cam_aspect(), 0, 0, 0), vec4(0, f, 0, 0), vec4(0, 0, -(fa + n) / (fa - n), -1), vec4(0, 0, -2.0 * fa * n / (fa - n), 0))

    proj_matrix = property(lambda s: s._proj())

    def cam_aspect(self):
        return self._r.width / self._r.height

    def colmap_view_trans(self):
        g = np.diag([1.0, -1.0, -1.0, 1.0])
        return vec3((g @ self._view())[:3, 3])

    def colmap_view_rot(self):
        """glm::normalize(glm::toQuat(GL_TO_COLMAP * view)) as a `quat` (memory order x, y, z, w)."""
        m = (np.diag([1.0, -1.0, -1.0, 1.0]) @ self._view())[:3, :3]
        w = math.sqrt(max(0.0, 1.0 + m[0, 0] + m[1, 1] + m[2, 2])) / 2.0
        x = math.copysign(math.sqrt(max(0.0, 1.0 + m[0, 0] - m[1, 1] - m[2, 2])) / 2.0, m[2, 1] - m[1, 2])
        y = math.copysign(math.sqrt(max(0.0, 1.0 - m[0, 0] + m[1, 1] - m[2, 2])) / 2.0, m[0, 2] - m[2, 0])
        z = math.copysign(math.sqrt(max(0.0, 1.0 - m[0, 0] - m[1, 1] + m[2, 2])) / 2.0, m[1, 0] - m[0, 1])
        n = math.sqrt(w * w + x * x + y * y + z * z)
        return quat(w / n, x / n, y / n, z / n)

    def colmap_focal_length(self):
        return self._r.height / (2.0 * math.tan(0.5 * math.radians(self._r.cam_fov)))

    @staticmethod
    def shutdown():
        pass
