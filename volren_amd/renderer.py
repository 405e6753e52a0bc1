"""Python mirror of the reference's Renderer object (src/renderer.h:16-63 as exposed by src/bindings.cpp:117-209),
backed by the HIP library through its C ABI.  Attribute names are the reference's field names."""
import ctypes as C

import numpy as np

from . import _lib

_INT_FIELDS = ("sample", "sppx", "seed", "bounces", "show_environment", "tonemapping", "integrator", "grid_frame_counter",
               "sample_pool_mb", "gpu_encoder", "fast_math", "tf_float_atlas", "launch_target_ms", "order_tiles", "coalesce_trace", "majorant_layout")
_FLOAT_FIELDS = {"tonemap_exposure": 1, "tonemap_gamma": 1, "albedo": 3, "phase": 1, "density_scale": 1,
                 "emission_scale": 1, "vol_clip_min": 3, "vol_clip_max": 3, "env_strength": 1, "env_transform": 9,
                 "tf_window_left": 1, "tf_window_width": 1, "cam_pos": 3, "cam_dir": 3, "cam_up": 3, "cam_fov": 1,
                 "volume_transform": 16}


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


class Renderer:
    """`Renderer(w, h)`; set fields; `load_volume/load_envmap/load_transferfunc`; `render(spp)`; `fbo_data()`."""

    def __init__(self, width, height, device=0, _borrowed=None):
        object.__setattr__(self, "_h", None)
        L = _lib.load()
        object.__setattr__(self, "_owned", _borrowed is None)
        if _borrowed is None:
            h = C.c_void_p()
            _lib.check(L.vr_create(C.byref(h), int(device), int(width), int(height)))
        else:
            h = C.c_void_p(_borrowed)                       # a part of a ShardedRenderer: the handle belongs to it
        object.__setattr__(self, "_h", h)
        object.__setattr__(self, "_L", L)
        object.__setattr__(self, "width", int(width))
        object.__setattr__(self, "height", int(height))

    def close(self):
        if self._h is not None:
            if self._owned:
                self._L.vr_destroy(self._h)
            object.__setattr__(self, "_h", None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- fields ----
    def __getattr__(self, name):
        if name in _INT_FIELDS or name in ("n_grid_frames", "last_launches", "pending_samples", "majorant_blocked", "env_div_safe", "env_compact", "kernel_variant", "kernel_variant_reason"):
            v = C.c_int()
            _lib.check(self._L.vr_get_int(self._h, name.encode(), C.byref(v)))
            return bool(v.value) if name in ("show_environment", "tonemapping") else v.value
        if name in _FLOAT_FIELDS:
            n = _FLOAT_FIELDS[name]
            buf = (C.c_float * n)()
            _lib.check(self._L.vr_get_float(self._h, name.encode(), buf, n))
            return buf[0] if n == 1 else np.array(buf[:], np.float32)
        raise AttributeError(name)

    def __setattr__(self, name, value):
        if name in _INT_FIELDS:
            _lib.check(self._L.vr_set_int(self._h, name.encode(), int(value)))
        elif name in _FLOAT_FIELDS or name == "env_rot":
            v = _f32(np.atleast_1d(value)).reshape(-1)
            _lib.check(self._L.vr_set_float(self._h, name.encode(), v.ctypes.data_as(C.POINTER(C.c_float)), v.size))
        else:
            object.__setattr__(self, name, value)

    def has_fast_math(self):
        """The library carries the opt-in tolerance-mode kernels (`fast_math = 1`); the default kernels are bit-exact."""
        return True

    # ---- scene ----
    def load_volume(self, path):
        _lib.check(self._L.vr_load_volume(self._h, str(path).encode()))

    def set_volume_path(self, path):
        """`renderer.volume = Volume(path)` of the reference's Python module: replaces the volume only (no commit)."""
        _lib.check(self._L.vr_set_volume_path(self._h, str(path).encode()))

    def volume_aabb(self, name="density"):
        out = (C.c_float * 6)()
        _lib.check(self._L.vr_volume_aabb(self._h, name.encode(), out))
        return np.array(out[:3], np.float32), np.array(out[3:], np.float32)

    def volume_minorant_majorant(self, name="density"):
        out = (C.c_float * 2)()
        _lib.check(self._L.vr_volume_minorant_majorant(self._h, name.encode(), out))
        return float(out[0]), float(out[1])

    def load_envmap(self, path):
        _lib.check(self._L.vr_load_envmap(self._h, str(path).encode()))

    def load_transferfunc(self, path):
        _lib.check(self._L.vr_load_transferfunc(self._h, str(path).encode()))

    def set_volume_dense(self, voxels_zyx, transform=None, name="density", unit_cube=True, commit=True):
        v = _f32(voxels_zyx)
        nz, ny, nx = v.shape
        t = _f32(transform).reshape(16) if transform is not None else None
        _lib.check(self._L.vr_set_volume_dense(self._h, name.encode(), v.ctypes.data, nx, ny, nz,
                                               t.ctypes.data if t is not None else None, 1 if unit_cube else 0))
        if commit:
            self.commit()

    def volume_add_grid_frame(self, voxels_zyx, name="density", transform=None):
        """voldata::Volume::add_grid_frame: one more animation frame holding the dense grid `name` (commit() afterwards)."""
        v = _f32(voxels_zyx)
        nz, ny, nx = v.shape
        t = _f32(transform).reshape(16) if transform is not None else None
        _lib.check(self._L.vr_volume_add_grid_frame_dense(self._h, name.encode(), v.ctypes.data, nx, ny, nz, t.ctypes.data if t is not None else None))

    def volume_update_grid_frame(self, frame, voxels_zyx, name="density", transform=None):
        """voldata::Volume::update_grid_frame: grid `name` of frame `frame` replaced (commit() afterwards)."""
        v = _f32(voxels_zyx)
        nz, ny, nx = v.shape
        t = _f32(transform).reshape(16) if transform is not None else None
        _lib.check(self._L.vr_volume_update_grid_frame_dense(self._h, int(frame), name.encode(), v.ctypes.data, nx, ny, nz, t.ctypes.data if t is not None else None))

    def volume_n_grid_frames(self):
        n = C.c_int()
        _lib.check(self._L.vr_volume_n_grid_frames(self._h, C.byref(n)))
        return n.value

    def set_volume_dense_f16(self, voxels_zyx, transform=None, name="density", unit_cube=True, commit=True):
        """Dense fp16 grid kept dense on the device (2 B/voxel, no brick indirection)."""
        v = np.ascontiguousarray(voxels_zyx, dtype=np.float16)
        nz, ny, nx = v.shape
        t = _f32(transform).reshape(16) if transform is not None else None
        _lib.check(self._L.vr_set_volume_dense_f16(self._h, name.encode(), v.ctypes.data, nx, ny, nz,
                                                   t.ctypes.data if t is not None else None, 1 if unit_cube else 0))
        if commit:
            self.commit()

    def set_volume_brick(self, transform, n_bricks, min_maj, indirection, rng, atlas_dim, atlas, mips=(), name="density",
                         unit_cube=True, commit=True):
        t = _f32(transform).reshape(16)
        nb = np.asarray(n_bricks, np.uint32)
        mm = _f32(min_maj)
        ind = np.ascontiguousarray(indirection, np.uint32)
        rg = np.ascontiguousarray(rng, np.uint32)
        ad = np.asarray(atlas_dim, np.uint32)
        at = np.ascontiguousarray(atlas, np.uint8)
        mip_arrays = [np.ascontiguousarray(a, np.uint32) for _, a in mips]
        mip_ptrs = (C.c_void_p * max(1, len(mips)))(*[a.ctypes.data for a in mip_arrays])
        mip_dims = np.asarray([d for d, _ in mips], np.uint32).reshape(-1, 3) if mips else np.zeros((1, 3), np.uint32)
        _lib.check(self._L.vr_set_volume_brick(self._h, name.encode(), t.ctypes.data, nb.ctypes.data, mm.ctypes.data,
                                               ind.ctypes.data, rg.ctypes.data, ad.ctypes.data, at.ctypes.data,
                                               len(mips), C.cast(mip_ptrs, C.c_void_p), mip_dims.ctypes.data, 1 if unit_cube else 0))
        if commit:
            self.commit()

    def set_envmap(self, rgb_top_first):
        a = _f32(rgb_top_first)
        _lib.check(self._L.vr_set_envmap(self._h, a.ctypes.data, a.shape[1], a.shape[0]))

    def set_transferfunc(self, lut):
        if lut is None:
            _lib.check(self._L.vr_set_transferfunc(self._h, None, 0))
            return
        a = _f32(lut)
        _lib.check(self._L.vr_set_transferfunc(self._h, a.ctypes.data, a.shape[0]))

    def commit(self):
        _lib.check(self._L.vr_commit(self._h))

    def reset(self):
        _lib.check(self._L.vr_reset(self._h))

    def resize(self, w, h):
        _lib.check(self._L.vr_resize(self._h, int(w), int(h)))
        object.__setattr__(self, "width", int(w))
        object.__setattr__(self, "height", int(h))

    def scale_and_move_to_unit_cube(self):
        _lib.check(self._L.vr_scale_and_move_to_unit_cube(self._h))

    # ---- rendering ----
    def trace(self):
        """One more sample (RendererOpenGL::trace).  Consecutive calls on an unchanged scene are launched together (include/volren_amd.h)."""
        _lib.check(self._L.vr_trace(self._h))

    def flush(self):
        """Launch the samples recorded by trace() without waiting for them."""
        _lib.check(self._L.vr_flush(self._h))

    def render(self, spp=0, sync=True):
        """`spp` more samples per pixel in one fused launch (the reference loops trace(); bindings.cpp:124-132)."""
        _lib.check(self._L.vr_render(self._h, int(spp)))
        if sync:
            self.synchronize()

    def synchronize(self):
        _lib.check(self._L.vr_synchronize(self._h))

    def last_kernel_ms(self):
        ms = C.c_double()
        _lib.check(self._L.vr_last_kernel_ms(self._h, C.byref(ms)))
        return ms.value

    def last_pathtrace_ms(self):
        """HIP-event duration of the path-tracing kernel alone (last sub-launch; no accumulation pass)."""
        ms = C.c_double()
        _lib.check(self._L.vr_last_pathtrace_ms(self._h, C.byref(ms)))
        return ms.value

    def framebuffer(self):
        """RGBA32F [H][W][4], row 0 = bottom (GL order)."""
        out = np.empty((self.height, self.width, 4), np.float32)
        _lib.check(self._L.vr_framebuffer(self._h, out.ctypes.data))
        return out

    def fbo_data(self):
        """float RGB of the framebuffer (bindings.cpp:141-148)."""
        return self.framebuffer()[..., :3].copy()

    def framebuffer_device_ptr(self):
        p = C.c_void_p()
        _lib.check(self._L.vr_framebuffer_device(self._h, C.byref(p)))
        return p.value

    def draw(self):
        _lib.check(self._L.vr_draw(self._h))

    def display(self):
        out = np.empty((self.height, self.width, 4), np.float32)
        _lib.check(self._L.vr_display(self._h, out.ctypes.data))
        return out

    def save(self, path):
        _lib.check(self._L.vr_save_png(self._h, str(path).encode()))

    # ---- additions ----
    def set_tiles(self, tile_ids):
        t = np.ascontiguousarray(tile_ids, np.int32)
        _lib.check(self._L.vr_set_tiles(self._h, t.ctypes.data if t.size else None, int(t.size)))

    def set_stream(self, stream_handle):
        _lib.check(self._L.vr_set_stream(self._h, C.c_void_p(stream_handle)))

    def pack_tiles(self, tile_ids_dev_ptr, n, packed_dev_ptr):
        _lib.check(self._L.vr_pack_tiles(self._h, C.c_void_p(tile_ids_dev_ptr), int(n), C.c_void_p(packed_dev_ptr)))

    def unpack_tiles(self, tile_ids_dev_ptr, n, packed_dev_ptr):
        _lib.check(self._L.vr_unpack_tiles(self._h, C.c_void_p(tile_ids_dev_ptr), int(n), C.c_void_p(packed_dev_ptr)))

    def uniforms_bytes(self):
        n = self._L.vr_uniforms_size()
        buf = (C.c_uint8 * n)()
        _lib.check(self._L.vr_get_uniforms(self._h, buf, n))
        return bytes(buf)

    def set_sched(self, thresholds):
        """Scheduler thresholds of THIS renderer's launches (8 ints: vr::PathtraceTuning::thr, csrc/vr_device.h)."""
        t = np.ascontiguousarray(thresholds, np.int32)
        assert t.size == 8
        _lib.check(self._L.vr_set_sched(self._h, t.ctypes.data))

    def sched_stats(self, enable=True, read=False):
        """Scheduler diagnostics of this renderer's path-tracing launches. Returns {state: (executions, active_lanes)} when read."""
        out = np.zeros(32, np.uint64) if read else None
        _lib.check(self._L.vr_sched_stats(self._h, 1 if enable else 0, out.ctypes.data if read else None))
        if not read:
            return None
        d = {n: (int(out[2 * i]), int(out[2 * i + 1])) for i, n in enumerate(STATE_NAMES)}
        d["resumes"], d["parks"] = int(out[14]), int(out[15])      # iterations in which the resume / the park block ran
        d["iterations"] = int(out[16])
        d["waves"] = int(out[17])
        d["cycles"] = {n: int(out[18 + i]) for i, n in enumerate(STATE_NAMES)}     # shader-clock ticks inside each state's block
        d["wave_cycles"] = int(out[25])                                            # summed lifetime of all wavefronts
        d["occupancy"] = {n: int(out[26 + i]) / max(1, int(out[16])) for i, n in enumerate(("marching", "ready", "nee", "postnee", "escape", "free"))}
        return d

    def wave_timeline(self, n_waves=8192):
        """(begin, queue empty, end) per wavefront of the last instrumented launch, in seconds relative to the earliest begin; rows of wavefronts that did not run are dropped."""
        out = np.zeros(3 * n_waves, np.uint64)
        _lib.check(self._L.vr_wave_timeline(self._h, out.ctypes.data, out.size))
        t = out.reshape(-1, 3)
        t = t[t[:, 2] > 0].astype(np.float64)
        t0 = t[:, 0].min() if len(t) else 0.0
        t[:, 1] = np.where(t[:, 1] > 0, t[:, 1], t[:, 2])
        return (t - t0) / 1e8

    def grid_checksums(self):
        out = (C.c_uint64 * 3)()
        _lib.check(self._L.vr_grid_checksums(self._h, out))
        return tuple(int(v) for v in out)

    def impmap(self):
        n = self._L.vr_impmap_floats(self._h)
        out = np.empty(n, np.float32)
        _lib.check(self._L.vr_get_impmap(self._h, out.ctypes.data, n))
        return out


class ShardedRenderer:
    """One frame on several devices in ONE process (include/volren_amd.h vr_sharded_*, csrc/sharded.h): `parts[i]` is an ordinary
    Renderer on `devices[i]` -- replicate the scene by applying the same calls to every part (`each`) -- `render(spp)` lets every
    part render its diagonal share of the 16x16 tiles and gathers them (RCCL all-gather between distinct devices, device-to-device
    copies between logical shards of one device); afterwards `parts[0]` holds the whole frame."""

    def __init__(self, width, height, devices):
        self._h = None
        L = _lib.load()
        devs = (C.c_int * len(devices))(*[int(d) for d in devices])
        h = C.c_void_p()
        _lib.check(L.vr_sharded_create(C.byref(h), devs, len(devices), int(width), int(height)))
        self._h, self._L = h, L
        self.devices = [int(d) for d in devices]
        self.parts = [Renderer(width, height, _borrowed=L.vr_sharded_part(h, i)) for i in range(L.vr_sharded_parts(h))]

    def close(self):
        if self._h is not None:
            for p in self.parts:
                p.close()
            self._L.vr_sharded_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def each(self, fn):
        """fn(part) for every part: how scene calls are replicated."""
        for p in self.parts:
            fn(p)
        return self

    @property
    def transport(self):
        return self._L.vr_sharded_transport(self._h).decode()

    @property
    def collective(self):
        """What the rccl transport runs per frame: "gather" (ncclSend / ncclRecv to part 0) or "allgather" (VR_SHARDED_COLLECTIVE)."""
        return self._L.vr_sharded_collective(self._h).decode()

    def reset(self):
        _lib.check(self._L.vr_sharded_reset(self._h))

    def render(self, spp=0, sync=True):
        _lib.check(self._L.vr_sharded_render(self._h, int(spp)))
        if sync:
            self.synchronize()

    def synchronize(self):
        _lib.check(self._L.vr_sharded_synchronize(self._h))

    def framebuffer(self):
        return self.parts[0].framebuffer()

    def save(self, path):
        self.parts[0].save(path)


def math_probe(fn, a, b=None):
    a = _f32(a).reshape(-1)
    b = _f32(b).reshape(-1) if b is not None else np.zeros_like(a)
    out = np.empty_like(a)
    _lib.check(_lib.load().vr_math_probe(int(fn), a.ctypes.data, b.ctypes.data, out.ctypes.data, a.size))
    return out


STATE_NAMES = ("new", "begin", "march", "collide", "nee", "postnee", "escape")
