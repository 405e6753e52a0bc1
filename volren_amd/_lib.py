"""ctypes loader for libvolren_amd.so (the C ABI declared in include/volren_amd.h).

The library is the product: hand-written HIP kernels for gfx950 plus the C++ host classes.  There is no
fallback -- if the shared object is missing, or no HIP device is visible when a compute entry point is called,
this module raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VOLREN_AMD_LIB") or os.path.join(_HERE, "libvolren_amd.so")      # override: tuning builds (tests/tools_build_variant.sh)

# every symbol include/volren_amd.h declares (checked by tests/test_capi_symbols.py)
SYMBOLS = [
    "vr_last_error", "vr_version", "vr_device_count", "vr_create", "vr_destroy", "vr_resize",
    "vr_load_volume", "vr_set_volume_path", "vr_volume_aabb", "vr_volume_minorant_majorant", "vr_load_envmap", "vr_load_transferfunc", "vr_set_volume_dense", "vr_set_volume_dense_f16", "vr_set_volume_brick",
    "vr_set_envmap", "vr_set_transferfunc", "vr_set_int", "vr_get_int", "vr_set_float", "vr_get_float",
    "vr_commit", "vr_reset", "vr_scale_and_move_to_unit_cube", "vr_trace", "vr_flush", "vr_render", "vr_synchronize",
    "vr_last_kernel_ms", "vr_last_pathtrace_ms", "vr_framebuffer", "vr_framebuffer_device", "vr_draw", "vr_display", "vr_save_png",
    "vr_set_tiles", "vr_set_stream", "vr_pack_tiles", "vr_unpack_tiles", "vr_get_uniforms", "vr_uniforms_size",
    "vr_volume_add_grid_frame_dense", "vr_volume_update_grid_frame_dense", "vr_volume_n_grid_frames", "vr_impmap_floats", "vr_get_impmap", "vr_test_alloc_cap_mb", "vr_set_sched", "vr_sched_stats", "vr_grid_checksums", "vr_math_probe", "vr_encode_dense_stats", "vr_write_brick_from_dense", "vr_write_dense",
    "vr_sharded_create", "vr_sharded_destroy", "vr_sharded_parts", "vr_sharded_part", "vr_sharded_transport", "vr_sharded_collective", "vr_sharded_reset", "vr_sharded_render", "vr_sharded_synchronize", "vr_tile_owners", "vr_wave_timeline",
]

_lib = None


class VolrenError(RuntimeError):
    pass


def load():
    """Load the library (torch first, so that both share one libamdhip64 when torch is used in the process)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise VolrenError("libvolren_amd.so is not built (%s). Run `make` or __graft_entry__.build(); there is no "
                          "CPU/PyTorch fallback for the HIP path." % LIB_PATH)
    try:  # noqa: SIM105 -- torch is optional plumbing (streams, torch.distributed); the renderer itself does not need it
        import torch  # noqa: F401
    except Exception:
        pass
    L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    vp, ci, cf, cd = C.c_void_p, C.c_int, C.c_float, C.c_double
    L.vr_last_error.restype = C.c_char_p
    L.vr_version.restype = C.c_char_p
    L.vr_device_count.restype = ci
    L.vr_create.argtypes = [C.POINTER(vp), ci, ci, ci]
    L.vr_destroy.argtypes = [vp]
    L.vr_destroy.restype = None
    L.vr_resize.argtypes = [vp, ci, ci]
    L.vr_volume_aabb.argtypes = [vp, C.c_char_p, C.POINTER(cf)]
    L.vr_volume_minorant_majorant.argtypes = [vp, C.c_char_p, C.POINTER(cf)]
    for n in ("vr_load_volume", "vr_set_volume_path", "vr_load_envmap", "vr_load_transferfunc", "vr_save_png"):
        getattr(L, n).argtypes = [vp, C.c_char_p]
    L.vr_set_volume_dense.argtypes = [vp, C.c_char_p, vp, ci, ci, ci, vp, ci]
    L.vr_set_volume_dense_f16.argtypes = [vp, C.c_char_p, vp, ci, ci, ci, vp, ci]
    L.vr_set_volume_brick.argtypes = [vp, C.c_char_p, vp, vp, vp, vp, vp, vp, vp, ci, vp, vp, ci]
    L.vr_set_envmap.argtypes = [vp, vp, ci, ci]
    L.vr_set_transferfunc.argtypes = [vp, vp, ci]
    L.vr_set_int.argtypes = [vp, C.c_char_p, ci]
    L.vr_get_int.argtypes = [vp, C.c_char_p, C.POINTER(ci)]
    L.vr_set_float.argtypes = [vp, C.c_char_p, C.POINTER(cf), ci]
    L.vr_get_float.argtypes = [vp, C.c_char_p, C.POINTER(cf), ci]
    for n in ("vr_commit", "vr_reset", "vr_scale_and_move_to_unit_cube", "vr_trace", "vr_synchronize", "vr_draw"):
        getattr(L, n).argtypes = [vp]
    L.vr_render.argtypes = [vp, ci]
    L.vr_last_kernel_ms.argtypes = [vp, C.POINTER(cd)]
    L.vr_last_pathtrace_ms.argtypes = [vp, C.POINTER(cd)]
    L.vr_framebuffer.argtypes = [vp, vp]
    L.vr_framebuffer_device.argtypes = [vp, C.POINTER(vp)]
    L.vr_display.argtypes = [vp, vp]
    L.vr_set_tiles.argtypes = [vp, vp, ci]
    L.vr_set_stream.argtypes = [vp, vp]
    L.vr_pack_tiles.argtypes = [vp, vp, ci, vp]
    L.vr_unpack_tiles.argtypes = [vp, vp, ci, vp]
    L.vr_get_uniforms.argtypes = [vp, vp, ci]
    L.vr_uniforms_size.restype = ci
    L.vr_impmap_floats.argtypes = [vp]
    L.vr_get_impmap.argtypes = [vp, vp, ci]
    L.vr_volume_add_grid_frame_dense.argtypes = [vp, C.c_char_p, vp, ci, ci, ci, vp]
    L.vr_volume_update_grid_frame_dense.argtypes = [vp, ci, C.c_char_p, vp, ci, ci, ci, vp]
    L.vr_volume_n_grid_frames.argtypes = [vp, vp]
    L.vr_test_alloc_cap_mb.argtypes = [C.c_longlong]
    L.vr_set_sched.argtypes = [vp, vp]
    L.vr_sched_stats.argtypes = [vp, ci, vp]
    L.vr_grid_checksums.argtypes = [vp, vp]
    L.vr_math_probe.argtypes = [ci, vp, vp, vp, ci]
    L.vr_write_brick_from_dense.argtypes = [vp, ci, ci, ci, vp, C.c_char_p]
    L.vr_write_dense.argtypes = [vp, ci, ci, ci, cf, cf, vp, C.c_char_p]
    L.vr_encode_dense_stats.argtypes = [vp, ci, ci, ci, vp, vp, vp]
    L.vr_sharded_create.argtypes = [C.POINTER(vp), C.POINTER(ci), ci, ci, ci]
    L.vr_sharded_destroy.argtypes = [vp]
    L.vr_sharded_destroy.restype = None
    L.vr_sharded_parts.argtypes = [vp]
    L.vr_sharded_part.argtypes = [vp, ci]
    L.vr_sharded_part.restype = vp
    L.vr_sharded_transport.argtypes = [vp]
    L.vr_sharded_transport.restype = C.c_char_p
    L.vr_sharded_collective.argtypes = [vp]
    L.vr_sharded_collective.restype = C.c_char_p
    L.vr_sharded_reset.argtypes = [vp]
    L.vr_sharded_render.argtypes = [vp, ci]
    L.vr_sharded_synchronize.argtypes = [vp]
    L.vr_tile_owners.argtypes = [ci, ci, ci, vp, ci]
    L.vr_wave_timeline.argtypes = [vp, vp, ci]
    _lib = L
    return L


def check(rc):
    if rc != 0:
        raise VolrenError(load().vr_last_error().decode("utf-8", "replace") or ("error code %d" % rc))
