"""CPU: the drop-in boundary.  libvolren_amd.so loads without a GPU, exports every function include/volren_amd.h
declares, and every compute entry point fails loudly (no CPU fallback) when no HIP device is present."""
import ctypes as C
import os
import re

import pytest

import scenes
import volren_amd

HEADER = os.path.join(scenes.ROOT, "include", "volren_amd.h")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vr_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported():
    lib = volren_amd.load()
    names = declared_functions()
    assert len(names) >= 35
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    assert sorted(volren_amd.SYMBOLS) == names, "volren_amd/_lib.py SYMBOLS out of sync with include/volren_amd.h"


def test_header_cites_the_reference():
    text = open(HEADER).read()
    for cite in ("src/renderer.cpp:78-145", "src/bindings.cpp:124-132", "src/main.cpp:37-81", "src/renderer.h:30-62", "src/environment.cpp:11-33"):
        assert cite in text


def test_no_device_fails_loudly():
    lib = volren_amd.load()
    if lib.vr_device_count() > 0:
        pytest.skip("a HIP device is present")
    h = C.c_void_p()
    rc = lib.vr_create(C.byref(h), 0, 64, 64)
    assert rc == 2 and not h.value                      # VR_ERR_NO_DEVICE
    assert b"no HIP device" in lib.vr_last_error()
    with pytest.raises(volren_amd.VolrenError):
        volren_amd.Renderer(32, 32)
    with pytest.raises(volren_amd.VolrenError):
        volren_amd.math_probe(0, [1.0])


def test_null_arguments_are_rejected():
    lib = volren_amd.load()
    assert lib.vr_trace(None) == 3 and b"null renderer" in lib.vr_last_error()      # VR_ERR_ARG
    assert lib.vr_render(None, 4) == 3
    assert lib.vr_create(None, 0, 8, 8) == 3
    lib.vr_destroy(None)                                                           # no-op


def test_product_does_not_reference_the_oracle():
    """The oracle is test infrastructure: nothing under volren_amd/ or include/ may import, include or link it."""
    bad = []
    for root in ("volren_amd", "include"):
        for d, _, files in os.walk(os.path.join(scenes.ROOT, root)):
            for f in files:
                if f.endswith((".so", ".pyc", ".o")):
                    continue
                t = open(os.path.join(d, f), errors="replace").read()
                if re.search(r"(from|import)\s+oracle|#include\s+[\"<].*oracle|liboracle|orc_[a-z_]+\(", t):
                    bad.append(os.path.join(d, f))
    assert not bad, bad


def test_host_encoder_statistics():
    """Dense -> brick encoder of the product (host C++) against the numpy reference encoder: same brick count, range."""
    import numpy as np
    import encoder_ref
    lib = volren_amd.load()
    dens = scenes.synthetic_density(40)
    nb = (C.c_uint32 * 3)()
    cnt = C.c_uint64()
    mm = (C.c_float * 2)()
    assert lib.vr_encode_dense_stats(dens.ctypes.data, 40, 40, 40, nb, C.byref(cnt), mm) == 0
    ref = encoder_ref.encode_arrays(dens)
    assert tuple(nb) == tuple(ref["n_bricks"])
    assert cnt.value == ref["brick_counter"]
    assert np.allclose(list(mm), ref["min_maj"])
