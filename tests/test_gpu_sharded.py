"""Multi-GPU inside the product (include/volren_amd.h vr_sharded_*, volren_amd/csrc/sharded.h, `volren --gpus N`):
one process, N renderers, the frame's 16x16 tiles dealt diagonally, one gather per frame.

A test box has ONE GPU: the parts are logical shards of device 0 (`devices = [0, 0, 0]`: device-to-device copies stand in for
the collective, everything else is the multi-device code path), and RCCL itself is exercised with a one-rank communicator
(VR_SHARDED_TRANSPORT=rccl: librccl opened at run time, ncclCommInitAll, and the frame's collective on the part's stream -- the gather's
group, which with one rank holds part 0's own copy only, and with VR_SHARDED_COLLECTIVE=allgather a grouped ncclAllGather).
N > 1 PHYSICAL devices is not verified anywhere (no such machine was available)."""
import os
import subprocess

import numpy as np
import pytest

import scenes

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def _sharded(name, w, h, devices):
    import volren_amd
    s = volren_amd.ShardedRenderer(w, h, devices)
    s.each(lambda p: scenes.configure(p, name, False))        # the scene is replicated part by part
    return s


@pytest.mark.parametrize("name,w,h,spp,parts", [("c1", 150, 90, 3, 3), ("c3", 96, 64, 4, 2), ("c2", 70, 52, 5, 5), ("c5:32", 80, 48, 3, 4), ("c4:64", 64, 64, 3, 8)])
def test_logical_shards_of_one_device_equal_the_unsharded_frame(name, w, h, spp, parts):
    ref = scenes.oracle_scene(name, w, h).render(spp)
    s = _sharded(name, w, h, [0] * parts)
    assert s.transport == "copy" and len(s.parts) == parts
    s.render(spp)
    assert np.array_equal(_bits(s.framebuffer()), _bits(ref))
    # frames back to back without a synchronisation in between (the copies of frame k+1 must not overtake the unpack of frame k),
    # and a frame accumulated in two calls
    s.reset(); s.render(spp, sync=False)
    s.reset(); s.render(spp - 1, sync=False); s.render(1)
    assert np.array_equal(_bits(s.framebuffer()), _bits(ref))
    assert all(p.sample == spp for p in s.parts)
    s.close()


def test_one_part_is_the_plain_renderer_and_rccl_carries_one_rank(monkeypatch):
    w, h, spp = 96, 64, 4
    ref = scenes.oracle_scene("c1", w, h).render(spp)
    s = _sharded("c1", w, h, [0])
    assert s.transport == "none"
    s.render(spp)
    assert np.array_equal(_bits(s.framebuffer()), _bits(ref))
    s.close()
    monkeypatch.setenv("VR_SHARDED_TRANSPORT", "rccl")
    for collective in ("gather", "allgather"):        # round 6: ncclSend / ncclRecv to part 0 by default, round 4's ncclAllGather behind the switch
        monkeypatch.setenv("VR_SHARDED_COLLECTIVE", collective)
        s = _sharded("c1", w, h, [0])
        assert s.transport == "rccl" and s.collective == collective      # a one-rank communicator: the collective runs, on the part's stream
        s.render(spp)
        assert np.array_equal(_bits(s.framebuffer()), _bits(ref))
        s.reset(); s.render(spp)
        assert np.array_equal(_bits(s.framebuffer()), _bits(ref))
        s.close()
    monkeypatch.setenv("VR_SHARDED_COLLECTIVE", "scatter")
    import volren_amd
    with pytest.raises(volren_amd.VolrenError):
        volren_amd.ShardedRenderer(w, h, [0])
    monkeypatch.delenv("VR_SHARDED_COLLECTIVE")
    import volren_amd
    with pytest.raises(volren_amd.VolrenError):       # RCCL refuses two ranks on one device: said up front, not found out in ncclCommInitAll
        volren_amd.ShardedRenderer(w, h, [0, 0])
    monkeypatch.setenv("VR_SHARDED_TRANSPORT", "copy")
    with pytest.raises(volren_amd.VolrenError):
        volren_amd.ShardedRenderer(w, h, [0, 99])     # no such device


def test_cli_gpus_flag_writes_the_same_png(tmp_path):
    """`volren ... --render --gpus 3 --devices 0,0,0` = the same PNG bytes as `--gpus 1` and as the plain command."""
    exe = scenes.ROOT + "/volren_amd/volren"
    base = [exe, scenes.SMOKE, scenes.HDR, "-w", "112", "-h", "80", "--render", "--spp", "6", "--bounces", "128", "--albedo", "0.8", "--phase", "0.3",
            "--density", "100", "--env_strength", "3", "--env_rot", "270", "--exposure", "3", "--gamma", "2.0", "--cam_fov", "40"]
    pngs = {}
    for tag, extra in (("plain", []), ("g1", ["--gpus", "1"]), ("g3", ["--gpus", "3", "--devices", "0,0,0"]), ("d2", ["--devices", "0,0"])):
        out = subprocess.run(base + extra + ["--output", tag + ".png"], cwd=tmp_path, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        pngs[tag] = (tmp_path / (tag + "_000000.png")).read_bytes()
        if tag == "g3":
            assert "tile exchange: copy" in out.stdout
    assert pngs["g1"] == pngs["plain"] and pngs["g3"] == pngs["plain"] and pngs["d2"] == pngs["plain"]
    bad = subprocess.run(base + ["--gpus", "2", "--devices", "0"], cwd=tmp_path, capture_output=True, text=True, timeout=300)
    assert bad.returncode != 0 and "disagree" in bad.stderr


def test_rccl_that_cannot_be_opened_is_an_error_message_not_a_crash():
    """the failing dlopen used to end in std::string + nullptr inside call_once.  VR_RCCL_LIBRARY names the library to open;
    a name that does not exist must give a clean error when RCCL is asked for by name, and the copy transport when it is not (two devices)."""
    def run(devices, env):
        code = ("import sys\nsys.path.insert(0, %r)\nimport volren_amd\n"
                "try:\n    s = volren_amd.ShardedRenderer(64, 48, %r)\n    print('TRANSPORT', s.transport)\n"
                "except volren_amd.VolrenError as e:\n    print('ERROR:', e)\n") % (scenes.ROOT, devices)
        out = subprocess.run([os.sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, (out.stdout[-500:], out.stderr[-2000:])
        return out.stdout
    env = dict(os.environ, VR_RCCL_LIBRARY="/nonexistent/librccl-not-here.so", VR_SHARDED_TRANSPORT="rccl")
    text = run([0], env)
    assert "ERROR:" in text and "could not be opened" in text and "librccl-not-here" in text, text
    import volren_amd
    if volren_amd.load().vr_device_count() >= 2:                  # not asked for by name: peer copies take over
        env.pop("VR_SHARDED_TRANSPORT")
        assert "TRANSPORT copy" in run([0, 1], env)


def test_more_parts_than_tile_diagonals():
    """8 parts on a 32x32 frame (2x2 tiles, 3 diagonals) -- five parts own no tile.  They must render nothing (an empty tile list means
    the whole frame to set_tiles) and the frame must still be the unsharded one."""
    w = h = 32
    spp = 3
    ref = scenes.oracle_scene("c1", w, h).render(spp)
    s = _sharded("c1", w, h, [0] * 8)
    s.render(spp)
    assert np.array_equal(_bits(s.framebuffer()), _bits(ref))
    assert all(p.sample == spp for p in s.parts)
    idle = [p for i, p in enumerate(s.parts) if i >= 3]
    assert all(p.last_launches == 0 for p in idle)                # nothing was launched for them
    assert all(not p.framebuffer().any() for p in idle)
    s.close()
