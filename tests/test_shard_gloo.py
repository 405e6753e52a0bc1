"""CPU, world_size 2 over gloo: the multi-GPU path of bench.py (tile ownership -> render own tiles -> pack -> ONE
all_gather -> unpack) with the oracle standing in for the HIP renderer and numpy for the pack/unpack kernels.
The sharded frame must be bit-identical to the single-process frame."""
import os
import socket
import sys

import numpy as np
import pytest

import scenes
from volren_amd.shard import TileShard, tile_owner_lists, pack_tiles_numpy, unpack_tiles_numpy

W, H, SPP = 70, 52, 3


def test_tile_ownership_is_a_partition():
    for w, h in ((1024, 1024), (1920, 1080), (70, 52), (16, 16)):
        tx, ty = (w + 15) // 16, (h + 15) // 16
        for n in (1, 2, 4, 8):
            lists = tile_owner_lists(w, h, n)
            flat = sorted(sum(lists, []))
            assert flat == list(range(tx * ty))
            if tx * ty >= 8 * n:
                sizes = [len(t) for t in lists]
                assert max(sizes) - min(sizes) <= max(tx, ty)          # diagonal interleave balances the counts
            s = TileShard(w, h, n, n - 1)
            assert len(s.pack_ids) == s.n_max and len(s.unpack_ids) == n * s.n_max
            assert sorted(t for t in s.unpack_ids if t >= 0) == flat


def test_product_tile_deal_equals_the_python_deal():
    """The sharded renderer inside the library (volren_amd/csrc/sharded.cpp) and the torch.distributed path (volren_amd/shard.py) deal the tiles
    identically: vr_tile_owners (host only) against tile_owner_lists, ragged frames and part counts up to 8 and beyond."""
    import volren_amd
    lib = volren_amd.load()
    for w, h in ((1024, 1024), (1920, 1080), (2048, 2048), (70, 52), (16, 16), (17, 1)):
        n_tiles = ((w + 15) // 16) * ((h + 15) // 16)
        for n in (1, 2, 3, 5, 8, 13):
            owner = np.full(n_tiles, -1, np.int32)
            assert lib.vr_tile_owners(w, h, n, owner.ctypes.data, n_tiles) == 0
            want = np.full(n_tiles, -1, np.int32)
            for p, tiles in enumerate(tile_owner_lists(w, h, n)):
                want[tiles] = p
            assert np.array_equal(owner, want), (w, h, n)
    assert lib.vr_tile_owners(64, 64, 2, None, 16) == 3 and lib.vr_tile_owners(64, 64, 2, np.zeros(15, np.int32).ctypes.data, 15) == 3      # VR_ERR_ARG


def test_hashed_deal_is_a_balanced_partition():
    """The alternative tile deal (tiles in the order of a hash of their id, dealt round robin): a partition with counts within one.
    Measured load balance (profiles/r3_rank_balance.txt): diagonal max/mean <= 1.021, hashed 1.05-1.06 -- the diagonal deal is the default."""
    for w, h in ((1024, 1024), (1920, 1080), (70, 52)):
        tx, ty = (w + 15) // 16, (h + 15) // 16
        for n in (2, 4, 8):
            lists = tile_owner_lists(w, h, n, "hashed")
            assert sorted(sum(lists, [])) == list(range(tx * ty))
            sizes = [len(t) for t in lists]
            assert max(sizes) - min(sizes) <= 1
            s = TileShard(w, h, n, 0, scheme="hashed")
            assert sorted(t for t in s.unpack_ids if t >= 0) == list(range(tx * ty))


def test_pack_unpack_roundtrip():
    rs = np.random.RandomState(0)
    fb = rs.rand(H, W, 4).astype(np.float32)
    ids = list(range(((W + 15) // 16) * ((H + 15) // 16)))
    out = unpack_tiles_numpy(pack_tiles_numpy(fb, ids), ids, np.zeros_like(fb))
    assert np.array_equal(out, fb)


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        shard = TileShard(W, H, world, rank)
        r = scenes.oracle_scene("c1", W, H)
        tiles_x = (W + 15) // 16
        for t in shard.mine:                      # render only the owned tiles
            tx, ty = t % tiles_x, t // tiles_x
            r.render(SPP, rect=(tx * 16, ty * 16, min(W, tx * 16 + 16), min(H, ty * 16 + 16)), threads=1)
            r.sample = 0
        packed = torch.from_numpy(pack_tiles_numpy(r.fb, shard.pack_ids))
        gathered = torch.empty(shard.gathered_floats, dtype=torch.float32)
        shard.all_gather(dist, gathered, packed)
        frame = unpack_tiles_numpy(gathered.numpy(), shard.unpack_ids, np.zeros((H, W, 4), np.float32))
        q.put((rank, frame))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_sharded_render_equals_single_process(world):
    """world_size 2 and 8 (the node size the north star names) over gloo: every rank renders its diagonal share with the oracle, ONE all_gather, every
    rank's unshuffled frame equals the single-process frame bit for bit."""
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(rk, world, port, q)) for rk in range(world)]
    for p in procs:
        p.start()
    frames = dict(q.get(timeout=600) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    ref = scenes.oracle_scene("c1", W, H).render(SPP)
    for rk in range(world):
        assert np.array_equal(frames[rk].view(np.uint32), ref.view(np.uint32)), "rank %d frame differs" % rk


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` must start its ranks itself (no external torchrun), before anything touches a GPU, and
    exit with their status.  --launch-check runs the launcher path only (spawn, rendezvous on 127.0.0.1, one all_gather of a
    per-rank tile buffer) with the gloo backend, so it works in the GPU-less build container."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env["VOLREN_DIST_BACKEND"] = "gloo"
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--launch-check", "--width", "256", "--height", "192"],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    j = json.loads(line)
    assert j["launch_check"] is True and j["n_gpus"] == 2 and j["backend"] == "gloo" and j["tiles_per_rank"] == 96
    # a rank that fails makes the launcher exit non-zero: without --launch-check the ranks need a HIP device
    if not _have_gpu():
        bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--cpu-budget", "0"],
                             env=env, capture_output=True, text=True, timeout=600)
        assert bad.returncode != 0


def _have_gpu():
    try:
        import volren_amd
        return volren_amd.load().vr_device_count() > 0
    except Exception:
        return False


def test_bench_host_sharded_argument_plumbing(monkeypatch):
    """`bench.py --gpus N --host sharded [--devices ...]` (the product's one-process multi-GPU host under the bench) and the child leg rank 0 runs
    after a torch.distributed measurement: arguments, device lists and the launcher variables that must NOT reach the child -- no GPU needed."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    a = bench.parse_args(["--gpus", "3", "--host", "sharded", "--devices", "0,0,0"])
    assert a.host == "sharded" and a.devices == "0,0,0" and bench.parse_args([]).host == "dist"
    # the child leg: one device per former rank (logical shards where the box has fewer devices), launcher variables stripped, errors kept as data
    seen = {}

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env

        class R:
            returncode, stdout, stderr = 0, 'noise\n{"value": 5.0, "frame_crc32": 7, "host": "sharded"}\n', ""
        return R()
    monkeypatch.setattr(bench.subprocess, "run", fake_run)
    for k, v in (("RANK", "0"), ("WORLD_SIZE", "4"), ("LOCAL_RANK", "0"), ("MASTER_PORT", "1234"), ("TORCHELASTIC_RUN_ID", "x")):
        monkeypatch.setenv(k, v)
    leg = bench.sharded_leg(bench.parse_args(["--gpus", "4", "--steps", "2", "--config", "c3", "--spp", "16"]), 4)
    assert leg == {"value": 5.0, "frame_crc32": 7, "host": "sharded"}
    cmd = seen["cmd"]
    assert cmd[cmd.index("--host") + 1] == "sharded" and cmd[cmd.index("--gpus") + 1] == "4" and cmd[cmd.index("--config") + 1] == "c3" and cmd[cmd.index("--spp") + 1] == "16"
    assert len(cmd[cmd.index("--devices") + 1].split(",")) == 4
    assert not any(k in seen["env"] for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "TORCHELASTIC_RUN_ID"))

    def failing_run(cmd, env=None, **kw):
        class R:
            returncode, stdout, stderr = 3, "", "boom"
        return R()
    monkeypatch.setattr(bench.subprocess, "run", failing_run)
    assert "error" in bench.sharded_leg(bench.parse_args(["--gpus", "2"]), 2)
    monkeypatch.undo()
    # under a launcher with more than one rank the mode refuses (it is ONE process driving all devices); without a device it says so
    env = dict(os.environ, WORLD_SIZE="2", RANK="0")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--host", "sharded"], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and "ONE process" in out.stderr
    if not _have_gpu():
        env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK")}
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--host", "sharded", "--devices", "0,0"], env=env, capture_output=True, text=True, timeout=300)
        assert out.returncode != 0 and "HIP device" in (out.stderr + out.stdout)
