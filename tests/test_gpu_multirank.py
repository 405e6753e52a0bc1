"""The multi-rank flow of bench.py with real HIP renderers, executed on the ONE GPU a test box has.

Every run is a fresh child `python bench.py --gpus N ...`: for N > 1 the launcher spawns its ranks (torch.distributed.run on
127.0.0.1) before anything in it touches the GPU, the ranks share device 0 (VOLREN_DIST_BACKEND=gloo: the collective is staged
through the host, everything else -- set_tiles, the fused render of the rank's tiles, pack_tiles, all_gather, unpack_tiles, and the
same frames pipelined over two streams -- is the path an 8-GPU node runs), and rank 0 prints the JSON line with `frame_crc32` of its RGBA32F frame after
the last step.  A pixel-sample depends on (seed, pixel, sample) only (shader/pathtracer_brick.glsl:28-36), so the CRC must be the
same for every N -- and equal to the CRC of the oracle's frame.  RCCL itself carries one rank's tiles in the `--force-dist` run
(nccl backend, world size 1): init_process_group, all_gather_into_tensor and the stream ordering around it run on the box.

A GPU box admits at most 6 processes of one user on its card (the pytest process and the launcher's elastic agent count: pytest + agent + 4 ranks would sit
exactly at the limit): the largest world here is 3; tests/tools_multirank_log.sh records 4 and 5 ranks (profiles/r4_multirank_*).
"""
import json
import os
import subprocess
import sys
import zlib

import numpy as np
import pytest

import scenes

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H, SPP = 256, 192, 8


def run_bench(n, backend=None, extra=(), config="c2", w=W, h=H, spp=SPP, steps=2, timeout=900):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "LOCAL_WORLD_SIZE", "GROUP_RANK", "TORCHELASTIC_RUN_ID"):
        env.pop(k, None)
    if backend:
        env["VOLREN_DIST_BACKEND"] = backend
    else:
        env.pop("VOLREN_DIST_BACKEND", None)
    env["VOLREN_SAMPLE_POOL_MB"] = "1024"               # several ranks x two pipelined renderers share one GPU
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--config", config, "--width", str(w), "--height", str(h), "--spp", str(spp),
           "--steps", str(steps), "--warmup", "1", "--cpu-budget", "0", "--extra-configs", "none"] + list(extra)
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
    assert out.returncode == 0, "bench.py --gpus %d failed (rc %d)\n%s\n%s" % (n, out.returncode, out.stdout[-2000:], out.stderr[-4000:])
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert lines, out.stdout[-2000:]
    return json.loads(lines[-1])


@pytest.fixture(scope="module")
def oracle_crc():
    ref = scenes.oracle_scene("c2", W, H).render(SPP)
    return zlib.crc32(np.ascontiguousarray(ref, np.float32).tobytes()) & 0xFFFFFFFF


@pytest.mark.timeout(1800)
def test_bench_multirank_frames_equal_the_single_gpu_frame(oracle_crc):
    j1 = run_bench(1)
    assert j1["n_gpus"] == 1 and j1["rccl_ranks"] == 1 and j1["dist_backend"] is None
    assert j1["frame_crc32"] == oracle_crc, "N=1 frame differs from the oracle's"
    assert j1["value"] > 0 and j1["value_pipelined"] is None            # one GPU: frames one after the other, nothing else measured
    for n in (2, 3):
        j = run_bench(n, backend="gloo")
        assert j["n_gpus"] == n and j["rccl_ranks"] == n and j["dist_backend"] == "gloo"
        assert j["value_pipelined"] > 0 and j["scaling"] == "strong"    # N > 1: also the frames pipelined over two renderers / streams
        assert j["frame_crc32"] == oracle_crc, "N=%d frame differs from the N=1 frame" % n
        assert j["roofline"]["samples_per_launch"] < W * H * SPP          # a rank rendered its share, not the frame


@pytest.mark.timeout(900)
def test_bench_one_rank_process_group_over_rccl(oracle_crc):
    """`--force-dist`: world size 1 on the nccl backend -- librccl loads, the communicator comes up on the device, and the frame that went
    through pack_tiles -> all_gather_into_tensor (RCCL) -> unpack_tiles on the renderer's stream is the frame."""
    j = run_bench(1, extra=["--force-dist"])
    assert j["n_gpus"] == 1 and j["rccl_ranks"] == 1 and j["dist_backend"] == "nccl"
    assert j["frame_crc32"] == oracle_crc


@pytest.mark.timeout(900)
def test_bench_pipelining_switch(oracle_crc, monkeypatch):
    """VOLREN_PIPELINE=0: N > 1 without the second renderer; =1: N = 1 with it.  The frame is the frame either way."""
    monkeypatch.setenv("VOLREN_PIPELINE", "0")
    j = run_bench(2, backend="gloo")
    assert j["value_pipelined"] is None and j["frame_crc32"] == oracle_crc
    monkeypatch.setenv("VOLREN_PIPELINE", "1")
    j = run_bench(1)
    assert j["value_pipelined"] > 0 and j["frame_crc32"] == oracle_crc
