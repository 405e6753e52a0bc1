#!/usr/bin/env python3
"""bench.py -- headline benchmark of the volume path-tracing hot path on MI355X.

A "step" is one full frame of BASELINE.json configs[1] ("c2": data/smoke.brick, no transfer function, 1024x1024,
1024 spp, 100 bounces, fov 40, seed 42): W*H*spp pixel-samples through the fused HIP path-tracing kernel, scene
resident in HBM before the timed region.  Metric: Msamples/s = W*H*spp / wall time of the sample loop.

Multi-GPU (launched by torch.distributed.run, one rank per GPU): the frame's 16x16 tiles are sharded over the ranks
(diagonal interleave), every rank renders its tiles, and the accumulated radiance is exchanged with ONE RCCL
all_gather of the compact per-rank tile buffers per frame.  Total work is fixed, so scaling is "strong".

Adds to the JSON line:
  roofline     -- algorithmic HBM bytes (SURVEY.md 8d formula, event counts from the oracle's instrumented counters on
                  the same config) / HIP-event duration of the path-tracing kernel, vs 8 TB/s.
  cpu_baseline -- the CPU oracle ("port") timed on the host cores on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md


def algorithmic_bytes_per_sample(counters, spp, use_tf, has_emission, dense=False):
    """B = 4*N_dda + b_tap*T*N_coll + b_em*N_coll_sv + 200*N_nee + 48*N_esc + B_fb, with
    B_fb = 32 B: 16 B written to the per-sample radiance pool + 16 B read back by the ordered accumulate pass
    (the same 32 B the reference's per-dispatch image read-modify-write costs)."""
    n = float(counters["samples"])
    n_dda = (counters["n_dda_sv"] + counters["n_dda_tr"]) / n
    n_coll = (counters["n_coll_sv"] + counters["n_coll_tr"]) / n
    n_coll_sv = counters["n_coll_sv"] / n
    n_nee = counters["n_nee"] / n
    n_esc = counters["n_esc"] / n
    taps = 8 if use_tf else 1
    b_tap = 2.0 if dense else 9.0          # dense fp16 voxel vs brick tap (4 B indirection + 4 B range + 1 B atlas)
    b = 4.0 * n_dda + b_tap * taps * n_coll + (9.0 if has_emission else 0.0) * n_coll_sv + 200.0 * n_nee + 48.0 * n_esc + 32.0
    return b, dict(N_dda=n_dda, N_coll=n_coll, N_coll_sv=n_coll_sv, N_nee=n_nee, N_esc=n_esc,
                   primary_miss=counters["n_primary_miss"] / n)


def cpu_baseline_and_counters(config, w, h, budget_s):
    """Oracle on the host cores: low-resolution full view of the same scene (same camera, so the same mix of
    box-missing and cloud pixels), spp chosen to fill ~budget_s seconds."""
    import scenes
    from oracle import binding as ob
    cw = ch = 512
    o = scenes.oracle_scene(config, cw, ch)
    o.render(1)                                          # library load, thread pool start
    probe = 1
    while True:                                          # probe long enough (>= 1 s) that burst clocks / quotas do not skew the estimate
        t0 = time.time()
        o.render(probe)
        dt = time.time() - t0
        if dt >= min(1.0, budget_s) or probe >= 1024:
            break
        probe *= 2
    rate = cw * ch * probe / max(dt, 1e-6)
    spp = int(max(1, min(4096, budget_s * rate / (cw * ch))))
    o2 = scenes.oracle_scene(config, cw, ch)
    t0 = time.time()
    o2.render(spp)
    dt = time.time() - t0
    cores = ob.lib().orc_num_threads()
    return dict(value=cw * ch * spp / dt / 1e6, unit="Msamples/s", cores=int(cores), kind="port",
                sample="%s scene at %dx%d, %d spp (%.1f s of oracle/liboracle.so, OpenMP over rows)" % (config, cw, ch, spp, dt)), o2.counters.as_dict()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="c2", help="c1 | c2 | c3 | readme | c4[:N] (synthetic N^3 dense fp16 grid, default 512) | c5[:N] (synthetic sparse brick grid + emission)")
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--height", type=int, default=1024)
    ap.add_argument("--spp", type=int, default=1024)
    ap.add_argument("--cpu-budget", type=float, default=12.0, help="seconds of CPU-oracle work for cpu_baseline (0 = skip)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
        raise SystemExit("WORLD_SIZE %d != --gpus %d" % (world, args.gpus))

    import torch
    import scenes
    import volren_amd

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the renderer has no CPU path")
    n_dev = torch.cuda.device_count()
    if os.environ.get("VOLREN_DIST_BACKEND", "nccl") != "nccl":
        local_rank = local_rank % max(1, n_dev)                # test mode: ranks may share a device
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL ("nccl") over xGMI is the real path; VOLREN_DIST_BACKEND=gloo exists only so that the multi-rank flow can be
        # exercised on a box where several ranks have to share one GPU (the collective is then staged through the host)
        backend = os.environ.get("VOLREN_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    w, h, spp = args.width, args.height, args.spp
    r = scenes.hip_scene(args.config, w, h, device=local_rank)
    stream = torch.cuda.current_stream()
    r.set_stream(stream.cuda_stream)

    from volren_amd.shard import TileShard
    shard = TileShard(w, h, world, rank)
    mine = shard.mine
    packed = gathered = tiles_dev = all_tiles_dev = None
    if world > 1:
        r.set_tiles(mine)
        tiles_dev = torch.from_numpy(shard.pack_ids).cuda()
        all_tiles_dev = torch.from_numpy(shard.unpack_ids).cuda()
        packed = torch.empty(shard.packed_floats, dtype=torch.float32, device="cuda")
        gathered = torch.empty(shard.gathered_floats, dtype=torch.float32, device="cuda")

    kernel_ms = []
    staged = world > 1 and os.environ.get("VOLREN_DIST_BACKEND", "nccl") != "nccl"

    def step():
        r.reset()
        r.render(spp, sync=False)                                           # ONE fused launch: all spp of all owned tiles
        if world > 1:
            r.pack_tiles(tiles_dev.data_ptr(), shard.n_max, packed.data_ptr())
            if staged:
                hg = torch.empty(gathered.shape, dtype=gathered.dtype)
                shard.all_gather(dist, hg, packed.cpu())
                gathered.copy_(hg)
            else:
                shard.all_gather(dist, gathered, packed)                    # RCCL over xGMI, once per frame
            r.unpack_tiles(all_tiles_dev.data_ptr(), world * shard.n_max, gathered.data_ptr())

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    r.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        kernel_ms.append(None)                                              # filled below without syncing inside the loop
    barrier()
    elapsed = time.perf_counter() - t0
    r.synchronize()                                                         # also raises if the kernel watchdog tripped
    last_ms = r.last_kernel_ms()                                            # HIP events around the last path-tracing launch
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    samples_per_step = float(w) * h * spp
    value = samples_per_step * args.steps / elapsed / 1e6

    out = None
    if rank == 0:
        use_tf = args.config == "c3"
        cpu = None
        counters = None
        if world == 1 and args.cpu_budget > 0:
            cpu, counters = cpu_baseline_and_counters(args.config, w, h, args.cpu_budget)
        else:
            _, counters = cpu_baseline_and_counters(args.config, w, h, 0.5)
        b_sample, events = algorithmic_bytes_per_sample(counters, spp, use_tf, args.config.startswith("c5"), dense=args.config.startswith("c4"))
        my_samples = len(mine) * 256.0 * spp if world > 1 else samples_per_step
        launches = max(1, r.last_launches)                       # a frame is split so that a sub-launch fits the sample pool
        launch_ms = last_ms / launches                           # HIP events on the renderer's stream around the frame's launches
        achieved = b_sample * (my_samples / launches) / (launch_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")   # PMC pass (FETCH_SIZE/WRITE_SIZE), collected separately
        if os.path.exists(tpath):
            tj = json.load(open(tpath))
            if tj.get("config") == args.config and tj.get("width") == w and tj.get("height") == h:
                traffic = tj["hbm_bytes_per_sample"] * (my_samples / launches)
        valu = None                                              # what actually bounds the kernel: VALU issue (PMC pass, profiles/)
        ppath = os.path.join(ROOT, "profiles", "r1_g_pmc_counters.json")
        if os.path.exists(ppath) and args.config == "c2" and (w, h) == (1024, 1024):
            c = json.load(open(ppath))["counters"]
            valu = {"busy": c["SQ_ACTIVE_INST_VALU"] / (c["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0 / 4.0),
                    "lane_utilisation": c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_INSTS_VALU"]),
                    "valu_instructions_per_sample": c["SQ_INSTS_VALU"] / float(json.load(open(ppath))["samples"]),
                    "source": "profiles/r1_g_pmc_counters.json (rocprofv3 --pmc, tests/tools_profile_run.py c2 1024 128)"}
        out = {
            "metric": "Msamples/s (pixels x spp / s), volume path tracing",
            "value": value, "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": ("synthetic grid (tests/scenes.py generator) + reference envmap" if args.config[:2] in ("c4", "c5") else
                                      "reference fixtures (smoke.brick, table_mountain_2_puresky_1k.hdr)" + (", lut.txt" if use_tf else "")),
            "config": {"workload": "BASELINE configs[%d] '%s': %s%s, %dx%d, %d spp, seed 42, fov 40" % (
                {"c1": 0, "c2": 1, "c3": 2, "c4": 3, "c5": 4}.get(args.config[:2], -1), args.config, "synthetic dense fp16 grid" if args.config.startswith("c4") else ("synthetic sparse brick grid + temperature grid (emission)" if args.config.startswith("c5") else "smoke.brick"), " + lut.txt" if use_tf else ", no transfer function", w, h, spp),
                "parallelism": "tiles16x16 diagonal-interleaved over %d GPU(s), 1 all_gather/frame" % world if world > 1 else "1 GPU, 1 fused launch/frame"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "kernel": "pathtrace_kernel<%s,false>" % ("true" if use_tf else "false"),
                         "kernel_ms": launch_ms, "launches_per_step": launches, "samples_per_launch": my_samples / launches,
                         "bytes_per_sample": b_sample, "events_per_sample": events,
                         "valu": valu,
                         "note": "bytes = algorithmic (SURVEY 8d); the scene is cache resident and the kernel is bound by VALU issue (vector ALUs busy ~92 % of the time, see `valu` and DESIGN.md 7), not by HBM bandwidth"},
        }
        if cpu is not None:
            out["cpu_baseline"] = cpu
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
