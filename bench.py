#!/usr/bin/env python3
"""bench.py -- headline benchmark of the volume path-tracing hot path on MI355X.

A "step" is one full frame of BASELINE.json configs[1] ("c2": data/smoke.brick, no transfer function, 1024x1024,
1024 spp, 100 bounces, fov 40, seed 42): W*H*spp pixel-samples through the fused HIP path-tracing kernel, scene
resident in HBM before the timed region.  Metric: Msamples/s = W*H*spp / wall time of the sample loop.

`python bench.py --gpus N` starts its own N ranks (one per GPU, torch.distributed.run on 127.0.0.1) BEFORE anything touches
the GPU and exits with their status; under an external torchrun (RANK/WORLD_SIZE set) it is a rank.  Multi-GPU: the
frame's 16x16 tiles are sharded over the ranks (diagonal interleave), every rank renders its tiles, and the accumulated
radiance is exchanged with ONE RCCL all_gather of the compact per-rank tile buffers per frame.  Total work is fixed, so
scaling is "strong".

`value` is K whole frames one after the other on one stream -- the same measurement for every N; for N > 1 `value_pipelined` adds the same frames
alternating between two renderers / streams (the drain of one frame overlapping the start of the next).

`python bench.py --gpus N --host sharded [--devices a,b,...]` measures the PRODUCT's multi-GPU host instead (include/volren_amd.h vr_sharded_*,
csrc/sharded.cpp: ONE process, N renderers on N devices, one grouped ncclAllGather per frame through librccl) and prints the same line with
`host: "sharded"`, `rccl_ranks: N`, `frame_crc32`.  Under the driver's torchrun command for N > 1 rank 0 also runs that mode as a child process once the
ranks have finished (they have left the GPUs by then) and reports it as `value_sharded` / `sharded` beside the torch.distributed `value`.

Adds to the JSON line:
  value_trace_loop -- (N=1) the same frame driven the way the reference drives it, `while (sample < sppx) trace();` (src/main.cpp:533-537,
                  src/bindings.cpp:124-132): spp x vr_trace + one vr_synchronize per step.  Consecutive trace() calls are coalesced into fused launches
                  (csrc/renderer.h), so this is expected within a few per cent of `value`.
  frame_crc32  -- CRC-32 of rank 0's RGBA32F frame after the last step: the same for every N (and equal to the oracle's frame).
  roofline     -- algorithmic HBM bytes (SURVEY.md 8d formula; event counts from the oracle's instrumented counters on the same config, 8
                  batches of one sample per pixel, their spread = bytes_per_sample_stderr) / HIP-event duration of the path-tracing kernel,
                  vs 8 TB/s; `traffic` from the committed PMC profile of the same configuration at the same frame (profiles/r4_hbm_traffic.json,
                  `stale` when the kernel sources changed since); roofline_issue: the resource that binds on the cached grids -- VALU
                  wave-instructions per sample (same profile) x this run's rate against the SIMDs' issue rate, lane utilisation beside it.
  cpu_baseline -- the CPU oracle ("port") timed on the host cores on a bounded sample of the same workload; cpu_baseline_raymarch: the
                  64-step ray marcher (common.glsl:506-566) SURVEY 8d names; cpu_baseline_c1: BASELINE configs[0] itself (256x256, 16 spp,
                  4 bounces) by the ray marcher and by the DDA trackers.
  configs      -- (N=1, headline config only) the same measurement for BASELINE configs[2] (c3) and configs[3]'s grid (c4, 512^3 dense fp16)
                  at 1024x1024 / 1024 spp, the resolution north_star quotes its 40 % target at, and for configs[3] / configs[4] at their own
                  frames (c4 at 1920x1080 x 4096 spp; c5full and c5cloud = 1024^3 sparse brick grid + emission at 2048x2048 x 4096 spp: the
                  rounds-1-3 stand-in and the grid at the occupancy SURVEY 8d names), two timed frames each.
  fast_math    -- (when built) the tolerance-mode kernels: speed and relative L2 against the bit-exact default.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
SIMDS = 1024                 # 256 CUs x 4 SIMDs
CLOCK_GHZ = 2.4              # peak engine clock, MI355X_MICROARCH.md
CYCLES_PER_VALU_OP = 1.8     # the CHEAPEST opcodes (fp32 add / mul / fma, and / or, right shifts with VGPR operands) occupy a SIMD for 1.73-1.9 cycles with 4 resident wavefronts; the
                             # kernels' own mix costs 2.3-2.4 (profiles/r5_issue_budget.json: static ISA per scheduler section x STATS execution counts x the per-opcode costs of
                             # profiles/r5_instruction_costs.txt) -- roofline_issue uses the mix's figure when the profile has one for the configuration
COUNTER_SPP = 8              # oracle samples per pixel behind events_per_sample (COUNTER_SPP batches of 1 spp: their spread is the standard error)
CONFIG_INDEX = {"c1": 0, "c2": 1, "c3": 2, "c4": 3, "c5": 4}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="c2", help="c1 | c2 | c3 | readme | c4[:N] (synthetic N^3 dense fp16 grid, default 512) | c5[:N] (synthetic sparse brick grid + emission, dense generator) | c5full (1024^3 sparse brick grid + emission built in brick form)")
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--height", type=int, default=1024)
    ap.add_argument("--spp", type=int, default=1024)
    ap.add_argument("--cpu-budget", type=float, default=12.0, help="seconds of CPU-oracle work for cpu_baseline (0 = skip)")
    ap.add_argument("--extra-configs", default=None, help="comma list of further configs measured at N=1 and reported under 'configs' (default: c3,c4 for the headline run; 'none' to skip)")
    ap.add_argument("--force-dist", action="store_true", help="with --gpus 1: still initialise the process group (RCCL) and run the sharded flow pack_tiles -> all_gather -> unpack_tiles with one rank")
    ap.add_argument("--host", default="dist", choices=("dist", "sharded"), help="dist: one process per GPU under torch.distributed (the contract's launcher); sharded: ONE process, the product's ShardedRenderer (vr_sharded_*) on --gpus devices")
    ap.add_argument("--devices", default=None, help="--host sharded: comma list of device ordinals, one per part (repeats = logical shards of one device); default 0..gpus-1")
    ap.add_argument("--launch-check", action="store_true", help="only start the ranks, rendezvous and exchange one tile buffer (no rendering): checks the launcher path")
    return ap.parse_args(argv)


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn_ranks(args):
    """--gpus N without a launcher: start N ranks as CHILD processes and wait.  Nothing in this process has touched the GPU
    (torch is not even imported yet), and no process that has is ever replaced by another program."""
    env = dict(os.environ)
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC (RCCL across processes)
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // args.gpus)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.run(cmd, env=env)
    return proc.returncode


def algorithmic_bytes_per_sample(counters, use_tf, has_emission, dense=False):
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


def cpu_baseline(config, budget_s, integrator=0, kind="port", aspect=1.0, size=None, spp=None):
    """Oracle on the host cores: low-resolution full view of the same scene (same camera and aspect ratio, so the same mix
    of box-missing and cloud pixels), spp chosen to fill ~budget_s seconds -- or the frame (size, spp) as given, repeated to fill the budget."""
    import scenes
    from oracle import binding as ob
    cw, ch = size if size else (512, max(16, int(round(512 / aspect / 2)) * 2))

    def scene():
        o = scenes.oracle_scene(config, cw, ch)
        o.integrator = integrator
        return o
    o = scene()
    o.render(1)                                          # library load, thread pool start
    reps = 1
    if spp is None:
        probe = 1
        while True:                                      # probe long enough (>= 1 s) that burst clocks / quotas do not skew the estimate
            t0 = time.time()
            o.render(probe)
            dt = time.time() - t0
            if dt >= min(1.0, budget_s) or probe >= 1024:
                break
            probe *= 2
        rate = cw * ch * probe / max(dt, 1e-6)
        spp = int(max(1, min(4096, budget_s * rate / (cw * ch))))
        o2 = scene()
        t0 = time.time()
        o2.render(spp)
        dt = time.time() - t0
    else:
        t0 = time.time()
        while True:                                      # the frame as BASELINE names it, as often as fits the budget (at least once)
            o2 = scene()
            o2.render(spp)
            dt = time.time() - t0
            if dt >= budget_s:
                break
            reps += 1
    cores = ob.lib().orc_num_threads()
    what = "oracle/liboracle.so, OpenMP over rows" + (", 64-step ray-marching trackers (common.glsl:506-566)" if integrator == 3 else "")
    return dict(value=cw * ch * spp * reps / dt / 1e6, unit="Msamples/s", cores=int(cores), kind=kind,
                sample="%s scene at %dx%d, %d spp%s (%.1f s of %s)" % (config, cw, ch, spp, " x %d frames" % reps if reps > 1 else "", dt, what))


def event_counters(config, aspect=1.0):
    """Event counts per sample from the oracle's instrumented counters on the same config and seed: COUNTER_SPP batches of one sample
    per pixel of a 256-pixel-wide full view.  Returns (summed counters, list of per-batch counters): the batches' spread gives the standard error
    of the bytes per sample behind roofline.frac."""
    import scenes
    cw = 256
    ch = max(16, int(round(cw / aspect / 2)) * 2)
    o = scenes.oracle_scene(config, cw, ch)
    batches, prev = [], {k: 0 for k in o.counters.as_dict()}
    for _ in range(COUNTER_SPP):
        o.render(1)
        cur = o.counters.as_dict()
        batches.append({k: cur[k] - prev[k] for k in cur})
        prev = cur
    return prev, batches


def profile_json(name):
    path = os.path.join(ROOT, "profiles", name)
    return json.load(open(path)) if os.path.exists(path) else None


KERNEL_SOURCES = ("vr_pathtrace.h", "vr_trace.h", "vr_math.h", "vr_scene.h", "vr_pathtrace.hip")


def kernel_source_sha():
    """Fingerprint of the path-tracing kernel's sources (comments and blank space stripped: only code counts): the PMC profile
    (tests/tools_collect_profiles.sh) records it, and a bench line that quotes the profile's counters for kernels built from other
    sources says so (`stale`)."""
    import hashlib
    import re
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        with open(os.path.join(ROOT, "volren_amd", "csrc", f), "r", encoding="utf-8", errors="replace") as fh:
            text = fh.read()
        text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)          # block comments
        text = re.sub(r"//[^\n]*", " ", text)                        # line comments (no string literal of these files contains //)
        h.update(" ".join(text.split()).encode())
    return h.hexdigest()[:16]


def issue_profile(cfg):
    """Opcode-weighted issue cost of the configuration's kernel (profiles/r5_issue_budget.json, tests/tools_issue_budget_all.sh): cycles per VALU wave-instruction of
    ITS instruction mix, the share of its issue cycles per scheduler section, and the counter SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES of the same run.  c5full runs the
    sibling of c5cloud's kernel (same code, linear majorant table): its entry is c5cloud's, marked approximate."""
    j = profile_json("r6_issue_budget.json") or profile_json("r5_issue_budget.json")
    if not j:
        return None, False
    key = {"c4:512": "c4"}.get(cfg, cfg)
    if key in j:
        return j[key], False
    if cfg.startswith("c5") and "c5cloud" in j:
        return j["c5cloud"], True
    return None, False


def traffic_profile(cfg, w, h):
    """(entry, file, stale) of the newest committed PMC profile taken on this configuration AT THIS FRAME SIZE (the instruction and traffic mix per sample
    depends on the view: c4 at 1920x1080 is not c4 at 1024x1024), or (None, None, None).  Round 4's file is keyed by the bench line's config names and
    records the frame; older files only have the square frames of their rounds."""
    for name in ("r6_hbm_traffic.json", "r5_hbm_traffic.json", "r4_hbm_traffic.json", "r3_hbm_traffic.json", "r2_hbm_traffic.json"):
        tj = profile_json(name)
        if not tj:
            continue
        for key, e in tj.get("configs", {}).items():
            scene = e.get("scene") or {"c4": "c4:512"}.get(key, key)
            frame = e.get("frame") or ([2048, 2048] if key.startswith("c5") else [1024, 1024])
            if scene in (cfg, {"c4": "c4:512"}.get(cfg, cfg)) and list(frame) == [w, h]:
                return e, "profiles/" + name, tj.get("kernel_source_sha") != kernel_source_sha()
    return None, None, None


class Bench:
    """One configuration on this rank's GPU: scene resident, tile shard set up, step() = one frame."""

    def __init__(self, config, w, h, spp, world, rank, local_rank, dist, fast_math=False, pipelined=None):
        import torch
        import scenes
        from volren_amd.shard import TileShard
        self.torch, self.dist, self.world, self.rank = torch, dist, world, rank
        self.config, self.w, self.h, self.spp = config, w, h, spp
        self.shard = TileShard(w, h, world, rank)
        self.sharded = dist is not None                     # pack -> all_gather -> unpack per frame (every N > 1; N = 1 with --force-dist)
        self.staged = self.sharded and dist.get_backend() != "nccl"
        # `value` is K frames one after the other on ONE stream, for every N alike (round 4: rounds 2-3 pipelined the frames of N > 1 only, which made
        # value(8) / value(1) compare two different things).  The fixed cost of a launch is the drain of the persistent wavefronts' path pools at its
        # end (4-5 ms, set by the deepest paths): 2 % of a whole frame on one GPU, 15 % of a rank's share of it on eight.  Consecutive frames are
        # independent, so two renderers on two streams let frame i+1's wavefronts move onto the CUs that frame i's draining workgroups free
        # (profiles/r2_launch_overhead.txt): measured additionally as `value_pipelined` for N > 1 (VOLREN_PIPELINE=1: also for N = 1, where it is worth
        # +0.5 % and makes the rocprofv3 durations of the overlapping kernels include their wait for the CUs; =0: never).
        env_pipe = os.environ.get("VOLREN_PIPELINE")
        self.pipelined = ((world > 1 and env_pipe != "0") or env_pipe == "1") if pipelined is None else bool(pipelined)
        pool_mb = os.environ.get("VOLREN_SAMPLE_POOL_MB")   # several ranks sharing one GPU (tests): a smaller radiance pool per renderer
        self.slots = []
        for k in range(2 if self.pipelined else 1):
            r = scenes.hip_scene(config, w, h, device=local_rank)
            if fast_math:
                r.fast_math = 1
            if pool_mb:
                r.sample_pool_mb = int(pool_mb)
            # a frame is split by the sample pool alone: no probe launch (the renderer's launch sizing by time, launch_target_ms, plans 2-second launches and
            # none of these frames has a longer one), so that every launch of a kernel in the rocprofv3 statistics of this command is a whole (sub-)frame
            r.launch_target_ms = 0
            stream = torch.cuda.Stream() if self.pipelined else torch.cuda.current_stream()
            r.set_stream(stream.cuda_stream)
            slot = dict(r=r, stream=stream)
            if self.sharded:
                r.set_tiles(self.shard.mine)
                slot["packed"] = torch.empty(self.shard.packed_floats, dtype=torch.float32, device="cuda")
                slot["gathered"] = torch.empty(self.shard.gathered_floats, dtype=torch.float32, device="cuda")
            self.slots.append(slot)
        self.r = self.slots[0]["r"]
        self.frame = 0
        if self.sharded:
            self.tiles_dev = torch.from_numpy(self.shard.pack_ids).cuda()
            self.all_tiles_dev = torch.from_numpy(self.shard.unpack_ids).cuda()

    def step(self, alternate=False):
        torch = self.torch
        slot = self.slots[self.frame % len(self.slots)] if alternate else self.slots[0]
        self.last_slot = slot
        self.frame += 1
        r = slot["r"]
        with torch.cuda.stream(slot["stream"]):
            r.reset()
            r.render(self.spp, sync=False)                                      # ONE fused launch: all spp of all owned tiles
            if self.sharded:
                r.pack_tiles(self.tiles_dev.data_ptr(), self.shard.n_max, slot["packed"].data_ptr())
                if self.staged:
                    r.synchronize()
                    hg = torch.empty(slot["gathered"].shape, dtype=slot["gathered"].dtype)
                    self.shard.all_gather(self.dist, hg, slot["packed"].cpu())
                    slot["gathered"].copy_(hg)
                else:
                    self.shard.all_gather(self.dist, slot["gathered"], slot["packed"])   # RCCL over xGMI, once per frame
                r.unpack_tiles(self.all_tiles_dev.data_ptr(), self.world * self.shard.n_max, slot["gathered"].data_ptr())

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def last_renderer(self):
        return self.last_slot["r"]

    def frame_crc32(self):
        """CRC-32 of the RGBA32F frame the last step left on this rank (after the gather: the whole frame on every rank)."""
        import zlib
        r = self.last_renderer()
        r.synchronize()
        return zlib.crc32(r.framebuffer().tobytes()) & 0xFFFFFFFF

    def measure(self, steps, warmup):
        torch, dist = self.torch, self.dist
        import gc
        gc.collect()                                                            # destructors of earlier renderers (16 GiB pools) run now, not inside the timed region
        for _ in range(max(warmup, 1)):
            self.step()
        if self.pipelined:                                                      # the second renderer has run once too (its pools are allocated)
            self.step(alternate=True)
            self.step(alternate=True)
        self.barrier()
        for slot in self.slots:
            slot["r"].synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            self.step()
        self.barrier()
        elapsed = time.perf_counter() - t0
        for slot in self.slots:
            slot["r"].synchronize()                                             # also raises if the kernel watchdog tripped
        piped = None
        if self.pipelined:                                                      # the same K frames alternating between the two renderers / streams
            self.barrier()
            t1 = time.perf_counter()
            for _ in range(steps):
                self.step(alternate=True)
            self.barrier()
            piped = time.perf_counter() - t1
            self.step()                                                         # the kernel's own duration: one more frame with nothing beside it
            self.barrier()
        last_r = self.last_renderer()
        pt_ms = last_r.last_pathtrace_ms()                                      # HIP events around the path-tracing kernels alone, summed over the frame's sub-launches
        last_ms = last_r.last_kernel_ms()                                       # HIP events on the renderer's stream around the last frame's launches
        if dist is not None:
            tmax = torch.tensor([elapsed, piped or 0.0], dtype=torch.float64, device="cuda" if not self.staged else "cpu")
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            elapsed, piped = float(tmax[0].item()), (float(tmax[1].item()) if piped else None)
        samples = float(self.w) * self.h * self.spp
        launches = max(1, last_r.last_launches)                                 # a frame is split so that a sub-launch fits the sample pool
        my_samples = len(self.shard.mine) * 256.0 * self.spp if self.world > 1 else samples
        # pt_ms is the sum over the frame's sub-launches (HIP events around each path-tracing kernel): kernel_ms = its AVERAGE launch duration,
        # samples_per_launch = the average samples of a launch -- what the rocprofv3 kernel statistics of the same run report
        return dict(value=samples * steps / elapsed / 1e6, ms_per_step=elapsed / steps * 1e3, value_pipelined=(samples * steps / piped / 1e6) if piped else None, kernel_ms=(pt_ms if pt_ms > 0 else last_ms) / launches, frame_gpu_ms=last_ms,
                    launches=launches, samples_per_launch=my_samples / launches)

    def measure_trace_loop(self, steps):
        """The reference's own call protocol (src/main.cpp:533-537, src/bindings.cpp:124-132): reset(); while (sample < sppx) trace(); -- spp calls of
        vr_trace and ONE vr_synchronize per frame, timed like measure()."""
        r = self.slots[0]["r"]
        torch = self.torch

        def frame():
            with torch.cuda.stream(self.slots[0]["stream"]):
                r.reset()
                for _ in range(self.spp):
                    r.trace()
                r.flush()
        frame()
        r.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            frame()
        r.synchronize()
        elapsed = time.perf_counter() - t0
        return dict(value=float(self.w) * self.h * self.spp * steps / elapsed / 1e6, ms_per_step=elapsed / steps * 1e3, launches_per_step=max(1, r.last_launches),
                    calls_per_step=self.spp, frame_crc32=self.frame_crc32())

    def roofline(self, m, counted):
        return roofline_of(self.config, self.w, self.h, self.spp, m, counted)


def roofline_of(config, w, h, spp, m, counted):
    """The contract's roofline object for one configuration: m = measure()'s dict (kernel_ms, launches, samples_per_launch), counted = event_counters()."""
    counters, batches = counted
    cfg = config
    use_tf = cfg == "c3"
    use_em = cfg.startswith("c5")
    dense = cfg.startswith("c4")
    b_sample, events = algorithmic_bytes_per_sample(counters, use_tf, use_em, dense=dense)
    per_batch = [algorithmic_bytes_per_sample(b, use_tf, use_em, dense=dense)[0] for b in batches]
    mean_b = sum(per_batch) / len(per_batch)
    stderr = (sum((x - mean_b) ** 2 for x in per_batch) / max(1, len(per_batch) - 1)) ** 0.5 / len(per_batch) ** 0.5
    events["oracle_samples"] = int(counters["samples"])
    events["oracle_spp"] = len(batches)
    achieved = b_sample * m["samples_per_launch"] / (m["kernel_ms"] * 1e-3) / 1e9
    traffic, traffic_src = None, None
    tp, tp_file, stale = traffic_profile(cfg, w, h)               # PMC passes (FETCH_SIZE / WRITE_SIZE), collected separately
    if tp and "hbm_bytes_per_sample" in tp:
        traffic = tp["hbm_bytes_per_sample"] * m["samples_per_launch"]
        traffic_src = {"from_profile": tp_file, "stale": bool(stale),
                       "note": "not measured by this run: rocprofv3 --pmc passes of %s (bytes = 2 x FETCH_SIZE + WRITE_SIZE: profiles/r3j_fetch_size_calibration.txt), scaled to this launch's samples%s" % (
                           tp.get("command", "?"), "; the kernel sources have changed since that profile was taken" if stale else "")}
    variant = "dense" if dense else ("emission" if use_em else "brick")
    kernel_rate = m["samples_per_launch"] / (m["kernel_ms"] * 1e-3)         # samples/s of the kernel alone
    # The resource that binds: vector-instruction issue at partial lane utilisation, not HBM.  VALU wave-instructions per sample come
    # from the PMC profile (SQ_INSTS_VALU of the same kernel; `stale` when the kernel sources changed since), the rate is this run's: achieved =
    # wave-instructions/s the kernel issued, peak = what 1024 SIMDs issue at 2.4 GHz when an instruction occupies a SIMD for 1.8 cycles.
    issue = None
    if tp and "per_sample" in tp and "valu" in tp["per_sample"]:
        valu = tp["per_sample"]["valu"]
        ip, approx = issue_profile(cfg)
        cyc = ip["cycles_per_valu_op"] if ip else CYCLES_PER_VALU_OP       # opcode-weighted: this kernel's instruction mix (verdict r4 #2a); 1.8 = the cheapest opcodes
        peak_issue = SIMDS * CLOCK_GHZ / cyc                                # G wave-instructions/s the SIMDs sustain at this mix
        ach_issue = valu * kernel_rate / 1e9
        issue = {"bound": "valu_issue", "achieved": ach_issue, "peak": peak_issue, "unit": "G wave-instructions/s", "frac": ach_issue / peak_issue,
                 "valu_wave_instructions_per_sample": valu, "lane_utilisation": tp.get("lane_utilisation"), "hbm_frac": achieved / HBM_PEAK_GBS,
                 "cycles_per_valu_op": cyc, "cycles_per_valu_op_cheapest": CYCLES_PER_VALU_OP, "frac_at_cheapest_opcode_cost": ach_issue / (SIMDS * CLOCK_GHZ / CYCLES_PER_VALU_OP),
                 "cycles_source": ("profiles/r6_issue_budget.json (static ISA per scheduler section x STATS execution counts x profiles/r5_instruction_costs.txt)" + (", entry of the sibling kernel c5cloud" if approx else "")) if ip else "profiles/r2_valu_issue_rate.txt",
                 "simds": SIMDS, "clock_ghz": CLOCK_GHZ, "counts_source": tp_file, "stale": bool(stale)}
        if ip:
            # the counter beside the model: quad-cycles a wavefront spends in VALU instructions / its resident quad-cycles, x resident wavefronts per SIMD = VALU pipelines'
            # worth per SIMD; a SIMD sustains 4 / cycles_per_valu_op of them
            share = ip.get("valu_active_share_of_wave_cycles")
            issue["sections_share_of_valu_issue_cycles"] = {k: round(v["share_of_valu_cycles"], 3) for k, v in ip["sections"].items() if v["share_of_valu_cycles"] > 0}
            issue["model_over_pmc_valu_count"] = ip.get("model_over_pmc")
            if share and not approx:                       # (the sibling kernel's counter says nothing about this scene's stalls)
                issue["counter_valu_active_share_of_wave_cycles"] = share
                issue["counter_frac"] = share * cyc            # = (4 x share) / (4 / cyc)
        issue["summary"] = "hbm %.2f / issue %.2f (counter %s) / lanes %s" % (achieved / HBM_PEAK_GBS, ach_issue / peak_issue, ("%.2f" % issue["counter_frac"]) if issue.get("counter_frac") else "?",
                                                                            ("%.2f" % tp["lane_utilisation"]) if tp.get("lane_utilisation") else "?")
    # the same with SURVEY 8d's FUSED framebuffer figure (B_fb = 16 B per pixel per frame = 16 / spp per sample) instead of the 32 B per sample this
    # build's sample pool moves (16 B written by the path-tracing kernel, 16 B read back by the ordered accumulation pass): the pool is the build's own
    # staging, so this is the fraction to hold against a renderer that accumulates in registers
    b_fused = b_sample - 32.0 + 16.0 / max(1, spp)
    achieved_fused = b_fused * m["samples_per_launch"] / (m["kernel_ms"] * 1e-3) / 1e9
    return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "frac_fused_fb": achieved_fused / HBM_PEAK_GBS, "bytes_per_sample_fused_fb": b_fused,
            "traffic": traffic, "traffic_source": traffic_src,
            "kernel": "pathtrace_kernel<TraceCfg<tf=%s, %s>, false>" % ("true" if use_tf else "false", variant + (", majorant levels 0-1 blocked" if cfg.startswith("c5cloud") else "")),
            "kernel_ms": m["kernel_ms"], "launches_per_step": m["launches"], "samples_per_launch": m["samples_per_launch"],
            "bytes_per_sample": b_sample, "bytes_per_sample_stderr": stderr, "events_per_sample": events, "roofline_issue": issue,
            "note": "bytes = algorithmic (SURVEY 8d), event counts from %d oracle samples per pixel (stderr over the batches); the kernel is limited by its work per sample (vector instructions and fully divergent vector-memory accesses at ~74 %% lane utilisation: roofline_issue), not by HBM bandwidth nor by where its gathers hit (DESIGN.md 5, profiles/r3c_whatif_voxel_taps_in_cache.txt): 2-9 useful bytes per 128-byte line" % len(batches)}


def workload_name(config, w, h, spp):
    what = "synthetic dense fp16 grid" if config.startswith("c4") else ("synthetic sparse brick grid + temperature grid (emission)" if config.startswith("c5") else "smoke.brick")
    tf = " + lut.txt" if config == "c3" else (", no transfer function" if not config.startswith(("c4", "c5")) else "")
    size = (" (1024^3 voxels, 16.8 % of 2 M bricks allocated, one connected cloud: SURVEY 8d's occupancy)" if config == "c5cloud" else
            " (1024^3 voxels, 3 % of the bricks allocated in 160 sealed blobs: rounds 1-3's stand-in)" if config.startswith("c5full") else (" (512^3 voxels)" if config in ("c4", "c4:512") else ""))
    return "BASELINE configs[%d] '%s': %s%s%s, %dx%d, %d spp, seed 42, fov 40" % (CONFIG_INDEX.get(config[:2], -1), config, what, size, tf, w, h, spp)


def run_sharded(args):
    """--host sharded: the product's own multi-GPU host.  ONE process; volren_amd.ShardedRenderer = vr_sharded_* (csrc/sharded.cpp): a renderer per device,
    the 16x16 tiles dealt diagonally, every part renders its tiles on its own stream, ONE grouped ncclAllGather per frame (librccl, opened at run time;
    device-to-device copies when parts share a device), part 0 holds the frame.  Same step, same timing brackets, same line as the torch.distributed host."""
    import zlib
    import scenes
    import volren_amd
    w, h, spp = args.width, args.height, args.spp
    devices = [int(x) for x in args.devices.split(",")] if args.devices else list(range(args.gpus))
    if len(devices) != args.gpus:
        raise SystemExit("--devices names %d parts, --gpus %d" % (len(devices), args.gpus))
    if volren_amd.load().vr_device_count() <= 0:
        raise SystemExit("bench.py needs a HIP device: the renderer has no CPU path")
    s = volren_amd.ShardedRenderer(w, h, devices)
    s.each(lambda p: scenes.configure(p, args.config, False))
    pool_mb = os.environ.get("VOLREN_SAMPLE_POOL_MB")
    for p in s.parts:
        p.launch_target_ms = 0                                  # as Bench: a frame is split by the sample pool alone, no probe launch
        if pool_mb:
            p.sample_pool_mb = int(pool_mb)

    def step():
        s.reset()
        s.render(spp, sync=False)
    for _ in range(max(args.warmup, 1)):
        step()
    s.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    s.synchronize()
    elapsed = time.perf_counter() - t0
    samples = float(w) * h * spp
    parts = []
    for i, p in enumerate(s.parts):
        launches = p.last_launches
        pt = p.last_pathtrace_ms()
        parts.append({"device": devices[i], "launches": launches, "pathtrace_ms": pt, "frame_gpu_ms": p.last_kernel_ms()})
    n_tiles = ((w + 15) // 16) * ((h + 15) // 16)
    own0 = sum(1 for t in range(n_tiles) if ((t % ((w + 15) // 16)) + (t // ((w + 15) // 16))) % len(devices) == 0)
    l0 = max(1, parts[0]["launches"])
    m = dict(value=samples * args.steps / elapsed / 1e6, ms_per_step=elapsed / args.steps * 1e3, kernel_ms=(parts[0]["pathtrace_ms"] or parts[0]["frame_gpu_ms"]) / l0,
             launches=l0, samples_per_launch=(own0 * 256.0 * spp if len(devices) > 1 else samples) / l0)
    crc = zlib.crc32(s.framebuffer().tobytes()) & 0xFFFFFFFF
    transport = s.transport
    collective = s.collective if transport == "rccl" else None
    s.close()
    out = {
        "metric": "Msamples/s (pixels x spp / s), volume path tracing",
        "value": m["value"], "unit": "Msamples/s", "n_gpus": len(devices), "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": m["ms_per_step"], "higher_is_better": True, "scaling": "strong", "host": "sharded", "transport": transport, "collective": collective,
        "devices": devices, "distinct_devices": len(set(devices)),
        "frame_crc32": crc, "vs_baseline": None, "dtype": "f32",
        "data": ("synthetic grid (tests/scenes.py generator) + reference envmap" if args.config[:2] in ("c4", "c5") else "reference fixtures (smoke.brick, table_mountain_2_puresky_1k.hdr)" + (", lut.txt" if args.config == "c3" else "")),
        "config": {"workload": workload_name(args.config, w, h, spp),
                   "parallelism": "ONE process, ShardedRenderer (vr_sharded_*): tiles16x16 diagonal-interleaved over %d part(s) on devices %s, 1 gather/frame by %s" % (len(devices), devices, transport)},
        "rccl_ranks": len(devices) if transport == "rccl" else 0, "parts": parts,
        "roofline": roofline_of(args.config, w, h, spp, m, event_counters(args.config, aspect=w / h)) if args.cpu_budget > 0 else None,
    }
    print(json.dumps(out), flush=True)
    return 0


def sharded_leg(args, world):
    """Rank 0 of the torch.distributed run, after the ranks have left the GPUs: the product's one-process host on the same N devices, as a CHILD process
    (a crash or hang in it must not lose the headline line).  Returns the child's line, or {"error": ...}."""
    try:
        import torch
        n_dev = torch.cuda.device_count()
        devices = list(range(world)) if n_dev >= world else [0] * world        # (a test box has one GPU: logical shards)
        env = {k: v for k, v in os.environ.items() if not (k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK", "ROLE_NAME", "ROLE_WORLD_SIZE",
                                                                 "GROUP_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "OMP_NUM_THREADS") or k.startswith(("TORCHELASTIC_", "TORCH_NCCL_", "NCCL_ASYNC")))}
        cmd = [sys.executable, os.path.abspath(__file__), "--host", "sharded", "--gpus", str(world), "--devices", ",".join(str(d) for d in devices), "--steps", str(args.steps),
               "--warmup", str(args.warmup), "--config", args.config, "--width", str(args.width), "--height", str(args.height), "--spp", str(args.spp), "--cpu-budget", "0"]
        proc = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=float(os.environ.get("VOLREN_SHARDED_LEG_TIMEOUT", "120")))
        lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
        if proc.returncode != 0 or not lines:
            return {"error": "exit %d: %s" % (proc.returncode, (proc.stderr or proc.stdout)[-600:])}
        return json.loads(lines[-1])
    except Exception as e:                                     # noqa: BLE001
        return {"error": "%s: %s" % (type(e).__name__, e)}


def main():
    args = parse_args()
    world_env = os.environ.get("WORLD_SIZE")
    if args.host == "sharded":
        if world_env is not None and int(world_env) > 1:
            raise SystemExit("--host sharded is ONE process driving all devices: run it without a launcher (python bench.py --gpus N --host sharded)")
        sys.exit(run_sharded(args))
    if world_env is None and args.gpus > 1:
        sys.exit(spawn_ranks(args))                       # parent: children do the work, rank 0 prints the JSON line
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(world_env or "1")
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE %d != --gpus %d" % (world, args.gpus))

    backend = os.environ.get("VOLREN_DIST_BACKEND", "nccl")
    import torch
    dist = None
    if args.launch_check:
        # launcher path only: rendezvous, one all_gather of a per-rank tile buffer, barrier -- no renderer, no GPU needed with gloo
        import numpy as np
        from volren_amd.shard import TileShard
        if world > 1:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if backend == "nccl":
                torch.cuda.set_device(local_rank)
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            else:
                dist.init_process_group(backend)
        shard = TileShard(args.width, args.height, world, rank)
        dev = "cuda" if backend == "nccl" and world > 1 else "cpu"
        packed = torch.full((shard.packed_floats,), float(rank), dtype=torch.float32, device=dev)
        gathered = torch.empty(shard.gathered_floats, dtype=torch.float32, device=dev)
        if world > 1:
            shard.all_gather(dist, gathered, packed)
            dist.barrier()
            ok = bool(np.array_equal(gathered.cpu().numpy().reshape(world, -1)[:, 0], np.arange(world, dtype=np.float32)))
        else:
            ok = True
        if rank == 0:
            print(json.dumps({"launch_check": ok, "n_gpus": world, "backend": backend if world > 1 else None, "tiles_per_rank": int(shard.n_max)}), flush=True)
        if world > 1:
            dist.destroy_process_group()
        sys.exit(0 if ok else 1)

    import scenes  # noqa: F401
    import volren_amd  # noqa: F401

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the renderer has no CPU path")
    n_dev = torch.cuda.device_count()
    if backend != "nccl":
        local_rank = local_rank % max(1, n_dev)                # test mode: ranks may share a device
    torch.cuda.set_device(local_rank)
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:                                         # --force-dist without a launcher: a one-rank group in this process
            os.environ.setdefault("MASTER_PORT", str(free_port()))
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        # RCCL ("nccl") over xGMI is the real path; VOLREN_DIST_BACKEND=gloo exists only so that the multi-rank flow can be
        # exercised on a box where several ranks have to share one GPU (the collective is then staged through the host)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    w, h, spp = args.width, args.height, args.spp
    b = Bench(args.config, w, h, spp, world, rank, local_rank, dist)
    m = b.measure(args.steps, args.warmup)
    crc = b.frame_crc32() if rank == 0 else None
    trace_loop = None
    if world == 1 and dist is None:
        try:
            trace_loop = b.measure_trace_loop(args.steps)
            trace_loop["same_frame"] = bool(trace_loop["frame_crc32"] == crc)
            trace_loop["ratio_to_value"] = trace_loop["value"] / m["value"]
        except Exception as e:                                 # noqa: BLE001
            trace_loop = {"error": str(e)}

    out = None
    if rank == 0:
        use_tf = args.config == "c3"
        cpu = None
        if world == 1 and dist is None and args.cpu_budget > 0:
            cpu = cpu_baseline(args.config, args.cpu_budget, aspect=w / h)
        counted = event_counters(args.config, aspect=w / h)
        out = {
            "metric": "Msamples/s (pixels x spp / s), volume path tracing",
            "value": m["value"], "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": m["ms_per_step"], "higher_is_better": True, "scaling": "strong",
            # `value`: K frames one after the other on one stream, for every N; value_pipelined (N > 1): the same frames alternating between two renderers
            # on two streams, the drain of frame i overlapping the start of frame i+1
            "value_pipelined": m["value_pipelined"], "pipelined": False,
            # the reference's protocol, `while (sample < sppx) trace();`: spp x vr_trace + one vr_synchronize per frame (coalesced into fused launches)
            "value_trace_loop": (trace_loop or {}).get("value"), "trace_loop": trace_loop,
            "host": "dist",
            "frame_crc32": crc,                                # of the RGBA32F frame on rank 0 after the last step: the same for every N
            "vs_baseline": None, "dtype": "f32", "data": ("synthetic grid (tests/scenes.py generator) + reference envmap" if args.config[:2] in ("c4", "c5") else
                                      "reference fixtures (smoke.brick, table_mountain_2_puresky_1k.hdr)" + (", lut.txt" if use_tf else "")),
            "config": {"workload": workload_name(args.config, w, h, spp),
                       "parallelism": ("tiles16x16 diagonal-interleaved over %d GPU(s), 1 all_gather/frame%s" % (world, "; value_pipelined: consecutive frames over 2 streams" if b.pipelined else "")) if world > 1 else "1 GPU, %d fused launch(es)/frame (sample pool budget %d GiB)%s%s" % (m["launches"], b.r.sample_pool_mb >> 10, "; value_pipelined: consecutive frames over 2 streams" if b.pipelined else "", ", one-rank process group: pack_tiles -> all_gather -> unpack_tiles per frame" if dist is not None else "")},
            "roofline": b.roofline(m, counted),
            "rccl_ranks": int(dist.get_world_size()) if dist is not None else 1,
            "dist_backend": (dist.get_backend() if dist is not None else None),
        }
        if cpu is not None:
            out["cpu_baseline"] = cpu
            for key, fn in (("cpu_baseline_raymarch", lambda: cpu_baseline(args.config, min(args.cpu_budget, 6.0), integrator=3, kind="raymarch-64", aspect=w / h)),
                            # BASELINE configs[0] itself ("c1": smoke.brick + envmap, 256x256, 16 spp, 4 bounces, CPU ray-march reference, no GPU): the frame as
                            # named, by the 64-step ray-marching trackers and by the DDA trackers the reference's kernels use
                            ("cpu_baseline_c1", lambda: dict(cpu_baseline("c1", min(args.cpu_budget, 3.0), integrator=3, kind="raymarch-64", size=(256, 256), spp=16),
                                                             dda=cpu_baseline("c1", min(args.cpu_budget, 3.0), integrator=0, kind="port", size=(256, 256), spp=16)))):
                try:
                    out[key] = fn()
                except Exception as e:                         # noqa: BLE001 -- a missing optional leg must not lose the headline line
                    out[key] = {"error": str(e)}

    # tolerance-mode kernels (v_log/v_rcp/v_sin hardware math): speed and distance from the bit-exact default, same frame
    if world == 1 and dist is None and args.extra_configs != "none" and getattr(b.r, "has_fast_math", lambda: False)():
        ref = b.r.framebuffer().copy()
        bf = Bench(args.config, w, h, spp, world, rank, local_rank, dist, fast_math=True)
        mf = bf.measure(max(1, args.steps - 1), 1)
        import numpy as np
        img = bf.r.framebuffer()
        rl2 = float(np.sqrt(((img[..., :3].astype(np.float64) - ref[..., :3]) ** 2).sum() / max((ref[..., :3].astype(np.float64) ** 2).sum(), 1e-30)))
        import zlib
        out["fast_math"] = {"value": mf["value"], "unit": "Msamples/s", "kernel_ms": mf["kernel_ms"], "speedup": mf["value"] / m["value"],
                            "rel_l2_vs_bit_exact": rl2, "tolerance": 1e-3, "within_tolerance": bool(rl2 <= 1e-3),
                            # the tolerance mode is deterministic too: ONE frame per (scene, frame size, spp, build) -- tests/tools_determinism.py, profiles/r6_determinism.txt;
                            # test_frames_are_reproducible_on_every_compiled_instance holds every instance to it.  (The value belongs to the build: contraction is allowed in
                            # this mode, so a change of the arithmetic's text -- round 6: the compact environment texels -- moves it; the bit-exact frame_crc32 never moves.)
                            "frame_crc32": zlib.crc32(np.ascontiguousarray(img).tobytes()) & 0xFFFFFFFF,
                            "note": "opt-in mode (vr_set_int fast_math 1): hardware transcendentals and reciprocal-based divisions; the headline value above is the bit-exact default"}
        del bf

    # the other single-GPU BASELINE configs at the resolution north_star quotes (driver-run, not builder-only)
    extra = args.extra_configs
    if extra is None:
        extra = "c3,c4,c4@1920x1080x4096,c5full@2048x2048x4096,c5cloud@2048x2048x4096" if (args.config == "c2" and world == 1 and dist is None) else "none"
    if rank == 0 and world == 1 and extra != "none":
        del b
        out["configs"] = []
        for spec in [x for x in extra.split(",") if x]:
            name, _, frame = spec.partition("@")                # name[@WxHxSPP]: a config at its own frame (BASELINE configs[3..4])
            fw, fh, fspp = (int(v) for v in frame.split("x")) if frame else (w, h, spp)
            steps_x = 2 if frame else 3
            try:
                bx = Bench(name, fw, fh, fspp, 1, 0, local_rank, None, pipelined=False)      # one frame at a time: BASELINE configs[3..4] are single frames
                mx = bx.measure(steps_x, 1)
                cx = event_counters(name, aspect=fw / fh)
                out["configs"].append({"name": spec, "workload": workload_name(name, fw, fh, fspp), "value": mx["value"], "unit": "Msamples/s",
                                       "ms_per_step": mx["ms_per_step"], "steps": steps_x, "warmup": 1, "pipelined": False, "roofline": bx.roofline(mx, cx)})
                del bx
            except Exception as e:                             # noqa: BLE001
                out["configs"].append({"name": spec, "error": str(e)})
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # N > 1: the product's own multi-GPU host (vr_sharded_*, one process, RCCL through librccl) on the same devices, beside the torch.distributed value.
        # The ranks have finished their collectives; this process gives its device memory back first.  VOLREN_SHARDED_LEG=0 skips it.
        if world > 1 and os.environ.get("VOLREN_SHARDED_LEG", "1") != "0":
            import gc
            b = None
            gc.collect()
            torch.cuda.empty_cache()
            time.sleep(1.0)                                    # the other ranks are exiting
            # the headline measurement is safe before the leg starts (a kill of rank 0 during the leg's up to VOLREN_SHARDED_LEG_TIMEOUT seconds would
            # otherwise lose it): the line as it stands goes to stderr and to gpurun_out/bench_headline.json; stdout still gets exactly ONE line, below
            try:
                sys.stderr.write("bench.py: headline before the sharded leg: " + json.dumps(out) + "\n")
                sys.stderr.flush()
                os.makedirs(os.path.join(os.path.dirname(os.path.abspath(__file__)), "gpurun_out"), exist_ok=True)
                with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "gpurun_out", "bench_headline.json"), "w") as f:
                    json.dump(out, f)
            except OSError:
                pass
            leg = sharded_leg(args, world)
            out["sharded"] = leg
            out["value_sharded"] = leg.get("value")
            if "frame_crc32" in leg:
                out["sharded_same_frame"] = bool(leg["frame_crc32"] == out["frame_crc32"])
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
