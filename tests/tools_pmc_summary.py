#!/usr/bin/env python3
"""Diagnostic: condenses rocprofv3 --pmc passes of tests/tools_profile_run.py into the per-sample figures of
profiles/r4_hbm_traffic.json (memory-side traffic, L1/L2 request counts, instruction mix).

    python3 tests/tools_pmc_summary.py <dir>  > summary.json          (<dir>/pmc_specs.json names the passes: tools_collect_profiles.sh writes it)

<dir> holds one sub-directory per pass, pmc_<tag>_<first counter of the set>/out_counter_collection.csv (the layout the
collection scripts under build/ write); <samples> = pixel-samples traced by ALL path-tracing dispatches of one pass.
Only dispatches of pathtrace_kernel are summed.  FETCH_SIZE / WRITE_SIZE are in KiB (rocprofv3 derived metrics).
"""
import csv
import glob
import json
import os
import sys


def sums(path):
    out = {}
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if "pathtrace_kernel" not in row["Kernel_Name"]:
                continue
            out[row["Counter_Name"]] = out.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
    return out


def main():
    root = sys.argv[1]
    specs = json.load(open(os.path.join(root, "pmc_specs.json")))        # written by tools_collect_profiles.sh: tag -> key, scene, frame, spp, samples, command
    for spec in sys.argv[2:]:                                             # legacy form <tag>=<samples>
        tag, samples = spec.split("=")
        specs.setdefault(tag, {})["samples"] = float(samples)
    res = {}
    for tag, sp in specs.items():
        samples = float(sp["samples"])
        c = {}
        for d in sorted(glob.glob(os.path.join(root, "pmc_%s_*" % tag))):
            f = os.path.join(d, "out_counter_collection.csv")
            if os.path.exists(f):
                c.update(sums(f))
        if not c:
            continue
        g = lambda k: c.get(k, float("nan"))
        # FETCH_SIZE tallies every memory-side read request at 64 bytes, but a request fills a whole 128-byte line -- for coalesced streams (the guide's
        # "reports exactly 1/2", MI355X_MICROARCH.md HBM) and, calibrated on this kernel's own patterns (tests/tools_fetch_calibration.hip,
        # profiles/r3j_fetch_size_calibration.txt), for divergent dword / dwordx4 gathers and 64-byte slot reads alike: 1 request per line, 64 B reported.
        # WRITE_SIZE is exact for its 32-byte sector and 16-byte coalesced writes.  Hence fetch = 2 x FETCH_SIZE.
        fetch_raw = g("FETCH_SIZE") * 1024.0 / samples
        fetch, write = 2.0 * fetch_raw, g("WRITE_SIZE") * 1024.0 / samples
        hit, miss = g("TCC_HIT_sum"), g("TCC_MISS_sum")
        e = {"key": sp.get("key", tag), "command": sp.get("command"), "scene": sp.get("scene"), "frame": [sp.get("width"), sp.get("height")], "spp": sp.get("spp"),
             "samples": int(samples), "fetch_bytes_per_sample": round(fetch, 1), "fetch_size_counter_bytes_per_sample": round(fetch_raw, 1), "write_bytes_per_sample": round(write, 1),
             "hbm_bytes_per_sample": round(fetch + write, 1), "l2_hit_rate": round(hit / (hit + miss), 3),
             "per_sample": {"valu": round(g("SQ_INSTS_VALU") / samples, 3),
                            "salu": round(g("SQ_INSTS_SALU") / samples, 3), "lds": round(g("SQ_INSTS_LDS") / samples, 4),
                            "vmem_read": round(g("SQ_INSTS_VMEM_RD") / samples, 4), "vmem_write": round(g("SQ_INSTS_VMEM_WR") / samples, 4),
                            "smem": round(g("SQ_INSTS_SMEM") / samples, 4),
                            "l1_accesses": round(g("TCP_TOTAL_ACCESSES_sum") / samples, 1), "l1_misses_to_l2": round(g("TCP_TCC_READ_REQ_sum") / samples, 1)},
             "lane_utilisation": round(g("SQ_THREAD_CYCLES_VALU") / (g("SQ_INSTS_VALU") * 64.0), 3) if "SQ_INSTS_VALU" in c else None,
             "wave_cycles_share": {"waiting_on_memory": round(g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES"), 3), "issue_stalled": round(g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES"), 3),
                                   "issuing": round(g("SQ_ACTIVE_INST_ANY") / g("SQ_WAVE_CYCLES"), 3),
                                   "issuing_valu": round(g("SQ_ACTIVE_INST_VALU") / g("SQ_WAVE_CYCLES"), 3) if "SQ_ACTIVE_INST_VALU" in c else None} if "SQ_WAVE_CYCLES" in c else None,
             "tcp": {"pending_stall_per_sample": round(g("TCP_PENDING_STALL_CYCLES_sum") / samples, 1),
                     "l2_read_latency_cycles": round(g("TCP_TCC_READ_REQ_LATENCY_sum") / g("TCP_TCC_READ_REQ_sum"), 0)} if "TCP_TCC_READ_REQ_sum" in c else None,
             "raw": {k: v for k, v in sorted(c.items())}}
        res[tag] = e
    json.dump(res, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
