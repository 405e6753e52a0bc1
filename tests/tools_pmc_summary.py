#!/usr/bin/env python3
"""Diagnostic: condenses rocprofv3 --pmc passes of tests/tools_profile_run.py into the per-sample figures of
profiles/r3_hbm_traffic.json (memory-side traffic, L1/L2 request counts, instruction mix).

    python3 tests/tools_pmc_summary.py <dir> <tag>=<samples> [...]  > summary.json
    python3 tests/tools_pmc_summary.py --merge summary.json profiles/r3_hbm_traffic.json c2=c2 c4=c4_512 c3=c3

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


def merge(summary, target, names):
    """--merge <profiles/r3_hbm_traffic.json> c2=c2 c4=c4_512 ...: refresh the measured fields of the committed profile"""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import kernel_source_sha
    old = json.load(open(target))
    old["kernel_source_sha"] = kernel_source_sha()          # bench.py marks the profile stale once the kernel sources differ
    for cfg, tag in names.items():
        if tag not in summary:
            continue
        o, n = old["configs"][cfg], summary[tag]
        o["samples"] = n["samples"]
        for f in ("fetch_bytes_per_sample", "fetch_size_counter_bytes_per_sample", "write_bytes_per_sample", "hbm_bytes_per_sample", "l2_hit_rate", "per_sample", "lane_utilisation", "wave_cycles_share"):
            o[f] = n[f]
        o["traffic_over_algorithmic"] = round(n["hbm_bytes_per_sample"] / o["algorithmic_bytes_per_sample"], 2)
        o.setdefault("tcp", {}).update({"l2_read_latency_cycles": n["tcp"]["l2_read_latency_cycles"], "pending_stall_cycles_per_sample": n["tcp"]["pending_stall_per_sample"]})
    json.dump(old, open(target, "w"), indent=1)


def main():
    if sys.argv[1] == "--merge":
        summary = json.load(open(sys.argv[2]))
        merge(summary, sys.argv[3], dict(a.split("=") for a in sys.argv[4:]))
        return
    root = sys.argv[1]
    res = {}
    for spec in sys.argv[2:]:
        tag, samples = spec.split("=")
        samples = float(samples)
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
        e = {"samples": int(samples), "fetch_bytes_per_sample": round(fetch, 1), "fetch_size_counter_bytes_per_sample": round(fetch_raw, 1), "write_bytes_per_sample": round(write, 1),
             "hbm_bytes_per_sample": round(fetch + write, 1), "l2_hit_rate": round(hit / (hit + miss), 3),
             "per_sample": {"valu": round(g("SQ_INSTS_VALU") / samples, 3),
                            "salu": round(g("SQ_INSTS_SALU") / samples, 3), "lds": round(g("SQ_INSTS_LDS") / samples, 4),
                            "vmem_read": round(g("SQ_INSTS_VMEM_RD") / samples, 4), "vmem_write": round(g("SQ_INSTS_VMEM_WR") / samples, 4),
                            "smem": round(g("SQ_INSTS_SMEM") / samples, 4),
                            "l1_accesses": round(g("TCP_TOTAL_ACCESSES_sum") / samples, 1), "l1_misses_to_l2": round(g("TCP_TCC_READ_REQ_sum") / samples, 1)},
             "lane_utilisation": round(g("SQ_THREAD_CYCLES_VALU") / (g("SQ_INSTS_VALU") * 64.0), 3) if "SQ_INSTS_VALU" in c else None,
             "wave_cycles_share": {"waiting_on_memory": round(g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES"), 3), "issue_stalled": round(g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES"), 3),
                                   "issuing": round(g("SQ_ACTIVE_INST_ANY") / g("SQ_WAVE_CYCLES"), 3)} if "SQ_WAVE_CYCLES" in c else None,
             "tcp": {"pending_stall_per_sample": round(g("TCP_PENDING_STALL_CYCLES_sum") / samples, 1),
                     "l2_read_latency_cycles": round(g("TCP_TCC_READ_REQ_LATENCY_sum") / g("TCP_TCC_READ_REQ_sum"), 0)} if "TCP_TCC_READ_REQ_sum" in c else None,
             "raw": {k: v for k, v in sorted(c.items())}}
        res[tag] = e
    json.dump(res, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
