"""Files what tests/tools_collect_profiles.sh left under gpurun_out/prof as profiles/r4_*: the bench line taken under rocprofv3, its kernel statistics,
the condensed PMC passes, and the traffic profile bench.py quotes (per bench-line key: memory-side bytes, instruction mix, lane utilisation, with the
algorithmic bytes of the bench line beside them and the kernel sources' fingerprint).  usage: python tests/tools_save_profiles.py [round-prefix]"""
import json
import os
import shutil
import subprocess
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import kernel_source_sha  # noqa: E402
R = sys.argv[1] if len(sys.argv) > 1 else "r5"
P = os.path.join(ROOT, "gpurun_out", "prof")
prof = lambda name: os.path.join(ROOT, "profiles", "%s_%s" % (R, name))
d = None
if os.path.exists(os.path.join(P, "bench_under_rocprof.json")):
    shutil.copy(os.path.join(P, "bench_kernel_stats.csv"), prof("bench_kernel_stats.csv"))
    line = [x for x in open(os.path.join(P, "bench_under_rocprof.json")) if x.startswith("{")][-1]
    d = json.loads(line)
    json.dump(d, open(prof("bench.json"), "w"), indent=1)
elif os.path.exists(prof("bench.json")):
    d = json.load(open(prof("bench.json")))
# the PMC passes, re-condensed here (a change of tools_pmc_summary.py then needs no new GPU run)
summary = json.loads(subprocess.check_output([sys.executable, os.path.join(ROOT, "tests", "tools_pmc_summary.py"), P]))
json.dump(summary, open(prof("pmc_summary.json"), "w"), indent=1)
alg = {}
if d:
    alg["c2"] = d["roofline"]["bytes_per_sample"]
    for c in d.get("configs", []):
        if "roofline" in c:
            alg[c["name"]] = c["roofline"]["bytes_per_sample"]
t = {"round": "%s: final kernels of the round; tests/tools_collect_profiles.sh (separate rocprofv3 --pmc passes of every bench configuration at its own frame, launches of 0.1-0.4 s), "
              "condensed by tests/tools_pmc_summary.py" % R,
     "source": "rocprofv3 --pmc, separate passes (FETCH_SIZE / WRITE_SIZE / TCC / SQ / TCP); pathtrace_kernel dispatches only, 2 launches each",
     "note": ("fetch_bytes_per_sample = 2 x rocprofv3 FETCH_SIZE (every read request fills a 128-byte line and is tallied at 64 B: MI355X_MICROARCH.md 'HBM' for coalesced streams, "
              "tests/tools_fetch_calibration.hip / profiles/r3j_fetch_size_calibration.txt for this kernel's gathers and slot reads); WRITE_SIZE is exact; both are L2<->fabric bytes "
              "(Infinity-Cache hits included)"),
     "kernel_source_sha": kernel_source_sha(),
     "note_sha": "fingerprint of the kernel sources with comments stripped (bench.py kernel_source_sha): bench.py marks quoted counters stale once the code differs",
     "configs": {}}
for tag, e in summary.items():
    key = e.get("key", tag)
    c = {k: e[k] for k in ("command", "scene", "frame", "spp", "samples", "fetch_bytes_per_sample", "fetch_size_counter_bytes_per_sample", "write_bytes_per_sample", "hbm_bytes_per_sample",
                           "l2_hit_rate", "per_sample", "lane_utilisation", "wave_cycles_share", "tcp")}
    if key in alg:
        c["algorithmic_bytes_per_sample"] = round(alg[key], 1)
        c["traffic_over_algorithmic"] = round(e["hbm_bytes_per_sample"] / alg[key], 2)
    t["configs"][key] = c
json.dump(t, open(prof("hbm_traffic.json"), "w"), indent=1)
if d:
    print("headline", round(d["value"], 1), round(d["roofline"]["frac"], 4), round(d["roofline"]["kernel_ms"], 2), (d["roofline"].get("roofline_issue") or {}).get("summary"))
    for c in d.get("configs", []):
        r = c.get("roofline", {})
        print(c.get("name"), round(c.get("value", 0), 1), round(r.get("frac", 0), 4), round(r.get("kernel_ms", 0), 2), (r.get("roofline_issue") or {}).get("summary"), c.get("error"))
for k, c in t["configs"].items():
    print(k, "traffic %.0f B/sample" % c["hbm_bytes_per_sample"], "= %sx algorithmic" % c.get("traffic_over_algorithmic"), "valu %.0f" % c["per_sample"]["valu"], "lanes", c["lane_utilisation"])
