"""Copies what tests/tools_collect_profiles.sh left under gpurun_out/prof into profiles/r3_* (bench line, kernel statistics, PMC summary, traffic profile
with the algorithmic bytes of the bench line and the kernel sources' fingerprint).  usage: python tests/tools_save_profiles.py"""
import json
import os
import shutil
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import kernel_source_sha  # noqa: E402
P = os.path.join(ROOT, "gpurun_out", "prof")
shutil.copy(os.path.join(P, "bench_kernel_stats.csv"), os.path.join(ROOT, "profiles", "r3_bench_kernel_stats.csv"))
shutil.copy(os.path.join(P, "pmc_summary.json"), os.path.join(ROOT, "profiles", "r3_pmc_summary.json"))
line = [x for x in open(os.path.join(P, "bench_under_rocprof.json")) if x.startswith("{")][-1]
d = json.loads(line)
json.dump(d, open(os.path.join(ROOT, "profiles", "r3_bench.json"), "w"), indent=1)
# the traffic profile: the committed skeleton refreshed from the PMC passes under gpurun_out/prof (re-condensed here, so that a change of
# tools_pmc_summary.py does not need a new GPU run)
import subprocess
summary = subprocess.check_output([sys.executable, os.path.join(ROOT, "tests", "tools_pmc_summary.py"), P, "c2=%d" % (2 * 1024 * 1024 * 128), "c4_512=%d" % (2 * 1024 * 1024 * 32),
                                   "c3=%d" % (2 * 1024 * 1024 * 128), "c5full=%d" % (2 * 2048 * 2048 * 16)])
open(os.path.join(P, "pmc_summary.json"), "wb").write(summary)
shutil.copy(os.path.join(P, "pmc_summary.json"), os.path.join(ROOT, "profiles", "r3_pmc_summary.json"))
shutil.copy(os.path.join(ROOT, "profiles", "r3_hbm_traffic.json"), os.path.join(P, "r3_hbm_traffic.json"))
subprocess.check_call([sys.executable, os.path.join(ROOT, "tests", "tools_pmc_summary.py"), "--merge", os.path.join(P, "pmc_summary.json"), os.path.join(P, "r3_hbm_traffic.json"),
                       "c2=c2", "c4=c4_512", "c3=c3", "c5full=c5full"])
t = json.load(open(os.path.join(P, "r3_hbm_traffic.json")))
t["note"] = ("fetch_bytes_per_sample = 2 x rocprofv3 FETCH_SIZE (every read request fills a 128-byte line and is tallied at 64 B: MI355X_MICROARCH.md 'HBM' for coalesced streams, "
             "tests/tools_fetch_calibration.hip / profiles/r3j_fetch_size_calibration.txt for this kernel's gathers and slot reads); WRITE_SIZE is exact; both are L2<->fabric bytes "
             "(Infinity-Cache hits included)")
alg = {"c2": d["roofline"]["bytes_per_sample"]}
for c in d.get("configs", []):
    key = {"c3": "c3", "c4": "c4", "c5full@2048x2048x4096": "c5full"}.get(c["name"])
    if key and "roofline" in c:
        alg[key] = c["roofline"]["bytes_per_sample"]
for k, v in alg.items():
    if k in t["configs"]:
        t["configs"][k]["algorithmic_bytes_per_sample"] = round(v, 1)
        t["configs"][k]["traffic_over_algorithmic"] = round(t["configs"][k]["hbm_bytes_per_sample"] / v, 2)
t["kernel_source_sha"] = kernel_source_sha()
t["round"] = ("r3, final kernels of the round (tests/tools_collect_profiles.sh: separate rocprofv3 --pmc passes; tools_pmc_summary.py --merge); c5full = the emission kernel at "
              "2048x2048, 16 spp (a short launch: the drain of the pools is ~20 % of it, which lowers its lane utilisation against the 4096-spp bench frame)")
json.dump(t, open(os.path.join(ROOT, "profiles", "r3_hbm_traffic.json"), "w"), indent=1)
print("headline", round(d["value"], 1), round(d["roofline"]["frac"], 4), round(d["roofline"]["kernel_ms"], 2), "stale" , d["roofline"]["traffic_source"]["stale"])
for c in d.get("configs", []):
    print(c.get("name"), round(c.get("value", 0), 1), round(c.get("roofline", {}).get("frac", 0), 4), round(c.get("roofline", {}).get("kernel_ms", 0), 2), c.get("error"))
