"""Rewrites the measured rows of DESIGN.md section 7 (and the kernel-statistics sentence) from profiles/r4_bench.json, r4_hbm_traffic.json and
r4_bench_kernel_stats.csv, so that the document quotes the filed records.  usage: python tests/tools_design_table.py"""
import csv
import json
import os
import re
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B = json.load(open(os.path.join(ROOT, "profiles", "r4_bench.json")))
T = json.load(open(os.path.join(ROOT, "profiles", "r4_hbm_traffic.json")))["configs"]
rows = [("c2", dict(value=B["value"], roofline=B["roofline"]))] + [(c["name"], c) for c in B["configs"]]
prefix = {"c2": "| c2:", "c3": "| c3:", "c4": "| c4: synthetic", "c4@1920x1080x4096": "| c4 at BASELINE", "c5full@2048x2048x4096": "| c5full:", "c5cloud@2048x2048x4096": "| **c5cloud**"}
doc = open(os.path.join(ROOT, "DESIGN.md")).read().split("\n")
for k, c in rows:
    r = c["roofline"]
    t = T.get(k, {})
    ri = r.get("roofline_issue") or {}
    for i, line in enumerate(doc):
        if line.startswith(prefix[k]):
            cells = line.split(" | ")
            cells[1] = "**%.0f**" % c["value"]
            cells[2] = "%.0f ± %.1f → %.0f (**%.3f**)" % (r["bytes_per_sample"], r["bytes_per_sample_stderr"], r["achieved"], r["frac"])
            cells[3] = "%.1f" % r["kernel_ms"]
            cells[4] = ("%.0f B (%.2f×)" % (t["hbm_bytes_per_sample"], t["hbm_bytes_per_sample"] / r["bytes_per_sample"])) if t else "—"
            cells[5] = ("%.0f → %.2f" % (ri["valu_wave_instructions_per_sample"], ri["frac"])) if ri else "—"
            cells[6] = ("%.2f" % t["lane_utilisation"]) if t else "—"
            doc[i] = " | ".join(cells)
ks = {row["Name"]: row for row in csv.DictReader(open(os.path.join(ROOT, "profiles", "r4_bench_kernel_stats.csv")))}


def kst(sub):
    for n, r in ks.items():
        if sub in n and "fastmath" not in n:
            return "%s launches, average %.2f ms (min %.2f, max %.2f)" % (r["Calls"], float(r["AverageNs"]) / 1e6, float(r["MinNs"]) / 1e6, float(r["MaxNs"]) / 1e6)
    return "?"


text = "\n".join(doc)
text = re.sub(r"\(c2\) \d+ launches, average [\d.]+ ms \(min [\d.]+, max [\d.]+\)", "(c2) " + kst("TraceCfg<false, 0, 0, 0, 0>, false"), text)
text = re.sub(r"\(c3\) \d+ launches, average [\d.]+ ms \(min [\d.]+, max [\d.]+\)", "(c3) " + kst("TraceCfg<true, 0, 0, 0, 0>, false"), text)
fm, cb, cr, c1 = B["fast_math"], B["cpu_baseline"], B["cpu_baseline_raymarch"], B["cpu_baseline_c1"]
text = re.sub(r"never the parity target\): c2 \d+ Msamples/s \(×[\d.]+\), relative L2 against the bit-exact frame [\d.e-]+\.",
              "never the parity target): c2 %.0f Msamples/s (×%.3f), relative L2 against the bit-exact frame %.1e." % (fm["value"], fm["speedup"], fm["rel_l2_vs_bit_exact"]), text)
text = re.sub(r"c2 by the DDA path tracer [\d.]+ Msamples/s, by the 64-step ray-marching trackers [\d.]+;", "c2 by the DDA path tracer %.1f Msamples/s, by the 64-step ray-marching trackers %.1f;" % (cb["value"], cr["value"]), text)
text = re.sub(r"by the ray marcher [\d.]+, by the DDA trackers [\d.]+ Msamples/s", "by the ray marcher %.1f, by the DDA trackers %.1f Msamples/s" % (c1["value"], c1["dda"]["value"]), text)
open(os.path.join(ROOT, "DESIGN.md"), "w").write(text)
print("DESIGN.md section 7 refreshed from profiles/r4_*")
