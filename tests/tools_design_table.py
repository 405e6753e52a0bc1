"""Rewrites the measured rows of DESIGN.md section 7 from profiles/r6_bench.json and profiles/r6_hbm_traffic.json, so that the document quotes the filed records
(the prose around the table quotes a few of the same numbers: the tool prints them for a manual check).  usage: python tests/tools_design_table.py"""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B = json.load(open(os.path.join(ROOT, "profiles", "r6_bench.json")))
T = json.load(open(os.path.join(ROOT, "profiles", "r6_hbm_traffic.json")))["configs"]
R5 = {"c2 (headline)": (5295, 0.277, 1.16), "c3": (10218, 0.494, 0.40), "c4": (2200, 0.250, 2.19), "c4 @1080p": (3712, 0.256, 2.14), "c5full @2048²": (4189, 0.216, 1.62),
      "c5cloud @2048²": (954, 0.177, 3.80)}
rows = [("c2 (headline)", B["value"], B["roofline"], T["c2"])]
for c in B["configs"]:
    key = c["name"]
    rows.append((key.split("@")[0] + (" @1080p" if "1920" in key else (" @2048²" if "2048" in key else "")), c["value"], c["roofline"], T[key]))
lines = []
for name, v, rf, tt in rows:
    o = R5[name]
    lines.append("| %s | **%.0f** | %.0f | **%.3f** (%.3f) | %.1f | %.0f (%.2f×) | %.2f | %d / %.3f / %.2f× |" % (
        name, v, rf["bytes_per_sample"], rf["frac"], rf["frac_fused_fb"], rf["kernel_ms"], tt["hbm_bytes_per_sample"], tt["traffic_over_algorithmic"], tt["lane_utilisation"], o[0], o[1], o[2]))
p = os.path.join(ROOT, "DESIGN.md")
s = open(p).read()
a = s.index("| c2 (headline) |")
b = s.index("\n\n", a)
s = s[:a] + "\n".join(lines) + s[b:]
open(p, "w").write(s)
print("\n".join(lines))
print("trace loop %.0f (x %.3f)  tolerance mode %.0f (x %.3f, rel L2 %.1e)  cpu: c2 dda %.1f, c2 raymarch %.1f, c1 raymarch %.1f, c1 dda %.1f" % (
    B["value_trace_loop"], B["value_trace_loop"] / B["value"], B["fast_math"]["value"], B["fast_math"]["value"] / B["value"], B["fast_math"]["rel_l2_vs_bit_exact"],
    B["cpu_baseline"]["value"], B["cpu_baseline_raymarch"]["value"], B["cpu_baseline_c1"]["value"], B["cpu_baseline_c1"]["dda"]["value"]))
for name, c in [("c2", B)] + [(c["name"], c) for c in B["configs"]]:
    ri = c["roofline"].get("roofline_issue") or {}
    print("issue", name, "%.2f" % ri.get("frac", float("nan")), "counter %.2f" % ri.get("counter_frac", float("nan")))
