#!/bin/bash
# Build container: the issue-cycle budget of every bench configuration's kernel from the inputs tests/tools_issue_reconcile.sh left under gpurun_out/issue
# (STATS counters + rocprofv3 instruction counters of the same launches).  Writes profiles/r5_issue_budget.txt and profiles/r5_issue_budget.json.
# usage: bash tests/tools_issue_budget_all.sh [hot pairs per iteration, default 4] [round prefix, default r6]
set -e
HP=${1:-4}; R=${2:-r6}
mkdir -p build/asm
for v in 0 1 4; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Iinclude -fno-slp-vectorize -DVR_PT_VARIANT=$v --cuda-device-only -gline-tables-only -S volren_amd/csrc/vr_pathtrace.hip -o build/asm/ptg_$v.s 2>/dev/null &
done
wait
python3 - "$HP" "$R" <<'PYEOF'
import csv, collections, json, os, subprocess, sys
hp, R = sys.argv[1], sys.argv[2]
root = os.getcwd()
cfgs = [("c2", "c2", "ptg_0.s", "TraceCfgILb0ELi0ELi0ELi0E"), ("c3", "c3", "ptg_0.s", "TraceCfgILb1ELi0ELi0ELi0E"), ("c4", "c4_512", "ptg_1.s", "TraceCfgILb0ELi0ELi0ELi1E"), ("c5cloud", "c5cloud", "ptg_4.s", "TraceCfgILb0ELi0ELi1ELi0ELi1E")]
out_txt, out_json = [], {}
for name, tag, asm, kernel in cfgs:
    sj = os.path.join(root, "gpurun_out/issue/%s_stats.json" % tag)
    cj = os.path.join(root, "gpurun_out/issue/%s_SQ_INSTS_VALU.csv" % tag)
    if not (os.path.exists(sj) and os.path.exists(cj)):
        continue
    st = json.load(open(sj))
    acc = collections.Counter()
    for row in csv.DictReader(open(cj)):
        if "pathtrace_kernel" in row["Kernel_Name"] and not row["Kernel_Name"].rstrip().endswith("true>(vr::KernelArgs)"):
            acc[row["Counter_Name"]] += float(row["Counter_Value"])
    # the plain (not instrumented) launches of the run rendered the frame once: warm-up render = probe launch + the rest
    per = {k: v / st["samples"] for k, v in acc.items()}
    wave = collections.Counter()
    wj = os.path.join(root, "gpurun_out/issue/%s_SQ_WAVE_CYCLES.csv" % tag)
    if os.path.exists(wj):
        for row in csv.DictReader(open(wj)):
            if "pathtrace_kernel" in row["Kernel_Name"] and not row["Kernel_Name"].rstrip().endswith("true>(vr::KernelArgs)"):
                wave[row["Counter_Name"]] += float(row["Counter_Value"])
    js = os.path.join(root, "build/asm/budget_%s.json" % name)
    cmd = [sys.executable, "tests/tools_issue_budget.py", "build/asm/" + asm] + (["--no-environment"] if name == "c3" else []) + ["--kernel", kernel, "--stats", sj, "--hot-pairs", hp, "--valu-per-sample", "%.3f" % per["SQ_INSTS_VALU"], "--msamples", "%.1f" % (st["samples"] / st["ms"] / 1e3 * 1.27), "--json", js]
    txt = subprocess.check_output(cmd).decode()
    j = json.load(open(js))
    j["pmc_per_sample"] = per
    if wave.get("SQ_WAVE_CYCLES"):
        j["valu_active_share_of_wave_cycles"] = wave["SQ_ACTIVE_INST_VALU"] / wave["SQ_WAVE_CYCLES"]       # counter: quad-cycles a wavefront spends in VALU instructions / its resident quad-cycles
        j["any_inst_active_share_of_wave_cycles"] = wave["SQ_ACTIVE_INST_ANY"] / wave["SQ_WAVE_CYCLES"]
    # the budget against the kernel's TIME: SIMD cycles per sample = SIMDs x clock / the kernel's own rate in the committed bench line (HIP events around the kernel)
    try:
        bl = json.load(open(os.path.join(root, "profiles/%s_bench.json" % R)))
        rf = bl["roofline"] if name == "c2" else next(c["roofline"] for c in bl["configs"] if c["name"].split("@")[0] == name)
        rate = rf["samples_per_launch"] / (rf["kernel_ms"] * 1e-3)
        simd_cycles = 1024 * 2.4e9 / rate
        valu_cyc = j["valu_issue_cycles_per_iteration"] * j["iterations_per_sample"] / j["model_over_pmc"]        # scaled to the hardware's instruction count
        j["simd_cycles_per_sample"] = simd_cycles
        j["valu_issue_cycles_per_sample"] = valu_cyc
        j["valu_issue_share_of_kernel_time"] = valu_cyc / simd_cycles
    except Exception as e:
        print("no bench line for", name, e)
    out_json[name] = j
    out_txt.append("== %s (%s, %s x %s x %s spp; kernel %s of %s)\n%s" % (name, st["config"], st["width"], st["height"], st["spp"], kernel, asm, "\n".join(l for l in txt.split("\n") if not l.startswith("SIMD cycles available"))))
    if "simd_cycles_per_sample" in j:
        out_txt.append("kernel time: %.0f SIMD cycles per sample (1024 SIMDs x 2.4 GHz / the kernel's %.0f Msamples/s, profiles/%s_bench.json) = %.0f cycles of VALU issue (model scaled to the PMC instruction count: %.0f %%) + %.0f cycles in which the SIMD issues no VALU instruction (all four wavefronts waiting on memory, LDS, the scalar unit or a dependency)" % (
            j["simd_cycles_per_sample"], 1024 * 2.4e3 / j["simd_cycles_per_sample"], R, j["valu_issue_cycles_per_sample"], 100 * j["valu_issue_share_of_kernel_time"], j["simd_cycles_per_sample"] - j["valu_issue_cycles_per_sample"]))
    if "valu_active_share_of_wave_cycles" in j:
        out_txt.append("counter: SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES = %.3f of a wavefront's resident time in VALU instructions; x 4 resident wavefronts = %.2f VALU pipelines' worth per SIMD, of the %.2f a SIMD sustains at this mix's %.2f cycles per instruction (4 / cycles per instruction) -> VALU issue at %.0f %% of its ceiling\n" % (
            j["valu_active_share_of_wave_cycles"], 4 * j["valu_active_share_of_wave_cycles"], 4.0 / j["cycles_per_valu_op"], j["cycles_per_valu_op"], 100 * j["valu_active_share_of_wave_cycles"] * j["cycles_per_valu_op"]))
open(os.path.join(root, "profiles/%s_issue_budget.txt" % R), "w").write(
    "# Round " + R[1:] + ": issue-cycle budget of the path-tracing kernel per scheduler section (tests/tools_issue_budget.py; inputs: tests/tools_issue_reconcile.sh on one MI355X;\n"
    "# round 6: the execution counts come from instrumented kernels built with -DVR_STATS_LEVEL=1 -- counters in LDS, the production kernels' registers, no scratch).\n"
    "# static ISA of the production kernel (per basic block, attributed to the scheduler's blocks by source line) x executions per iteration (STATS counters of the same run)\n"
    "# x issue cycles per opcode class (profiles/r5_instruction_costs.txt).  `model / PMC`: the model's VALU wave-instructions per sample against rocprofv3 SQ_INSTS_VALU of the\n"
    "# plain launches of the same run.\n\n" + "\n".join(out_txt))
json.dump(out_json, open(os.path.join(root, "profiles/%s_issue_budget.json" % R), "w"), indent=1)
print("\n".join(out_txt))
PYEOF
