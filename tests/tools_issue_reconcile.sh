#!/bin/bash
# Diagnostic (round 5): the inputs of tests/tools_issue_budget.py from ONE run per configuration -- the STATS kernels' execution counts (JSON) and the
# instruction counters of the same launches (rocprofv3 --pmc: the instrumented launch and the plain warm-up launch before it are separate dispatches).
# Round 6: run it with VOLREN_AMD_LIB=$PWD/build/exp_stats1/libvolren_amd.so (VARIANTS="0 1 2 4" bash tests/tools_build_variant.sh stats1 -DVR_STATS_LEVEL=1): instrumented
# kernels that only count executions, with the production kernels' registers and no scratch -- their counts describe the schedule of the kernels the budget prices.
# usage: bash tests/tools_issue_reconcile.sh "c2 1024 128" "c4:512 1024 32" ...
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/issue
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for spec in "$@"; do
  set -- $spec; cfg=$1; size=$2; spp=$3
  tag=${cfg//[:@]/_}
  for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VALU" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES"; do
    first=${set%% *}
    d=$OUT/${tag}_$first
    rm -rf $d
    VR_STATS_JSON=$OUT/${tag}_stats.json timeout -k 10 300 rocprofv3 --pmc $set -d $d -o out --output-format csv -- python3 $ROOT/tests/tools_sched_stats.py $cfg $size $spp > $d.log 2>&1 || echo "pass $tag $first failed"
    f=$(find $d -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && cp $f $OUT/${tag}_$first.csv
    echo "== $tag [$set]"; grep -E "Msamples|iterations" $d.log | head -3
  done
done
python3 - <<PYEOF
import csv, glob, collections, os
for f in sorted(glob.glob("$OUT/*.csv")):
    acc = collections.defaultdict(float); n = collections.Counter()
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "")
        if "pathtrace_kernel" not in k: continue
        kind = "STATS" if "Lb1EEEv" in k else "plain"
        acc[(kind, row["Counter_Name"])] += float(row["Counter_Value"]); n[(kind, row["Counter_Name"])] += 1
    print(os.path.basename(f), {("%s %s" % k): (v, n[k]) for k, v in sorted(acc.items())})
PYEOF
