#!/bin/bash
# Collects the profiles that DESIGN.md / bench.py cite, on the GPU box (one gpurun call):
#   1. rocprofv3 --kernel-trace --stats of the bench command            -> gpurun_out/prof/bench_kernel_stats.csv, bench line
#   2. rocprofv3 --pmc passes (separate runs, counters only) of tests/tools_profile_run.py for every configuration of the bench line AT ITS OWN FRAME
#      (round 4: launches of >= 0.1 s, so that the drain of the pools does not weigh on the lane utilisation; a frame of its own for c4 at 1920x1080)
#      -> gpurun_out/prof/pmc_<tag>_<first counter>/out_counter_collection.csv + pmc_specs.json; condensed by tests/tools_pmc_summary.py,
#      filed under profiles/ by tests/tools_save_profiles.py
# usage: bash tests/tools_collect_profiles.sh [bench|pmc|all]
set -o pipefail
what=${1:-all}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp

if [ "$what" = bench ] || [ "$what" = all ]; then
  rm -rf $OUT/bench_trace
  rocprofv3 --kernel-trace --stats -d $OUT/bench_trace -o out --output-format csv -- python3 $ROOT/bench.py --steps 3 --warmup 1 > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err
  f=$(find $OUT/bench_trace -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $OUT/bench_kernel_stats.csv
  echo "bench under rocprofv3: rc=$? stats=$f"
  tail -c 400 $OUT/bench_under_rocprof.json
fi

if [ "$what" = pmc ] || [ "$what" = all ]; then
  SETS=("FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"
        "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_THREAD_CYCLES_VALU"
        "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU"
        "TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum")
  # tag  bench-line key            scene   frame      spp
  SPECS=("c2 c2 c2 1024x1024 512" "c3 c3 c3 1024x1024 512" "c4 c4 c4:512 1024x1024 256" "c4_1080p c4@1920x1080x4096 c4:512 1920x1080 128"
         "c5full c5full@2048x2048x4096 c5full 2048x2048 128" "c5cloud c5cloud@2048x2048x4096 c5cloud 2048x2048 32")
  [ -n "$PMC_ONLY" ] && SPECS=("$PMC_ONLY")
  echo "{" > $OUT/pmc_specs.json
  sep=""
  for spec in "${SPECS[@]}"; do
    set -- $spec; tag=$1; key=$2; cfg=$3; frame=$4; spp=$5
    w=${frame%x*}; h=${frame#*x}
    # tools_profile_run.py renders twice (warm-up + measured): 2 dispatches of the path-tracing kernel per pass
    echo "$sep \"$tag\": {\"key\": \"$key\", \"scene\": \"$cfg\", \"width\": $w, \"height\": $h, \"spp\": $spp, \"samples\": $((2*w*h*spp)), \"command\": \"tests/tools_profile_run.py $cfg $frame $spp\"}" >> $OUT/pmc_specs.json
    sep=","
    for s in "${SETS[@]}"; do
      first=${s%% *}
      d=$OUT/pmc_${tag}_${first}
      rm -rf $d
      timeout -k 10 400 rocprofv3 --pmc $s -d $d -o out --output-format csv -- python3 $ROOT/tests/tools_profile_run.py $cfg $frame $spp > $d.log 2>&1 || echo "pass $tag $first failed"
      f=$(find $d -name "*counter_collection.csv" | head -1)
      if [ -n "$f" ] && [ "$f" != "$d/out_counter_collection.csv" ]; then cp $f $d/out_counter_collection.csv; fi
      echo "pmc $tag [$s]: $(grep 'kernel ms' $d.log | tail -1)"
    done
  done
  echo "}" >> $OUT/pmc_specs.json
  python3 $ROOT/tests/tools_pmc_summary.py $OUT > $OUT/pmc_summary.json
  python3 - <<PYEOF
import json
s = json.load(open("$OUT/pmc_summary.json"))
for k, v in s.items():
    print(k, {f: v[f] for f in ("fetch_bytes_per_sample", "write_bytes_per_sample", "hbm_bytes_per_sample", "l2_hit_rate", "lane_utilisation", "wave_cycles_share")}, v["per_sample"], v["tcp"])
PYEOF
fi
