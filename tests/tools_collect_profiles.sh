#!/bin/bash
# Collects the profiles that DESIGN.md / bench.py cite, on the GPU box (one gpurun call):
#   1. rocprofv3 --kernel-trace --stats of the bench command            -> gpurun_out/prof/bench_kernel_stats.csv, bench line
#   2. rocprofv3 --pmc passes (separate runs, counters only) of tests/tools_profile_run.py for c2 / c4:512 / c3 / c5full
#      -> gpurun_out/prof/pmc_<tag>_<first counter>/out_counter_collection.csv, condensed by tests/tools_pmc_summary.py
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
  tail -c 600 $OUT/bench_under_rocprof.json
fi

if [ "$what" = pmc ] || [ "$what" = all ]; then
  SETS=("FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"
        "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_THREAD_CYCLES_VALU"
        "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
        "TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum")
  for spec in "c2 c2 1024 128" "c4_512 c4:512 1024 32" "c3 c3 1024 128" "c5full c5full 2048 16"; do
    set -- $spec; tag=$1; cfg=$2; size=$3; spp=$4
    for s in "${SETS[@]}"; do
      first=${s%% *}
      d=$OUT/pmc_${tag}_${first}
      rm -rf $d
      timeout -k 10 300 rocprofv3 --pmc $s -d $d -o out --output-format csv -- python3 $ROOT/tests/tools_profile_run.py $cfg $size $spp > $d.log 2>&1 || echo "pass $tag $first failed"
      f=$(find $d -name "*counter_collection.csv" | head -1)
      if [ -n "$f" ] && [ "$f" != "$d/out_counter_collection.csv" ]; then cp $f $d/out_counter_collection.csv; fi
      echo "pmc $tag [$s]: $(grep 'kernel ms' $d.log | tail -1)"
    done
  done
  # tools_profile_run.py renders twice (warm-up + measured): 2 dispatches of the path-tracing kernel per pass
  python3 $ROOT/tests/tools_pmc_summary.py $OUT c2=$((2*1024*1024*128)) c4_512=$((2*1024*1024*32)) c3=$((2*1024*1024*128)) c5full=$((2*2048*2048*16)) > $OUT/pmc_summary.json
  cp $ROOT/profiles/r3_hbm_traffic.json $OUT/r3_hbm_traffic.json
  python3 $ROOT/tests/tools_pmc_summary.py --merge $OUT/pmc_summary.json $OUT/r3_hbm_traffic.json c2=c2 c4=c4_512 c3=c3 c5full=c5full
  python3 - <<EOF
import json
s = json.load(open("$OUT/pmc_summary.json"))
for k, v in s.items():
    print(k, {f: v[f] for f in ("fetch_bytes_per_sample", "write_bytes_per_sample", "hbm_bytes_per_sample", "l2_hit_rate", "lane_utilisation", "wave_cycles_share")}, v["per_sample"], v["tcp"])
EOF
fi
