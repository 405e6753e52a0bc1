#!/bin/bash
# diagnostic: memory-side traffic (FETCH_SIZE / WRITE_SIZE passes) of one configuration; usage: tools_traffic_quick.sh <cfg> <size> <spp>
ROOT=${GRAFT_REPO_ROOT:-$PWD}; OUT=$ROOT/gpurun_out/traffic; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
cfg=$1; size=$2; spp=$3
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $OUT/$c; rocprofv3 --pmc $c -d $OUT/$c -o out --output-format csv -- python3 $ROOT/tests/tools_profile_run.py $cfg $size $spp > $OUT/$c.log 2>&1
  f=$(find $OUT/$c -name "*counter_collection.csv" | head -1)
  python3 - <<PY
import csv
t=0.0
for r in csv.DictReader(open("$f")):
    if "pathtrace_kernel" in r["Kernel_Name"] and r["Counter_Name"]=="$c": t+=float(r["Counter_Value"])
print("$cfg $c: %.1f B per sample" % (t*1024.0/(2.0*$size*$size*$spp)))
PY
done
