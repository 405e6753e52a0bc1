#!/bin/bash
# runs build/tools/fetch_calibration under rocprofv3 --pmc (separate passes) and prints bytes per access for every kernel
ROOT=${GRAFT_REPO_ROOT:-$PWD}; OUT=$ROOT/gpurun_out/calib; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  first=${c%% *}; rm -rf $OUT/$first
  rocprofv3 --pmc $c -d $OUT/$first -o out --output-format csv -- $ROOT/build/tools/fetch_calibration > $OUT/$first.log 2>&1 || echo "pass $first failed"
  f=$(find $OUT/$first -name "*counter_collection.csv" | head -1)
  python3 - <<PY
import csv, collections
t=collections.defaultdict(float)
for r in csv.DictReader(open("$f")): t[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])] += float(r["Counter_Value"])
n = float(1 << 22)
for (k, c), v in sorted(t.items()):
    if k.startswith("__amd"): continue
    unit = 1024.0 if c in ("FETCH_SIZE", "WRITE_SIZE") else 1.0
    print("%-12s %-24s %14.1f  -> %8.2f %s per access" % (k, c, v, v * unit / n, "bytes" if unit > 1 else "requests"))
PY
done
