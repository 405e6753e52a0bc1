#!/bin/bash
# Round 6, GPU call 8: -DVR_WORLD_SLOT=1 (collision events without their cold read): the suite under the experiment library, then the A/B
set -o pipefail
O=gpurun_out/r6h; mkdir -p $O
VOLREN_AMD_LIB=$PWD/build/exp_ws/libvolren_amd.so python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $O/pytest_ws.log 2>&1; echo "pytest (world slot) rc $?" | tee -a $O/summary.txt
tail -n 6 $O/pytest_ws.log | tee -a $O/summary.txt
AB_CASES="c2:1024:256 c4:512:1024:64 c4:512:1920x1080:32" bash tests/tools_ab.sh default ws > $O/ab_ws.txt 2>&1
grep "^==" $O/ab_ws.txt | tee -a $O/summary.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in default ws; do
  if [ "$v" = default ]; then unset VOLREN_AMD_LIB; else export VOLREN_AMD_LIB=$R/build/exp_$v/libvolren_amd.so; fi
  for s in FETCH_SIZE WRITE_SIZE; do
    d=$R/$O/pmc_${v}_$s; rm -rf $d
    timeout -k 10 300 rocprofv3 --pmc $s -d $d -o out --output-format csv -- python3 $R/tests/tools_profile_run.py c4:512 1024x1024 256 > $d.log 2>&1 || echo "pass failed"
    f=$(find $d -name "*counter_collection.csv" | head -1)
    python3 - "$f" "c4 $v $s" <<'PYEOF' | tee -a $R/$O/traffic_ws.txt
import csv, sys, collections
acc = collections.Counter()
for row in csv.DictReader(open(sys.argv[1])):
    if "pathtrace_kernel" in row["Kernel_Name"]:
        acc[row["Counter_Name"]] += float(row["Counter_Value"])
print(sys.argv[2], dict(acc))
PYEOF
    rm -rf $d
  done
done
