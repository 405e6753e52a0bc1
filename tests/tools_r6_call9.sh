#!/bin/bash
# Round 6, GPU call 9: collision events without their cold read + one sector per event (the new default): the suite, then the A/B against the same without the swapped
# slot layout (ws) and without the experiment (ws0); the emission kernels with it (wse, 6 spilled VGPRs)
set -o pipefail
O=gpurun_out/r6i; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" | tee -a $O/summary.txt
tail -n 4 $O/pytest.log | tee -a $O/summary.txt
AB_CASES="c2:1024:256 c4:512:1024:64 c4:512:1920x1080:32" bash tests/tools_ab.sh ws0 ws default > $O/ab_ws.txt 2>&1
grep "^==" $O/ab_ws.txt | tee -a $O/summary.txt
AB_CASES="c5full:2048:32 c5cloud:2048:16" bash tests/tools_ab.sh default wse > $O/ab_wse.txt 2>&1
grep "^==" $O/ab_wse.txt | tee -a $O/summary.txt
VOLREN_AMD_LIB=$PWD/build/exp_wse/libvolren_amd.so python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "c5 or emission or scheduler or stale or reproducible or fuzz" > $O/pytest_wse.log 2>&1; echo "pytest (emission kernels with the world slot) rc $?" | tee -a $O/summary.txt
tail -n 3 $O/pytest_wse.log | tee -a $O/summary.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for s in FETCH_SIZE WRITE_SIZE; do
  d=$R/$O/pmc_default_$s; rm -rf $d
  timeout -k 10 300 rocprofv3 --pmc $s -d $d -o out --output-format csv -- python3 $R/tests/tools_profile_run.py c4:512 1024x1024 256 > $d.log 2>&1 || echo "pass failed"
  f=$(find $d -name "*counter_collection.csv" | head -1)
  python3 - "$f" "c4 default(ws+swap) $s" <<'PYEOF' | tee -a $R/$O/traffic.txt
import csv, sys, collections
acc = collections.Counter()
for row in csv.DictReader(open(sys.argv[1])):
    if "pathtrace_kernel" in row["Kernel_Name"]:
        acc[row["Counter_Name"]] += float(row["Counter_Value"])
print(sys.argv[2], dict(acc))
PYEOF
  rm -rf $d
done
