#!/bin/bash
# Round 6, GPU call 5: the compact (RGBE) environment map -- the suite, an A/B through VR_ENV_RGBE=0/1 (same library), and the fabric traffic of c4 / c5cloud both ways
set -o pipefail
O=gpurun_out/r6e; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" | tee -a $O/summary.txt
tail -n 4 $O/pytest.log | tee -a $O/summary.txt
for round in 1 2; do
  for c in "c2 1024 256" "c3 1024 256" "c4:512 1024 64" "c4:512 1920x1080 32" "c5full 2048 32" "c5cloud 2048 16"; do
    for m in 0 1; do
      VR_ENV_RGBE=$m timeout -k 10 200 python tests/tools_profile_run.py $c 2>&1 | grep "kernel ms" | sed "s|^|== rgbe=$m $c: |" | tee -a $O/ab_rgbe.txt
    done
  done
done
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in "c4:512 1024x1024 256" "c5cloud 2048x2048 32"; do
  set -- $c; tag=${1//[:@]/_}
  for m in 0 1; do
    for s in FETCH_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
      first=${s%% *}; d=$R/$O/pmc_${tag}_rgbe${m}_$first; rm -rf $d
      VR_ENV_RGBE=$m timeout -k 10 300 rocprofv3 --pmc $s -d $d -o out --output-format csv -- python3 $R/tests/tools_profile_run.py $c > $d.log 2>&1 || echo "pass failed"
      f=$(find $d -name "*counter_collection.csv" | head -1)
      python3 - "$f" "$tag rgbe=$m $first" <<'PYEOF' | tee -a $R/$O/traffic_rgbe.txt
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
done
