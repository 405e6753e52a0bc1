#!/bin/bash
# Round 6, GPU call 7: final profiles, part 2 -- PMC passes of the 2048^2 frames, the issue budget's inputs (level-1 instrumented library), rank balance, every instance's throughput
set -o pipefail
for spec in "c5full c5full@2048x2048x4096 c5full 2048x2048 128" "c5cloud c5cloud@2048x2048x4096 c5cloud 2048x2048 32"; do
  PMC_ONLY="$spec" bash tests/tools_collect_profiles.sh pmc 2>&1 | grep -E "^pmc|failed"
  set -- $spec; cp gpurun_out/prof/pmc_specs.json gpurun_out/prof/pmc_specs_$1.json
done
O=gpurun_out/r6g; mkdir -p $O
VOLREN_AMD_LIB=$PWD/build/exp_stats1/libvolren_amd.so bash tests/tools_issue_reconcile.sh "c2 1024 128" "c3 1024 128" "c4:512 1024 32" "c5cloud 2048 8" > $O/issue_reconcile.txt 2>&1
echo "issue reconcile done"
python tests/tools_rank_balance.py c2 1024 1024 1024 diagonal > $O/rank_balance.txt 2>&1
python tests/tools_rank_balance.py c4:512 1920 1080 512 diagonal >> $O/rank_balance.txt 2>&1
python tests/tools_rank_balance.py c5cloud 2048 2048 128 diagonal >> $O/rank_balance.txt 2>&1
grep -E "full frame|N=8" $O/rank_balance.txt
python tests/tools_variant_throughput.py > $O/variant_throughput.txt 2>&1
tail -n 11 $O/variant_throughput.txt | cut -c1-200
for c in "c2 1024 128" "c3 1024 128" "c4:512 1024 32" "c5full 2048 16" "c5cloud 2048 8"; do
  timeout -k 10 200 python tests/tools_sched_stats.py $c >> $O/sched_stats.txt 2>&1
done
