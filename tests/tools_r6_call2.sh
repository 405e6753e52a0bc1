#!/bin/bash
# Round 6, second GPU call: the suite on the round's code changes, the rare kernel instances A/B, scheduler statistics of the re-instrumented kernels,
# and the inputs of the issue-cycle budget from the level-1 instrumented library.
set -o pipefail
O=gpurun_out/r6b; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" | tee -a $O/summary.txt
tail -n 15 $O/pytest.log | tee -a $O/summary.txt
for v in default rv_base rv_p4 rv_p2; do
  if [ "$v" = default ]; then unset VOLREN_AMD_LIB; else export VOLREN_AMD_LIB=$PWD/build/exp_$v/libvolren_amd.so; fi
  timeout -k 10 240 python tests/tools_rare_variants_ab.py $v 2>&1 | grep "^==" >> $O/rare_variants.txt
done
unset VOLREN_AMD_LIB
echo "rare variants done" | tee -a $O/summary.txt
for c in "c2 1024 128" "c4:512 1024 32"; do
  timeout -k 10 200 python tests/tools_sched_stats.py $c >> $O/sched_stats_lds_counters.txt 2>&1
done
echo "sched stats done" | tee -a $O/summary.txt
VOLREN_AMD_LIB=$PWD/build/exp_stats1/libvolren_amd.so bash tests/tools_issue_reconcile.sh "c2 1024 128" "c3 1024 128" "c4:512 1024 32" "c5cloud 2048 8" > $O/issue_reconcile.txt 2>&1
echo "issue reconcile done" | tee -a $O/summary.txt
