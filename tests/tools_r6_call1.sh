#!/bin/bash
# Round 6, first GPU call: the suite, the bench line, per-block scheduler statistics of the current kernels, occupancy scaling and the two padding
# diagnostics (the inputs of tests/tools_latency_model.py), the drain experiment, and the reproducibility loop of both arithmetic modes.
set -o pipefail
O=gpurun_out/r6a; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" | tee -a $O/summary.txt
tail -n 3 $O/pytest.log | tee -a $O/summary.txt
python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?" | tee -a $O/summary.txt
for c in "c2 1024 128" "c3 1024 128" "c4:512 1024 32" "c5full 2048 16" "c5cloud 2048 8"; do
  timeout -k 10 200 python tests/tools_sched_stats.py $c >> $O/sched_stats.txt 2>&1
done
echo "sched stats done" | tee -a $O/summary.txt
for b in 2 3 4; do
  for c in "c2 1024 256" "c4:512 1024 64" "c5cloud 2048 16"; do
    VR_BLOCKS_PER_CU=$b timeout -k 10 200 python tests/tools_profile_run.py $c 2>&1 | grep "kernel ms" | sed "s|^|== blocks_per_cu $b $c: |" >> $O/occupancy.txt
  done
done
echo "occupancy done" | tee -a $O/summary.txt
AB_CASES="c2:1024:256 c4:512:1024:64" bash tests/tools_ab.sh default sleep4 sleep16 valu64 drain > $O/ab_padding.txt 2>&1
echo "ab done" | tee -a $O/summary.txt
python tests/tools_rank_balance.py c2 1024 1024 1024 diagonal > $O/rank_balance_default.txt 2>&1
VOLREN_AMD_LIB=$PWD/build/exp_drain/libvolren_amd.so python tests/tools_rank_balance.py c2 1024 1024 1024 diagonal > $O/rank_balance_drain.txt 2>&1
echo "rank balance done" | tee -a $O/summary.txt
timeout -k 10 400 python tests/tools_determinism.py c2 1024 1024 100 poison > $O/determinism_plain.txt 2>&1
tail -n 2 $O/determinism_plain.txt | tee -a $O/summary.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_det -- python3 $GRAFT_REPO_ROOT/tests/tools_determinism.py c2 1024 1024 30 poison > $GRAFT_REPO_ROOT/$O/determinism_rocprof.txt 2>&1
tail -n 2 $GRAFT_REPO_ROOT/$O/determinism_rocprof.txt | tee -a $GRAFT_REPO_ROOT/$O/summary.txt
rm -rf $GRAFT_REPO_ROOT/$O/prof_det
