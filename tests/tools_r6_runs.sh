#!/bin/bash
# Round 6: the GPU calls of the round, one function each, in the order they were made (records under profiles/r6*; `gpurun -- bash tests/tools_r6_runs.sh <name>`).
# Each was its own script while the round ran; they are kept as the record of the exact commands behind the numbers.
set -o pipefail

# Round 6, first GPU call: the suite, the bench line, per-block scheduler statistics of the current kernels, occupancy scaling and the two padding
# diagnostics (the inputs of tests/tools_latency_model.py), the drain experiment, and the reproducibility loop of both arithmetic modes.
call1() {
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
}

# Round 6, second GPU call: the suite on the round's code changes, the rare kernel instances A/B, scheduler statistics of the re-instrumented kernels,
# and the inputs of the issue-cycle budget from the level-1 instrumented library.
call2() {
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
}

# Round 6, third GPU call: the single-frame split what-if (verdict r5 #4) and the environment warp with single-record loads again (the model says the shared
# memory path, not latency, is the scarce resource: does the round-5 trade of 15 more accesses for 3 fewer round trips still hold?)
call3() {
O=gpurun_out/r6c; mkdir -p $O
python tests/tools_split_whatif.py c2 1024 1024 > $O/split_whatif.txt 2>&1
python tests/tools_split_whatif.py c4:512 1024 256 >> $O/split_whatif.txt 2>&1
grep "one launch" $O/split_whatif.txt | tee -a $O/summary.txt
AB_CASES="c2:1024:256 c4:512:1024:64 c3:1024:256" bash tests/tools_ab.sh default envs > $O/ab_env_singles.txt 2>&1
grep "^==" $O/ab_env_singles.txt | tee -a $O/summary.txt
}

# Round 6, GPU call 4: the bench line under rocprofv3 --kernel-trace --stats and the PMC passes of c2, c3, c4 (tests/tools_collect_profiles.sh)
call4() {
bash tests/tools_collect_profiles.sh bench 2>&1 | tail -5
for spec in "c2 c2 c2 1024x1024 512" "c3 c3 c3 1024x1024 512" "c4 c4 c4:512 1024x1024 256"; do
  PMC_ONLY="$spec" bash tests/tools_collect_profiles.sh pmc 2>&1 | grep -E "^pmc|failed" 
  set -- $spec; cp gpurun_out/prof/pmc_specs.json gpurun_out/prof/pmc_specs_$1.json
done
}

# Round 6, GPU call 5: the compact (RGBE) environment map -- the suite, an A/B through VR_ENV_RGBE=0/1 (same library), and the fabric traffic of c4 / c5cloud both ways
call5() {
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
}

# Round 6, GPU call 6: final profiles, part 1 -- the bench line under rocprofv3 and the PMC passes of c2, c3, c4 (final kernels)
call6() {
bash tests/tools_collect_profiles.sh bench 2>&1 | tail -3
for spec in "c2 c2 c2 1024x1024 512" "c3 c3 c3 1024x1024 512" "c4 c4 c4:512 1024x1024 256" "c4_1080p c4@1920x1080x4096 c4:512 1920x1080 128"; do
  PMC_ONLY="$spec" bash tests/tools_collect_profiles.sh pmc 2>&1 | grep -E "^pmc|failed"
  set -- $spec; cp gpurun_out/prof/pmc_specs.json gpurun_out/prof/pmc_specs_$1.json
done
}

# Round 6, GPU call 7: final profiles, part 2 -- PMC passes of the 2048^2 frames, the issue budget's inputs (level-1 instrumented library), rank balance, every instance's throughput
call7() {
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
}

# Round 6, GPU call 8: -DVR_WORLD_SLOT=1 (collision events without their cold read): the suite under the experiment library, then the A/B
call8() {
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
}

# Round 6, GPU call 9: collision events without their cold read + one sector per event (the new default): the suite, then the A/B against the same without the swapped
# slot layout (ws) and without the experiment (ws0); the emission kernels with it (wse, 6 spilled VGPRs)
call9() {
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
}

# Round 6, GPU call 10: the inputs of tests/tools_latency_model.py on the FINAL kernels -- occupancy (2 / 3 / 4 wavefronts per SIMD) and the padding A/B
call10() {
O=gpurun_out/r6j; mkdir -p $O
for b in 2 3 4; do
  for c in "c2 1024 256" "c4:512 1024 64" "c5cloud 2048 16"; do
    VR_BLOCKS_PER_CU=$b timeout -k 10 200 python tests/tools_profile_run.py $c 2>&1 | grep "kernel ms" | sed "s|^|== blocks_per_cu $b $c: |" >> $O/occupancy.txt
  done
done
AB_CASES="c2:1024:256 c4:512:1024:64" bash tests/tools_ab.sh default sleep4 sleep16 valu64 > $O/ab_padding.txt 2>&1
cat $O/occupancy.txt; grep "^==" $O/ab_padding.txt
}

# Round 6, GPU call 11: the tolerance mode with and without the world slot (the bench line under rocprofv3 had it 6 % SLOWER than the exact kernels)
call11() {
O=gpurun_out/r6k; mkdir -p $O
for round in 1 2; do
python tests/tools_fast_math.py c2 1024 256 2>&1 | grep "spp:" | sed "s|^|== default: |" | tee -a $O/fast.txt
VOLREN_AMD_LIB=$PWD/build/exp_fws0/libvolren_amd.so python tests/tools_fast_math.py c2 1024 256 2>&1 | grep "spp:" | sed "s|^|== tolerance kernel without the world slot: |" | tee -a $O/fast.txt
done
python tests/tools_fast_math.py c2 1024 1024 2>&1 | grep "spp:" | sed "s|^|== default, 1024 spp: |" | tee -a $O/fast.txt
}

# Round 6, last GPU call: what the driver runs at round end -- the GPU suite, smoke(), the bench command -- on the final tree, and the bench line once more under
# rocprofv3 --kernel-trace --stats (profiles/r6_bench.json, r6_bench_kernel_stats.csv)
final_check() {
O=gpurun_out/r6z; mkdir -p $O; rm -f $O/summary.txt
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" | tee -a $O/summary.txt
tail -n 4 $O/pytest.log | tee -a $O/summary.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?" | tee -a $O/summary.txt
grep -E "smoke" $O/smoke.log | tail -3 | tee -a $O/summary.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc $?" | tee -a $O/summary.txt
grep "^{" $O/bench.json | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('value', d['value'], 'ms_per_step', d['ms_per_step'], 'frac', d['roofline']['frac'], 'traffic stale', d['roofline'].get('traffic_source',{}).get('stale'), 'crc', d['frame_crc32'], 'fast', d['fast_math']['speedup'])
for c in d['configs']: print(c['name'], c.get('value'), c.get('roofline',{}).get('frac'))
" | tee -a $O/summary.txt
bash tests/tools_collect_profiles.sh bench 2>&1 | tail -2
}

# GPU call 12: non-temporal stores for the cold slots and / or the sample pool on the final kernels (with the world slot nobody reads a cold sector back from the L2)
call12() {
O=gpurun_out/r6l; mkdir -p $O
AB_CASES="c2:1024:256 c4:512:1024:64 c4:512:1920x1080:32" bash tests/tools_ab.sh default coldnt snt coldnt_snt > $O/ab_nt.txt 2>&1
grep "^==" $O/ab_nt.txt
}

# GPU call 13: the events' cold stores as full sectors (two dwordx4 per event, unconditional) + non-temporal sample stores by default: the suite, then the A/B against the
# library before both (prev) and with the store change only (nosnt)
call13() {
O=gpurun_out/r6m; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" | tee -a $O/summary.txt
tail -n 4 $O/pytest.log | tee -a $O/summary.txt
AB_CASES="c2:1024:256 c4:512:1024:64 c4:512:1920x1080:32" bash tests/tools_ab.sh prev nosnt default > $O/ab.txt 2>&1
grep "^==" $O/ab.txt | tee -a $O/summary.txt
}

# the other kernels' (transfer function, emission, run-time) events as full 16-byte stores: head = the commit before
call14() {
O=gpurun_out/r6n; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" | tee -a $O/summary.txt
tail -n 4 $O/pytest.log | tee -a $O/summary.txt
AB_CASES="c3:1024:256 c5full:512:2048:8 c5cloud:512:2048:8 c5full:512:1024:32" bash tests/tools_ab.sh head default > $O/ab.txt 2>&1
grep "^==" $O/ab.txt | tee -a $O/summary.txt
}

# the scatter event's loads as four 16-byte loads issued together: preq = the build before, head = before the full-sector stores of call14
call15() {
O=gpurun_out/r6o; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" | tee -a $O/summary.txt
tail -n 4 $O/pytest.log | tee -a $O/summary.txt
AB_CASES="c2:1024:256 c3:1024:256 c4:512:1024:64 c5full:512:2048:8 c5cloud:512:2048:8" bash tests/tools_ab.sh preq default > $O/ab.txt 2>&1
grep "^==" $O/ab.txt | tee -a $O/summary.txt
}

# environment lookups with fewer loads (a row's two texels as one 8-byte load; the last warp record with its texels' importances): head2 = the commit before
call16() {
O=gpurun_out/r6p; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" | tee -a $O/summary.txt
tail -n 4 $O/pytest.log | tee -a $O/summary.txt
AB_CASES="c2:1024:256 c3:1024:256 c4:512:1024:64 c5full:512:2048:8 c5cloud:512:2048:8" bash tests/tools_ab.sh head2 default > $O/ab.txt 2>&1
grep "^==" $O/ab.txt | tee -a $O/summary.txt
}

# full GPU suite, then an A/B: bash tests/tools_r6_runs.sh test_ab <out dir> <variant> ...
test_ab() {
O=gpurun_out/$1; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" | tee -a $O/summary.txt
tail -n 4 $O/pytest.log | tee -a $O/summary.txt
ab "$@"
}
# generic A/B of library variants (each is first checked against the oracle by smoke()): bash tests/tools_r6_runs.sh ab <out dir> <variant> ...
ab() {
O=gpurun_out/$1; shift; mkdir -p $O
AB_CASES=${AB_CASES:-"c2:1024:256 c3:1024:256 c4:512:1024:64 c5full:512:2048:8 c5cloud:512:2048:8"} bash tests/tools_ab.sh "$@" > $O/ab.txt 2>&1
grep -E "^==|smoke" $O/ab.txt | tee -a $O/summary.txt
}

case "$1" in
  ab) shift; ab "$@"; exit $? ;;
  test_ab) shift; test_ab "$@"; exit $? ;;
  call1|call2|call3|call4|call5|call6|call7|call8|call9|call10|call11|final_check|call12|call13|call14|call15|call16) "$1" ;;
  *) echo "usage: bash tests/tools_r6_runs.sh {call1|call2|call3|call4|call5|call6|call7|call8|call9|call10|call11|final_check|call12|call13|call14|call15|call16}"; exit 2 ;;
esac
