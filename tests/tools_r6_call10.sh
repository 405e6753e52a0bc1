#!/bin/bash
# Round 6, GPU call 10: the inputs of tests/tools_latency_model.py on the FINAL kernels -- occupancy (2 / 3 / 4 wavefronts per SIMD) and the padding A/B
set -o pipefail
O=gpurun_out/r6j; mkdir -p $O
for b in 2 3 4; do
  for c in "c2 1024 256" "c4:512 1024 64" "c5cloud 2048 16"; do
    VR_BLOCKS_PER_CU=$b timeout -k 10 200 python tests/tools_profile_run.py $c 2>&1 | grep "kernel ms" | sed "s|^|== blocks_per_cu $b $c: |" >> $O/occupancy.txt
  done
done
AB_CASES="c2:1024:256 c4:512:1024:64" bash tests/tools_ab.sh default sleep4 sleep16 valu64 > $O/ab_padding.txt 2>&1
cat $O/occupancy.txt; grep "^==" $O/ab_padding.txt
