#!/bin/bash
# Round 6, third GPU call: the single-frame split what-if (verdict r5 #4) and the environment warp with single-record loads again (the model says the shared
# memory path, not latency, is the scarce resource: does the round-5 trade of 15 more accesses for 3 fewer round trips still hold?)
set -o pipefail
O=gpurun_out/r6c; mkdir -p $O
python tests/tools_split_whatif.py c2 1024 1024 > $O/split_whatif.txt 2>&1
python tests/tools_split_whatif.py c4:512 1024 256 >> $O/split_whatif.txt 2>&1
grep "one launch" $O/split_whatif.txt | tee -a $O/summary.txt
AB_CASES="c2:1024:256 c4:512:1024:64 c3:1024:256" bash tests/tools_ab.sh default envs > $O/ab_env_singles.txt 2>&1
grep "^==" $O/ab_env_singles.txt | tee -a $O/summary.txt
