set -o pipefail
( python tests/tools_rank_balance.py c2 1024 1024 1024; python tests/tools_rank_balance.py c4 1920 1080 4096; python tests/tools_rank_balance.py c5full 2048 2048 4096 ) 2>&1 | grep -v "^/opt\|Preparing\|load volume" > gpurun_out/r3k_rank_balance.txt
cat gpurun_out/r3k_rank_balance.txt
