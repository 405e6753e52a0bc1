# usage: bash tests/tools_sweep.sh  (diagnostic: scheduler threshold sweep on the GPU box)
for t in "16,4,24,2,12,12,16" "16,4,16,2,12,12,16" "24,8,16,2,16,16,24" "32,8,8,2,24,24,32" "32,16,8,2,32,32,32" "24,8,16,1,16,16,24" "24,8,16,3,16,16,24"; do python tests/tools_sched_stats.py c2 1024 32 $t 2>&1 | grep "thr\|collide\|march\|iterations\|new\|nee\|begin\|escape"; done
