# usage: bash tests/tools_sweep.sh  (diagnostic: scheduler threshold sweep on the GPU box)
for t in "16,4,24,2,12,12,16" "16,4,32,2,12,12,16" "16,4,16,2,12,12,16" "24,4,24,2,20,20,24" "12,4,24,2,8,8,12" "16,4,24,1,12,12,16" "16,4,24,3,12,12,16" "32,4,24,2,28,28,32" "32,4,12,2,28,28,32"; do python tests/tools_sched_stats.py c2 1024 32 $t 2>&1 | grep "thr\|collide\|nee \|resident"; done
