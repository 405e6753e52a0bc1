# usage: bash tests/tools_sweep.sh  (diagnostic: scheduler threshold sweep on the GPU box)
for t in "64,0,32,2,48,48,64" "64,0,48,2,48,48,64" "64,0,56,2,56,56,64" "64,0,64,2,64,64,64" "48,0,40,2,40,40,48" "64,0,24,2,56,56,64"; do python tests/tools_sched_stats.py c2 1024 32 $t 2>&1 | grep "thr\|collide\|nee \|new \|escape\|march"; done
