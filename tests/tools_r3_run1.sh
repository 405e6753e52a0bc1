export VOLREN_AMD_LIB=$PWD/build/exp_statsched/libvolren_amd.so VR_STAT_SECTIONS=1
for c in "c2 1024 128" "c4:512 1024 32" "c5full 2048 32" "c3 1024 128"; do python tests/tools_sched_stats.py $c 2>&1 | grep -E "sections|Msamples|unaccounted"; done
