export AB_CASES="c5full:2048:64"
bash tests/tools_ab_env.sh "VR_SPU=8" "VR_SPU=4" "VR_SPU=2" "VR_SPU=16" 2>&1 | grep "=="
