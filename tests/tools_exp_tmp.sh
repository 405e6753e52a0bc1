cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export AB_CASES="c2:1024:256 c4:512:1024:64 c5full:2048:64 c5cloud:2048:64"
bash tests/tools_ab.sh default w5 2>&1 | grep -v "^load\|Preparing\|Loading" > gpurun_out/r4e_five_waves.log
cat gpurun_out/r4e_five_waves.log
