cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export AB_CASES="c2:1024:256 c4:512:1024:64 c5full:2048:64 c5cloud:2048:64"
bash tests/tools_ab.sh default majb snt 2>&1 | grep -v "^load\|Preparing\|Loading" > gpurun_out/r4d_majb_snt.log
echo "--- VR_SPU (samples per work unit; default 8, dense kernel 4)" >> gpurun_out/r4d_majb_snt.log
AB_CASES="c5cloud:2048:64 c5full:2048:64" bash tests/tools_ab_env.sh VR_SPU=4 VR_SPU=8 VR_SPU=16 2>&1 | grep -v "^load\|Preparing\|Loading" >> gpurun_out/r4d_majb_snt.log
cat gpurun_out/r4d_majb_snt.log
