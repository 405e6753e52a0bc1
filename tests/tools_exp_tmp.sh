cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export AB_CASES="c2:1024:256 c5full:2048:64 c5cloud:2048:64"
bash tests/tools_ab.sh default xrng 2>&1 | grep -v "^load\|Preparing\|Loading" > gpurun_out/r4c_extra_gather.log
cat gpurun_out/r4c_extra_gather.log
python bench.py > gpurun_out/bench_r4_first.json 2> gpurun_out/bench_r4_first.err; echo "bench rc $?"; tail -c 1500 gpurun_out/bench_r4_first.json; tail -5 gpurun_out/bench_r4_first.err
