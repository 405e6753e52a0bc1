cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export AB_CASES="c2:1024:256 c3:1024:256 c5full:2048:64 c5cloud:2048:64"
bash tests/tools_ab.sh hdr0 default 2>&1 | grep -v "^load\|Preparing\|Loading" > gpurun_out/r4c_brick_headers.log
cat gpurun_out/r4c_brick_headers.log
bash tests/tools_traffic_quick.sh c5cloud 2048 64 2>&1 | tail -2
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "render_matches_oracle or emission or synthetic or odd_brick or encoder or c5_cloud or transfer_function or glsl or animation or dense_and_raw" 2>&1 | tail -5
