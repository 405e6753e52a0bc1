set -o pipefail
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "emission or c5 or synthetic_baseline or global_majorant or scheduler or stale_cold or flags or grid_frames" > gpurun_out/r3q_tests.log 2>&1; echo "pytest rc $?"; tail -2 gpurun_out/r3q_tests.log
AB_CASES="c5full:2048:64 c5:512:1024:64" bash tests/tools_ab.sh default noempt > gpurun_out/r3q_ab.log 2>&1
cat gpurun_out/r3q_ab.log
