set -o pipefail
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "render_matches or degenerate or scheduler or random_parameter or stale_cold or synthetic_baseline or transfer_function_lut or global_majorant or flags or odd_brick or emission or dense_fp16 or tuning" > gpurun_out/r3s_tests.log 2>&1; echo "pytest rc $?"; tail -2 gpurun_out/r3s_tests.log
AB_CASES="c2:1024:256 c3:1024:256 c4:512:1024:64 c5full:2048:64" bash tests/tools_ab.sh default wg4 wg16nomaj > gpurun_out/r3s_ab.log 2>&1
cat gpurun_out/r3s_ab.log
