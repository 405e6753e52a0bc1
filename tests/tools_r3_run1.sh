set -o pipefail
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -s -k "readme_command or unmodified_reference or 1e3_of_reference or tuning_state" > gpurun_out/r3j_tests.log 2>&1; echo "pytest rc $?"
grep -E "PSNR|passed|failed|Error|assert" gpurun_out/r3j_tests.log | head -20
