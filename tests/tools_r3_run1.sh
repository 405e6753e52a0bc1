set -o pipefail
python -m pytest tests -x -q -m gpu > gpurun_out/r3m_tests.log 2>&1; rc=$?; echo "pytest rc $rc"; tail -5 gpurun_out/r3m_tests.log
