set -o pipefail
python -m pytest tests -x -q -m gpu > gpurun_out/r3p_tests.log 2>&1; rc=$?; echo "pytest rc $rc"; tail -3 gpurun_out/r3p_tests.log
[ $rc = 0 ] || exit $rc
bash tests/tools_collect_profiles.sh all > gpurun_out/r3p_collect.log 2>&1; echo "collect rc $?"; tail -12 gpurun_out/r3p_collect.log
