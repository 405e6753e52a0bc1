set -o pipefail
python -m pytest tests -x -q -m gpu > gpurun_out/r3t_tests.log 2>&1; rc=$?; echo "pytest rc $rc"; tail -3 gpurun_out/r3t_tests.log
[ $rc = 0 ] || exit $rc
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
bash tests/tools_collect_profiles.sh all > gpurun_out/r3t_collect.log 2>&1; echo "collect rc $?"; tail -4 gpurun_out/r3t_collect.log | cut -c1-300
