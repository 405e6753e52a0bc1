set -o pipefail
python -m pytest tests -x -q -m gpu > gpurun_out/r3l_tests.log 2>&1; rc=$?; echo "pytest rc $rc"; tail -3 gpurun_out/r3l_tests.log
[ $rc = 0 ] || exit $rc
AB_CASES="c2:1024:256 c3:1024:256 c5full:2048:64" bash tests/tools_ab.sh default coldmem > gpurun_out/r3l_ab.log 2>&1
cat gpurun_out/r3l_ab.log
