set -o pipefail
python -m pytest tests -x -q -m gpu > gpurun_out/r3f_tests.log 2>&1; rc=$?; echo "pytest rc $rc"; tail -5 gpurun_out/r3f_tests.log
[ $rc = 0 ] || exit $rc
( time python bench.py ) > gpurun_out/r3f_bench.json 2> gpurun_out/r3f_bench.err; echo "bench rc $?"; tail -3 gpurun_out/r3f_bench.err
python - <<'PY'
import json
l=[x for x in open("gpurun_out/r3f_bench.json") if x.startswith("{")][-1]
d=json.loads(l)
print(d["value"], d["roofline"]["frac"], d["roofline"]["kernel_ms"], d.get("fast_math",{}).get("value"))
for c in d.get("configs",[]):
    print(c.get("name"), c.get("value"), c.get("roofline",{}).get("frac"), c.get("roofline",{}).get("kernel_ms"), c.get("roofline",{}).get("launches_per_step"), c.get("error"))
PY
