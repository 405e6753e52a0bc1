#!/bin/bash
# Round 6, last GPU call: what the driver runs at round end -- the GPU suite, smoke(), the bench command -- on the final tree, and the bench line once more under
# rocprofv3 --kernel-trace --stats (profiles/r6_bench.json, r6_bench_kernel_stats.csv)
set -o pipefail
O=gpurun_out/r6z; mkdir -p $O; rm -f $O/summary.txt
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" | tee -a $O/summary.txt
tail -n 4 $O/pytest.log | tee -a $O/summary.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?" | tee -a $O/summary.txt
grep -E "smoke" $O/smoke.log | tail -3 | tee -a $O/summary.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc $?" | tee -a $O/summary.txt
grep "^{" $O/bench.json | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('value', d['value'], 'ms_per_step', d['ms_per_step'], 'frac', d['roofline']['frac'], 'traffic stale', d['roofline'].get('traffic_source',{}).get('stale'), 'crc', d['frame_crc32'], 'fast', d['fast_math']['speedup'])
for c in d['configs']: print(c['name'], c.get('value'), c.get('roofline',{}).get('frac'))
" | tee -a $O/summary.txt
bash tests/tools_collect_profiles.sh bench 2>&1 | tail -2
