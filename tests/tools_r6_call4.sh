#!/bin/bash
# Round 6, GPU call 4: the bench line under rocprofv3 --kernel-trace --stats and the PMC passes of c2, c3, c4 (tests/tools_collect_profiles.sh)
set -o pipefail
bash tests/tools_collect_profiles.sh bench 2>&1 | tail -5
for spec in "c2 c2 c2 1024x1024 512" "c3 c3 c3 1024x1024 512" "c4 c4 c4:512 1024x1024 256"; do
  PMC_ONLY="$spec" bash tests/tools_collect_profiles.sh pmc 2>&1 | grep -E "^pmc|failed" 
  set -- $spec; cp gpurun_out/prof/pmc_specs.json gpurun_out/prof/pmc_specs_$1.json
done
