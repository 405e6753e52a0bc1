#!/bin/bash
# Round 6, GPU call 6: final profiles, part 1 -- the bench line under rocprofv3 and the PMC passes of c2, c3, c4 (final kernels)
set -o pipefail
bash tests/tools_collect_profiles.sh bench 2>&1 | tail -3
for spec in "c2 c2 c2 1024x1024 512" "c3 c3 c3 1024x1024 512" "c4 c4 c4:512 1024x1024 256" "c4_1080p c4@1920x1080x4096 c4:512 1920x1080 128"; do
  PMC_ONLY="$spec" bash tests/tools_collect_profiles.sh pmc 2>&1 | grep -E "^pmc|failed"
  set -- $spec; cp gpurun_out/prof/pmc_specs.json gpurun_out/prof/pmc_specs_$1.json
done
