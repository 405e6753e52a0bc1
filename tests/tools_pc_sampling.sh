#!/bin/bash
# Diagnostic: rocprofv3 PC sampling of the path-tracing kernel (one configuration), summarised per source section by tests/tools_pc_sections.py.
# usage: bash tests/tools_pc_sampling.sh [cfg] [frame] [spp] [method] [unit] [interval]
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cfg=${1:-c2}; frame=${2:-1024x1024}; spp=${3:-128}; method=${4:-stochastic}; unit=${5:-cycles}; interval=${6:-1048576}
OUT=$ROOT/gpurun_out/pcs
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
tag=${cfg//[:@]/_}_${method}
d=$OUT/$tag
rm -rf $d
ROCPROFILER_PC_SAMPLING_BETA_ENABLED=1 timeout -k 10 300 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method $method --pc-sampling-unit $unit --pc-sampling-interval $interval \
    --kernel-trace -d $d -o out --output-format csv json -- python3 $ROOT/tests/tools_profile_run.py $cfg $frame $spp > $d.log 2>&1
echo "pc sampling $tag rc=$?"
tail -5 $d.log
find $d -type f | head -20
for f in $(find $d -name "*.csv"); do echo "== $f"; wc -l $f; head -3 $f | cut -c1-600; done
