#!/bin/bash
# diagnostic: A/B of run-time environment settings with the default library; usage: tools_ab_env.sh "VAR=val" ["VAR=val" ...]
CASES=${AB_CASES:-"c2:1024:256 c4:512:1024:64"}
for round in 1 2; do
for e in "$@"; do
  for c in $CASES; do
    cfg=${c%:*:*}; rest=${c#$cfg:}; size=${rest%:*}; spp=${rest#*:}
    env $e timeout -k 10 120 python tests/tools_profile_run.py $cfg $size $spp 2>&1 | grep "kernel ms" | sed "s|^|== [$e] $cfg $size $spp: |"
  done
done
done
