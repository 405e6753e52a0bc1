#!/bin/bash
# Diagnostic: A/B timing of library variants on the GPU box (same box, back to back, two rounds so that drift shows).
#   usage: bash tests/tools_ab.sh <variant> [<variant> ...]     variant = "default" (volren_amd/libvolren_amd.so) or the <name> of
#   build/exp_<name>/libvolren_amd.so (tests/tools_build_variant.sh).  Every variant first renders the smoke scenes against the oracle.
set -o pipefail
CASES=${AB_CASES:-"c2:1024:256 c3:1024:256 c4:512:1024:64"}
for round in 1 2; do
for v in "$@"; do
  if [ "$v" = default ]; then unset VOLREN_AMD_LIB; else export VOLREN_AMD_LIB=$PWD/build/exp_$v/libvolren_amd.so; fi
  if [ $round = 1 ]; then python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -E "smoke ok|Error|error" || { echo "== $v: smoke FAILED"; continue; }; fi
  for c in $CASES; do
    cfg=${c%:*:*}; rest=${c#$cfg:}; size=${rest%:*}; spp=${rest#*:}
    timeout -k 10 120 python tests/tools_profile_run.py $cfg $size $spp 2>&1 | grep "kernel ms" | sed "s|^|== $v $cfg $size $spp: |"
  done
done
done
