#!/bin/bash
# Diagnostic: builds build/exp_<name>/libvolren_amd.so with extra compiler flags for the path-tracing kernel
# (e.g. -DVR_WAVES_PER_SIMD=5 -DVR_NSLOT=126); run with VOLREN_AMD_LIB=build/exp_<name>/libvolren_amd.so.
#   usage: [VARIANTS="0 1"] bash tests/tools_build_variant.sh <name> <flags...>
#   VARIANTS: the kernel variants to recompile with the flags (default: all five, bit-exact and tolerance mode); the others, and everything else when VARIANTS is set,
#   are taken from the default build in build/ (run `make` first)
set -e
name=$1; shift
out=build/exp_$name; mkdir -p $out
if [ -n "$VARIANTS" ]; then
  FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-result -Iinclude"
  objs=""
  pids=""
  for v in 0 1 2 3 4; do
    if echo " $VARIANTS " | grep -q " $v "; then
      /opt/rocm/bin/hipcc $FLAGS -fno-slp-vectorize -DVR_PT_VARIANT=$v "$@" -Rpass-analysis=kernel-resource-usage -c volren_amd/csrc/vr_pathtrace.hip -o $out/vr_pathtrace_$v.o 2> $out/res_$v.txt &
      pids="$pids $!"
      objs="$objs $out/vr_pathtrace_$v.o"
    else
      objs="$objs build/vr_pathtrace_$v.o"
    fi
    objs="$objs build/vr_ptfast_$v.o"
  done
  for p in $pids; do wait $p; done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $out/libvolren_amd.so build/vr_kernels.o $objs build/grids.o build/imageio.o build/environment.o build/transferfunc.o build/renderer.o build/sharded.o build/capi.o -lz -ldl
  for v in $VARIANTS; do grep -h -A8 "pathtrace_kernel" $out/res_$v.txt | grep -E "Function Name|VGPRs:|SGPRs Spill|VGPRs Spill|ScratchSize" | sed 's/.*remark: [^ ]* *//; s/ \[-R.*//' | paste - - - - - | sed "s/^/v$v: /"; done
  exit 0
fi
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-result -Iinclude"
pids=""
for v in 0 1 2 3 4; do
  /opt/rocm/bin/hipcc $FLAGS -fno-slp-vectorize -DVR_PT_VARIANT=$v "$@" -Rpass-analysis=kernel-resource-usage -c volren_amd/csrc/vr_pathtrace.hip -o $out/vr_pathtrace_$v.o 2> $out/res_$v.txt &
  pids="$pids $!"
done
/opt/rocm/bin/hipcc $FLAGS "$@" -c volren_amd/csrc/vr_kernels.hip -o $out/vr_kernels.o 2>/dev/null &
pids="$pids $!"
/opt/rocm/bin/hipcc $FLAGS "$@" -x hip -c volren_amd/csrc/renderer.cpp -o $out/renderer.o 2>/dev/null &
pids="$pids $!"
/opt/rocm/bin/hipcc $FLAGS "$@" -x hip -c volren_amd/csrc/environment.cpp -o $out/environment.o 2>/dev/null &
pids="$pids $!"
for v in 0 1 2 3 4; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -fno-hip-fp32-correctly-rounded-divide-sqrt -Wno-unused-result -Iinclude -DVR_FAST_MATH=1 -fno-slp-vectorize -DVR_PT_VARIANT=$v "$@" -c volren_amd/csrc/vr_pathtrace.hip -o $out/vr_pathtrace_fast_$v.o 2>/dev/null &
  pids="$pids $!"
done
for p in $pids; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $out/libvolren_amd.so $out/vr_kernels.o $out/vr_pathtrace_0.o $out/vr_pathtrace_1.o $out/vr_pathtrace_2.o $out/vr_pathtrace_3.o $out/vr_pathtrace_4.o $out/vr_pathtrace_fast_0.o $out/vr_pathtrace_fast_1.o $out/vr_pathtrace_fast_2.o $out/vr_pathtrace_fast_3.o $out/vr_pathtrace_fast_4.o build/grids.o build/imageio.o $out/environment.o build/transferfunc.o $out/renderer.o build/sharded.o build/capi.o -lz -ldl
grep -h -A8 "TraceCfgILb0ELi0ELi0ELi[01]EEELb0E" $out/res_0.txt $out/res_1.txt | grep -E "VGPRs:|ScratchSize|Occupancy|LDS" | sed 's/.*remark: [^ ]* *//; s/ \[-R.*//' | paste - - - - 
