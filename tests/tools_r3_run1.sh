set -o pipefail
L=$PWD/build
for round in 1 2; do
for p in 128 16; do
  VOLREN_AMD_LIB=$L/exp_coldmem/libvolren_amd.so python tests/tools_whatif_wrap.py $p 1024 64 2>&1 | grep period | sed "s|^|== stored   |"
  VOLREN_AMD_LIB=$L/exp_wrap$p/libvolren_amd.so python tests/tools_whatif_wrap.py $p 1024 64 2>&1 | grep period | sed "s|^|== wrapped  |"
done
done > gpurun_out/r3d_whatif_wrap.log 2>&1
cat gpurun_out/r3d_whatif_wrap.log
