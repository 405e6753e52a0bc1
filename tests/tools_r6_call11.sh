#!/bin/bash
# Round 6, GPU call 11: the tolerance mode with and without the world slot (the bench line under rocprofv3 had it 6 % SLOWER than the exact kernels)
O=gpurun_out/r6k; mkdir -p $O
for round in 1 2; do
python tests/tools_fast_math.py c2 1024 256 2>&1 | grep "spp:" | sed "s|^|== default: |" | tee -a $O/fast.txt
VOLREN_AMD_LIB=$PWD/build/exp_fws0/libvolren_amd.so python tests/tools_fast_math.py c2 1024 256 2>&1 | grep "spp:" | sed "s|^|== tolerance kernel without the world slot: |" | tee -a $O/fast.txt
done
python tests/tools_fast_math.py c2 1024 1024 2>&1 | grep "spp:" | sed "s|^|== default, 1024 spp: |" | tee -a $O/fast.txt
