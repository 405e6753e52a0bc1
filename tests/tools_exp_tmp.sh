cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/ -q -m gpu > gpurun_out/t_gpu_all.log 2>&1; echo "gpu tests rc $?"; tail -8 gpurun_out/t_gpu_all.log
python tests/tools_variant_throughput.py 2>&1 | grep -v "^load\|Preparing\|Loading\|amdgpu.ids" > gpurun_out/r4_variant_throughput.txt; cat gpurun_out/r4_variant_throughput.txt
