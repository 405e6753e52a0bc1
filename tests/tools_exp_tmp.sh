cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( python tests/tools_wave_timeline.py c2 1024 128 1; python tests/tools_wave_timeline.py c2 1024 1024 8; python tests/tools_wave_timeline.py c4:512 1024 64 1 ) 2>&1 | grep -v "^load\|Preparing\|Loading\|amdgpu" > gpurun_out/r4f_wave_timeline.log
cat gpurun_out/r4f_wave_timeline.log
