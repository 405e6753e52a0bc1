for round in 1 2; do
for c in "c2 1024 256" "c4:512 1024 64" "c3 1024 256"; do
  for t in "64,0,56,0,60,60,64" "64,0,56,0,64,64,64" "64,0,48,0,64,64,64" "64,0,60,0,62,62,64" "64,0,40,0,64,64,64" "48,0,56,0,60,60,64" "64,0,64,0,64,64,64"; do
    timeout -k 10 120 python tests/tools_profile_run.py $c "$t,0" 2>&1 | grep "kernel ms" | sed "s|^|== $c thr $t: |"
  done
done
done > gpurun_out/r3u_thr_sweep.log 2>&1
cat gpurun_out/r3u_thr_sweep.log
