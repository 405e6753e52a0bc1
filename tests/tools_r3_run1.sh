set -o pipefail
python - <<'PY' > gpurun_out/r3i_defer_check.log 2>&1
import sys, os
sys.path[:0]=[os.getcwd(), os.path.join(os.getcwd(),"tests")]
import numpy as np, scenes
for cfg in ("c2","c3","c4:64","c5:32","c1"):
    r = scenes.hip_scene(cfg, 72, 56); r.render(6); want = r.framebuffer().copy()
    for s in ([64,0,56,24,60,60,64,60],[64,0,56,48,60,60,64,48],[64,0,56,65,60,60,64,32],[8,0,8,40,8,8,8,8],[1,66,1,1,1,1,1,1],[64,0,64,63,64,64,64,64]):
        r.set_sched(s); r.reset(); r.render(6)
        same = np.array_equal(r.framebuffer().view(np.uint32), want.view(np.uint32))
        print(cfg, s, "bit-identical" if same else "DIFFERENT")
PY
cat gpurun_out/r3i_defer_check.log | grep -c "bit-identical"; grep -c DIFFERENT gpurun_out/r3i_defer_check.log
for round in 1 2; do
for c in "c5full 2048 64" "c2 1024 256" "c3 1024 256" "c4:512 1024 64"; do
  for t in "24,0" "24,60" "40,60" "48,60" "56,60" "65,60" "48,48" "32,60"; do
    a=${t%,*}; b=${t#*,}
    timeout -k 10 120 python tests/tools_profile_run.py $c "64,0,56,$a,60,60,64,$b" 2>&1 | grep "kernel ms" | sed "s|^|== $c collide $a batch $b: |"
  done
done
done > gpurun_out/r3i_defer_sweep.log 2>&1
cat gpurun_out/r3i_defer_sweep.log
