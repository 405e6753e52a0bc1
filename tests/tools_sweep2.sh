# diagnostic: threshold sweep timed with the production (non-stats) kernel; usage: bash tests/tools_sweep2.sh
for t in "64,0,32,2,48,48,64" "48,0,40,2,52,38,36" "48,0,32,2,52,38,36" "40,0,40,2,48,36,32" "56,0,40,2,56,40,40" "48,0,48,2,52,38,36" "48,0,40,3,52,38,36" "48,0,40,1,52,38,36"; do python - $t <<'PY'
import sys, os
sys.path[:0]=[os.getcwd(), os.path.join(os.getcwd(),"tests")]
import scenes, volren_amd as va
thr=[int(x) for x in sys.argv[1].split(",")]
r = scenes.hip_scene("c2",1024,1024)
va.set_sched(thr+[0])
r.render(32); r.reset(); r.render(32); a=r.last_kernel_ms(); r.reset(); r.render(32); b=r.last_kernel_ms()
print(thr, "ms %.2f %.2f  Msamples/s %.0f"%(a,b,1024*1024*32/min(a,b)/1e3))
PY
done 2>&1 | grep Msamples
