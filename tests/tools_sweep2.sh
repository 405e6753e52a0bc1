# diagnostic: threshold sweep timed with the production (non-stats) kernel; usage: bash tests/tools_sweep2.sh
for t in "64,0,32,2,48,48,64" "64,0,32,1,48,48,64" "64,0,32,3,48,48,64" "64,0,32,4,48,48,64" "32,0,32,2,48,48,32" "32,0,48,3,48,48,32"; do python - $t <<'PY'
import sys, os
sys.path[:0]=[os.getcwd(), os.path.join(os.getcwd(),"tests")]
import scenes, volren_amd as va
thr=[int(x) for x in sys.argv[1].split(",")]
r = scenes.hip_scene("c2",1024,1024)
va.set_sched(thr+[0])
r.render(64); r.reset(); r.render(64); a=r.last_kernel_ms(); r.reset(); r.render(64); b=r.last_kernel_ms()
print(thr, "ms %.2f %.2f  Msamples/s %.0f"%(a,b,1024*1024*64/min(a,b)/1e3))
PY
done 2>&1 | grep Msamples
