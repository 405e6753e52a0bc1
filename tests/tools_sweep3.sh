# diagnostic: scheduler threshold sweep (run-time parameters) with the production kernels; usage: bash tests/tools_sweep3.sh
# thr = NEW, cap, hungry (low-water mark of live paths), -, NEE, POSTNEE, ESCAPE
python - <<'PY'
import sys, os
sys.path[:0]=[os.getcwd(), os.path.join(os.getcwd(),"tests")]
import scenes
sets = ["64,0,56,2,60,60,64", "64,0,48,2,60,60,64", "64,0,60,2,60,60,64", "64,0,56,2,64,64,64", "64,0,56,2,56,56,64", "64,0,56,2,48,48,64",
        "56,0,56,2,60,60,56", "48,0,56,2,60,60,48", "64,0,56,2,60,60,56", "64,0,64,2,64,64,64", "64,0,40,2,60,60,64", "64,0,56,2,52,60,64", "64,0,56,2,60,52,64"]
for cfg, spp in (("c2", 128), ("c4:512", 32)):
    r = scenes.hip_scene(cfg, 1024, 1024)
    r.render(spp)
    for t in sets:
        thr = [int(x) for x in t.split(",")]
        r.set_sched(thr + [0])
        best = 1e9
        for k in range(2):
            r.reset(); r.render(spp); best = min(best, r.last_kernel_ms())
        print("%s thr %s  ms %.2f  Msamples/s %.0f" % (cfg, t, best, 1024 * 1024 * spp / best / 1e3), flush=True)
PY
