import sys, json, time
import os; sys.path[:0]=[os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__))]
import numpy as np, scenes, volren_amd as va
cfg = sys.argv[1] if len(sys.argv)>1 else "c2"
w=h=int(sys.argv[2]) if len(sys.argv)>2 else 512
spp=int(sys.argv[3]) if len(sys.argv)>3 else 64
thr = [int(x) for x in sys.argv[4].split(",")] if len(sys.argv)>4 else None
r = scenes.hip_scene(cfg,w,h)
if thr: r.set_sched(thr+[0]*(8-len(thr)))
r.render(spp); r.reset()
r.sched_stats(True)
r.render(spp); ms=r.last_kernel_ms()
st = r.sched_stats(False, read=True)
print(cfg,w,h,spp,"thr",thr,"ms %.2f  Msamples/s %.1f"%(ms, w*h*spp/ms/1e3))
if os.environ.get("VR_STATS_JSON"):       # for tests/tools_issue_budget.py: executions per state and iterations of the instrumented launch
    json.dump({"config": cfg, "width": w, "height": h, "spp": spp, "samples": w*h*spp, "iterations": st["iterations"], "resumes": st["resumes"], "parks": st["parks"], "waves": st["waves"], "ms": ms,
               **{k: list(st[k]) for k in va.renderer.STATE_NAMES}}, open(os.environ["VR_STATS_JSON"], "w"))
ns = w*h*spp
tot_exec=0
for k in va.renderer.STATE_NAMES:
    e,l = st[k]; tot_exec+=e
    cyc = st["cycles"][k]
    print("  %-8s exec %10d  lanes/exec %5.1f  lane-steps/sample %6.2f  exec/wave-sample %6.2f  cyc/exec %7.0f  share %5.1f%%"%(k,e,l/max(e,1),l/ns,e/(ns/64), cyc/max(e,1), 100.0*cyc/max(st["wave_cycles"],1)))
print("  wave lifetime cycles/wave-sample %.0f ; unaccounted (scheduler) %.1f%%"%(st["wave_cycles"]/(ns/64), 100.0*(1-sum(st["cycles"].values())/max(st["wave_cycles"],1))))
print("  pool occupancy per iteration:", {k: round(v,1) for k,v in st.get("occupancy",{}).items()})
print("  avg resident waves (at 2.4 GHz) %.0f  (waves launched %d)"%(st["wave_cycles"]/(ms*2.4e6), st["waves"]))
print("  iterations/wave-sample %.1f  blocks/iter %.2f"%(st["iterations"]/(ns/64), tot_exec/st["iterations"]))
if os.environ.get("VR_STAT_SECTIONS"):      # library built with -DVR_STAT_SCHED=1: the occupancy slots carry section cycles
    sec = {k: v * st["iterations"] for k, v in st["occupancy"].items()}
    names = {"marching": "resume", "ready": "hot pair", "nee": "park", "postnee": "decision + event batches", "escape": "loop tail/head", "free": "-"}
    ev = sum(st["cycles"][k] for k in ("new", "begin", "nee", "postnee", "escape"))
    print("  sections (share of wave lifetime):", {names[k]: round(100.0 * v / st["wave_cycles"], 1) for k, v in sec.items() if k != "free"},
          " of which event code %.1f" % (100.0 * ev / st["wave_cycles"]))
