"""The two analysis tools of round 6 stay runnable: the closed-queueing model of the kernel's time (on the committed measurements) and the L2 breakdown by access class
(the lane code compiled for the host with its access hooks, on a small scene).  CPU only."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_latency_model_reproduces_the_measurements_it_was_not_fitted_to():
    """tests/tools_latency_model.py fits two numbers per configuration to the rates with 3 and 4 wavefronts per SIMD and must then land within 2 points of every
    measured ratio it has not seen (the two idle-cycle paddings, the instruction padding; 6 points for 2 wavefronts per SIMD) and within 5 points of the SIMD's idle share
    (the bar of verdict r5 #1) on c2 and c4 -- from the records committed under profiles/."""
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "tests", "tools_latency_model.py")], cwd=ROOT).decode()
    blocks = re.split(r"^== ", out, flags=re.M)[1:]
    seen = {}
    for b in blocks:
        name = b.split("\n", 1)[0].split()[0]
        rows = re.findall(r"^   (.{68}) +([0-9.]+) +([0-9.]+)   \(([-+0-9.]+) points\)", b, flags=re.M)
        seen[name] = {r[0].strip(): (float(r[1]), float(r[2])) for r in rows}
    assert {"c2", "c4"} <= set(seen), seen.keys()
    for name in ("c2", "c4"):
        rows = seen[name]
        assert len(rows) >= 6, rows
        for label, (measured, model) in rows.items():
            # (the 2-wavefront point is the model's weakest: half the population, a different balance between the stations -- c4's final kernels are 5 points off there)
            tol = 0.05 if "issues no vector instruction" in label else (0.06 if "2 wavefronts" in label else 0.02)
            assert abs(model - measured) <= tol, (name, label, measured, model)


def test_l2_breakdown_runs_on_a_small_scene():
    """tests/tools_l2_breakdown.py on the 64^3 two-grid scene: every class that the scene exercises reports accesses, misses never exceed the accesses that can miss,
    and the two taps of a collision are counted once each (the emission tap exists on camera / scatter segments and is loaded -- from cell 0 -- on shadow ones)."""
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "tests", "tools_l2_breakdown.py"), "c5:64", "128", "2", "2", "2048", "0.25"], cwd=ROOT).decode()
    rows = {}
    for ln in out.split("\n"):
        m = re.match(r"^  (.{30}) +([0-9.]+) +([0-9.]+) +([0-9.]+)", ln)
        if m:
            rows[m.group(1).strip()] = tuple(float(m.group(k)) for k in (2, 3, 4))
    for cls in ("majorant table, levels 0-1", "density tap", "emission tap", "environment warp table", "environment texels", "cold path state, reads", "cold path state, writes", "sample pool, writes"):
        assert cls in rows and rows[cls][0] > 0, (cls, rows)
        acc, newline, miss = rows[cls]
        assert miss <= newline <= acc, (cls, rows[cls])
    assert rows["density tap"][0] == rows["emission tap"][0]
    assert abs(rows["sample pool, writes"][0] - 1.0) < 1e-9
