"""Diagnostic: table of the path-tracing kernel instances' resources from build/vr_pathtrace*.resources.txt
(hipcc -Rpass-analysis=kernel-resource-usage, written by `make`).  usage: python tests/tools_kernel_resources.py > profiles/r2_kernel_resources.txt"""
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rows = []
for path in sorted(glob.glob(os.path.join(ROOT, "build", "vr_pathtrace_[0-9].resources.txt")) + glob.glob(os.path.join(ROOT, "build", "vr_ptfast_[0-9].resources.txt"))):
    mode = "fast " if "ptfast" in os.path.basename(path) else "exact"
    cur = None
    for line in open(path):
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            n = re.search(r"TraceCfgILb([01])ELi(\d)ELi(\d)ELi(\d)E(?:Li(\d)E)?EELb([01])E", m.group(1))
            cur = dict(mode=mode, tf=n.group(1), glob=n.group(2), em=n.group(3), dense=n.group(4), majb=n.group(5) or "0", stats=n.group(6)) if n else None
            if cur:
                rows.append(cur)
            continue
        if cur is None:
            continue
        for key, pat in (("vgpr", r"\bVGPRs: (\d+)"), ("sgpr", r"TotalSGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"), ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"),
                         ("sspill", r"SGPRs Spill: (\d+)"), ("vspill", r"VGPRs Spill: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
            m = re.search(pat, line)
            if m:
                cur[key] = int(m.group(1))
print("# hipcc -Rpass-analysis=kernel-resource-usage, current build (make): path-tracing kernel instances")
print("# round 1, pathtrace_kernel<false,false>: 128 VGPRs, 217 SGPR spills, 10 VGPR spills, 44 B scratch, 6460 instructions")
print('# "scratch 36 B" of some rare variants = the register scavenger\'s emergency slot (kernels with SGPR spills); their code contains no scratch instruction')
for r in rows:
    print("%s  TraceCfg<tf=%s, global=%s, emission=%s, dense=%s, majb=%s> stats=%s : VGPRs %3d  SGPRs %3d  scratch %3d B  SGPR spills %3d  VGPR spills %2d  LDS %5d B  waves/SIMD %d" % (
        r["mode"], r["tf"], r["glob"], r["em"], r["dense"], r["majb"], r["stats"], r.get("vgpr", -1), r.get("sgpr", -1), r.get("scratch", -1), r.get("sspill", -1), r.get("vspill", -1), r.get("lds", -1), r.get("occ", -1)))
