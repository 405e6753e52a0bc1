"""Diagnostic: the path-tracing kernel's device assembly attributed to source regions.

Input: assembly of vr_pathtrace.hip compiled with -gline-tables-only (same code as the production build; `.loc file line` directives name the
source line of every instruction).  Every instruction is attributed to the innermost function of vr_trace.h / scheduler section of
vr_pathtrace.h whose line it carries; instructions of vr_math.h / vr_scene.h / the HIP headers (inlined callees) go to the region of the
last instruction before them that carried a line of those two files inside the same basic block run.  Output: per region, instruction
counts by class and issue cycles by the per-opcode costs of profiles/r2_instruction_costs.txt (wave64, 4 wavefronts per SIMD).

usage: tools_isa_sections.py build/asm/ptg_0.s [kernel-substring] [--blocks]
"""
import collections
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# issue cycles per wave64 instruction with 4 resident wavefronts per SIMD (profiles/r2_instruction_costs.txt); classes by opcode
COST = {"simple": 1.72, "fma": 1.90, "vop3": 2.9, "cvt": 2.53, "cmp": 3.27, "cnd": 2.78, "trans": 5.1, "div": 3.3, "dpp": 2.9, "lane": 3.27,
        "salu": 1.0, "snop": 1.22, "smem": 1.0, "vmem": 4.0, "lds": 4.0, "branch": 1.0, "wait": 1.0, "pk": 2.86, "other": 2.0, "u64": 2.86}


def classify(op):
    if op.startswith("s_nop"):
        return "snop"
    if op.startswith("s_waitcnt") or op.startswith("s_barrier") or op.startswith("s_sleep"):
        return "wait"
    if op.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc", "s_swappc")):
        return "branch"
    if op.startswith(("s_load", "s_buffer_load", "s_memtime", "s_memrealtime", "s_dcache", "s_atc")):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("v_pk_"):
        return "pk"
    if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane", "v_permlane", "v_mbcnt")):
        return "lane"
    if op.startswith("v_cmp") or op.startswith("v_cmpx"):
        return "cmp"
    if op.startswith("v_cndmask"):
        return "cnd"
    if op.startswith(("v_rcp", "v_rsq", "v_sqrt", "v_log", "v_exp", "v_sin", "v_cos")):
        return "trans"
    if op.startswith(("v_div_scale", "v_div_fmas", "v_div_fixup")):
        return "div"
    if op.startswith(("v_cvt", "v_floor", "v_ceil", "v_trunc", "v_rndne", "v_fract", "v_frexp", "v_ldexp")):
        return "cvt"
    if op.startswith(("v_fma_", "v_fmac", "v_fmaak", "v_fmamk")):
        return "fma"
    if op.startswith(("v_lshl_add_u64", "v_mad_u64", "v_mad_i64", "v_lshlrev_b64", "v_lshrrev_b64", "v_ashrrev_i64")):
        return "u64"
    if op.endswith("_e64") or op.startswith(("v_mad_", "v_mul_lo", "v_mul_hi", "v_lshl_add", "v_lshl_or", "v_and_or", "v_or3", "v_add3", "v_xad", "v_bfe", "v_bfi", "v_min3", "v_max3",
                                             "v_med3", "v_add_lshl", "v_alignbit", "v_perm", "v_xor3", "v_mul_u32_u24", "v_mul_i32_i24", "v_sad")):
        return "vop3"
    if op.startswith(("v_lshlrev", "v_lshrrev", "v_ashrrev")):
        return "vop3"            # measured 2.57: nearer the VOP3 group than the 1.7-cycle group
    if op.startswith("v_"):
        return "simple"
    return "other"


def function_ranges(path):
    """(first line, last line, name) of every function / struct method body that starts at column 0 with VR_HD, template or a type, crude but enough"""
    out = []
    lines = open(path).read().split("\n")
    cur = None
    for i, ln in enumerate(lines, 1):
        m = re.match(r"^(?:template <[^>]*>\s*)?(?:VR_HD|__device__ __forceinline__|static|inline|constexpr)\b.*?\b([A-Za-z_][A-Za-z0-9_]*)\s*\(", ln)
        if m and not ln.startswith(" "):
            cur = [i, i, m.group(1)]
            out.append(cur)
        elif cur is not None:
            cur[1] = i
    return [(a, b, n) for a, b, n in out]


def scheduler_sections(path):
    """line ranges of the scheduler loop's sections in vr_pathtrace.h, delimited by its VR_SECTION(k) markers"""
    lines = open(path).read().split("\n")
    marks = [(i, int(re.search(r"VR_SECTION\((\d)\)", ln).group(1))) for i, ln in enumerate(lines, 1) if re.search(r"^\s*VR_SECTION\(\d\);", ln)]
    loop = next(i for i, ln in enumerate(lines, 1) if re.match(r"^\s*for \(;;\) \{", ln))
    names = {0: "sched:resume", 1: "sched:hot-pair glue", 2: "sched:park", 3: "sched:batches+decision"}
    out, prev = [], loop
    for i, k in marks:
        if k == 3:
            # the decision logic, then the four event batches (each with its routing code: VR_ROUTE_B expands at its call line)
            cuts = [(prev, "sched:decision")]
            for key, nm in (("if (want_esc) {", "batch:escape"), ("if (want_post) {", "batch:postnee"), ("if (want_new) {", "batch:new"), ("if (want_nee) {", "batch:nee")):
                cuts.append((next(j for j, ln in enumerate(lines, 1) if prev <= j <= i and ln.strip().startswith(key)), nm))
            cuts.append((next(j for j, ln in enumerate(lines, 1) if prev <= j <= i and ln.startswith("#if !VR_BATCH_REGS") and j > cuts[-1][0]), "sched:decision"))
            for (a, nm), (b, _) in zip(cuts, cuts[1:] + [(i + 1, None)]):
                out.append((a, b - 1, nm))
        else:
            out.append((prev, i, names[k]))
        prev = i + 1
    end = next(i for i, ln in enumerate(lines, 1) if i > prev and "every path of the pool has finished" in ln)
    out.append((prev, end, "sched:tail"))
    out.append((1, loop - 1, "prologue"))
    out.append((end + 1, len(lines), "epilogue"))
    return out


def pathtrace_ranges():
    """vr_pathtrace.h: helper functions above the kernel keep their names, the kernel body is cut into the scheduler's sections"""
    path = os.path.join(ROOT, "volren_amd/csrc/vr_pathtrace.h")
    lines = open(path).read().split("\n")
    kline = next(i for i, ln in enumerate(lines, 1) if ln.startswith("pathtrace_kernel(const KernelArgs A)"))
    funcs = function_ranges(path)
    out = []
    for a, b, n in funcs:
        if a < kline:
            out.append((a, min(b, kline - 1), "helper:" + n))
    return out + [(max(a, kline), b, n) for a, b, n in scheduler_sections(path) if b >= kline]


def region_of(fileno_name, line, tr_ranges, pt_ranges):
    if fileno_name == "vr_trace.h":
        for a, b, n in tr_ranges:
            if a <= line <= b:
                return "trace:" + n
        return "trace:?"
    if fileno_name == "vr_pathtrace.h":
        # helper functions defined above the kernel (shle_park, cold_fetch, HotStore...) -> by function; inside the kernel -> by section
        best = None
        for a, b, n in pt_ranges:
            if a <= line <= b:
                best = n
        return best or "pathtrace:?"
    return None


def main():
    path = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else "TraceCfgILb0E"
    txt = open(path).read()
    files = {int(m.group(1)): os.path.basename(m.group(2)) for m in re.finditer(r'\.file\s+(\d+)\s+"[^"]*"\s+"([^"]+)"', txt)}
    files.update({int(m.group(1)): os.path.basename(m.group(2)) for m in re.finditer(r'\.file\s+(\d+)\s+"([^"]+)"\s+md5', txt)})
    tr = function_ranges(os.path.join(ROOT, "volren_amd/csrc/vr_trace.h"))
    pt_funcs = function_ranges(os.path.join(ROOT, "volren_amd/csrc/vr_pathtrace.h"))
    pt_secs = scheduler_sections(os.path.join(ROOT, "volren_amd/csrc/vr_pathtrace.h"))
    kernel_start = min(a for a, b, n in pt_secs if n == "prologue")
    # helper functions above the kernel keep their names; the kernel body is cut into sections
    pt_ranges = pathtrace_ranges()
    m = None
    for mm in re.finditer(r"\n(_ZN2vr[a-z_0-9]*16pathtrace_kernelINS[^\n:]*):[^\n]*\n", txt):
        if want in mm.group(1) and "Lb0EEEv" in mm.group(1):          # non-STATS instance
            m = mm
            break
    if not m:
        raise SystemExit("kernel not found")
    body = txt[m.end():]
    body = body[:body.index(".Lfunc_end")]
    cur_file, cur_line, region = None, 0, "prologue"
    per = collections.defaultdict(lambda: collections.Counter())
    ops_by_region = collections.defaultdict(lambda: collections.Counter())
    for ln in body.split("\n"):
        mloc = re.match(r"\s+\.loc\s+(\d+)\s+(\d+)", ln)
        if mloc:
            cur_file, cur_line = files.get(int(mloc.group(1)), "?"), int(mloc.group(2))
            r = region_of(cur_file, cur_line, tr, pt_ranges)
            if r:
                region = r
            continue
        mi = re.match(r"\s+([a-z][a-z0-9_]+)\s", ln + " ")
        if not mi or mi.group(1).startswith("."):
            continue
        op = mi.group(1)
        per[region][classify(op)] += 1
        ops_by_region[region][op] += 1
    tot = collections.Counter()
    print("%-34s %6s %6s %6s %6s %6s %6s %6s %6s %8s" % ("region", "instr", "valu", "cmp", "cnd", "div", "trans", "vmem", "lds", "issue cyc"))
    rows = []
    for r, c in per.items():
        n = sum(c.values())
        valu = sum(v for k, v in c.items() if k in ("simple", "fma", "vop3", "cvt", "cmp", "cnd", "trans", "div", "pk", "lane", "u64", "dpp"))
        cyc = sum(COST[k] * v for k, v in c.items())
        rows.append((r, n, valu, c["cmp"], c["cnd"], c["div"], c["trans"], c["vmem"], c["lds"], cyc))
        tot.update(c)
    for row in sorted(rows, key=lambda x: -x[-1]):
        print("%-34s %6d %6d %6d %6d %6d %6d %6d %6d %8.0f" % row)
    n = sum(tot.values())
    print("%-34s %6d %6d %6d %6d %6d %6d %6d %6d %8.0f" % ("TOTAL", n, sum(v for k, v in tot.items() if k in ("simple", "fma", "vop3", "cvt", "cmp", "cnd", "trans", "div", "pk", "lane", "u64", "dpp")),
                                                             tot["cmp"], tot["cnd"], tot["div"], tot["trans"], tot["vmem"], tot["lds"], sum(COST[k] * v for k, v in tot.items())))
    if "--ops" in sys.argv:
        for r, c in ops_by_region.items():
            print(r, c.most_common(25))


if __name__ == "__main__":
    main()
