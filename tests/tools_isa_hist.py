"""Diagnostic: instruction histogram of the path-tracing kernels from the device assembly (build/asm/vr_kernels.s)."""
import re, collections, sys
txt = open(sys.argv[1] if len(sys.argv) > 1 else "build/asm/vr_kernels.s").read()
for m in re.finditer(r"\n(_ZN2vr16pathtrace_kernelINS_8TraceCfgILb([01])E[A-Za-z0-9]*EELb0E[^\n:]*):[^\n]*\n", txt):
    body = txt[m.end():]
    body = body[:body.index(".Lfunc_end")]
    ops = collections.Counter()
    for l in body.split("\n"):
        mm = re.match(r"\s+([a-z][a-z0-9_]+)\s", l + " ")
        if mm and not mm.group(1).startswith("."):
            ops[mm.group(1)] += 1
    valu = sum(v for k, v in ops.items() if k.startswith("v_"))
    print("USE_TF=%s total %d valu %d salu %d pk %d scratch %d lds %d vmem %d" % (
        m.group(2), sum(ops.values()), valu, sum(v for k, v in ops.items() if k.startswith("s_")),
        sum(v for k, v in ops.items() if k.startswith("v_pk")), sum(v for k, v in ops.items() if k.startswith("scratch")),
        sum(v for k, v in ops.items() if k.startswith("ds_")), sum(v for k, v in ops.items() if k.startswith(("global_", "buffer_", "flat_")))))
    print(" ", ops.most_common(45))
