#!/usr/bin/env python
"""Count gfx950 instructions of one kernel by class between source-line markers of the hipcc --save-temps
assembly ("fp64 instructions per pixel from the ISA").

    python tools/isa_count.py kernel.s [first_line last_line]...

Classes: f64 = VALU ops on doubles (v_*_f64, conversions to/from f64), f32fast = v_add/sub/mul/fma_f32 (the only
VALU ops that issue a wave64 in 2 cycles on CDNA4), valu = every other vector ALU op (4 cycles), trans = v_rsq/rcp/sqrt_f64
(quarter rate: 16 cycles), lds = ds_*, vmem = buffer_/global_ loads and stores, salu = s_* (own issue port).
"""
import sys as _sys
if {"-h", "--help"} & set(_sys.argv[1:]):          # every tool answers --help without touching the GPU
    print(__doc__)
    raise SystemExit(0)
import re
import sys


def classify(op):
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("buffer_", "global_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("v_"):
        if re.match(r"v_(rsq|rcp|sqrt)_f64", op):
            return "trans64"
        if re.match(r"v_(rsq|rcp|sqrt|exp|log|sin|cos)_f32", op):
            return "trans32"
        if "f64" in op:
            return "f64"
        if re.match(r"v_(add|sub|subrev|mul|fma|fmac|mac|mad)_f32", op):
            return "f32fast"
        return "valu"
    return "other"


def main():
    path = sys.argv[1]
    lines = open(path).read().split("\n")
    ranges = [(int(sys.argv[i]), int(sys.argv[i + 1])) for i in range(2, len(sys.argv) - 1, 2)] or [(1, len(lines))]
    for a, b in ranges:
        counts, ops = {}, {}
        for ln in lines[a - 1:b]:
            t = ln.strip()
            if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
                continue
            op = t.split()[0]
            c = classify(op)
            counts[c] = counts.get(c, 0) + 1
            ops[op] = ops.get(op, 0) + 1
        cyc = 4 * (counts.get("f64", 0) + counts.get("valu", 0)) + 2 * counts.get("f32fast", 0) + 16 * counts.get("trans64", 0) \
            + 8 * counts.get("trans32", 0)
        print("lines %d-%d: %s  | VALU issue cycles %d" % (a, b, " ".join("%s=%d" % kv for kv in sorted(counts.items())), cyc))
        print("   " + " ".join("%s:%d" % kv for kv in sorted(ops.items(), key=lambda kv: -kv[1])[:40]))


if __name__ == "__main__":
    main()
