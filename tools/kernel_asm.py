#!/usr/bin/env python
"""Extract one kernel's instructions (comments and directives stripped) from a `hipcc -save-temps` gfx950 assembly file,
and its register / LDS footprint -- to diff the code of a kernel before and after a refactoring, or to feed
tools/isa_count.py.

    python tools/kernel_asm.py file.s <substring of the mangled kernel name> [out.s]
"""
import sys as _sys
if {"-h", "--help"} & set(_sys.argv[1:]):          # every tool answers --help without touching the GPU
    print(__doc__)
    raise SystemExit(0)
import re
import sys


def extract(path, needle):
    lines = open(path).read().split("\n")
    start = None
    for i, ln in enumerate(lines):
        if ln.startswith("_Z") and needle in ln.split(":")[0] and ln.split(":")[0] + ":" == ln.split(";")[0].strip():
            start = i
            break
    if start is None:
        raise SystemExit("no kernel matching %r" % needle)
    name = lines[start].split(":")[0]
    body, meta = [], {}
    for ln in lines[start + 1:]:
        if ln.startswith(".Lfunc_end"):
            break
        t = ln.split(";")[0].rstrip()
        if not t.strip() or t.strip().startswith("."):
            if t.strip().endswith(":"):
                body.append(t.strip())
            continue
        body.append(t.strip())
    for ln in lines[start:]:
        m = re.match(r"\s*\.set %s\.(num_vgpr|num_agpr|numbered_sgpr|private_seg_size), (\d+)" % re.escape(name), ln)
        if m:
            meta[m.group(1)] = int(m.group(2))
        m = re.match(r"; LDSByteSize: (\d+)", ln)
        if m and "num_vgpr" in meta:
            meta["lds_bytes"] = int(m.group(1))
            break
    return name, body, meta


def main():
    name, body, meta = extract(sys.argv[1], sys.argv[2])
    print(name, meta, "%d instructions/labels" % len(body))
    if len(sys.argv) > 3:
        open(sys.argv[3], "w").write("\n".join(body) + "\n")


if __name__ == "__main__":
    main()
