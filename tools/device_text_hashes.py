#!/usr/bin/env python
"""SHA-256 of the .text section of every gfx950 code object of a library: two builds whose hashes agree contain byte-identical
kernels (what a refactoring of the kernel sources must show).    python tools/device_text_hashes.py [lib.so]"""
import sys as _sys
if {"-h", "--help"} & set(_sys.argv[1:]):          # every tool answers --help without touching the GPU
    print(__doc__)
    raise SystemExit(0)
import hashlib
import os
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import isa_hazards as ih  # noqa: E402


def hashes(lib):
    out = []
    with tempfile.TemporaryDirectory() as wd:
        for obj in ih.code_objects(lib, wd):
            text = obj + ".text"
            subprocess.run([os.path.join(ih.LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.text", obj, text], check=True)
            out.append((os.path.basename(obj).split(".hipv4")[0].split(".")[-1], os.path.getsize(text), hashlib.sha256(open(text, "rb").read()).hexdigest()))
    return out


if __name__ == "__main__":
    for idx, size, h in hashes(sys.argv[1] if len(sys.argv) > 1 else ih.DEFAULT_LIB):
        print("code object %s: .text %d bytes  sha256 %s" % (idx, size, h))
