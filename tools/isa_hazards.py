#!/usr/bin/env python
"""Scan every gfx950 code object of libdiscorpy_hip.so for three things the compiler does not guard:

  store-data   a VMEM store of MORE than 64 bits of data (buffer_store_dwordx3 / x4, buffer_store_format_xyz / xyzw, tbuffer_store_*,
               global_ / flat_ / scratch_store_dwordx3 / x4) followed too closely by a VALU instruction that writes one of its data
               registers.  The ISA asks for one wait state when the store's soffset is NOT an SGPR, and LLVM's hazard recognizer
               (GCNHazardRecognizer::createsVALUHazard) pads exactly that case.  On gfx950 the same pair with an SGPR soffset stored
               the NEW register contents in about 1 of 10^4 stores (found in round 4 on remap_wg_color_kernel's buffer_store_dwordx3
               ... sN offen nt); csrc/color_kernels.hip keeps the data registers live across an `s_nop 1` behind such stores.
               Rule here: >= 2 wait states between a wide MUBUF / MTBUF store with an SGPR soffset and a VALU write of its vdata,
               >= 1 for every other wide store.
  dma-wait     remap_wg_kernel / stack_wg_kernel (and the colour kernel built the same way) stream the NEXT source box into one LDS
               slab with untracked LDS-DMA (`buffer_load_dwordx4 ... lds` in an asm statement) while they blend out of the other.  An
               `s_waitcnt vmcnt(0)` between such a fill and the LDS reads of the blend that runs under it serialises the two
               (what SIInsertWaitcnts did until c2b86c5).  Rule here: walking forward from a fill, no `s_waitcnt` with vmcnt(0)
               may be followed by an LDS read (ds_read* / ds_load*) before the next s_barrier -- a full wait is legal only as the
               explicit "wait, barrier" pair in front of the slab's first read.  And (tracked-fill): those kernels must not contain
               a fill issued through the compiler's builtin at all -- hipcc waits for such a fill in front of the next LDS read of
               the wave, whichever slab that read is from.
  scratch      no shipped kernel may spill inside a loop (a scratch_load / scratch_store within the address range of a backward
               branch).  Spills outside every loop are listed as notes with the kernel's .private_segment_fixed_size.

    python tools/isa_hazards.py [path/to/libdiscorpy_hip.so] [--json out.json] [--summary out.txt]

Exit status 1 if anything is flagged.  Uses llvm-objdump / llvm-readelf of the ROCm LLVM (/opt/rocm/lib/llvm/bin); no GPU needed.
"""
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = os.environ.get("DCP_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
DEFAULT_LIB = os.path.join(ROOT, "discorpy_amd", "lib", "libdiscorpy_hip.so")

WIDE_STORE = re.compile(r"^(buffer_store_dwordx[34]|buffer_store_format_xyzw?|tbuffer_store_format_xyzw?|"
                        r"(global|flat|scratch)_store_dwordx[34])$")
VREG = re.compile(r"^v(\d+)$|^v\[(\d+):(\d+)\]$")
DMA_KERNELS = ("remap_wg_kernel", "stack_wg_kernel", "remap_wg_color_kernel", "remap_wg_batch_kernel")


def code_objects(lib, workdir):
    """The gfx950 ELF images of `lib`, extracted into workdir: a shared library / host object carries one bundle per translation
    unit in .hip_fatbin; `hipcc --cuda-device-only -c` writes a bare offload bundle."""
    local = os.path.join(workdir, os.path.basename(lib))
    shutil.copy(lib, local)
    head = open(local, "rb").read(32)
    if head.startswith(b"__CLANG_OFFLOAD_BUNDLE__"):
        out = local + ".gfx950"
        subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                        "--input=" + local, "--output=" + out], check=True, capture_output=True)
        return [out]
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", local], check=True, capture_output=True)
    return sorted(os.path.join(workdir, f) for f in os.listdir(workdir) if "amdgcn-amd-amdhsa--gfx950" in f)


def vregs(tok):
    m = VREG.match(tok)
    if not m:
        return set()
    if m.group(1) is not None:
        return {int(m.group(1))}
    return set(range(int(m.group(2)), int(m.group(3)) + 1))


def parse(obj):
    """{kernel: [(address, mnemonic, [operands], text)]} from llvm-objdump -d."""
    txt = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", obj], check=True, capture_output=True, text=True).stdout
    funcs, cur = {}, None
    for ln in txt.split("\n"):
        m = re.match(r"^([0-9a-f]+) <(.+)>:$", ln)
        if m:
            cur = funcs.setdefault(m.group(2), [])
            continue
        if cur is None or not ln.startswith("\t"):
            continue
        body = ln.split("//")[0].strip()
        addr = None
        ma = re.search(r"//\s*([0-9A-Fa-f]+):", ln)
        if ma:
            addr = int(ma.group(1), 16)
        if not body:
            continue
        parts = body.split(None, 1)
        ops = [o.strip() for o in re.split(r",\s*(?![^\[]*\])", parts[1])] if len(parts) > 1 else []
        if parts[0].startswith(("s_branch", "s_cbranch")):
            mt = re.search(r"<[^>]*\+0x([0-9a-fA-F]+)>\s*$", ln)      # target as symbol + offset
            if mt:
                body += " -> +0x" + mt.group(1)
            elif re.search(r"<[^>+]*>\s*$", ln):
                body += " -> +0x0"
        cur.append((addr, parts[0], ops, body))
    return funcs


def loop_ranges(ins):
    """[(first address, last address)] of every backward branch of a function: the address ranges that are loop bodies."""
    if not ins or ins[0][0] is None:
        return []
    start, out = ins[0][0], []
    for addr, mn, _, body in ins:
        m = re.search(r"-> \+0x([0-9a-f]+)$", body)
        if m and addr is not None and start + int(m.group(1), 16) <= addr:
            out.append((start + int(m.group(1), 16), addr))
    return out


def kernel_meta(obj):
    """{kernel symbol: private_segment_fixed_size} from the code object's metadata note."""
    txt = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", obj], capture_output=True, text=True).stdout
    meta, name = {}, None
    for ln in txt.split("\n"):
        m = re.match(r"\s*\.name:\s+(\S+)", ln)
        if m:
            name = m.group(1)
        m = re.match(r"\s*\.private_segment_fixed_size:\s+(\d+)", ln)
        if m:
            meta[("pending", len(meta))] = int(m.group(1))
        m = re.match(r"\s*\.symbol:\s+(\S+)\.kd", ln)
        if m:
            # the fields of one kernel arrive in alphabetical order: .private_segment_fixed_size before .symbol
            for k in [k for k in meta if isinstance(k, tuple)]:
                meta[m.group(1)] = meta.pop(k)
    return {k: v for k, v in meta.items() if not isinstance(k, tuple)}


def wait_states(mn, ops):
    if mn == "s_nop":
        try:
            return int(ops[0], 0) + 1
        except (ValueError, IndexError):
            return 1
    return 1


def valu_writes(mn, ops):
    """VGPRs a VALU instruction writes (none for compares, readlane / readfirstlane, which write SGPRs / VCC / EXEC)."""
    if not mn.startswith("v_") or mn.startswith(("v_cmp", "v_readlane", "v_readfirstlane", "v_nop")):
        return set()
    w = vregs(ops[0]) if ops else set()
    if mn.startswith("v_swap") and len(ops) > 1:
        w |= vregs(ops[1])
    return w


def store_parts(mn, ops):
    """(vdata registers, soffset-is-an-SGPR) of a wide store."""
    data = set()
    if mn.startswith(("buffer_", "tbuffer_")):
        data = vregs(ops[0]) if ops else set()
        # buffer_store vdata, vaddr|off, srsrc, soffset [modifiers on the last operand]
        soff = ops[3].split()[0] if len(ops) > 3 else "0"
        return data, bool(re.match(r"^(s\d+|m0|ttmp\d+)$", soff))
    # global_store vaddr, vdata, saddr|off ; flat_store vaddr, vdata ; scratch_store vaddr|off, vdata, saddr|off
    data = vregs(ops[1]) if len(ops) > 1 else set()
    return data, False


def scan_store_hazard(name, ins):
    found = []
    for i, (addr, mn, ops, body) in enumerate(ins):
        if not WIDE_STORE.match(mn):
            continue
        data, sgpr_soff = store_parts(mn, ops)
        need = 2 if sgpr_soff else 1
        ws, j = 0, i + 1
        while j < len(ins) and ws < need:
            _, mn2, ops2, body2 = ins[j]
            hit = valu_writes(mn2, ops2) & data
            if hit:
                found.append({"kind": "store-data", "kernel": name, "address": addr, "store": body, "writer": body2,
                              "wait_states_between": ws, "needed": need, "registers": sorted(hit)})
                break
            if mn2.startswith(("s_branch", "s_cbranch", "s_endpgm", "s_setpc")):
                break                        # (the instruction at a branch target starts at least one wait state later; x3/x4 stores in
                                             # this library are never the last instruction before a taken branch into a VALU write)
            ws += wait_states(mn2, ops2)
            j += 1
    return found


def is_lds_dma(mn, body):
    return mn.startswith("buffer_load") and re.search(r"\blds\b", body) is not None


def is_untracked_fill(ins, i):
    """lds_dma16_untracked's asm statement (csrc/dcp_device.h): `s_mov_b32 m0, sA / s_nop 0 / buffer_load_dwordx4 ... lds / s_mov_b32 m0, sB`.
    A fill the compiler issued itself (__builtin_amdgcn_raw_ptr_buffer_load_lds) is tracked: hipcc waits for it in front of the next
    LDS read by construction -- the float32 stack kernel keeps that on purpose (A/B in the source: untracked is 1.5 % slower there)."""
    if not is_lds_dma(ins[i][1], ins[i][3]) or i < 2 or i + 1 >= len(ins):
        return False
    p2, p1, nx = ins[i - 2], ins[i - 1], ins[i + 1]
    return (p1[1] == "s_nop" and p2[1] == "s_mov_b32" and p2[2][:1] == ["m0"] and nx[1] == "s_mov_b32" and nx[2][:1] == ["m0"])


def must_fill_untracked(name):
    """Kernels whose design is "stream the next box under the blend of this one": every LDS-DMA fill in them must be the untracked
    asm statement.  (stack_wg_kernel on float32 / float64 keeps the compiler's builtin on purpose -- measured faster; the colour,
    spline and per-wave-box kernels fill, wait and blend in turn.)"""
    if "remap_wg_kernel" in name or "remap_wg_batch_kernel" in name:
        return True
    m = re.match(r"_ZN3dcp15stack_wg_kernelILi(n?\d+)ELi\d+E([a-z])EEv", name)
    return bool(m) and m.group(2) in "ahstij"


def scan_dma_wait(name, ins):
    if not any(k in name for k in DMA_KERNELS):
        return [], 0
    found, fills, seen = [], 0, set()
    if must_fill_untracked(name):
        tracked = [(a, body) for i, (a, mn, _, body) in enumerate(ins) if is_lds_dma(mn, body) and not is_untracked_fill(ins, i)]
        if tracked:
            found.append({"kind": "tracked-fill", "kernel": name, "count": len(tracked), "first_address": tracked[0][0], "first": tracked[0][1]})
    for i, (addr, mn, ops, body) in enumerate(ins):
        if not is_untracked_fill(ins, i):
            continue
        fills += 1
        waited_at = None
        for j in range(i + 1, len(ins)):
            a2, mn2, ops2, body2 = ins[j]
            if mn2 == "s_barrier" or mn2 == "s_endpgm":
                break
            if is_lds_dma(mn2, body2):
                waited_at = None if waited_at is None else waited_at
            if mn2 == "s_waitcnt" and re.search(r"vmcnt\(0\)", body2):
                waited_at = (a2, body2)
            elif mn2 == "s_waitcnt" and not re.search(r"vmcnt", body2) and re.match(r"s_waitcnt\s+(0x0+|0)$", body2):
                waited_at = (a2, body2)
            if waited_at and re.match(r"ds_(read|load)", mn2):
                if waited_at[0] not in seen:
                    seen.add(waited_at[0])
                    found.append({"kind": "dma-wait", "kernel": name, "fill_address": addr, "fill": body, "wait_address": waited_at[0],
                                  "wait": waited_at[1], "lds_read": body2})
                break
    return found, fills


def scan_scratch(name, ins, priv):
    """Spills.  In a loop (inside the address range of a backward branch) they are findings: a reload is a VMEM operation per
    iteration, and its wait also waits for an untracked fill.  Outside every loop (a value parked across the prologue's branches)
    they cost a few instructions per workgroup and are reported as notes."""
    found, notes = [], []
    loops = loop_ranges(ins)
    hot = [(a, body) for a, mn, _, body in ins if mn.startswith("scratch_") and any(lo <= a <= hi for lo, hi in loops)]
    cold = [(a, body) for a, mn, _, body in ins if mn.startswith("scratch_") and not any(lo <= a <= hi for lo, hi in loops)]
    if hot:
        found.append({"kind": "scratch", "kernel": name, "private_segment_fixed_size": priv, "scratch_instructions_in_loops": len(hot),
                      "first": hot[0][1]})
    if not hot and (cold or priv):
        notes.append({"kind": "scratch-outside-loops", "kernel": name, "private_segment_fixed_size": priv, "scratch_instructions": len(cold)})
    return found, notes


def scan_library(lib):
    report = {"library": os.path.relpath(lib, ROOT) if lib.startswith(ROOT) else lib, "code_objects": 0, "kernels": 0, "instructions": 0,
              "wide_stores": 0, "wide_stores_sgpr_soffset": 0, "lds_dma_fills_in_streaming_kernels": 0, "findings": [], "notes": []}
    with tempfile.TemporaryDirectory() as wd:
        objs = code_objects(lib, wd)
        report["code_objects"] = len(objs)
        for obj in objs:
            meta = kernel_meta(obj)
            for name, ins in parse(obj).items():
                if name.endswith(".kd") or not ins:
                    continue
                report["kernels"] += 1
                report["instructions"] += len(ins)
                for _, mn, ops, _ in ins:
                    if WIDE_STORE.match(mn):
                        report["wide_stores"] += 1
                        report["wide_stores_sgpr_soffset"] += int(store_parts(mn, ops)[1])
                report["findings"] += scan_store_hazard(name, ins)
                f, fills = scan_dma_wait(name, ins)
                report["findings"] += f
                report["lds_dma_fills_in_streaming_kernels"] += fills
                f, notes = scan_scratch(name, ins, meta.get(name, 0))
                report["findings"] += f
                report["notes"] += notes
    return report


def demangle(names):
    try:
        r = subprocess.run([os.path.join(LLVM, "llvm-cxxfilt")] if os.path.exists(os.path.join(LLVM, "llvm-cxxfilt")) else ["c++filt"],  # noqa
                           input="\n".join(names), capture_output=True, text=True)
        out = r.stdout.split("\n")
        return dict(zip(names, out)) if r.returncode == 0 and len(out) >= len(names) else {n: n for n in names}
    except OSError:
        return {n: n for n in names}


def summary(report):
    lines = ["isa_hazards: %s" % report["library"],
             "  %d gfx950 code objects, %d kernels / device functions, %d instructions" % (report["code_objects"], report["kernels"], report["instructions"]),
             "  wide (> 64-bit) VMEM stores: %d, of them with an SGPR soffset: %d" % (report["wide_stores"], report["wide_stores_sgpr_soffset"]),
             "  untracked LDS-DMA fills in the streaming kernels: %d" % report["lds_dma_fills_in_streaming_kernels"],
             "  findings: %d, notes: %d" % (len(report["findings"]), len(report["notes"]))]
    names = demangle(sorted({f["kernel"] for f in report["findings"] + report["notes"]}))
    for f in report["findings"]:
        k = names.get(f["kernel"], f["kernel"])
        if f["kind"] == "store-data":
            lines.append("  [store-data] %s @%x: `%s` then `%s` after %d wait state(s) (need %d), registers v%s" % (
                k, f["address"] or 0, f["store"], f["writer"], f["wait_states_between"], f["needed"], f["registers"]))
        elif f["kind"] == "tracked-fill":
            lines.append("  [tracked-fill] %s: %d LDS-DMA fill(s) the compiler tracks (first @%x `%s`): it will wait for them in front of "
                         "every LDS read" % (k, f["count"], f["first_address"] or 0, f["first"]))
        elif f["kind"] == "dma-wait":
            lines.append("  [dma-wait] %s: fill @%x, `%s` @%x in front of `%s`" % (k, f["fill_address"] or 0, f["wait"], f["wait_address"] or 0, f["lds_read"]))
        else:
            lines.append("  [scratch] %s: %s" % (k, {a: b for a, b in f.items() if a not in ("kind", "kernel")}))
    for f in report["notes"]:
        lines.append("  (note) %s: %d bytes of scratch, %d spill instruction(s), none inside a loop" % (
            names.get(f["kernel"], f["kernel"]).split("(")[0], f["private_segment_fixed_size"], f["scratch_instructions"]))
    return "\n".join(lines)


def main(argv):
    import argparse
    ap = argparse.ArgumentParser(description="gfx950 ISA hazard scan of libdiscorpy_hip.so (see the module docstring)")
    ap.add_argument("library", nargs="?", default=DEFAULT_LIB)
    ap.add_argument("--json", default=None)
    ap.add_argument("--summary", default=None)
    a = ap.parse_args(argv)
    rep = scan_library(a.library)
    text = summary(rep)
    print(text)
    if a.json:
        json.dump(rep, open(a.json, "w"), indent=1)
    if a.summary:
        open(a.summary, "w").write(text + "\n")
    return 1 if rep["findings"] else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
