"""CPU suite: tools/isa_hazards.py over every gfx950 code object of the shipped library.

Three things hipcc does not guard and a GPU test catches once in 10^4 pixels, or not at all:
  * the gfx950 store-data hazard: a > 64-bit buffer store with an SGPR soffset followed directly by a VALU write of its data
    registers (csrc/color_kernels.hip pads it with `s_nop 1`; LLVM's recognizer pads only the non-register soffset);
  * a full `s_waitcnt vmcnt(0)` between an untracked LDS-DMA fill and the blend that is meant to run under it (c2b86c5), or a
    fill issued through the compiler's builtin in a kernel designed around untracked fills;
  * spills inside a loop of a shipped instantiation.
The rules are exercised on hand-written listings, on the library itself (must be clean) and on color_kernels.hip rebuilt
WITHOUT the pad (must be flagged)."""
import os
import shutil
import subprocess
import sys

import pytest

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_hazards as ih  # noqa: E402

LIB = os.path.join(ROOT, "discorpy_amd", "lib", "libdiscorpy_hip.so")
needs_llvm = pytest.mark.skipif(not os.path.exists(os.path.join(ih.LLVM, "llvm-objdump")), reason="ROCm LLVM tools not installed")


def _ins(lines, start=0x1000):
    """[(address, mnemonic, operands, text)] from assembly lines, four bytes apart (branch targets as `-> +0xOFFSET`)."""
    out = []
    for i, ln in enumerate(lines):
        parts = ln.split(None, 1)
        ops = [o.strip() for o in ih.re.split(r",\s*(?![^\[]*\])", parts[1].split(" -> ")[0])] if len(parts) > 1 else []
        out.append((start + 4 * i, parts[0], ops, ln))
    return out


def test_store_data_rule_on_listings():
    wide = "buffer_store_dwordx3 v[22:24], v2, s[12:15], s16 offen nt"
    # the pair as hipcc emitted it: one unrelated instruction in between is ONE wait state -- the SGPR-soffset case needs two
    assert ih.scan_store_hazard("k", _ins([wide, "s_add_i32 s0, s0, 1", "v_cvt_i32_f32_e32 v22, v8"]))
    assert ih.scan_store_hazard("k", _ins([wide, "v_cvt_i32_f32_e32 v23, v8"]))
    # padded as color_kernels.hip does (s_nop 1 = two wait states), or the writer far enough away, or other registers
    assert not ih.scan_store_hazard("k", _ins([wide, "s_nop 1", "v_cvt_i32_f32_e32 v22, v8"]))
    assert not ih.scan_store_hazard("k", _ins([wide, "s_add_i32 s0, s0, 1", "s_add_i32 s1, s1, 1", "v_cvt_i32_f32_e32 v22, v8"]))
    assert not ih.scan_store_hazard("k", _ins([wide, "v_cvt_i32_f32_e32 v25, v8", "v_mov_b32_e32 v21, v8"]))
    # a 64-bit store has no such hazard; a compare writes no VGPR
    assert not ih.scan_store_hazard("k", _ins(["buffer_store_dwordx2 v[22:23], v2, s[12:15], s16 offen", "v_mov_b32_e32 v22, v8"]))
    assert not ih.scan_store_hazard("k", _ins([wide, "v_cmp_eq_f32_e32 vcc, 1.0, v22"]))
    # soffset not a register: the ISA's own rule, one wait state (what LLVM pads by itself)
    imm = "buffer_store_dwordx4 v[4:7], v2, s[12:15], 0 offen"
    assert ih.scan_store_hazard("k", _ins([imm, "v_mov_b32_e32 v5, v8"]))
    assert not ih.scan_store_hazard("k", _ins([imm, "s_nop 0", "v_mov_b32_e32 v5, v8"]))
    assert ih.scan_store_hazard("k", _ins(["global_store_dwordx4 v[0:1], v[4:7], off", "v_mov_b32_e32 v7, v8"]))


UNTRACKED = ["s_mov_b32 s9, m0", "s_mov_b32 m0, s4", "s_nop 0", "buffer_load_dwordx4 v11, s[0:3], 0 offen lds", "s_mov_b32 m0, s9"]


def test_dma_wait_rule_on_listings():
    name = "_ZN3dcp15remap_wg_kernelILi0ELi5ELi2EfEEvNS_9ImageArgsENS_7MapArgsE"
    blend = ["ds_read2_b32 v[36:37], v17 offset1:1", "v_fma_f64 v[0:1], v[2:3], v[4:5], v[6:7]"]
    # fill, then the blend of the OTHER slab, then the explicit wait + barrier: clean
    ok, fills = ih.scan_dma_wait(name, _ins(UNTRACKED + blend + ["s_waitcnt vmcnt(0)", "s_barrier"] + blend))
    assert not ok and fills == 1
    # a partial wait (the stack kernel's lazy one) is not a full wait
    assert not ih.scan_dma_wait(name, _ins(UNTRACKED + ["s_waitcnt vmcnt(16)"] + blend + ["s_barrier"]))[0]
    # the regression: a full vector-memory wait in front of the blend that was meant to run under the fill
    bad, _ = ih.scan_dma_wait(name, _ins(UNTRACKED + ["s_waitcnt vmcnt(0)"] + blend + ["s_barrier"]))
    assert [f["kind"] for f in bad] == ["dma-wait"]
    # ... which is what a spill reload brings with it
    bad, _ = ih.scan_dma_wait(name, _ins(UNTRACKED + ["scratch_load_dwordx2 v[0:1], off, off", "s_waitcnt vmcnt(0)"] + blend + ["s_barrier"]))
    assert bad
    # a fill through the compiler's builtin in a kernel built around untracked fills
    tracked = ["s_add_i32 m0, s97, s19", "s_mov_b32 s0, s25", "buffer_load_dwordx4 v11, s[0:3], 0 offen lds", "s_or_b64 exec, exec, s[12:13]"]
    bad, _ = ih.scan_dma_wait(name, _ins(tracked + blend + ["s_barrier"]))
    assert [f["kind"] for f in bad] == ["tracked-fill"]
    # the float32 stack kernel keeps the builtin on purpose; its uint16 twin must not
    f32 = "_ZN3dcp15stack_wg_kernelILi5ELi2EfEEvNS_9StackArgsENS_7MapArgsE"
    u16 = "_ZN3dcp15stack_wg_kernelILi5ELi1EtEEvNS_9StackArgsENS_7MapArgsE"
    assert not ih.scan_dma_wait(f32, _ins(tracked + ["s_waitcnt vmcnt(0)"] + blend + ["s_barrier"]))[0]
    assert ih.scan_dma_wait(u16, _ins(tracked + blend + ["s_barrier"]))[0]
    # other kernels are not this rule's business
    assert ih.scan_dma_wait("_ZN3dcp16spline_wg_kernelILi0ELi3ELi5ELb1EEEvNS_10SplineArgsE", _ins(tracked + ["s_waitcnt vmcnt(0)"] + blend)) == ([], 0)


def test_scratch_rule_tells_loops_from_prologues():
    loop = ["v_mov_b32_e32 v0, 0", "scratch_load_dword v1, off, off offset:16", "v_add_u32_e32 v0, v0, v1", "s_cbranch_scc1 65532 -> +0x4"]
    found, notes = ih.scan_scratch("k", _ins(loop), 20)
    assert found and found[0]["scratch_instructions_in_loops"] == 1 and not notes
    prologue = ["scratch_store_dword off, v1, off", "v_mov_b32_e32 v0, 0", "v_add_u32_e32 v0, v0, v1", "s_cbranch_scc1 65534 -> +0x4", "s_endpgm"]
    found, notes = ih.scan_scratch("k", _ins(prologue), 4)
    assert not found and notes and notes[0]["scratch_instructions"] == 1
    assert ih.scan_scratch("k", _ins(["v_mov_b32_e32 v0, 0", "s_endpgm"]), 0) == ([], [])


@needs_llvm
def test_shipped_library_is_clean():
    assert os.path.exists(LIB), "build the library first (__graft_entry__.build())"
    rep = ih.scan_library(LIB)
    assert rep["code_objects"] >= 4 and rep["kernels"] > 300 and rep["instructions"] > 500000
    assert rep["wide_stores_sgpr_soffset"] > 0 and rep["lds_dma_fills_in_streaming_kernels"] > 0      # the scan saw what it is about
    assert rep["findings"] == [], ih.summary(rep)
    # spills outside loops are tolerated but must stay what they are today: the fallback branch of the integer stack kernels
    assert all("stack_wg_kernel" in n["kernel"] and n["private_segment_fixed_size"] <= 32 for n in rep["notes"]), ih.summary(rep)


@needs_llvm
def test_colour_kernels_without_the_pad_are_flagged(tmp_path):
    """DCP_WIDE_STORE_PAD_ON=0 reproduces the code that corrupted ~1 of 10^4 stores in round 4: the scan must see it."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not installed")
    obj = str(tmp_path / "color_nopad.o")
    src = os.path.join(ROOT, "discorpy_amd", "csrc", "color_kernels.hip")
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fvisibility=hidden",
                        "-DDCP_WIDE_STORE_PAD_ON=0", "--cuda-device-only", "-c", src, "-o", obj], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    rep = ih.scan_library(obj)
    bad = [f for f in rep["findings"] if f["kind"] == "store-data"]
    assert bad, ih.summary(rep)
    assert all("remap_wg_color_kernel" in f["kernel"] and f["needed"] == 2 and "buffer_store_dwordx" in f["store"] for f in bad)
