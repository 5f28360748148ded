"""The static instruction budget of the MFMA loops (tools/isa_audit.py on the in-tree libvinet_hip.so; no GPU needed: the code
objects are disassembled with llvm-objdump).  Round 4 measured these loops ISSUE-bound: every non-MFMA instruction per MFMA beyond
~2 costs time (profiles/r4_wrs_phases.txt), so the mix is held here as a regression gate -- a source or toolchain change that bloats
a K loop fails the CPU suite instead of showing up as a slower step a round later.  Baseline: profiles/r5_isa_audit_baseline.txt."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "vinet_amd", "libvinet_hip.so")

spec = importlib.util.spec_from_file_location("isa_audit", os.path.join(ROOT, "tools", "isa_audit.py"))
isa_audit = importlib.util.module_from_spec(spec)
spec.loader.exec_module(isa_audit)

# (substring of the demangled name, MFMAs in the K loop, ceiling on non-MFMA : MFMA) -- measured 2.65 / 2.67 / 3.19 / 2.51 / 1.88
BUDGET = [
    ("conv_ht_kernel<6, 16, 3, false, false, false, false>", 48, 2.8),
    ("conv_ht_kernel<6, 32, 3, false, false, false, false>", 48, 2.8),
    ("conv_wgrad_rs4_kernel<3, 48>", None, 3.35),
    ("conv_wgrad_rs4_kernel<3, 96>", None, 2.65),
    ("conv_dma_kernel<4, 8, 4, 1, 3, false, false>", 32, 2.0),
]


@pytest.mark.skipif(not (os.path.exists(LIB) and os.path.exists(isa_audit.OBJDUMP)), reason="needs the built library and llvm-objdump")
def test_mfma_loops_keep_their_instruction_budget():
    rows = isa_audit.audit(LIB, ["conv_ht_kernel", "conv_wgrad_rs4_kernel", "conv_dma_kernel"], min_mfma=8)
    assert rows, "no MFMA loops found: the disassembly or the loop detection broke"
    for name, mfma, ceiling in BUDGET:
        mine = [r for r in rows if r["name"].startswith(name)]
        assert mine, "%s: kernel or its K loop not found (%d loops audited)" % (name, len(rows))
        main = max(mine, key=lambda r: r["mfma"])            # the K loop proper (prologue / tail loops hold fewer MFMAs)
        if mfma is not None:
            assert main["mfma"] == mfma, (name, main)
        assert main["non_mfma_per_mfma"] <= ceiling, "%s: %.2f non-MFMA instructions per MFMA (ceiling %.2f): %s" % (
            name, main["non_mfma_per_mfma"], ceiling, {k: main[k] for k in ("n", "mfma", "valu", "salu", "lds", "vmem", "sync", "branch")})
        assert "16x16x32" in main["shape"]


@pytest.mark.skipif(not (os.path.exists(LIB) and os.path.exists(isa_audit.OBJDUMP)), reason="needs the built library and llvm-objdump")
def test_no_packed_fp32_instruction_reads_the_high_half_of_its_second_source():
    """MI355X erratum found in round 5 (csrc/common.h): `v_pk_add_f32 / v_pk_mul_f32 ... op_sel:[x,1]` -- the LOW result
    half takes the HIGH half of src1 -- reads zero in lanes 48..63 now and then while a wave of another kernel issues MFMAs on the same
    SIMD (tools/reduce_race_repro.py --pkvariants: 576...784 wrong of 2e9 results for the three affected selections, 0 for the five
    others; inside a training step it dropped one voxel's term of SoundNet's last-layer bias gradient in 7 % of the launches of
    channel_reduce8_kernel<bf16, 0>, the only kernel of the library that contained the form).  hipcc emits it wherever it allocated a
    register pair in swapped order, so this is a gate on the BUILT library: a kernel that shows up here takes VN_NO_PK_F32."""
    hits = isa_audit.packed_fp32_high_half_reads(LIB)
    assert not hits, "packed fp32 instructions with op_sel:[x,1] (erratum form) in: %s" % hits


def test_erratum_form_matcher_on_known_lines():
    """the operand-selection matcher itself, on lines of llvm-objdump output (affected: the LOW result half reads the HIGH half of the
    second or third source; src0 selections and `op_sel_hi`-only forms are not)"""
    hit = lambda ops: bool((m := isa_audit.PK_F32_SRC1_HI.search(ops)) and (m.group(2) == "1" or m.group(3) == "1"))
    assert hit("v[20:21], v[20:21], v[34:35] op_sel:[0,1] op_sel_hi:[1,0]")
    assert hit("v[2:3], v[6:7], v[8:9] op_sel:[0,1] op_sel_hi:[1,1]")
    assert hit("v[2:3], v[6:7], v[8:9], v[10:11] op_sel:[0,0,1] op_sel_hi:[1,1,0]")
    assert not hit("v[2:3], v[6:7], v[8:9] op_sel:[1,0] op_sel_hi:[0,1]")
    assert not hit("v[2:3], v[6:7], v[8:9] op_sel_hi:[1,0]")
    assert not hit("v[2:3], v[6:7], v[2:3], v[8:9] op_sel:[1,0,0] op_sel_hi:[0,1,1]")
    assert not hit("v[2:3], v[6:7], v[8:9]")
