#!/usr/bin/env python3
"""Static audit of the MFMA loops in libvinet_hip.so (runs in the CPU container: no GPU).

    python tools/isa_audit.py [--lib vinet_amd/libvinet_hip.so] [--kernels conv_ht,wgrad_rs4] [--min-mfma 8] [--md]

The library's gfx950 code objects are extracted (llvm-objdump --offloading, in a scratch directory) and disassembled.  For
every kernel, every LOOP (a backward branch to an earlier address inside the function) that contains MFMAs and no inner loop
with MFMAs is a "K loop body"; per body the tool prints the instruction mix

    MFMA | VALU | SALU | LDS (ds_*) | VMEM (global_/buffer_/flat_/scratch_) | waitcnt / barrier / nop | branch

the ratio of non-MFMA to MFMA instructions, and the matrix FLOPs per issued instruction.  Why it matters (round 4's finding,
profiles/r4_wrs_phases.txt): these loops are bound by instruction ISSUE -- a 16x16x32 bf16 MFMA occupies the matrix pipe
for 16 cycles and a wave issues about one instruction per 4-5 cycles, so only ~2 other instructions per MFMA hide in its shadow
at one wave per SIMD (MI355X_MICROARCH.md, "one wave per SIMD ... <= 5 besides the MFMA" for the 32-cycle 32x32x16).  The table
is the static side of that budget; `s_nop` / `s_waitcnt` count as issue slots too.
"""
import argparse
import collections
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
CXXFILT = "c++filt"

MFMA_FLOPS = {  # matrix flops of one instruction (2 * M * N * K)
    "16x16x32": 2 * 16 * 16 * 32, "32x32x16": 2 * 32 * 32 * 16, "16x16x16": 2 * 16 * 16 * 16, "32x32x8": 2 * 32 * 32 * 8,
    "16x16x4": 2 * 16 * 16 * 4, "32x32x2": 2 * 32 * 32 * 2, "4x4x4": 2 * 4 * 4 * 4 * 16,
}


def classify(mn):
    if mn.startswith("v_mfma") or mn.startswith("v_smfmac"):
        return "mfma"
    if mn.startswith("ds_"):
        return "lds"
    if mn.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if mn.startswith(("s_waitcnt", "s_barrier", "s_nop", "s_sleep", "s_setprio", "s_sethalt")):
        return "sync"
    if mn.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc", "s_swappc", "s_call")):
        return "branch"
    if mn.startswith("s_"):
        return "salu"
    if mn.startswith("v_"):
        return "valu"
    return "other"


def mfma_flops(mn):
    m = re.search(r"_(\d+x\d+x\d+)", mn)
    return MFMA_FLOPS.get(m.group(1), 0) if m else 0


def disassemble(lib):
    tmp = tempfile.mkdtemp(prefix="isa_audit_")
    try:
        dst = os.path.join(tmp, "lib.so")
        shutil.copy(lib, dst)
        subprocess.run([OBJDUMP, "--offloading", dst], cwd=tmp, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
        for f in sorted(os.listdir(tmp)):
            if "amdgcn" in f:
                yield subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", os.path.join(tmp, f)], capture_output=True, text=True, check=True).stdout
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def functions(text):
    """[(mangled name, [(address, mnemonic, operands)])]"""
    cur, name = None, None
    for line in text.splitlines():
        m = re.match(r"^([0-9a-f]+) <([^>]+)>:", line)
        if m:
            if cur:
                yield name, cur
            name, cur = m.group(2), []
            continue
        m = re.match(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):(.*)$", line)
        if m and cur is not None:
            cur.append((int(m.group(3), 16), m.group(1), m.group(2) + " " + m.group(4)))
    if cur:
        yield name, cur


def loops(ins):
    """innermost loops that contain MFMAs: (start index, end index) over `ins` (a backward s_cbranch / s_branch to an address
    inside the function closes a loop)"""
    addr2i = {a: i for i, (a, _, _) in enumerate(ins)}
    found = []
    for i, (a, mn, ops) in enumerate(ins):
        if mn.startswith(("s_cbranch", "s_branch")):
            # objdump prints the target as an absolute address in a trailing comment or as <sym+0x..>; recompute from the simm16 when given
            m = re.search(r"<[^>]*\+0x([0-9a-fA-F]+)>", ops)
            tgt = None
            if m:
                base = ins[0][0]
                tgt = base + int(m.group(1), 16)
            else:
                m = re.search(r"(-?\d+)\s*$", ops)
                if m:
                    simm = int(m.group(1))
                    if simm >= 32768:
                        simm -= 65536
                    tgt = a + 4 + 4 * simm
            if tgt is not None and tgt <= a and tgt in addr2i:
                found.append((addr2i[tgt], i))
    has_mfma = lambda lo, hi: any(ins[k][1].startswith(("v_mfma", "v_smfmac")) for k in range(lo, hi + 1))
    found = [l for l in found if has_mfma(*l)]
    inner = [l for l in found if not any(o != l and l[0] <= o[0] and o[1] <= l[1] for o in found)]
    return sorted(set(inner))


def demangle(names):
    try:
        out = subprocess.run([CXXFILT], input="\n".join(names), capture_output=True, text=True, check=True).stdout.splitlines()
        return dict(zip(names, out))
    except (OSError, subprocess.CalledProcessError):
        return {n: n for n in names}


def audit(lib, kernels=None, min_mfma=8):
    rows = []
    for text in disassemble(lib):
        for name, ins in functions(text):
            if kernels and not any(k in name for k in kernels):
                continue
            for lo, hi in loops(ins):
                c = collections.Counter(classify(mn) for _, mn, _ in ins[lo:hi + 1])
                if c["mfma"] < min_mfma:
                    continue
                flops = sum(mfma_flops(mn) for _, mn, _ in ins[lo:hi + 1] if classify(mn) == "mfma")
                total = sum(c.values())
                shapes = collections.Counter(re.search(r"_(\d+x\d+x\d+)", mn).group(1) for _, mn, _ in ins[lo:hi + 1]
                                             if classify(mn) == "mfma" and re.search(r"_(\d+x\d+x\d+)", mn))
                tr = sum(1 for _, mn, _ in ins[lo:hi + 1] if mn.startswith("ds_read_b64_tr"))
                rows.append(dict(kernel=name, at="0x%x" % ins[lo][0], n=total, mfma=c["mfma"], valu=c["valu"], salu=c["salu"], lds=c["lds"],
                                 vmem=c["vmem"], sync=c["sync"], branch=c["branch"], other=c["other"], lds_tr=tr,
                                 shape=",".join("%s x%d" % kv for kv in shapes.items()),
                                 non_mfma_per_mfma=(total - c["mfma"]) / c["mfma"], kflop_per_instr=flops / total / 1e3))
    dm = demangle(sorted({r["kernel"] for r in rows}))
    for r in rows:
        r["name"] = re.sub(r"^void ", "", dm[r["kernel"]])
    return rows


PK_F32_SRC1_HI = re.compile(r"op_sel:\[(\d),(\d)(?:,(\d))?\]")


def packed_fp32_high_half_reads(lib):
    """{demangled kernel: count} of packed fp32 VALU instructions whose LOW result half reads the HIGH half of its second (or third)
    source (`op_sel:[x,1]` / `[x,x,1]`): the form the MI355X erratum of round 5 hits (csrc/common.h, VN_NO_PK_F32; minimal
    reproduction: tools/reduce_race_repro.py --pkvariants).  The library must not contain it."""
    hits = collections.Counter()
    for text in disassemble(lib):
        for name, ins in functions(text):
            for _, mn, ops in ins:
                if mn.startswith("v_pk_") and mn.endswith("_f32"):
                    m = PK_F32_SRC1_HI.search(ops)
                    if m and (m.group(2) == "1" or m.group(3) == "1"):
                        hits[name] += 1
    dm = demangle(sorted(hits))
    return {dm[k]: v for k, v in hits.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=os.path.join(ROOT, "vinet_amd", "libvinet_hip.so"))
    ap.add_argument("--kernels", default="", help="comma-separated substrings of (mangled) kernel names; default: all")
    ap.add_argument("--min-mfma", type=int, default=8, help="ignore loops with fewer MFMAs (split-K tails, prologues)")
    ap.add_argument("--md", action="store_true", help="markdown table")
    ap.add_argument("--pk-erratum", action="store_true", help="list the kernels that contain a packed fp32 instruction reading the high half of its second source (must be none)")
    args = ap.parse_args()
    if args.pk_erratum:
        hits = packed_fp32_high_half_reads(args.lib)
        for k, v in sorted(hits.items()):
            print("%4d  %s" % (v, k))
        print("%d kernel(s) contain the affected form" % len(hits))
        return 1 if hits else 0
    rows = audit(args.lib, [k for k in args.kernels.split(",") if k], args.min_mfma)
    rows.sort(key=lambda r: (r["name"], r["at"]))
    hdr = ["kernel (loop at)", "instr", "MFMA", "VALU", "SALU", "LDS", "(tr)", "VMEM", "sync", "br", "non-MFMA : MFMA", "kFLOP / instr", "MFMA shape"]
    sep = " | " if args.md else "  "
    if args.md:
        print("| " + " | ".join(hdr) + " |")
        print("|" + "---|" * len(hdr))
    else:
        print("%-86s %6s %5s %5s %5s %5s %5s %5s %5s %3s %8s %8s  %s" % tuple(hdr))
    for r in rows:
        nm = "%s (%s)" % (r["name"][:72], r["at"])
        vals = (nm, r["n"], r["mfma"], r["valu"], r["salu"], r["lds"], r["lds_tr"], r["vmem"], r["sync"], r["branch"], r["non_mfma_per_mfma"], r["kflop_per_instr"], r["shape"])
        if args.md:
            print("| %s | %d | %d | %d | %d | %d | %d | %d | %d | %d | %.2f | %.2f | %s |" % vals)
        else:
            print("%-86s %6d %5d %5d %5d %5d %5d %5d %5d %3d %8.2f %8.2f  %s" % vals)
    return 0


if __name__ == "__main__":
    sys.exit(main())
