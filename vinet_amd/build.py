"""Build libvinet_hip.so (gfx950) in-tree with hipcc.

    python -m vinet_amd.build [--force]

One translation unit per kernel family, compiled in parallel, linked into
``vinet_amd/libvinet_hip.so``.  The .so is git-ignored but travels to the GPU
box with the gpurun snapshot.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "obj")
LIB = os.path.join(HERE, "libvinet_hip.so")
SOURCES = ["conv_api.hip", "conv_bf16.hip", "conv_bnb.hip", "conv_f32.hip", "conv_wgrad.hip", "wgrad_dma.hip", "wgrad_pp.hip", "wgrad_ts.hip", "conv_ts.hip", "wgrad_hs.hip", "wgrad_rs.hip", "wgrad_tf.hip", "conv_hs.hip", "layout.hip", "bn.hip", "pool.hip", "resample.hip", "loss_adam.hip", "postproc.hip", "preproc.hip"]
import glob
# every header of csrc/ and include/ is a dependency of every object (a stale object travelling to the GPU box is worse
# than a rebuild of 20 files)
HEADERS = sorted(glob.glob(os.path.join(CSRC, "*.h"))) + sorted(glob.glob(os.path.join(ROOT, "include", "*.h")))
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC,
         "-Wno-unused-result", "-ffp-contract=off"]


# Kernels whose hand-counted `s_waitcnt` / EXEC-mask inline asm assumes that the compiler neither spills nor copies the registers
# the asm owns (conv_pw.h: the global-load ring with counted vmcnt; pool.hip: routing on the EXEC mask).  A vector-register spill
# in one of them would read in-flight data: the build FAILS on it instead of shipping silently wrong kernels.  Checked from
# hipcc's own -Rpass-analysis=kernel-resource-usage remarks of the same compilation (pinned toolchain: ROCm 7.2 / clang 22).
GUARDED = {"conv_bf16.hip": ("conv_pw_kernel",), "pool.hip": ("maxpool_bwd",), "conv_bnb.hip": ()}     # (conv_bnb.hip: resource table only)
RESOURCES = os.path.join(CSRC, "obj", "kernel_resources.json")


def _parse_resources(text):
    """hipcc remarks -> {mangled kernel name: {vgpr, agpr, sgpr, scratch, vgpr_spill, sgpr_spill, occupancy}}"""
    import re
    out, cur = {}, None
    keys = {"VGPRs": "vgpr", "AGPRs": "agpr", "TotalSGPRs": "sgpr", "ScratchSize [bytes/lane]": "scratch", "VGPRs Spill": "vgpr_spill",
            "SGPRs Spill": "sgpr_spill", "Occupancy [waves/SIMD]": "occupancy"}
    for line in text.splitlines():
        m = re.search(r"remark:\s+Function Name: (\S+)", line)
        if m:
            cur = out.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark:\s+([A-Za-z \[\]/]+): (\d+)", line)
        if m and cur is not None and m.group(1).strip() in keys:
            cur[keys[m.group(1).strip()]] = int(m.group(2))
    return out


def check_guarded(src, text):
    res = _parse_resources(text)
    bad = [(k, v) for k, v in res.items() if any(g in k for g in GUARDED.get(src, ())) and v.get("vgpr_spill", 0) > 0]
    if bad:
        raise RuntimeError("vector-register spills in kernels whose inline asm owns in-flight registers (%s): %s" % (src, bad))
    return res


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_variant(tag, extra_flags, verbose=True):
    """side build for A/B experiments: vinet_amd/csrc/obj_<tag>/ -> /tmp/libvinet_hip_<tag>.so"""
    global OBJ, LIB, FLAGS
    save = (OBJ, LIB, FLAGS)
    OBJ, LIB, FLAGS = os.path.join(CSRC, "obj_" + tag), os.path.join(HERE, "libvinet_hip_%s.so" % tag), FLAGS + list(extra_flags)
    try:
        return build(force=False, verbose=verbose)
    finally:
        OBJ, LIB, FLAGS = save


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()
    hdrs = [h if os.path.isabs(h) else os.path.join(CSRC, h) for h in HEADERS]
    jobs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src.replace(".hip", ".o"))
        if force or _stale(o, [s] + hdrs):
            jobs.append((s, o))

    def compile_one(job):
        s, o = job
        guarded = os.path.basename(s) in GUARDED
        cmd = [hipcc] + FLAGS + (["-Rpass-analysis=kernel-resource-usage"] if guarded else []) + ["-c", s, "-o", o]
        r = subprocess.run(cmd, capture_output=True, text=True)
        out = r.stdout + r.stderr
        if guarded and r.returncode == 0:
            try:
                res = check_guarded(os.path.basename(s), out)
            except RuntimeError:
                os.remove(o)
                raise
            out = "\n".join(l for l in out.splitlines() if "kernel-resource-usage" not in l and not l.startswith("   ") )
            return s, r.returncode, out, res
        return s, r.returncode, out, None

    if jobs:
        if verbose:
            print("[vinet_amd.build] compiling %d file(s) for gfx950 ..." % len(jobs), flush=True)
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4)) as ex:
            resources, failed = {}, None
            for s, rc, out, res in ex.map(compile_one, jobs):
                if rc != 0:
                    failed = failed or RuntimeError("hipcc failed on %s:\n%s" % (s, out))
                    continue
                if res is not None:
                    resources[os.path.basename(s)] = res
                if verbose and out.strip():
                    print(out)
        if resources:
            # (written ONCE, here, after every compile job has finished: the jobs run on a thread pool; the file is a build
            #  artefact -- git-ignored -- that tools read for register counts)
            import json
            try:
                allres = json.load(open(RESOURCES)) if os.path.exists(RESOURCES) else {}
            except ValueError:
                allres = {}
            allres.update(resources)
            with open(RESOURCES, "w") as f:
                json.dump(allres, f, indent=1, sort_keys=True)
        if failed is not None:
            raise failed
    objs = [os.path.join(OBJ, s.replace(".hip", ".o")) for s in SOURCES]
    if force or jobs or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stdout + r.stderr)
        if verbose:
            print("[vinet_amd.build] linked", LIB, flush=True)
        try:
            audit_packed_fp32(LIB, verbose)
        except RuntimeError:
            os.remove(LIB)          # never leave a library with the erratum form where _lib.load() would pick it up
            raise
    return LIB


def audit_packed_fp32(lib, verbose=True):
    """Gate on the LINKED library, main build and build_variant() alike: hipcc emits `v_pk_*_f32 ... op_sel:[x,1]`
    wherever it happened to allocate a register pair in swapped order, and that form returns wrong sums beside foreign MFMA waves
    (csrc/common.h).  A kernel that shows up here must take VN_NO_PK_F32; the build fails instead of shipping it.
    Without llvm-objdump (not this image) the audit cannot run: say so, loudly, rather than pass in silence."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("vinet_isa_audit", os.path.join(ROOT, "tools", "isa_audit.py"))
    ia = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ia)
    if not os.path.exists(ia.OBJDUMP):
        print("[vinet_amd.build] WARNING: %s missing -- the packed-fp32 erratum audit of %s did NOT run" % (ia.OBJDUMP, lib), file=sys.stderr, flush=True)
        return None
    hits = ia.packed_fp32_high_half_reads(lib)
    if hits:
        raise RuntimeError("packed fp32 instructions with op_sel:[x,1] (MI355X erratum form) in %s: %s -- build those kernels with VN_NO_PK_F32" % (lib, hits))
    if verbose:
        print("[vinet_amd.build] packed-fp32 erratum audit: clean", flush=True)
    return hits


if __name__ == "__main__":
    build(force="--force" in sys.argv)
