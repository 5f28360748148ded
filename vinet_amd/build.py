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
SOURCES = ["conv_api.hip", "conv_bf16.hip", "conv_f32.hip", "conv_wgrad.hip", "wgrad_dma.hip", "wgrad_pp.hip", "wgrad_ts.hip", "conv_ts.hip", "wgrad_hs.hip", "wgrad_rs.hip", "wgrad_tf.hip", "conv_hs.hip", "layout.hip", "bn.hip", "pool.hip", "resample.hip", "loss_adam.hip", "postproc.hip", "preproc.hip"]
import glob
# every header of csrc/ and include/ is a dependency of every object (a stale object travelling to the GPU box is worse
# than a rebuild of 20 files)
HEADERS = sorted(glob.glob(os.path.join(CSRC, "*.h"))) + sorted(glob.glob(os.path.join(ROOT, "include", "*.h")))
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC,
         "-Wno-unused-result", "-ffp-contract=off"]


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
        cmd = [hipcc] + FLAGS + ["-c", s, "-o", o]
        r = subprocess.run(cmd, capture_output=True, text=True)
        return s, r.returncode, r.stdout + r.stderr

    if jobs:
        if verbose:
            print("[vinet_amd.build] compiling %d file(s) for gfx950 ..." % len(jobs), flush=True)
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4)) as ex:
            for s, rc, out in ex.map(compile_one, jobs):
                if rc != 0:
                    raise RuntimeError("hipcc failed on %s:\n%s" % (s, out))
                if verbose and out.strip():
                    print(out)
    objs = [os.path.join(OBJ, s.replace(".hip", ".o")) for s in SOURCES]
    if force or jobs or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stdout + r.stderr)
        if verbose:
            print("[vinet_amd.build] linked", LIB, flush=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
