#!/usr/bin/env python3
"""Do the channel reductions of a training step return the same bits twice?  (GPU.)

    python tools/reduce_dup_check.py [steps] [vinet|avinet] [--cfg=name=value,...]

Round 5 found the last SoundNet layer's bias / BatchNorm gradients differing from run to run: the 256-thread reduction kernels, beside
the weight-gradient stream, dropped ONE voxel's term of one register for 16 lanes in ~7 % of their launches on that 6-voxel tensor --
transient (an identical launch right behind is exact).  On a large tensor such a loss would be invisible to every parity test.  This
tool makes it visible for EVERY reduction of the step: each `vinet_bn_bwd_reduce`, `vinet_channel_stats` and `vinet_channel_sum` the
engine issues writes its partial rows into a preallocated arena (copied from there to the engine's own buffer, which the finalize
pass reads) and is followed by ONE identical launch into the next arena slot; after the step the two are compared bit for bit (no
allocation and no synchronisation inside the step).  The kernels are deterministic
(no atomics in the reduce passes), so any difference is a defect of the kind above.
`--cfg=lib.reduce_small=0` switches the small-tensor kernel off (the defect then shows on the <= 64-voxel tensors of AViNet).
"""
import collections
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from vinet_amd import engine as E, loss as VL, model as VM, optim as VO, synth, _lib

steps = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 10
net = "avinet" if "avinet" in sys.argv else "vinet"
_lib.load()
for a_ in sys.argv:
    if a_.startswith("--cfg="):
        E.configure_from_string(a_[6:])
DEV = torch.device("cuda:0")
E.set_default_dtype("bf16")
av = net == "avinet"
B, T, H, W = (2, 32, 224, 384) if av else (8, 32, 224, 384)
x = synth.clip(B, T, H, W, 11).permute(0, 2, 1, 3, 4).to(DEV).contiguous()
ins = (x, synth.audio(B, 70560, 11).to(DEV)) if av else (x,)
gt = synth.gt_map(B, H, W, 11).to(DEV)
ARENA = torch.empty(3 << 28, dtype=torch.float32, device=DEV)       # 3 GiB of fp32 partial rows
scratch_out = torch.empty(4096, dtype=torch.float32, device=DEV)
top = 0
pairs = []          # (name, (B, T, H, W, C), offset of the engine's own launch, offset of the repeat, floats)
PART_ARG = {"vinet_bn_bwd_reduce": 6, "vinet_channel_stats": 2, "vinet_channel_sum": 2}
orig_call = E.Ctx.call
lib = _lib.load()
hip = C.CDLL("libamdhip64.so")
hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]


def call2(self, name, *args, **kw):
    """the engine's own launch writes its partial rows into the arena (and they are copied to where the engine wants them: the
    finalize pass reads that), ONE identical launch follows into the next arena slot"""
    global top
    pi = PART_ARG.get(name)
    if pi is None:
        return orig_call(self, name, *args, **kw)
    t = args[0]._obj
    rows = lib.vinet_stats_rows(C.byref(t))
    n = rows * 2 * t.C
    if top + 2 * n > ARENA.numel():
        return orig_call(self, name, *args, **kw)
    a1 = list(args)
    a1[pi] = ARENA.data_ptr() + 4 * top
    rc = orig_call(self, name, *a1, **kw)
    assert hip.hipMemcpyAsync(args[pi], a1[pi], 4 * n, 3, args[-1]) == 0
    a2 = list(args)
    a2[pi] = ARENA.data_ptr() + 4 * (top + n)
    if name == "vinet_channel_sum":
        a2[4], a2[5] = scratch_out.data_ptr(), 0
    orig_call(self, name, *a2)
    pairs.append((name, (t.B, t.T, t.H, t.W, t.C), top, top + n, n))
    top += 2 * n
    return rc


E.Ctx.call = call2
bad = collections.Counter()
seen = collections.Counter()
for s in range(steps):
    # a fresh model per round, two steps each (the soak's pattern, tools/fork_soak.py: the defect showed in 1 of 4 such rounds)
    m = (VM.VideoAudioSaliencyModel if av else VM.VideoSaliencyModel)(num_clips=T)
    m.load_state_dict(synth.synth_state_dict(m.state_dict(), 11))
    m = m.to(DEV).train()
    opt = VO.Adam([p for p in m.parameters() if p.requires_grad], lr=1e-4)
    nb = npairs = 0
    for it in range(2):
        top = 0
        pairs.clear()
        opt.zero_grad()
        l = VL.kldiv(m(*ins), gt)
        l.backward()
        torch.cuda.synchronize()
        for name, dims, oa, ob, n in pairs:
            key = (name, dims)
            seen[key] += 1
            if not torch.equal(ARENA[oa:oa + n], ARENA[ob:ob + n]):
                bad[key] += 1
                nb += 1
                d = (ARENA[oa:oa + n] - ARENA[ob:ob + n]).nonzero().flatten()
                print("   round %d.%d: %s on %s: %d of %d partial values differ (first at %d: %.9g vs %.9g)" % (
                    s, it, name, dims, d.numel(), n, int(d[0]), float(ARENA[oa + int(d[0])]), float(ARENA[ob + int(d[0])])), flush=True)
        npairs += len(pairs)
        opt.step()
    print("round %d: loss %.6f, %d reductions checked against an identical launch right behind them, %d pairs differ" % (s, float(l), npairs, nb), flush=True)
    del m, opt
print("\n%s, %d rounds, config %s: %d reduction launches checked, %d pairs differ" % (net, steps, E.config(changed_only=True), sum(seen.values()), sum(bad.values())))
for key, c in sorted(bad.items(), key=lambda kv: -kv[1]):
    dims = key[1]
    print("   %-22s B%d T%d H%d W%d C%d (%d voxels): %d of %d pairs" % (key[0], *dims, dims[0] * dims[1] * dims[2] * dims[3], c, seen[key]))
