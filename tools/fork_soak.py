#!/usr/bin/env python3
"""Hunt for the intermittent mismatch of the forked backward schedule: it showed once in three full-suite runs,
never alone -- the signature of caching-allocator history.  This loop alternates (a) a churn phase that leaves the allocator's
per-stream pools in a different state every round (models and clips of other shapes, with and without forks, freed in odd
orders) with (b) the comparison itself: AViNet / ViNet training steps, forked forward + backward against the one-stream schedule.

    python tools/fork_soak.py [rounds] [net] [--record-stream]
"""
import os
import sys
import random

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from vinet_amd import engine as E, loss as VL, model as VM, optim as VO, synth, _lib

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 10
net = sys.argv[2] if len(sys.argv) > 2 else "avinet"
forked = "--one-stream" not in sys.argv      # --one-stream: the comparison runs the ONE-stream schedule too (control: what varies without forks?)
_lib.load()
for a_ in sys.argv:
    if a_.startswith("--cfg="):
        E.configure_from_string(a_[6:])
DEV = torch.device("cuda:0")
E.set_default_dtype("bf16")
av = net == "avinet"
B, T, H, W = (2, 32, 224, 384) if av else (4, 16, 128, 192)
x = synth.clip(B, T, H, W, 11).permute(0, 2, 1, 3, 4).to(DEV).contiguous()
ins = (x, synth.audio(B, 70560, 11).to(DEV)) if av else (x,)
gt = synth.gt_map(B, H, W, 11).to(DEV)


def run(vox, bwd, steps=2):
    E.configure(branch_streams_train_vox=vox, branch_streams_bwd=bwd, branch_streams_bwd_min_batch=1)
    m = (VM.VideoAudioSaliencyModel if av else VM.VideoSaliencyModel)(num_clips=T)
    m.load_state_dict(synth.synth_state_dict(m.state_dict(), 11))
    m = m.to(DEV).train()
    opt = VO.Adam([p for p in m.parameters() if p.requires_grad], lr=1e-4)
    losses = []
    for _ in range(steps):
        opt.zero_grad()
        l = VL.kldiv(m(*ins), gt)
        l.backward()
        losses.append(float(l))
    torch.cuda.synchronize()
    names = [n for n, p in m.named_parameters() if p.requires_grad]
    sizes = [p.numel() for n, p in m.named_parameters() if p.requires_grad]
    return losses, opt.flat_g.clone(), names, sizes


def churn(seed):
    rnd = random.Random(seed)
    keep = []
    for i in range(rnd.randint(2, 5)):
        t = rnd.choice([8, 16])
        b = rnd.choice([1, 2, 3])
        h, w = rnd.choice([(64, 96), (96, 192), (128, 192)])
        E.configure(branch_streams_train_vox=rnd.choice([0, 1 << 30]), branch_streams_bwd=rnd.choice([False, True]), branch_streams_bwd_min_batch=1)
        m = VM.VideoSaliencyModel(num_clips=t)
        m.load_state_dict(synth.synth_state_dict(m.state_dict(), seed + i))
        m = m.to(DEV).train()
        xx = synth.clip(b, t, h, w, seed).permute(0, 2, 1, 3, 4).to(DEV).contiguous()
        g = synth.gt_map(b, h, w, seed).to(DEV)
        VL.kldiv(m(xx), g).backward()
        keep.append((m, xx, torch.empty(rnd.randint(1, 64) << 20, device=DEV)))
        if rnd.random() < 0.5 and keep:
            keep.pop(rnd.randrange(len(keep)))
        if rnd.random() < 0.3:
            torch.cuda.empty_cache()
    del keep


ref = run(0, False)
bad = 0
for r in range(rounds):
    churn(100 + r)
    got = run(1 << 30, True) if forked else run(0, False)
    rel = float((got[1] - ref[1]).norm() / ref[1].norm())
    ok = got[0] == ref[0] and rel < 3e-6
    print("round %d: losses %s rel %.3e %s" % (r, "equal" if got[0] == ref[0] else "DIFFER %s vs %s" % (got[0], ref[0]), rel, "ok" if ok else "MISMATCH"), flush=True)
    if not ok:
        bad += 1
        off = 0
        nd = (got[1] != ref[1]).nonzero().flatten()
        print("    elements of the flat gradient that differ at all: %d of %d, first %s last %s" % (nd.numel(), ref[1].numel(), nd[:3].tolist(), nd[-3:].tolist()))
        for n, s in zip(got[2], got[3]):
            d = float((got[1][off:off + s] - ref[1][off:off + s]).norm() / (ref[1][off:off + s].norm() + 1e-30))
            if d > 1e-4:
                print("    %-60s rel %.3e   |ref| %.3e |got| %.3e (whole gradient %.3e)" % (n, d, float(ref[1][off:off + s].norm()), float(got[1][off:off + s].norm()), float(ref[1].norm())))
                dd = (got[1][off:off + s] - ref[1][off:off + s]).abs()
                idx = (dd > 0).nonzero().flatten()
                print("      flat offset %d, %d of %d elements differ, indices %s ... %s; largest |diff| %.3e at %d (ref %.3e got %.3e)" % (
                    off, idx.numel(), s, idx[:6].tolist(), idx[-3:].tolist(), float(dd.max()), int(dd.argmax()), float(ref[1][off + int(dd.argmax())]), float(got[1][off + int(dd.argmax())])))
            off += s
print("mismatches: %d of %d" % (bad, rounds))
