#!/usr/bin/env python3
"""Soak: N bf16 training steps on one fixed synthetic batch; the loss must fall monotonically-ish and stay finite.
Catches rare wrong tiles / races that a single-step parity test cannot.   python tools/soak.py [steps] [batch]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from vinet_amd import engine, loss as VL, model as VM, optim as VO, synth

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 120
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device("cuda:0")
engine.set_default_dtype("bf16")
m = VM.VideoSaliencyModel(num_clips=32)
m.load_state_dict(synth.synth_state_dict(m.state_dict(), 0))
m = m.to(dev).train()
x = synth.clip(B, 32, 224, 384, 1).to(dev).permute(0, 2, 1, 3, 4)
gt = synth.gt_map(B, 224, 384, 1).to(dev)
opt = VO.Adam([p for p in m.parameters() if p.requires_grad], lr=1e-4)
hist = []
for i in range(steps):
    opt.zero_grad()
    l = VL.kldiv(m(x), gt)
    l.backward()
    opt.step()
    if i % 10 == 0 or i == steps - 1:
        v = float(l.detach())
        hist.append(v)
        print("step %4d  loss %.5f" % (i, v), flush=True)
        assert v == v and v < 1e3, "loss diverged"
assert hist[-1] < 0.5 * hist[0], "loss did not fall: %s" % hist
bad = [k for k, p in m.named_parameters() if not torch.isfinite(p).all()]
assert not bad, bad
print("soak ok: %.4f -> %.4f over %d steps" % (hist[0], hist[-1], steps))
