#!/usr/bin/env python3
"""fp32s training step at the train_step golden's size: per-parameter gradient error against the fp64 oracle, sorted by the
share of the whole-vector error (which tensors carry it), for the three-launch bf16-plane weight gradients and for the dedicated
split kernel (VINET_SPLIT_WGRAD_BF16=0).  python tools/dbg_split_grad.py [fp32s|fp32|bf16]"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests import goldens as G
from oracle import vinet_cpu as O
from vinet_amd import engine as E, loss as VL, model as VM, synth

dt = sys.argv[1] if len(sys.argv) > 1 else "fp32s"
E.set_default_dtype(dt)
z, meta = G.load("train_step")
B, T, H, W = meta["B"], meta["T"], meta["H"], meta["W"]
x = synth.clip(B, T, H, W, meta["seed"]).permute(0, 2, 1, 3, 4)
gt = synth.gt_map(B, H, W, meta["seed"])
dev = torch.device("cuda:0")
m = VM.VideoSaliencyModel(num_clips=8)
m.load_state_dict(G.state_dict_for(m, meta["seed"], z, meta))
m = m.to(dev).train()
pred = m(x.to(dev))
VL.kldiv(pred, gt.to(dev)).backward()
o = O.VideoSaliencyModel(num_clips=8)
o.load_state_dict(G.state_dict_for(o, meta["seed"], z, meta))
o = o.double().train()
O.kldiv(o(x.double()), gt.double()).backward()
truth = {k: p.grad.double() for k, p in o.named_parameters()}
rows, num, den = [], 0.0, 0.0
for k, p in m.named_parameters():
    t = truth[k]
    d2 = float((p.grad.double().cpu() - t).pow(2).sum())
    t2 = float(t.pow(2).sum())
    rows.append((d2, t2, k))
    num += d2; den += t2
print("dtype %s SPLIT_WGRAD_BF16=%s global rel L2 %.3e" % (dt, os.environ.get("VINET_SPLIT_WGRAD_BF16", "1"), (num / den) ** 0.5))
rows.sort(reverse=True)
for d2, t2, k in rows[:14]:
    print("  share %5.1f%%  rel %.3e  norm %.3e  %s" % (100 * d2 / num, (d2 / (t2 + 1e-300)) ** 0.5, t2 ** 0.5, k))

print("-- in reverse order of the forward pass (decoder first): rel err, cosine - 1, best-fit scale")
names = [k for k, _ in m.named_parameters()]
params = dict(m.named_parameters())
for k in names[::-1][:12] + names[::-1][60:66] + names[:4]:
    a, t = params[k].grad.double().cpu().reshape(-1), truth[k].reshape(-1)
    cos = float(a @ t / (a.norm() * t.norm() + 1e-300))
    print("  rel %.3e  cos-1 %+.2e  scale %.5f  %s" % (float((a - t).norm() / (t.norm() + 1e-300)), cos - 1, float(a @ t / (t @ t + 1e-300)), k))
po = o(x.double()).detach()
print("pred max abs err vs fp64 oracle: %.3e" % float((pred.detach().double().cpu() - po).abs().max()))
