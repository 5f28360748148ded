#!/usr/bin/env python3
"""Follow-up of tools/fork_soak.py: the run-to-run mismatch sits in SoundNet's last layer (audionet.conv7.bias / batchnorm7.weight) with or
without forks.  Capture what its backward sees in every run -- dz (the gradient behind BatchNorm + ReLU), z, the statistics, dy after the
apply pass -- and report the first tensor that differs from the reference run."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = [sys.argv[0], "0", "avinet", "--one-stream"]
import torch
import tools.fork_soak as S          # (runs its reference pass on import; rounds = 0)
from vinet_amd import engine as E

orig = E._conv_backward
cap = {}


def hooked(ctx, plan, x, res, bn, act, train_bn, keep, M):
    tag = None
    if bn is not None and not isinstance(bn, E.JointBN) and plan.N == 1024 and plan.Cin == 512 and plan.bias is not None:
        tag = "conv7"
        dz = res.grad_view()
        cap["dz"] = dz.buf.clone(); cap["z"] = res.v.buf.clone(); cap["mean"] = keep["mean"].clone(); cap["invstd"] = keep["invstd"].clone()
        cap["scale"] = res.scale.clone(); cap["shift"] = res.shift.clone(); cap["x"] = x.v.buf.clone()
    orig(ctx, plan, x, res, bn, act, train_bn, keep, M)
    if tag:
        cap["dy"] = res.grad_view().buf.clone()
        cap["gb"] = plan.bias.grad.clone()


E._conv_backward = hooked
rounds = int(os.environ.get("ROUNDS", "14"))
S.run(0, False)
ref = {k: v.clone() for k, v in cap.items()}
for r in range(rounds):
    S.churn(100 + r)
    S.run(0, False)
    torch.cuda.synchronize()
    diffs = [(k, float((cap[k].float() - ref[k].float()).abs().max())) for k in ref if not torch.equal(cap[k], ref[k])]
    print("round", r, "differing:", diffs, flush=True)
