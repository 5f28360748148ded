#!/usr/bin/env python3
"""Where does the split-bf16 (fp32s) training step leave the exact fp32 step?  (GPU; no oracle: the engine's own fp32 path is
the baseline -- tests/test_gpu_model.py::test_train_step_well_conditioned[fp32] holds it within 1.06 x the reference's fp32 error)

    python tools/split_grad_probe.py [--cfg=name=value,...] [--fixture=train_step_wc] [--dtype=fp32s|bf16]

Runs the `train_step_wc` problem (ViNet-8, B = 12, 8 x 128 x 192, tests/golden/train_step_wc.npz weights) once per dtype and prints
the relative L2 distance of every parameter gradient, in REVERSE module order (the loss end first): the first layer whose error
jumps is where the backward pass of the split form goes wrong.
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from vinet_amd import engine as E, loss as VL, model as VM, synth, _lib

fixture = "train_step_wc"
cfg = ""
for a in sys.argv[1:]:
    if a.startswith("--cfg="):
        cfg = a[6:]
    if a.startswith("--fixture="):
        fixture = a[10:]
_lib.load()
DEV = torch.device("cuda:0")
z = np.load(os.path.join(ROOT, "tests", "golden", fixture + ".npz"))
meta = json.loads(str(z["meta"]))
B, T, H, W = meta["B"], meta["T"], meta["H"], meta["W"]
x = synth.clip(B, T, H, W, meta["seed"]).permute(0, 2, 1, 3, 4).to(DEV)
gt = synth.gt_map(B, H, W, meta["seed"]).to(DEV)


def grads(dtype, cfg_text=""):
    E.set_default_dtype(dtype)
    old = E.configure_from_string(cfg_text) if cfg_text else None
    m = VM.VideoSaliencyModel(num_clips=8)
    sd = synth.synth_state_dict(m.state_dict(), meta["seed"])
    if "head_w" in z.files:
        sd[meta["head_w_key"]] = torch.from_numpy(np.asarray(z["head_w"]))
        sd[meta["head_b_key"]] = torch.from_numpy(np.asarray(z["head_b"]))
    m.load_state_dict(sd)
    m = m.to(DEV).train()
    pred = m(x)
    loss = VL.kldiv(pred, gt)
    loss.backward()
    torch.cuda.synchronize()
    out = [(n, p.grad.detach().double().cpu()) for n, p in m.named_parameters() if p.grad is not None]
    if old:
        E.configure(**old)
    return float(loss), pred.detach().double().cpu(), out


other = "fp32s"
for a in sys.argv[1:]:
    if a.startswith("--dtype="):
        other = a[8:]            # bf16: the benchmarked path against the same baseline
l0, p0, g0 = grads("fp32")
l1, p1, g1 = grads(other, cfg)
print("loss fp32 %.9f %s %.9f   pred rel %.3e" % (l0, other, l1, float((p1 - p0).norm() / p0.norm())))
num = den = 0.0
rows = []
for (n, a), (n2, b) in zip(g0, g1):
    assert n == n2
    num += float((a - b).pow(2).sum()); den += float(a.pow(2).sum())
    rows.append((n, float((a - b).norm() / (a.norm() + 1e-30)), float(a.norm()), float((a * b).sum() / (a.norm() * b.norm() + 1e-30))))
print("whole gradient vector: rel L2 %.4e" % ((num / den) ** 0.5))
for n, r, nrm, cos in reversed(rows):
    print("%-58s rel %.3e   |g| %.3e   cos %.6f" % (n, r, nrm, cos))
