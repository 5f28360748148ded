#!/usr/bin/env python3
"""Where do the hipMemcpy / aten fill launches of one training step come from?  (torch.profiler with Python stacks)"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import vinet_amd.model as M, vinet_amd.optim as O, vinet_amd.loss as Lo, vinet_amd.engine as E, vinet_amd.parallel as P

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda:0")
torch.manual_seed(0)
from vinet_amd import synth
E.set_default_dtype("bf16")
m = M.VideoSaliencyModel(num_clips=32)
m.load_state_dict(synth.synth_state_dict(m.state_dict(), 0))
m = m.to(dev).train()
opt = O.Adam(P.trainable_parameters(m), lr=1e-4)
buckets = P.GradientBuckets(opt)
x = torch.randn(B, 32, 3, 224, 384, device=dev).permute(0, 2, 1, 3, 4)
gt = synth.gt_map(B, 224, 384, 0).to(dev)


def step():
    opt.zero_grad()
    buckets.begin_step()
    l = Lo.kldiv(m(x), gt)
    l.backward()
    buckets.finish()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
cnt = collections.Counter()
kc = collections.Counter()
for ev in prof.events():
    n = ev.name
    if "Memcpy" in n or "copyBuffer" in n or "Fill" in n or "fill" in n:
        kc[n[:90]] += 1
    if any(k in n for k in ("aten::copy_", "aten::fill_", "aten::zero_", "aten::zeros", "aten::to", "aten::_to_copy", "aten::clone", "aten::contiguous", "aten::add", "aten::mul", "aten::sum", "aten::empty")):
        st = [s for s in (ev.stack or []) if "vinet_amd" in s or "find_copies" in s]
        cnt[(n, tuple(st[:3]))] += 1
for (n, st), c in sorted(cnt.items(), key=lambda kv: -kv[1])[:40]:
    print(c, n, " <- ".join(st))
rt = collections.Counter(); rtt = collections.Counter()
for ev in prof.events():
    if ev.name.startswith("hip"):
        rt[ev.name] += 1; rtt[ev.name] += ev.cpu_time_total
print("---- runtime API (count, total us)")
for n, c in rt.most_common(12):
    print(c, int(rtt[n]), n)
import time
for _ in range(2):
    step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(3):
    step()
torch.cuda.synchronize(); print("ms/step", (time.perf_counter() - t0) / 3 * 1e3)
print("---- device-side")
for n, c in kc.most_common(12):
    print(c, n)
