#!/usr/bin/env python3
"""Host-side cost of one training step (cProfile at batch 1, where the GPU work hides nothing): which Python functions the
~1100 kernel launches spend their time in.   python tools/py_overhead.py [batch]"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from vinet_amd import engine as E, loss as Lo, model as M, optim as O, parallel as P, synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device("cuda:0")
E.set_default_dtype("bf16")
m = M.VideoSaliencyModel(num_clips=32)
m.load_state_dict(synth.synth_state_dict(m.state_dict(), 0))
m = m.to(dev).train()
opt = O.Adam(P.trainable_parameters(m), lr=1e-4)
x = torch.randn(B, 32, 3, 224, 384, device=dev).permute(0, 2, 1, 3, 4)
gt = synth.gt_map(B, 224, 384, 0).to(dev)


def step():
    opt.zero_grad()
    l = Lo.kldiv(m(x), gt)
    l.backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    step()
t1 = time.perf_counter()          # host time only: no synchronize
torch.cuda.synchronize()
t2 = time.perf_counter()
print("batch %d: host %.2f ms/step issue time, %.2f ms/step wall" % (B, (t1 - t0) * 100, (t2 - t0) * 100))
pr = cProfile.Profile()
pr2 = cProfile.Profile()            # the tape's backward runs in autograd's worker thread: its own profile
orig = E._TapeFn.backward


def traced(ctxa, *g):
    pr2.enable()
    try:
        return orig(ctxa, *g)
    finally:
        pr2.disable()


E._TapeFn.backward = staticmethod(traced)
pr.enable()
for _ in range(5):
    step()
pr.disable()
torch.cuda.synchronize()
print("==== forward / main thread")
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
print("==== backward (tape) thread")
pstats.Stats(pr2).sort_stats("tottime").print_stats(24)
