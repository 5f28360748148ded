"""One point of bench.py's local-batch sweep: the training step at a small batch, eager and as a replayed hipGraph.
usage: python tools/sweep_small.py <batch> [steps]   (knobs come from the environment: VINET_WGRAD_GROUP, ...)"""
import sys
import time

import torch

sys.path.insert(0, ".")
from vinet_amd import engine, loss, model, optim, parallel, synth  # noqa: E402
from vinet_amd.graph import GraphedTrainStep  # noqa: E402

B = int(sys.argv[1])
K = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda:0")
engine.set_default_dtype("bf16")
m = model.VideoSaliencyModel(num_clips=32)
m.load_state_dict(synth.synth_state_dict(m.state_dict(), 0))
m = m.to(dev).train()
x = torch.randn((B, 32, 3, 224, 384), device=dev).permute(0, 2, 1, 3, 4).contiguous()
gt = synth.gt_map(B, 224, 384, 0).to(dev)
opt = optim.Adam(parallel.trainable_parameters(m), lr=1e-4)


def step():
    opt.zero_grad()
    l = loss.kldiv(m(x), gt)
    l.backward()
    opt.step()
    return l


def rate(fn, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return B * n / (time.perf_counter() - t0)


for _ in range(3):
    step()
e = max(rate(step, K) for _ in range(2))
import os  # noqa: E402
if os.environ.get("SWEEP_GRAPH_ONE_STREAM", "0") != "0":      # A/B: the captured step on one stream
    engine.WGRAD_SIDE_STREAM = False
g = GraphedTrainStep(m, opt, loss.kldiv, (x,), gt)
g((x,), gt)
gr = max(rate(lambda: g((x,), gt), K) for _ in range(2))
print("batch %d  eager %.1f  graph %.1f clips/s" % (B, e, gr))
