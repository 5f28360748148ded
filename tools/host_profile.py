"""Host-side cost of one eager training step at a small batch: cProfile over K steps after warm-up (GPU box).
usage: python tools/host_profile.py [batch] [steps]"""
import cProfile
import io
import pstats
import sys
import time

import torch

sys.path.insert(0, ".")
from vinet_amd import engine, loss, model, optim, parallel, synth  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
K = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda:0")
engine.set_default_dtype("bf16")
m = model.VideoSaliencyModel(num_clips=32)
m.load_state_dict(synth.synth_state_dict(m.state_dict(), 0))
m = m.to(dev).train()
x = torch.randn((B, 32, 3, 224, 384), device=dev).permute(0, 2, 1, 3, 4)
gt = synth.gt_map(B, 224, 384, 0).to(dev)
opt = optim.Adam(parallel.trainable_parameters(m), lr=1e-4)


def step():
    opt.zero_grad()
    l = loss.kldiv(m(x), gt)
    l.backward()
    opt.step()
    return l


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(K):
    step()
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print("batch %d: host issue %.2f ms / step, wall %.2f ms / step" % (B, t_issue / K * 1e3, t_all / K * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(K):
    step()
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(45)
print(s.getvalue()[:9000])
