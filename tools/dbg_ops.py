import torch, sys
sys.path.insert(0, '.')
from vinet_amd import _lib as L, ops, synth
L.load()
dev = torch.device('cuda:0')
x = synth.normal("gopx", (2, 16, 4, 12, 16), 1)
w = synth.normal("gopw", (32, 16, 1, 3, 3), 2) * 0.1
dy = synth.normal("gopp", (2, 32, 4, 12, 16), 4)
res = {}
for dt in (torch.float32, torch.bfloat16):
    xc = x.permute(0, 2, 3, 4, 1).contiguous().to(dev, dt)
    dyc = dy.permute(0, 2, 3, 4, 1).contiguous().to(dev, dt)
    dx = torch.ops.vinet.conv3d_bwd_data(dyc, xc, w.to(dev), [1, 1, 1], [0, 1, 1]).float().cpu()
    res[dt] = dx
a, b = res[torch.float32], res[torch.bfloat16]
print("rel", float((a - b).norm() / a.norm()))
d = (a - b).abs()
bad = (d > 0.1).nonzero()
print(bad.shape, bad[:20].tolist())
print("per-channel err", [round(float(d[..., c].mean()), 3) for c in range(16)])
print("per-w err", [round(float(d[:, :, :, wi].mean()), 3) for wi in range(16)])
print("per-b/t", [[round(float(d[bi, ti].mean()), 3) for ti in range(4)] for bi in range(2)])
