import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vinet_amd import _lib, engine as E, loss as VL, model as VM, optim as VO, synth
from vinet_amd.graph import GraphedTrainStep
_lib.load(); E.set_default_dtype("bf16")
if os.environ.get("NO_SIDE"): E.WGRAD_SIDE_STREAM = False
DEV = torch.device("cuda:0")
B, T, H, W = 2, 8, 64, 96
x = synth.clip(B, T, H, W, 3).permute(0, 2, 1, 3, 4).to(DEV).contiguous()
gt = synth.gt_map(B, H, W, 3).to(DEV)
res = {}
for mode in ("eager", "graph"):
    m = VM.VideoSaliencyModel(num_clips=T)
    m.load_state_dict(synth.synth_state_dict(m.state_dict(), 3))
    m = m.to(DEV).train()
    opt = VO.Adam([p for p in m.parameters() if p.requires_grad], lr=1e-4)
    names = [n for n, p in m.named_parameters() if p.requires_grad]
    if mode == "graph":
        step = GraphedTrainStep(m, opt, VL.kldiv, (x,), gt)
        l = float(step((x,), gt))
    else:
        opt.zero_grad(); l = VL.kldiv(m(x), gt); l.backward(); opt.step(); l = float(l)
    torch.cuda.synchronize()
    res[mode] = (l, opt.flat_g.clone(), opt.flat_p.clone(), names, opt._offs, [p.numel() for p in opt._params])
(le, ge, pe, names, offs, nums), (lg, gg, pg, *_ ) = res["eager"], res["graph"]
print("loss", le, lg)
print("grad diff max", float((ge - gg).abs().max()), "rel", float((ge - gg).norm() / ge.norm()))
print("param diff max", float((pe - pg).abs().max()), "mean", float((pe - pg).abs().mean()))
bad = []
for n, o, k in zip(names, offs, nums):
    d = float((ge[o:o + k] - gg[o:o + k]).norm() / (ge[o:o + k].norm() + 1e-30))
    if d > 1e-6:
        bad.append((d, n))
print(len(bad), "of", len(names), "parameters differ in gradient")
for d, n in sorted(bad, reverse=True)[:int(os.environ.get("SHOW", "25"))]:
    print("%.3e %s" % (d, n))
