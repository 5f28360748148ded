#!/usr/bin/env python3
"""audionet.conv7.bias gradient: is the value in the flat gradient buffer the channel sum of the dy its backward produced?  References
to the tensors are kept (no kernel added to the step) and compared after the step's final synchronisation."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = [sys.argv[0], "0", "avinet", "--one-stream"] + sys.argv[1:]
import torch
import tools.fork_soak as S          # (runs its reference pass on import; rounds = 0)
from vinet_amd import engine as E

orig = E.conv_forward
held = {}


def hooked(ctx, plan, x, bn=None, **kw):
    res = orig(ctx, plan, x, bn=bn, **kw)
    if bn is not None and getattr(plan, "bias", None) is not None and plan.N == 1024 and plan.Cin == 512:
        held.clear()
        held.update(act=res, plan=plan, bn=bn)
    return res


E.conv_forward = hooked
# a SECOND channel sum of the same dy right behind the first one, into a scratch vector: transient (the data arrived in between) or not?
extra = {}
orig_call = E.Ctx.call


FIRST = os.environ.get("EXTRA_FIRST", "0") == "1"
if "g2" not in extra:
    extra["ws"] = torch.empty(4096, device=S.DEV)
    extra["g2"] = torch.zeros(1024, device=S.DEV)


def call2(self, name, *args, **kw):
    hit = name == "vinet_channel_sum" and args[3] == 1024
    if hit and FIRST:
        orig_call(self, name, args[0], args[1], extra["ws"].data_ptr(), args[3], extra["g2"].data_ptr(), 0, args[6])
    orig_call(self, name, *args, **kw)
    if hit and not FIRST:
        orig_call(self, name, args[0], args[1], extra["ws"].data_ptr(), args[3], extra["g2"].data_ptr(), 0, args[6])


E.Ctx.call = call2
rounds = int(os.environ.get("ROUNDS", "16"))
for r in range(rounds):
    got = S.run(0, False)
    torch.cuda.synchronize()
    a, plan, bn = held["act"], held["plan"], held["bn"]
    dy = a._grad.buf.float().view(-1, 1024)
    sdy = dy.double().sum(0)
    gb = plan.bias.grad.double()
    d = (gb - sdy).abs()
    idx = (d > 1e-6).nonzero().flatten()
    d2 = (extra["g2"].double() - sdy).abs()
    print("run %d: |bias grad| %.3e  |sum dy| %.3e  max |bias grad - sum dy| %.3e at %s; differing (>1e-6): %d %s; SECOND sum right behind it: max err %.3e, differing %d" % (
        r, float(gb.norm()), float(sdy.norm()), float(d.max()), int(d.argmax()), idx.numel(), idx[:10].tolist(), float(d2.max()), int((d2 > 1e-6).sum())), flush=True)
    if idx.numel():
        i = int(idx[0])
        print("     element %d: bias grad %.6e, sum dy %.6e, dy column %s" % (i, float(gb[i]), float(sdy[i]), dy[:, i].tolist()))
