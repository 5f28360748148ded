"""Where does the bf16 path's map error come from?  (tuning / analysis tool, not a test; uses the CPU oracle, so it lives under tests/)

    python tests/bf16_error_budget.py [8x96x192 8x128x192 16x64x96 48x64x96]

Emulates the bf16 pipeline on the oracle -- conv inputs and weights rounded to bf16, fp32 accumulation, activations stored
in bf16 behind every ReLU / upsample / pool -- against the golden map of tests/golden/e2e_*.npz, then switches groups of
layers back to fp32.  Round-3 result (profiles/r3_bf16_error_budget.txt): the emulation reproduces the GPU's bf16 error
(0.0099 vs 0.0096 measured at 8x96x192); keeping the decoder tail (convtsp4.3/.6/.8 + sigmoid, 3.5 of 229 GFLOP) in fp32 --
the cheap fix VERDICT r2 proposed -- changes max |err| by -11 % ... +19 % depending on the shape; the WHOLE decoder in fp32
leaves 0.0046 ... 0.0104, the whole encoder in fp32 0.0059 ... 0.0083: the error is spread over all ~80 layers and adds up like
a random walk, there is no cheap layer subset that buys north_star's 1e-3.  Only the fp32 path (bench.py: fp32_path) meets it.
"""
import os, sys, json, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import vinet_cpu as O
from tests import goldens as G
from vinet_amd import synth
torch.set_num_threads(8)
def rb(t): return t.to(torch.bfloat16).to(torch.float32)
def run(tag, keep_fp32=lambda name: False, out_fp32=lambda name: False, verbose=False):
    z, meta = G.load("e2e_"+tag)
    m = O.VideoSaliencyModel(num_clips=meta["clips"]).eval()
    m.load_state_dict(G.state_dict_for(m, meta["weight_seed"], z, meta))
    x = synth.clip(1, meta["clips"], meta["H"], meta["W"], meta["clip_seed"]).permute(0,2,1,3,4)
    # names of conv modules
    convs = {n:mod for n,mod in m.named_modules() if isinstance(mod, torch.nn.Conv3d)}
    bns = {n:mod for n,mod in m.named_modules() if isinstance(mod, torch.nn.BatchNorm3d)}
    hooks=[]
    saved={}
    for n,mod in convs.items():
        if keep_fp32(n): continue
        saved[n]=mod.weight.data.clone()
        mod.weight.data = rb(mod.weight.data)
        hooks.append(mod.register_forward_pre_hook(lambda md, inp: (rb(inp[0]),)))
    # outputs: relu outputs rounded: hook on ReLU modules and Upsample modules
    for n,mod in m.named_modules():
        if isinstance(mod,(torch.nn.ReLU, torch.nn.Upsample, torch.nn.MaxPool3d)):
            if out_fp32(n): continue
            hooks.append(mod.register_forward_hook(lambda md, inp, out: rb(out)))
    with torch.no_grad():
        y = m(x)
    for h in hooks: h.remove()
    for n,w in saved.items(): convs[n].weight.data = w
    ref = torch.as_tensor(z["y"])
    d = float((y-ref).abs().max())
    am = int(y.reshape(-1).argmax())
    return d, am==meta["argmax"], meta["top2_gap"]
tags = sys.argv[1:] or ["8x96x192","8x128x192","16x64x96","48x64x96"]
for tag in tags:
    print(tag, "all bf16:", run(tag))
    tail = lambda n: n.startswith("decoder.convtsp4") and n not in ("decoder.convtsp4.0",)
    print(tag, "tail(convtsp4.3/6/8) fp32 (conv inputs+weights):", run(tag, keep_fp32=tail))
    print(tag, "tail fp32 + tail outputs fp32:", run(tag, keep_fp32=tail, out_fp32=lambda n: n.startswith("decoder.convtsp4") and n not in ("decoder.convtsp4.1","decoder.convtsp4.2")))
    dec = lambda n: n.startswith("decoder")
    print(tag, "whole decoder fp32:", run(tag, keep_fp32=dec, out_fp32=dec))
    enc = lambda n: n.startswith("backbone")
    print(tag, "whole encoder fp32:", run(tag, keep_fp32=enc, out_fp32=enc))
