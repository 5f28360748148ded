#!/usr/bin/env python3
"""How far can a training gradient move when the FORWARD pass carries a rounding error?  (CPU, fp64 oracle, ~3 minutes on 24 cores;
lives under tests/ because it runs the oracle.  Output of the committed run: profiles/r5_gate_flip_law.txt)

    python tests/experiments/gate_flip_law.py [noise ...]          # default: 1e-5 1e-6 1e-7

The `train_step_wc` problem (ViNet-8, B = 12, 8 x 128 x 192: 288 samples per channel in the deepest BatchNorms) runs in fp64 twice:
clean, and with every Conv3d output perturbed by `noise * rms(output)` white noise -- forward only (the perturbation is detached, the
backward arithmetic stays exact fp64).  ReLU gates and max-pool argmaxes are discontinuous in the activations, so the gradient moves
with the SQUARE ROOT of the forward error: the fraction of flipped gates is proportional to the perturbation, each flip changes its
gradient entry by O(1), and the L2 norm of a sparse O(1) change goes with the root of its density.  Measured (see the profile):

    per-conv noise 1e-5  ->  prediction rel 1.5e-4  ->  whole gradient vector 16.9 %
    per-conv noise 1e-6  ->  prediction rel 1.5e-5  ->  whole gradient vector  5.3 %      (x 3.2 = sqrt(10) per decade)

i.e. gradient_rel ~ 14 * sqrt(prediction_rel) on this fixture.  The split-bf16 path (fp32s: 16 significant bits per operand) has a
prediction error of 6.3e-5 -> 11.0 % predicted, 10.9 % measured (tests/test_gpu_model.py::test_train_step_well_conditioned[fp32s],
tools/split_grad_probe.py); the reference's own fp32 gradient sits 2.2 % from fp64, which is what a forward error of 2.6e-6 gives.
A gate of "2 x the reference's fp32 error" therefore asks for a forward error <= 1e-5, ~ 20 significant bits per operand: exact fp32
arithmetic passes it (1.06 x), a 16-bit split cannot, whatever its kernels do.
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from oracle import vinet_cpu as O
from tests import goldens as G
from vinet_amd import synth

torch.set_num_threads(min(24, os.cpu_count() or 1))
z, meta = G.load("train_step_wc")
B, T, H, W = meta["B"], meta["T"], meta["H"], meta["W"]
x = synth.clip(B, T, H, W, meta["seed"]).permute(0, 2, 1, 3, 4).double()
gt = synth.gt_map(B, H, W, meta["seed"]).double()


def run(noise, seed=0):
    o = O.VideoSaliencyModel(num_clips=8)
    o.load_state_dict(G.state_dict_for(o, meta["seed"], z, meta))
    o = o.double().train()
    gen = torch.Generator().manual_seed(seed)
    if noise:
        for m in o.modules():
            if isinstance(m, torch.nn.Conv3d):
                m.register_forward_hook(lambda mod, i, out: out + (noise * out.detach().pow(2).mean().sqrt()) *
                                        torch.randn(out.shape, generator=gen, dtype=out.dtype))
    pred = o(x)
    O.kldiv(pred, gt).backward()
    return pred.detach(), {k: p.grad.clone() for k, p in o.named_parameters()}


def main():
    noises = [float(a) for a in sys.argv[1:]] or [1e-5, 1e-6, 1e-7]
    t = time.time()
    p0, g0 = run(0.0)
    print("clean fp64 step: %.0f s" % (time.time() - t), flush=True)
    den = sum(float(g0[k].pow(2).sum()) for k in g0)
    for noise in noises:
        p1, g1 = run(noise)
        num = sum(float((g1[k] - g0[k]).pow(2).sum()) for k in g0)
        rel = lambda k: float((g1[k] - g0[k]).norm() / g0[k].norm())
        pr = float((p1 - p0).norm() / p0.norm())
        print("per-conv noise %.0e: prediction rel %.3e   whole gradient vector rel L2 %.3e (= %.1f x sqrt(prediction rel))   "
              "decoder.convtsp4.3.weight %.3e   decoder.convtsp1.0.weight %.3e   backbone.base1.0.conv_s.weight %.3e" % (
                  noise, pr, (num / den) ** 0.5, (num / den) ** 0.5 / pr ** 0.5, rel("decoder.convtsp4.3.weight"),
                  rel("decoder.convtsp1.0.weight"), rel("backbone.base1.0.conv_s.weight")), flush=True)


if __name__ == "__main__":
    main()
