#!/usr/bin/env python3
"""Per-phase cycle budget of conv_pp_kernel from a -DVINET_CONV_TIMING build (s_memtime):
mean cycles per phase spent in [reads+issue+vmcnt] [barrier X] [lgkm+MFMA] [barrier Y],
for the upper (g0) and lower (g1) wave group.  Tuning tool."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from tools.conv_ab import SITES, bind
from vinet_amd import _lib as L

lib = bind(os.path.join(ROOT, "vinet_amd", "libvinet_hip_timing.so"))
lib.vinet_set_option(b"pp", 2)
dev = torch.device("cuda:0")
stream = torch.cuda.current_stream().cuda_stream
Bsz = int(sys.argv[1]) if len(sys.argv) > 1 else 32
for (name, B, T, H, W, Cin, N, k, s, p) in SITES:
    B = Bsz
    oT, oH, oW = [(d + 2 * pp - kk) // ss + 1 for d, kk, ss, pp in zip((T, H, W), k, s, p)]
    x = torch.randn(B * T * H * W * Cin, device=dev).bfloat16()
    y = torch.empty(B * oT * oH * oW * N, device=dev, dtype=torch.bfloat16)
    ntaps = k[0] * k[1] * k[2]
    Kp = (Cin + 31) // 32 * 32
    w = (torch.randn(ntaps * N * Kp, device=dev) * 0.05).bfloat16()
    taps = torch.tensor([(a - p[0], b - p[1], c - p[2], (a * k[1] + b) * k[2] + c) for a in range(k[0]) for b in range(k[1]) for c in range(k[2])], dtype=torch.int32, device=dev)
    dbg = torch.zeros(8 * 65536, device=dev)
    ones = torch.ones(8 * 65536, device=dev)
    d = L.CConvDesc()
    d.dtype = d.out_dtype = L.BF16
    d.x = L.CTensor(x.data_ptr(), B, T, H, W, Cin, Cin, T * H * W * Cin)
    d.y = L.CTensor(y.data_ptr(), B, oT, oH, oW, N, N, oT * oH * oW * N)
    d.oT, d.oH, d.oW = oT, oH, oW
    d.sT, d.sH, d.sW = s
    d.omT = d.omH = d.omW = 1
    d.ntaps, d.taps, d.w, d.Kp = ntaps, taps.data_ptr(), w.data_ptr(), Kp
    d.out_shift, d.out_scale = dbg.data_ptr(), ones.data_ptr()
    for _ in range(2):
        assert lib.vinet_conv3d(C.byref(d), stream) == 0, lib.vinet_last_error()
    torch.cuda.synchronize()
    dbg.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); lib.vinet_conv3d(C.byref(d), stream); e1.record(); torch.cuda.synchronize()
    nb = ((B * oT * oH * oW + 255) // 256) * ((N + 255) // 256)
    t = dbg[:nb * 8].view(nb, 2, 4).cpu()
    g0, g1 = t[:, 0].mean(0), t[:, 1].mean(0)
    print("%-26s blocks %5d  %.3f ms | g0: M %6.0f X %6.0f C %6.0f Y %6.0f | g1: M %6.0f X %6.0f C %6.0f Y %6.0f" % (
        name, nb, e0.elapsed_time(e1), *g0.tolist(), *g1.tolist()), flush=True)
