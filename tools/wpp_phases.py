#!/usr/bin/env python3
"""Per-phase cycle budget of conv_wgrad_pp_kernel from a -DVINET_CONV_TIMING build (see pp_phases.py)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from tools.conv_ab import SITES, bind
from vinet_amd import _lib as L

lib = bind(os.path.join(ROOT, "vinet_amd", "libvinet_hip_timing.so"))
lib.vinet_debug_wgrad_ptr.argtypes = [C.c_void_p]
lib.vinet_debug_wgrad_ptr.restype = None
shape = int(sys.argv[2]) if len(sys.argv) > 2 else 3
lib.vinet_set_option(b"wgrad_pp", shape)
dev = torch.device("cuda:0")
stream = torch.cuda.current_stream().cuda_stream
Bsz = int(sys.argv[1]) if len(sys.argv) > 1 else 32
for (name, B, T, H, W, Cin, N, k, s, p) in SITES:
    B = Bsz
    oT, oH, oW = [(d + 2 * pp - kk) // ss + 1 for d, kk, ss, pp in zip((T, H, W), k, s, p)]
    x = torch.randn(B * T * H * W * Cin, device=dev).bfloat16()
    y = torch.randn(B * oT * oH * oW * N, device=dev).bfloat16()
    ntaps = k[0] * k[1] * k[2]
    Kp = (Cin + 31) // 32 * 32
    taps = torch.tensor([(a - p[0], b - p[1], c - p[2], (a * k[1] + b) * k[2] + c) for a in range(k[0]) for b in range(k[1]) for c in range(k[2])], dtype=torch.int32, device=dev)
    dw = torch.zeros(ntaps * N * Kp, device=dev)
    sc, sh = torch.rand(Cin, device=dev) + 0.5, torch.randn(Cin, device=dev)
    dbg = torch.zeros(8 * 65536, device=dev)
    lib.vinet_debug_wgrad_ptr(dbg.data_ptr())
    for pre in (False, True):
        d = L.CWgradDesc()
        d.dtype, d.mode = L.BF16, 0
        d.x = L.CTensor(x.data_ptr(), B, T, H, W, Cin, Cin, T * H * W * Cin)
        d.dy = L.CTensor(y.data_ptr(), B, oT, oH, oW, N, N, oT * oH * oW * N)
        d.sT, d.sH, d.sW = s
        d.ntaps, d.taps, d.dw, d.Kp = ntaps, taps.data_ptr(), dw.data_ptr(), Kp
        d.pre = L.CAffine(sc.data_ptr(), sh.data_ptr(), 1) if pre else L.CAffine(None, None, 0)
        for _ in range(2):
            assert lib.vinet_conv3d_wgrad(C.byref(d), stream) == 0, lib.vinet_last_error()
        torch.cuda.synchronize()
        dbg.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); lib.vinet_conv3d_wgrad(C.byref(d), stream); e1.record(); torch.cuda.synchronize()
        t = dbg.view(-1, 2, 4).cpu()
        nb = int((t[:, 0, 2] > 0).sum())
        t = t[:nb]
        g0, g1 = t[:, 0].mean(0), t[:, 1].mean(0)
        print("%-24s %-5s blocks %5d  %.3f ms | g0: M %6.0f X %6.0f C %6.0f Y %6.0f | g1: M %6.0f X %6.0f C %6.0f Y %6.0f" % (
            name, "pre" if pre else "plain", nb, e0.elapsed_time(e1), *g0.tolist(), *g1.tolist()), flush=True)
