#!/usr/bin/env python3
"""Per-phase cycle counts of conv_dma_kernel (prologue / K loop / epilogue) from a
-DVINET_CONV_TIMING build: s_memtime stamps dumped per workgroup.  Tuning tool."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from tools.conv_ab import SITES, bind
from vinet_amd import _lib as L

lib = bind(os.path.join(ROOT, "vinet_amd", "libvinet_hip_timing.so"))
dev = torch.device("cuda:0")
stream = torch.cuda.current_stream().cuda_stream
print("%-26s %8s %9s %9s %9s %7s   (cycles per workgroup, mean; s_memtime ticks = 100 MHz? see ratio)" % ("site", "blocks", "prologue", "kloop", "epilogue", "chunks"))
ONLY = os.environ.get("AB_ONLY", "")
for (name, B, T, H, W, Cin, N, k, s, p) in SITES:
    if ONLY and ONLY not in name:
        continue
    B = int(os.environ.get("AB_BATCH", B))
    oT, oH, oW = [(d + 2 * pp - kk) // ss + 1 for d, kk, ss, pp in zip((T, H, W), k, s, p)]
    x = torch.randn(B * T * H * W * Cin, device=dev).bfloat16()
    y = torch.empty(B * oT * oH * oW * N, device=dev, dtype=torch.bfloat16)
    ntaps = k[0] * k[1] * k[2]
    Kp = (Cin + 31) // 32 * 32
    w = (torch.randn(ntaps * N * Kp, device=dev) * 0.05).bfloat16()
    taps = torch.tensor([(a - p[0], b - p[1], c - p[2], (a * k[1] + b) * k[2] + c) for a in range(k[0]) for b in range(k[1]) for c in range(k[2])], dtype=torch.int32, device=dev)
    dbg = torch.zeros(4 * 65536 * 4, device=dev)
    d = L.CConvDesc()
    d.dtype = d.out_dtype = L.BF16
    d.x = L.CTensor(x.data_ptr(), B, T, H, W, Cin, Cin, T * H * W * Cin)
    d.y = L.CTensor(y.data_ptr(), B, oT, oH, oW, N, N, oT * oH * oW * N)
    d.oT, d.oH, d.oW = oT, oH, oW
    d.sT, d.sH, d.sW = s
    d.omT = d.omH = d.omW = 1
    d.ntaps, d.taps, d.w, d.Kp = ntaps, taps.data_ptr(), w.data_ptr(), Kp
    d.out_shift = dbg.data_ptr()     # timing build: dump buffer (zeros, so the affine is a no-op)
    dbg2 = torch.ones(4 * 65536 * 4, device=dev)
    d.out_scale = dbg2.data_ptr()
    for _ in range(3):
        assert lib.vinet_conv3d(C.byref(d), stream) == 0, lib.vinet_last_error()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); lib.vinet_conv3d(C.byref(d), stream); e1.record(); torch.cuda.synchronize()
    t = dbg.view(-1, 4).cpu()
    nb = int((t[:, 3] > 0).sum())
    t = t[:nb]
    t2 = dbg2.view(-1, 4).cpu()[:nb]
    print("%-26s %8d %9.0f %9.0f %9.0f %7.0f   kernel %.3f ms | epilogue: lds-write %6.0f  read+store %6.0f  total %6.0f" % (
        name, nb, t[:, 0].mean(), t[:, 1].mean(), t[:, 2].mean(), t[0, 3], e0.elapsed_time(e1), t2[:, 0].mean(), t2[:, 1].mean(), t2[:, 2].mean()))
