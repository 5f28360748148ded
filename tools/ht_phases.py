#!/usr/bin/env python3
"""Per-phase cycle counts of conv_ht_kernel (prologue / K loop incl. halo staging / epilogue) from a
-DVINET_CONV_TIMING build: s_memtime stamps dumped per workgroup.  Tuning tool."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from tools.conv_ab import SITES, bind
from vinet_amd import _lib as L

lib = bind(os.path.join(ROOT, "vinet_amd", os.environ.get("VINET_TIMING_LIB", "libvinet_hip_timing.so")))
lib.vinet_set_option(b"ht", 2)
dev = torch.device("cuda:0")
stream = torch.cuda.current_stream().cuda_stream
Bn = int(sys.argv[1]) if len(sys.argv) > 1 else 64
STATS = os.environ.get("VINET_PHASES_STATS", "0") == "1"
print("%-26s %8s %9s %9s %9s %9s   (s_memtime ticks per workgroup, mean)" % ("site", "blocks", "prologue", "kloop", "epilogue", "halo-wait"))
for (name, B, T, H, W, Cin, N, k, s, p) in SITES:
    if not (k[1:] == (3, 3) and W % 16 == 0):
        continue
    B = Bn
    oT, oH, oW = [(d + 2 * pp - kk) // ss + 1 for d, kk, ss, pp in zip((T, H, W), k, s, p)]
    x = torch.randn(B * T * H * W * Cin, device=dev).bfloat16()
    y = torch.empty(B * oT * oH * oW * N, device=dev, dtype=torch.bfloat16)
    ntaps = k[0] * k[1] * k[2]
    Kp = (Cin + 31) // 32 * 32
    w = (torch.randn(ntaps * N * Kp, device=dev) * 0.05).bfloat16()
    taps = torch.tensor([(a - p[0], b - p[1], c - p[2], (a * k[1] + b) * k[2] + c) for a in range(k[0]) for b in range(k[1]) for c in range(k[2])], dtype=torch.int32, device=dev)
    nblk = 4 * 1024 * 1024
    dbg = torch.zeros(nblk * 4, device=dev)
    d = L.CConvDesc()
    d.dtype = d.out_dtype = L.BF16
    d.x = L.CTensor(x.data_ptr(), B, T, H, W, Cin, Cin, T * H * W * Cin)
    d.y = L.CTensor(y.data_ptr(), B, oT, oH, oW, N, N, oT * oH * oW * N)
    d.oT, d.oH, d.oW = oT, oH, oW
    d.sT, d.sH, d.sW = s
    d.omT = d.omH = d.omW = 1
    d.ntaps, d.taps, d.w, d.Kp = ntaps, taps.data_ptr(), w.data_ptr(), Kp
    d.tline = 5
    d.out_shift = dbg.data_ptr()
    for _ in range(2):
        assert lib.vinet_conv3d(C.byref(d), stream) == 0, lib.vinet_last_error()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); lib.vinet_conv3d(C.byref(d), stream); e1.record(); torch.cuda.synchronize()
    t = dbg.view(-1, 4).cpu()
    nb = int((t[:, 1] > 0).sum())
    t = t[:nb]
    ms = e0.elapsed_time(e1)
    print("%-26s %8d %9.0f %9.0f %9.0f %9.0f   kernel %.3f ms  %.0f TF/s" % (
        name, nb, t[:, 0].mean(), t[:, 1].mean(), t[:, 2].mean(), t[:, 3].mean(), ms, 2.0 * B * oT * oH * oW * N * Cin * ntaps / ms / 1e9))
