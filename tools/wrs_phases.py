#!/usr/bin/env python3
"""Per-phase cycle budget of conv_wgrad_rsm_kernel (the row-streaming weight gradient of the decoder / 1x3x3 layers at W = 48, 24)
from a -DVINET_CONV_TIMING build:

    python -c "from vinet_amd import build; build.build_variant('timing', ['-DVINET_CONV_TIMING'])"
    python tools/wrs_phases.py [batch]

per wave and step: MFMA phase (global loads issued, 66 transpose reads, 54 MFMAs), wait at barrier 1, LDS write phase, wait at
barrier 2; per item: prologue (ring fill)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from tools.conv_ab import bind
from vinet_amd import _lib as L

lib = bind(os.path.join(ROOT, "vinet_amd", os.environ.get("WRS_LIB", "libvinet_hip_timing.so")))
lib.vinet_debug_wrs_ptr.argtypes = [C.c_void_p]
lib.vinet_debug_wrs_ptr.restype = None
if os.environ.get("WRS_OPT"):        # e.g. WRS_OPT=wgrad_rs4=0: the eight-wave kernels
    for kv in os.environ["WRS_OPT"].split(","):
        lib.vinet_set_option(kv.split("=")[0].encode(), int(kv.split("=")[1]))
dev = torch.device("cuda:0")
stream = torch.cuda.current_stream().cuda_stream
Bsz = int(sys.argv[1]) if len(sys.argv) > 1 else 64
SITES = [("dec3 480->192 5x3x3/5", 20, 28, 48, 480, 192, 5), ("3c s 128->192 1x3x3", 16, 28, 48, 128, 192, 1),
         ("3b s 96->128 1x3x3", 16, 28, 48, 96, 128, 1), ("4x s 160->320 1x3x3", 8, 14, 24, 160, 320, 1)]
for name, T, H, W, Cin, N, kT in SITES:
    B = Bsz
    oT = T // kT
    x = torch.randn(B * T * H * W * Cin, device=dev).bfloat16()
    y = torch.randn(B * oT * H * W * N, device=dev).bfloat16()
    ntaps = kT * 9
    Kp = (Cin + 31) // 32 * 32
    taps = torch.tensor([(a, b - 1, c - 1, (a * 3 + b) * 3 + c) for a in range(kT) for b in range(3) for c in range(3)], dtype=torch.int32, device=dev)
    dw = torch.zeros(ntaps * N * Kp, device=dev)
    dbg = torch.zeros(1024 * 8 * 8, device=dev)
    lib.vinet_debug_wrs_ptr(dbg.data_ptr())
    d = L.CWgradDesc()
    d.dtype, d.mode = L.BF16, 0
    d.x = L.CTensor(x.data_ptr(), B, T, H, W, Cin, Cin, T * H * W * Cin)
    d.dy = L.CTensor(y.data_ptr(), B, oT, H, W, N, N, oT * H * W * N)
    d.sT, d.sH, d.sW = kT, 1, 1
    d.ntaps, d.taps, d.dw, d.Kp = ntaps, taps.data_ptr(), dw.data_ptr(), Kp
    d.pre = L.CAffine(None, None, 0)
    d.tline, d.max_cus = 4, 256
    buf = C.create_string_buffer(96)
    lib.vinet_conv3d_wgrad_kernel_name(C.byref(d), buf, 96)
    for _ in range(2):
        assert lib.vinet_conv3d_wgrad(C.byref(d), stream) == 0, lib.vinet_last_error()
    torch.cuda.synchronize()
    dbg.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); lib.vinet_conv3d_wgrad(C.byref(d), stream); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    t = dbg.view(-1, 8).cpu()
    t = t[t[:, 6] > 0]
    flops = 2.0 * B * oT * H * W * N * Cin * ntaps
    m = t.mean(0)
    steps, items = float(m[6]), float(m[7])
    print("%-24s %-28s %.3f ms %6.0f TF/s | waves %4d  total %8.0f cyc | per item: prologue %6.0f | per step: mfma %6.0f  barrier1 %6.0f  write %5.0f  barrier2 %5.0f  (sum %6.0f; 54 MFMAs = 864 cyc of pipe per wave, 1728 per SIMD)"
          % (name, buf.value.decode(), ms, flops / ms / 1e9, t.shape[0], float(m[0]), float(m[1]) / items, float(m[2]) / steps, float(m[3]) / steps,
             float(m[4]) / steps, float(m[5]) / steps, float(m[2] + m[3] + m[4] + m[5]) / steps), flush=True)
    nw = 4 if os.environ.get("WRS_OPT", "") == "" else 8
    w = t.view(-1, nw, 8)[:, :, 2].mean(0) / steps if t.shape[0] % nw == 0 else None
    if w is not None:
        print("      mfma phase by wave:", " ".join("%6.0f" % v for v in w.tolist()))
