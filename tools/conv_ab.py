#!/usr/bin/env python3
"""In-process A/B timing of vinet_conv3d / vinet_conv3d_wgrad on real layer shapes.

    python tools/conv_ab.py [--lib PATH ...] [--sites fwd|all]

Each library given with --lib (default: the in-tree one) is dlopen'ed separately;
every site is timed with HIP events, interleaved over the variants (library x
option), median of several rounds.  This is the tool behind the tile / pipeline
choices recorded in DESIGN.md -- never compare numbers from different runs."""
import argparse
import ctypes as C
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from vinet_amd import _lib as L

# (name, B,T,H,W, Cin, N, k, s, p)  -- ViNet-32 layers at 32x224x384, batch 8
SITES = [
    ("stem_t 64->64 7x1x1/2", 8, 32, 112, 192, 64, 64, (7, 1, 1), (2, 1, 1), (3, 0, 0)),
    ("b1.3s 64->192 1x3x3", 8, 16, 56, 96, 64, 192, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
    ("b1.3t 192->192 3x1x1", 8, 16, 56, 96, 192, 192, (3, 1, 1), (1, 1, 1), (1, 0, 0)),
    ("3c pw 256->128", 8, 16, 28, 48, 256, 128, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    ("snd1 pw 64->16 (B x2)", 8, 35281, 1, 1, 64, 16, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    ("3c entry 256->288 pw", 8, 16, 28, 48, 256, 288, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    ("3c dg 288->256 pw", 8, 16, 28, 48, 288, 256, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    ("4x entry 512->256 pw", 8, 8, 14, 24, 512, 256, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    ("3c s 128->192 1x3x3", 8, 16, 28, 48, 128, 192, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
    ("4f pw 528->256", 8, 8, 14, 24, 528, 256, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    ("5c s 192->384 1x3x3", 8, 4, 7, 12, 192, 384, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
    ("dec2 832->480 3x3x3/3", 8, 12, 14, 24, 832, 480, (3, 3, 3), (3, 1, 1), (0, 1, 1)),
    ("dec3 480->192 5x3x3/5", 8, 20, 28, 48, 480, 192, (5, 3, 3), (5, 1, 1), (0, 1, 1)),
    ("dec4 192->64 5x3x3/5", 8, 20, 56, 96, 192, 64, (5, 3, 3), (5, 1, 1), (0, 1, 1)),
    # dgrad phases of the T-strided decoder convs: stride-1 1x3x3 correlations, channels swapped
    ("dg dec1 832->1024", 8, 4, 7, 12, 832, 1024, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
    ("dg dec2 480->832 /ph", 8, 4, 14, 24, 480, 832, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
    ("dg dec3 192->480 /ph", 8, 4, 28, 48, 192, 480, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
    ("dg dec4 64->192 /ph", 8, 4, 56, 96, 64, 192, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
    ("dg 3c s 192->128", 8, 16, 28, 48, 192, 128, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
    ("dg 4x s 320->160", 8, 8, 14, 24, 320, 160, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
    ("dg 4x t 320->320", 8, 8, 14, 24, 320, 320, (3, 1, 1), (1, 1, 1), (1, 0, 0)),
    # 64-channel outputs at huge M: short K loops, epilogue / latency bound
    ("dg stem_t 64->64 4taps", 8, 16, 112, 192, 64, 64, (4, 1, 1), (1, 1, 1), (1, 0, 0)),
    ("dg b1.3s 192->64 1x3x3", 8, 16, 56, 96, 192, 64, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
    ("stem folded 32->64 7taps", 8, 32, 118, 100, 32, 64, (1, 7, 1), (1, 1, 1), (0, 0, 0)),
    ("3c t 192->192 3x1x1", 8, 16, 28, 48, 192, 192, (3, 1, 1), (1, 1, 1), (1, 0, 0)),
    ("3b t 128->128 3x1x1", 8, 16, 28, 48, 128, 128, (3, 1, 1), (1, 1, 1), (1, 0, 0)),
    ("4f t 320->320 3x1x1", 8, 8, 14, 24, 320, 320, (3, 1, 1), (1, 1, 1), (1, 0, 0)),
    ("4d t 256->256 3x1x1", 8, 8, 14, 24, 256, 256, (3, 1, 1), (1, 1, 1), (1, 0, 0)),
    ("3b s 96->128 1x3x3", 8, 16, 28, 48, 96, 128, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
    ("dg 3b s 128->96", 8, 16, 28, 48, 128, 96, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
    ("3c b2 32->96 1x3x3", 8, 16, 28, 48, 32, 96, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
    ("dec5 64->32 2x3x3/2", 8, 4, 112, 192, 64, 32, (2, 3, 3), (2, 1, 1), (0, 1, 1)),
    ("dg dec5 32->64 /ph", 8, 2, 112, 192, 32, 64, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
    # more pointwise layers and their data gradients (--pw)
    ("3b entry 192->176 pw", 8, 16, 28, 48, 192, 176, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    ("3b dg 176->192 pw", 8, 16, 28, 48, 176, 192, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    ("2b 64->64 pw", 8, 16, 56, 96, 64, 64, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    ("3c b3 256->64 pw", 8, 16, 28, 48, 256, 64, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    ("3c b3 dg 64->256 pw", 8, 16, 28, 48, 64, 256, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    ("3b b3 192->32 pw", 8, 16, 28, 48, 192, 32, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    ("3b b3 dg 32->192 pw", 8, 16, 28, 48, 32, 192, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    ("4b entry 480->304 pw", 8, 8, 14, 24, 480, 304, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    ("4b dg 304->480 pw", 8, 8, 14, 24, 304, 480, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    ("4c b3 512->64 pw", 8, 8, 14, 24, 512, 64, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    ("4c b3 dg 64->512 pw", 8, 8, 14, 24, 64, 512, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    ("4f entry 528->448 pw", 8, 8, 14, 24, 528, 448, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    ("4c entry 512->296 pw", 8, 8, 14, 24, 512, 296, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    ("5b entry 832->624 pw", 8, 4, 7, 12, 832, 624, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    ("5c entry 832->448 pw", 8, 4, 7, 12, 832, 448, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    ("5b b3 832->128 pw", 8, 4, 7, 12, 832, 128, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
]


def bind(path):
    lib = C.CDLL(path)
    for name, argtypes in L.SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = L._RESTYPE.get(name, C.c_int)
    return lib


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", action="append")
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--wgrad", action="store_true")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--stats", action="store_true", help="forward convs also write BN statistics partials (training epilogue)")
    ap.add_argument("--only", default="", help="substring filter on site names")
    ap.add_argument("--ht", action="store_true", help="A/B of the halo-tile kernel (conv_ht.h) against the default dispatch on the 3x3-spatial sites")
    ap.add_argument("--pw", action="store_true", help="A/B of the pointwise streaming kernel (conv_pw.h) against the default dispatch on the 1x1x1 sites")
    ap.add_argument("--opt", action="append", default=[], help="name=v0,v1,...: A/B of one library option on the library's own dispatch (one variant per value)")
    ap.add_argument("--default", action="store_true", help="one variant per library: its own dispatch, no option overrides (A/B of two builds)")
    ap.add_argument("--tm", action="store_true", help="(3,1,1) temporal sites: the halo-tile temporal mode against the ping-pong GEMM kernel (taps re-staged, no halo reuse) and conv_dma")
    ap.add_argument("--acc", action="store_true", help="accumulate into y (the epilogue of a data gradient that joins an existing gradient)")
    ap.add_argument("--f32s", action="store_true", help="the split-bf16 form (VINET_F32S): fp32 tensors, hi / lo weight planes from vinet_pack_weights; TF/s = useful (one third of the MFMA rate)")
    args = ap.parse_args()
    libs = [(os.path.basename(p), bind(p)) for p in (args.lib or [L.LIB_PATH])]
    libs_all = [lib for _, lib in libs]
    dev = torch.device("cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    variants = []
    if args.wgrad:
        for ln, lib in libs:
            variants.append((ln + ":wpp256", lib, dict(wgrad_dma=1, wgrad_tg=0, wgrad_pp=3, tperm=1), False))
            variants.append((ln + ":wpp192", lib, dict(wgrad_dma=1, wgrad_tg=0, wgrad_pp=4, tperm=1), False))
            variants.append((ln + ":wpp192+pre", lib, dict(wgrad_dma=1, wgrad_tg=0, wgrad_pp=4, tperm=1), True))
            variants.append((ln + ":wdma", lib, dict(wgrad_dma=1, wgrad_tg=0, wgrad_pp=0, tperm=1), False))
            variants.append((ln + ":wdma-noperm", lib, dict(wgrad_dma=1, wgrad_tg=0, wgrad_pp=0, tperm=0), False))
            variants.append((ln + ":wdma+pre", lib, dict(wgrad_dma=1, wgrad_tg=0, wgrad_pp=0, tperm=1), True))
        libs = []
    if args.opt:
        for ln, lib in libs:
            for o in args.opt:
                name, vals = o.split("=")
                for v in vals.split(","):
                    variants.append((ln + ":%s=%s" % (name, v), lib, {name: int(v)}, False))
        libs = []
    if args.default:
        for ln, lib in libs:
            variants.append((ln + ":default", lib, dict(), False))
        libs = []
    if args.pw:
        for ln, lib in libs:
            variants.append((ln + ":default", lib, dict(pw=0), False))
            variants.append((ln + ":pw", lib, dict(pw=2, pw_maxtn=99), False))
            variants.append((ln + ":default+pre", lib, dict(pw=0), True))
            variants.append((ln + ":pw+pre", lib, dict(pw=2, pw_maxtn=99), True))
        libs = []
    if args.tm:
        for ln, lib in libs:
            variants.append((ln + ":ht_t", lib, dict(ht=1, ht_t=1, pp=1), False))
            variants.append((ln + ":pp192", lib, dict(ht=1, ht_t=0, pp=4), False))
            variants.append((ln + ":pp256", lib, dict(ht=1, ht_t=0, pp=3), False))
            variants.append((ln + ":dma", lib, dict(ht=1, ht_t=0, pp=0), False))
            variants.append((ln + ":ht_t+pre", lib, dict(ht=1, ht_t=1, pp=1), True))
            variants.append((ln + ":dma+pre", lib, dict(ht=1, ht_t=0, pp=0), True))
        libs = []
    if args.ht:
        for ln, lib in libs:
            variants.append((ln + ":default", lib, dict(ht=0), False))
            variants.append((ln + ":ht", lib, dict(ht=2), False))
            variants.append((ln + ":ht+pre", lib, dict(ht=2), True))
        libs = []
    for ln, lib in libs:
        variants.append((ln + ":pp256", lib, dict(dma=1, pp=3, tperm=0, n64_tile=0, n192_tile=0, n128_tile=0), False))
        variants.append((ln + ":pp192", lib, dict(dma=1, pp=4, tperm=0, n64_tile=0, n192_tile=0, n128_tile=0), False))
        variants.append((ln + ":dma", lib, dict(dma=1, pp=0, tperm=0, n64_tile=0, n192_tile=0, n128_tile=0), False))
        variants.append((ln + ":dma-n128a", lib, dict(dma=1, pp=0, tperm=0, n64_tile=0, n192_tile=1, n128_tile=1), False))
        variants.append((ln + ":dma-n128b", lib, dict(dma=1, pp=0, tperm=0, n64_tile=0, n192_tile=1, n128_tile=2), False))
        variants.append((ln + ":dma-n192", lib, dict(dma=1, pp=0, tperm=0, n64_tile=0, n192_tile=1, n128_tile=0), False))
        variants.append((ln + ":dma-n192+pre", lib, dict(dma=1, pp=0, tperm=0, n64_tile=0, n192_tile=1, n128_tile=0), True))
        variants.append((ln + ":dma-bm128", lib, dict(dma=1, pp=0, tperm=0, n64_tile=1, n192_tile=0, n128_tile=0), False))
        variants.append((ln + ":dma-bm64", lib, dict(dma=1, pp=0, tperm=0, n64_tile=2, n192_tile=0, n128_tile=0), False))
        variants.append((ln + ":dma+pre", lib, dict(dma=1, pp=0, tperm=0, n64_tile=0, n192_tile=0, n128_tile=0), True))

    print("%-26s" % "site" + "".join("%22s" % (v[0][-21:].replace("libvinet_hip", "")) for v in variants) + "   (ms | TF/s)")
    for (name, B, T, H, W, Cin, N, k, s, p) in SITES:
        if args.only and args.only not in name:
            continue
        if args.pw and k != (1, 1, 1):
            continue
        if args.ht and not ((k[1:] == (3, 3) and W % 16 == 0) or (k == (3, 1, 1) and (H * W) % 16 == 0)):
            continue
        if args.tm and k != (3, 1, 1):
            continue
        B = args.batch
        oT, oH, oW = [(d + 2 * pp - kk) // ss + 1 for d, kk, ss, pp in zip((T, H, W), k, s, p)]
        ntaps = k[0] * k[1] * k[2]
        Kp = (Cin + 31) // 32 * 32
        if args.f32s:
            x = torch.randn(B * T * H * W * Cin, device=dev)
            y = torch.empty(B * oT * oH * oW * N, device=dev, dtype=torch.float32)
            w = torch.empty(ntaps * N * Kp, device=dev, dtype=torch.float32)
            wm = torch.randn(N * Cin * ntaps, device=dev) * 0.05
            rc = libs_all[0].vinet_pack_weights(wm.data_ptr(), N, Cin, ntaps, 0, 0, L.F32S, w.data_ptr(), stream)
            assert rc == 0, libs_all[0].vinet_last_error()
        else:
            x = torch.randn(B * T * H * W * Cin, device=dev).bfloat16()
            y = torch.empty(B * oT * oH * oW * N, device=dev, dtype=torch.bfloat16)
            w = (torch.randn(ntaps * N * Kp, device=dev) * 0.05).bfloat16()
        taps = torch.tensor([(a - p[0], b - p[1], c - p[2], (a * k[1] + b) * k[2] + c) for a in range(k[0]) for b in range(k[1]) for c in range(k[2])],
                            dtype=torch.int32, device=dev)
        sc, sh = torch.rand(Cin, device=dev) + 0.5, torch.randn(Cin, device=dev)
        dw = torch.zeros(ntaps * N * Kp, device=dev)
        flops = 2.0 * B * oT * oH * oW * N * Cin * ntaps
        stats = torch.zeros(((B * oT * oH * oW + 63) // 64 + 8 * B * oT) * 2 * N, device=dev)

        def desc(pre):
            d = L.CConvDesc()
            d.dtype = d.out_dtype = L.BF16
            if args.f32s:
                d.dtype, d.out_dtype = L.F32S, L.F32
            d.mode = 0
            d.x = L.CTensor(x.data_ptr(), B, T, H, W, Cin, Cin, T * H * W * Cin)
            d.y = L.CTensor(y.data_ptr(), B, oT, oH, oW, N, N, oT * oH * oW * N)
            d.oT, d.oH, d.oW = oT, oH, oW
            d.sT, d.sH, d.sW = s
            d.omT = d.omH = d.omW = 1
            d.ntaps, d.taps, d.w, d.Kp = ntaps, taps.data_ptr(), w.data_ptr(), Kp
            d.pre = L.CAffine(sc.data_ptr(), sh.data_ptr(), 1) if pre else L.CAffine(None, None, 0)
            if args.stats:
                d.stats = stats.data_ptr()
            if args.acc:
                d.accumulate = 1
            if k == (1, 1, 1):
                d.tline = 6
            elif k[1:] == (3, 3):
                d.tline = 5
            elif k[1:] == (1, 1) and k[0] > 1:
                d.tline, d.tpad = 1, p[0]
            return d

        def wdesc(pre):
            d = L.CWgradDesc()
            d.dtype, d.mode = L.BF16, 0
            d.x = L.CTensor(x.data_ptr(), B, T, H, W, Cin, Cin, T * H * W * Cin)
            d.dy = L.CTensor(y.data_ptr(), B, oT, oH, oW, N, N, oT * oH * oW * N)
            d.sT, d.sH, d.sW = s
            d.ntaps, d.taps, d.dw, d.Kp = ntaps, taps.data_ptr(), dw.data_ptr(), Kp
            d.pre = L.CAffine(sc.data_ptr(), sh.data_ptr(), 1) if pre else L.CAffine(None, None, 0)
            if k[1:] == (3, 3) and s == (k[0], 1, 1) and p == (0, 1, 1):
                d.tline = 4          # kT x 3 x 3 / (kT,1,1) tap order promised: the row-streaming kernels (wgrad_rs.hip) are eligible
            elif k[1:] == (1, 1) and k[0] > 1:
                d.tline, d.tpad = 1, p[0]
            return d

        times = [[] for _ in variants]
        for r in range(args.rounds + 1):
            for vi, (vn, lib, opts, pre) in enumerate(variants):
                for kname, val in opts.items():
                    lib.vinet_set_option(kname.encode(), val)
                d = wdesc(pre) if args.wgrad else desc(pre)
                fn = lib.vinet_conv3d_wgrad if args.wgrad else lib.vinet_conv3d
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.iters):
                    rc = fn(C.byref(d), stream)
                    assert rc == 0, lib.vinet_last_error()
                e1.record()
                torch.cuda.synchronize()
                if r > 0:
                    times[vi].append(e0.elapsed_time(e1) / args.iters)
        row = "%-26s" % name
        for t in times:
            m = statistics.median(t)
            row += "%12.3f |%7.1f " % (m, flops / m / 1e9)
        print(row, flush=True)


if __name__ == "__main__":
    main()
