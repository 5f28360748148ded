#!/usr/bin/env python3
"""Stress of the BatchNorm-backward chain on a TINY tensor (SoundNet's last layer: 6 voxels x 1024 channels), where the run-to-run
mismatch of audionet.conv7.bias / batchnorm7.weight sits (tools/fork_soak.py): reduce -> finalize -> apply (in place) -> channel sum,
thousands of times, beside a second stream that keeps the chip busy; every result against torch."""
import ctypes as C
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from vinet_amd import _lib as L, engine as E

lib = L.load()
dev = torch.device("cuda:0")
B, T, Cc = 2, 3, 1024
n = B * T
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
busy = torch.cuda.Stream()
a = torch.randn(4096, 4096, device=dev, dtype=torch.bfloat16)
g = torch.Generator(device=dev).manual_seed(1)
bad = 0
# ASYNCHRONOUS form: `chunk` chains are queued back to back (no host synchronisation in between: the kernels of a chain start
# the moment their predecessor retires, as inside a training step), each with its own buffers; checked afterwards
chunk = 64
fwd_keep = []
for it0 in range(0, iters, chunk):
    jobs = []
    with torch.cuda.stream(busy):
        for _ in range(8):
            a @ a
    for it in range(it0, min(it0 + chunk, iters)):
        dz = (torch.randn(n, Cc, generator=g, device=dev) * 1e-3).bfloat16()
        z = torch.randn(n, Cc, generator=g, device=dev).bfloat16()
        sc = torch.rand(Cc, generator=g, device=dev) + 0.5
        sh = torch.randn(Cc, generator=g, device=dev) * 0.3
        mean = torch.randn(Cc, generator=g, device=dev) * 0.1
        invstd = torch.rand(Cc, generator=g, device=dev) + 0.5
        jobs.append([dz, z, sc, sh, mean, invstd])
    for j in jobs:
        dz, z, sc, sh, mean, invstd = j
        dzv = E.View(dz.view(-1).clone(), 0, B, T, 1, 1, Cc, Cc, T * Cc, E.BF16)
        zv = E.View(z.view(-1), 0, B, T, 1, 1, Cc, Cc, T * Cc, E.BF16)
        st = torch.cuda.current_stream().cuda_stream
        rows = lib.vinet_stats_rows(C.byref(dzv.ct()))
        ws = torch.empty(rows * 2 * Cc, device=dev)
        c1, c2 = torch.empty(Cc, device=dev), torch.empty(Cc, device=dev)
        flat = torch.zeros(3 * Cc + 7, device=dev)
        gb, dgm, dbt = flat[1:1 + Cc], flat[1 + Cc:1 + 2 * Cc], flat[1 + 2 * Cc:1 + 3 * Cc]
        fwd = L.CAffine(sc.data_ptr(), sh.data_ptr(), 1)
        assert lib.vinet_bn_bwd_reduce(C.byref(dzv.ct()), C.byref(zv.ct()), E.BF16, fwd, mean.data_ptr(), invstd.data_ptr(), ws.data_ptr(), st) == 0
        assert lib.vinet_bn_bwd_finalize(ws.data_ptr(), rows, Cc, 0, float(n), sc.data_ptr(), 1, dgm.data_ptr(), dbt.data_ptr(), invstd.data_ptr(), c1.data_ptr(), c2.data_ptr(), st) == 0
        assert lib.vinet_bn_bwd_apply(C.byref(dzv.ct()), C.byref(zv.ct()), E.BF16, fwd, mean.data_ptr(), invstd.data_ptr(), c1.data_ptr(), c2.data_ptr(), C.byref(dzv.ct()), st) == 0
        ws2 = torch.empty(rows * 2 * Cc, device=dev)
        assert lib.vinet_channel_sum(C.byref(dzv.ct()), E.BF16, ws2.data_ptr(), Cc, gb.data_ptr(), 1, st) == 0
        j.extend([dzv, flat, ws, ws2, c1, c2])
    torch.cuda.synchronize()
    for k, j in enumerate(jobs):
        dz, z, sc, sh, mean, invstd, dzv, flat = j[:8]
        gb, dgm, dbt = flat[1:1 + Cc], flat[1 + Cc:1 + 2 * Cc], flat[1 + 2 * Cc:1 + 3 * Cc]
        gate = (z.float() * sc + sh) > 0
        gm = dz.float() * gate
        xhat = (z.float() - mean) * invstd
        r_dg, r_db = (gm * xhat).double().sum(0), gm.double().sum(0)
        r_gb = dzv.buf.view(n, Cc).float().double().sum(0)
        e1 = float((dgm.double() - r_dg).abs().max()); e2 = float((dbt.double() - r_db).abs().max()); e3 = float((gb.double() - r_gb).abs().max())
        if e1 > 1e-6 or e2 > 1e-6 or e3 > 1e-6:
            bad += 1
            w = (gb.double() - r_gb).abs()
            print("iter %d: dgamma err %.3e dbeta err %.3e bias-sum err %.3e (indices %s)" % (it0 + k, e1, e2, e3, (w > 1e-6).nonzero().flatten()[:8].tolist()), flush=True)
print("bad iterations: %d of %d" % (bad, iters))
