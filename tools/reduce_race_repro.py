#!/usr/bin/env python3
"""Minimal stand-alone reproduction attempt of the transient defect of `vinet_channel_sum` on a 6-voxel x 1024-channel bf16 tensor
(tools/reduce_dup_check.py finds it inside a training step: 33 of 240 pairs).  No engine, no model:

    main stream :  repeat { refresh dy from a master copy (a tiny kernel, as the step's producer) ; channel_sum(dy) -> slot 2i ;
                            channel_sum(dy) -> slot 2i + 1 }
    second stream: a co-runner that keeps the chip busy for the whole time:
                   --co=wgrad  the library's weight-gradient kernels on a decoder layer (what the step runs there)
                   --co=conv   the library's forward conv on the same layer
                   --co=gemm   rocBLAS (torch.matmul)
                   --co=copy   a large device-to-device copy
                   --co=atomics / stores / ratomics   synthetic kernels of tools/ubench/vmem_return_probe.hip: fp32 global atomics
                               without return, plain stores to the same addresses, atomics with return
                   --co=ldstr / lds   synthetic: LDS transpose reads (ds_read_b64_tr_b16) / plain ds_read_b64 in a loop
                   --co=mfma   synthetic: MFMAs on registers
                   --co=none   nothing

    python tools/reduce_race_repro.py [--co=wgrad] [--iters=4000] [--cfg=lib.reduce_small=0] [--voxels=6] [--channels=1024]

Every slot pair is compared bit for bit at the end (the kernel has no atomics: identical launches must return identical bits), and
every slot against the exact sums computed by torch in fp64 of the bf16 inputs.
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from vinet_amd import _lib as L, engine as E

opts = dict(co="wgrad", iters="4000", cfg="lib.reduce_small=0", voxels="6", channels="1024", batch="2")
for a in sys.argv[1:]:
    if a.startswith("--") and "=" in a:
        k, v = a[2:].split("=", 1)
        opts[k] = v
lib = L.load()
if opts["cfg"]:
    E.configure_from_string(opts["cfg"])
dev = torch.device("cuda:0")
nv, Cc, iters = int(opts["voxels"]), int(opts["channels"]), int(opts["iters"])
g = torch.Generator(device=dev).manual_seed(3)
master = (torch.randn(nv, Cc, generator=g, device=dev) * 1e-3).bfloat16()
dy = torch.empty_like(master)
dyv = E.View(dy.view(-1), 0, 2, nv // 2, 1, 1, Cc, Cc, (nv // 2) * Cc, E.BF16)      # (2 clips x voxels / 2 frames, as SoundNet's tail)
rows = lib.vinet_stats_rows(C.byref(dyv.ct()))
n = rows * 2 * Cc
slots = torch.empty(2 * iters, n, device=dev)
out = torch.empty(Cc, device=dev)

# ---- the co-runner ---------------------------------------------------------------------------------------------------------
co = opts["co"]
side = torch.cuda.Stream()
B = int(opts["batch"])
T, H, W, Cin, N, k, s, p = 20, 28, 48, 480, 192, (5, 3, 3), (5, 1, 1), (0, 1, 1)          # the decoder's 480 -> 192 layer
oT, oH, oW = [(d + 2 * pp - kk) // ss + 1 for d, kk, ss, pp in zip((T, H, W), k, s, p)]
xa = torch.randn(B * T * H * W * Cin, device=dev).bfloat16()
ya = torch.randn(B * oT * oH * oW * N, device=dev).bfloat16()
ntaps = k[0] * k[1] * k[2]
Kp = (Cin + 31) // 32 * 32
wa = (torch.randn(ntaps * N * Kp, device=dev) * 0.05).bfloat16()
dw = torch.zeros(ntaps * N * Kp, device=dev)
taps = torch.tensor([(a - p[0], b - p[1], c - p[2], (a * k[1] + b) * k[2] + c) for a in range(k[0]) for b in range(k[1]) for c in range(k[2])],
                    dtype=torch.int32, device=dev)
wd = L.CWgradDesc()
wd.dtype, wd.mode = L.BF16, 0
wd.x = L.CTensor(xa.data_ptr(), B, T, H, W, Cin, Cin, T * H * W * Cin)
wd.dy = L.CTensor(ya.data_ptr(), B, oT, oH, oW, N, N, oT * oH * oW * N)
wd.sT, wd.sH, wd.sW = s
wd.ntaps, wd.taps, wd.dw, wd.Kp = ntaps, taps.data_ptr(), dw.data_ptr(), Kp
wd.max_cus = 208
cd = L.CConvDesc()
cd.dtype = cd.out_dtype = L.BF16
cd.x, cd.y = wd.x, wd.dy
cd.oT, cd.oH, cd.oW = oT, oH, oW
cd.sT, cd.sH, cd.sW = s
cd.omT = cd.omH = cd.omW = 1
cd.ntaps, cd.taps, cd.w, cd.Kp, cd.tline = ntaps, taps.data_ptr(), wa.data_ptr(), Kp, 5
ga = torch.randn(4096, 4096, device=dev, dtype=torch.bfloat16)
big_a, big_b = torch.empty(1 << 28, device=dev, dtype=torch.uint8), torch.empty(1 << 28, device=dev, dtype=torch.uint8)


_pl = None


def probe_lib():
    global _pl
    if _pl is None:
        _pl = C.CDLL(os.path.join(ROOT, "tools", "ubench", "libvmem_return_probe.so"))
        _pl.probe_fill.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
        _pl.probe_launch.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
        _pl.corun_launch.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_void_p]
        _pl.corun_lds_launch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        _pl.corun_mfma_launch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        _pl.pk_probe_launch.argtypes = [C.c_int, C.c_uint32, C.c_void_p, C.c_void_p]
        _pl.pk_variant_launch.argtypes = [C.c_int, C.c_int, C.c_uint32, C.c_void_p, C.c_void_p]
        _pl.seq_probe_launch.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_uint32, C.c_void_p, C.c_void_p]
        _pl.reduce_victim_launch.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_long, C.c_void_p, C.c_void_p]
    return _pl


cobuf = torch.zeros(1 << 22, device=dev)


def co_launch():
    st = side.cuda_stream
    if co == "mfma":                                  # synthetic: MFMAs on registers, few registers per wave
        assert probe_lib().corun_mfma_launch(cobuf.data_ptr(), 100000, 416, st) == 0
        return
    if co in ("ldstr", "lds"):                        # synthetic: LDS transpose reads (ds_read_b64_tr_b16) / plain LDS reads in a loop
        assert probe_lib().corun_lds_launch(cobuf.data_ptr(), 200000, 3 if co == "ldstr" else 4, 416, st) == 0
        return
    if co in ("atomics", "stores", "ratomics"):       # synthetic: 208 x 2 workgroups of fp32 atomics / plain stores / returning atomics
        assert probe_lib().corun_launch(cobuf.data_ptr(), cobuf.numel(), 2000, {"atomics": 0, "stores": 1, "ratomics": 2}[co], 416, st) == 0
        return
    if co == "wgrad":
        assert lib.vinet_conv3d_wgrad(C.byref(wd), st) == 0, lib.vinet_last_error()
    elif co == "conv":
        assert lib.vinet_conv3d(C.byref(cd), st) == 0, lib.vinet_last_error()
    elif co == "gemm":
        with torch.cuda.stream(side):
            ga @ ga
    elif co == "copy":
        with torch.cuda.stream(side):
            big_b.copy_(big_a)


if co == "wgrad":
    nm = C.create_string_buffer(256)
    lib.vinet_conv3d_wgrad_kernel_name(C.byref(wd), nm, 256)
    print("co-runner kernel:", nm.value.decode(), flush=True)

# ---- the loop: the host keeps both streams fed (a co-runner launch for every `per` pairs) ------------------------------------
main = torch.cuda.current_stream().cuda_stream
per = 8
VICTIM = int(opts.get("victim", "-1"))
if "--pkself" in sys.argv:
    # the swapped packed add in waves 0..3 of a 512-thread workgroup whose waves 4..7 issue MFMAs: same kernel, same SIMDs
    pl = probe_lib()
    pl.pk_self_launch.argtypes = [C.c_int, C.c_uint32, C.c_void_p, C.c_void_p]
    pout = torch.zeros(16, dtype=torch.int32, device=dev)
    for i in range(iters):
        assert pl.pk_self_launch(4000, i, pout.data_ptr(), main) == 0
    torch.cuda.synchronize()
    o = [int(v) & 0xffffffff for v in pout.tolist()]
    print("v_pk_add_f32 op_sel:[0,1] op_sel_hi:[1,0] beside MFMA waves of its OWN workgroup: %d wrong of %.2e results" % (o[1], o[0] * 256 * 4000 * 2))
    pl.pk_grid_launch.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    pout = torch.zeros(16, dtype=torch.int32, device=dev)
    for i in range(max(1, iters // 100)):
        assert pl.pk_grid_launch(4000, 2048, pout.data_ptr(), main) == 0
    torch.cuda.synchronize()
    o = [int(v) & 0xffffffff for v in pout.tolist()]
    print("... beside MFMA workgroups of its OWN dispatch (odd / even workgroups of one launch): %d wrong of %.2e results" % (o[1], o[0] * 256 * 4000 * 2))
    # two dispatches of the SAME kernel on two queues: every workgroup of one multiplies, every workgroup of the other adds
    pl.pk_grid_launch2.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
    pout = torch.zeros(16, dtype=torch.int32, device=dev)
    for i in range(max(1, iters // 100)):
        assert pl.pk_grid_launch2(4000, 1024, pout.data_ptr(), side.cuda_stream, 1) == 0
        assert pl.pk_grid_launch2(4000, 1024, pout.data_ptr(), main, 0) == 0
    torch.cuda.synchronize()
    o = [int(v) & 0xffffffff for v in pout.tolist()]
    print("... beside MFMA workgroups of ANOTHER dispatch of the OWN kernel (same code and registers, second queue): %d wrong of %.2e results" % (o[1], o[0] * 256 * 4000 * 2))
    sys.exit(0)
if "--pkvariants" in sys.argv:
    # every operand selection of the packed fp32 instructions the library contains, on registers, beside the co-runner
    pl = probe_lib()
    names = ["v_pk_add_f32 (no op_sel)", "v_pk_add_f32 op_sel:[0,1] op_sel_hi:[1,0] (src1 halves swapped)", "v_pk_add_f32 op_sel_hi:[1,0] (src1.lo to both)",
             "v_pk_add_f32 op_sel:[0,1] op_sel_hi:[1,1] (src1.hi to both)", "v_pk_add_f32 op_sel:[1,0] op_sel_hi:[0,1] (src0 halves swapped)",
             "v_pk_mul_f32 op_sel:[0,1] op_sel_hi:[1,0] (src1 halves swapped)", "v_pk_mul_f32 op_sel_hi:[0,1] (src0.lo to both)", "v_pk_mul_f32 op_sel:[1,0] (src0.hi to both)"]
    for which, nm in enumerate(names):
        pout = torch.zeros(16, dtype=torch.int32, device=dev)
        for i in range(iters):
            if co != "none" and i % per == 0:
                co_launch()
            assert pl.pk_variant_launch(which, 2000, i, pout.data_ptr(), main) == 0
        torch.cuda.synchronize()
        o = [int(v) & 0xffffffff for v in pout.tolist()]
        extra = ""
        if o[1]:
            extra = "   first: lane %d, %s half: got %.4f, expected %.4f" % (o[4], "high" if o[6] else "low", torch.tensor([o[7]], dtype=torch.int64).to(torch.int32).view(torch.float32).item(),
                                                                          torch.tensor([o[8]], dtype=torch.int64).to(torch.int32).view(torch.float32).item())
        print("%-72s beside %s: %d wrong of %.2e results%s" % (nm, co, o[1], o[0] * 256 * 2000 * 2, extra), flush=True)
    sys.exit(0)
if "seq" in opts:
    # instruction-sequence probes (tools/ubench/vmem_return_probe.hip, seq_probe_kernel<mode>)
    pl = probe_lib()
    mode = int(opts["seq"])
    sbuf = torch.randint(0, 1 << 31, (65536,), dtype=torch.int32, device=dev)
    pout = torch.zeros(16, dtype=torch.int32, device=dev)
    for i in range(iters):
        if co != "none" and i % per == 0:
            co_launch()
        assert pl.seq_probe_launch(mode, sbuf.data_ptr(), 64, i, pout.data_ptr(), main) == 0
    torch.cuda.synchronize()
    o = [int(v) & 0xffffffff for v in pout.tolist()]
    print("sequence probe %d beside co-runner %s: %d launches x 256 lanes x 64 repetitions: %d wrong values" % (mode, co, o[0], o[1]))
    if o[1]:
        print("   first: lane %d repetition %d slot %d: got 0x%08x, expected 0x%08x (launch %d)" % (o[4], o[5], o[6], o[7], o[8], o[9]))
    sys.exit(0)
if "--pkprobe" in sys.argv:
    # a VALU-only victim: v_pk_add_f32 / v_pk_mul_f32 on registers against the scalar results, 2000 pairs per lane and launch
    pl = probe_lib()
    pout = torch.zeros(16, dtype=torch.int32, device=dev)
    for i in range(iters):
        if co != "none" and i % per == 0:
            co_launch()
        assert pl.pk_probe_launch(2000, i, pout.data_ptr(), main) == 0
    torch.cuda.synchronize()
    o = [int(v) & 0xffffffff for v in pout.tolist()]
    print("packed-fp32 probe beside co-runner %s: %d launches x 256 lanes x 2000 (add, mul) pairs: %d wrong adds, %d wrong multiplies" % (co, o[0], o[1], o[2]))
    sys.exit(0)
if "--probe" in sys.argv:
    # the instruction-level probe instead of the library's kernel (tools/ubench/vmem_return_probe.hip: load, s_waitcnt vmcnt(0), copy
    # the four destination registers at once and again ~150 cycles later)
    pl = probe_lib()
    pbuf = torch.zeros(1024, dtype=torch.int32, device=dev)
    pout = torch.zeros(16, dtype=torch.int32, device=dev)
    for i in range(iters):
        if co != "none" and i % per == 0:
            co_launch()
        assert pl.probe_fill(pbuf.data_ptr(), i, main) == 0
        for j in range(2):
            assert pl.probe_launch(pbuf.data_ptr(), i, 2 * i + j, pout.data_ptr(), main) == 0
    torch.cuda.synchronize()
    o = [int(v) & 0xffffffff for v in pout.tolist()]
    print("probe beside co-runner %s: %d launches, dwords wrong right behind s_waitcnt vmcnt(0): %d, still wrong ~150 cycles later: %d" % (co, o[0], o[1], o[2]))
    if o[1]:
        print("   first: lane %d dword %d: early 0x%08x, late 0x%08x, expected 0x%08x (launch %d; 0xdeadbeef = the register's content before the load)" % (
            o[4], o[5], o[6], o[7], o[8], o[9]))
    sys.exit(0)
for i in range(iters):
    if co != "none" and i % per == 0:
        co_launch()
    dy.copy_(master)                                    # the producer of the tensor, as in the step
    for j in range(2):
        if VICTIM >= 0:       # the stand-alone restatement of the kernel (tools/ubench/vmem_return_probe.hip), variant VICTIM
            assert probe_lib().reduce_victim_launch(VICTIM, dy.data_ptr(), Cc, nv, slots[2 * i + j].data_ptr(), main) == 0
            continue
        rc = lib.vinet_channel_sum(C.byref(dyv.ct()), E.BF16, slots[2 * i + j].data_ptr(), Cc, out.data_ptr(), 0, main)
        assert rc == 0, lib.vinet_last_error()
torch.cuda.synchronize()
ref = master.float().double().sum(0)
pairs_bad = int((slots[0::2] != slots[1::2]).any(dim=1).sum())
s_all = slots.view(2 * iters, rows, 2, Cc)[:, :, 0, :].double().sum(1)
wrong = ((s_all - ref).abs() > 1e-9).any(dim=1)
print("co-runner %s, %d voxels x %d channels, %s: %d of %d pairs differ, %d of %d launches have a wrong sum" % (
    co, nv, Cc, E.config(changed_only=True), pairs_bad, iters, int(wrong.sum()), 2 * iters))
if int(wrong.sum()):
    i = int(wrong.nonzero()[0])
    d = (s_all[i] - ref).abs()
    idx = (d > 1e-9).nonzero().flatten()
    print("   first wrong launch %d: %d channels off: %s ... (sum %.9g, exact %.9g; column %s)" % (
        i, idx.numel(), idx[:20].tolist(), float(s_all[i][idx[0]]), float(ref[idx[0]]), master[:, idx[0]].float().tolist()))
