"""Every C-ABI entry point of libvinet_hip.so on a real MI355X against the CPU
model of the same contract (tests/abi_emulator.py), on identical descriptors and
seeded inputs.  fp32 must agree to fp32 round-off; bf16 to bf16 round-off."""
import ctypes as C
import math
import os

import numpy as np
import pytest
import torch

from tests.abi_emulator import AbiEmulator
from vinet_amd import _lib as L
from vinet_amd import engine as E
from vinet_amd import synth

pytestmark = pytest.mark.gpu

DTS = [E.F32, E.BF16]
TOL = {E.F32: 2e-5, E.BF16: 2e-2}


def _dev():
    return torch.device("cuda:0")


def _lib():
    return L.load()


def _stream():
    return torch.cuda.current_stream().cuda_stream


# ---- exact-arithmetic mode -----------------------------------------------------------------------------------------
# Inside `exact_mode()` every tensor the case runners draw holds SMALL INTEGERS (activations, gradients and weights in
# [-2, 2]; pending-affine / epilogue scales in {1, 2}, shifts in {-1, 0, 1}): exactly representable in bf16, every product
# and every partial sum (|.| <= 4 K <= 2^17) exact in fp32 in ANY summation order, the bf16 rounding of the exact result
# unique.  The kernels must then equal the ABI model BIT FOR BIT: a wrong tap, pad, stride phase, swizzle or a dropped
# K tile cannot hide under a 2e-2 tolerance.  (Per-tile sums of squares are not exact: statistics keep their tolerance.)
_EXACT = [False]


class exact_mode:
    def __enter__(self):
        _EXACT[0] = True

    def __exit__(self, *a):
        _EXACT[0] = False
        return False


def _ints(name, shape, seed, lo, hi):
    return torch.floor(synth.uniform(name, tuple(shape), seed, float(lo), float(hi) + 1.0)).clamp_(lo, hi)


def _rand(name, shape, seed=0, scale=1.0):
    if _EXACT[0]:
        return _ints(name, shape, seed, -2, 2)
    return synth.normal(name, shape, seed) * scale


class Pair:
    """the same tensor on CPU (for the emulator) and on the GPU (for the library)."""

    def __init__(self, t):
        self.cpu = t.contiguous().clone()
        self.gpu = self.cpu.to(_dev())

    def ptr(self, side):
        return (self.cpu if side == "cpu" else self.gpu).data_ptr()

    def get(self, side):
        return self.cpu if side == "cpu" else self.gpu.cpu()


def view_pair(B, T, H, W, Cc, dt, name, seed=0, ld=None, c_off=0, t_total=None, t_off=0, fill=None):
    """a View backed by a Pair; optionally a channel slice (ld > C) and/or T slice of a bigger buffer."""
    ld = Cc if ld is None else ld
    tt = T if t_total is None else t_total
    n = B * tt * H * W * ld
    base = _rand(name, (n,), seed) if fill is None else torch.full((n,), float(fill))
    base = base.to(E.TORCH_DT[dt])
    p = Pair(base)
    off = t_off * H * W * ld + c_off

    def mk(side):
        buf = p.cpu if side == "cpu" else p.gpu
        return E.View(buf, off, B, T, H, W, Cc, ld, tt * H * W * ld, dt)
    return p, mk


def fvec(name, n, seed=0, lo=None, hi=None):
    if _EXACT[0]:
        return Pair((_ints(name, (n,), seed, -1, 1) if lo is None else _ints(name, (n,), seed, 1, 2)).float())
    t = _rand(name, (n,), seed) if lo is None else synth.uniform(name, (n,), seed, lo, hi)
    return Pair(t.float())


def _cmp(a, b, tol, what=""):
    if _EXACT[0] and "stats" not in what:
        assert a.shape == b.shape and a.dtype == b.dtype, (what, a.shape, b.shape, a.dtype, b.dtype)
        if not torch.equal(a, b):
            bad = (a.float() != b.float()).nonzero()
            raise AssertionError("%s: NOT bit-identical on exact-arithmetic inputs: %d of %d elements differ, first at %s (got %s, expected %s)"
                                 % (what, bad.shape[0], a.numel(), bad[0].tolist(), a.float()[tuple(bad[0])].item(), b.float()[tuple(bad[0])].item()))
        return
    a, b = a.float(), b.float()
    scale = max(1.0, float(b.abs().max()))
    d = float((a - b).abs().max())
    assert d <= tol * scale, "%s: max abs diff %g (scale %g) > %g" % (what, d, scale, tol * scale)


def run_both(fn, make_args):
    """make_args(side) -> args list.  Runs the emulator on CPU args and the library on GPU args."""
    emu = AbiEmulator()
    rc = getattr(emu, fn)(*make_args("cpu"))
    assert rc == 0
    lib = _lib()
    rc = getattr(lib, fn)(*make_args("gpu"))
    if rc != 0:
        raise AssertionError("%s rc=%d: %s" % (fn, rc, lib.vinet_last_error().decode()))
    torch.cuda.synchronize()


# ----------------------------------------------------------------------------
def test_library_loads_and_abi():
    lib = _lib()
    assert lib.vinet_abi_version() == L.ABI_VERSION


def test_tr16_fragment_mapping():
    """ds_read_tr16_b64: lane l must receive tile[k = (l>>4)*8 + j][i = l&15], j = 0..7."""
    lib = _lib()
    fn = lib.vinet_selftest_tr16
    fn.argtypes = [C.c_void_p, C.c_void_p]
    fn.restype = C.c_int
    out = torch.zeros(64 * 8, dtype=torch.int16, device=_dev())
    assert fn(out.data_ptr(), _stream()) == 0
    torch.cuda.synchronize()
    got = out.cpu().view(64, 8).numpy().astype(np.int64)
    exp = np.array([[((l >> 4) * 8 + j) * 16 + (l & 15) for j in range(8)] for l in range(64)])
    if not (got == exp).all():
        bad = [(l, got[l].tolist(), exp[l].tolist()) for l in range(64) if (got[l] != exp[l]).any()][:6]
        raise AssertionError("tr16 mapping differs, e.g. (lane, got, expected): %s" % bad)


# ---- convolution ---------------------------------------------------------------
CONV_CASES = [
    # name, (B,T,H,W), Cin, N, k, s, p, extras
    ("pw_192_64", (2, 3, 7, 9), 192, 64, (1, 1, 1), (1, 1, 1), (0, 0, 0), {}),
    ("pw_pre_stats", (2, 3, 7, 9), 96, 48, (1, 1, 1), (1, 1, 1), (0, 0, 0), dict(pre=True, stats=True)),
    ("sp_3x3", (1, 2, 9, 11), 32, 96, (1, 3, 3), (1, 1, 1), (0, 1, 1), dict(pre=True)),
    ("tm_3x1", (1, 5, 6, 7), 64, 208, (3, 1, 1), (1, 1, 1), (1, 0, 0), dict(stats=True)),
    ("tm_7s2", (1, 9, 5, 6), 64, 64, (7, 1, 1), (2, 1, 1), (3, 0, 0), dict(epi=True, act=1)),
    ("dec_3x3x3", (2, 6, 5, 7), 64, 160, (3, 3, 3), (3, 1, 1), (0, 1, 1), dict(act=1)),
    ("dec_5x3x3_bigK", (1, 5, 4, 6), 480, 192, (5, 3, 3), (5, 1, 1), (0, 1, 1), dict(act=1)),
    ("small_n16", (1, 2, 8, 8), 192, 16, (1, 1, 1), (1, 1, 1), (0, 0, 0), {}),
    ("cin24", (1, 2, 6, 6), 24, 64, (1, 3, 3), (1, 1, 1), (0, 1, 1), dict(pre=True)),
    ("wide_n384", (1, 1, 7, 12), 832, 384, (1, 1, 1), (1, 1, 1), (0, 0, 0), dict(stats=True)),
    ("concat_slice_out", (1, 2, 6, 6), 64, 32, (1, 1, 1), (1, 1, 1), (0, 0, 0), dict(out_ld=96, out_coff=32, stats=True)),
    ("tslice_in", (2, 3, 5, 5), 32, 32, (3, 3, 3), (3, 1, 1), (0, 1, 1), dict(in_ttotal=5, in_toff=1)),
    ("accumulate", (1, 2, 6, 6), 64, 64, (1, 3, 3), (1, 1, 1), (0, 1, 1), dict(accumulate=True)),
    ("head_pad", (2, 1, 8, 12), 32, 1, (1, 1, 1), (1, 1, 1), (0, 0, 0), dict(epi_shift=True, act=2, out_f32=True, head=True)),
    ("big_m", (2, 4, 28, 48), 64, 64, (1, 1, 1), (1, 1, 1), (0, 0, 0), dict(stats=True)),
    # input = channel slice of a wider buffer (the fused Inception entry conv's reduce outputs, the block output
    # behind them): rows ld apart, first channel not on a row start
    ("xslice_pre", (1, 2, 9, 11), 96, 128, (1, 3, 3), (1, 1, 1), (0, 1, 1), dict(pre=True, stats=True, in_ld=368, in_coff=16)),
    ("xslice_pw", (2, 2, 6, 8), 256, 176, (1, 1, 1), (1, 1, 1), (0, 0, 0), dict(pre=True, stats=True, in_ld=368, in_coff=112,
                                                                              out_ld=480, out_coff=0)),
    ("xslice_plain", (1, 3, 6, 6), 64, 192, (3, 1, 1), (1, 1, 1), (1, 0, 0), dict(in_ld=104, in_coff=40, act=1)),
    # a stride phase of a data gradient: rows stored to every 3rd frame of a longer tensor (plain store: the 16-byte epilogue path)
    ("phase_store", (2, 4, 7, 9), 96, 160, (1, 3, 3), (1, 1, 1), (0, 1, 1), dict(om=(3, 1))),
]


def _fwd_taps(k, p):
    return [(kt - p[0], kh - p[1], kw - p[2], (kt * k[1] + kh) * k[2] + kw)
            for kt in range(k[0]) for kh in range(k[1]) for kw in range(k[2])]


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("case", CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv3d(case, dt):
    _run_conv_case(case, dt)


@pytest.mark.parametrize("case", CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv3d_split_bf16_arithmetic(case):
    """VINET_F32S: fp32 tensors, every operand split into hi + lo bf16 (16 significant bits), three bf16 MFMAs per product.
    Against the fp32 model: the products carry a relative error of ~2^-17, i.e. ~1e-5 on O(1) outputs (the exact-fp32 form
    holds 2e-5 on the same cases; bf16 2e-2)."""
    _run_conv_case(case, E.F32, cdt=L.F32S, tol=1e-4)


# the 256x256x64 ping-pong kernel, forced on shapes it would not normally be chosen for as well:
# partial tiles in M and N, N > 256, odd and even K-tile counts, Kp % 64 == 32, 45 taps, strides,
# T-sliced inputs, channel-sliced outputs, statistics, accumulate, fp32 head
PP_CASES = [c for c in CONV_CASES if not c[7].get("pre")] + [
    ("pp_multi_tile", (2, 4, 14, 24), 192, 480, (1, 3, 3), (1, 1, 1), (0, 1, 1), dict(stats=True)),
    ("pp_kp96_odd", (1, 3, 9, 10), 96, 272, (3, 1, 1), (1, 1, 1), (1, 0, 0), dict(act=1)),
    ("pp_45taps", (1, 10, 6, 8), 64, 320, (5, 3, 3), (5, 1, 1), (0, 1, 1), dict(stats=True, act=1)),
    ("pp_one_ktile", (1, 2, 20, 20), 64, 256, (1, 1, 1), (1, 1, 1), (0, 0, 0), {}),
    ("pp_stride2", (1, 9, 12, 12), 128, 128, (3, 3, 3), (2, 2, 2), (1, 1, 1), {}),
]


# the 128 x 192 tile of conv_dma (waves 2 x 2) on shapes it is not normally chosen for: partial row tiles, pending BN + ReLU,
# statistics, epilogues, sliced views, two column tiles (N = 384 pads to nothing here because 384 % 192 == 0)
N192_CASES = [
    ("n192_s", (2, 3, 10, 12), 64, 192, (1, 3, 3), (1, 1, 1), (0, 1, 1), dict(stats=True)),
    ("n192_t_pre", (1, 5, 9, 10), 192, 192, (3, 1, 1), (1, 1, 1), (1, 0, 0), dict(pre=True, stats=True, act=1)),
    ("n192_pw_slices", (2, 2, 7, 9), 96, 192, (1, 1, 1), (1, 1, 1), (0, 0, 0), dict(epi=True, act=1, in_ld=160, in_coff=32, out_ld=256, out_coff=32)),
    ("n192_acc", (1, 4, 8, 8), 128, 192, (1, 3, 3), (1, 1, 1), (0, 1, 1), dict(accumulate=True)),
]


@pytest.mark.parametrize("case", N192_CASES, ids=[c[0] for c in N192_CASES])
def test_conv3d_n192_tile(case):
    lib = _lib()
    assert lib.vinet_set_option(b"n192_tile", 2) == 0 and lib.vinet_set_option(b"pp", 0) == 0
    try:
        d0 = _run_conv_case(case, E.BF16, forced=True)
        buf = C.create_string_buffer(128)
        assert lib.vinet_conv3d_kernel_name(C.byref(d0), buf, 128) == 0 and buf.value.startswith(b"conv_dma_kernel<4,6,2,2"), buf.value
        assert lib.vinet_conv3d_tile_m(C.byref(d0)) == 128
    finally:
        lib.vinet_set_option(b"n192_tile", 1)
        lib.vinet_set_option(b"pp", 1)


@pytest.mark.parametrize("shape", [3, 4], ids=["bn256", "bn192"])
@pytest.mark.parametrize("case", PP_CASES, ids=[c[0] for c in PP_CASES])
def test_conv3d_pingpong(case, shape):
    lib = _lib()
    assert lib.vinet_set_option(b"pp", shape) == 0     # 3 / 4: force the 256- / 192-wide shape
    try:
        d0 = _run_conv_case(case, E.BF16, forced=True)
        buf = C.create_string_buffer(128)
        assert lib.vinet_conv3d_kernel_name(C.byref(d0), buf, 128) == 0
        assert buf.value == (b"conv_pp_kernel<256>" if shape == 3 else b"conv_pp_kernel<192>")
    finally:
        lib.vinet_set_option(b"pp", 1)


# the halo-tile kernel (conv_ht.h), forced: 32- and 16-wide tiles, partial row tiles (H not a multiple of 8 / 16), one /
# several temporal taps with and without temporal stride, 1 / 2 / 3 channel chunks incl. partial ones (Cin = 24, 96, 160),
# 32- / 64- / 96-wide column tiles with padded N, statistics, activations, accumulate into a strided stride-phase placement,
# channel- and T-sliced views on both sides
HT_CASES = [
    ("ht_64_192", (2, 2, 8, 32), 64, 192, (1, 3, 3), (1, 1, 1), (0, 1, 1), dict(stats=True)),
    ("ht_192_64_5t", (1, 5, 10, 32), 192, 64, (5, 3, 3), (5, 1, 1), (0, 1, 1), dict(act=1)),
    ("ht_tw16_rows20", (2, 3, 20, 16), 96, 128, (1, 3, 3), (1, 1, 1), (0, 1, 1), dict(stats=True, act=1)),
    ("ht_tw16_w48", (1, 2, 28, 48), 128, 192, (1, 3, 3), (1, 1, 1), (0, 1, 1), {}),
    ("ht_cin24_n32", (1, 4, 12, 64), 24, 32, (2, 3, 3), (2, 1, 1), (0, 1, 1), dict(act=1)),
    ("ht_cin160_n80", (1, 2, 9, 32), 160, 80, (1, 3, 3), (1, 1, 1), (0, 1, 1), dict(epi=True, act=1)),
    ("ht_3t_s1", (1, 4, 8, 32), 64, 64, (3, 3, 3), (1, 1, 1), (1, 1, 1), dict(stats=True)),
    ("ht_slices", (2, 2, 8, 32), 64, 96, (1, 3, 3), (1, 1, 1), (0, 1, 1), dict(in_ld=160, in_coff=32, out_ld=256, out_coff=64, stats=True)),
    ("ht_tslice_in", (2, 6, 8, 16), 32, 48, (3, 3, 3), (3, 1, 1), (0, 1, 1), dict(in_ttotal=8, in_toff=1)),
    ("ht_phase_acc", (2, 3, 8, 32), 64, 192, (1, 3, 3), (1, 1, 1), (0, 1, 1), dict(accumulate=True, om=(5, 2))),
    ("ht_phase_store", (2, 3, 12, 32), 64, 192, (1, 3, 3), (1, 1, 1), (0, 1, 1), dict(om=(5, 3))),
    ("ht_f32_out", (1, 1, 16, 32), 32, 8, (1, 3, 3), (1, 1, 1), (0, 1, 1), dict(out_f32=True, epi_shift=True)),
    # pending BatchNorm + ReLU applied once per staged element, in LDS (padding must stay zero behind the affine)
    ("ht_pre_64_96", (2, 2, 8, 32), 64, 96, (1, 3, 3), (1, 1, 1), (0, 1, 1), dict(pre=True, stats=True)),
    ("ht_pre_tw16_cin96", (1, 3, 20, 16), 96, 64, (3, 3, 3), (1, 1, 1), (1, 1, 1), dict(pre=True, act=1)),
    ("ht_pre_slices", (2, 2, 9, 32), 128, 192, (1, 3, 3), (1, 1, 1), (0, 1, 1), dict(pre=True, in_ld=160, in_coff=32, stats=True)),
    # temporal mode: (3,1,1) taps, 64 positions x 4 output frames per workgroup; frames / positions not multiples of 4 / 64,
    # plain and pending-affine inputs, accumulate (the data gradient joining an existing one), sliced views
    ("htt_192", (2, 6, 8, 16), 192, 192, (3, 1, 1), (1, 1, 1), (1, 0, 0), dict(tline=True, stats=True)),
    ("htt_pre_128", (1, 5, 14, 24), 128, 128, (3, 1, 1), (1, 1, 1), (1, 0, 0), dict(tline=True, pre=True, stats=True, act=1)),
    ("htt_acc_slices", (2, 7, 6, 8), 64, 96, (3, 1, 1), (1, 1, 1), (1, 0, 0), dict(tline=True, accumulate=True, in_ld=160, in_coff=32, out_ld=256, out_coff=64)),
    ("htt_pre_cin160", (1, 4, 8, 8), 160, 80, (3, 1, 1), (1, 1, 1), (1, 0, 0), dict(tline=True, pre=True, epi=True)),
    # round 6: a last half chunk (Cin = 96: the weight DMA's 64-bytes-back redirection) over five temporal groups, four column
    # tiles at TW = 16, temporal mode with a pending affine and partial tiles, accumulate into a wide slice
    ("ht_cin96_5t", (1, 5, 9, 32), 96, 192, (5, 3, 3), (5, 1, 1), (0, 1, 1), dict(stats=True, act=1)),
    ("ht_n384_tw16", (2, 1, 7, 16), 160, 384, (1, 3, 3), (1, 1, 1), (0, 1, 1), dict(epi=True)),
    ("htt_pre_192", (1, 5, 14, 24), 192, 192, (3, 1, 1), (1, 1, 1), (1, 0, 0), dict(tline=True, pre=True, stats=True, act=1)),
    ("htt_acc_slices_192", (2, 7, 6, 8), 64, 192, (3, 1, 1), (1, 1, 1), (1, 0, 0), dict(tline=True, accumulate=True, in_ld=160, in_coff=32, out_ld=448, out_coff=64)),
]


@pytest.mark.parametrize("case", HT_CASES, ids=[c[0] for c in HT_CASES])
def test_conv3d_halo_tile(case):
    """the halo-tile kernel (conv_ht.h, v_mfma_f32_16x16x32_bf16) forced onto every case: spatial / temporal mode, PRE, accumulate,
    channel slices, partial chunks (the 32x32x16 form of round 3 measured 8-17 % slower and was deleted in round 6)"""
    lib = _lib()
    assert lib.vinet_set_option(b"ht", 2) == 0
    try:
        ex = dict(case[7])
        ex.setdefault("tline", 5)
        d0 = _run_conv_case(case[:7] + (ex,), E.BF16, forced=True)
        buf = C.create_string_buffer(128)
        assert lib.vinet_conv3d_kernel_name(C.byref(d0), buf, 128) == 0 and buf.value.startswith(b"conv_ht_kernel<"), buf.value
    finally:
        lib.vinet_set_option(b"ht", 1)


@pytest.mark.parametrize("case", HT_CASES, ids=[c[0] for c in HT_CASES])
def test_conv3d_halo_tile_split_bf16(case):
    """the halo-tile kernel in the VINET_F32S form (conv_ht.h, SPLIT: fp32 halo image, K steps of 32 channels, hi / lo weight
    planes, three MFMAs per product) on the same forced cases, held to the split form's 1e-4"""
    lib = _lib()
    assert lib.vinet_set_option(b"ht", 2) == 0
    try:
        ex = dict(case[7])
        ex.setdefault("tline", 5)
        d0 = _run_conv_case(case[:7] + (ex,), E.F32, forced=True, cdt=L.F32S, tol=1e-4)
        buf = C.create_string_buffer(128)
        d0.dtype = L.F32S
        assert lib.vinet_conv3d_kernel_name(C.byref(d0), buf, 128) == 0 and buf.value.startswith(b"conv_ht3_kernel<"), buf.value
        assert lib.vinet_set_option(b"ht3", 0) == 0
        assert lib.vinet_conv3d_kernel_name(C.byref(d0), buf, 128) == 0 and buf.value.startswith(b"conv_dma3_kernel<"), buf.value
    finally:
        lib.vinet_set_option(b"ht", 1)
        lib.vinet_set_option(b"ht3", 1)


# the pointwise streaming kernel (conv_pw.h), forced on small grids: 32- / 64- / 96-column weight tiles with padded and partial
# column tiles (N = 176, 288, 40), 1 / 2 / 6 / 8 / 9 K steps with a channel tail inside the last one (Cin = 176, 40), row tails,
# pending BatchNorm + ReLU (NaN-page padding), statistics (one row per workgroup), affine / ReLU epilogue,
# channel- and T-sliced views on both sides, and two grids large enough that every wave walks several tiles (ring across tiles)
PW_CASES = [
    ("pw_256_288_pre_stats", (2, 4, 14, 24), 256, 288, (1, 1, 1), (1, 1, 1), (0, 0, 0), dict(pre=True, stats=True)),
    ("pw_288_256", (2, 3, 7, 9), 288, 256, (1, 1, 1), (1, 1, 1), (0, 0, 0), {}),
    ("pw_64_64_relu", (2, 3, 7, 9), 64, 64, (1, 1, 1), (1, 1, 1), (0, 0, 0), dict(stats=True, act=1, epi=True)),
    ("pw_192_176_slices", (2, 2, 6, 8), 192, 176, (1, 1, 1), (1, 1, 1), (0, 0, 0), dict(pre=True, stats=True, in_ld=368, in_coff=112,
                                                                                   out_ld=480, out_coff=32)),
    ("pw_cin176_relu", (1, 3, 9, 11), 176, 192, (1, 1, 1), (1, 1, 1), (0, 0, 0), dict(act=1)),
    ("pw_cin40_n40_pre", (1, 2, 9, 7), 40, 40, (1, 1, 1), (1, 1, 1), (0, 0, 0), dict(pre=True, stats=True)),
    ("pw_32_32", (2, 1, 8, 12), 32, 32, (1, 1, 1), (1, 1, 1), (0, 0, 0), dict(epi_shift=True)),
    ("pw_tslices", (2, 3, 5, 7), 64, 96, (1, 1, 1), (1, 1, 1), (0, 0, 0), dict(in_ttotal=5, in_toff=1, out_ttotal=4, out_toff=1, stats=True)),
    ("pw_many_tiles_k8", (2, 8, 56, 96), 256, 288, (1, 1, 1), (1, 1, 1), (0, 0, 0), dict(pre=True, stats=True)),
    ("pw_many_tiles_k2", (3, 8, 56, 96), 64, 64, (1, 1, 1), (1, 1, 1), (0, 0, 0), dict(epi=True)),
    ("pw_many_tiles_k1", (3, 8, 56, 96), 32, 64, (1, 1, 1), (1, 1, 1), (0, 0, 0), dict(stats=True, act=1)),
]


# Data gradients that also leave the BatchNorm-backward partial sums of the layer(s) behind their output (VinetConvDesc::bnb_*,
# conv_bnb.hip): halo tiles in both modes and widths, conv_dma in every tile family (pointwise streaming off), store and accumulate
# forms, ragged M / N, sliced gradient and z views, with and without the ReLU gate
BNB_CASES = [
    ("bnb_ht_64_64", (2, 2, 8, 32), 64, 64, (1, 3, 3), (1, 1, 1), (0, 1, 1), dict(bnb=dict())),
    ("bnb_ht_128_96_tw16", (1, 3, 20, 16), 128, 96, (1, 3, 3), (1, 1, 1), (0, 1, 1), dict(bnb=dict(ld=160, coff=32), out_ld=256, out_coff=64)),
    ("bnb_ht_64_32", (1, 2, 9, 32), 64, 32, (1, 3, 3), (1, 1, 1), (0, 1, 1), dict(bnb=dict(relu=False))),
    ("bnb_ht_acc_192", (2, 2, 8, 32), 64, 192, (1, 3, 3), (1, 1, 1), (0, 1, 1), dict(bnb=dict(), accumulate=True)),
    ("bnb_htt_192", (2, 6, 8, 16), 192, 192, (3, 1, 1), (1, 1, 1), (1, 0, 0), dict(tline=True, bnb=dict())),
    ("bnb_htt_acc_96", (2, 7, 6, 8), 64, 96, (3, 1, 1), (1, 1, 1), (1, 0, 0), dict(tline=True, accumulate=True, bnb=dict(ld=128, coff=16))),
    ("bnb_htt_64", (1, 5, 14, 24), 128, 64, (3, 1, 1), (1, 1, 1), (1, 0, 0), dict(tline=True, bnb=dict())),
    ("bnb_pw_176_192", (2, 3, 10, 13), 176, 192, (1, 1, 1), (1, 1, 1), (0, 0, 0), dict(tline=6, bnb=dict(), accumulate=True)),
    ("bnb_pw_288_256", (1, 4, 11, 17), 288, 256, (1, 1, 1), (1, 1, 1), (0, 0, 0), dict(tline=6, bnb=dict(ld=320, coff=64))),
    ("bnb_pw_64_48", (3, 2, 7, 9), 64, 48, (1, 1, 1), (1, 1, 1), (0, 0, 0), dict(tline=6, bnb=dict(relu=False), accumulate=True)),
    ("bnb_pw_96_16", (2, 2, 5, 8), 96, 16, (1, 1, 1), (1, 1, 1), (0, 0, 0), dict(tline=6, bnb=dict())),
    ("bnb_pw_64_128", (2, 5, 9, 16), 64, 128, (1, 1, 1), (1, 1, 1), (0, 0, 0), dict(tline=6, bnb=dict(), accumulate=True)),
    ("bnb_dma_3t_32_64", (2, 6, 5, 12), 32, 64, (3, 1, 1), (1, 1, 1), (1, 0, 0), dict(tline=True, bnb=dict())),
]


@pytest.mark.parametrize("exact", [False, True], ids=["random", "exact"])
@pytest.mark.parametrize("case", BNB_CASES, ids=[c[0] for c in BNB_CASES])
def test_conv3d_bn_bwd_stats_from_the_epilogue(case, exact):
    """vinet_conv3d with bnb_partials: the output equals the ABI model's (and the plain launch's: same arithmetic), and the partial
    rows sum to the reduce pass's two sums over the gradient the launch stored -- exactly, on exact-arithmetic inputs"""
    lib = _lib()
    ht = case[0].startswith("bnb_ht")
    assert lib.vinet_set_option(b"ht", 2 if ht else 0) == 0 and lib.vinet_set_option(b"pw", 0) == 0 and lib.vinet_set_option(b"pp", 0) == 0
    ex = dict(case[7])
    if ht and not case[0].startswith("bnb_htt"):
        ex.setdefault("tline", 5)
    case = case[:7] + (ex,)
    try:
        if exact:
            with exact_mode():
                _run_conv_case(case, E.BF16, forced=True)
        else:
            _run_conv_case(case, E.BF16, forced=True)
    finally:
        lib.vinet_set_option(b"ht", 1)
        lib.vinet_set_option(b"pw", 1)
        lib.vinet_set_option(b"pp", 1)


@pytest.mark.parametrize("case", PW_CASES, ids=[c[0] for c in PW_CASES])
def test_conv3d_pointwise_stream(case):
    lib = _lib()
    assert lib.vinet_set_option(b"pw", 2) == 0
    try:
        ex = dict(case[7])
        ex["tline"] = 6
        d0 = _run_conv_case(case[:7] + (ex,), E.BF16, forced=True)
        buf = C.create_string_buffer(128)
        assert lib.vinet_conv3d_kernel_name(C.byref(d0), buf, 128) == 0 and buf.value.startswith(b"conv_pw_kernel<"), buf.value
    finally:
        lib.vinet_set_option(b"pw", 1)


# conv epilogue option epi_rows = 1 (whole-row stores through a wave-private LDS image, conv_igemm.h): off by default (measured
# neutral to slower), kept working: row tails, channel tails (N = 48, 208, 80), channel-sliced outputs, stride-phase placement
# (non-linear row offsets), spatial and temporal halo tiles, statistics
EPI_ROWS_CASES = [c for c in CONV_CASES if c[0] in ("pw_pre_stats", "tm_3x1", "concat_slice_out", "phase_store", "big_m", "xslice_pw")]
EPI_ROWS_HT = [c for c in HT_CASES if c[0] in ("ht_64_192", "ht_cin160_n80", "ht_slices", "ht_phase_store", "htt_192", "htt_pre_cin160")]


@pytest.mark.parametrize("case", EPI_ROWS_CASES + EPI_ROWS_HT, ids=[c[0] for c in EPI_ROWS_CASES + EPI_ROWS_HT])
def test_conv3d_whole_row_epilogue(case):
    lib = _lib()
    ht = case in EPI_ROWS_HT
    if lib.vinet_set_option(b"epi_rows", 1) == -2:
        pytest.skip("the whole-row epilogue is compiled into -DVINET_EXPERIMENTS side builds only (profiles/r3_epi_rows_ab.txt)")
    assert lib.vinet_set_option(b"epi_rows", 1) == 0 and lib.vinet_set_option(b"ht", 2 if ht else 0) == 0
    try:
        ex = dict(case[7])
        if ht:
            ex.setdefault("tline", 5)
        _run_conv_case(case[:7] + (ex,), E.BF16, forced=True)
    finally:
        lib.vinet_set_option(b"epi_rows", 0)
        lib.vinet_set_option(b"ht", 1)


# split-K (grids too small for the chip): long-K decoder shape, placement through a concat slice, padded fp32
# head, a T-sliced input, a pending affine, a chunk count that does not divide over the splits
SPLITK_CASES = [
    ("sk_dec_bigK", (1, 5, 4, 6), 480, 192, (5, 3, 3), (5, 1, 1), (0, 1, 1), dict(act=1, epi=True)),
    ("sk_concat", (1, 2, 6, 6), 256, 32, (1, 3, 3), (1, 1, 1), (0, 1, 1), dict(out_ld=96, out_coff=32, act=1)),
    ("sk_head", (2, 1, 8, 12), 224, 1, (1, 3, 3), (1, 1, 1), (0, 1, 1), dict(epi_shift=True, act=2, out_f32=True, head=True)),
    ("sk_tslice", (2, 3, 5, 5), 96, 48, (3, 3, 3), (3, 1, 1), (0, 1, 1), dict(in_ttotal=5, in_toff=1)),
    ("sk_pre", (1, 2, 9, 11), 160, 96, (1, 3, 3), (1, 1, 1), (0, 1, 1), dict(pre=True)),
    ("sk_m300", (1, 3, 10, 10), 832, 384, (3, 1, 1), (1, 1, 1), (1, 0, 0), dict(epi=True, act=1)),
    # long K on >= 1024 voxels: the large tiles that split-K makes affordable (128 x 192 for N = 192, 128 x 128 otherwise)
    ("sk_tile192", (1, 5, 28, 48), 480, 192, (5, 3, 3), (5, 1, 1), (0, 1, 1), dict(act=1, epi=True)),
    ("sk_tile128", (1, 3, 28, 48), 160, 480, (3, 3, 3), (3, 1, 1), (0, 1, 1), dict(act=1)),
]


@pytest.mark.parametrize("min_per", [1, 2, 7], ids=["default", "per2", "per7"])
@pytest.mark.parametrize("case", SPLITK_CASES, ids=[c[0] for c in SPLITK_CASES])
def test_conv3d_splitk(case, min_per):
    lib = _lib()
    assert lib.vinet_set_option(b"splitk", min_per) == 0
    try:
        ex = dict(case[7], splitk=True)
        _run_conv_case(case[:7] + (ex,), E.BF16)
        # deterministic: the slabs are added in a fixed order
        a = _run_conv_case(case[:7] + (ex,), E.BF16, want_y=True)
        b = _run_conv_case(case[:7] + (ex,), E.BF16, want_y=True)
        assert torch.equal(a, b)
    finally:
        lib.vinet_set_option(b"splitk", 1)


# frame-streaming temporal conv (conv_ts.hip): forward with statistics / folded BN, stride 2 and 1, the two stride
# phases of the stem partner's data gradient (taps in descending order, output frames interleaved, accumulate),
# sliced views, windows running past the last frame
CONV_TS_CASES = [
    ("ts_fwd_k7s2", (2, 9, 8, 16), 64, 64, (7, 1, 1), (2, 1, 1), (3, 0, 0), dict(pre=True, stats=True)),
    ("ts_fwd_k7s2_eval", (1, 12, 8, 8), 64, 64, (7, 1, 1), (2, 1, 1), (3, 0, 0), dict(epi=True, act=1)),
    ("ts_fwd_k3s1", (2, 5, 16, 8), 64, 64, (3, 1, 1), (1, 1, 1), (1, 0, 0), dict(pre=True, stats=True, in_ld=160, in_coff=32, out_ld=96, out_coff=16)),
    ("ts_fwd_k2s2", (1, 8, 8, 8), 64, 64, (2, 1, 1), (2, 1, 1), (0, 0, 0), dict(act=1)),
    ("ts_fwd_long", (1, 32, 8, 24), 64, 64, (7, 1, 1), (2, 1, 1), (3, 0, 0), dict(pre=True, stats=True)),
    ("ts_acc", (2, 6, 8, 8), 64, 64, (3, 1, 1), (1, 1, 1), (1, 0, 0), dict(accumulate=True)),
    # without statistics the frames of a patch split into segments (small batches): 4 x 4 output frames; 6 + 5 with accumulation;
    # a stride phase's placement (every second frame of y)
    ("ts_segs_k7s2", (1, 32, 8, 8), 64, 64, (7, 1, 1), (2, 1, 1), (3, 0, 0), dict(epi=True, act=1)),
    ("ts_segs_acc", (1, 11, 8, 8), 64, 64, (3, 1, 1), (1, 1, 1), (1, 0, 0), dict(accumulate=True)),
    ("ts_segs_phase", (1, 10, 8, 8), 64, 64, (3, 1, 1), (1, 1, 1), (1, 0, 0), dict(om=(2, 1))),
]


@pytest.mark.parametrize("case", CONV_TS_CASES, ids=[c[0] for c in CONV_TS_CASES])
def test_conv3d_tstream(case):
    lib = _lib()
    assert lib.vinet_set_option(b"conv_ts", 2) == 0
    try:
        ex = dict(case[7], tline=True)
        d0 = _run_conv_case(case[:7] + (ex,), E.BF16, forced=True)
        buf = C.create_string_buffer(128)
        assert lib.vinet_conv3d_kernel_name(C.byref(d0), buf, 128) == 0 and buf.value.startswith(b"conv_ts_kernel<")
        assert lib.vinet_conv3d_tile_m(C.byref(d0)) == 64
    finally:
        lib.vinet_set_option(b"conv_ts", 1)


@pytest.mark.parametrize("case", CONV_TS_CASES, ids=[c[0] for c in CONV_TS_CASES])
def test_conv3d_tstream_split_bf16(case):
    """the frame-streaming kernel in the VINET_F32S form (conv_ts3_kernel: fp32 tensors, frames split into hi / lo planes on the way
    into the LDS ring, three MFMAs per product) on the same forced cases, held to the split form's 1e-4"""
    lib = _lib()
    assert lib.vinet_set_option(b"conv_ts", 2) == 0
    try:
        ex = dict(case[7], tline=True)
        d0 = _run_conv_case(case[:7] + (ex,), E.F32, forced=True, cdt=L.F32S, tol=1e-4)
        d0.dtype = L.F32S
        buf = C.create_string_buffer(128)
        assert lib.vinet_conv3d_kernel_name(C.byref(d0), buf, 128) == 0 and buf.value.startswith(b"conv_ts3_kernel<"), buf.value
    finally:
        lib.vinet_set_option(b"conv_ts", 1)


@pytest.mark.parametrize("ksp", [(7, 2, 3), (3, 2, 1), (2, 2, 0), (5, 3, 2)], ids=["k7s2p3", "k3s2p1", "k2s2p0", "k5s3p2"])
@pytest.mark.parametrize("acc", [0, 1])
def test_conv3d_tstream_dgrad_fused(ksp, acc):
    """the whole data gradient of a strided temporal 64 -> 64 conv in one launch (tline == 3)"""
    lib = _lib()
    dt = E.BF16
    k, s, p = ksp
    B, Ti, H, W, Cc = 2, 11, 8, 8, 64
    To = (Ti + 2 * p - k) // s + 1
    xp, xmk = view_pair(B, To, H, W, Cc, dt, "fdy", 1, ld=96, c_off=16)
    yp, ymk = view_pair(B, Ti, H, W, Cc, dt, "fdx", 2, fill=0.25)
    wp = Pair((_rand("fdw", (k * Cc * Cc,), 3, 0.05)).to(E.TORCH_DT[dt]))

    def mk(side):
        d = L.CConvDesc()
        d.dtype, d.out_dtype, d.mode = dt, dt, 0
        d.x, d.y = xmk(side).ct(), ymk(side).ct()
        d.oT, d.oH, d.oW = Ti, H, W
        d.sT, d.sH, d.sW = s, 1, 1
        d.omT = d.omH = d.omW = 1
        d.ntaps, d.taps, d.w, d.Kp = k, None, wp.ptr(side), 64
        d.pre = L.CAffine(None, None, 0)
        d.accumulate = acc
        d.tline, d.tpad = 3, p
        return [C.byref(d), _stream() if side == "gpu" else 0]

    assert lib.vinet_set_option(b"conv_ts", 2) == 0
    try:
        assert lib.vinet_conv3d_fuses_dgrad_phases(mk("gpu")[0]) == 1
        run_both("vinet_conv3d", mk)
    finally:
        lib.vinet_set_option(b"conv_ts", 1)
    _cmp(yp.get("gpu"), yp.get("cpu"), TOL[dt], "fused temporal dgrad")


@pytest.mark.parametrize("relu", [1, 0], ids=["relu", "linear"])
@pytest.mark.parametrize("ksp", [(7, 2, 3), (3, 2, 1)], ids=["k7s2p3", "k3s2p1"])
def test_conv3d_tstream_dgrad_fused_bn_bwd_stats(ksp, relu):
    """tline == 3 with VinetConvDesc::bnb_*: the launch also leaves the partial sums of vinet_bn_bwd_reduce(dx, z) -- against that
    very pass run on the gradient the launch wrote (the ABI model's, summed over rows), on a sliced z view; dx itself unchanged"""
    lib = _lib()
    dt = E.BF16
    k, s, p = ksp
    B, Ti, H, W, Cc = 2, 11, 8, 16, 64
    To = (Ti + 2 * p - k) // s + 1
    xp, xmk = view_pair(B, To, H, W, Cc, dt, "bdy", 1, ld=96, c_off=16)
    yp, ymk = view_pair(B, Ti, H, W, Cc, dt, "bdx", 2, fill=0.25)
    yp0, ymk0 = view_pair(B, Ti, H, W, Cc, dt, "bdx0", 2, fill=0.5)
    zp, zmk = view_pair(B, Ti, H, W, Cc, dt, "bz", 5, ld=160, c_off=32)
    wp = Pair((_rand("bdw", (k * Cc * Cc,), 3, 0.05)).to(E.TORCH_DT[dt]))
    sc, sh = fvec("bsc", Cc, 6, 0.5, 1.5), fvec("bsh", Cc, 7)
    mean, invstd = fvec("bmu", Cc, 8), fvec("bis", Cc, 9, 0.5, 2.0)

    def mk(side, bnb, ymk_):
        d = L.CConvDesc()
        d.dtype, d.out_dtype, d.mode = dt, dt, 0
        d.x, d.y = xmk(side).ct(), ymk_(side).ct()
        d.oT, d.oH, d.oW = Ti, H, W
        d.sT, d.sH, d.sW = s, 1, 1
        d.omT = d.omH = d.omW = 1
        d.ntaps, d.taps, d.w, d.Kp = k, None, wp.ptr(side), 64
        d.pre = L.CAffine(None, None, 0)
        d.tline, d.tpad = 3, p
        if bnb:
            z = zmk(side)
            d.bnb_z, d.bnb_ld, d.bnb_sB = z.ptr(), z.ld, z.sB
            d.bnb_fwd = L.CAffine(sc.ptr(side), sh.ptr(side), relu)
            d.bnb_mean, d.bnb_invstd = mean.ptr(side), invstd.ptr(side)
        return d

    assert lib.vinet_set_option(b"conv_ts", 2) == 0
    try:
        d = mk("gpu", True, ymk)
        rows = lib.vinet_conv3d_bn_bwd_stats_rows(C.byref(d))
        assert rows == B * (H * W) // 64
        part = torch.full((rows * 2 * Cc,), float("nan"), device="cuda")
        d.bnb_partials = part.data_ptr()
        assert lib.vinet_conv3d(C.byref(d), _stream()) == 0, lib.vinet_last_error()
        d0 = mk("gpu", False, ymk0)
        assert lib.vinet_conv3d(C.byref(d0), _stream()) == 0, lib.vinet_last_error()
        d.accumulate = 1
        assert lib.vinet_conv3d_bn_bwd_stats_rows(C.byref(d)) == 0      # (an accumulating launch does not see the whole gradient)
        torch.cuda.synchronize()
    finally:
        lib.vinet_set_option(b"conv_ts", 1)
    assert torch.equal(yp.get("gpu"), yp0.get("gpu")), "dx differs with the statistics folded in"
    # the separate pass of the ABI model on the gradient the GPU wrote
    emu = AbiEmulator()
    g_cpu = Pair(yp.get("gpu").clone())
    gv = E.View(g_cpu.cpu, 0, B, Ti, H, W, Cc, Cc, Ti * H * W * Cc, dt)
    zc = zmk("cpu")
    r2 = emu.vinet_stats_rows(gv.ct())
    ws = torch.zeros(r2 * 2 * Cc)
    assert emu.vinet_bn_bwd_reduce(gv.ct(), zc.ct(), dt, L.CAffine(sc.ptr("cpu"), sh.ptr("cpu"), relu), mean.ptr("cpu"), invstd.ptr("cpu"),
                                   ws.data_ptr(), 0) == 0
    ref = ws.view(r2, 2, Cc).double().sum(0)
    got = part.cpu().view(rows, 2, Cc).double().sum(0)
    assert torch.isfinite(got).all()
    _cmp(got, ref, 1e-3, "BatchNorm-backward partial sums out of the fused temporal data gradient")


@pytest.mark.parametrize("r", [0, 1])
def test_conv3d_tstream_dgrad_phase(r):
    """one stride phase of the 7x1x1 / 2 data gradient: taps (e - j, slice d0 + 2j) in descending offset order, output
    frames r, r+2, ... of a 2x longer tensor, accumulated"""
    lib = _lib()
    dt = E.BF16
    B, To, H, W, Cc = 2, 8, 8, 8, 64
    Ti = 2 * To
    d0_, e = (r + 3) % 2, (r + 3 - (r + 3) % 2) // 2
    rows = [(e - j, 0, 0, d0_ + 2 * j) for j in range(4) if d0_ + 2 * j < 7]
    xp, xmk = view_pair(B, To, H, W, Cc, dt, "tdy", 1)
    yp, ymk = view_pair(B, Ti, H, W, Cc, dt, "tdx", 2)
    wp = Pair((_rand("tdw", (7 * Cc * Cc,), 3, 0.05)).to(E.TORCH_DT[dt]))
    taps = Pair(torch.tensor(rows, dtype=torch.int32))

    def mk(side):
        d = L.CConvDesc()
        d.dtype, d.out_dtype, d.mode = dt, dt, 0
        d.x, d.y = xmk(side).ct(), ymk(side).ct()
        d.oT, d.oH, d.oW = To, H, W
        d.sT = d.sH = d.sW = 1
        d.omT, d.omH, d.omW = 2, 1, 1
        d.ooT, d.ooH, d.ooW = r, 0, 0
        d.ntaps, d.taps, d.w, d.Kp = len(rows), taps.ptr(side), wp.ptr(side), 64
        d.pre = L.CAffine(None, None, 0)
        d.accumulate = 1
        d.tline, d.tpad = 1, -min(t[0] for t in rows)
        return [C.byref(d), _stream() if side == "gpu" else 0]

    assert lib.vinet_set_option(b"conv_ts", 2) == 0
    try:
        run_both("vinet_conv3d", mk)
        buf = C.create_string_buffer(128)
        assert lib.vinet_conv3d_kernel_name(mk("gpu")[0], buf, 128) == 0 and buf.value.startswith(b"conv_ts_kernel<")
    finally:
        lib.vinet_set_option(b"conv_ts", 1)
    _cmp(yp.get("gpu"), yp.get("cpu"), TOL[dt], "temporal dgrad phase")


def _run_conv_case(case, dt, forced=False, want_y=False, cdt=None, tol=None):
    name, (B, T, H, W), Cin, N, k, s, p, ex = case
    oT, oH, oW = [(d + 2 * pp - kk) // ss + 1 for d, kk, ss, pp in zip((T, H, W), k, s, p)]
    xp, xmk = view_pair(B, T, H, W, Cin, dt, "x" + name, 1, t_total=ex.get("in_ttotal"), t_off=ex.get("in_toff", 0),
                        ld=ex.get("in_ld"), c_off=ex.get("in_coff", 0))
    head = ex.get("head", False)
    Ny = (E.EG[dt] if head else N)
    odt = E.F32 if ex.get("out_f32") else dt
    omT, ooT = ex.get("om", (1, 0))
    yp, ymk = view_pair(B, oT * omT, oH, oW, Ny, odt, "y" + name, 2, ld=ex.get("out_ld"), c_off=ex.get("out_coff", 0),
                        t_total=ex.get("out_ttotal"), t_off=ex.get("out_toff", 0))
    Kp = E.rup(Cin, 32)
    ntaps = k[0] * k[1] * k[2]
    wmaster = _rand("w" + name, (N, Cin, ntaps), 3, 1.0 / math.sqrt(Cin * ntaps))
    wpk = torch.zeros(ntaps, N, Kp)
    wpk[:, :, :Cin] = wmaster.permute(2, 0, 1)
    wp = Pair(wpk.to(E.TORCH_DT[dt]))
    if cdt == L.F32S:      # the split form reads hi / lo bf16 planes (vinet_pack_weights with the arithmetic dtype): pack the GPU side
        wsrc = wmaster.contiguous().cuda()
        assert _lib().vinet_pack_weights(wsrc.data_ptr(), N, Cin, ntaps, 0, 0, L.F32S, wp.gpu.data_ptr(), _stream()) == 0
        torch.cuda.synchronize()
    taps = Pair(torch.tensor(_fwd_taps(k, p), dtype=torch.int32))
    pre_s, pre_h = fvec("ps" + name, Cin, 4, 0.5, 1.5), fvec("ph" + name, Cin, 5)
    os_, oh_ = fvec("os" + name, N, 6, 0.5, 1.5), fvec("oh" + name, N, 7)
    M = B * oT * oH * oW
    rows = (M + 63) // 64 + B * oT * 8
    stats = Pair(torch.zeros(rows * 2 * N))
    bnb = ex.get("bnb")
    if bnb is not None:      # the launch is a data gradient behind BatchNorm + ReLU layers: it also writes their backward partial sums
        zp, zmk = view_pair(B, oT, oH, oW, Ny, odt, "z" + name, 11, ld=bnb.get("ld"), c_off=bnb.get("coff", 0))
        b_sc, b_sh = fvec("bsc" + name, N, 12, 0.5, 1.5), fvec("bsh" + name, N, 13)
        b_mu, b_is = fvec("bmu" + name, N, 14), fvec("bis" + name, N, 15, 0.5, 2.0)
        bpart = Pair(torch.full((rows * 2 * N,), float("nan")))

    def mk(side):
        d = L.CConvDesc()
        d.dtype, d.out_dtype, d.mode = (dt if cdt is None else cdt), odt, 0
        d.x, d.y = xmk(side).ct(), ymk(side).ct()
        d.oT, d.oH, d.oW = oT, oH, oW
        d.sT, d.sH, d.sW = s
        d.omT = d.omH = d.omW = 1
        d.ntaps, d.taps, d.w, d.Kp = ntaps, taps.ptr(side), wp.ptr(side), Kp
        d.pre = L.CAffine(pre_s.ptr(side), pre_h.ptr(side), 1) if ex.get("pre") else L.CAffine(None, None, 0)
        d.out_scale = os_.ptr(side) if ex.get("epi") else None
        d.out_shift = oh_.ptr(side) if (ex.get("epi") or ex.get("epi_shift")) else None
        d.act = ex.get("act", 0)
        d.accumulate = 1 if ex.get("accumulate") else 0
        d.stats = stats.ptr(side) if ex.get("stats") else None
        d.n_valid = N if head else 0
        if ex.get("tline"):
            d.tline, d.tpad = (1 if ex["tline"] is True else ex["tline"]), p[0]
        if ex.get("om"):            # output placement of a stride phase: positions (to*omT + ooT, ...) of a larger y
            d.omT, d.ooT = ex["om"]
        if bnb is not None:
            z = zmk(side)
            d.bnb_z, d.bnb_ld, d.bnb_sB = z.ptr(), z.ld, z.sB
            d.bnb_fwd = L.CAffine(b_sc.ptr(side), b_sh.ptr(side), 1 if bnb.get("relu", True) else 0)
            d.bnb_mean, d.bnb_invstd = b_mu.ptr(side), b_is.ptr(side)
            d.bnb_partials = bpart.ptr(side)
        if ex.get("splitk") and side == "gpu":
            nb = _lib().vinet_conv3d_splitk_bytes(C.byref(d))
            assert nb >= 2 * M * Ny * 4, "split-K plan expected for " + name
            ws = torch.full((nb // 4,), float("nan"), device="cuda")     # scratch contents must not matter
            keep_ws.append(ws)
            d.splitk_ws, d.splitk_ws_bytes = ws.data_ptr(), nb
        return [C.byref(d), _stream() if side == "gpu" else 0]

    keep_ws = []
    if bnb is not None:
        d0 = mk("gpu")[0]._obj
        nbuf = C.create_string_buffer(128)
        _lib().vinet_conv3d_kernel_name(C.byref(d0), nbuf, 128)
        br = _lib().vinet_conv3d_bn_bwd_stats_rows(C.byref(d0))
        assert 0 < br <= rows, "no BatchNorm-backward statistics for %s (%s)" % (name, nbuf.value.decode())
        assert br == _lib().vinet_conv3d_stats_rows(C.byref(d0))
    run_both("vinet_conv3d", mk)
    _cmp(yp.get("gpu"), yp.get("cpu"), TOL[dt] if tol is None else tol, "conv " + name)
    if bnb is not None:
        # against the reduce pass's definition, in double, on the gradient the GPU wrote: (sum g gate, sum g gate (z - mean) invstd)
        yv, zv = ymk("cpu"), zmk("cpu")
        g = E.View(yp.get("gpu"), yv.off, yv.B, yv.T, yv.H, yv.W, yv.C, yv.ld, yv.sB, odt).torch5().double().reshape(-1, Ny)
        zz = zv.torch5().double().reshape(-1, Ny)
        if bnb.get("relu", True):
            g = g * ((zz * b_sc.cpu.double() + b_sh.cpu.double()) > 0)
        ref = torch.stack([g.sum(0), (g * (zz - b_mu.cpu.double()) * b_is.cpu.double()).sum(0)])
        got = bpart.get("gpu")[:br * 2 * N].view(br, 2, N).double().sum(0)
        assert torch.isfinite(got).all(), "unwritten rows of the partial-sum table"
        if _EXACT[0]:
            assert torch.equal(got, ref), "bn-backward sums of %s not exact: max diff %g" % (name, float((got - ref).abs().max()))
        else:
            scale = g.abs().sum(0).clamp_min(1.0) * 4
            assert float(((got - ref).abs() / scale).max()) < 2e-3, "bn-backward sums of %s: %g" % (name, float(((got - ref).abs() / scale).max()))
    if ex.get("stats"):
        d0 = mk("gpu")[0]._obj
        bm = _lib().vinet_conv3d_tile_m(C.byref(d0))
        assert forced or bm == AbiEmulator().vinet_conv3d_tile_m(d0)
        r = _lib().vinet_conv3d_stats_rows(C.byref(d0))
        nbuf = C.create_string_buffer(128)
        _lib().vinet_conv3d_kernel_name(C.byref(d0), nbuf, 128)
        per_item = nbuf.value.startswith(b"conv_ts") or nbuf.value.startswith(b"conv_hs")    # one row per 64-position item, all its frames / rows
        assert (r >= (M + bm - 1) // bm or ex.get("tline") == 6 or per_item) and (forced or r == AbiEmulator().vinet_conv3d_stats_rows(d0))   # (pointwise: one row per workgroup)
        assert r <= rows
        rc = AbiEmulator().vinet_conv3d_stats_rows(d0)
        sg = stats.get("gpu")[:r * 2 * N].view(r, 2, N).double().sum(0)
        sc = stats.get("cpu")[:rc * 2 * N].view(rc, 2, N).double().sum(0)
        _cmp(sg, sc, 1e-4 if dt == E.F32 else 2e-2, "conv stats " + name)
    if want_y:
        return yp.get("gpu").clone()
    return mk("gpu")[0]._obj


@pytest.mark.parametrize("dt", DTS)
def test_conv3d_stem_mode(dt):
    B, T, H, W, N = 2, 3, 18, 22, 64
    oH, oW = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
    x = torch.zeros(B, T, H, W, 4)
    x[..., :3] = _rand("stemx", (B, T, H, W, 3), 1)
    xp = Pair(x.reshape(-1).to(E.TORCH_DT[dt]))
    yp, ymk = view_pair(B, T, oH, oW, N, dt, "stemy", 2)
    wm = _rand("stemw", (N, 3, 49), 3, 1.0 / math.sqrt(147))
    wpk = torch.zeros(7, N, 32)
    for kh in range(7):
        for kw in range(7):
            wpk[kh, :, kw * 4:kw * 4 + 3] = wm[:, :, kh * 7 + kw]
    wp = Pair(wpk.to(E.TORCH_DT[dt]))
    taps = Pair(torch.tensor([(0, kh - 3, -3, kh) for kh in range(7)], dtype=torch.int32))

    def mk(side):
        d = L.CConvDesc()
        d.dtype, d.out_dtype, d.mode = dt, dt, 1
        buf = xp.cpu if side == "cpu" else xp.gpu
        d.x = E.View(buf, 0, B, T, H, W, 4, 4, T * H * W * 4, dt).ct()
        d.y = ymk(side).ct()
        d.oT, d.oH, d.oW = T, oH, oW
        d.sT, d.sH, d.sW = 1, 2, 2
        d.omT = d.omH = d.omW = 1
        d.ntaps, d.taps, d.w, d.Kp = 7, taps.ptr(side), wp.ptr(side), 32
        d.pre = L.CAffine(None, None, 0)
        return [C.byref(d), _stream() if side == "gpu" else 0]

    run_both("vinet_conv3d", mk)
    _cmp(yp.get("gpu"), yp.get("cpu"), TOL[dt], "stem conv")
    # and against torch's conv3d directly (fp32 only): the emulator itself is checked here
    if dt == E.F32:
        ref = torch.nn.functional.conv3d(x[..., :3].permute(0, 4, 1, 2, 3), wm.view(N, 3, 1, 7, 7), stride=(1, 2, 2), padding=(0, 3, 3))
        _cmp(yp.get("gpu").view(B, T, oH, oW, N).permute(0, 4, 1, 2, 3), ref, 2e-5, "stem vs torch")


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("hw", [(18, 22), (17, 23), (32, 48), (20, 128), (16, 256), (60, 128), (200, 128)])     # (60, 128): 4 row segments, the last ragged; (200, 128): 14 asked for, 13 x 8 rows cover it
def test_stem_folded(dt, hw):
    """padded import + overlapped [.., (W+8)/2, C=32] ld=8 view: the stem as a generic 7-tap conv,
    forward and weight gradient, vs the emulator and (fp32) torch's conv3d"""
    B, T, N = 2, 3, 64
    H, W = hw
    Hp, Wp = H + 6, W + 8 + (W & 1)
    oH, oW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    src = Pair(_rand("sfx", (B, 3, T, H, W), 1))
    n = B * T * Hp * Wp * 4
    buf = Pair(torch.full((n + 64,), 7.0).to(E.TORCH_DT[dt]))
    yp, ymk = view_pair(B, T, oH, oW, N, dt, "sfy", 2)
    wm = _rand("sfw", (N, 3, 49), 3, 1.0 / math.sqrt(147))
    wpk = torch.zeros(7, N, 32)
    for kh in range(7):
        for kw in range(7):
            wpk[kh, :, kw * 4:kw * 4 + 3] = wm[:, :, kh * 7 + kw]
    wp = Pair(wpk.to(E.TORCH_DT[dt]))
    taps = Pair(torch.tensor([(0, kh, 0, kh) for kh in range(7)], dtype=torch.int32))
    sB = T * Hp * Wp * 4

    def b(side):
        return buf.cpu if side == "cpu" else buf.gpu

    def mk_imp(side):
        t = src.cpu if side == "cpu" else src.gpu
        sb, sc, st, sh, sw = t.stride()
        dst = E.View(b(side), 0, B, T, Hp, Wp, 4, 4, sB, dt)
        return [t.data_ptr(), sb, sc, st, sh, sw, 3, H, W, 3, 3, C.byref(dst.ct()), dt, _stream() if side == "gpu" else 0]

    run_both("vinet_import_ncdhw_pad", mk_imp)
    _cmp(buf.get("gpu")[:n], buf.get("cpu")[:n], 0.0, "padded import")

    def mk(side):
        d = L.CConvDesc()
        d.dtype, d.out_dtype, d.mode = dt, dt, 0
        d.x = E.View(b(side), 0, B, T, Hp, Wp // 2, 32, 8, sB, dt).ct()
        d.y = ymk(side).ct()
        d.oT, d.oH, d.oW = T, oH, oW
        d.sT, d.sH, d.sW = 1, 2, 1
        d.omT = d.omH = d.omW = 1
        d.ntaps, d.taps, d.w, d.Kp = 7, taps.ptr(side), wp.ptr(side), 32
        d.pre = L.CAffine(None, None, 0)
        return [C.byref(d), _stream() if side == "gpu" else 0]

    run_both("vinet_conv3d", mk)
    _cmp(yp.get("gpu"), yp.get("cpu"), TOL[dt], "folded stem conv")
    if dt == E.BF16 and oW % 64 == 0:
        # the row-streaming strip kernel of the real stem (conv_hs.hip), with statistics and a folded-BN epilogue
        lib = _lib()
        os_, oh_ = fvec("sfos", N, 6, 0.5, 1.5), fvec("sfoh", N, 7)
        M = B * T * oH * oW
        for with_stats in (True, False):
            stats = Pair(torch.full(((M // 64) * 2 * N,), float("nan")))     # (rows past vinet_conv3d_stats_rows are never read)

            def mk_hs(side):
                args = mk(side)
                d = args[0]._obj
                d.tline = 2
                if with_stats:
                    d.stats = stats.ptr(side)
                else:
                    d.out_scale, d.out_shift, d.act = os_.ptr(side), oh_.ptr(side), 1
                return args
            lib.vinet_set_option(b"conv_hs", 2)
            try:
                run_both("vinet_conv3d", mk_hs)
                nbuf = C.create_string_buffer(128)
                d0 = mk_hs("gpu")[0]._obj
                assert lib.vinet_conv3d_kernel_name(C.byref(d0), nbuf, 128) == 0 and nbuf.value == b"conv_hs_kernel"
                assert lib.vinet_conv3d_tile_m(C.byref(d0)) == 64
                bm_cpu = AbiEmulator().vinet_conv3d_tile_m(d0)
                rg = lib.vinet_conv3d_stats_rows(C.byref(d0))
            finally:
                lib.vinet_set_option(b"conv_hs", 1)
            _cmp(yp.get("gpu"), yp.get("cpu"), TOL[dt], "folded stem conv (row-streaming strips)")
            if with_stats:
                assert 0 < rg <= M // 64
                sg = stats.get("gpu").view(M // 64, 2, N)[:rg].double().sum(0)
                rc = (M + bm_cpu - 1) // bm_cpu
                sc = stats.get("cpu")[:rc * 2 * N].view(rc, 2, N).double().sum(0)
                _cmp(sg, sc, 2e-2, "folded stem conv stats")
        run_both("vinet_conv3d", mk)      # (restore y for the checks below)
    if dt == E.F32 and oW % 64 == 0:
        # the strip kernel in the split-bf16 form (conv_hs3_kernel: fp32 folded clip, hi / lo weight planes from vinet_pack_weights),
        # with statistics / with a folded-BN epilogue, against the fp32 model
        lib = _lib()
        os_, oh_ = fvec("sfos3", N, 6, 0.5, 1.5), fvec("sfoh3", N, 7)
        M = B * T * oH * oW
        w3 = torch.zeros(7 * N * 32, device="cuda")
        assert lib.vinet_pack_weights(wm.contiguous().cuda().data_ptr(), N, 3, 49, 0, 1, L.F32S, w3.data_ptr(), _stream()) == 0
        y_ref = yp.get("cpu").clone()
        for with_stats in (True, False):
            stats = Pair(torch.zeros((M // 64) * 2 * N))

            def mk_hs3(side):
                args = mk(side)
                d = args[0]._obj
                d.tline = 2
                if side == "gpu":
                    d.dtype, d.w = L.F32S, w3.data_ptr()
                if with_stats:
                    d.stats = stats.ptr(side)
                else:
                    d.out_scale, d.out_shift, d.act = os_.ptr(side), oh_.ptr(side), 1
                return args
            lib.vinet_set_option(b"conv_hs", 2)
            try:
                run_both("vinet_conv3d", mk_hs3)
                nbuf = C.create_string_buffer(128)
                d0 = mk_hs3("gpu")[0]._obj
                assert lib.vinet_conv3d_kernel_name(C.byref(d0), nbuf, 128) == 0 and nbuf.value == b"conv_hs3_kernel", nbuf.value
                rows3 = lib.vinet_conv3d_stats_rows(C.byref(d0))
            finally:
                lib.vinet_set_option(b"conv_hs", 1)
            _cmp(yp.get("gpu"), yp.get("cpu"), 1e-4, "folded stem conv (split-bf16 strips)")
            if with_stats:
                assert rows3 == B * T * (oW // 64)
                sg = stats.get("gpu").view(M // 64, 2, N).double().sum(0)
                sc = stats.get("cpu").view(M // 64, 2, N).double().sum(0)
                _cmp(sg, sc, 1e-3, "folded stem conv stats (split-bf16 strips)")
        run_both("vinet_conv3d", mk)      # (restore y for the checks below)
    if dt == E.F32:
        ref = torch.nn.functional.conv3d(src.cpu, wm.view(N, 3, 1, 7, 7), stride=(1, 2, 2), padding=(0, 3, 3))
        _cmp(yp.get("gpu").view(B, T, oH, oW, N).permute(0, 4, 1, 2, 3), ref, 2e-5, "folded stem vs torch")

    dp, dmk = view_pair(B, T, oH, oW, N, dt, "sfd", 4)
    dw = Pair(torch.zeros(7 * N * 32))

    def mkw(side):
        d = L.CWgradDesc()
        d.dtype, d.mode = dt, 0
        d.x = E.View(b(side), 0, B, T, Hp, Wp // 2, 32, 8, sB, dt).ct()
        d.dy = dmk(side).ct()
        d.sT, d.sH, d.sW = 1, 2, 1
        d.ntaps, d.taps, d.dw, d.Kp = 7, taps.ptr(side), dw.ptr(side), 32
        d.pre = L.CAffine(None, None, 0)
        return [C.byref(d), _stream() if side == "gpu" else 0]

    run_both("vinet_conv3d_wgrad", mkw)
    _cmp(dw.get("gpu"), dw.get("cpu"), 3e-5 if dt == E.F32 else 2e-2, "folded stem wgrad")
    if dt == E.BF16:
        # the 7-taps-per-group, 32-channel-tile kernel the real stem (22 M voxels) gets
        lib = _lib()
        lib.vinet_set_option(b"wgrad_tg", 7)
        try:
            dw.gpu.zero_()
            dw.cpu.zero_()
            run_both("vinet_conv3d_wgrad", mkw)
            nbuf = C.create_string_buffer(128)
            assert lib.vinet_conv3d_wgrad_kernel_name(mkw("gpu")[0], nbuf, 128) == 0 and nbuf.value == b"conv_wgrad_dma_kernel<64,32,7,plain>"
        finally:
            lib.vinet_set_option(b"wgrad_tg", 0)
        _cmp(dw.get("gpu"), dw.get("cpu"), 2e-2, "folded stem wgrad (32-channel tile)")
        if oW % 64 == 0:
            # the row-streaming strip kernel of the real stem (wgrad_hs.hip); the caller promises the tap geometry
            def mkw_hs(side):
                args = mkw(side)
                args[0]._obj.tline = 2
                return args
            lib.vinet_set_option(b"wgrad_hs", 2)
            try:
                dw.gpu.zero_()
                dw.cpu.zero_()
                run_both("vinet_conv3d_wgrad", mkw_hs)
                assert lib.vinet_conv3d_wgrad_kernel_name(mkw_hs("gpu")[0], nbuf, 128) == 0 and nbuf.value == b"conv_wgrad_hs_kernel"
            finally:
                lib.vinet_set_option(b"wgrad_hs", 1)
            _cmp(dw.get("gpu"), dw.get("cpu"), 2e-2, "folded stem wgrad (row-streaming strips)")
            # ... with the BatchNorm(+ReLU) backward of the stem's BN applied to the dz operand on the fly
            zp, zmk = view_pair(B, T, oH, oW, N, dt, "sfz", 8)
            f_sc, f_sh = fvec("sffs", N, 9, 0.5, 1.5), fvec("sffh", N, 10)
            mu, istd = fvec("sfmu", N, 11), fvec("sfis", N, 12, 0.5, 2.0)
            c1, c2 = fvec("sfc1", N, 13, -0.1, 0.1), fvec("sfc2", N, 14, -0.1, 0.1)

            def mkw_bnb(side):
                args = mkw_hs(side)
                d = args[0]._obj
                zv = zmk(side)
                d.bnb_z, d.bnb_ld, d.bnb_sB = zv.ptr(), zv.ld, zv.sB
                d.bnb_fwd = L.CAffine(f_sc.ptr(side), f_sh.ptr(side), 1)
                d.bnb_mean, d.bnb_invstd, d.bnb_c1, d.bnb_c2 = mu.ptr(side), istd.ptr(side), c1.ptr(side), c2.ptr(side)
                return args
            lib.vinet_set_option(b"wgrad_hs", 2)
            try:
                assert lib.vinet_conv3d_wgrad_fuses_bn_bwd(mkw_bnb("gpu")[0]) == 1
                dw.gpu.zero_()
                dw.cpu.zero_()
                run_both("vinet_conv3d_wgrad", mkw_bnb)
            finally:
                lib.vinet_set_option(b"wgrad_hs", 1)
            _cmp(dw.get("gpu"), dw.get("cpu"), 2e-2, "folded stem wgrad with fused BN backward")


@pytest.mark.parametrize("dt", DTS)
def test_conv3d_phase_output_mapping(dt):
    """dgrad-style launch: iteration space Q, output written at o*om+oo into a larger tensor"""
    B, Q, N, Cin = 1, (3, 4, 5), 32, 64
    xp, xmk = view_pair(B, 3, 4, 5, Cin, dt, "phx", 1)
    yp, ymk = view_pair(B, 6, 4, 10, N, dt, "phy", 2, fill=0.5)
    Kp = 64
    wp = Pair(_rand("phw", (2 * N * Kp,), 3, 0.1).to(E.TORCH_DT[dt]))
    taps = Pair(torch.tensor([(0, 0, 0, 0), (1, 0, -1, 1)], dtype=torch.int32))

    def mk(side):
        d = L.CConvDesc()
        d.dtype, d.out_dtype, d.mode = dt, dt, 0
        d.x, d.y = xmk(side).ct(), ymk(side).ct()
        d.oT, d.oH, d.oW = Q
        d.sT = d.sH = d.sW = 1
        d.omT, d.omH, d.omW = 2, 1, 2
        d.ooT, d.ooH, d.ooW = 1, 0, 1
        d.ntaps, d.taps, d.w, d.Kp = 2, taps.ptr(side), wp.ptr(side), Kp
        d.pre = L.CAffine(None, None, 0)
        d.accumulate = 1
        return [C.byref(d), _stream() if side == "gpu" else 0]

    run_both("vinet_conv3d", mk)
    _cmp(yp.get("gpu"), yp.get("cpu"), TOL[dt], "phase mapping")


WGRAD_CASES = [
    ("pw", (2, 3, 7, 9), 96, 48, (1, 1, 1), (1, 1, 1), (0, 0, 0), False),
    ("sp3", (1, 2, 9, 11), 32, 96, (1, 3, 3), (1, 1, 1), (0, 1, 1), True),
    ("t7s2", (1, 9, 5, 6), 64, 64, (7, 1, 1), (2, 1, 1), (3, 0, 0), False),
    ("dec", (2, 6, 5, 7), 64, 160, (3, 3, 3), (3, 1, 1), (0, 1, 1), True),
    ("splitk", (2, 4, 28, 48), 64, 64, (1, 1, 1), (1, 1, 1), (0, 0, 0), False),
    ("cin24_n208", (1, 2, 6, 6), 24, 208, (1, 3, 3), (1, 1, 1), (0, 1, 1), False),
    # x and dy as channel slices of wider buffers (fused Inception entry conv: dy = [b1r | b2r | b0] of the
    # block's gradient buffer, x = the previous block's output behind its own reduce channels)
    ("pw_slices", (2, 3, 7, 9), 96, 48, (1, 1, 1), (1, 1, 1), (0, 0, 0), True, dict(x_ld=160, x_coff=32, dy_ld=112, dy_coff=16)),
    ("sp3_xslice", (1, 2, 9, 11), 32, 96, (1, 3, 3), (1, 1, 1), (0, 1, 1), True, dict(x_ld=80, x_coff=16)),
    ("pw_big_slices", (4, 8, 28, 48), 192, 176, (1, 1, 1), (1, 1, 1), (0, 0, 0), True, dict(x_ld=304, x_coff=112, dy_ld=432, dy_coff=0)),
]


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("case", WGRAD_CASES, ids=[c[0] for c in WGRAD_CASES])
def test_conv3d_wgrad(case, dt):
    _run_wgrad_case(case, dt)


@pytest.mark.parametrize("case", WGRAD_CASES, ids=[c[0] for c in WGRAD_CASES])
def test_conv3d_wgrad_split_bf16_arithmetic(case):
    """the weight gradient in the VINET_F32S form (fp32 tensors, hi / lo bf16 operands, three MFMAs per product)"""
    _run_wgrad_case(case, E.F32, cdt=L.F32S)


# the 256x256x64 ping-pong wgrad, forced: partial row tiles (N < 256, N > 256), partial segment tiles,
# Cin not a multiple of 64, pending affine, T stride, odd K-tile counts, split-K
WGRAD_PP_CASES = WGRAD_CASES + [
    ("pp_n320_c192", (2, 4, 10, 12), 192, 320, (1, 3, 3), (1, 1, 1), (0, 1, 1), True),
    ("pp_c480_t5", (1, 10, 6, 8), 480, 192, (5, 3, 3), (5, 1, 1), (0, 1, 1), False),
    ("pp_c96_3t", (2, 5, 9, 9), 96, 128, (3, 1, 1), (1, 1, 1), (1, 0, 0), True),
    ("pp_bigm", (4, 8, 28, 48), 64, 192, (1, 3, 3), (1, 1, 1), (0, 1, 1), True),
]


@pytest.mark.parametrize("shape", [3, 4], ids=["tn256", "tn192"])
@pytest.mark.parametrize("case", WGRAD_PP_CASES, ids=[c[0] for c in WGRAD_PP_CASES])
def test_conv3d_wgrad_pingpong(case, shape):
    lib = _lib()
    assert lib.vinet_set_option(b"wgrad_pp", shape) == 0      # 3 / 4: force the 256- / 192-row tile
    try:
        d0 = _run_wgrad_case(case, E.BF16)
        buf = C.create_string_buffer(128)
        assert lib.vinet_conv3d_wgrad_kernel_name(C.byref(d0), buf, 128) == 0 and buf.value.startswith(b"conv_wgrad_pp_kernel<")
    finally:
        lib.vinet_set_option(b"wgrad_pp", 1)


# frame-streaming wgrad of temporal 64 -> 64 convs: stride 2 and 1, 7 / 3 / 2 taps, pending affine, sliced views,
# odd frame counts (window running past the last frame), one clip / several clips
WGRAD_TS_CASES = [
    ("ts_k7s2", (2, 9, 8, 16), 64, 64, (7, 1, 1), (2, 1, 1), (3, 0, 0), True),
    ("ts_k7s2_plain", (1, 12, 8, 8), 64, 64, (7, 1, 1), (2, 1, 1), (3, 0, 0), False),
    ("ts_k3s1", (2, 5, 16, 8), 64, 64, (3, 1, 1), (1, 1, 1), (1, 0, 0), True),
    ("ts_k2s2", (1, 8, 8, 8), 64, 64, (2, 1, 1), (2, 1, 1), (0, 0, 0), False),
    ("ts_k5s1_slices", (3, 6, 8, 8), 64, 64, (5, 1, 1), (1, 1, 1), (2, 0, 0), True, dict(x_ld=160, x_coff=32, dy_ld=112, dy_coff=16)),
    ("ts_k7s2_long", (1, 32, 8, 24), 64, 64, (7, 1, 1), (2, 1, 1), (3, 0, 0), True),
]


@pytest.mark.parametrize("case", WGRAD_TS_CASES, ids=[c[0] for c in WGRAD_TS_CASES])
def test_conv3d_wgrad_tstream(case):
    lib = _lib()
    assert lib.vinet_set_option(b"wgrad_ts", 2) == 0      # force (the heuristic wants thousands of patches)
    try:
        ex = dict(case[8]) if len(case) > 8 else {}
        ex["tline"] = True
        d0 = _run_wgrad_case(case[:8] + (ex,), E.BF16)
        buf = C.create_string_buffer(128)
        assert lib.vinet_conv3d_wgrad_kernel_name(C.byref(d0), buf, 128) == 0 and buf.value.startswith(b"conv_wgrad_ts_kernel<")
    finally:
        lib.vinet_set_option(b"wgrad_ts", 1)


# row-streaming wgrad of kT x 3 x 3 / (kT,1,1) convs with 64 output channels (the 192 -> 64 decoder layer): one, two
# and three K steps per row, 1 / 2 / 5 temporal taps, several channel chunks, sliced views, a 2-row image
WGRAD_RS_CASES = [
    ("rs_w32_k5", (2, 10, 6, 32), 128, 64, (5, 3, 3), (5, 1, 1), (0, 1, 1), False),
    ("rs_w64_k2", (1, 4, 5, 64), 64, 64, (2, 3, 3), (2, 1, 1), (0, 1, 1), False),
    ("rs_w96_k1", (2, 2, 4, 96), 192, 64, (1, 3, 3), (1, 1, 1), (0, 1, 1), False),
    ("rs_slices", (1, 6, 2, 32), 64, 64, (3, 3, 3), (3, 1, 1), (0, 1, 1), False, dict(x_ld=160, x_coff=32, dy_ld=112, dy_coff=16)),
    ("rs_n192", (2, 2, 5, 64), 64, 192, (1, 3, 3), (1, 1, 1), (0, 1, 1), False),
    ("rs_n128_k2", (1, 4, 4, 32), 128, 128, (2, 3, 3), (2, 1, 1), (0, 1, 1), False),
    ("rs_partial_chunks", (1, 2, 4, 64), 96, 80, (1, 3, 3), (1, 1, 1), (0, 1, 1), False),
    # several image rows per step (W = 48 / 24), image heights that do not divide into steps, partial chunks
    ("rs_w48", (2, 4, 7, 48), 96, 128, (2, 3, 3), (2, 1, 1), (0, 1, 1), False),
    ("rs_w24", (1, 3, 14, 24), 160, 64, (3, 3, 3), (3, 1, 1), (0, 1, 1), False),
    ("rs_w24_h3", (2, 1, 3, 24), 64, 64, (1, 3, 3), (1, 1, 1), (0, 1, 1), False, dict(x_ld=96, x_coff=16)),
    # rows of 128 / 160 / 192 positions (4 - 6 K steps, > 64 KB of LDS), a 32-channel output (64 -> 32 k2x3x3 of the decoder)
    ("rs_w192_n32", (1, 4, 5, 192), 64, 32, (2, 3, 3), (2, 1, 1), (0, 1, 1), False),
    ("rs_w128", (2, 1, 4, 128), 96, 64, (1, 3, 3), (1, 1, 1), (0, 1, 1), False),
    ("rs_w160", (1, 3, 3, 160), 64, 72, (3, 3, 3), (3, 1, 1), (0, 1, 1), False),
]


@pytest.mark.parametrize("case", WGRAD_RS_CASES, ids=[c[0] for c in WGRAD_RS_CASES])
def test_conv3d_wgrad_rowstream(case):
    lib = _lib()
    assert lib.vinet_set_option(b"wgrad_rs", 2) == 0
    try:
        ex = dict(case[8]) if len(case) > 8 else {}
        ex["tline"] = 4
        d0 = _run_wgrad_case(case[:8] + (ex,), E.BF16)
        buf = C.create_string_buffer(128)
        assert lib.vinet_conv3d_wgrad_kernel_name(C.byref(d0), buf, 128) == 0 and buf.value.startswith(b"conv_wgrad_rs_kernel<")
    finally:
        lib.vinet_set_option(b"wgrad_rs", 1)


# frame-streaming wgrad of wide 3x1x1 / s1 / p1 temporal convs (H*W a multiple of 96): 192 / 128 / 64 output channels
# per workgroup, pending BN + ReLU and plain inputs, partial channel chunks, sliced views, one- and two-frame clips
WGRAD_TF_CASES = [
    ("tf_192_pre", (2, 5, 8, 12), 192, 192, (3, 1, 1), (1, 1, 1), (1, 0, 0), True),
    ("tf_128_plain", (1, 4, 8, 24), 128, 128, (3, 1, 1), (1, 1, 1), (1, 0, 0), False),
    ("tf_96_pre", (3, 3, 4, 24), 96, 96, (3, 1, 1), (1, 1, 1), (1, 0, 0), True),
    ("tf_208_pre", (1, 6, 8, 12), 208, 208, (3, 1, 1), (1, 1, 1), (1, 0, 0), True),
    ("tf_32_64", (2, 2, 16, 12), 32, 64, (3, 1, 1), (1, 1, 1), (1, 0, 0), True),
    ("tf_t1", (2, 1, 8, 12), 64, 160, (3, 1, 1), (1, 1, 1), (1, 0, 0), True),
    ("tf_slices", (2, 7, 8, 12), 128, 192, (3, 1, 1), (1, 1, 1), (1, 0, 0), True, dict(x_ld=160, x_coff=32, dy_ld=256, dy_coff=48)),
]


@pytest.mark.parametrize("case", WGRAD_TF_CASES, ids=[c[0] for c in WGRAD_TF_CASES])
def test_conv3d_wgrad_tframes(case):
    lib = _lib()
    assert lib.vinet_set_option(b"wgrad_tf", 2) == 0
    try:
        ex = dict(case[8]) if len(case) > 8 else {}
        ex["tline"] = True
        d0 = _run_wgrad_case(case[:8] + (ex,), E.BF16)
        buf = C.create_string_buffer(128)
        assert lib.vinet_conv3d_wgrad_kernel_name(C.byref(d0), buf, 128) == 0 and buf.value.startswith(b"conv_wgrad_tf_kernel<")
    finally:
        lib.vinet_set_option(b"wgrad_tf", 1)


def _run_wgrad_case(case, dt, cdt=None):
    name, (B, T, H, W), Cin, N, k, s, p, pre = case[:8]
    ex = case[8] if len(case) > 8 else {}
    oT, oH, oW = [(d + 2 * pp - kk) // ss + 1 for d, kk, ss, pp in zip((T, H, W), k, s, p)]
    xp, xmk = view_pair(B, T, H, W, Cin, dt, "wx" + name, 1, ld=ex.get("x_ld"), c_off=ex.get("x_coff", 0))
    dp, dmk = view_pair(B, oT, oH, oW, N, dt, "wd" + name, 2, ld=ex.get("dy_ld"), c_off=ex.get("dy_coff", 0))
    ntaps = k[0] * k[1] * k[2]
    Kp = E.rup(Cin, 32)
    dw = Pair(torch.zeros(ntaps * N * Kp))
    taps = Pair(torch.tensor(_fwd_taps(k, p), dtype=torch.int32))
    ps, ph = fvec("wps" + name, Cin, 4, 0.5, 1.5), fvec("wph" + name, Cin, 5)

    def mk(side):
        d = L.CWgradDesc()
        d.dtype, d.mode = (dt if cdt is None else cdt), 0
        d.x, d.dy = xmk(side).ct(), dmk(side).ct()
        d.sT, d.sH, d.sW = s
        d.ntaps, d.taps, d.dw, d.Kp = ntaps, taps.ptr(side), dw.ptr(side), Kp
        d.pre = L.CAffine(ps.ptr(side), ph.ptr(side), 1) if pre else L.CAffine(None, None, 0)
        if ex.get("tline"):
            d.tline, d.tpad = (1 if ex["tline"] is True else ex["tline"]), p[0]
        return [C.byref(d), _stream() if side == "gpu" else 0]

    run_both("vinet_conv3d_wgrad", mk)
    _cmp(dw.get("gpu"), dw.get("cpu"), (3e-5 if cdt is None else 2e-4) if dt == E.F32 else 2e-2, "wgrad " + name)
    return mk("gpu")[0]._obj


@pytest.mark.parametrize("dt", DTS)
def test_conv3d_wgrad_stem(dt):
    B, T, H, W, N = 1, 2, 18, 22, 64
    oH, oW = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
    x = torch.zeros(B, T, H, W, 4)
    x[..., :3] = _rand("wsx", (B, T, H, W, 3), 1)
    xp = Pair(x.reshape(-1).to(E.TORCH_DT[dt]))
    dp, dmk = view_pair(B, T, oH, oW, N, dt, "wsd", 2)
    dw = Pair(torch.zeros(7 * N * 32))
    taps = Pair(torch.tensor([(0, kh - 3, -3, kh) for kh in range(7)], dtype=torch.int32))

    def mk(side):
        d = L.CWgradDesc()
        d.dtype, d.mode = dt, 1
        buf = xp.cpu if side == "cpu" else xp.gpu
        d.x = E.View(buf, 0, B, T, H, W, 4, 4, T * H * W * 4, dt).ct()
        d.dy = dmk(side).ct()
        d.sT, d.sH, d.sW = 1, 2, 2
        d.ntaps, d.taps, d.dw, d.Kp = 7, taps.ptr(side), dw.ptr(side), 32
        d.pre = L.CAffine(None, None, 0)
        return [C.byref(d), _stream() if side == "gpu" else 0]

    run_both("vinet_conv3d_wgrad", mk)
    _cmp(dw.get("gpu"), dw.get("cpu"), 3e-5 if dt == E.F32 else 2e-2, "wgrad stem")


@pytest.mark.parametrize("dt", DTS)
def test_pack_unpack(dt):
    N, Cin, nt = 48, 24, 9
    w = Pair(_rand("pw", (N * Cin * nt,), 1))
    for transpose in (0, 1):
        n = nt * (N if not transpose else Cin) * E.rup(Cin if not transpose else N, 32)
        out = Pair(torch.zeros(n).to(E.TORCH_DT[dt]))
        run_both("vinet_pack_weights", lambda s: [w.ptr(s), N, Cin, nt, transpose, 0, dt, out.ptr(s), _stream() if s == "gpu" else 0])
        assert torch.equal(out.get("gpu"), out.get("cpu"))
    ws = Pair(_rand("pws", (64 * 3 * 49,), 2))
    outs = Pair(torch.zeros(7 * 64 * 32).to(E.TORCH_DT[dt]))
    run_both("vinet_pack_weights", lambda s: [ws.ptr(s), 64, 3, 49, 0, 1, dt, outs.ptr(s), _stream() if s == "gpu" else 0])
    assert torch.equal(outs.get("gpu"), outs.get("cpu"))
    if dt == E.F32:
        dw = Pair(_rand("udw", (nt * N * 32,), 3))
        g = Pair(_rand("ug", (N * Cin * nt,), 4))
        run_both("vinet_unpack_wgrad", lambda s: [dw.ptr(s), N, Cin, nt, 0, 1, g.ptr(s), _stream() if s == "gpu" else 0])
        _cmp(g.get("gpu"), g.get("cpu"), 1e-6, "unpack")
        dws = Pair(_rand("udws", (7 * 64 * 32,), 5))
        gs = Pair(torch.zeros(64 * 3 * 49))
        run_both("vinet_unpack_wgrad", lambda s: [dws.ptr(s), 64, 3, 49, 1, 0, gs.ptr(s), _stream() if s == "gpu" else 0])
        _cmp(gs.get("gpu"), gs.get("cpu"), 1e-6, "unpack stem")
        # both jobs in one launch (vinet_unpack_wgrad_multi): accumulate + hand the workspaces back zeroed
        dw2, dws2 = Pair(_rand("udw2", (nt * N * 32,), 6)), Pair(_rand("udws2", (7 * 64 * 32,), 7))
        g2, gs2 = Pair(_rand("ug2", (N * Cin * nt,), 8)), Pair(_rand("ugs2", (64 * 3 * 49,), 9))

        def table(s):
            rows = [[dw2.ptr(s), g2.ptr(s), N, Cin, nt, 0, 0, 0], [dws2.ptr(s), gs2.ptr(s), 64, 3, 49, 1, nt * N * 32, 0],
                    [0, 0, 0, 0, 0, 0, nt * N * 32 + 7 * 64 * 32, 0]]
            t = torch.tensor(rows, dtype=torch.int64)
            return t.cuda() if s == "gpu" else t
        tabs = {}
        run_both("vinet_unpack_wgrad_multi", lambda s: [tabs.setdefault(s, table(s)).data_ptr(), 2, nt * N * 32 + 7 * 64 * 32, 3, _stream() if s == "gpu" else 0])
        _cmp(g2.get("gpu"), g2.get("cpu"), 1e-6, "multi unpack")
        _cmp(gs2.get("gpu"), gs2.get("cpu"), 1e-6, "multi unpack stem")
        assert float(dw2.get("gpu").abs().max()) == 0.0 and float(dws2.get("gpu").abs().max()) == 0.0
        assert float(dw2.get("cpu").abs().max()) == 0.0 and float(dws2.get("cpu").abs().max()) == 0.0


PACK_SHAPES = [(48, 24, 9), (100, 40, 27), (192, 480, 45), (16, 1, 64), (32, 16, 32), (7, 5, 1), (130, 70, 3), (256, 832, 1),
               (8, 8, 300), (33, 65, 2), (64, 64, 7)]


@pytest.mark.parametrize("tiled", [1, 0], ids=["tiled", "elementwise"])
@pytest.mark.parametrize("pdt", [L.F32, L.BF16, L.F32S], ids=["f32", "bf16", "f32s"])
def test_pack_weights_multi_matches_single_packs(pdt, tiled):
    """vinet_pack_weights_multi (LDS-tiled kernel and its element-wise predecessor, option pack_tiled) against one
    vinet_pack_weights per job, bit for bit: plain and transposed packs of odd shapes (row / column tails, 1 ... 300 taps, the
    64-tap SoundNet conv, padding columns), the 7 x 7 stem, and the side-by-side transposed form of the joint entry convs
    (ld / col: only the job's own columns may be written)"""
    lib = _lib()
    dev = _dev()
    es = 4 if pdt in (L.F32, L.F32S) else 2
    tdt = torch.float32 if es == 4 else torch.bfloat16
    jobs, refs, outs, keep = [], [], [], []
    off = 0
    for i, (N, Cin, nt) in enumerate(PACK_SHAPES):
        w = _rand("pmw%d" % i, (N * Cin * nt,), 1).cuda()
        keep.append(w)
        for tr in (0, 1):
            rows, kp = (N, E.rup(Cin, 32)) if not tr else (Cin, E.rup(N, 32))
            n = nt * rows * kp
            ref = torch.full((n,), 7.0, dtype=tdt, device=dev)
            assert lib.vinet_pack_weights(w.data_ptr(), N, Cin, nt, tr, 0, pdt, ref.data_ptr(), _stream()) == 0
            out = torch.full((n,), 7.0, dtype=tdt, device=dev)
            jobs.append([w.data_ptr(), out.data_ptr(), N, Cin, nt, tr, off, 0])
            off += n
            refs.append(ref); outs.append(out)
    # the stem
    ws = _rand("pmws", (64 * 3 * 49,), 2).cuda()
    refs_ = torch.full((7 * 64 * 32,), 7.0, dtype=tdt, device=dev)
    assert lib.vinet_pack_weights(ws.data_ptr(), 64, 3, 49, 0, 1, pdt, refs_.data_ptr(), _stream()) == 0
    outs_ = torch.full((7 * 64 * 32,), 7.0, dtype=tdt, device=dev)
    jobs.append([ws.data_ptr(), outs_.data_ptr(), 64, 3, 49, 2, off, 0])
    off += 7 * 64 * 32
    refs.append(refs_); outs.append(outs_)
    # two pointwise convs over one 40-channel input, transposed side by side: rows of ld = 96 elements, columns [0, 24) and [32, 82)
    if pdt != L.F32S:      # (the split form's pack permutes K inside 32-column groups: the engine packs joint convs for it the same way, covered by the model tests)
        joint = torch.full((40 * 96,), 7.0, dtype=tdt, device=dev)
        jref = joint.clone().view(40, 96)
        for (Nm, col, seed) in ((24, 0, 3), (50, 32, 4)):
            wm = _rand("pmj%d" % col, (Nm * 40,), seed).cuda()
            keep.append(wm)
            jobs.append([wm.data_ptr(), joint.data_ptr(), Nm, 40, 1, 1, off, 96 | (col << 32)])
            off += 40 * E.rup(Nm, 32)
            jref[:, col:col + Nm] = wm.view(Nm, 40).t().to(tdt)
        refs.append(jref.reshape(-1)); outs.append(joint)
    table = torch.tensor(jobs + [[0, 0, 0, 0, 0, 0, off, 0]], dtype=torch.int64).cuda()
    assert lib.vinet_set_option(b"pack_tiled", tiled) == 0
    try:
        rc = lib.vinet_pack_weights_multi(table.data_ptr(), len(jobs), off, pdt, _stream())
        assert rc == 0, lib.vinet_last_error()
        torch.cuda.synchronize()
    finally:
        lib.vinet_set_option(b"pack_tiled", 1)
    for k, (o, r) in enumerate(zip(outs, refs)):
        assert torch.equal(o.view(torch.int32 if es == 4 else torch.int16), r.view(torch.int32 if es == 4 else torch.int16)), "job %d differs" % k


@pytest.mark.parametrize("tiled", [1, 0], ids=["tiled", "elementwise"])
def test_unpack_wgrad_multi_matches_single_unpacks(tiled):
    """vinet_unpack_wgrad_multi (LDS-tiled / element-wise) against one vinet_unpack_wgrad per job: grad += dw bit for bit, every
    packed element (padding columns included) handed back zero"""
    lib = _lib()
    jobs, checks = [], []
    off = 0
    for i, (N, Cin, nt) in enumerate(PACK_SHAPES + [(64, 3, 49)]):
        stem = 1 if (N, Cin, nt) == (64, 3, 49) else 0
        nsl, kp = (7, 32) if stem else (nt, E.rup(Cin, 32))
        dw = _rand("umd%d" % i, (nsl * N * kp,), 5).cuda()
        g0 = _rand("umg%d" % i, (N * Cin * nt,), 6).cuda()
        ref, dwr = g0.clone(), dw.clone()
        assert lib.vinet_unpack_wgrad(dwr.data_ptr(), N, Cin, nt, stem, 3, ref.data_ptr(), _stream()) == 0
        g = g0.clone()
        jobs.append([dw.data_ptr(), g.data_ptr(), N, Cin, nt, stem, off, 0])
        off += nsl * N * kp
        checks.append((dw, g, ref, dwr))
    table = torch.tensor(jobs + [[0, 0, 0, 0, 0, 0, off, 0]], dtype=torch.int64).cuda()
    assert lib.vinet_set_option(b"pack_tiled", tiled) == 0
    try:
        rc = lib.vinet_unpack_wgrad_multi(table.data_ptr(), len(jobs), off, 3, _stream())
        assert rc == 0, lib.vinet_last_error()
        torch.cuda.synchronize()
    finally:
        lib.vinet_set_option(b"pack_tiled", 1)
    for k, (dw, g, ref, dwr) in enumerate(checks):
        assert torch.equal(g, ref), "job %d: gradients differ" % k
        assert float(dw.abs().max()) == 0.0 and float(dwr.abs().max()) == 0.0, "job %d: workspace not handed back zeroed" % k


@pytest.mark.parametrize("dt", DTS)
def test_import_export_copy(dt):
    B, Cc, T, H, W = 2, 3, 4, 6, 10
    base = _rand("imp", (B, T, Cc, H, W), 1)     # train.py hands over a permuted view of this
    src = Pair(base)
    sb, st, sc, sh, sw = base.stride()
    dp, dmk = view_pair(B, T, H, W, 4, dt, "impd", 2)
    run_both("vinet_import_ncdhw", lambda s: [src.ptr(s), sb, sc, st, sh, sw, Cc, C.byref(dmk(s).ct()), dt, _stream() if s == "gpu" else 0])
    assert torch.equal(dp.get("gpu"), dp.get("cpu"))
    got = dp.get("gpu").float().view(B, T, H, W, 4)
    assert torch.equal(got[..., 3], torch.zeros_like(got[..., 3]))
    _cmp(got[..., :3], base.permute(0, 1, 3, 4, 2).to(E.TORCH_DT[dt]).float(), 0.0, "import values")
    # export with a pending affine
    Cx = 16
    xp, xmk = view_pair(B, T, H, W, Cx, dt, "expx", 3, ld=24, c_off=8)
    sc_, sh_ = fvec("exps", Cx, 4, 0.5, 1.5), fvec("exph", Cx, 5)
    out = Pair(torch.zeros(B, Cx, T, H, W))
    osb, osc, ost, osh, osw = out.cpu.stride()
    run_both("vinet_export_ncdhw", lambda s: [C.byref(xmk(s).ct()), dt, L.CAffine(sc_.ptr(s), sh_.ptr(s), 1), out.ptr(s), osb, osc, ost, osh, osw, 0, _stream() if s == "gpu" else 0])
    _cmp(out.get("gpu"), out.get("cpu"), 1e-6, "export")
    for odt in DTS:
        yp, ymk = view_pair(B, T, H, W, Cx, odt, "cpy", 6, t_total=T + 2, t_off=1)
        run_both("vinet_copy_affine", lambda s: [C.byref(xmk(s).ct()), dt, L.CAffine(sc_.ptr(s), sh_.ptr(s), 1), C.byref(ymk(s).ct()), odt, 1, _stream() if s == "gpu" else 0])
        _cmp(yp.get("gpu"), yp.get("cpu"), 1e-6 if odt == E.F32 else 1e-2, "copy_affine")


@pytest.mark.parametrize("acc", [0, 1])
@pytest.mark.parametrize("aff", [0, 1, 2], ids=["plain", "affine_relu", "relu_only"])
def test_copy_affine_streaming(acc, aff):
    """the 8-channel streaming copy (bf16, >= 65536 voxels): sliced views on both sides, affine + ReLU, accumulate"""
    B, T, H, W, Cc = 2, 2, 128, 128, 24
    dt = E.BF16
    sp, smk = view_pair(B, T, H, W, Cc, dt, "cas", 1, ld=40, c_off=8)
    dp, dmk = view_pair(B, T, H, W, Cc, dt, "cad", 2, ld=56, c_off=16)
    ps, ph = fvec("caps", Cc, 3, 0.5, 1.5), fvec("caph", Cc, 4)

    def pre(side):
        if aff == 1:
            return L.CAffine(ps.ptr(side), ph.ptr(side), 1)
        return L.CAffine(None, None, 1 if aff == 2 else 0)
    run_both("vinet_copy_affine", lambda s: [C.byref(smk(s).ct()), dt, pre(s), C.byref(dmk(s).ct()), dt, acc, _stream() if s == "gpu" else 0])
    _cmp(dp.get("gpu"), dp.get("cpu"), 1e-2, "copy_affine streaming")


@pytest.mark.parametrize("aff", [0, 1])
def test_split_bf16_planes(aff):
    """vinet_split_bf16: fp32 view (+ pending BatchNorm + ReLU) -> hi = bf16(v), lo = bf16(v - hi) planes, bit-identical with the model;
    hi + lo reproduces v to 2^-16 relative"""
    B, T, H, W, Cc = 2, 3, 5, 7, 24
    sp, smk = view_pair(B, T, H, W, Cc, E.F32, "spx", 1, ld=40, c_off=8)
    hp, hmk = view_pair(B, T, H, W, Cc, E.BF16, "sph", 2, ld=32, c_off=8)
    lp, lmk = view_pair(B, T, H, W, Cc, E.BF16, "spl", 3)
    ps, ph = fvec("spps", Cc, 3, 0.5, 1.5), fvec("spph", Cc, 4)
    run_both("vinet_split_bf16", lambda s: [C.byref(smk(s).ct()), L.CAffine(ps.ptr(s), ph.ptr(s), 1) if aff else L.CAffine(None, None, 0),
                                            C.byref(hmk(s).ct()), C.byref(lmk(s).ct()), _stream() if s == "gpu" else 0])
    if not aff:      # (with the affine the library's fused multiply-add and the model's multiply + add differ in the last bit of v)
        assert torch.equal(hp.get("gpu"), hp.get("cpu")) and torch.equal(lp.get("gpu"), lp.get("cpu"))
    recg = hmk("gpu").torch5().float().cpu() + lmk("gpu").torch5().float().cpu()
    v = smk("cpu").torch5().float()
    if aff:
        v = torch.relu(v * ps.cpu + ph.cpu)
    rec = hmk("cpu").torch5().float() + lmk("cpu").torch5().float()
    for r in (rec, recg):
        assert float((r - v).abs().max()) <= 2.0 ** -15 * max(1.0, float(v.abs().max()))


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("Cc", [16, 24, 64, 208, 528, 1024])
def test_bn_kernels(dt, Cc):
    B, T, H, W = 2, 3, 5, 7
    xp, xmk = view_pair(B, T, H, W, Cc, dt, "bnx", 1)
    gp, gmk = view_pair(B, T, H, W, Cc, dt, "bng", 2)
    rows = AbiEmulator().vinet_stats_rows(xmk("cpu").ct())
    assert _lib().vinet_stats_rows(C.byref(xmk("gpu").ct())) == rows
    part = Pair(torch.zeros(rows * 2 * Cc))
    run_both("vinet_channel_stats", lambda s: [C.byref(xmk(s).ct()), dt, part.ptr(s), _stream() if s == "gpu" else 0])
    _cmp(part.get("gpu").view(rows, 2, Cc).double().sum(0), part.get("cpu").view(rows, 2, Cc).double().sum(0), 1e-5, "channel_stats")
    # finalize
    n = float(B * T * H * W)
    gam, bet = fvec("g", Cc, 3, 0.5, 1.5), fvec("b", Cc, 4)
    rm, rv = fvec("rm", Cc, 5), fvec("rv", Cc, 6, 0.5, 1.5)
    mean, istd, sc, sh = [Pair(torch.zeros(Cc)) for _ in range(4)]
    stats = Pair(part.cpu.clone())
    run_both("vinet_bn_finalize", lambda s: [stats.ptr(s), rows, Cc, 0, n, gam.ptr(s), bet.ptr(s), 1e-3, 0.001, rm.ptr(s), rv.ptr(s), mean.ptr(s), istd.ptr(s), sc.ptr(s), sh.ptr(s), _stream() if s == "gpu" else 0])
    for a, nm in ((mean, "mean"), (istd, "invstd"), (sc, "scale"), (sh, "shift"), (rm, "rm"), (rv, "rv")):
        _cmp(a.get("gpu"), a.get("cpu"), 2e-6, "bn_finalize " + nm)
    # fold
    fs, fh, fi = [Pair(torch.zeros(Cc)) for _ in range(3)]
    cb = fvec("cb", Cc, 7)
    run_both("vinet_bn_fold", lambda s: [gam.ptr(s), bet.ptr(s), rm.ptr(s), rv.ptr(s), cb.ptr(s), 1e-3, Cc, fs.ptr(s), fh.ptr(s), fi.ptr(s), _stream() if s == "gpu" else 0])
    for a in (fs, fh, fi):
        _cmp(a.get("gpu"), a.get("cpu"), 2e-6, "bn_fold")
    # backward
    aff = lambda s: L.CAffine(sc.ptr(s), sh.ptr(s), 1)
    part2 = Pair(torch.zeros(rows * 2 * Cc))
    run_both("vinet_bn_bwd_reduce", lambda s: [C.byref(gmk(s).ct()), C.byref(xmk(s).ct()), dt, aff(s), mean.ptr(s), istd.ptr(s), part2.ptr(s), _stream() if s == "gpu" else 0])
    _cmp(part2.get("gpu").view(rows, 2, Cc).double().sum(0), part2.get("cpu").view(rows, 2, Cc).double().sum(0), 2e-5, "bn_bwd_reduce")
    dg, db, c1, c2 = [Pair(torch.zeros(Cc)) for _ in range(4)]
    p2 = Pair(part2.cpu.clone())
    run_both("vinet_bn_bwd_finalize", lambda s: [p2.ptr(s), rows, Cc, 0, n, sc.ptr(s), 1, dg.ptr(s), db.ptr(s), istd.ptr(s), c1.ptr(s), c2.ptr(s), _stream() if s == "gpu" else 0])
    for a in (dg, db, c1, c2):
        _cmp(a.get("gpu"), a.get("cpu"), 2e-6, "bn_bwd_finalize")
    dxp, dxmk = view_pair(B, T, H, W, Cc, dt, "bndx", 8)
    run_both("vinet_bn_bwd_apply", lambda s: [C.byref(gmk(s).ct()), C.byref(xmk(s).ct()), dt, aff(s), mean.ptr(s), istd.ptr(s), c1.ptr(s), c2.ptr(s), C.byref(dxmk(s).ct()), _stream() if s == "gpu" else 0])
    _cmp(dxp.get("gpu"), dxp.get("cpu"), 1e-5 if dt == E.F32 else 2e-2, "bn_bwd_apply")
    if dt == E.F32 and Cc % 8 == 0:
        # the VINET_F32S training form: the same pass also writes the hi / lo bf16 planes of its result (vinet_split_bf16's arithmetic)
        dsp, dsmk = view_pair(B, T, H, W, Cc, dt, "bnds", 9)
        hp, hmk = view_pair(B, T, H, W, Cc, E.BF16, "bnhi", 10, ld=Cc + 8)
        lp, lmk = view_pair(B, T, H, W, Cc, E.BF16, "bnlo", 11)
        run_both("vinet_bn_bwd_apply_split", lambda s: [C.byref(gmk(s).ct()), C.byref(xmk(s).ct()), aff(s), mean.ptr(s), istd.ptr(s), c1.ptr(s), c2.ptr(s),
                                                        C.byref(dsmk(s).ct()), C.byref(hmk(s).ct()), C.byref(lmk(s).ct()), _stream() if s == "gpu" else 0])
        assert torch.equal(dsp.get("gpu"), dxp.get("gpu")), "bn_bwd_apply_split: dx differs from vinet_bn_bwd_apply's"
        v = dsp.get("gpu").view(B, T, H, W, Cc)
        hi = E.View(hp.get("gpu"), 0, B, T, H, W, Cc, Cc + 8, T * H * W * (Cc + 8), E.BF16).torch5()
        lo = lp.get("gpu").view(B, T, H, W, Cc)
        assert torch.equal(hi, v.bfloat16()) and torch.equal(lo, (v - v.bfloat16().float()).bfloat16()), "hi / lo planes are not the split of dx"
    # channel_sum with folding
    out = Pair(torch.ones(Cc // 4))
    ws = Pair(torch.zeros(rows * 2 * Cc))
    run_both("vinet_channel_sum", lambda s: [C.byref(xmk(s).ct()), dt, ws.ptr(s), Cc // 4, out.ptr(s), 1, _stream() if s == "gpu" else 0])
    _cmp(out.get("gpu"), out.get("cpu"), 2e-5, "channel_sum")


@pytest.mark.parametrize("rows,Cc,out_rows", [(5000, 64, 256), (4097, 176, 256), (300, 24, 7), (64, 832, 64)])
def test_bn_partials_fold(rows, Cc, out_rows):
    per = (rows + out_rows - 1) // out_rows
    out_rows = (rows + per - 1) // per
    p = Pair(_rand("pf", (rows * 2 * Cc,), 1))
    o = Pair(torch.zeros(out_rows * 2 * Cc))
    run_both("vinet_bn_partials_fold", lambda s: [p.ptr(s), rows, Cc, o.ptr(s), out_rows, _stream() if s == "gpu" else 0])
    _cmp(o.get("gpu"), o.get("cpu"), 1e-6, "bn_partials_fold")


@pytest.mark.parametrize("dt", DTS)
def test_act_bwd(dt):
    B, T, H, W, Cc = 2, 2, 5, 6, 32
    zp, zmk = view_pair(B, T, H, W, Cc, dt, "az", 1)
    gp, gmk = view_pair(B, T, H, W, Cc, dt, "ag", 2)
    for act in (1, 2):
        yp, ymk = view_pair(B, T, H, W, Cc, dt, "ay", 3)
        run_both("vinet_act_bwd", lambda s: [C.byref(gmk(s).ct()), dt, C.byref(zmk(s).ct()), dt, act, C.byref(ymk(s).ct()), dt, _stream() if s == "gpu" else 0])
        _cmp(yp.get("gpu"), yp.get("cpu"), 1e-6 if dt == E.F32 else 1e-2, "act_bwd")
    # mixed: fp32 dz and z (head), activation-dtype dy
    zp, zmk = view_pair(1, 1, 1, 96, 4, E.F32, "az2", 4)
    gp, gmk = view_pair(1, 1, 1, 96, 4, E.F32, "ag2", 5)
    yp, ymk = view_pair(1, 1, 1, 96, 4, dt, "ay2", 6)
    run_both("vinet_act_bwd", lambda s: [C.byref(gmk(s).ct()), E.F32, C.byref(zmk(s).ct()), E.F32, 2, C.byref(ymk(s).ct()), dt, _stream() if s == "gpu" else 0])
    _cmp(yp.get("gpu"), yp.get("cpu"), 1e-6 if dt == E.F32 else 1e-2, "act_bwd mixed")


POOLS = [((1, 3, 3), (1, 2, 2), (0, 1, 1)), ((3, 3, 3), (2, 2, 2), (1, 1, 1)), ((3, 3, 3), (1, 1, 1), (1, 1, 1)),
         ((2, 1, 1), (2, 1, 1), (0, 0, 0)), ((1, 2, 2), (1, 2, 2), (0, 0, 0)), ((4, 1, 1), (2, 1, 2), (0, 0, 0)),
         ((8, 1, 1), (8, 1, 1), (0, 0, 0))]


@pytest.mark.parametrize("form", [2, 3, 4])
@pytest.mark.parametrize("dt", DTS)
def test_maxpool_k3s1_twalk_backward(dt, form):
    """the T-walking 3x3x3/s1 backward (chosen for large tensors only) forced on the small test shape
    (2: bf16 takes the all-loads-up-front form with EXEC-mask routing; 3: the conditional-load form for every dtype;
    4: bf16 all-loads-up-front with compare / select / add routing)"""
    lib = _lib()
    assert lib.vinet_set_option(b"pool_twalk", form) == 0 and lib.vinet_set_option(b"pool_lds", 0) == 0
    try:
        test_maxpool(dt, ((3, 3, 3), (1, 1, 1), (1, 1, 1)))
    finally:
        lib.vinet_set_option(b"pool_twalk", 1)
        lib.vinet_set_option(b"pool_lds", 1)


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("acc", [0, 1])
@pytest.mark.parametrize("Cc,hw", [(136, (17, 8)), (64, (8, 24))])
def test_maxpool_k3s2_block_backward_shapes(Cc, hw, acc, dt):
    """the 2x2x2-block 3x3x3/s2 backward on more shapes: odd and even extents, several channel octets, store and accumulate"""
    test_maxpool(dt, POOLS[1], Cc=Cc, acc=acc, hw=hw)


@pytest.mark.parametrize("acc", [0, 1])
@pytest.mark.parametrize("Cc,hw", [(136, (17, 8)), (64, (8, 24))])
def test_maxpool_k3s1_backward_shapes(Cc, hw, acc):
    """the bf16 T-walking 3x3x3/s1 backward on more shapes: several channel octets per voxel, store and accumulate"""
    lib = _lib()
    assert lib.vinet_set_option(b"pool_twalk", 2) == 0
    try:
        test_maxpool(E.BF16, ((3, 3, 3), (1, 1, 1), (1, 1, 1)), Cc=Cc, acc=acc, hw=hw)
    finally:
        lib.vinet_set_option(b"pool_twalk", 1)


@pytest.mark.parametrize("dt", DTS)
def test_maxpool_133s2_generic_backward(dt):
    """1x3x3/s(1,2,2) and 3x3x3/s2 through the generic gather (the 2x2 / 2x2x2-block kernels are their defaults)"""
    lib = _lib()
    assert lib.vinet_set_option(b"pool_blk", 0) == 0
    try:
        test_maxpool(dt, POOLS[0])
        test_maxpool(dt, POOLS[1])
    finally:
        lib.vinet_set_option(b"pool_blk", 1)


def test_maxpool_k3s1_lds_forward_bf16_fp32_compare():
    """bf16 through the fp32-compare LDS kernel (the packed-key kernel is the bf16 default)"""
    lib = _lib()
    assert lib.vinet_set_option(b"pool_pk", 0) == 0
    try:
        test_maxpool_k3s1_lds_forward(E.BF16)
        for ksp in POOLS[:3]:                      # the generic 8-channel kernel, fp32 compare
            test_maxpool(E.BF16, ksp)
    finally:
        lib.vinet_set_option(b"pool_pk", 1)


@pytest.mark.parametrize("dt", DTS)
def test_maxpool_k3s1_lds_forward(dt):
    """the LDS halo-tile 3x3x3/s1 forward (chosen for large tensors only) forced on the small test shape
    (partial spatial tiles, partial channel group)"""
    lib = _lib()
    assert lib.vinet_set_option(b"pool_lds", 2) == 0
    try:
        test_maxpool(dt, ((3, 3, 3), (1, 1, 1), (1, 1, 1)))
    finally:
        lib.vinet_set_option(b"pool_lds", 1)


@pytest.mark.parametrize("T", [7, 1])
@pytest.mark.parametrize("Cc,hw", [(24, (9, 10)), (136, (17, 8)), (64, (34, 32))])
@pytest.mark.parametrize("ksp", POOLS[:2], ids=["k133s2", "k333s2"])
def test_maxpool_strided_odd_extents(ksp, Cc, hw, T):
    """the strided 3 x 3 spatial pools (forward, and the 2x2 / 2x2x2-block backward kernels) on odd / unit clip lengths, odd and
    even H / W, several channel octets"""
    test_maxpool(E.BF16, ksp, Cc=Cc, hw=hw, T=T)


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("ksp", POOLS, ids=[str(p[0]) + str(p[1]) for p in POOLS])
def test_maxpool(dt, ksp, Cc=24, acc=1, hw=(9, 10), T=8):
    k, s, p = ksp
    B, (H, W) = 2, hw
    od = [(d + 2 * pp - kk) // ss + 1 for d, kk, ss, pp in zip((T, H, W), k, s, p)]
    xp, xmk = view_pair(B, T, H, W, Cc, dt, "px", 1, ld=Cc + 16, c_off=8)      # input / its gradient: slices of a wider buffer
    yp, ymk = view_pair(B, od[0], od[1], od[2], Cc, dt, "py", 2)
    am = Pair(torch.zeros(B * od[0] * od[1] * od[2] * Cc, dtype=torch.uint8))
    ps, ph = fvec("pps", Cc, 3, -1.5, 1.5), fvec("pph", Cc, 4)

    def pd():
        return L.CPoolDesc(dt, k[0], k[1], k[2], s[0], s[1], s[2], p[0], p[1], p[2])
    run_both("vinet_maxpool3d", lambda sd: [C.byref(pd()), C.byref(xmk(sd).ct()), L.CAffine(ps.ptr(sd), ph.ptr(sd), 1), C.byref(ymk(sd).ct()), am.ptr(sd), _stream() if sd == "gpu" else 0])
    _cmp(yp.get("gpu"), yp.get("cpu"), 1e-6 if dt == E.F32 else 1e-2, "maxpool")
    # ReLU creates ties at 0: indices may only differ where the pooled value is 0
    diff = am.get("gpu") != am.get("cpu")
    assert not bool((diff & (yp.get("cpu").float() != 0)).any())
    gp, gmk = view_pair(B, od[0], od[1], od[2], Cc, dt, "pg", 5)
    dxp, dxmk = view_pair(B, T, H, W, Cc, dt, "pdx", 6, ld=Cc + 16, c_off=8)
    amc = Pair(am.cpu.clone())
    run_both("vinet_maxpool3d_bwd", lambda sd: [C.byref(pd()), C.byref(gmk(sd).ct()), amc.ptr(sd), C.byref(dxmk(sd).ct()), acc, _stream() if sd == "gpu" else 0])
    _cmp(dxp.get("gpu"), dxp.get("cpu"), 1e-5 if dt == E.F32 else 2e-2, "maxpool bwd")


@pytest.mark.parametrize("dt", DTS)
def test_upsample_quad_kernels(dt):
    """the 4-channel one-output-per-lane kernels (8-channel block kernels are the default)"""
    lib = _lib()
    assert lib.vinet_set_option(b"up_blk", 0) == 0
    try:
        test_upsample(dt)
    finally:
        lib.vinet_set_option(b"up_blk", 1)


@pytest.mark.parametrize("dt", DTS)
def test_upsample(dt):
    B, T, H, W, Cc = 2, 3, 5, 7, 32
    xp, xmk = view_pair(B, T, H, W, Cc, dt, "ux", 1)
    yp, ymk = view_pair(B, T, 2 * H, 2 * W, Cc, dt, "uy", 2, t_total=T + 2, t_off=2)
    run_both("vinet_upsample2x", lambda s: [C.byref(xmk(s).ct()), C.byref(ymk(s).ct()), dt, _stream() if s == "gpu" else 0])
    _cmp(yp.get("gpu"), yp.get("cpu"), 1e-6 if dt == E.F32 else 1e-2, "upsample")
    dxp, dxmk = view_pair(B, T, H, W, Cc, dt, "udx", 3)
    run_both("vinet_upsample2x_bwd", lambda s: [C.byref(ymk(s).ct()), C.byref(dxmk(s).ct()), dt, 1, _stream() if s == "gpu" else 0])
    _cmp(dxp.get("gpu"), dxp.get("cpu"), 1e-5 if dt == E.F32 else 2e-2, "upsample bwd")
    # the backward of a ReLU in front of the upsample folded in: dx = [xf > 0] * upsample^T(dy), stored
    dmp, dmmk = view_pair(B, T, H, W, Cc, dt, "udm", 4, ld=48, c_off=8)
    run_both("vinet_upsample2x_bwd_relu", lambda s: [C.byref(ymk(s).ct()), C.byref(dmmk(s).ct()), C.byref(xmk(s).ct()), dt, _stream() if s == "gpu" else 0])
    _cmp(dmp.get("gpu"), dmp.get("cpu"), 1e-5 if dt == E.F32 else 2e-2, "upsample bwd + ReLU gate")
    gated = E.View(dmp.get("gpu"), 8, B, T, H, W, Cc, 48, T * H * W * 48, dt).torch5()
    assert bool((gated[xmk("cpu").torch5() <= 0] == 0).all()) and bool((gated != 0).any())


@pytest.mark.parametrize("which", [0, 1, 2])
@pytest.mark.parametrize("g64", [0, 1])
def test_losses(which, g64):
    B, H, W = 3, 40, 56
    s = Pair(synth.uniform("ls", (B, H, W), 1, 0.01, 0.99))
    g = synth.gt_map(B, H, W, 2)
    g = Pair(g.double() if g64 else g)
    saved = Pair(torch.zeros(B * 8, dtype=torch.float64))
    loss = Pair(torch.zeros(1))
    run_both("vinet_loss_fwd", lambda sd: [which, s.ptr(sd), g.ptr(sd), g64, B, H * W, saved.ptr(sd), loss.ptr(sd), _stream() if sd == "gpu" else 0])
    _cmp(loss.get("gpu"), loss.get("cpu"), 1e-6, "loss fwd")
    gs = Pair(torch.tensor([0.7]))
    ds = Pair(torch.zeros(B, H, W))
    savedg = saved.gpu.clone()

    def mk(sd):
        sv = saved.cpu if sd == "cpu" else savedg
        return [which, s.ptr(sd), g.ptr(sd), g64, B, H * W, sv.data_ptr(), gs.ptr(sd), -1.0, 0, ds.ptr(sd), _stream() if sd == "gpu" else 0]
    run_both("vinet_loss_bwd", mk)
    a, b = ds.get("gpu"), ds.get("cpu")
    assert float((a - b).abs().max()) <= 1e-6 * float(b.abs().max()) + 1e-12, "loss bwd %g vs scale %g" % (float((a - b).abs().max()), float(b.abs().max()))


@pytest.mark.parametrize("g64", [0, 1])
def test_nss_forward(g64):
    B, H, W = 3, 40, 56
    s = Pair(synth.uniform("ns", (B, H, W), 1, 0.01, 0.99))
    g = synth.gt_map(B, H, W, 2)
    g = (g > 0.5 * g.amax(dim=(1, 2), keepdim=True)).float()
    g = Pair(g.double() if g64 else g)
    saved = Pair(torch.zeros(B * 8, dtype=torch.float64))
    out = Pair(torch.zeros(1))
    run_both("vinet_loss_fwd", lambda sd: [3, s.ptr(sd), g.ptr(sd), g64, B, H * W, saved.ptr(sd), out.ptr(sd), _stream() if sd == "gpu" else 0])
    _cmp(out.get("gpu"), out.get("cpu"), 1e-6, "nss")
    lib = _lib()
    ds = torch.zeros(B * H * W, device=_dev())
    assert lib.vinet_loss_bwd(3, s.gpu.data_ptr(), g.gpu.data_ptr(), g64, B, H * W, saved.gpu.data_ptr(), None, 1.0, 0, ds.data_ptr(), _stream()) != 0


def test_adam_and_fill():
    n = 10007
    n4 = (n + 3) // 4 * 4
    p, g = Pair(_rand("ap", (n4,), 1)), Pair(_rand("ag", (n4,), 2, 0.01))
    m, v = Pair(torch.zeros(n4)), Pair(torch.zeros(n4))
    for step in (1, 2, 3):
        bc1, bc2 = 1 - 0.9 ** step, 1 - 0.999 ** step
        run_both("vinet_adam_step", lambda s: [p.ptr(s), g.ptr(s), m.ptr(s), v.ptr(s), n, 1e-4, 0.9, 0.999, 1e-8, bc1, bc2, 1.0, _stream() if s == "gpu" else 0])
    _cmp(p.get("gpu"), p.get("cpu"), 1e-6, "adam p")
    # against torch.optim.Adam
    tp = torch.nn.Parameter(_rand("ap", (n4,), 1)[:n].clone())
    opt = torch.optim.Adam([tp], lr=1e-4)
    for _ in range(3):
        tp.grad = _rand("ag", (n4,), 2, 0.01)[:n].clone()
        opt.step()
    _cmp(p.get("gpu")[:n], tp.detach(), 1e-6, "adam vs torch")
    f = Pair(torch.zeros(1000))
    run_both("vinet_fill_f32", lambda s: [f.ptr(s), 1000, 3.5, _stream() if s == "gpu" else 0])
    assert torch.equal(f.get("gpu"), f.get("cpu"))


@pytest.mark.parametrize("dt", DTS)
def test_bilinear(dt):
    B, Cc, I, J, O = 2, 64, 42, 3, 336
    x1 = Pair(_rand("b1", (B * I * Cc,), 1).to(E.TORCH_DT[dt]))
    x2 = Pair(_rand("b2", (B * J * Cc,), 2).to(E.TORCH_DT[dt]))
    w, bias = Pair(_rand("bw", (O * I * J,), 3, 0.1)), Pair(_rand("bb", (O,), 4))
    out = Pair(torch.zeros(B * O * Cc).to(E.TORCH_DT[dt]))
    run_both("vinet_bilinear_fwd", lambda s: [x1.ptr(s), x2.ptr(s), dt, w.ptr(s), bias.ptr(s), B, Cc, I, J, O, out.ptr(s), _stream() if s == "gpu" else 0])
    _cmp(out.get("gpu"), out.get("cpu"), 2e-5 if dt == E.F32 else 2e-2, "bilinear fwd")
    do = Pair(_rand("bdo", (B * O * Cc,), 5).to(E.TORCH_DT[dt]))
    d1 = Pair(torch.zeros(B * I * Cc).to(E.TORCH_DT[dt]))
    d2 = Pair(torch.zeros(B * J * Cc).to(E.TORCH_DT[dt]))
    dw, db = Pair(torch.zeros(O * I * J)), Pair(torch.zeros(O))
    run_both("vinet_bilinear_bwd", lambda s: [x1.ptr(s), x2.ptr(s), do.ptr(s), dt, w.ptr(s), B, Cc, I, J, O, d1.ptr(s), d2.ptr(s), dw.ptr(s), db.ptr(s), _stream() if s == "gpu" else 0])
    for a, nm in ((d1, "dx1"), (d2, "dx2"), (dw, "dw"), (db, "dbias")):
        _cmp(a.get("gpu"), a.get("cpu"), 5e-5 if dt == E.F32 else 3e-2, "bilinear bwd " + nm)


def test_pingpong_kernels_race_screen():
    """The ping-pong kernels order LDS-DMA writes against fragment reads by counted vmcnt + barriers only; a
    misplaced wait shows up as RARE wrong tiles.  Screen: large layer shapes (many workgroups per CU, every
    pipeline stage exercised), repeated launches, compared with the conv_dma / wgrad_dma result of the same
    descriptor (same math, different accumulation order)."""
    lib = _lib()
    dev = _dev()
    B, T, H, W, Cin, N, k = 4, 8, 28, 48, 256, 384, (1, 3, 3)
    x = (_rand("rsx", (B * T * H * W * Cin,), 1)).to(torch.bfloat16).to(dev)
    ntaps = 9
    w = (_rand("rsw", (ntaps * N * Cin,), 2, 1.0 / math.sqrt(Cin * ntaps))).to(torch.bfloat16).to(dev)
    taps = torch.tensor(_fwd_taps(k, (0, 1, 1)), dtype=torch.int32, device=dev)
    ys = [torch.empty(B * T * H * W * N, dtype=torch.bfloat16, device=dev) for _ in range(2)]

    def conv(y):
        d = L.CConvDesc()
        d.dtype = d.out_dtype = E.BF16
        d.mode = 0
        d.x = L.CTensor(x.data_ptr(), B, T, H, W, Cin, Cin, T * H * W * Cin)
        d.y = L.CTensor(y.data_ptr(), B, T, H, W, N, N, T * H * W * N)
        d.oT, d.oH, d.oW = T, H, W
        d.sT = d.sH = d.sW = 1
        d.omT = d.omH = d.omW = 1
        d.ntaps, d.taps, d.w, d.Kp = ntaps, taps.data_ptr(), w.data_ptr(), Cin
        d.pre = L.CAffine(None, None, 0)
        assert lib.vinet_conv3d(C.byref(d), _stream()) == 0, lib.vinet_last_error()

    try:
        lib.vinet_set_option(b"pp", 0)
        conv(ys[0])
        torch.cuda.synchronize()
        ref = ys[0].float()
        scale = float(ref.abs().max())
        for shape in (3, 4):
            lib.vinet_set_option(b"pp", shape)
            for it in range(12):
                ys[1].zero_()
                conv(ys[1])
                torch.cuda.synchronize()
                d = float((ys[1].float() - ref).abs().max())
                assert d <= 2e-2 * scale, "conv_pp shape %d, launch %d: max diff %g (scale %g)" % (shape, it, d, scale)
    finally:
        lib.vinet_set_option(b"pp", 1)

    # weight gradient: dy = the conv output above, x as is
    dws = [torch.zeros(ntaps * N * Cin, device=dev) for _ in range(2)]

    def wgrad(dw):
        d = L.CWgradDesc()
        d.dtype, d.mode = E.BF16, 0
        d.x = L.CTensor(x.data_ptr(), B, T, H, W, Cin, Cin, T * H * W * Cin)
        d.dy = L.CTensor(ys[0].data_ptr(), B, T, H, W, N, N, T * H * W * N)
        d.sT = d.sH = d.sW = 1
        d.ntaps, d.taps, d.dw, d.Kp = ntaps, taps.data_ptr(), dw.data_ptr(), Cin
        d.pre = L.CAffine(None, None, 0)
        assert lib.vinet_conv3d_wgrad(C.byref(d), _stream()) == 0, lib.vinet_last_error()

    try:
        lib.vinet_set_option(b"wgrad_pp", 0)
        wgrad(dws[0])
        torch.cuda.synchronize()
        ref = dws[0].clone()
        scale = float(ref.abs().max())
        for shape in (3, 4):
            lib.vinet_set_option(b"wgrad_pp", shape)
            for it in range(8):
                dws[1].zero_()
                wgrad(dws[1])
                torch.cuda.synchronize()
                d = float((dws[1] - ref).abs().max())
                assert d <= 5e-3 * scale, "wgrad_pp tile %d, launch %d: max diff %g (scale %g)" % (shape, it, d, scale)
    finally:
        lib.vinet_set_option(b"wgrad_pp", 1)


def test_pingpong_kernels_random_shapes():
    """a seeded slice of tools/fuzz_pp.py: random geometry forced through conv_pp / conv_wgrad_pp vs the ABI model"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_pp.py"), "16", "7"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


# ---- saliency-map post-processing: generate_result.py:95-104, train.py:251-253, utils.py:61-78 ----
# up- and down-scaling, the harness shape (224x384 -> 360x640), a full-HD image, maps smaller than the blur radius (repeated
# reflection), one-row / one-column maps, tile edges (sizes that are not multiples of 32 x 64), batches
POSTPROC_CASES = [
    ("harness", 2, 224, 384, 360, 640),
    ("fullhd", 1, 224, 384, 1080, 1920),
    ("down", 3, 224, 384, 100, 150),
    ("same", 2, 96, 192, 96, 192),
    ("tiny_up", 2, 7, 9, 20, 31),
    ("tiny_down", 1, 8, 8, 3, 5),
    ("row", 1, 1, 20, 1, 33),
    ("col", 2, 17, 1, 65, 1),
    ("tile_edges", 1, 40, 70, 33, 129),
]


@pytest.mark.parametrize("case", POSTPROC_CASES, ids=[c[0] for c in POSTPROC_CASES])
def test_resize_blur_normalize_u8(case):
    """the device maps must be the oracle's BIT FOR BIT: float32 blurred maps, min / max keys, uint8 images."""
    from oracle import postproc_cpu as P
    name, B, H, W, oH, oW = case
    lib = _lib()
    src = torch.sigmoid(_rand("pp" + name, (B, H, W), 3) * 2.0 - 1.0).contiguous()
    ref = P.resize_blur(src.numpy(), oH, oW)
    ref8 = P.normalize_u8(ref)
    s = src.to(_dev())
    out = torch.empty((B, oH, oW), dtype=torch.float32, device=_dev())
    mm = torch.empty((B, 2), dtype=torch.int32, device=_dev())
    assert lib.vinet_resize_blur(s.data_ptr(), B, H, W, out.data_ptr(), oH, oW, mm.data_ptr(), _stream()) == 0, lib.vinet_last_error()
    u8 = torch.empty((B, oH, oW), dtype=torch.uint8, device=_dev())
    assert lib.vinet_normalize_u8(out.data_ptr(), mm.data_ptr(), B, oH * oW, u8.data_ptr(), _stream()) == 0, lib.vinet_last_error()
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    assert np.abs(got - ref).max() <= 1e-6, "resize_blur %s: max abs diff %g" % (name, np.abs(got - ref).max())
    assert np.array_equal(got, ref), "resize_blur %s: %d of %d samples differ in the last bit" % (name, (got != ref).sum(), ref.size)
    assert np.array_equal(u8.cpu().numpy(), ref8), "normalize_u8 %s: %d bytes differ" % (name, (u8.cpu().numpy() != ref8).sum())
    # the keys alone (maps that did not come out of resize_blur), then the bytes from them
    mm2 = torch.empty((B, 2), dtype=torch.int32, device=_dev())
    assert lib.vinet_minmax(out.data_ptr(), B, oH * oW, mm2.data_ptr(), _stream()) == 0
    torch.cuda.synchronize()
    assert torch.equal(mm.cpu(), mm2.cpu())
    # no keys requested
    out2 = torch.empty_like(out)
    assert lib.vinet_resize_blur(s.data_ptr(), B, H, W, out2.data_ptr(), oH, oW, None, _stream()) == 0
    torch.cuda.synchronize()
    assert torch.equal(out, out2)


def test_normalize_u8_negative_and_constant_maps():
    """min / max keys order floats of either sign; a constant map divides by 1e-5 and lands on 0"""
    from oracle import postproc_cpu as P
    lib = _lib()
    x = torch.stack([_rand("ppneg", (50, 70), 1), torch.full((50, 70), 0.25), -torch.rand(50, 70, generator=torch.Generator().manual_seed(1)) - 1.0])
    ref8 = P.normalize_u8(x.numpy())
    xd = x.to(_dev()).contiguous()
    mm = torch.empty((3, 2), dtype=torch.int32, device=_dev())
    u8 = torch.empty((3, 50, 70), dtype=torch.uint8, device=_dev())
    assert lib.vinet_minmax(xd.data_ptr(), 3, 3500, mm.data_ptr(), _stream()) == 0
    assert lib.vinet_normalize_u8(xd.data_ptr(), mm.data_ptr(), 3, 3500, u8.data_ptr(), _stream()) == 0
    torch.cuda.synchronize()
    assert np.array_equal(u8.cpu().numpy(), ref8)
    assert lib.vinet_resize_blur(None, 1, 4, 4, u8.data_ptr(), 4, 4, None, _stream()) < 0       # bad arguments fail loudly


# ---- input pipeline: dataloader.py:243-250, 283-296; generate_result.py:77-88 --------------------------
PREPROC_CASES = [
    ("dhf1k", 3, 360, 640, 224, 384),
    ("hd", 1, 720, 1280, 224, 384),
    ("upscale", 2, 100, 150, 224, 384),
    ("same", 2, 224, 384, 224, 384),
    ("w_only", 1, 224, 500, 224, 384),
    ("tiny", 2, 5, 7, 20, 31),
    ("odd", 1, 37, 53, 20, 31),
]


@pytest.mark.parametrize("case", PREPROC_CASES, ids=[c[0] for c in PREPROC_CASES])
def test_frames_and_gt_preprocess(case):
    """device img_transform == the oracle's (which is pinned byte for byte against Pillow), bit for bit; the same for the
    ground-truth path"""
    from oracle import preproc_cpu as Q
    from vinet_amd import preprocess as PR
    name, N, H, W, oH, oW = case
    rng = np.random.default_rng(len(name) * 100 + H)
    u8 = rng.integers(0, 256, (N, H, W, 3), dtype=np.uint8)
    got = PR.frames_to_tensor(torch.from_numpy(u8).to(_dev()), (oH, oW)).cpu().numpy()
    ref = Q.frames_preprocess(u8, oH, oW)
    assert got.shape == ref.shape and np.array_equal(got, ref), "%s: %d of %d values differ" % (name, (got != ref).sum(), ref.size)
    g8 = rng.integers(0, 256, (N, H, W), dtype=np.uint8)
    g8[-1] = (g8[-1] > 200).astype(np.uint8)                  # a 0/1 map: not divided by 255
    for size in ((oH, oW), None):
        gg = PR.gt_to_tensor(torch.from_numpy(g8).to(_dev()), size).cpu().numpy()
        gr = Q.gt_preprocess(g8, *(size or (None, None)))
        assert np.array_equal(gg, gr), "%s gt %s: max abs diff %g" % (name, size, np.abs(gg - gr).max())


def test_audio_excerpt():
    """dataloader.py:89-122 on device: exact zeros outside the excerpt, float32 round-off inside (the window is a double
    cosine evaluated by two different libms)"""
    from oracle import preproc_cpu as Q
    from vinet_amd import preprocess as PR
    rng = np.random.default_rng(3)
    wav = (rng.standard_normal(200000) * 2 ** -8).astype(np.float32)
    wd = torch.from_numpy(wav).to(_dev())
    for (s, e) in [(0, 999), (5000, 5000 + 70559), (1234, 1234 + 47040), (199000, 260000), (10, 10), (300, 299)]:
        got = PR.audio_excerpt(wd, s, e).cpu().numpy()
        ref = Q.audio_excerpt(wav, s, e)
        assert np.array_equal(got == 0, ref == 0) or np.abs(got - ref).max() < 1e-9
        assert np.abs(got - ref).max() <= 1e-9 + 2e-7 * np.abs(ref).max()
    with pytest.raises(Exception):
        PR.audio_excerpt(wd, 0, 80000)                   # longer than the window: the reference's assignment fails too


def test_io_pipeline_kernels_match_committed_fixtures():
    """the device pre- / post-processing against tests/golden/io_frames.npz (real Pillow outputs) and io_maps.npz"""
    from tests import goldens as G
    from vinet_amd import preprocess as PR
    from vinet_amd import utils as U
    z, meta = G.load("io_frames")
    for name, (n, h, w, oh, ow) in meta["cases"].items():
        got = PR.frames_to_tensor(torch.from_numpy(z[name + "_in"]).to(_dev()), (oh, ow)).cpu().numpy()
        assert np.array_equal(got, z[name + "_out"]), name
    gt = torch.from_numpy(z["gt_in"]).to(_dev())
    assert np.array_equal(PR.gt_to_tensor(gt, meta["gt_train_size"]).cpu().numpy(), z["gt_train"])
    assert np.array_equal(PR.gt_to_tensor(gt).cpu().numpy(), z["gt_val"])
    m, _ = G.load("io_maps")
    src = torch.from_numpy(m["src"]).to(_dev())
    for name, size in {"up_45x80": (45, 80), "same": (28, 48), "down_9x13": (9, 13)}.items():
        assert np.array_equal(U.resize_blur(src, size).cpu().numpy(), m[name + "_blur"]), name
        assert np.array_equal(U.postprocess(src, size).cpu().numpy(), m[name + "_u8"]), name


@pytest.mark.parametrize("dt", DTS)
def test_unfold1d(dt):
    """SoundNet conv1's im2col along the waveform: windows of k samples every `stride`, zero padding at both ends, channel 0
    of a channel-padded input"""
    B, L, Cpad, k, stride, pad = 3, 501, 8, 64, 2, 32
    Lo = (L + 2 * pad - k) // stride + 1
    xp, xmk = view_pair(B, L, 1, 1, Cpad, dt, "unf_x", 1)
    yp, ymk = view_pair(B, Lo, 1, 1, k, dt, "unf_y", 2, ld=72, c_off=8)       # a channel slice of a wider buffer

    def mk(side):
        return [C.byref(xmk(side).ct()), C.byref(ymk(side).ct()), dt, stride, pad, _stream() if side == "gpu" else 0]

    run_both("vinet_unfold1d", mk)
    assert torch.equal(yp.get("gpu"), yp.get("cpu"))
    lib = _lib()
    bad = ymk("gpu").ct()
    bad.T += 1
    assert lib.vinet_unfold1d(C.byref(xmk("gpu").ct()), C.byref(bad), dt, stride, pad, _stream()) < 0


@pytest.mark.parametrize("cin", [8, 32, 64])
def test_conv3d_wgrad_skinny(cin):
    """pointwise weight gradient with a channel-padded 8-wide dy (the 32 -> 1 head): the reduction kernel, on views with
    a channel offset and a voxel count that is not a multiple of anything"""
    lib = _lib()
    assert lib.vinet_set_option(b"wgrad_skinny", 2) == 0
    try:
        case = ("skinny%d" % cin, (2, 3, 7, 11), cin, 8, (1, 1, 1), (1, 1, 1), (0, 0, 0), False, dict(x_ld=cin + 16, x_coff=8, dy_ld=24, dy_coff=16))
        d0 = _run_wgrad_case(case, E.BF16)
        buf = C.create_string_buffer(128)
        assert lib.vinet_conv3d_wgrad_kernel_name(C.byref(d0), buf, 128) == 0 and buf.value == b"wgrad_skinny_kernel"
    finally:
        lib.vinet_set_option(b"wgrad_skinny", 1)


# ---- exact arithmetic: every conv / weight-gradient kernel variant bit-identical with the ABI model ------------------
def _exact_targets():
    t = []
    for c in CONV_CASES:
        if c[7].get("act") != 2:                       # (a sigmoid is not exact)
            t.append(("conv3d-" + c[0], test_conv3d, dict(case=c, dt=E.BF16)))
    for shape in (3, 4):
        for c in PP_CASES:
            if c[7].get("act") != 2:
                t.append(("pingpong%d-%s" % (shape, c[0]), test_conv3d_pingpong, dict(case=c, shape=shape)))
    for c in N192_CASES:
        t.append(("n192-" + c[0], test_conv3d_n192_tile, dict(case=c)))
    for c in CONV_TS_CASES:
        t.append(("tstream-" + c[0], test_conv3d_tstream, dict(case=c)))
    for c in HT_CASES:
        if not c[7].get("out_f32"):
            t.append(("halo16-" + c[0], test_conv3d_halo_tile, dict(case=c)))
    for c in PW_CASES:
        t.append(("pw-" + c[0], test_conv3d_pointwise_stream, dict(case=c)))
    for c in EPI_ROWS_CASES + EPI_ROWS_HT:
        t.append(("epi-rows-" + c[0], test_conv3d_whole_row_epilogue, dict(case=c)))
    for ksp in [(7, 2, 3), (3, 2, 1), (5, 3, 2)]:
        for acc in (0, 1):
            t.append(("tsd-k%ds%dp%d-acc%d" % (ksp + (acc,)), test_conv3d_tstream_dgrad_fused, dict(ksp=ksp, acc=acc)))
    for r in (0, 1):
        t.append(("ts-phase%d" % r, test_conv3d_tstream_dgrad_phase, dict(r=r)))
    t.append(("stem-mode", test_conv3d_stem_mode, dict(dt=E.BF16)))
    for hw in [(18, 22), (17, 23), (20, 128)]:
        t.append(("stem-folded-%dx%d" % hw, test_stem_folded, dict(dt=E.BF16, hw=hw)))
    for c in WGRAD_CASES:
        t.append(("wgrad-" + c[0], test_conv3d_wgrad, dict(case=c, dt=E.BF16)))
    for shape in (3, 4):
        for c in WGRAD_PP_CASES:
            t.append(("wgrad-pp%d-%s" % (shape, c[0]), test_conv3d_wgrad_pingpong, dict(case=c, shape=shape)))
    for c in WGRAD_TS_CASES:
        t.append(("wgrad-ts-" + c[0], test_conv3d_wgrad_tstream, dict(case=c)))
    for c in WGRAD_RS_CASES:
        t.append(("wgrad-rs-" + c[0], test_conv3d_wgrad_rowstream, dict(case=c)))
    for c in WGRAD_TF_CASES:
        t.append(("wgrad-tf-" + c[0], test_conv3d_wgrad_tframes, dict(case=c)))
    t.append(("wgrad-stem", test_conv3d_wgrad_stem, dict(dt=E.BF16)))
    for cin in (32, 64):
        t.append(("wgrad-skinny-%d" % cin, test_conv3d_wgrad_skinny, dict(cin=cin)))
    return t


_EXACT_TARGETS = _exact_targets()


@pytest.mark.parametrize("target", _EXACT_TARGETS, ids=[t[0] for t in _EXACT_TARGETS])
def test_exact_arithmetic_bit_identity(target):
    """small-integer bf16 inputs through every conv and weight-gradient kernel variant on the existing case tables:
    the result must equal the ABI model bit for bit (see exact_mode above)"""
    _, fn, kw = target
    with exact_mode():
        fn(**kw)


# ---- the packed-fp32 erratum (round 5) ------------------------------------------------------------------------------------------------
PROBE_SO = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "ubench", "libvmem_return_probe.so")


def _probe_lib():
    if not os.path.exists(PROBE_SO):
        pytest.skip("tools/ubench/libvmem_return_probe.so not built (__graft_entry__.build_probe)")
    pl = C.CDLL(PROBE_SO)
    pl.corun_mfma_launch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    pl.pk_variant_launch.argtypes = [C.c_int, C.c_int, C.c_uint32, C.c_void_p, C.c_void_p]
    return pl


@pytest.mark.parametrize("small", [0, 1])
def test_channel_reductions_are_exact_beside_foreign_mfma_waves(small):
    """Round 5's run-to-run gradient mismatch, as a regression test: `vinet_channel_sum` / `vinet_channel_stats` /
    `vinet_bn_bwd_reduce` on SoundNet's 6-voxel x 1024-channel tail, 1500 launches each, while a second stream keeps register-only
    MFMA waves of another kernel resident on every SIMD.  Before the fix (csrc/common.h VN_NO_PK_F32: the generic reduction contained
    `v_pk_add_f32 ... op_sel:[0,1]`, which reads zero in lanes 48..63 under exactly this condition) HALF of the launches returned a sum
    with one voxel's term missing; identical launches must return identical bits, and the sums must be the exact ones.  Both
    dispatches: the thread-per-channel kernel for tiny tensors (`reduce_small` = 1, the default) and the generic 256-thread kernels."""
    lib, pl, dev = _lib(), _probe_lib(), _dev()
    L.set_option("reduce_small", small)
    try:
        nv, Cc, iters = 6, 1024, 1500
        g = torch.Generator(device=dev).manual_seed(3)
        dy = (torch.randn(nv, Cc, generator=g, device=dev) * 1e-3).bfloat16()
        z = torch.randn(nv, Cc, generator=g, device=dev).bfloat16()
        mean, invstd = torch.randn(Cc, generator=g, device=dev) * 0.1, torch.rand(Cc, generator=g, device=dev) + 0.5
        sc, sh = torch.rand(Cc, generator=g, device=dev) + 0.5, torch.randn(Cc, generator=g, device=dev) * 0.3
        dyv = E.View(dy.view(-1), 0, 2, nv // 2, 1, 1, Cc, Cc, (nv // 2) * Cc, E.BF16)
        zv = E.View(z.view(-1), 0, 2, nv // 2, 1, 1, Cc, Cc, (nv // 2) * Cc, E.BF16)
        rows = lib.vinet_stats_rows(C.byref(dyv.ct()))
        n = rows * 2 * Cc
        slots = torch.empty(3, iters, n, device=dev)
        out = torch.empty(Cc, device=dev)
        cobuf = torch.zeros(1024, device=dev)
        side = torch.cuda.Stream()
        fwd = L.CAffine(sc.data_ptr(), sh.data_ptr(), 1)
        st = _stream()
        for i in range(iters):
            if i % 8 == 0:
                assert pl.corun_mfma_launch(cobuf.data_ptr(), 100000, 416, side.cuda_stream) == 0
            assert lib.vinet_channel_sum(C.byref(dyv.ct()), E.BF16, slots[0, i].data_ptr(), Cc, out.data_ptr(), 0, st) == 0
            assert lib.vinet_channel_stats(C.byref(dyv.ct()), E.BF16, slots[1, i].data_ptr(), st) == 0
            assert lib.vinet_bn_bwd_reduce(C.byref(dyv.ct()), C.byref(zv.ct()), E.BF16, fwd, mean.data_ptr(), invstd.data_ptr(),
                                           slots[2, i].data_ptr(), st) == 0
        torch.cuda.synchronize()
        for k, what in enumerate(("channel_sum", "channel_stats", "bn_bwd_reduce")):
            differ = int((slots[k] != slots[k, 0]).any(dim=1).sum())
            assert differ == 0, "%s (reduce_small=%d): %d of %d launches differ from the first one beside MFMA waves" % (what, small, differ, iters)
        ref = dy.float().double().sum(0)
        got = slots[0, 0].view(rows, 2, Cc)[:, 0, :].double().sum(0)
        assert float((got - ref).abs().max()) < 1e-9
        gate = (z.float() * sc + sh) > 0
        ref2 = (dy.float() * gate).double().sum(0)
        got2 = slots[2, 0].view(rows, 2, Cc)[:, 0, :].double().sum(0)
        assert float((got2 - ref2).abs().max()) < 1e-9
    finally:
        L.set_option("reduce_small", 1)


def test_packed_fp32_erratum_is_still_there():
    """Documentation as a test (never fails on a healthy result): the affected instruction form on bare registers beside MFMA waves.
    Records the counts in the parity report so a board / firmware on which the erratum is gone shows up as zeros."""
    pl, dev = _probe_lib(), _dev()
    cobuf = torch.zeros(1024, device=dev)
    side = torch.cuda.Stream()
    res = {}
    for which, name in ((0, "v_pk_add_f32"), (1, "v_pk_add_f32 op_sel:[0,1] op_sel_hi:[1,0]"), (4, "v_pk_add_f32 op_sel:[1,0] op_sel_hi:[0,1]")):
        pout = torch.zeros(16, dtype=torch.int32, device=dev)
        for i in range(600):
            if i % 8 == 0:
                assert pl.corun_mfma_launch(cobuf.data_ptr(), 100000, 416, side.cuda_stream) == 0
            assert pl.pk_variant_launch(which, 2000, i, pout.data_ptr(), _stream()) == 0
        torch.cuda.synchronize()
        o = pout.tolist()
        res[name] = {"wrong": int(o[1]) & 0xffffffff, "of": int(o[0]) * 256 * 2000 * 2}
    try:
        from tests.test_gpu_model import _note
        _note("packed_fp32_erratum_beside_mfma", res)
    except Exception:
        pass
    assert res["v_pk_add_f32"]["wrong"] == 0 and res["v_pk_add_f32 op_sel:[1,0] op_sel_hi:[0,1]"]["wrong"] == 0, res
