"""CPU model of the libvinet_hip.so C ABI  --  TEST INFRASTRUCTURE ONLY.

Implements every entry point of include/vinet_hip.h in plain numpy/torch-CPU,
straight from the documented contract (descriptors with raw host pointers).
Installed with ``vinet_amd._lib._install_test_double(AbiEmulator())`` by the
CPU tests so the *host logic* of vinet_amd (tap tables, dgrad phase
decomposition, BN bookkeeping, concat plumbing, the backward tape, the
optimizer) can be checked against the oracle without a GPU.  The product never
imports this file; on a GPU box the real library is used and each kernel is
checked against the same contract.
"""
import ctypes as C

import numpy as np
import torch

from vinet_amd import _lib as L

F32, BF16 = 0, 1


def _deref(a):
    return a._obj if hasattr(a, "_obj") else a


def _raw(ptr, n, dt):
    if dt == F32:
        return np.ctypeslib.as_array((C.c_float * n).from_address(ptr))
    return np.ctypeslib.as_array((C.c_uint16 * n).from_address(ptr))


def _f32(ptr, n):
    return np.ctypeslib.as_array((C.c_float * n).from_address(ptr)) if ptr else None


def _bf2f(u):
    return (u.astype(np.uint32) << 16).view(np.float32)


def _f2bf(f):
    u = np.ascontiguousarray(f, dtype=np.float32).view(np.uint32)
    r = (u + 0x7FFF + ((u >> 16) & 1)) >> 16
    nan = (u & 0x7FFFFFFF) > 0x7F800000
    r = np.where(nan, (u >> 16) | 0x40, r)
    return r.astype(np.uint16)


def _span(t):
    return (t.B - 1) * t.sB + ((t.T - 1) * t.H * t.W + (t.H - 1) * t.W + (t.W - 1)) * t.ld + t.C


def _strided(t, dt):
    raw = _raw(t.ptr, _span(t), dt)
    es = raw.itemsize
    return np.lib.stride_tricks.as_strided(raw, (t.B, t.T, t.H, t.W, t.C),
                                           (t.sB * es, t.H * t.W * t.ld * es, t.W * t.ld * es, t.ld * es, es))


def rd(t, dt):
    v = _strided(t, dt)
    return v.astype(np.float32) if dt == F32 else _bf2f(np.ascontiguousarray(v))


def wr(t, dt, val, accumulate=False):
    v = _strided(t, dt)
    if accumulate:
        val = val + (v if dt == F32 else _bf2f(np.ascontiguousarray(v)))
    if dt == F32:
        v[...] = val.astype(np.float32)
    else:
        v[...] = _f2bf(val).reshape(v.shape)


def affine(x, a, C_):
    if a.scale:
        x = x * _f32(a.scale, C_) + _f32(a.shift, C_)
        x = x.astype(np.float32)
    if a.relu:
        x = np.maximum(x, 0)
    return x


def _gather(x, idx, axis, size):
    """x indexed along `axis` with zero fill where idx is out of [0,size)."""
    ok = (idx >= 0) & (idx < size)
    g = np.take(x, np.clip(idx, 0, size - 1), axis=axis)
    shape = [1] * x.ndim
    shape[axis] = -1
    return g * ok.reshape(shape)


def _tile_m(dtype, mode, M, N, kchunks=0):
    if mode == 1:
        return 256 if dtype == BF16 else 128
    nts = [8, 6, 4, 3, 2, 1]
    best = min((N + nt * 16 - 1) // (nt * 16) * nt * 16 for nt in nts)
    nt = 1
    for c in nts:
        pad = (N + c * 16 - 1) // (c * 16) * c * 16
        if pad * 4 <= best * 5:
            nt = c
            break
    if dtype == F32:
        return 128
    if nt == 4 and 0 < kchunks <= 64 and M >= (1 << 20):
        return 128
    if nt == 6 and ((M + 255) // 256) * ((N + 95) // 96) < 64:
        nt = 8
    if nt == 6 and N % 192 == 0 and ((M + 127) // 128) * (N // 192) >= 512:
        return 128
    if nt == 8 and 0 < kchunks <= 64 and ((M + 127) // 128) * ((N + 127) // 128) >= 1024:
        return 128
    bm = 256
    if nt in (8, 4):
        bn = nt * 16
        tn = (N + bn - 1) // bn
        if ((M + 255) // 256) * tn < 512:
            bm = 128
            if ((M + 127) // 128) * tn < 512:
                bm = 64
    return bm


def _plain_f32(fn):
    """descriptor entry points: VINET_F32S (fp32 tensors, split-bf16 arithmetic) is modelled as exact fp32"""
    import functools

    @functools.wraps(fn)
    def wrapper(self, d, *a, **k):
        dd = d._obj if hasattr(d, "_obj") else d
        if getattr(dd, "dtype", None) == L.F32S:
            dd.dtype = F32
        return fn(self, d, *a, **k)
    return wrapper


class AbiEmulator:
    def __init__(self):
        self.err = b""
        self.calls = []

    # -- misc --------------------------------------------------------------
    def vinet_abi_version(self):
        return L.ABI_VERSION

    def vinet_last_error(self):
        return self.err

    def vinet_set_option(self, name, value):
        return 0

    def vinet_split_bf16(self, src, pre, hi, lo, stream):
        src, hi, lo = (t._obj if hasattr(t, "_obj") else t for t in (src, hi, lo))
        v = affine(rd(src, F32), pre, src.C)
        h = _bf2f(_f2bf(v.reshape(-1))).reshape(v.shape)
        wr(hi, BF16, h)
        wr(lo, BF16, v - h)
        return 0

    def vinet_debug_spin(self, cycles, stream):
        return 0

    def vinet_fill_f32(self, p, n, value, stream):
        _f32(p, n)[:] = value
        return 0

    # -- conv ----------------------------------------------------------------
    @_plain_f32
    def vinet_conv3d_splitk_bytes(self, d):
        return 0          # the model never splits; results are identical by construction

    @_plain_f32
    def vinet_conv3d_tile_m(self, d):
        d = _deref(d)
        M, N = d.x.B * d.oT * d.oH * d.oW, d.y.C
        # conv_ts.hip::vinet_conv_use_ts -- the frame-streaming kernel of temporal 64 -> 64 convs: one stats row per 64 positions
        HW = d.oH * d.oW
        if (d.tline == 1 and d.dtype == BF16 and d.out_dtype == BF16 and d.mode == 0 and d.x.C == 64 and N == 64 and d.Kp == 64 and
                2 <= d.ntaps <= 7 and d.sT in (1, 2) and d.ntaps >= d.sT and (d.sH, d.sW, d.omH, d.omW, d.ooH, d.ooW) == (1, 1, 1, 1, 0, 0) and
                (d.x.H, d.x.W, d.y.H, d.y.W) == (d.oH, d.oW, d.oH, d.oW) and HW % 64 == 0 and 0 <= d.tpad < d.ntaps and
                d.oT >= 4 and d.x.B * (HW // 64) * max(1, min(-(-512 // (d.x.B * (HW // 64))), d.oT // 4)) >= 384 and
                not (d.pre.relu and not d.pre.scale)):       # (frame segments: vinet_conv_ts_segments)
            return 64
        # conv_hs.hip::vinet_conv_use_hs -- the row-streaming strip kernel of the folded RGB stem
        if (d.tline == 2 and d.dtype == BF16 and d.out_dtype == BF16 and d.mode == 0 and d.x.C == 32 and d.x.ld == 8 and d.Kp == 32 and
                d.ntaps == 7 and (d.sT, d.sH, d.sW) == (1, 2, 1) and N == 64 and d.oW % 64 == 0 and not d.pre.scale and not d.pre.relu and
                not d.accumulate and d.oH >= 8 and
                d.x.B * d.oT * (d.oW // 64) * max(1, min(-(-768 // (d.x.B * d.oT * (d.oW // 64))), d.oH // 7)) >= 384):   # (row segments)
            return 64
        # conv_api.hip::use_pp -- the 256x256x64 kernel takes large plain bf16 convs
        if d.dtype == BF16 and d.mode == 0 and not d.pre.scale and not d.pre.relu and d.ntaps <= 64:
            nkt = d.ntaps * ((d.Kp + 63) // 64)
            bn = 192 if ((N + 191) // 192 * 192) * 20 <= ((N + 255) // 256 * 256) * 17 else 256
            tiles = ((M + 255) // 256) * ((N + bn - 1) // bn)
            if N >= 160 and nkt >= (8 if d.ntaps == 1 else 16) and tiles >= 128:
                return 256
        return _tile_m(d.dtype, d.mode, M, N, d.ntaps * (d.Kp // 32))

    @_plain_f32
    def vinet_conv3d_kernel_name(self, d, buf, n):
        return 0

    @_plain_f32
    def vinet_conv3d_wgrad_kernel_name(self, d, buf, n):
        return 0

    @_plain_f32
    def vinet_conv3d_wgrad_fuses_bn_bwd(self, d):
        d = _deref(d)
        # (as the library: only the folded-stem strip kernel; no size threshold here so that the CPU tests cover it)
        return 1 if (d.bnb_z and d.tline == 2 and d.dtype == BF16 and not d.pre.scale) else 0

    def _taps(self, d):
        return np.ctypeslib.as_array((C.c_int32 * (4 * d.ntaps)).from_address(d.taps)).reshape(-1, 4)

    def _gathered(self, x, d, dt_, dh_, dw_, oT, oH, oW):
        """x[b, to*sT+dt, ho*sH+dh, wo*sW+dw, :] with zero fill -> [B,oT,oH,oW,C]"""
        g = _gather(x, np.arange(oT) * d.sT + dt_, 1, x.shape[1])
        g = _gather(g, np.arange(oH) * d.sH + dh_, 2, x.shape[2])
        g = _gather(g, np.arange(oW) * d.sW + dw_, 3, x.shape[3])
        return g

    def _bnb_epi_ok(self, d):
        """conv_api.hip::bnb_epi_ok -- the shared conv epilogue forms the BatchNorm-backward sums for bf16 data gradients that cover y
        densely in one launch, with whole 8-channel groups, plain inputs (the library also excludes the kernels with their own
        epilogues; the model has no kernel selection, so it folds wherever the CONTRACT allows)"""
        if not (d.bnb_z and d.bnb_mean and d.bnb_invstd) or (d.bnb_fwd.relu and not (d.bnb_fwd.scale and d.bnb_fwd.shift)):
            return False
        if d.dtype != BF16 or d.out_dtype != BF16 or d.mode != 0 or d.stats or d.act or d.out_scale or d.out_shift:
            return False
        if d.n_valid > 0 and d.n_valid != d.y.C:
            return False
        if (d.omT, d.omH, d.omW, d.ooT, d.ooH, d.ooW) != (1, 1, 1, 0, 0, 0) or (d.y.T, d.y.H, d.y.W) != (d.oT, d.oH, d.oW):
            return False
        if d.y.C % 8 or d.y.ld % 8 or d.y.sB % 8 or d.bnb_ld % 8 or d.bnb_sB % 8 or d.bnb_ld < d.y.C or d.pre.scale or d.pre.relu:
            return False
        return True

    def vinet_conv3d_bn_bwd_stats_rows(self, d):
        d = _deref(d)
        if d.tline == 3:
            return 0      # (the fused temporal data gradient of the stem: not modelled with statistics)
        return self.vinet_conv3d_stats_rows(d) if self._bnb_epi_ok(d) else 0

    @_plain_f32
    def vinet_conv3d_fuses_dgrad_phases(self, d):
        d = _deref(d)
        return 1 if (d.tline == 3 and d.dtype == BF16 and d.x.C == 64 and d.y.C == 64 and d.sT >= 2) else 0

    def _conv3d_tsd(self, d):
        """tline == 3: dx[ti] (+)= sum_{kt: (ti + p - kt) % s == 0} dy[(ti + p - kt) / s] @ wt[kt]^T"""
        assert self.vinet_conv3d_fuses_dgrad_phases(d)
        dy = rd(d.x, d.dtype)
        k, s, p = d.ntaps, d.sT, d.tpad
        Cc, N = d.y.C, d.x.C
        wraw = _raw(d.w, k * Cc * d.Kp, d.dtype)
        w = (wraw.astype(np.float32) if d.dtype == F32 else _bf2f(wraw)).reshape(k, Cc, d.Kp)[:, :, :N]
        B, To = dy.shape[0], dy.shape[1]
        out = np.zeros((B, d.y.T, d.y.H, d.y.W, Cc), np.float32)
        for ti in range(d.y.T):
            for kt in range(k):
                num = ti + p - kt
                if num % s == 0 and 0 <= num // s < To:
                    out[:, ti] += (torch.from_numpy(np.ascontiguousarray(dy[:, num // s]).reshape(-1, N)) @
                                   torch.from_numpy(w[kt].T.copy())).numpy().reshape(B, d.y.H, d.y.W, Cc)
        wr(d.y, d.out_dtype, out, bool(d.accumulate))
        return 0

    @_plain_f32
    def vinet_conv3d_applies_pre_once(self, d):
        return 0

    @_plain_f32
    def vinet_conv3d_stats_rows(self, d):
        d = _deref(d)
        M = d.x.B * d.oT * d.oH * d.oW
        bm = self.vinet_conv3d_tile_m(d)
        return (M + bm - 1) // bm

    @_plain_f32
    def vinet_conv3d(self, d, stream):
        d = _deref(d)
        self.calls.append("conv3d")
        if d.tline == 3:
            return self._conv3d_tsd(d)
        x = affine(rd(d.x, d.dtype), d.pre, d.x.C)
        N = d.y.C
        Nw = d.n_valid if d.n_valid > 0 else N
        Kp = d.Kp
        taps = self._taps(d)
        nsl = int(taps[:, 3].max()) + 1
        wraw = _raw(d.w, nsl * Nw * Kp, d.dtype)
        w = (wraw.astype(np.float32) if d.dtype == F32 else _bf2f(wraw)).reshape(nsl, Nw, Kp)
        B, oT, oH, oW = d.x.B, d.oT, d.oH, d.oW
        acc = np.zeros((B, oT, oH, oW, N), np.float32)
        for (dt_, dh_, dw_, sl) in taps:
            if d.mode == 0:
                g = self._gathered(x, d, dt_, dh_, dw_, oT, oH, oW)
                acc[..., :Nw] += (torch.from_numpy(np.ascontiguousarray(g).reshape(-1, d.x.C)) @
                                  torch.from_numpy(w[sl, :, :d.x.C].T.copy())).numpy().reshape(B, oT, oH, oW, Nw)
            else:
                for p in range(8):
                    g = self._gathered(x, d, dt_, dh_, dw_ + p, oT, oH, oW)
                    acc[..., :Nw] += (torch.from_numpy(np.ascontiguousarray(g).reshape(-1, 4)) @
                                      torch.from_numpy(w[sl, :, p * 4:p * 4 + 4].T.copy())).numpy().reshape(B, oT, oH, oW, Nw)
        if d.out_scale:
            acc[..., :Nw] *= _f32(d.out_scale, Nw)
        if d.out_shift:
            acc[..., :Nw] += _f32(d.out_shift, Nw)
        if d.stats:
            M = B * oT * oH * oW
            rows = (M + self.vinet_conv3d_tile_m(d) - 1) // self.vinet_conv3d_tile_m(d)
            st = _f32(d.stats, rows * 2 * N).reshape(rows, 2, N)
            st[...] = 0
            flat = acc.reshape(-1, N).astype(np.float64)
            st[0, 0] = flat.sum(0)
            st[0, 1] = (flat * flat).sum(0)
        if d.act == 1:
            acc = np.maximum(acc, 0)
        elif d.act == 2:
            acc = 1.0 / (1.0 + np.exp(-acc))
        y = _strided(d.y, d.out_dtype)
        sl = (slice(None), slice(d.ooT, d.ooT + (oT - 1) * d.omT + 1, d.omT), slice(d.ooH, d.ooH + (oH - 1) * d.omH + 1, d.omH),
              slice(d.ooW, d.ooW + (oW - 1) * d.omW + 1, d.omW), slice(None))
        if d.accumulate:
            old = y[sl]
            acc = acc + (old if d.out_dtype == F32 else _bf2f(np.ascontiguousarray(old)))
        y[sl] = acc.astype(np.float32) if d.out_dtype == F32 else _f2bf(acc).reshape(acc.shape)
        if d.bnb_partials:
            # the partial sums of vinet_bn_bwd_reduce(y, bnb_z) on the values just stored (rounded, accumulated)
            assert self._bnb_epi_ok(d), "bnb_partials set for a problem whose rows query returns 0"
            zt = L.CTensor(d.bnb_z, d.y.B, d.y.T, d.y.H, d.y.W, d.y.C, d.bnb_ld, d.bnb_sB)
            g, xhat = self._bn_bwd_terms(d.y, zt, d.out_dtype, d.bnb_fwd, d.bnb_mean, d.bnb_invstd)
            rows = self.vinet_conv3d_stats_rows(d)
            P = _f32(d.bnb_partials, rows * 2 * N).reshape(rows, 2, N)
            P[...] = 0
            P[0, 0] = g.reshape(-1, N).astype(np.float64).sum(0)
            P[0, 1] = (g * xhat).reshape(-1, N).astype(np.float64).sum(0)
            self.calls.append("conv3d+bnb")
        return 0

    @_plain_f32
    def vinet_conv3d_wgrad(self, d, stream):
        d = _deref(d)
        self.calls.append("wgrad")
        x = affine(rd(d.x, d.dtype), d.pre, d.x.C)
        dy = rd(d.dy, d.dtype)
        if d.bnb_z:     # fused BN backward: dy is the gradient behind the BatchNorm, bnb_z the raw conv output
            assert self.vinet_conv3d_wgrad_fuses_bn_bwd(d)
            zt = L.CTensor(d.bnb_z, d.dy.B, d.dy.T, d.dy.H, d.dy.W, d.dy.C, d.bnb_ld, d.bnb_sB)
            g, xhat = self._bn_bwd_terms(d.dy, zt, d.dtype, d.bnb_fwd, d.bnb_mean, d.bnb_invstd)
            Cc = d.dy.C
            dy = _f32(d.bnb_fwd.scale, Cc) * (g - _f32(d.bnb_c1, Cc) - xhat * _f32(d.bnb_c2, Cc))
            if d.dtype != F32:
                dy = _bf2f(_f2bf(dy.reshape(-1))).reshape(dy.shape)
        B, oT, oH, oW, N = dy.shape
        taps = self._taps(d)
        nsl = int(taps[:, 3].max()) + 1
        dw = _f32(d.dw, nsl * N * d.Kp).reshape(nsl, N, d.Kp)
        dyt = torch.from_numpy(np.ascontiguousarray(dy).reshape(-1, N).T.copy())
        for (dt_, dh_, dw_, sl) in taps:
            if d.mode == 0:
                g = self._gathered(x, d, dt_, dh_, dw_, oT, oH, oW)
                dw[sl, :, :d.x.C] += (dyt @ torch.from_numpy(np.ascontiguousarray(g).reshape(-1, d.x.C))).numpy()
            else:
                for p in range(8):
                    g = self._gathered(x, d, dt_, dh_, dw_ + p, oT, oH, oW)
                    dw[sl, :, p * 4:p * 4 + 4] += (dyt @ torch.from_numpy(np.ascontiguousarray(g).reshape(-1, 4))).numpy()
        return 0

    def vinet_pack_weights(self, w, N, Cin, ntaps, transpose, stem, dtype, out, stream):
        dtype = F32 if dtype == L.F32S else dtype      # (the emulator models the split-bf16 form as exact fp32, plain fp32 packs)
        W = _f32(w, N * Cin * ntaps).reshape(N, Cin, ntaps)
        if stem:
            o = np.zeros((7, N, 32), np.float32)
            for kh in range(7):
                for kw in range(7):
                    o[kh, :, kw * 4:kw * 4 + Cin] = W[:, :, kh * 7 + kw]
        elif not transpose:
            Kp = (Cin + 31) // 32 * 32
            o = np.zeros((ntaps, N, Kp), np.float32)
            o[:, :, :Cin] = W.transpose(2, 0, 1)
        else:
            Kp = (N + 31) // 32 * 32
            o = np.zeros((ntaps, Cin, Kp), np.float32)
            o[:, :, :N] = W.transpose(2, 1, 0)
        raw = _raw(out, o.size, dtype)
        raw[:] = o.reshape(-1) if dtype == F32 else _f2bf(o.reshape(-1))
        return 0

    def vinet_pack_weights_multi(self, table, njobs, total, dtype, stream):
        dtype = F32 if dtype == L.F32S else dtype
        tab = np.ctypeslib.as_array((C.c_int64 * (8 * (njobs + 1))).from_address(table)).reshape(njobs + 1, 8)
        assert int(tab[njobs, 6]) == total
        for j in range(njobs):
            w, out, N, Cin, ntaps, flags = (int(v) for v in tab[j, :6])
            ld, col = int(tab[j, 7]) & 0xffffffff, int(tab[j, 7]) >> 32
            if ld:
                assert flags == 1
                W = _f32(w, N * Cin * ntaps).reshape(N, Cin, ntaps)
                raw = _raw(out, ntaps * Cin * ld, dtype).reshape(ntaps, Cin, ld)
                o = W.transpose(2, 1, 0)
                raw[:, :, col:col + N] = o if dtype == F32 else _f2bf(o.reshape(-1)).reshape(o.shape)
                continue
            self.vinet_pack_weights(w, N, Cin, ntaps, flags & 1, (flags >> 1) & 1, dtype, out, stream)
        return 0

    def vinet_unpack_wgrad(self, dw, N, Cin, ntaps, stem, accumulate, grad, stream):
        g = _f32(grad, N * Cin * ntaps).reshape(N, Cin, ntaps)
        if stem:
            D = _f32(dw, 7 * N * 32).reshape(7, N, 32)
            v = np.zeros((N, Cin, ntaps), np.float32)
            for kh in range(7):
                for kw in range(7):
                    v[:, :, kh * 7 + kw] = D[kh, :, kw * 4:kw * 4 + Cin]
        else:
            Kp = (Cin + 31) // 32 * 32
            D = _f32(dw, ntaps * N * Kp).reshape(ntaps, N, Kp)   # rows beyond N (padded heads) are never read
            v = D[:, :N, :Cin].transpose(1, 2, 0)
        g[...] = g + v if (accumulate & 1) else v
        if accumulate & 2:      # hand the workspace back zeroed (rows [0, N) of every slice)
            D[:, :N, :] = 0
        return 0

    def vinet_unpack_wgrad_multi(self, table, njobs, total, flags, stream):
        tab = np.ctypeslib.as_array((C.c_int64 * (8 * (njobs + 1))).from_address(table)).reshape(njobs + 1, 8)
        assert int(tab[njobs, 6]) == total
        for j in range(njobs):
            dw, grad, N, Cin, ntaps, stem = (int(v) for v in tab[j, :6])
            self.vinet_unpack_wgrad(dw, N, Cin, ntaps, stem, flags, grad, stream)
        return 0

    # -- layout ----------------------------------------------------------------
    def vinet_import_ncdhw(self, src, sb, sc, st, sh, sw, Cc, dst, dst_dtype, stream):
        dst = _deref(dst)
        span = (dst.B - 1) * sb + (Cc - 1) * sc + (dst.T - 1) * st + (dst.H - 1) * sh + (dst.W - 1) * sw + 1
        raw = _f32(src, span)
        s = np.lib.stride_tricks.as_strided(raw, (dst.B, dst.T, dst.H, dst.W, Cc), (sb * 4, st * 4, sh * 4, sw * 4, sc * 4))
        v = np.zeros((dst.B, dst.T, dst.H, dst.W, dst.C), np.float32)
        v[..., :Cc] = s
        wr(dst, dst_dtype, v)
        return 0

    def vinet_import_ncdhw_pad(self, src, sb, sc, st, sh, sw, Cc, Hs, Ws, pad_top, pad_left, dst, dst_dtype, stream):
        dst = _deref(dst)
        span = (dst.B - 1) * sb + (Cc - 1) * sc + (dst.T - 1) * st + (Hs - 1) * sh + (Ws - 1) * sw + 1
        raw = _f32(src, span)
        s = np.lib.stride_tricks.as_strided(raw, (dst.B, dst.T, Hs, Ws, Cc), (sb * 4, st * 4, sh * 4, sw * 4, sc * 4))
        v = np.zeros((dst.B, dst.T, dst.H, dst.W, dst.C), np.float32)
        v[:, :, pad_top:pad_top + Hs, pad_left:pad_left + Ws, :Cc] = s
        wr(dst, dst_dtype, v)
        return 0

    def vinet_export_ncdhw(self, src, src_dtype, pre, dst, sb, sc, st, sh, sw, accumulate, stream):
        src = _deref(src)
        v = affine(rd(src, src_dtype), pre, src.C)
        span = (src.B - 1) * sb + (src.C - 1) * sc + (src.T - 1) * st + (src.H - 1) * sh + (src.W - 1) * sw + 1
        raw = _f32(dst, span)
        d = np.lib.stride_tricks.as_strided(raw, (src.B, src.T, src.H, src.W, src.C), (sb * 4, st * 4, sh * 4, sw * 4, sc * 4))
        d[...] = d + v if accumulate else v
        return 0

    def vinet_copy_affine(self, src, src_dtype, pre, dst, dst_dtype, accumulate, stream):
        src, dst = _deref(src), _deref(dst)
        wr(dst, dst_dtype, affine(rd(src, src_dtype), pre, src.C), bool(accumulate))
        return 0

    # -- BN ----------------------------------------------------------------------
    def vinet_bn_partials_fold(self, partials, rows, Cc, out, out_rows, stream):
        per = (rows + out_rows - 1) // out_rows
        assert (out_rows - 1) * per < rows
        P = _f32(partials, rows * 2 * Cc).reshape(rows, 2, Cc).astype(np.float64)
        O = _f32(out, out_rows * 2 * Cc).reshape(out_rows, 2, Cc)
        for i in range(out_rows):
            O[i] = P[i * per:(i + 1) * per].sum(0)
        return 0

    def vinet_bn_finalize(self, partials, rows, Cc, ld, count, gamma, beta, eps, momentum, rm, rv, mean, invstd, scale, shift, stream):
        ld = ld or Cc
        P = _f32(partials, (rows * 2 - 1) * ld + Cc).copy()
        P = np.lib.stride_tricks.as_strided(P, (rows, 2, Cc), (8 * ld, 4 * ld, 4)).astype(np.float64)
        s, q = P[:, 0].sum(0), P[:, 1].sum(0)
        mu = s / count
        var = np.maximum(q / count - mu * mu, 0)
        istd = 1.0 / np.sqrt(var + np.float64(np.float32(eps)))
        g = _f32(gamma, Cc) if gamma else np.ones(Cc, np.float32)
        b = _f32(beta, Cc) if beta else np.zeros(Cc, np.float32)
        sc = (g * istd).astype(np.float32)
        if mean:
            _f32(mean, Cc)[:] = mu
        if invstd:
            _f32(invstd, Cc)[:] = istd
        _f32(scale, Cc)[:] = sc
        _f32(shift, Cc)[:] = b - mu.astype(np.float32) * sc
        m = np.float32(momentum)
        if rm:
            r = _f32(rm, Cc)
            r[:] = (1 - m) * r + m * mu.astype(np.float32)
        if rv:
            r = _f32(rv, Cc)
            unb = var * count / (count - 1) if count > 1 else var
            r[:] = (1 - m) * r + m * unb.astype(np.float32)
        return 0

    def vinet_bn_fold(self, gamma, beta, rm, rv, conv_bias, eps, Cc, scale, shift, invstd, stream):
        istd = (1.0 / np.sqrt(_f32(rv, Cc) + np.float32(eps))).astype(np.float32)
        g = _f32(gamma, Cc) if gamma else np.ones(Cc, np.float32)
        b = _f32(beta, Cc) if beta else np.zeros(Cc, np.float32)
        cb = _f32(conv_bias, Cc) if conv_bias else np.zeros(Cc, np.float32)
        sc = g * istd
        _f32(scale, Cc)[:] = sc
        _f32(shift, Cc)[:] = b + (cb - _f32(rm, Cc)) * sc
        if invstd:
            _f32(invstd, Cc)[:] = istd
        return 0

    def vinet_stats_rows(self, x):
        x = _deref(x)
        n = x.B * x.T * x.H * x.W
        return max(1, min(1024, (n + 63) // 64))

    def vinet_channel_stats(self, x, dtype, partials, stream):
        x = _deref(x)
        rows = self.vinet_stats_rows(x)
        v = rd(x, dtype).reshape(-1, x.C).astype(np.float64)
        P = _f32(partials, rows * 2 * x.C).reshape(rows, 2, x.C)
        P[...] = 0
        P[0, 0], P[0, 1] = v.sum(0), (v * v).sum(0)
        return 0

    def _bn_bwd_terms(self, dz, x_raw, dtype, fwd, mean, invstd):
        Cc = x_raw.C
        xv = rd(x_raw, dtype)
        g = rd(dz, dtype)
        if fwd.relu:
            z = xv * _f32(fwd.scale, Cc) + _f32(fwd.shift, Cc) if fwd.scale else xv
            g = g * (z > 0)
        xhat = (xv - _f32(mean, Cc)) * _f32(invstd, Cc)
        return g, xhat

    def vinet_bn_bwd_reduce(self, dz, x_raw, dtype, fwd, mean, invstd, partials, stream):
        dz, x_raw = _deref(dz), _deref(x_raw)
        g, xhat = self._bn_bwd_terms(dz, x_raw, dtype, fwd, mean, invstd)
        rows = self.vinet_stats_rows(x_raw)
        P = _f32(partials, rows * 2 * x_raw.C).reshape(rows, 2, x_raw.C)
        P[...] = 0
        P[0, 0] = g.reshape(-1, x_raw.C).astype(np.float64).sum(0)
        P[0, 1] = (g * xhat).reshape(-1, x_raw.C).astype(np.float64).sum(0)
        return 0

    def vinet_bn_bwd_finalize(self, partials, rows, Cc, ld, count, scale, train, dgamma, dbeta, invstd, c1, c2, stream):
        ld = ld or Cc
        P = _f32(partials, (rows * 2 - 1) * ld + Cc).copy()
        P = np.lib.stride_tricks.as_strided(P, (rows, 2, Cc), (8 * ld, 4 * ld, 4)).astype(np.float64)
        s, p = P[:, 0].sum(0), P[:, 1].sum(0)
        if dgamma:
            _f32(dgamma, Cc)[:] += p.astype(np.float32)
        if dbeta:
            _f32(dbeta, Cc)[:] += s.astype(np.float32)
        if c1:
            _f32(c1, Cc)[:] = s / count if train else 0
        if c2:
            _f32(c2, Cc)[:] = p / count if train else 0
        return 0

    def vinet_bn_bwd_apply(self, dz, x_raw, dtype, fwd, mean, invstd, c1, c2, dx, stream):
        dz, x_raw, dx = _deref(dz), _deref(x_raw), _deref(dx)
        Cc = x_raw.C
        g, xhat = self._bn_bwd_terms(dz, x_raw, dtype, fwd, mean, invstd)
        wr(dx, dtype, _f32(fwd.scale, Cc) * (g - _f32(c1, Cc) - xhat * _f32(c2, Cc)))
        return 0

    def vinet_act_bwd(self, dz, dz_dtype, z, z_dtype, act, dy, dy_dtype, stream):
        dz, z, dy = _deref(dz), _deref(z), _deref(dy)
        g, zv = rd(dz, dz_dtype), rd(z, z_dtype)
        if act == 1:
            g = g * (zv > 0)
        elif act == 2:
            g = g * zv * (1 - zv)
        wr(dy, dy_dtype, g)
        return 0

    def vinet_channel_sum(self, x, dtype, workspace, Cout, out, accumulate, stream):
        x = _deref(x)
        s = rd(x, dtype).reshape(-1, x.C).astype(np.float64).sum(0).reshape(-1, Cout).sum(0)
        o = _f32(out, Cout)
        o[:] = o + s if accumulate else s
        return 0

    # -- pooling / upsample ---------------------------------------------------------
    def vinet_maxpool3d(self, d, x, pre, y, argmax, stream):
        d, x, y = _deref(d), _deref(x), _deref(y)
        xa = affine(rd(x, d.dtype), pre, x.C)
        if d.dtype != F32 and pre.scale:
            # pooling sees the pending affine applied AND rounded to the activation dtype (pool.hip: pool_round)
            xa = _bf2f(_f2bf(np.ascontiguousarray(xa).reshape(-1))).reshape(xa.shape)
        xv = torch.from_numpy(np.ascontiguousarray(xa)).permute(0, 4, 1, 2, 3)
        out, idx = torch.nn.functional.max_pool3d(xv, (d.kT, d.kH, d.kW), (d.sT, d.sH, d.sW), (d.pT, d.pH, d.pW),
                                                  return_indices=True)
        wr(y, d.dtype, out.permute(0, 2, 3, 4, 1).numpy())
        if argmax:
            # flat input index -> window-relative tap
            it = idx // (x.H * x.W)
            ih = (idx // x.W) % x.H
            iw = idx % x.W
            ot = torch.arange(y.T).view(1, 1, -1, 1, 1)
            oh = torch.arange(y.H).view(1, 1, 1, -1, 1)
            ow = torch.arange(y.W).view(1, 1, 1, 1, -1)
            tap = ((it - (ot * d.sT - d.pT)) * d.kH + (ih - (oh * d.sH - d.pH))) * d.kW + (iw - (ow * d.sW - d.pW))
            am = np.ctypeslib.as_array((C.c_uint8 * (y.B * y.T * y.H * y.W * y.C)).from_address(argmax))
            am[:] = tap.permute(0, 2, 3, 4, 1).contiguous().numpy().astype(np.uint8).reshape(-1)
        return 0

    def vinet_maxpool3d_bwd(self, d, dy, argmax, dx, accumulate, stream):
        d, dy, dx = _deref(d), _deref(dy), _deref(dx)
        g = rd(dy, d.dtype)
        am = np.ctypeslib.as_array((C.c_uint8 * g.size).from_address(argmax)).reshape(g.shape).astype(np.int64)
        out = np.zeros((dx.B, dx.T, dx.H, dx.W, dx.C), np.float32)
        kt, r = am // (d.kH * d.kW), am % (d.kH * d.kW)
        kh, kw = r // d.kW, r % d.kW
        b, ot, oh, ow, c = np.meshgrid(*[np.arange(s) for s in g.shape], indexing="ij")
        np.add.at(out, (b, ot * d.sT - d.pT + kt, oh * d.sH - d.pH + kh, ow * d.sW - d.pW + kw, c), g)
        wr(dx, d.dtype, out, bool(accumulate))
        return 0

    def vinet_unfold1d(self, x, y, dtype, stride, pad, stream):
        x, y = _deref(x), _deref(y)
        src = rd(x, dtype)[..., 0].reshape(x.B, x.T)                       # channel 0 of [B,T,1,1,C]
        k = y.C
        padded = np.zeros((x.B, x.T + 2 * pad), np.float32)
        padded[:, pad:pad + x.T] = src
        idx = np.arange(y.T)[:, None] * stride + np.arange(k)[None, :]
        wr(y, dtype, padded[:, idx].reshape(y.B, y.T, 1, 1, k))
        return 0

    def vinet_upsample2x(self, x, y, dtype, stream):
        x, y = _deref(x), _deref(y)
        xv = torch.from_numpy(np.ascontiguousarray(rd(x, dtype))).permute(0, 4, 1, 2, 3)
        out = torch.nn.functional.interpolate(xv, scale_factor=(1, 2, 2), mode="trilinear", align_corners=False)
        wr(y, dtype, out.permute(0, 2, 3, 4, 1).numpy())
        return 0

    def vinet_upsample2x_bwd(self, dy, dx, dtype, accumulate, stream):
        dy, dx = _deref(dy), _deref(dx)
        g = torch.from_numpy(np.ascontiguousarray(rd(dy, dtype))).permute(0, 4, 1, 2, 3)
        # transpose of the separable stencil, written out (no autograd: the custom-op tests call this below the autograd key)
        def U(n):       # [n in, 2n out]: out[o] = sum_i U[i, o] x[i]
            return torch.nn.functional.interpolate(torch.eye(n).unsqueeze(0), scale_factor=2, mode="linear", align_corners=False)[0]
        gx = torch.einsum("io,bctop,jp->bctij", U(dx.H), g.float(), U(dx.W))
        wr(dx, dtype, gx.permute(0, 2, 3, 4, 1).contiguous().numpy(), bool(accumulate))
        return 0

    def vinet_bn_bwd_apply_split(self, dz, x_raw, fwd, mean, invstd, c1, c2, dx, hi, lo, stream):
        rc = self.vinet_bn_bwd_apply(dz, x_raw, F32, fwd, mean, invstd, c1, c2, dx, stream)
        return rc or self.vinet_split_bf16(dx, L.CAffine(None, None, 0), hi, lo, stream)

    def vinet_upsample2x_bwd_relu(self, dy, dx, xf, dtype, stream):
        rc = self.vinet_upsample2x_bwd(dy, dx, dtype, 0, stream)
        dx, xf = _deref(dx), _deref(xf)
        wr(dx, dtype, rd(dx, dtype) * (rd(xf, dtype) > 0))
        return rc

    # -- losses / optimizer -----------------------------------------------------------
    @staticmethod
    def _loss_value(which, s, g):
        eps = 2.2204e-16
        if which == 0:
            p, q = s / s.sum(1, keepdim=True), g / g.sum(1, keepdim=True)
            return (q * torch.log(eps + q / (p + eps))).sum(1)
        if which == 1:
            a = (s - s.mean(1, keepdim=True)) / s.std(1, keepdim=True)
            b = (g - g.mean(1, keepdim=True)) / g.std(1, keepdim=True)
            return (a * b).sum(1) / torch.sqrt((a * a).sum(1) * (b * b).sum(1))
        if which == 3:
            z = (s - s.mean(1, keepdim=True)) / (s.std(1, keepdim=True) + eps)
            return (z * g).sum(1) / g.sum(1)
        ns = (s - s.min(1, keepdim=True)[0]) / (s.max(1, keepdim=True)[0] - s.min(1, keepdim=True)[0])
        ng = (g - g.min(1, keepdim=True)[0]) / (g.max(1, keepdim=True)[0] - g.min(1, keepdim=True)[0])
        return torch.min(ns / ns.sum(1, keepdim=True), ng / ng.sum(1, keepdim=True)).sum(1)

    def _loss_inputs(self, s, gt, gt_is_f64, B, n):
        sv = torch.from_numpy(_f32(s, B * n).reshape(B, n).astype(np.float64))
        if gt_is_f64:
            gv = torch.from_numpy(np.ctypeslib.as_array((C.c_double * (B * n)).from_address(gt)).reshape(B, n).copy())
        else:
            gv = torch.from_numpy(_f32(gt, B * n).reshape(B, n).astype(np.float64))
        return sv, gv

    def vinet_loss_fwd(self, which, s, gt, gt_is_f64, B, n, saved, loss, stream):
        sv, gv = self._loss_inputs(s, gt, gt_is_f64, B, n)
        per = self._loss_value(which, sv, gv)
        _f32(loss, 1)[0] = float(per.mean())
        return 0

    def vinet_loss_bwd(self, which, s, gt, gt_is_f64, B, n, saved, gscale, coeff, accumulate, ds, stream):
        sv, gv = self._loss_inputs(s, gt, gt_is_f64, B, n)

        def run():
            with torch.enable_grad():
                sv.requires_grad_(True)
                self._loss_value(which, sv, gv).mean().backward()
        # (a fresh thread has default dispatch keys: when a torch.library custom op calls this below the autograd key,
        #  autograd is off for the calling thread whatever enable_grad says)
        import threading
        th = threading.Thread(target=run)
        th.start()
        th.join()
        gs = float(_f32(gscale, 1)[0]) if gscale else 1.0
        g = (sv.grad * gs * coeff).numpy().astype(np.float32)
        d = _f32(ds, B * n).reshape(B, n)
        d[...] = d + g if accumulate else g
        return 0

    def vinet_adam_step(self, p, g, m, v, n, lr, b1, b2, eps, bc1, bc2, gs, stream):
        P, G, M_, V = _f32(p, n), _f32(g, n), _f32(m, n), _f32(v, n)
        f = np.float32
        ge = G * f(gs)
        M_[:] = f(b1) * M_ + (f(1) - f(b1)) * ge
        V[:] = f(b2) * V + (f(1) - f(b2)) * ge * ge
        denom = np.sqrt(V) * f(1.0 / np.sqrt(f(bc2))) + f(eps)
        P[:] = P - f(lr / bc1) * (M_ / denom)
        return 0

    # -- saliency-map post-processing (oracle/postproc_cpu.py is the CPU statement of these) -------------
    @staticmethod
    def _mm_key(f):
        u = np.asarray(f, dtype=np.float32).view(np.uint32)
        return np.where(u & 0x80000000, ~u, u | np.uint32(0x80000000)).astype(np.uint32)

    @staticmethod
    def _mm_unkey(k):
        k = np.asarray(k, dtype=np.uint32)
        return np.where(k & 0x80000000, k & np.uint32(0x7FFFFFFF), ~k).astype(np.uint32).view(np.float32)

    def _store_minmax(self, minmax, maps):
        mm = np.ctypeslib.as_array((C.c_uint32 * (2 * maps.shape[0])).from_address(minmax)).reshape(-1, 2)
        flat = maps.reshape(maps.shape[0], -1)
        mm[:, 0] = self._mm_key(flat.min(1))
        mm[:, 1] = self._mm_key(flat.max(1))

    def vinet_resize_blur(self, src, B, H, W, dst, oH, oW, minmax, stream):
        from oracle import postproc_cpu as P
        out = P.resize_blur(_f32(src, B * H * W).reshape(B, H, W), oH, oW)
        _f32(dst, B * oH * oW)[:] = out.reshape(-1)
        if minmax:
            self._store_minmax(minmax, out)
        return 0

    def vinet_minmax(self, src, B, n, minmax, stream):
        self._store_minmax(minmax, _f32(src, B * n).reshape(B, n))
        return 0

    def vinet_normalize_u8(self, src, minmax, B, n, dst, stream):
        from oracle import postproc_cpu as P
        x = _f32(src, B * n).reshape(B, n)
        mm = self._mm_unkey(np.ctypeslib.as_array((C.c_uint32 * (2 * B)).from_address(minmax)).reshape(B, 2))
        out = np.ctypeslib.as_array((C.c_uint8 * (B * n)).from_address(dst)).reshape(B, n)
        for b in range(B):
            assert mm[b, 0] == x[b].min() and mm[b, 1] == x[b].max(), "normalize_u8: stale min/max keys"
            out[b] = P.normalize_u8(x[b].reshape(1, -1)).reshape(-1)
        return 0

    # -- input pipeline (oracle/preproc_cpu.py) -----------------------------------------------------------
    def vinet_frames_preprocess_ws_bytes(self, N, H, W, oH, oW):
        return 256 + N * H * oW * 3

    def vinet_frames_preprocess(self, src, N, H, W, dst, oH, oW, mean_std, ws, stream):
        from oracle import preproc_cpu as Q
        assert ws
        frames = np.ctypeslib.as_array((C.c_uint8 * (N * H * W * 3)).from_address(src)).reshape(N, H, W, 3)
        ms = [float(mean_std[i]) for i in range(6)]
        _f32(dst, N * 3 * oH * oW)[:] = Q.frames_preprocess(frames, oH, oW, ms[:3], ms[3:]).reshape(-1)
        return 0

    def vinet_audio_excerpt(self, wav, n_samples, start, end, out, win, stream):
        from oracle import preproc_cpu as Q
        w = _f32(wav, n_samples) if n_samples else np.zeros(0, np.float32)
        if min(end + 1, n_samples) - start > win:
            return -1
        _f32(out, win)[:] = Q.audio_excerpt(w, start, end, win)
        return 0

    def vinet_gt_preprocess_ws_bytes(self, N, oH, oW):
        return 256 + N * oH * oW * 8

    def vinet_gt_preprocess(self, src, N, H, W, dst, oH, oW, ws, stream):
        from oracle import preproc_cpu as Q
        assert ws
        g = np.ctypeslib.as_array((C.c_uint8 * (N * H * W)).from_address(src)).reshape(N, H, W)
        _f32(dst, N * oH * oW)[:] = Q.gt_preprocess(g, oH, oW).reshape(-1)
        return 0

    # -- bilinear ----------------------------------------------------------------------
    @staticmethod
    def _rdflat(ptr, n, dt):
        r = _raw(ptr, n, dt)
        return r.astype(np.float32) if dt == F32 else _bf2f(r)

    def vinet_bilinear_fwd(self, x1, x2, dtype, w, bias, B, Cc, I, J, O, out, stream):
        a = self._rdflat(x1, B * I * Cc, dtype).reshape(B, I, Cc)
        b = self._rdflat(x2, B * J * Cc, dtype).reshape(B, J, Cc)
        W = _f32(w, O * I * J).reshape(O, I, J)
        o = np.einsum("bic,oij,bjc->boc", a, W, b, optimize=True)
        if bias:
            o = o + _f32(bias, O).reshape(1, O, 1)
        raw = _raw(out, B * O * Cc, dtype)
        raw[:] = o.reshape(-1).astype(np.float32) if dtype == F32 else _f2bf(o.reshape(-1))
        return 0

    def vinet_bilinear_bwd(self, x1, x2, dout, dtype, w, B, Cc, I, J, O, dx1, dx2, dw, dbias, stream):
        a = self._rdflat(x1, B * I * Cc, dtype).reshape(B, I, Cc)
        b = self._rdflat(x2, B * J * Cc, dtype).reshape(B, J, Cc)
        g = self._rdflat(dout, B * O * Cc, dtype).reshape(B, O, Cc)
        W = _f32(w, O * I * J).reshape(O, I, J)
        if dx1:
            v = np.einsum("boc,oij,bjc->bic", g, W, b, optimize=True).reshape(-1)
            _raw(dx1, v.size, dtype)[:] = v.astype(np.float32) if dtype == F32 else _f2bf(v)
        if dx2:
            v = np.einsum("boc,oij,bic->bjc", g, W, a, optimize=True).reshape(-1)
            _raw(dx2, v.size, dtype)[:] = v.astype(np.float32) if dtype == F32 else _f2bf(v)
        if dw:
            _f32(dw, O * I * J)[:] += np.einsum("boc,bic,bjc->oij", g, a, b, optimize=True).reshape(-1)
        if dbias:
            _f32(dbias, O)[:] += g.sum((0, 2))
        return 0
