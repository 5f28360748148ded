"""Saliency losses on MI355X -- drop-in for the reference's loss.py:13-99.

`kldiv(s_map, gt)`, `cc(s_map, gt)`, `similarity(s_map, gt)` keep the reference's
signatures ([B,H,W] maps -> scalar, batch mean) and are differentiable w.r.t.
`s_map`.  Forward and backward each run as one workgroup-per-sample HIP kernel
with fp64 accumulators (libvinet_hip.so: vinet_loss_fwd / vinet_loss_bwd).
Ground truth may be float32 or float64 (the DIEM loader hands over float64,
SURVEY.md F11); like the reference the result is then float64.
"""
import torch

from . import _lib as L
from . import engine as E

_WHICH = {"kldiv": 0, "cc": 1, "similarity": 2, "nss": 3}


class _LossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, s_map, gt, which):
        assert s_map.size() == gt.size()
        assert s_map.dim() == 3, "expected [B,H,W] maps"
        s = s_map.detach()
        if s.dtype != torch.float32 or not s.is_contiguous():
            s = s.float().contiguous()
        g = gt.detach()
        if g.dtype not in (torch.float32, torch.float64):
            g = g.float()
        g = g.contiguous()
        B, n = s.shape[0], s.shape[1] * s.shape[2]
        lib = L.get()
        stream = E._stream_for(s.device)
        saved = torch.empty(B * 8, dtype=torch.float64, device=s.device)
        loss = torch.empty((), dtype=torch.float32, device=s.device)
        L.check(lib.vinet_loss_fwd(which, s.data_ptr(), g.data_ptr(), 1 if g.dtype == torch.float64 else 0, B, n,
                                   saved.data_ptr(), loss.data_ptr(), stream), "vinet_loss_fwd")
        ctx.save_for_backward(s, g, saved)
        ctx.which, ctx.shape = which, s_map.shape
        return loss.double() if g.dtype == torch.float64 else loss

    @staticmethod
    def backward(ctx, gout):
        s, g, saved = ctx.saved_tensors
        B, n = s.shape[0], s.shape[1] * s.shape[2]
        lib = L.get()
        stream = E._stream_for(s.device)
        gs = gout.detach().float().contiguous()
        ds = torch.empty_like(s)
        L.check(lib.vinet_loss_bwd(ctx.which, s.data_ptr(), g.data_ptr(), 1 if g.dtype == torch.float64 else 0, B, n,
                                   saved.data_ptr(), gs.data_ptr(), 1.0, 0, ds.data_ptr(), stream), "vinet_loss_bwd")
        return ds.view(ctx.shape), None, None


def kldiv(s_map, gt):
    """loss.py:13-38."""
    return _LossFn.apply(s_map, gt, 0)


def cc(s_map, gt):
    """loss.py:80-99."""
    return _LossFn.apply(s_map, gt, 1)


def similarity(s_map, gt):
    """loss.py:52-78 (with normalize_map, loss.py:41-50)."""
    return _LossFn.apply(s_map, gt, 2)


@torch.no_grad()
def nss(s_map, gt):
    """loss.py:101-120 for maps of equal size (the reference `cv2.resize`s s_map to gt's size first when they
    differ, which is host post-processing): z-score of the saliency map with the unbiased
    std (+2.2204e-16), mean over the fixation mask; batch mean.  A validation metric: no gradient."""
    assert s_map.size() == gt.size(), "nss: resize the saliency map to the fixation map first"
    assert s_map.dim() == 3, "expected [B,H,W] maps"
    s = s_map.detach().float().contiguous()
    g = gt.detach()
    if g.dtype not in (torch.float32, torch.float64):
        g = g.float()
    g = g.contiguous()
    B, n = s.shape[0], s.shape[1] * s.shape[2]
    lib = L.get()
    saved = torch.empty(B * 8, dtype=torch.float64, device=s.device)
    out = torch.empty((), dtype=torch.float32, device=s.device)
    L.check(lib.vinet_loss_fwd(3, s.data_ptr(), g.data_ptr(), 1 if g.dtype == torch.float64 else 0, B, n,
                               saved.data_ptr(), out.data_ptr(), E._stream_for(s.device)), "vinet_loss_fwd")
    return out.double() if g.dtype == torch.float64 else out
