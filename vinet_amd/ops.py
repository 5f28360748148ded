"""torch.library custom ops over the C ABI of libvinet_hip.so (north_star: "a thin C-ABI .so loaded via PyTorch-ROCm
custom ops"; SURVEY.md section 8(b): `torch.library.custom_op` + `register_fake` + `register_autograd`).

The drop-in modules of vinet_amd.model / model_utils run a whole network call as ONE autograd node (engine tape: fused
BatchNorm bookkeeping, concat-free buffers, side-stream weight gradients).  This file exposes the same kernels one
operator at a time, the way torch expects operators: schemas, fake (meta) implementations for shape inference under
`torch.compile` / FakeTensor / DDP tracing, and autograd formulas whose backward calls the `_bwd_*` operators, which
are the data-gradient / weight-gradient entry points of the library (convolution_backward of train.py:216).

Tensors are channels-last `[B, T, H, W, C]` (contiguous), bf16 or fp32, C a multiple of 16 bytes worth of elements;
weights and biases are fp32 in torch's own layout (`[N, Cin, kT, kH, kW]`), exactly the reference's parameters.

    torch.ops.vinet.conv3d(x, weight, bias, stride, padding, act)        nn.Conv3d (+ ReLU)   model_utils.py:131,144,148; model.py:256-282
    torch.ops.vinet.conv3d_bwd_data / conv3d_bwd_weight                  its two backward halves
    torch.ops.vinet.maxpool3d / maxpool3d_bwd                            nn.MaxPool3d          model.py:696-714, model_utils.py:178
    torch.ops.vinet.upsample2x / upsample2x_bwd                          nn.Upsample((1,2,2), trilinear)  model.py:254
    torch.ops.vinet.saliency_loss / saliency_loss_bwd                    kldiv / cc / similarity          loss.py:13-99
    torch.ops.vinet.adam_step_                                           torch.optim.Adam.step over a flat buffer   train.py:188,217

There is no CPU kernel behind them: on a CPU tensor they raise (vinet_amd._lib), except under the tests' ABI double.
"""
import ctypes as C
from typing import List, Optional, Tuple

import torch
from torch import Tensor

from . import _lib as L
from . import engine as E

_DT = {torch.float32: E.F32, torch.bfloat16: E.BF16}


def _view(t):
    assert t.dim() == 5 and t.is_contiguous(), "expected a contiguous channels-last [B,T,H,W,C] tensor"
    B, T, H, W, Cc = t.shape
    dt = _DT[t.dtype]
    assert Cc % E.EG[dt] == 0, "C must be a multiple of %d for %s" % (E.EG[dt], t.dtype)
    return E.View(t.view(-1), 0, B, T, H, W, Cc, Cc, T * H * W * Cc, dt)


def _ctx(t, record=False):
    return E.Ctx(t.device, _DT[t.dtype], training=False, record=record)


def _out_thw(thw, k, s, p):
    return [(d + 2 * pp - kk) // ss + 1 for d, kk, ss, pp in zip(thw, k, s, p)]


def _plan(weight, bias, stride, padding):
    return E.ConvPlan(weight, bias, tuple(weight.shape[2:]), tuple(stride), tuple(padding))


# ---- convolution ---------------------------------------------------------------------------------------------------------
@torch.library.custom_op("vinet::conv3d_fwd", mutates_args=())
def conv3d_fwd(x: Tensor, weight: Tensor, bias: Optional[Tensor], stride: List[int], padding: List[int], act: int) -> Tensor:
    ctx = _ctx(x)
    plan = _plan(weight.detach(), None if bias is None else bias.detach(), stride, padding)
    y = E.conv_forward(ctx, plan, E.Act(_view(x)), act=act)
    v = y.v
    return v.buf.view(v.B, v.T, v.H, v.W, v.C)


@conv3d_fwd.register_fake
def _(x, weight, bias, stride, padding, act):
    oT, oH, oW = _out_thw(x.shape[1:4], weight.shape[2:], stride, padding)
    return x.new_empty((x.shape[0], oT, oH, oW, weight.shape[0]))


def _standalone_backward(x, dy, weight, stride, padding, want_dx, want_dw):
    """one conv's backward through the engine's own routine (stride-phase data gradient, side-stream weight gradient)"""
    ctx = _ctx(dy, record=True)
    w = weight.detach().clone().requires_grad_(want_dw) if want_dw else weight.detach()
    plan = _plan(w, None, stride, padding)
    xa = E.Act(_view(x), needs_grad=want_dx)
    res = E.Act(_view(dy))
    res._grad, res.grad_ready = res.v, True
    M = dy.shape[0] * dy.shape[1] * dy.shape[2] * dy.shape[3]
    ctx._deferred, ctx._side_keep, ctx.side_used, ctx._unpack_jobs = [], [], False, []
    E._conv_backward(ctx, plan, xa, res, None, L.ACT_NONE, False, {}, M)
    ctx.flush_deferred()
    ctx.flush_unpack()
    if getattr(ctx, "side_used", False):
        for st in ctx.side_streams():
            torch.cuda.current_stream(ctx.device).wait_stream(st)
    dx = None
    if want_dx:
        g = xa.grad_view()
        dx = g.buf.view(g.B, g.T, g.H, g.W, g.C)
    return dx, (w.grad if want_dw else None)


@torch.library.custom_op("vinet::conv3d_bwd_data", mutates_args=())
def conv3d_bwd_data(dy: Tensor, x: Tensor, weight: Tensor, stride: List[int], padding: List[int]) -> Tensor:
    """dx of nn.Conv3d; `x` supplies the input extent only"""
    return _standalone_backward(x, dy.contiguous(), weight, stride, padding, True, False)[0]


@conv3d_bwd_data.register_fake
def _(dy, x, weight, stride, padding):
    return x.new_empty(x.shape)


@torch.library.custom_op("vinet::conv3d_bwd_weight", mutates_args=())
def conv3d_bwd_weight(dy: Tensor, x: Tensor, weight: Tensor, stride: List[int], padding: List[int]) -> Tensor:
    """dW of nn.Conv3d in torch layout, fp32; `weight` supplies the shape only"""
    return _standalone_backward(x, dy.contiguous(), weight, stride, padding, False, True)[1]


@conv3d_bwd_weight.register_fake
def _(dy, x, weight, stride, padding):
    return weight.new_empty(weight.shape, dtype=torch.float32)


@torch.library.custom_op("vinet::act_bwd", mutates_args=())
def act_bwd(dy: Tensor, y: Tensor, act: int) -> Tensor:
    """gradient through ReLU (1) / sigmoid (2) given the activation's OUTPUT"""
    out = torch.empty_like(dy)
    ctx = _ctx(dy)
    ctx.call("vinet_act_bwd", C.byref(_view(dy).ct()), _DT[dy.dtype], C.byref(_view(y).ct()), _DT[y.dtype], act, C.byref(_view(out).ct()),
             _DT[out.dtype], ctx.stream)
    return out


@act_bwd.register_fake
def _(dy, y, act):
    return torch.empty_like(dy)


def _conv3d_setup(ctx, inputs, output):
    x, weight, bias, stride, padding, act = inputs
    ctx.save_for_backward(x, weight, output)
    ctx.stride, ctx.padding, ctx.act, ctx.has_bias = list(stride), list(padding), act, bias is not None


def _conv3d_backward(ctx, gy):
    x, weight, y = ctx.saved_tensors
    gy = gy.contiguous()
    if ctx.act != L.ACT_NONE:
        gy = torch.ops.vinet.act_bwd(gy, y, ctx.act)
    dx = torch.ops.vinet.conv3d_bwd_data(gy, x, weight, ctx.stride, ctx.padding) if ctx.needs_input_grad[0] else None
    dw = torch.ops.vinet.conv3d_bwd_weight(gy, x, weight, ctx.stride, ctx.padding) if ctx.needs_input_grad[1] else None
    db = gy.float().sum(dim=(0, 1, 2, 3)) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
    return dx, dw, db, None, None, None


conv3d_fwd.register_autograd(_conv3d_backward, setup_context=_conv3d_setup)


def conv3d(x, weight, bias=None, stride=(1, 1, 1), padding=(0, 0, 0), act=L.ACT_NONE):
    """differentiable nn.Conv3d (+ ReLU / sigmoid epilogue) on a channels-last tensor"""
    return torch.ops.vinet.conv3d_fwd(x, weight, bias, list(stride), list(padding), act)


# ---- max pooling -----------------------------------------------------------------------------------------------------------
@torch.library.custom_op("vinet::maxpool3d_fwd", mutates_args=())
def maxpool3d_fwd(x: Tensor, kernel: List[int], stride: List[int], padding: List[int]) -> Tuple[Tensor, Tensor]:
    xv = _view(x)
    oT, oH, oW = _out_thw(x.shape[1:4], kernel, stride, padding)
    y = x.new_empty((x.shape[0], oT, oH, oW, x.shape[4]))
    am = torch.empty(y.numel(), dtype=torch.uint8, device=x.device)
    ctx = _ctx(x)
    pd = L.CPoolDesc(xv.dt, *kernel, *stride, *padding)
    ctx.call("vinet_maxpool3d", C.byref(pd), C.byref(xv.ct()), L.CAffine(None, None, 0), C.byref(_view(y).ct()), am.data_ptr(), ctx.stream)
    return y, am


@maxpool3d_fwd.register_fake
def _(x, kernel, stride, padding):
    oT, oH, oW = _out_thw(x.shape[1:4], kernel, stride, padding)
    y = x.new_empty((x.shape[0], oT, oH, oW, x.shape[4]))
    return y, x.new_empty((y.numel(),), dtype=torch.uint8)


@torch.library.custom_op("vinet::maxpool3d_bwd", mutates_args=())
def maxpool3d_bwd(dy: Tensor, argmax: Tensor, in_shape: List[int], kernel: List[int], stride: List[int], padding: List[int]) -> Tensor:
    dy = dy.contiguous()
    dx = dy.new_empty(in_shape)
    ctx = _ctx(dy)
    pd = L.CPoolDesc(_DT[dy.dtype], *kernel, *stride, *padding)
    ctx.call("vinet_maxpool3d_bwd", C.byref(pd), C.byref(_view(dy).ct()), argmax.data_ptr(), C.byref(_view(dx).ct()), 0, ctx.stream)
    return dx


@maxpool3d_bwd.register_fake
def _(dy, argmax, in_shape, kernel, stride, padding):
    return dy.new_empty(in_shape)


def _pool_setup(ctx, inputs, output):
    x, kernel, stride, padding = inputs
    ctx.save_for_backward(output[1])
    ctx.args = (list(x.shape), list(kernel), list(stride), list(padding))


def _pool_backward(ctx, gy, gam):
    (am,) = ctx.saved_tensors
    return torch.ops.vinet.maxpool3d_bwd(gy, am, *ctx.args), None, None, None


maxpool3d_fwd.register_autograd(_pool_backward, setup_context=_pool_setup)


def maxpool3d(x, kernel, stride, padding=(0, 0, 0)):
    return torch.ops.vinet.maxpool3d_fwd(x, list(kernel), list(stride), list(padding))[0]


# ---- (1,2,2) trilinear upsample ---------------------------------------------------------------------------------------------
@torch.library.custom_op("vinet::upsample2x_fwd", mutates_args=())
def upsample2x_fwd(x: Tensor) -> Tensor:
    B, T, H, W, Cc = x.shape
    y = x.new_empty((B, T, 2 * H, 2 * W, Cc))
    ctx = _ctx(x)
    ctx.call("vinet_upsample2x", C.byref(_view(x).ct()), C.byref(_view(y).ct()), _DT[x.dtype], ctx.stream)
    return y


@upsample2x_fwd.register_fake
def _(x):
    B, T, H, W, Cc = x.shape
    return x.new_empty((B, T, 2 * H, 2 * W, Cc))


@torch.library.custom_op("vinet::upsample2x_bwd", mutates_args=())
def upsample2x_bwd(dy: Tensor) -> Tensor:
    dy = dy.contiguous()
    B, T, H2, W2, Cc = dy.shape
    dx = dy.new_empty((B, T, H2 // 2, W2 // 2, Cc))
    ctx = _ctx(dy)
    ctx.call("vinet_upsample2x_bwd", C.byref(_view(dy).ct()), C.byref(_view(dx).ct()), _DT[dy.dtype], 0, ctx.stream)
    return dx


@upsample2x_bwd.register_fake
def _(dy):
    B, T, H2, W2, Cc = dy.shape
    return dy.new_empty((B, T, H2 // 2, W2 // 2, Cc))


upsample2x_fwd.register_autograd(lambda ctx, gy: torch.ops.vinet.upsample2x_bwd(gy), setup_context=lambda ctx, inputs, output: None)


def upsample2x(x):
    return torch.ops.vinet.upsample2x_fwd(x)


# ---- saliency losses ----------------------------------------------------------------------------------------------------------
@torch.library.custom_op("vinet::saliency_loss", mutates_args=())
def saliency_loss(s_map: Tensor, gt: Tensor, which: int) -> Tuple[Tensor, Tensor]:
    """which: 0 kldiv, 1 cc, 2 similarity (loss.py:13-99) -> (loss scalar fp32, per-sample saved statistics)"""
    s = s_map.float().contiguous()
    g = gt.contiguous()
    B, n = s.shape[0], s.shape[1] * s.shape[2]
    saved = torch.empty(B * 8, dtype=torch.float64, device=s.device)
    out = torch.empty((), dtype=torch.float32, device=s.device)
    L.check(L.get().vinet_loss_fwd(which, s.data_ptr(), g.data_ptr(), 1 if g.dtype == torch.float64 else 0, B, n, saved.data_ptr(),
                                   out.data_ptr(), E._stream_for(s.device)), "vinet_loss_fwd")
    return out, saved


@saliency_loss.register_fake
def _(s_map, gt, which):
    return s_map.new_empty((), dtype=torch.float32), s_map.new_empty((s_map.shape[0] * 8,), dtype=torch.float64)


@torch.library.custom_op("vinet::saliency_loss_bwd", mutates_args=())
def saliency_loss_bwd(gout: Tensor, s_map: Tensor, gt: Tensor, saved: Tensor, which: int) -> Tensor:
    s = s_map.float().contiguous()
    g = gt.contiguous()
    B, n = s.shape[0], s.shape[1] * s.shape[2]
    ds = torch.empty_like(s)
    gs = gout.detach().float().contiguous()
    L.check(L.get().vinet_loss_bwd(which, s.data_ptr(), g.data_ptr(), 1 if g.dtype == torch.float64 else 0, B, n, saved.data_ptr(),
                                   gs.data_ptr(), 1.0, 0, ds.data_ptr(), E._stream_for(s.device)), "vinet_loss_bwd")
    return ds


@saliency_loss_bwd.register_fake
def _(gout, s_map, gt, saved, which):
    return s_map.new_empty(s_map.shape, dtype=torch.float32)


def _loss_setup(ctx, inputs, output):
    s_map, gt, which = inputs
    ctx.save_for_backward(s_map, gt, output[1])
    ctx.which = which


def _loss_backward(ctx, gl, gsaved):
    s_map, gt, saved = ctx.saved_tensors
    return torch.ops.vinet.saliency_loss_bwd(gl, s_map, gt, saved, ctx.which), None, None


saliency_loss.register_autograd(_loss_backward, setup_context=_loss_setup)


# ---- fused Adam over a flat buffer ----------------------------------------------------------------------------------------------
@torch.library.custom_op("vinet::adam_step_", mutates_args=("p", "m", "v"))
def adam_step_(p: Tensor, g: Tensor, m: Tensor, v: Tensor, lr: float, beta1: float, beta2: float, eps: float, step: int, grad_scale: float) -> None:
    """torch.optim.Adam's update (no weight decay, no amsgrad) of flat fp32 buffers, in place"""
    bc1, bc2 = 1.0 - beta1 ** step, 1.0 - beta2 ** step
    L.check(L.get().vinet_adam_step(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), lr, beta1, beta2, eps, bc1, bc2,
                                    grad_scale, E._stream_for(p.device)), "vinet_adam_step")


@adam_step_.register_fake
def _(p, g, m, v, lr, beta1, beta2, eps, step, grad_scale):
    return None
