"""AViNet audio-visual fusion: nn.Bilinear(42, 3, 336) over channels
(model.py:230,235-237) as HIP kernels on channels-last tensors."""
import ctypes as C

from . import _lib as L
from . import engine as E


def bilinear_forward(ctx, bil, y0, audio, out_thw):
    """y0 [B,1,7,6,C] (x1: I = 42 positions), audio [B,3,1,1,C] (x2: J = 3) ->
    Act [B,4,7,12,C] with out[b, o, c] = sum_ij x1[b,i,c] W[o,i,j] x2[b,j,c] + bias[o]."""
    x1 = E.materialize(ctx, y0) if not _dense_plain(y0) else y0
    x2 = E.materialize(ctx, audio) if not _dense_plain(audio) else audio
    E._note_reader(ctx, x1)
    E._note_reader(ctx, x2)
    v1, v2 = x1.v, x2.v
    B, Cc = v1.B, v1.C
    I, J = v1.T * v1.H * v1.W, v2.T * v2.H * v2.W
    O = bil.out_features
    assert (I, J) == (bil.in1_features, bil.in2_features) and O == out_thw[0] * out_thw[1] * out_thw[2]
    out = E.Act(E.View.alloc(B, out_thw[0], out_thw[1], out_thw[2], Cc, ctx.dt, ctx.device), needs_grad=True)
    w, bias = bil.weight, bil.bias
    ctx.call("vinet_bilinear_fwd", v1.ptr(), v2.ptr(), ctx.dt, w.data_ptr(), E._ptr(bias), B, Cc, I, J, O,
             out.v.ptr(), ctx.stream)
    if ctx.recording:
        def bwd():
            dg = out.grad_view()
            d1 = x1.grad_view() if x1.needs_grad else None
            d2 = x2.grad_view() if x2.needs_grad else None
            assert not (x1.is_grad_ready() or x2.is_grad_ready()), "bilinear inputs have a single consumer"
            gw = E._param_grad(w) if w.requires_grad else None
            gb = E._param_grad(bias) if (bias is not None and bias.requires_grad) else None
            ctx.call("vinet_bilinear_bwd", v1.ptr(), v2.ptr(), dg.ptr(), ctx.dt, w.data_ptr(), B, Cc, I, J, O,
                     d1.ptr() if d1 else None, d2.ptr() if d2 else None, E._ptr(gw), E._ptr(gb), ctx.stream)
            E._note_param_grad(ctx, w, bias)
            if d1 is not None:
                x1.mark_grad_ready()
            if d2 is not None:
                x2.mark_grad_ready()
        ctx.record(bwd)
    return out


def _dense_plain(a):
    v = a.v
    return a.plain and v.ld == v.C and v.sB == v.T * v.H * v.W * v.C
