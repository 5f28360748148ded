"""torch.library custom ops of vinet_amd.ops on CPU through the ABI emulator: values and
gradients against torch's own operators, fake (meta) implementations and schemas through torch.library.opcheck."""
import pytest
import torch
import torch.nn.functional as F

from tests.abi_emulator import AbiEmulator
from vinet_amd import _lib as L
from vinet_amd import engine as E
from vinet_amd import synth


@pytest.fixture(autouse=True)
def _emulated_abi():
    L._install_test_double(AbiEmulator())
    old = E.default_dtype()
    E.set_default_dtype("fp32")
    yield
    E.set_default_dtype("bf16" if old == E.BF16 else "fp32")
    L._install_test_double(None)


def _cl(t):      # NCDHW -> channels-last [B,T,H,W,C]
    return t.permute(0, 2, 3, 4, 1).contiguous()


def _nc(t):
    return t.permute(0, 4, 1, 2, 3)


@pytest.mark.parametrize("k,s,p", [((1, 3, 3), (1, 1, 1), (0, 1, 1)), ((3, 1, 1), (1, 1, 1), (1, 0, 0)), ((2, 3, 3), (2, 1, 1), (0, 1, 1)),
                                   ((1, 1, 1), (1, 1, 1), (0, 0, 0))])
def test_conv3d_op_values_and_gradients(k, s, p):
    from vinet_amd import ops
    x = synth.normal("opx", (2, 8, 4, 6, 8), 1).requires_grad_(True)
    w = (synth.normal("opw", (12, 8) + k, 2) * 0.2).requires_grad_(True)
    b = synth.normal("opb", (12,), 3).requires_grad_(True)
    xc = _cl(x.detach()).requires_grad_(True)
    y = ops.conv3d(xc, w, b, s, p, act=L.ACT_RELU)
    ref = F.relu(F.conv3d(x, w, b, stride=s, padding=p))
    assert torch.allclose(_nc(y), ref, atol=1e-5)
    proj = synth.normal("opp", tuple(ref.shape), 4)
    (_nc(y) * proj).sum().backward()
    gw, gb = w.grad.clone(), b.grad.clone()
    w.grad = b.grad = None
    (ref * proj).sum().backward()
    assert torch.allclose(_nc(xc.grad), x.grad, atol=1e-4)
    assert torch.allclose(gw, w.grad, atol=1e-4) and torch.allclose(gb, b.grad, atol=1e-4)


def test_pool_upsample_ops():
    from vinet_amd import ops
    x = synth.normal("opool", (1, 8, 4, 8, 10), 5).requires_grad_(True)
    xc = _cl(x.detach()).requires_grad_(True)
    y = ops.upsample2x(ops.maxpool3d(xc, (1, 3, 3), (1, 2, 2), (0, 1, 1)))
    ref = F.interpolate(F.max_pool3d(x, (1, 3, 3), (1, 2, 2), (0, 1, 1)), scale_factor=(1, 2, 2), mode="trilinear")
    assert torch.allclose(_nc(y), ref, atol=1e-6)
    proj = synth.normal("opoolp", tuple(ref.shape), 6)
    (_nc(y) * proj).sum().backward()
    (ref * proj).sum().backward()
    assert torch.allclose(_nc(xc.grad), x.grad, atol=1e-5)


def test_loss_and_adam_ops():
    from vinet_amd import loss as VL
    from vinet_amd import ops  # noqa: F401
    s = synth.uniform("ops", (2, 24, 40), 7, 0.01, 0.99).requires_grad_(True)
    g = synth.gt_map(2, 24, 40, 7)
    for which, fn in ((0, VL.kldiv), (1, VL.cc), (2, VL.similarity)):
        v, _ = torch.ops.vinet.saliency_loss(s, g, which)
        v.backward()
        g1, s.grad = s.grad.clone(), None
        w = fn(s, g)
        w.backward()
        assert torch.allclose(v, w.detach()) and torch.allclose(g1, s.grad)
        s.grad = None
    p = synth.normal("adp", (1000,), 8)
    gr = synth.normal("adg", (1000,), 9)
    pt = p.clone().requires_grad_(True)
    opt = torch.optim.Adam([pt], lr=1e-3)
    m, v = torch.zeros(1000), torch.zeros(1000)
    for step in (1, 2, 3):
        pt.grad = gr.clone()
        opt.step()
        torch.ops.vinet.adam_step_(p, gr, m, v, 1e-3, 0.9, 0.999, 1e-8, step, 1.0)
    assert torch.allclose(p, pt.detach(), atol=1e-6)


def test_opcheck_schemas_and_fake_implementations():
    """torch.library.opcheck: schema, fake tensor (register_fake) and autograd registration consistency"""
    from vinet_amd import ops  # noqa: F401
    x = _cl(synth.normal("ocx", (1, 8, 2, 4, 4), 1))
    w = synth.normal("ocw", (8, 8, 1, 3, 3), 2)
    tests = ("test_schema", "test_faketensor")
    torch.library.opcheck(torch.ops.vinet.conv3d_fwd.default, (x, w, None, [1, 1, 1], [0, 1, 1], 1), test_utils=tests)
    y = torch.ops.vinet.conv3d_fwd(x, w, None, [1, 1, 1], [0, 1, 1], 0)
    torch.library.opcheck(torch.ops.vinet.conv3d_bwd_data.default, (y, x, w, [1, 1, 1], [0, 1, 1]), test_utils=tests)
    torch.library.opcheck(torch.ops.vinet.conv3d_bwd_weight.default, (y, x, w, [1, 1, 1], [0, 1, 1]), test_utils=tests)
    torch.library.opcheck(torch.ops.vinet.maxpool3d_fwd.default, (x, [1, 3, 3], [1, 2, 2], [0, 1, 1]), test_utils=tests)
    torch.library.opcheck(torch.ops.vinet.upsample2x_fwd.default, (x,), test_utils=tests)
    s = synth.uniform("ocs", (2, 8, 8), 3, 0.01, 0.99)
    torch.library.opcheck(torch.ops.vinet.saliency_loss.default, (s, synth.gt_map(2, 8, 8, 3), 0), test_utils=tests)
    # a fake-tensor trace sees the right shapes without touching the library
    from torch._subclasses.fake_tensor import FakeTensorMode
    with FakeTensorMode():
        fx = torch.empty(2, 8, 16, 32, 64, dtype=torch.bfloat16)
        fw = torch.empty(192, 64, 1, 3, 3)
        out = torch.ops.vinet.conv3d_fwd(fx, fw, None, [1, 1, 1], [0, 1, 1], 1)
        assert tuple(out.shape) == (2, 8, 16, 32, 192) and out.dtype == torch.bfloat16
        pooled, am = torch.ops.vinet.maxpool3d_fwd(out, [1, 3, 3], [1, 2, 2], [0, 1, 1])
        assert tuple(pooled.shape) == (2, 8, 8, 16, 192) and am.dtype == torch.uint8
