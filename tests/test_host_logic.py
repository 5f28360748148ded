"""Host logic of vinet_amd (tap tables, dgrad phases, BN bookkeeping, concat
plumbing, backward tape, losses glue, fused optimizer) on CPU, with the C ABI
served by tests/abi_emulator.py, against the golden vectors of the reference."""
import json

import numpy as np
import pytest
import torch

from oracle import vinet_cpu as O
from tests import goldens as G
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


def _close(a, b, tol):
    a, b = torch.as_tensor(a), torch.as_tensor(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    d = float((a.double() - b.double()).abs().max())
    assert d <= tol, "max abs diff %g > %g" % (d, tol)


def _blocks():
    from vinet_amd import model_utils as MU
    return {
        "basic_16_32": lambda: MU.BasicConv3d(16, 32, 1, 1),
        "sep_16_32_k3": lambda: MU.SepConv3d(16, 32, 3, 1, 1),
        "sep_3_64_k7s2": lambda: MU.SepConv3d(3, 64, 7, 2, 3),
        "mixed_3b": lambda: MU.Mixed_3b(),
    }


@pytest.mark.parametrize("name", ["basic_16_32", "sep_16_32_k3", "sep_3_64_k7s2", "mixed_3b"])
@pytest.mark.parametrize("mode", ["eval", "train"])
def test_blocks(name, mode):
    z, meta = G.load("block_" + name)
    m = _blocks()[name]()
    m.load_state_dict(synth.synth_state_dict(m.state_dict(), meta["seed"]))
    m.train(mode == "train")
    x = synth.normal("x_" + name, tuple(meta["in_shape"]), meta["seed"]).requires_grad_(True)
    y = m(x)
    _close(y.detach(), z[mode + "_y"], 2e-5)
    proj = synth.normal("proj_" + name, tuple(y.shape), meta["seed"])
    (y * proj).sum().backward()
    _close(x.grad, z[mode + "_gx"], 2e-4)
    for k, p in m.named_parameters():
        ref = torch.as_tensor(z[mode + "_g:" + k])
        tol = 2e-4 * max(1.0, float(ref.abs().max()))
        _close(p.grad, ref, tol)
    if mode == "train":
        for k, v in m.state_dict().items():
            if "running" in k:
                _close(v, z["train_stat:" + k], 1e-5)


def test_dgrad_phase_decomposition_matches_conv_transpose():
    """every (k, s, p) the nets use, plus a few odd ones, as 1-D identities"""
    for I, k, s, p in [(32, 7, 2, 3), (12, 3, 3, 0), (20, 5, 5, 0), (9, 3, 1, 1), (70560, 64, 2, 32), (11, 4, 2, 2), (7, 2, 3, 0)]:
        O = (I + 2 * p - k) // s + 1
        if I > 1000:
            I, O = 200, (200 + 2 * p - k) // s + 1
        w = torch.randn(1, 1, k, dtype=torch.float64)
        dy = torch.randn(1, 1, O, dtype=torch.float64)
        x = torch.zeros(1, 1, I, dtype=torch.float64, requires_grad=True)
        torch.nn.functional.conv1d(x, w, stride=s, padding=p).backward(dy)
        got = torch.zeros(I, dtype=torch.float64)
        for r, Q, taps in E._phase_taps_1d(I, O, k, s, p):
            for q in range(Q):
                acc = 0.0
                for off, ki in taps:
                    o = q + off
                    if 0 <= o < O:
                        acc += float(dy[0, 0, o]) * float(w[0, 0, ki])
                got[r + q * s] = acc
        assert torch.allclose(got, x.grad[0, 0], atol=1e-12), (I, k, s, p)


def test_losses_match_reference_goldens():
    from vinet_amd import loss as VL
    z, meta = G.load("loss")
    for tag, (B, H, W) in {"full": (2, 224, 384), "small": (3, 40, 56)}.items():
        s = synth.uniform("loss_s_" + tag, (B, H, W), meta["seed"], 0.01, 0.99)
        g = synth.gt_map(B, H, W, meta["seed"])
        for fn in ("kldiv", "cc", "similarity"):
            si = s.clone().requires_grad_(True)
            v = getattr(VL, fn)(si, g)
            v.backward()
            _close(v.detach(), z["%s_%s" % (tag, fn)], 2e-6)
            if tag == "small":
                _close(si.grad, z["small_%s_grad" % fn], 1e-7)
        assert abs(float(VL.kldiv(s, g.double())) - float(z["%s_kldiv_gt64" % tag])) < 1e-6


def test_decoder8_forward_backward():
    from vinet_amd import model as VM
    z, meta = G.load("decoder8")
    m = VM.DecoderConvUp8()
    m.load_state_dict(G.state_dict_for(m, meta["seed"], z, meta))
    ys = [synth.normal("dec_y%d" % i, tuple(s), meta["seed"]).abs().requires_grad_(True) for i, s in enumerate(meta["shapes"])]
    o = m(*ys)
    _close(o.detach(), z["out"], 2e-5)
    proj = synth.normal("dec_proj", tuple(o.shape), meta["seed"])
    (o * proj).sum().backward()
    _close(ys[0].grad, z["gy0"], 2e-4 * max(1.0, float(np.abs(z["gy0"]).max())))
    _close(ys[1].grad, z["gy1"], 2e-4 * max(1.0, float(np.abs(z["gy1"]).max())))
    for i in (2, 3):
        _close(ys[i].grad.reshape(-1)[:4096], z["gy%d_head" % i], 2e-4 * max(1.0, float(np.abs(z["gy%d_head" % i]).max())))
    for k, p in m.named_parameters():
        ref = z["gp_head:" + k]
        _close(p.grad.reshape(-1)[:2048], ref, 3e-4 * max(1.0, float(np.abs(ref).max())))


@pytest.mark.slow
def test_e2e_8x96x192_inference():
    from vinet_amd import model as VM
    z, meta = G.load("e2e_8x96x192")
    m = VM.VideoSaliencyModel(num_clips=8).eval()
    m.load_state_dict(G.state_dict_for(m, meta["weight_seed"], z, meta))
    x = synth.clip(1, 8, meta["H"], meta["W"], meta["clip_seed"]).permute(0, 2, 1, 3, 4)
    with torch.no_grad():
        y = m(x)
    assert y.shape == (1, meta["H"], meta["W"])
    _close(y, z["y"], 1e-4)
    assert int(y.reshape(-1).argmax()) == meta["argmax"]


@pytest.mark.slow
def test_train_step_matches_reference():
    from vinet_amd import loss as VL
    from vinet_amd import model as VM
    from vinet_amd import optim as VO
    z, meta = G.load("train_step")
    B, T, H, W = meta["B"], meta["T"], meta["H"], meta["W"]
    x = synth.clip(B, T, H, W, meta["seed"]).permute(0, 2, 1, 3, 4)
    gt = synth.gt_map(B, H, W, meta["seed"])
    m = VM.VideoSaliencyModel(num_clips=8)
    m.load_state_dict(G.state_dict_for(m, meta["seed"], z, meta))
    m.train()
    opt = VO.Adam([p for p in m.parameters() if p.requires_grad], lr=meta["lr"])
    opt.zero_grad()
    pred = m(x)
    loss0 = VL.kldiv(pred, gt)
    loss0.backward()
    _close(pred.detach(), z["pred"], 2e-5)
    _close(loss0.detach(), z["loss0"], 1e-5)
    # Gradients: with B=2 the deepest BatchNorms see 12 samples per channel and the
    # reference's own fp32 gradients sit ~1.5e-2 (relative) from the fp64 truth there.
    # Criterion: we must be as close to the fp64 oracle as the fp32 reference is.
    params = dict(m.named_parameters())
    truth, ref32 = {}, {}
    for dt, store in ((torch.float64, truth), (torch.float32, ref32)):
        o = O.VideoSaliencyModel(num_clips=8)
        o.load_state_dict(G.state_dict_for(o, meta["seed"], z, meta))
        o = o.to(dt).train()
        O.kldiv(o(x.to(dt)), gt.to(dt)).backward()
        store.update({k: p.grad.double() for k, p in o.named_parameters()})
    worst = 0.0
    for k, p in params.items():
        t = truth[k]
        e_ref = float((ref32[k] - t).norm() / (t.norm() + 1e-30))
        e_me = float((p.grad.double() - t).norm() / (t.norm() + 1e-30))
        assert e_me <= 3.0 * e_ref + 2e-4, "%s: rel err %.3e vs reference-fp32 %.3e" % (k, e_me, e_ref)
        worst = max(worst, e_me)
    assert worst < 5e-2
    names = json.loads(str(z["grad_names"]))
    gq = np.array([float((params[k].grad.double() ** 2).sum()) for k in names])
    np.testing.assert_allclose(gq, z["grad_sqsum"], rtol=8e-2, atol=1e-12)
    opt.step()
    with torch.no_grad():
        loss1 = VL.kldiv(m(x), gt)      # train-mode forward: second running-stat update, as in the fixture
    # Adam's first step is sign descent (m/sqrt(v) = +-1): fp32-noise-level gradient
    # entries flip sign between implementations, so loss1 agrees to ~1e-3, not 1e-5
    _close(loss1, z["loss1"], 3e-3)
    assert float(loss1) < float(loss0) - 0.2
    sd = m.state_dict()
    for k in [n for n in z.files if n.startswith("state:")]:
        # base4 statistics come from 12 samples/channel on weights that already took one
        # sign-descent step: only the large-M stem statistics are tight
        _close(sd[k[6:]], z[k], 5e-5 if "base1" in k else 2e-3)
