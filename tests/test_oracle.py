"""The CPU oracle (oracle/vinet_cpu.py) against the golden vectors captured
from the real reference (tests/golden/make_goldens.py).  CPU only."""
import json

import numpy as np
import pytest
import torch

from oracle import vinet_cpu as O
from tests import goldens as G
from vinet_amd import synth

TOL = 2e-6


def _close(a, b, tol=TOL):
    a, b = torch.as_tensor(a), torch.as_tensor(b)
    assert a.shape == b.shape
    assert float((a.double() - b.double()).abs().max()) <= tol


BLOCKS = {
    "basic_16_32": lambda: O.BasicConv3d(16, 32, 1, 1),
    "sep_16_32_k3": lambda: O.SepConv3d(16, 32, 3, 1, 1),
    "sep_3_64_k7s2": lambda: O.SepConv3d(3, 64, 7, 2, 3),
    "mixed_3b": lambda: O.Mixed_3b(),
}


@pytest.mark.parametrize("name", list(BLOCKS))
def test_blocks(name):
    z, meta = G.load("block_" + name)
    m = BLOCKS[name]()
    sd = synth.synth_state_dict(m.state_dict(), meta["seed"])
    x = synth.normal("x_" + name, tuple(meta["in_shape"]), meta["seed"])
    for mode in ("eval", "train"):
        m.load_state_dict(sd)
        m.train(mode == "train")
        xi = x.clone().requires_grad_(True)
        y = m(xi)
        proj = synth.normal("proj_" + name, tuple(y.shape), meta["seed"])
        m.zero_grad()
        (y * proj).sum().backward()
        _close(y.detach(), z[mode + "_y"])
        _close(xi.grad, z[mode + "_gx"])
        for k, p in m.named_parameters():
            _close(p.grad, z[mode + "_g:" + k], 1e-5)
        if mode == "train":
            for k, v in m.state_dict().items():
                if "running" in k:
                    _close(v, z["train_stat:" + k])


def test_losses():
    z, meta = G.load("loss")
    for tag, (B, H, W) in {"full": (2, 224, 384), "small": (3, 40, 56)}.items():
        s = synth.uniform("loss_s_" + tag, (B, H, W), meta["seed"], 0.01, 0.99)
        g = synth.gt_map(B, H, W, meta["seed"])
        for fn in ("kldiv", "cc", "similarity"):
            si = s.clone().requires_grad_(True)
            v = getattr(O, fn)(si, g)
            v.backward()
            _close(v.detach(), z["%s_%s" % (tag, fn)])
            if tag == "small":
                _close(si.grad, z["small_%s_grad" % fn])
        assert abs(float(O.kldiv(s, g.double())) - float(z["%s_kldiv_gt64" % tag])) < 1e-12


def test_decoder8():
    z, meta = G.load("decoder8")
    m = O.DecoderConvUp8()
    sd = G.state_dict_for(m, meta["seed"], z, meta)
    m.load_state_dict(sd)
    ys = [synth.normal("dec_y%d" % i, tuple(s), meta["seed"]).abs().requires_grad_(True) for i, s in enumerate(meta["shapes"])]
    o = m(*ys)
    proj = synth.normal("dec_proj", tuple(o.shape), meta["seed"])
    (o * proj).sum().backward()
    _close(o.detach(), z["out"])
    _close(ys[0].grad, z["gy0"])
    _close(ys[1].grad, z["gy1"])
    for i in (2, 3):
        _close(ys[i].grad.reshape(-1)[:4096], z["gy%d_head" % i])


@pytest.mark.parametrize("tag", ["8x96x192", "8x128x192"])
def test_e2e_small(tag):
    z, meta = G.load("e2e_" + tag)
    m = O.VideoSaliencyModel(num_clips=meta["clips"]).eval()
    m.load_state_dict(G.state_dict_for(m, meta["weight_seed"], z, meta))
    x = synth.clip(1, meta["clips"], meta["H"], meta["W"], meta["clip_seed"]).permute(0, 2, 1, 3, 4)
    with torch.no_grad():
        y = m(x)
    _close(y, z["y"])
    assert int(y.reshape(-1).argmax()) == meta["argmax"]
    assert meta["top2_gap"] > 1e-3


def test_state_dict_keys_match_reference_layout():
    """470 keys for ViNet-32 (checkpoint compatibility)."""
    m = O.VideoSaliencyModel(num_clips=32)
    keys = list(m.state_dict().keys())
    assert len(keys) == 470
    assert "backbone.base1.0.conv_s.weight" in keys
    assert "backbone.base1.0.bn_s.running_mean" in keys
    assert "decoder.convtsp4.8.bias" in keys
    assert sum(p.numel() for p in m.parameters()) == 31099585
    z, meta = G.load("train_step")
    assert json.loads(str(z["state_names"])) == list(O.VideoSaliencyModel(num_clips=8).state_dict().keys())


@pytest.mark.slow
def test_train_step():
    z, meta = G.load("train_step")
    B, T, H, W = meta["B"], meta["T"], meta["H"], meta["W"]
    x = synth.clip(B, T, H, W, meta["seed"]).permute(0, 2, 1, 3, 4)
    gt = synth.gt_map(B, H, W, meta["seed"])
    m = O.VideoSaliencyModel(num_clips=8)
    m.load_state_dict(G.state_dict_for(m, meta["seed"], z, meta))
    m.train()
    opt = torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr=meta["lr"])
    opt.zero_grad()
    pred = m(x)
    loss0 = O.kldiv(pred, gt)
    loss0.backward()
    opt.step()
    _close(pred.detach(), z["pred"])
    _close(loss0.detach(), z["loss0"])
    names = json.loads(str(z["grad_names"]))
    gs = np.array([float(dict(m.named_parameters())[k].grad.double().sum()) for k in names])
    np.testing.assert_allclose(gs, z["grad_sum"], rtol=1e-4, atol=1e-6)
    with torch.no_grad():
        loss1 = O.kldiv(m(x), gt)
    _close(loss1, z["loss1"], 1e-5)
