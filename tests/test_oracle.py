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


COMPACT_BLOCKS = {"mixed_3c": lambda: O.Mixed_3c(), "mixed_4b": lambda: O.Mixed_4b(), "mixed_4f": lambda: O.Mixed_4f(),
                  "mixed_5c": lambda: O.Mixed_5c(), "mixed_5b": lambda: O.Mixed_5b(), "soundnet": lambda: O.SoundNet()}


@pytest.mark.parametrize("name", list(COMPACT_BLOCKS))
def test_compact_inception_blocks(name):
    """the oracle against the reference's compact block goldens (strided samples + L2 norms: make_goldens.py _compact): Mixed_3c / 4b /
    4f / 5c (round 6) and 5b (round 4), eval and train mode, outputs, input gradient and every parameter gradient"""
    z, meta = G.load("block_" + name)
    m = COMPACT_BLOCKS[name]()
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
        got = {"y": y.detach(), "gx": xi.grad}
        got.update({"g:" + k: p.grad for k, p in m.named_parameters() if p.grad is not None})
        for k, t in got.items():
            key = mode + "_" + k
            stride = int(z[key + "#stride"])
            np.testing.assert_allclose(t.reshape(-1)[::stride].numpy(), z[key], rtol=1e-5, atol=1e-6)
            np.testing.assert_allclose(float(t.double().norm()), float(z[key + "#norm"]), rtol=1e-6)


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


@pytest.mark.parametrize("clips", [8, 16, 32, 48])
def test_decoders(clips):
    """DecoderConvUp8 / 16 / 48 (model.py:375-435, 313-373, 437-498) and the 32-frame DecoderConvUp (model.py:251-311, round 6) against
    the reference's outputs and gradients"""
    z, meta = G.load("decoder%d" % clips)
    m = {8: O.DecoderConvUp8, 16: O.DecoderConvUp16, 48: O.DecoderConvUp48, 32: O.DecoderConvUp}[clips]()
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


def test_loss_func_matches_reference():
    """utils.loss_func / get_loss (utils.py:9-39): flag / coefficient combinations and the 4-D multi-frame path"""
    from tests.golden_args import LossArgs
    z, meta = G.load("loss_func")
    s3 = synth.uniform("lf_s3", (2, 40, 56), meta["seed"], 0.01, 0.99)
    g3 = synth.gt_map(2, 40, 56, meta["seed"])
    s4 = synth.uniform("lf_s4", (2, 3, 24, 40), meta["seed"], 0.01, 0.99)
    g4 = synth.gt_map(6, 24, 40, meta["seed"] + 1).reshape(2, 3, 24, 40)
    for name, flags in meta["combos"].items():
        for tag, (s_, g_) in {"3d": (s3, g3), "4d": (s4, g4)}.items():
            si = s_.clone().requires_grad_(True)
            v = O.loss_func(si, g_, LossArgs(**flags))
            v.sum().backward()
            _close(v.detach(), z["%s_%s" % (name, tag)])
            _close(si.grad, z["%s_%s_grad" % (name, tag)])


def test_64_frame_variant_is_labelled_build_defined():
    """BASELINE config 5: no reference decoder exists for 64 frames; the restatement says so and ends at T = 1"""
    assert "NO REFERENCE PARITY" in open(O.__file__).read()
    m = O.VideoSaliencyModel(num_clips=64).eval()
    with torch.no_grad():
        y = m(synth.clip(1, 64, 32, 32, 1).permute(0, 2, 1, 3, 4))
    assert y.shape == (1, 32, 32)


@pytest.mark.parametrize("tag", ["8x96x192", "8x128x192", "16x64x96", "48x64x96"])
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


@pytest.mark.slow
def test_training_trajectory_first_steps():
    """tests/golden/train_trajectory.npz is the REFERENCE's own training loop (train.py:208-217 semantics, Adam lr 1e-4, 48 steps over
    a rotation of 4 synthetic batches; the generator refuses to write it unless the oracle reproduces all 48 losses and the final
    state).  Here, on every CPU run: the oracle's first 6 steps land on the stored losses (the first epoch over the rotation and the
    start of the second: the Adam moments, BatchNorm statistics and batch order are all in play), the stored curve descends by the
    factor the generator printed, and every stored quantity is finite."""
    from tests import model_cases as MC
    z, meta = G.load("train_trajectory")
    ref = z["losses"]
    assert meta["steps"] == len(ref) == 48 and meta["batches"] == 4 and np.isfinite(ref).all()
    assert ref[-4:].mean() < 0.2 * ref[:4].mean(), "the fixture is meant to be a trajectory that trains"
    # the chaos yardstick stored with it: 8 perturbed-weight runs of the reference + one fp64 run; they agree early and spread later
    ens, f64 = z["ensemble_losses"], z["fp64_losses"]
    assert ens.shape == (8, 48) and f64.shape == (48,) and np.isfinite(ens).all() and np.isfinite(f64).all()
    allm = np.vstack([ens, ref[None], f64[None]])
    assert float((allm[:, :6].std(0) / allm[:, :6].mean(0)).max()) < 2e-3, "members must still agree over steps 0..5 (the tight gate's window)"
    assert float((allm[:, 16:].std(0) / allm[:, 16:].mean(0)).mean()) > 0.1, "... and be O(0.1..1) apart later: that is why the late gate is statistical"
    m = O.VideoSaliencyModel(num_clips=meta["T"])
    m.load_state_dict(G.state_dict_for(m, meta["seed"], z, meta))
    got = MC.trajectory_run(m, O.kldiv, lambda ps: torch.optim.Adam(ps, lr=meta["lr"]), MC.trajectory_batches(meta), 6, torch.device("cpu"))
    np.testing.assert_allclose(got, ref[:6], rtol=2e-5, atol=0)


# ---- oracle/postproc_cpu.py: cv2 / torchvision are absent here ("parity unpinned"); the restatement is checked against two
# independent implementations of the same published algorithms and against closed-form properties -------------------------
def test_postproc_resize_matches_half_pixel_bilinear():
    from oracle import postproc_cpu as P
    rng = np.random.default_rng(0)
    for (H, W, oh, ow) in [(224, 384, 360, 640), (224, 384, 100, 150), (7, 9, 20, 31), (8, 8, 3, 5), (1, 20, 1, 33)]:
        s = rng.random((2, H, W), dtype=np.float32)
        a = P.resize_linear(s, oh, ow)
        b = torch.nn.functional.interpolate(torch.from_numpy(s)[None], size=(oh, ow), mode="bilinear", align_corners=False)[0].numpy()
        assert a.shape == (2, oh, ow) and a.dtype == np.float32
        assert np.abs(a - b).max() < 1e-4          # torch forms the source index from a float32 scale, cv2 from a double
    s = rng.random((3, 17, 23), dtype=np.float32)
    assert np.array_equal(P.resize_linear(s, 17, 23), s)          # same size: weights (1, 0), the identity
    assert np.allclose(P.resize_linear(np.full((5, 6), 0.375, np.float32), 11, 13), 0.375, atol=1e-7)


def test_postproc_gaussian_blur_matches_separable_mirror_correlation():
    import scipy.ndimage as ndi
    from oracle import postproc_cpu as P
    k = P.gaussian_kernel()
    assert k.dtype == np.float32 and k.shape == (11,) and abs(float(k.sum()) - 1.0) < 2e-7 and np.array_equal(k, k[::-1])
    assert abs(float(k[5]) / float(k[4]) - np.exp(1.0 / 8.0)) < 1e-6          # sigma = 2: k5 / k4 = exp(1 / (2 sigma^2))
    rng = np.random.default_rng(1)
    for (H, W) in [(360, 640), (7, 9), (3, 4), (1, 20), (11, 11)]:           # incl. maps smaller than the radius
        s = rng.random((2, H, W), dtype=np.float32)
        a = P.gaussian_blur11(s)
        k64 = k.astype(np.float64)
        b = ndi.correlate1d(ndi.correlate1d(s.astype(np.float64), k64, axis=2, mode="mirror"), k64, axis=1, mode="mirror")
        assert np.abs(a - b).max() < 1e-6
    assert np.allclose(P.gaussian_blur11(np.full((9, 30), 0.5, np.float32)), 0.5, atol=2e-7)
    assert list(P.reflect101(np.array([-6, -1, 0, 3, 4, 9]), 4)) == [0, 1, 0, 3, 2, 3]


def test_postproc_normalize_u8_known_answers():
    from oracle import postproc_cpu as P
    ramp = (np.arange(256, dtype=np.float32) / 255.0).reshape(16, 16)
    u = P.normalize_u8(ramp)
    # x*255 + 0.5 is then ROUNDED (half to even), not truncated (utils.py:71): i + 0.5 - eps -> i or i + 1
    assert u.dtype == np.uint8 and u[0, 0] == 0 and u[-1, -1] == 255 and np.all(np.diff(u.reshape(-1).astype(int)) >= 0)
    assert np.array_equal(P.normalize_u8(np.full((4, 5), 0.7, np.float32)), np.zeros((4, 5), np.uint8))   # 0 / 1e-5
    two = np.array([[0.0, 1.0]], np.float32)
    assert list(P.normalize_u8(two)[0]) == [0, 255]            # 1 / (1 + 1e-5) * 255 + 0.5 = 255.497 -> 255
    b = P.normalize_u8(np.stack([ramp, 1.0 - ramp]))           # a batch is normalised map by map
    assert np.array_equal(b[0], u) and b[1][0, 0] == 255


# ---- oracle/preproc_cpu.py: the frame resize is PINNED against the real Pillow resampler (present in this image) ----------
PIL_SHAPES = [(360, 640, 224, 384), (100, 150, 224, 384), (224, 384, 224, 384), (720, 1280, 224, 384), (37, 53, 20, 31),
              (5, 7, 224, 384), (300, 384, 224, 384), (224, 500, 224, 384), (1, 9, 4, 3)]


@pytest.mark.parametrize("shape", PIL_SHAPES, ids=["%dx%d_to_%dx%d" % s for s in PIL_SHAPES])
def test_preproc_resize_equals_pillow_byte_for_byte(shape):
    from PIL import Image
    from oracle import preproc_cpu as Q
    H, W, oh, ow = shape
    rng = np.random.default_rng(H * 1000 + W)
    img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    ours = Q.pil_resize_bilinear(img, oh, ow)
    pil = np.asarray(Image.fromarray(img).resize((ow, oh), Image.BILINEAR))
    assert np.array_equal(ours, pil)
    # img_transform as a whole: ToTensor + Normalize are plain float32 arithmetic on those bytes
    x = Q.frames_preprocess(img[None], oh, ow)
    want = (torch.from_numpy(pil.copy()).permute(2, 0, 1).float().div(255) - torch.tensor(Q.MEAN)[:, None, None]) / torch.tensor(Q.STD)[:, None, None]
    assert x.shape == (1, 3, oh, ow) and np.array_equal(x[0], want.numpy())


def test_preproc_gt_maps():
    from oracle import preproc_cpu as Q
    rng = np.random.default_rng(5)
    g = rng.integers(0, 256, (2, 36, 64), dtype=np.uint8)
    g[1] = (g[1] > 250).astype(np.uint8)                      # a 0/1 map is NOT divided (dataloader.py:294: max > 1.0)
    same = Q.gt_preprocess(g)
    assert np.array_equal(same[0], (g[0].astype(np.float64) / 255.0).astype(np.float32)) and np.array_equal(same[1], g[1].astype(np.float32))
    r = Q.gt_preprocess(g, 22, 38)
    ref = torch.nn.functional.interpolate(torch.from_numpy(g[:1].astype(np.float64))[None], size=(22, 38), mode="bilinear", align_corners=False)[0, 0].numpy() / 255.0
    assert r.shape == (2, 22, 38) and r.dtype == np.float32 and np.abs(r[0] - ref).max() < 1e-4


def test_preproc_audio_excerpt_and_frame_bounds():
    from oracle import preproc_cpu as Q
    for M in (1, 2, 7, 1470, 70560):
        assert np.allclose(Q.hanning_np118(M), np.hanning(M), rtol=0, atol=1e-12)       # the two formulas are algebraically equal
        assert np.abs(Q.hanning_np118(M).astype(np.float32) - np.hanning(M).astype(np.float32)).max() <= 6e-8
    rng = np.random.default_rng(3)
    wav = (rng.standard_normal(200000) * 2 ** -8).astype(np.float32)
    for (s, e) in [(0, 999), (5000, 5000 + 70559), (1234, 1234 + 47040), (199000, 260000), (10, 10)]:
        out = Q.audio_excerpt(wav, s, e)
        M = min(e + 1, wav.shape[0]) - s
        lo = 70560 // 2 - M // 2
        assert out.shape == (70560,) and np.all(out[:lo] == 0) and np.all(out[lo + M:] == 0)
        assert np.array_equal(out[lo:lo + M], Q.hanning_np118(M).astype(np.float32) * wav[s:s + M])
    st, en = Q.audio_frame_bounds(5, 25.0, 22050, 4000)          # 882 samples per frame, clipped by the waveform's end
    assert list(st) == [0, 0, 441, 1323, 2205, 3087] and list(en) == [0, 441, 1323, 2205, 3087, 3969]
    st, en = Q.audio_frame_bounds(6, 25.0, 22050, 4000)
    assert en[6] == 4000


def test_io_pipeline_oracle_matches_committed_fixtures():
    """tests/golden/io_frames.npz holds outputs of the real Pillow resampler (+ ToTensor / Normalize); io_maps.npz regression
    vectors of the post-processing restatement (tests/golden/make_io_goldens.py)"""
    from oracle import postproc_cpu as P
    from oracle import preproc_cpu as Q
    z, meta = G.load("io_frames")
    for name, (n, h, w, oh, ow) in meta["cases"].items():
        assert np.array_equal(Q.pil_resize_bilinear(z[name + "_in"], oh, ow), z[name + "_resized"])
        assert np.array_equal(Q.frames_preprocess(z[name + "_in"], oh, ow), z[name + "_out"])
    assert np.array_equal(Q.gt_preprocess(z["gt_in"], *meta["gt_train_size"]), z["gt_train"])
    assert np.array_equal(Q.gt_preprocess(z["gt_in"]), z["gt_val"])
    m, _ = G.load("io_maps")
    for name, (oh, ow) in {"up_45x80": (45, 80), "same": (28, 48), "down_9x13": (9, 13)}.items():
        b = P.resize_blur(m["src"], oh, ow)
        assert np.array_equal(b, m[name + "_blur"]) and np.array_equal(P.normalize_u8(b), m[name + "_u8"])
