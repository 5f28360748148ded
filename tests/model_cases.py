"""Model-level parity cases shared by the CPU host-logic tests (C ABI served by
tests/abi_emulator.py) and the GPU parity tests (real libvinet_hip.so).  Every
case compares vinet_amd against golden vectors captured from the reference."""
import json
import os

import numpy as np
import torch

from oracle import vinet_cpu as O
from tests import goldens as G
from vinet_amd import engine as E
from vinet_amd import synth


def close(a, b, tol, what=""):
    a, b = torch.as_tensor(a).detach().cpu(), torch.as_tensor(b).detach().cpu()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    d = float((a.double() - b.double()).abs().max())
    assert d <= tol, "%s: max abs diff %g > %g" % (what, d, tol)
    return d


def relerr(a, b):
    a, b = torch.as_tensor(a).detach().cpu().double(), torch.as_tensor(b).detach().cpu().double()
    assert a.shape == b.shape, (a.shape, b.shape)
    return float((a - b).norm() / (b.norm() + 1e-30))


def blocks():
    from vinet_amd import model_utils as MU
    return {
        "mixed_5b": lambda: MU.Mixed_5b(),
        "mixed_3c": lambda: MU.Mixed_3c(),
        "mixed_4b": lambda: MU.Mixed_4b(),
        "mixed_4f": lambda: MU.Mixed_4f(),
        "mixed_5c": lambda: MU.Mixed_5c(),
        "soundnet": lambda: __import__("vinet_amd.model", fromlist=["SoundNet"]).SoundNet(),
        "basic_16_32": lambda: MU.BasicConv3d(16, 32, 1, 1),
        "sep_16_32_k3": lambda: MU.SepConv3d(16, 32, 3, 1, 1),
        "sep_3_64_k7s2": lambda: MU.SepConv3d(3, 64, 7, 2, 3),
        "mixed_3b": lambda: MU.Mixed_3b(),
    }


def block_case_bf16(name, mode, dev, ftol=2e-2, gtol=0.12, note=None):
    """bf16 path: relative L2 error per tensor.  Elementwise max-abs is not meaningful
    for bf16 gradients: a pre-activation that rounds across 0, or a max-pool argmax
    that flips between two near-equal inputs, moves one gradient element by O(1)."""
    z, meta = G.load("block_" + name)
    m = blocks()[name]()
    m.load_state_dict(synth.synth_state_dict(m.state_dict(), meta["seed"]))
    m = m.to(dev)
    m.train(mode == "train")
    x = synth.normal("x_" + name, tuple(meta["in_shape"]), meta["seed"]).to(dev).requires_grad_(True)
    y = m(x)
    errs = {"y": relerr(y, z[mode + "_y"])}
    proj = synth.normal("proj_" + name, tuple(y.shape), meta["seed"]).to(dev)
    (y * proj).sum().backward()
    errs["gx"] = relerr(x.grad, z[mode + "_gx"])
    for k, p in m.named_parameters():
        errs["g:" + k] = relerr(p.grad, z[mode + "_g:" + k])
    if note is not None:
        note(errs)
    # a ReLU whose bf16 pre-activation rounds across 0 flips one gradient element by O(1):
    # relative L2 error ~ sqrt(flip fraction) ~ several % per layer
    assert errs["y"] <= ftol, "%s y rel err %g" % (name, errs["y"])
    assert errs["gx"] <= gtol, "%s gx rel err %g" % (name, errs["gx"])
    for k, e in errs.items():
        assert e <= gtol, "%s %s rel err %g" % (name, k, e)
    return errs


def block_case(name, mode, dev, ftol=2e-5, gtol=2e-4):
    z, meta = G.load("block_" + name)
    m = blocks()[name]()
    m.load_state_dict(synth.synth_state_dict(m.state_dict(), meta["seed"]))
    m = m.to(dev)
    m.train(mode == "train")
    x = synth.normal("x_" + name, tuple(meta["in_shape"]), meta["seed"]).to(dev).requires_grad_(True)
    y = m(x)
    close(y, z[mode + "_y"], ftol * max(1.0, float(np.abs(z[mode + "_y"]).max())), name + " y")
    proj = synth.normal("proj_" + name, tuple(y.shape), meta["seed"]).to(dev)
    (y * proj).sum().backward()
    close(x.grad, z[mode + "_gx"], gtol * max(1.0, float(np.abs(z[mode + "_gx"]).max())), name + " gx")
    for k, p in m.named_parameters():
        ref = torch.as_tensor(z[mode + "_g:" + k])
        close(p.grad, ref, gtol * max(1.0, float(ref.abs().max())), name + " grad " + k)
    if mode == "train":
        for k, v in m.state_dict().items():
            if "running" in k:
                close(v, z["train_stat:" + k], max(1e-5, ftol), name + " " + k)


def block_case_compact(name, mode, dev, sample_tol, norm_tol, l2=False, zero_tol=1e-4):
    """block goldens stored as strided samples + L2 norms (tests/golden/make_goldens.py::_compact: the M = 336-voxel Inception
    blocks have 1.3 M weights).  Per tensor: the same strided sample of ours against the reference's, max abs error relative to
    the sample's largest magnitude <= sample_tol, and the L2 norm within norm_tol (relative).
    l2 (bf16): the sample is held to a relative L2 error <= sample_tol instead, like every other bf16 block test -- a bf16
    pre-activation that rounds across zero flips a ReLU gate (or a max-pool argmax) and moves single gradient elements by O(1)
    of the largest one (0.37-0.42 measured on gx here), which an elementwise bound cannot tell from a bug."""
    z, meta = G.load("block_" + name)
    m = blocks()[name]()
    m.load_state_dict(synth.synth_state_dict(m.state_dict(), meta["seed"]))
    m = m.to(dev)
    m.train(mode == "train")
    # (the waveform is an input of the net, not an activation: vinet_amd's SoundNet takes no gradient for it -- engine.unfold1d_forward)
    wants_gx = name != "soundnet"
    x = synth.normal("x_" + name, tuple(meta["in_shape"]), meta["seed"]).to(dev).requires_grad_(wants_gx)
    y = m(x)
    proj = synth.normal("proj_" + name, tuple(y.shape), meta["seed"]).to(dev)
    (y * proj).sum().backward()
    got = {"y": y}
    if wants_gx:
        got["gx"] = x.grad
    got.update({"g:" + k: p.grad for k, p in m.named_parameters() if p.grad is not None})
    # (parameters the forward never uses -- SoundNet's two classification heads -- have no gradient on either side)
    assert {mode + "_" + k for k in got} | ({mode + "_gx"} if not wants_gx else set()) == {
        f for f in z.files if f.startswith(mode + "_") and "#" not in f and not f.startswith(mode + "_stat:")}, "gradient set differs from the reference's"
    errs = {}
    # a conv bias in front of a training-mode BatchNorm (SoundNet: model.py:757-790) has a gradient that is ZERO in exact arithmetic
    # (the BatchNorm subtracts the mean): the reference holds round-off there (norms 6e-6 ... 7e-4 beside 2e3 for the weights).
    # Such tensors -- reference norm below 1e-5 of the largest parameter gradient's -- must be equally negligible on our side.
    gmax = max([float(z[f]) for f in z.files if f.startswith(mode + "_g:") and f.endswith("#norm")] or [0.0])
    for k, t in got.items():
        key = mode + "_" + k
        if k.startswith("g:") and float(z[key + "#norm"]) < 1e-5 * gmax:
            mine_n = float(t.detach().double().norm())
            # (ours = sum over M voxels of a BatchNorm-backward output whose exact sum is zero: M x the round-off of the mean term)
            assert mine_n <= zero_tol * gmax, "%s %s %s: reference gradient is numerically zero (%g), ours is %g (largest gradient norm %g)" % (
                name, mode, k, float(z[key + "#norm"]), mine_n, gmax)
            continue
        ref = torch.as_tensor(z[key])
        stride = int(z[key + "#stride"])
        mine = t.detach().reshape(-1)[::stride].float().cpu()
        scale = max(1e-6, float(ref.abs().max()))
        es = float((mine - ref).abs().max()) / scale
        if l2:
            es = float((mine.double() - ref.double()).norm() / max(1e-30, float(ref.double().norm())))
        en = abs(float(t.detach().double().norm()) - float(z[key + "#norm"])) / max(1e-12, float(z[key + "#norm"]))
        errs[k] = (es, en)
        assert es <= sample_tol, "%s %s %s: sample error %g (%s) > %g" % (name, mode, k, es, "relative L2" if l2 else "relative to the largest sample", sample_tol)
        assert en <= norm_tol, "%s %s %s: L2 norm off by %g > %g" % (name, mode, k, en, norm_tol)
    return errs


def losses_case(dev, vtol=2e-6, gtol=1e-7):
    from vinet_amd import loss as VL
    z, meta = G.load("loss")
    for tag, (B, H, W) in {"full": (2, 224, 384), "small": (3, 40, 56)}.items():
        s = synth.uniform("loss_s_" + tag, (B, H, W), meta["seed"], 0.01, 0.99).to(dev)
        g = synth.gt_map(B, H, W, meta["seed"]).to(dev)
        for fn in ("kldiv", "cc", "similarity"):
            si = s.clone().requires_grad_(True)
            v = getattr(VL, fn)(si, g)
            v.backward()
            close(v, z["%s_%s" % (tag, fn)], vtol, "%s %s" % (tag, fn))
            if tag == "small":
                close(si.grad, z["small_%s_grad" % fn], gtol, "%s grad" % fn)
            else:
                st = z["full_%s_gradstats" % fn]
                gd = si.grad.double().cpu()
                assert abs(float((gd * gd).sum()) - st[1]) <= 1e-4 * st[1]
        v64 = VL.kldiv(s, g.double())
        assert v64.dtype == torch.float64
        assert abs(float(v64) - float(z["%s_kldiv_gt64" % tag])) < 1e-6
        fix = (g > 0.5 * g.amax(dim=(1, 2), keepdim=True)).float()
        close(VL.nss(s, fix), z["%s_nss" % tag], 5e-6, "%s nss" % tag)


def decoder8_case(dev, ftol=2e-5, gtol=3e-4):
    return decoder_case(8, dev, ftol, gtol)


def decoder_case(clips, dev, ftol=2e-5, gtol=3e-4):
    """DecoderConvUp{8,16,48} alone against the reference's outputs, input gradients and parameter gradients
    (model.py:375-435, 313-373, 437-498; the 48-frame tail has a biased (3,1,1) conv, model.py:466)."""
    from vinet_amd import model as VM
    z, meta = G.load("decoder%d" % clips)
    m = {8: VM.DecoderConvUp8, 16: VM.DecoderConvUp16, 48: VM.DecoderConvUp48, 32: VM.DecoderConvUp}[clips]()
    m.load_state_dict(G.state_dict_for(m, meta["seed"], z, meta))
    m = m.to(dev)
    ys = [synth.normal("dec_y%d" % i, tuple(s), meta["seed"]).abs().to(dev).requires_grad_(True) for i, s in enumerate(meta["shapes"])]
    o = m(*ys)
    close(o, z["out"], ftol, "decoder out")
    proj = synth.normal("dec_proj", tuple(o.shape), meta["seed"]).to(dev)
    (o * proj).sum().backward()
    for i in (0, 1):
        ref = z["gy%d" % i]
        close(ys[i].grad, ref, gtol * max(1.0, float(np.abs(ref).max())), "gy%d" % i)
    for i in (2, 3):
        ref = z["gy%d_head" % i]
        close(ys[i].grad.reshape(-1)[:4096], ref, gtol * max(1.0, float(np.abs(ref).max())), "gy%d" % i)
    for k, p in m.named_parameters():
        ref = z["gp_head:" + k]
        close(p.grad.reshape(-1)[:2048], ref, gtol * max(1.0, float(np.abs(ref).max())), "decoder grad " + k)


def loss_func_case(dev, vtol=3e-6, gtol=2e-7):
    """utils.loss_func / get_loss (utils.py:9-39) against values and gradients captured from the reference: default flags,
    kldiv + cc + sim with the reference's default coefficients (train.py:36-37), other coefficients, cc alone; 3-D maps and
    the 4-D multi-frame path (utils.py:27-37)."""
    from tests.golden_args import LossArgs
    from vinet_amd import utils as VU
    z, meta = G.load("loss_func")
    s3 = synth.uniform("lf_s3", (2, 40, 56), meta["seed"], 0.01, 0.99)
    g3 = synth.gt_map(2, 40, 56, meta["seed"])
    s4 = synth.uniform("lf_s4", (2, 3, 24, 40), meta["seed"], 0.01, 0.99)
    g4 = synth.gt_map(6, 24, 40, meta["seed"] + 1).reshape(2, 3, 24, 40)
    for name, flags in meta["combos"].items():
        a = LossArgs(**flags)
        for tag, (s_, g_) in {"3d": (s3, g3), "4d": (s4, g4)}.items():
            si = s_.clone().to(dev).requires_grad_(True)
            v = VU.loss_func(si, g_.to(dev), a)
            assert tuple(v.shape) == (1,)
            v.sum().backward()
            close(v, z["%s_%s" % (name, tag)], vtol, "loss_func %s %s" % (name, tag))
            close(si.grad, z["%s_%s_grad" % (name, tag)], gtol, "loss_func grad %s %s" % (name, tag))


def e2e_bf16_case(tag, dev, tol=2.5e-2, cc_min=0.999, topk=5):
    """the throughput (bf16) path against the reference's fp32 map: max abs error, linear correlation with the
    golden map, and the golden fixation (argmax) must stay among the bf16 map's top-k pixels."""
    from vinet_amd import model as VM
    z, meta = G.load("e2e_" + tag)
    m = VM.VideoSaliencyModel(num_clips=meta["clips"]).eval()
    m.load_state_dict(G.state_dict_for(m, meta["weight_seed"], z, meta))
    m = m.to(dev)
    x = synth.clip(1, meta["clips"], meta["H"], meta["W"], meta["clip_seed"]).to(dev).permute(0, 2, 1, 3, 4)
    with torch.no_grad():
        y = m(x).cpu()
    ref = torch.as_tensor(z["y"])
    d = close(y, ref, tol, "e2e bf16 " + tag)
    a, b = y.double().reshape(-1), ref.double().reshape(-1)
    a, b = a - a.mean(), b - b.mean()
    cc = float((a * b).sum() / (a.norm() * b.norm()))
    top = torch.topk(y.reshape(-1), topk).indices.tolist()
    am = int(y.reshape(-1).argmax())
    info = dict(max_abs=d, cc=cc, argmax_matches=(am == meta["argmax"]), golden_argmax_rank=(top.index(meta["argmax"]) if meta["argmax"] in top else None),
                top2_gap=meta["top2_gap"])
    assert cc >= cc_min, "bf16 map decorrelated from the reference: cc %.6f" % cc
    assert meta["argmax"] in top, "the reference's fixation is not among the bf16 map's top-%d pixels" % topk
    return info


def e2e_case(tag, dev, tol=1e-4, argmax=True):
    """forward parity on the float map (north_star: 1e-3 abs) and bit-exact argmax."""
    from vinet_amd import model as VM
    z, meta = G.load("e2e_" + tag)
    m = VM.VideoSaliencyModel(num_clips=meta["clips"]).eval()
    m.load_state_dict(G.state_dict_for(m, meta["weight_seed"], z, meta))
    m = m.to(dev)
    x = synth.clip(1, meta["clips"], meta["H"], meta["W"], meta["clip_seed"]).to(dev).permute(0, 2, 1, 3, 4)
    with torch.no_grad():
        y = m(x)
    assert y.shape == (1, meta["H"], meta["W"])
    d = close(y, z["y"], tol, "e2e " + tag)
    if argmax:
        assert int(y.reshape(-1).argmax()) == meta["argmax"], "argmax differs (top-2 gap %g)" % meta["top2_gap"]
    return d, meta


def train_step_case(dev, make_optimizer=None, pred_tol=2e-5, loss_tol=1e-5, grad_factor=3.0, grad_floor=2e-3, worst_max=0.15,
                    global_tol=None, sq_rtol=8e-2, global_factor=None, fixture="train_step", min_drop=0.2):
    from vinet_amd import loss as VL
    from vinet_amd import model as VM
    from vinet_amd import optim as VO
    z, meta = G.load(fixture)
    B, T, H, W = meta["B"], meta["T"], meta["H"], meta["W"]
    x = synth.clip(B, T, H, W, meta["seed"]).permute(0, 2, 1, 3, 4)
    gt = synth.gt_map(B, H, W, meta["seed"])
    m = VM.VideoSaliencyModel(num_clips=8)
    m.load_state_dict(G.state_dict_for(m, meta["seed"], z, meta))
    m = m.to(dev).train()
    xd, gd = x.to(dev), gt.to(dev)
    opt = (make_optimizer or VO.Adam)([p for p in m.parameters() if p.requires_grad], lr=meta["lr"])
    opt.zero_grad()
    pred = m(xd)
    loss0 = VL.kldiv(pred, gd)
    loss0.backward()
    close(pred, z["pred"], pred_tol, "train pred")
    close(loss0, z["loss0"], loss_tol, "train loss0")
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
    worst, table = 0.0, []
    for k, p in params.items():
        t = truth[k]
        e_ref = float((ref32[k] - t).norm() / (t.norm() + 1e-30))
        e_me = float((p.grad.double().cpu() - t).norm() / (t.norm() + 1e-30))
        table.append((e_me, e_ref, k))
        # floor 2e-3: a forward that differs from torch's by fp32 round-off (1e-7 at the stem, 1e-5 after
        # the 12-sample BNs) flips a ~1e-5 fraction of ReLU gates, and the gradient's L2 error goes with
        # the square root of that fraction; a 2e-6 perturbation of the oracle's own input moves these
        # gradients by 1e-4..6e-4.  A wrong tap / pad / stride shows up at >= 1e-1.
        # (the split-bf16 form passes grad_factor 8, grad_floor 0.15 and a bound on the error of the WHOLE gradient vector: with
        #  B = 2 the deep BatchNorms see 12 samples per channel and single small parameters -- a BatchNorm bias of base3, the stem's
        #  weights, residuals of cancelling sums where the reference's OWN fp32 gradient sits 1.6-3.8 % from fp64 -- amplify a 2^-17
        #  operand error to 5-20 %; what a wrong kernel would do, an O(1) error on a large tensor, the global bound catches)
        worst = max(worst, e_me)
    table.sort(reverse=True)
    train_step_case.last_table = table[:16]
    train_step_case.worst_ratio = max(((e_me - grad_floor) / max(e_ref, 1e-30), k) for e_me, e_ref, k in table)   # (reported; gated below)
    num = sum(float((params[k].grad.double().cpu() - truth[k]).pow(2).sum()) for k in params)
    nref = sum(float((ref32[k] - truth[k]).pow(2).sum()) for k in params)
    den = sum(float(truth[k].pow(2).sum()) for k in params)
    train_step_case.global_rel = (num / den) ** 0.5
    train_step_case.global_ref = (nref / den) ** 0.5
    for e_me, e_ref, k in table:
        assert e_me <= grad_factor * e_ref + grad_floor, "%s: rel err %.3e vs reference-fp32 %.3e" % (k, e_me, e_ref)
    assert worst < worst_max, table[:5]
    if global_tol is not None or global_factor is not None:
        # the whole gradient vector against fp64, as an absolute bound and / or relative to the reference's OWN fp32 error against
        # fp64 (the `train_step` fixture is ill-conditioned on purpose -- B = 2, 12 samples per channel in the deepest BatchNorms: the
        # reference's fp32 gradient vector sits 3 % from fp64, ours in exact fp32 3.4 %; an operand error of 2^-17 instead of 2^-24
        # grows the same way, to 15 % measured (direction, not scale: tools/dbg_split_grad.py))
        bound = min(b for b in (global_tol, None if global_factor is None else global_factor * train_step_case.global_ref + 1e-3) if b is not None)
        assert train_step_case.global_rel <= bound, "whole gradient vector: relative L2 error %.3e > %.3e (reference fp32 vs fp64: %.3e)" % (
            train_step_case.global_rel, bound, train_step_case.global_ref)
    names = json.loads(str(z["grad_names"]))
    gq = np.array([float((params[k].grad.double() ** 2).sum()) for k in names])
    np.testing.assert_allclose(gq, z["grad_sqsum"], rtol=sq_rtol, atol=1e-12)
    opt.step()
    with torch.no_grad():
        loss1 = VL.kldiv(m(xd), gd)      # train-mode forward: second running-stat update, as in the fixture
    # Adam's first step is sign descent (m/sqrt(v) = +-1): fp32-noise-level gradient
    # entries flip sign between implementations, so loss1 agrees to ~1e-3, not 1e-5
    close(loss1, z["loss1"], 3e-3, "train loss1")
    assert float(loss1) < float(loss0) - min_drop
    sd = m.state_dict()
    for k in [n for n in z.files if n.startswith("state:")]:
        # base4 statistics come from 12 samples/channel on weights that already took one
        # sign-descent step: only the large-M stem statistics are tight
        close(sd[k[6:]], z[k], 5e-5 if "base1" in k else 2e-3, k)
    return worst


def weight_shared_case(dev, dt, tol):
    """A block whose forward calls ONE BasicConv3d twice (a module used twice in a forward, which parallel.expect_reports
    supports): both tape nodes share the plan's persistent weight-gradient workspace, so the multi-job unpack must run one job
    per workspace (the workspace already holds the sum).  Truth: the same graph on torch autograd (oracle blocks, fp32 CPU)."""
    from vinet_amd import model_utils as MU

    class Twice(MU._Block):
        def __init__(self):
            super().__init__()
            self.c = MU.BasicConv3d(16, 16, (1, 3, 3), 1, (0, 1, 1))

        def _fwd(self, ctx, x, dst=None):
            return self.c._fwd(ctx, self.c._fwd(ctx, x))

    class TwiceRef(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.c = O.BasicConv3d(16, 16, (1, 3, 3), 1, (0, 1, 1))

        def forward(self, x):
            return self.c(self.c(x))

    m, r = Twice(), TwiceRef()
    sd = synth.synth_state_dict(r.state_dict(), 5)
    r.load_state_dict(sd)
    m.load_state_dict(sd)
    m.compute_dtype = dt
    m = m.to(dev).train()
    r.train()
    x = synth.normal("x_twice", (2, 16, 3, 12, 16), 5)
    xr = x.clone().requires_grad_(True)
    xg = x.to(dev).requires_grad_(True)
    proj = synth.normal("proj_twice", (2, 16, 3, 12, 16), 5)
    errs = []
    for _ in range(3):      # (the second and third passes find the persistent workspace handed back zeroed -- or not)
        for p in m.parameters():
            p.grad = None
        xg.grad = None
        (m(xg) * proj.to(dev)).sum().backward()
    (r(xr) * proj).sum().backward()
    errs.append(("gx", relerr(xg.grad, xr.grad)))
    for (k, p), (_, q) in zip(m.named_parameters(), r.named_parameters()):
        errs.append((k, relerr(p.grad, q.grad)))
    for k, e in errs:
        assert e <= tol, "weight-shared conv: %s rel err %g > %g" % (k, e, tol)
    return dict(errs)


def trajectory_batches(meta):
    """the fixture's rotation of synthetic batches, by recipe (tests/golden/make_goldens.py trajectory_batches)"""
    B, T, H, W, seed = meta["B"], meta["T"], meta["H"], meta["W"], meta["seed"]
    return [(synth.clip(B, T, H, W, seed + 100 * i).permute(0, 2, 1, 3, 4).contiguous(), synth.gt_map(B, H, W, seed + 100 * i))
            for i in range(meta["batches"])]


def trajectory_run(model, kldiv, make_optimizer, batches, steps, dev):
    """train.py:208-217: zero_grad -> model -> kldiv -> backward -> step, `steps` times over the rotation; the per-step losses"""
    model.train()
    opt = make_optimizer([p for p in model.parameters() if p.requires_grad])
    bs = [(x.to(dev), g.to(dev)) for x, g in batches]
    losses = []
    for i in range(steps):
        x, gt = bs[i % len(bs)]
        opt.zero_grad()
        loss = kldiv(model(x), gt)
        loss.backward()
        opt.step()
        losses.append(loss.detach())
    return np.array([float(l) for l in losses])


def trajectory_case(dev, dtype, steps=None):
    """the vinet_amd path against the REFERENCE's training trajectory (tests/golden/train_trajectory.npz): returns
    dict(losses, ref, rel = |l - ref| / ref per step, end_rel, eval_after, eval_after_ref, state_rel = worst per-tensor
    || p_end - p_ref_end ||-proxy from the stored checksums)"""
    from vinet_amd import loss as VL
    from vinet_amd import model as VM
    from vinet_amd import optim as VO
    z, meta = G.load("train_trajectory")
    steps = steps or meta["steps"]
    E.set_default_dtype(dtype)
    m = VM.VideoSaliencyModel(num_clips=meta["T"])
    m.load_state_dict(G.state_dict_for(m, meta["seed"], z, meta))
    m = m.to(dev)
    batches = trajectory_batches(meta)
    losses = trajectory_run(m, VL.kldiv, lambda ps: VO.Adam(ps, lr=meta["lr"]), batches, steps, dev)
    ref = z["losses"][:steps]
    out = dict(losses=losses, ref=ref, rel=np.abs(losses - ref) / np.abs(ref))
    if steps == meta["steps"]:
        m.eval()
        with torch.no_grad():
            out["eval_after"] = float(VL.kldiv(m(batches[0][0].to(dev)), batches[0][1].to(dev)))
        out["eval_after_ref"] = float(z["eval_loss_after"])
        # parameter space: per tensor |sum(p) - sum(p_ref)| against how far training moved that tensor (||p_end - p_0||, stored)
        names = json.loads(str(z["state_names"]))
        sd = {k: v.detach().double().cpu() for k, v in m.state_dict().items()}
        sq = np.array([float((sd[k] ** 2).sum()) for k in names])
        moved = z["state_delta_norm"]
        big = moved > 1e-3
        out["state_norm_rel"] = float(np.max(np.abs(np.sqrt(sq[big]) - np.sqrt(z["state_sqsum"][big])) / moved[big]))
    return out
