#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ FROM THE REAL REFERENCE.

Runs only in the build container (needs /root/reference).  It
  1. puts three empty stub packages (block, torchvision, cv2 -- imported but
     unused on this path) ahead of /root/reference on
     sys.path and imports the reference's model / model_utils / loss modules
     unmodified (no bytecode written);
  2. fills reference modules with the repo's procedural weights
     (vinet_amd/synth.py) and runs the golden cases on the reference;
  3. runs the same cases on oracle/vinet_cpu.py and REFUSES to write a fixture
     unless the oracle reproduces the reference (max abs diff recorded);
  4. writes inputs-by-recipe + expected outputs as .npz (data only).

Usage:  python tests/golden/make_goldens.py
"""
import json
import os
import sys
import tempfile
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import numpy as np
import torch

from vinet_amd import synth
from oracle import vinet_cpu as O

REF = "/root/reference"
TOL_ORACLE = 2e-6  # oracle-vs-reference; observed 0.0 on every case


def _install_stubs():
    d = tempfile.mkdtemp(prefix="vinet_stubs_")
    for pkg, body in {
        "block": "fusions = None\n",
        "cv2": "",
        "torchvision": "from . import models, transforms, utils\n",
    }.items():
        os.makedirs(os.path.join(d, pkg))
        with open(os.path.join(d, pkg, "__init__.py"), "w") as f:
            f.write(body)
    for sub, body in {"models": "vgg19 = None\n", "transforms": "", "utils": ""}.items():
        with open(os.path.join(d, "torchvision", sub + ".py"), "w") as f:
            f.write(body)
    sys.path[:0] = [d, REF]
    return d


def _import_reference():
    _install_stubs()
    import model as RM  # noqa
    import model_utils as RU  # noqa
    import loss as RL  # noqa
    return RM, RU, RL


def _maxdiff(a, b):
    return float((a.double() - b.double()).abs().max())


def _check(name, ref, ora, meta):
    d = _maxdiff(ref, ora)
    meta.setdefault("oracle_vs_reference_maxabs", {})[name] = d
    if not d <= TOL_ORACLE:
        raise SystemExit("ORACLE DOES NOT REPRODUCE THE REFERENCE on %s: %g" % (name, d))


def _np(t):
    return t.detach().cpu().numpy()


def _top2(map2d):
    flat = map2d.reshape(-1)
    v, i = torch.topk(flat, 2)
    return int(i[0]), float(v[0] - v[1])


# ----------------------------------------------------------------------------
def _compact(res, key, t, limit=16384):
    """large tensors are stored as a strided sample plus their L2 norm (the M = 336-voxel Inception blocks have 1.3 M weights)"""
    flat = t.detach().reshape(-1)
    stride = max(1, -(-flat.numel() // limit))
    res[key] = _np(flat[::stride])
    res[key + "#norm"] = np.array(float(flat.double().norm()))
    res[key + "#stride"] = np.array(stride)


def block_case(name, make_ref, make_ora, in_shape, seed, out, compact=False):
    """eval output; train output + updated running stats + input/weight grads."""
    meta = {}
    ref, ora = make_ref(), make_ora()
    sd = synth.synth_state_dict(ref.state_dict(), seed)
    x = synth.normal("x_" + name, in_shape, seed)
    r = synth.normal("r_" + name, (1,), seed)  # placeholder to fix key order
    res = {}
    for mode in ("eval", "train"):
        outs = []
        for m in (ref, ora):
            m.load_state_dict(sd)
            m.train(mode == "train")
            xi = x.clone().requires_grad_(True)
            y = m(xi)
            proj = synth.normal("proj_" + name, tuple(y.shape), seed)
            m.zero_grad()
            (y * proj).sum().backward()
            outs.append((y.detach(), xi.grad.detach(), {k: v.grad.detach().clone() for k, v in m.named_parameters() if v.grad is not None},   # (SoundNet's two heads get none)
                         {k: v.detach().clone() for k, v in m.state_dict().items()}))
        (y_r, gx_r, gp_r, st_r), (y_o, gx_o, gp_o, st_o) = outs
        _check("%s/%s/y" % (name, mode), y_r, y_o, meta)
        _check("%s/%s/gx" % (name, mode), gx_r, gx_o, meta)
        for k in gp_r:
            _check("%s/%s/g_%s" % (name, mode, k), gp_r[k], gp_o[k], meta)
        if compact:
            _compact(res, mode + "_y", y_r)
            _compact(res, mode + "_gx", gx_r)
            for k, v in gp_r.items():
                _compact(res, mode + "_g:" + k, v)
        else:
            res[mode + "_y"] = _np(y_r)
            res[mode + "_gx"] = _np(gx_r)
            for k, v in gp_r.items():
                res[mode + "_g:" + k] = _np(v)
        if mode == "train":
            for k, v in st_r.items():
                if "running" in k:
                    _check("%s/train/%s" % (name, k), v, st_o[k], meta)
                    res["train_stat:" + k] = _np(v)
    res["meta"] = np.array(json.dumps(dict(meta, seed=seed, in_shape=list(in_shape), name=name)))
    np.savez_compressed(os.path.join(out, "block_%s.npz" % name), **res)
    print("block", name, "ok", {k: v for k, v in list(meta["oracle_vs_reference_maxabs"].items())[:2]})


def _calibrated_sd(model, seed, run_logits):
    """procedural weights + head calibrated to logits ~ N(-3, 1)."""
    sd = synth.synth_state_dict(model.state_dict(), seed)
    prefix = "visual_model." if any(k.startswith("visual_model.") for k in sd) else ""
    wk = prefix + "decoder.convtsp4.%d.weight"
    last = max(int(k.split(".")[-2]) for k in sd if k.startswith(prefix + "decoder.convtsp4.") and k.endswith(".bias"))
    wk, bk = (prefix + "decoder.convtsp4.%d.weight" % last), (prefix + "decoder.convtsp4.%d.bias" % last)
    model.load_state_dict(sd)
    logits = run_logits(model)
    w, b = synth.calibrate_head(sd[wk], sd[bk], float(logits.mean()), float(logits.std()))
    sd[wk], sd[bk] = w, b
    return sd, wk, bk


def _logits_hook(model, decoder):
    """pre-sigmoid logits of one forward: input of the trailing nn.Sigmoid."""
    box = {}
    sig = decoder.convtsp4[-1]
    h = sig.register_forward_hook(lambda m, i, o: box.__setitem__("l", i[0].detach()))
    return box, h


def e2e_case(RM, clips, H, W, seed, out, tag):
    meta = {}
    ref = RM.VideoSaliencyModel(num_clips=clips).eval()
    ora = O.VideoSaliencyModel(num_clips=clips).eval()
    x = synth.clip(1, clips, H, W, seed).permute(0, 2, 1, 3, 4)

    def run_logits(m):
        box, h = _logits_hook(m, m.decoder)
        with torch.no_grad():
            m(x)
        h.remove()
        return box["l"]

    best = None
    for s in range(seed, seed + 3):  # keep the seed with the largest top-2 gap
        sd, wk, bk = _calibrated_sd(ref, s, run_logits)
        ref.load_state_dict(sd)
        with torch.no_grad():
            y = ref(x)
        idx, gap = _top2(y[0])
        if best is None or gap > best[3]:
            best = (s, sd, y, gap, idx, wk, bk)
    s, sd, y_r, gap, idx, wk, bk = best
    ora.load_state_dict(sd)
    with torch.no_grad():
        y_o = ora(x)
    _check("e2e/%s" % tag, y_r, y_o, meta)
    np.savez_compressed(
        os.path.join(out, "e2e_%s.npz" % tag),
        y=_np(y_r), head_w=_np(sd[wk]), head_b=_np(sd[bk]),
        meta=np.array(json.dumps(dict(meta, weight_seed=s, clip_seed=seed, clips=clips, H=H, W=W, argmax=idx,
                                      top2_gap=gap, head_w_key=wk, head_b_key=bk,
                                      ymin=float(y_r.min()), ymax=float(y_r.max()), ystd=float(y_r.std())))))
    print("e2e", tag, "ok: range [%.4f, %.4f] std %.4f argmax %d gap %.3g" % (y_r.min(), y_r.max(), y_r.std(), idx, gap))


def decoder_case(RM, seed, out, clips=8):
    """DecoderConvUp{8,16,48} alone (T-concat seams, 5 upsamples, the clip-length specific tail) incl. input and
    parameter grads.  Encoder feature maps of a `clips`-frame clip: T = clips/8, /4, /2, /2 (model.py:690-743)."""
    meta = {}
    cls = {8: "DecoderConvUp8", 16: "DecoderConvUp16", 48: "DecoderConvUp48", 32: "DecoderConvUp"}[clips]
    ref, ora = getattr(RM, cls)(), getattr(O, cls)()
    sd = synth.synth_state_dict(ref.state_dict(), seed)
    t0 = clips // 8
    hw = (3, 6) if clips == 8 else (2, 3)
    shapes = [(1, 1024, t0, hw[0], hw[1]), (1, 832, 2 * t0, 2 * hw[0], 2 * hw[1]), (1, 480, 4 * t0, 4 * hw[0], 4 * hw[1]),
              (1, 192, 4 * t0, 8 * hw[0], 8 * hw[1])]
    ys = [synth.normal("dec_y%d" % i, s, seed).abs() for i, s in enumerate(shapes)]
    # un-calibrated logits are far from 0; calibrate as for e2e
    box, h = _logits_hook(ref, ref)
    ref.load_state_dict(sd)
    with torch.no_grad():
        ref(*ys)
    h.remove()
    last = max(int(k.split(".")[1]) for k in sd if k.startswith("convtsp4.") and k.endswith(".bias"))
    wk, bk = "convtsp4.%d.weight" % last, "convtsp4.%d.bias" % last
    w, b = synth.calibrate_head(sd[wk], sd[bk], float(box["l"].mean()), float(box["l"].std()))
    sd[wk], sd[bk] = w, b
    res = {}
    outs = []
    for m in (ref, ora):
        m.load_state_dict(sd)
        yi = [y.clone().requires_grad_(True) for y in ys]
        o = m(*yi)
        proj = synth.normal("dec_proj", tuple(o.shape), seed)
        m.zero_grad()
        (o * proj).sum().backward()
        outs.append((o.detach(), [t.grad for t in yi], {k: p.grad.clone() for k, p in m.named_parameters()}))
    (o_r, gy_r, gp_r), (o_o, gy_o, gp_o) = outs
    _check("dec/out", o_r, o_o, meta)
    for i in range(4):
        _check("dec/gy%d" % i, gy_r[i], gy_o[i], meta)
    for k in gp_r:
        _check("dec/g_" + k, gp_r[k], gp_o[k], meta)
    res["out"] = _np(o_r)
    res["gy0"], res["gy1"] = _np(gy_r[0]), _np(gy_r[1])
    # big grads: keep per-tensor (sum, sum of squares, first 64 values)
    for i in (2, 3):
        g = gy_r[i].double()
        res["gy%d_stats" % i] = np.array([float(g.sum()), float((g * g).sum())])
        res["gy%d_head" % i] = _np(gy_r[i].reshape(-1)[:4096])
    for k, g in gp_r.items():
        gd = g.double()
        res["gp_stats:" + k] = np.array([float(gd.sum()), float((gd * gd).sum())])
        res["gp_head:" + k] = _np(g.reshape(-1)[:2048])
    res["head_w"], res["head_b"] = _np(w), _np(b)
    res["meta"] = np.array(json.dumps(dict(meta, seed=seed, shapes=[list(s) for s in shapes], clips=clips, head_w_key=wk, head_b_key=bk)))
    np.savez_compressed(os.path.join(out, "decoder%d.npz" % clips), **res)
    print("decoder%d ok" % clips)


class _Args:
    """the loss flags of train.py:21-66 that utils.get_loss reads"""

    def __init__(self, **kw):
        self.kldiv, self.cc, self.sim, self.l1 = True, False, False, False
        self.kldiv_coeff, self.cc_coeff, self.sim_coeff, self.l1_coeff = 1.0, -1.0, -1.0, 1.0
        self.batch_size = 2
        self.__dict__.update(kw)


def loss_func_case(seed, out):
    """utils.loss_func / get_loss (utils.py:9-39) of the REAL reference: the default flag set, kldiv + cc + sim with the
    reference's default coefficients (train.py:36-37), non-default coefficients, and the 4-D multi-frame path
    (utils.py:27-37).  The reference allocates its accumulator with `.cuda()`: for this capture
    `Tensor.cuda` is the identity, nothing else is patched."""
    import utils as RUt      # the reference's utils.py (stubs for cv2 / torchvision are on sys.path)
    meta, res = {}, {}
    real_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        combos = {"default": _Args(), "kl_cc_sim": _Args(cc=True, sim=True),
                  "coeffs": _Args(cc=True, sim=True, kldiv_coeff=0.5, cc_coeff=-2.0, sim_coeff=-0.25), "cc_only": _Args(kldiv=False, cc=True)}
        s3 = synth.uniform("lf_s3", (2, 40, 56), seed, 0.01, 0.99)
        g3 = synth.gt_map(2, 40, 56, seed)
        s4 = synth.uniform("lf_s4", (2, 3, 24, 40), seed, 0.01, 0.99)
        g4 = synth.gt_map(6, 24, 40, seed + 1).reshape(2, 3, 24, 40)
        for name, a in combos.items():
            for tag, (s_, g_) in {"3d": (s3, g3), "4d": (s4, g4)}.items():
                si = s_.clone().requires_grad_(True)
                v = RUt.loss_func(si, g_, a)
                v.sum().backward()
                res["%s_%s" % (name, tag)] = _np(v.detach())
                res["%s_%s_grad" % (name, tag)] = _np(si.grad)
        res["meta"] = np.array(json.dumps(dict(meta, seed=seed, combos={k: {f: getattr(v, f) for f in ("kldiv", "cc", "sim", "kldiv_coeff", "cc_coeff", "sim_coeff")} for k, v in combos.items()})))
    finally:
        torch.Tensor.cuda = real_cuda
    np.savez_compressed(os.path.join(out, "loss_func.npz"), **res)
    print("loss_func ok", {k: float(v.reshape(-1)[0]) for k, v in res.items() if v.ndim <= 1 and k != "meta" and not k.endswith("grad")})


def loss_case(RL, seed, out):
    meta, res = {}, {}
    for tag, (B, H, W) in {"full": (2, 224, 384), "small": (3, 40, 56)}.items():
        s = synth.uniform("loss_s_" + tag, (B, H, W), seed, 0.01, 0.99)
        g = synth.gt_map(B, H, W, seed)
        for fn in ("kldiv", "cc", "similarity"):
            vals = []
            for mod in (RL, O):
                si = s.clone().requires_grad_(True)
                v = getattr(mod, fn)(si, g)
                v.backward()
                vals.append((v.detach(), si.grad.detach()))
            _check("loss/%s/%s" % (tag, fn), vals[0][0], vals[1][0], meta)
            _check("loss/%s/%s/grad" % (tag, fn), vals[0][1], vals[1][1], meta)
            res["%s_%s" % (tag, fn)] = _np(vals[0][0])
            if tag == "small":
                res["%s_%s_grad" % (tag, fn)] = _np(vals[0][1])
            else:
                gd = vals[0][1].double()
                res["%s_%s_gradstats" % (tag, fn)] = np.array([float(gd.sum()), float((gd * gd).sum()), float(gd.abs().max())])
        # float64 ground truth as on the DIEM path
        v64 = RL.kldiv(s, g.double())
        res["%s_kldiv_gt64" % tag] = _np(v64)
        # nss (loss.py:101-120, equal-size branch) against a binary fixation mask
        fix = (g > 0.5 * g.amax(dim=(1, 2), keepdim=True)).float()
        n_r, n_o = RL.nss(s, fix), O.nss(s, fix)
        _check("loss/%s/nss" % tag, n_r, n_o, meta)
        res["%s_nss" % tag] = _np(n_r)
    res["meta"] = np.array(json.dumps(dict(meta, seed=seed)))
    np.savez_compressed(os.path.join(out, "loss.npz"), **res)
    print("loss ok", {k: float(v) for k, v in res.items() if k.startswith("full_") and v.ndim == 0})


def train_step_case(RM, RL, seed, out, shape=(2, 8, 64, 96), name="train_step"):
    """one Adam step (train.py:208-217) on ViNet-8: B=2, 8x64x96 (`train_step`: 12 samples per channel in the deepest BatchNorms --
    ill-conditioned on purpose) and B=12, 8x128x192 (`train_step_wc`, round 5: 288 samples per channel there, so that a gradient
    gate can tell a 2^-17 operand error from a bug)."""
    meta, res = {}, {}
    B, T, H, W = shape
    x = synth.clip(B, T, H, W, seed).permute(0, 2, 1, 3, 4)
    gt = synth.gt_map(B, H, W, seed)
    ref = RM.VideoSaliencyModel(num_clips=8)
    ora = O.VideoSaliencyModel(num_clips=8)

    def run_logits(m):
        m.eval()
        box, h = _logits_hook(m, m.decoder)
        with torch.no_grad():
            m(x)
        h.remove()
        return box["l"]

    sd, wk, bk = _calibrated_sd(ref, seed, run_logits)
    results = []
    for m, kld in ((ref, RL.kldiv), (ora, O.kldiv)):
        m.load_state_dict(sd)
        m.train()
        opt = torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr=1e-4)
        opt.zero_grad()
        pred = m(x)
        loss0 = kld(pred, gt)
        loss0.backward()
        grads = {k: p.grad.detach().clone() for k, p in m.named_parameters()}
        opt.step()
        with torch.no_grad():
            loss1 = kld(m(x), gt)
        results.append((pred.detach(), loss0.detach(), loss1.detach(), grads, {k: v.detach().clone() for k, v in m.state_dict().items()}))
    (p_r, l0_r, l1_r, g_r, sd_r), (p_o, l0_o, l1_o, g_o, sd_o) = results
    _check("train/pred", p_r, p_o, meta)
    _check("train/loss0", l0_r, l0_o, meta)
    _check("train/loss1", l1_r, l1_o, meta)
    for k in g_r:
        _check("train/g_" + k, g_r[k], g_o[k], meta)
    res["pred"] = _np(p_r)
    res["loss0"], res["loss1"] = _np(l0_r), _np(l1_r)
    names = list(g_r.keys())
    res["grad_names"] = np.array(json.dumps(names))
    res["grad_sum"] = np.array([float(g_r[k].double().sum()) for k in names])
    res["grad_sqsum"] = np.array([float((g_r[k].double() ** 2).sum()) for k in names])
    keep = ["backbone.base1.0.conv_s.weight", "backbone.base1.0.bn_s.weight", "backbone.base1.0.bn_s.bias",
            "backbone.base2.0.branch3.1.conv.weight", "backbone.base4.1.branch1.1.conv_t.weight",
            "decoder.convtsp4.3.weight", "decoder.convtsp4.6.weight", "decoder.convtsp4.6.bias"]
    for k in keep:
        res["grad:" + k] = _np(g_r[k].reshape(-1)[:8192])
    sn = list(sd_r.keys())
    res["state_names"] = np.array(json.dumps(sn))
    res["state_sum"] = np.array([float(sd_r[k].double().sum()) for k in sn])
    res["state_sqsum"] = np.array([float((sd_r[k].double() ** 2).sum()) for k in sn])
    for k in ("backbone.base1.0.bn_s.running_mean", "backbone.base1.0.bn_s.running_var",
              "backbone.base4.1.branch0.0.bn.running_mean", "backbone.base4.1.branch0.0.bn.running_var"):
        res["state:" + k] = _np(sd_r[k])
    res["head_w"], res["head_b"] = _np(sd[wk]), _np(sd[bk])
    res["meta"] = np.array(json.dumps(dict(meta, seed=seed, B=B, T=T, H=H, W=W, lr=1e-4, head_w_key=wk, head_b_key=bk)))
    np.savez_compressed(os.path.join(out, name + ".npz"), **res)
    print("%s ok: loss %.6f -> %.6f" % (name, float(l0_r), float(l1_r)))


def trajectory_batches(nb, B, T, H, W, seed):
    """the fixed rotation of synthetic batches of the trajectory fixture: [(x [B,3,T,H,W], gt [B,H,W])] * nb (recipe, not data)"""
    return [(synth.clip(B, T, H, W, seed + 100 * i).permute(0, 2, 1, 3, 4).contiguous(), synth.gt_map(B, H, W, seed + 100 * i)) for i in range(nb)]


def trajectory_case(RM, RL, seed, out, steps=48, nb=4, shape=(8, 8, 128, 192), name="train_trajectory", ensemble=8):
    """the REAL reference run as a training loop -- train.py:208-217: `optimizer.zero_grad(); pred = model(img);
    loss = kldiv(pred, gt); loss.backward(); optimizer.step()` with Adam(lr 1e-4, train.py:187) -- for `steps` steps over a fixed
    rotation of `nb` synthetic batches (ViNet-8, B = 8, 8 x 128 x 192, procedural weights with the calibrated head).  Stored: the loss
    of every step, per-tensor checksums of the final state_dict, the eval-mode loss on batch 0 before and after.  The oracle must
    reproduce the whole trajectory or nothing is written."""
    B, T, H, W = shape
    batches = trajectory_batches(nb, B, T, H, W, seed)
    ref = RM.VideoSaliencyModel(num_clips=T)
    ora = O.VideoSaliencyModel(num_clips=T)
    x0 = batches[0][0]

    def run_logits(m):
        m.eval()
        box, h = _logits_hook(m, m.decoder)
        with torch.no_grad():
            m(x0)
        h.remove()
        return box["l"]

    sd, wk, bk = _calibrated_sd(ref, seed, run_logits)
    runs = []
    import time
    for tag, m, kld in (("reference", ref, RL.kldiv), ("oracle", ora, O.kldiv)):
        m.load_state_dict(sd)
        m.eval()
        with torch.no_grad():
            ev0 = float(kld(m(batches[0][0]), batches[0][1]))
        m.train()
        opt = torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr=1e-4)
        losses, t0 = [], time.time()
        for i in range(steps):
            x, gt = batches[i % nb]
            opt.zero_grad()
            loss = kld(m(x), gt)
            loss.backward()
            opt.step()
            losses.append(float(loss))
            if i % 8 == 0:
                print("  %s step %d loss %.6f (%.0f s)" % (tag, i, losses[-1], time.time() - t0), flush=True)
        m.eval()
        with torch.no_grad():
            ev1 = float(kld(m(batches[0][0]), batches[0][1]))
        runs.append((np.array(losses), ev0, ev1, {k: v.detach().clone() for k, v in m.state_dict().items()}))
    (l_r, e0_r, e1_r, sd_r), (l_o, e0_o, e1_o, sd_o) = runs
    meta = {}
    d = float(np.abs(l_r - l_o).max())
    meta["oracle_vs_reference_loss_maxabs"] = d
    # (the oracle builds the same aten graph: the two runs are bit-identical on one machine; allow round-off only)
    if not d <= 1e-5 * float(np.abs(l_r).max()):
        raise SystemExit("ORACLE DOES NOT REPRODUCE THE REFERENCE'S TRAINING TRAJECTORY: %g" % d)
    dp = max(_maxdiff(sd_r[k], sd_o[k]) for k in sd_r)
    meta["oracle_vs_reference_state_maxabs"] = dp
    if not dp <= 1e-4:
        raise SystemExit("ORACLE'S FINAL STATE DIFFERS FROM THE REFERENCE'S: %g" % dp)
    # The trajectory is CHAOTIC beyond ~10 steps (Adam's sign-like steps amplify round-off: the HIP fp32 path, 3e-4 from the
    # reference for 6 steps, is O(1) away by step 16 -- and so is the reference from ITSELF): the yardstick for "trains like the
    # reference" is therefore the reference's own spread.  An ensemble of the reference with its initial weights perturbed by
    # (1 + 2^-20 xi), xi ~ N(0, 1) -- a few fp32 ulps -- and one fp64 run of the unperturbed weights.
    ens = []
    for member in range(ensemble):
        g = torch.Generator().manual_seed(1000 + member)
        sdm = {k: (v * (1 + 2.0 ** -20 * torch.randn(v.shape, generator=g)) if v.is_floating_point() and v.dim() > 0 else v) for k, v in sd.items()}
        ref.load_state_dict(sdm)
        ref.train()
        opt = torch.optim.Adam([p for p in ref.parameters() if p.requires_grad], lr=1e-4)
        ls = []
        for i in range(steps):
            x, gt = batches[i % nb]
            opt.zero_grad()
            loss = RL.kldiv(ref(x), gt)
            loss.backward()
            opt.step()
            ls.append(float(loss))
        print("  ensemble member %d: end %.4f" % (member, np.mean(ls[-4:])), flush=True)
        ens.append(ls)
    l64 = None
    if ensemble:
        ref.load_state_dict(sd)
        ref.double().train()
        opt = torch.optim.Adam([p for p in ref.parameters() if p.requires_grad], lr=1e-4)
        l64 = []
        for i in range(steps):
            x, gt = batches[i % nb]
            opt.zero_grad()
            loss = RL.kldiv(ref(x.double()), gt.double())
            loss.backward()
            opt.step()
            l64.append(float(loss))
        ref.float()
        print("  fp64 run: end %.4f" % np.mean(l64[-4:]), flush=True)
    res = dict(losses=l_r, eval_loss_before=np.array(e0_r), eval_loss_after=np.array(e1_r))
    if ensemble:
        res["ensemble_losses"], res["fp64_losses"] = np.array(ens), np.array(l64)
    sn = list(sd_r.keys())
    res["state_names"] = np.array(json.dumps(sn))
    res["state_sum"] = np.array([float(sd_r[k].double().sum()) for k in sn])
    res["state_sqsum"] = np.array([float((sd_r[k].double() ** 2).sum()) for k in sn])
    # how far training moved each tensor: || p_end - p_0 || (the yardstick for a parameter-space comparison)
    res["state_delta_norm"] = np.array([float((sd_r[k].double() - sd[k].double()).norm()) for k in sn])
    res["head_w"], res["head_b"] = _np(sd[wk]), _np(sd[bk])
    res["meta"] = np.array(json.dumps(dict(meta, seed=seed, steps=steps, batches=nb, B=B, T=T, H=H, W=W, lr=1e-4, head_w_key=wk, head_b_key=bk,
                                           batch_seed_rule="seed + 100 * i")))
    np.savez_compressed(os.path.join(out, name + ".npz"), **res)
    print("%s ok: loss %.6f -> %.6f over %d steps; eval(batch 0) %.6f -> %.6f" % (name, l_r[0], l_r[-1], steps, e0_r, e1_r))


def avinet_case(RM, seed, out):
    meta = {}
    cwd = os.getcwd()
    tmp = tempfile.mkdtemp(prefix="vinet_snd_")
    os.chdir(tmp)
    try:
        torch.save(RM.SoundNet().state_dict(), "soundnet8_final.pth")  # model.py:224 reads it
        ref = RM.VideoAudioSaliencyModel(num_clips=32).eval()
    finally:
        os.chdir(cwd)
    ora = O.VideoAudioSaliencyModel(num_clips=32).eval()
    x = synth.clip(1, 32, 224, 384, seed).permute(0, 2, 1, 3, 4)
    a = synth.audio(1, 70560, seed)

    def run_logits(m):
        box, h = _logits_hook(m, m.visual_model.decoder)
        with torch.no_grad():
            m(x, a)
        h.remove()
        return box["l"]

    sd, wk, bk = _calibrated_sd(ref, seed, run_logits)
    ref.load_state_dict(sd)
    ora.load_state_dict(sd)
    with torch.no_grad():
        y_r = ref(x, a)
        y_o = ora(x, a)
        aud = ref.audionet(a)
    _check("avinet/y", y_r, y_o, meta)
    idx, gap = _top2(y_r[0])
    np.savez_compressed(os.path.join(out, "avinet32.npz"), y=_np(y_r), audio_feat=_np(aud), head_w=_np(sd[wk]), head_b=_np(sd[bk]),
                        meta=np.array(json.dumps(dict(meta, seed=seed, argmax=idx, top2_gap=gap, head_w_key=wk, head_b_key=bk))))
    print("avinet ok: range [%.4f, %.4f] argmax %d gap %.3g" % (y_r.min(), y_r.max(), idx, gap))


def main():
    torch.set_num_threads(os.cpu_count())
    RM, RU, RL = _import_reference()
    out = HERE
    if sys.argv[1:] == ["loss"]:        # regenerate one fixture
        loss_case(RL, 3, out)
        return
    if sys.argv[1:] == ["round6_blocks"]:    # block goldens for the Inception stages that had none (VERDICT r5 coverage rows a3 / a4 / a5)
        block_case("mixed_3c", lambda: RU.Mixed_3c(), lambda: O.Mixed_3c(), (1, 256, 4, 14, 24), 16, out, compact=True)
        block_case("mixed_4b", lambda: RU.Mixed_4b(), lambda: O.Mixed_4b(), (1, 480, 4, 7, 12), 17, out, compact=True)
        block_case("mixed_4f", lambda: RU.Mixed_4f(), lambda: O.Mixed_4f(), (1, 528, 4, 7, 12), 18, out, compact=True)
        block_case("mixed_5c", lambda: RU.Mixed_5c(), lambda: O.Mixed_5c(), (2, 832, 2, 7, 12), 19, out, compact=True)
        return
    if sys.argv[1:] == ["round6_decoder32"]:  # the headline configuration's decoder on its own (row a7: model.py:251-311)
        decoder_case(RM, 24, out, clips=32)
        return
    if sys.argv[1:] == ["round6_soundnet"]:  # the audio branch on its own (row a9): model.py:746-825, eval + train mode, every gradient
        block_case("soundnet", lambda: RM.SoundNet(), lambda: O.SoundNet(), (2, 1, 70560, 1), 20, out, compact=True)
        return
    if sys.argv[1:2] == ["round6"]:      # the reference's training loop as a loss trajectory
        trajectory_case(RM, RL, 61, out, steps=int(sys.argv[2]) if len(sys.argv) > 2 else 48)
        return
    if sys.argv[1:] == ["round5"]:      # the well-conditioned training-step fixture
        train_step_case(RM, RL, 43, out, shape=(12, 8, 128, 192), name="train_step_wc")
        return
    if sys.argv[1:] == ["round4"]:      # a block golden at the M = 336-voxel stage (4 x 7 x 12: base4 of a 32 x 224 x 384 clip)
        block_case("mixed_5b", lambda: RU.Mixed_5b(), lambda: O.Mixed_5b(), (1, 832, 4, 7, 12), 15, out, compact=True)
        return
    if sys.argv[1:] == ["round2"]:      # the fixtures added in round 2 (the others are left untouched)
        loss_func_case(5, out)
        decoder_case(RM, 22, out, clips=16)
        decoder_case(RM, 23, out, clips=48)
        e2e_case(RM, 16, 64, 96, 34, out, "16x64x96")
        e2e_case(RM, 48, 64, 96, 35, out, "48x64x96")
        return
    block_case("basic_16_32", lambda: RU.BasicConv3d(16, 32, 1, 1), lambda: O.BasicConv3d(16, 32, 1, 1), (2, 16, 4, 6, 8), 11, out)
    block_case("sep_16_32_k3", lambda: RU.SepConv3d(16, 32, 3, 1, 1), lambda: O.SepConv3d(16, 32, 3, 1, 1), (2, 16, 4, 6, 8), 12, out)
    block_case("sep_3_64_k7s2", lambda: RU.SepConv3d(3, 64, 7, 2, 3), lambda: O.SepConv3d(3, 64, 7, 2, 3), (1, 3, 8, 16, 24), 13, out)
    block_case("mixed_3b", lambda: RU.Mixed_3b(), lambda: O.Mixed_3b(), (1, 192, 4, 6, 8), 14, out)
    loss_case(RL, 3, out)
    decoder_case(RM, 21, out)
    e2e_case(RM, 8, 96, 192, 31, out, "8x96x192")
    e2e_case(RM, 8, 128, 192, 32, out, "8x128x192")
    train_step_case(RM, RL, 41, out)
    e2e_case(RM, 32, 224, 384, 33, out, "32x224x384")
    avinet_case(RM, 51, out)
    loss_func_case(5, out)
    decoder_case(RM, 22, out, clips=16)
    decoder_case(RM, 23, out, clips=48)
    e2e_case(RM, 16, 64, 96, 34, out, "16x64x96")
    e2e_case(RM, 48, 64, 96, 35, out, "48x64x96")
    block_case("mixed_5b", lambda: RU.Mixed_5b(), lambda: O.Mixed_5b(), (1, 832, 4, 7, 12), 15, out, compact=True)


if __name__ == "__main__":
    main()
