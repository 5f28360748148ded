"""CPU oracle for the ViNet / AViNet saliency path  --  TEST INFRASTRUCTURE ONLY.

This file is a from-scratch PyTorch-CPU (fp32, aten ops) restatement of the
reference's hot path.  It exists so that the HIP kernels in ``vinet_amd`` can
be checked for parity on a box where ``/root/reference`` does not exist.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import it.  The product package never does; the
product path raises if the HIP library is missing.

Pinning: ``tests/golden/make_goldens.py`` imports the real reference (with
stub modules for its unused imports), loads the same procedural weights into
both, and asserts this oracle reproduces the reference bit-for-bit on every
golden case before the fixture is written.  The goldens under
``tests/golden/`` are therefore outputs OF THE REFERENCE, and
``tests/test_oracle.py`` re-checks the oracle against them on every run.

Reference citations (``/root/reference``):
  BasicConv3d          model_utils.py:128-139
  SepConv3d            model_utils.py:141-160
  Mixed_3b..5c         model_utils.py:162-420
  BackBoneS3D          model.py:690-743
  DecoderConvUp{,8,16,48}  model.py:251-311, 375-435, 313-373, 437-498
  VideoSaliencyModel   model.py:72-112
  SoundNet             model.py:746-825
  VideoAudioSaliencyModel  model.py:191-249
  kldiv / normalize_map / similarity / cc   loss.py:13-99
  loss_func / get_loss utils.py:9-39
"""
import torch
from torch import nn

# --------------------------------------------------------------------------
# conv blocks (model_utils.py:128-160)
# --------------------------------------------------------------------------


class BasicConv3d(nn.Module):
    """conv(bias=False) -> BN3d(eps 1e-3, momentum 1e-3) -> ReLU  (model_utils.py:128-139)."""

    def __init__(self, in_planes, out_planes, kernel_size, stride, padding=0):
        super().__init__()
        self.conv = nn.Conv3d(in_planes, out_planes, kernel_size, stride, padding, bias=False)
        self.bn = nn.BatchNorm3d(out_planes, eps=1e-3, momentum=0.001)
        self.relu = nn.ReLU()

    def forward(self, x):
        return self.relu(self.bn(self.conv(x)))


class SepConv3d(nn.Module):
    """(1,k,k) conv/BN/ReLU then (k,1,1) conv/BN/ReLU, stride and padding split
    the same way (model_utils.py:141-160)."""

    def __init__(self, in_planes, out_planes, kernel_size, stride, padding=0):
        super().__init__()
        k, s, p = kernel_size, stride, padding
        self.conv_s = nn.Conv3d(in_planes, out_planes, (1, k, k), (1, s, s), (0, p, p), bias=False)
        self.bn_s = nn.BatchNorm3d(out_planes, eps=1e-3, momentum=0.001)
        self.relu_s = nn.ReLU()
        self.conv_t = nn.Conv3d(out_planes, out_planes, (k, 1, 1), (s, 1, 1), (p, 0, 0), bias=False)
        self.bn_t = nn.BatchNorm3d(out_planes, eps=1e-3, momentum=0.001)
        self.relu_t = nn.ReLU()

    def forward(self, x):
        x = self.relu_s(self.bn_s(self.conv_s(x)))
        return self.relu_t(self.bn_t(self.conv_t(x)))


# Inception widths: name -> (in, b0, b1_reduce, b1_out, b2_reduce, b2_out, b3)
# (model_utils.py:162-420)
MIXED_WIDTHS = {
    "3b": (192, 64, 96, 128, 16, 32, 32),
    "3c": (256, 128, 128, 192, 32, 96, 64),
    "4b": (480, 192, 96, 208, 16, 48, 64),
    "4c": (512, 160, 112, 224, 24, 64, 64),
    "4d": (512, 128, 128, 256, 24, 64, 64),
    "4e": (512, 112, 144, 288, 32, 64, 64),
    "4f": (528, 256, 160, 320, 32, 128, 128),
    "5b": (832, 256, 160, 320, 32, 128, 128),
    "5c": (832, 384, 192, 384, 48, 128, 128),
}


class _Mixed(nn.Module):
    """Four-branch Inception block (model_utils.py:162-189 and siblings)."""

    def __init__(self, name):
        super().__init__()
        cin, b0, b1r, b1, b2r, b2, b3 = MIXED_WIDTHS[name]
        self.branch0 = nn.Sequential(BasicConv3d(cin, b0, kernel_size=1, stride=1))
        self.branch1 = nn.Sequential(
            BasicConv3d(cin, b1r, kernel_size=1, stride=1),
            SepConv3d(b1r, b1, kernel_size=3, stride=1, padding=1),
        )
        self.branch2 = nn.Sequential(
            BasicConv3d(cin, b2r, kernel_size=1, stride=1),
            SepConv3d(b2r, b2, kernel_size=3, stride=1, padding=1),
        )
        self.branch3 = nn.Sequential(
            nn.MaxPool3d(kernel_size=(3, 3, 3), stride=1, padding=1),
            BasicConv3d(cin, b3, kernel_size=1, stride=1),
        )

    def forward(self, x):
        return torch.cat((self.branch0(x), self.branch1(x), self.branch2(x), self.branch3(x)), 1)


def _mixed_cls(name):
    return type("Mixed_" + name, (_Mixed,), {"__init__": lambda self: _Mixed.__init__(self, name)})


Mixed_3b, Mixed_3c = _mixed_cls("3b"), _mixed_cls("3c")
Mixed_4b, Mixed_4c, Mixed_4d = _mixed_cls("4b"), _mixed_cls("4c"), _mixed_cls("4d")
Mixed_4e, Mixed_4f = _mixed_cls("4e"), _mixed_cls("4f")
Mixed_5b, Mixed_5c = _mixed_cls("5b"), _mixed_cls("5c")


# --------------------------------------------------------------------------
# backbone (model.py:690-743)
# --------------------------------------------------------------------------


class BackBoneS3D(nn.Module):
    def __init__(self):
        super().__init__()
        self.base1 = nn.Sequential(
            SepConv3d(3, 64, kernel_size=7, stride=2, padding=3),
            nn.MaxPool3d((1, 3, 3), (1, 2, 2), (0, 1, 1)),
            BasicConv3d(64, 64, kernel_size=1, stride=1),
            SepConv3d(64, 192, kernel_size=3, stride=1, padding=1),
        )
        self.maxp2 = nn.MaxPool3d((1, 3, 3), (1, 2, 2), (0, 1, 1))
        self.base2 = nn.Sequential(Mixed_3b(), Mixed_3c())
        self.maxp3 = nn.MaxPool3d((3, 3, 3), (2, 2, 2), (1, 1, 1))
        self.base3 = nn.Sequential(Mixed_4b(), Mixed_4c(), Mixed_4d(), Mixed_4e(), Mixed_4f())
        self.maxt4 = nn.MaxPool3d((2, 1, 1), (2, 1, 1), (0, 0, 0))
        self.maxp4 = nn.MaxPool3d((1, 2, 2), (1, 2, 2), (0, 0, 0))
        self.base4 = nn.Sequential(Mixed_5b(), Mixed_5c())

    def forward(self, x):
        y3 = self.base1(x)
        y2 = self.base2(self.maxp2(y3))
        y1 = self.base3(self.maxp3(y2))
        y0 = self.base4(self.maxp4(self.maxt4(y1)))
        return [y0, y1, y2, y3]


# --------------------------------------------------------------------------
# decoders (model.py:251-498).  The four clip lengths differ only in the tail.
# tail spec per num_clips: list of (kind, args)
# --------------------------------------------------------------------------

DECODER_TAILS = {
    # (conv5 kT, tail convs after the last upsample) -- model.py:277-283 / 339-346 / 401-408 / 463-469
    32: dict(k5=2, tail=[(32, 32, 2, False), "relu", (32, 1, 1, True)]),
    16: dict(k5=2, tail=[(32, 1, 1, True)]),
    8: dict(k5=1, tail=[(32, 1, 1, True)]),
    48: dict(k5=2, tail=[(32, 32, 3, True), "relu", (32, 1, 1, True)]),
    # BUILD-DEFINED, NO REFERENCE PARITY (BASELINE config 5): the reference wires no decoder for 64-frame
    # clips (model.py:91-99); this is DecoderConvUp (model.py:251-311) with its last temporal conv (2,1,1)/s2 widened
    # to (4,1,1)/s4 so that T ends at 1.  Checked only HIP-vs-this-restatement; no golden from the reference exists.
    64: dict(k5=2, tail=[(32, 32, 4, False), "relu", (32, 1, 1, True)]),
}


class _DecoderConvUp(nn.Module):
    def __init__(self, num_clips):
        super().__init__()
        spec = DECODER_TAILS[num_clips]
        self.upsampling = nn.Upsample(scale_factor=(1, 2, 2), mode="trilinear")

        def stage(cin, cout, kt):
            return [nn.Conv3d(cin, cout, (kt, 3, 3), (kt, 1, 1), (0, 1, 1), bias=False), nn.ReLU(), self.upsampling]

        self.convtsp1 = nn.Sequential(*stage(1024, 832, 1))
        self.convtsp2 = nn.Sequential(*stage(832, 480, 3))
        self.convtsp3 = nn.Sequential(*stage(480, 192, 5))
        tail = []
        for item in spec["tail"]:
            if item == "relu":
                tail.append(nn.ReLU())
            else:
                cin, cout, kt, bias = item
                tail.append(nn.Conv3d(cin, cout, (kt, 1, 1), (kt, 1, 1), bias=bias))
        self.convtsp4 = nn.Sequential(*stage(192, 64, 5), *stage(64, 32, spec["k5"]), *tail, nn.Sigmoid())

    def forward(self, y0, y1, y2, y3):
        z = self.convtsp1(y0)
        z = self.convtsp2(torch.cat((z, y1), 2))
        z = self.convtsp3(torch.cat((z, y2), 2))
        z = self.convtsp4(torch.cat((z, y3), 2))
        return z.view(z.size(0), z.size(3), z.size(4))


class DecoderConvUp(_DecoderConvUp):
    def __init__(self):
        super().__init__(32)


class DecoderConvUp16(_DecoderConvUp):
    def __init__(self):
        super().__init__(16)


class DecoderConvUp8(_DecoderConvUp):
    def __init__(self):
        super().__init__(8)


class DecoderConvUp48(_DecoderConvUp):
    def __init__(self):
        super().__init__(48)


class DecoderConvUp64(_DecoderConvUp):
    """build-defined 64-frame variant, no reference counterpart (see DECODER_TAILS[64])"""

    def __init__(self):
        super().__init__(64)


class VideoSaliencyModel(nn.Module):
    """model.py:72-112 (use_upsample=True, num_hier=3 path only; the ablation
    decoders are out of scope)."""

    def __init__(self, transformer_in_channel=32, nhead=4, use_upsample=True, num_hier=3, num_clips=32):
        super().__init__()
        if not use_upsample or num_hier != 3 or num_clips not in DECODER_TAILS:
            raise NotImplementedError("only use_upsample=True, num_hier=3, num_clips in {8,16,32,48} (+ build-defined 64)")
        self.backbone = BackBoneS3D()
        self.num_hier = num_hier
        self.decoder = {8: DecoderConvUp8, 16: DecoderConvUp16, 32: DecoderConvUp, 48: DecoderConvUp48, 64: DecoderConvUp64}[num_clips]()

    def forward(self, x):
        y0, y1, y2, y3 = self.backbone(x)
        return self.decoder(y0, y1, y2, y3)


# --------------------------------------------------------------------------
# audio branch (model.py:746-825, 191-249)
# --------------------------------------------------------------------------

# (cin, cout, k, pad, pool) per SoundNet layer; stride is always (2,1)
SOUNDNET_LAYERS = [
    (1, 16, 64, 32, 8),
    (16, 32, 32, 16, 8),
    (32, 64, 16, 8, 0),
    (64, 128, 8, 4, 0),
    (128, 256, 4, 2, 4),
    (256, 512, 4, 2, 0),
    (512, 1024, 4, 2, 0),
]


class SoundNet(nn.Module):
    def __init__(self):
        super().__init__()
        for i, (cin, cout, k, p, pool) in enumerate(SOUNDNET_LAYERS, 1):
            setattr(self, "conv%d" % i, nn.Conv2d(cin, cout, (k, 1), (2, 1), (p, 0)))
            setattr(self, "batchnorm%d" % i, nn.BatchNorm2d(cout, eps=1e-5, momentum=0.1))
            setattr(self, "relu%d" % i, nn.ReLU(True))
            if pool:
                setattr(self, "maxpool%d" % i, nn.MaxPool2d((pool, 1), (pool, 1)))
        # present in the checkpoint, never used in forward (model.py:788-791)
        self.conv8_objs = nn.Conv2d(1024, 1000, (8, 1), (2, 1))
        self.conv8_scns = nn.Conv2d(1024, 401, (8, 1), (2, 1))

    def forward(self, waveform):
        x = waveform
        for i, (_, _, _, _, pool) in enumerate(SOUNDNET_LAYERS, 1):
            x = getattr(self, "conv%d" % i)(x)
            x = getattr(self, "batchnorm%d" % i)(x)
            x = getattr(self, "relu%d" % i)(x)
            if pool:
                x = getattr(self, "maxpool%d" % i)(x)
        return x


class VideoAudioSaliencyModel(nn.Module):
    """model.py:191-249, use_transformer=False path.  Unlike the reference the
    SoundNet weights are not read from ./soundnet8_final.pth at construction;
    callers load a state_dict (the parity tests use procedural weights)."""

    def __init__(self, use_transformer=False, transformer_in_channel=32, num_encoder_layers=3, nhead=4,
                 use_upsample=True, num_hier=3, num_clips=32):
        super().__init__()
        if use_transformer:
            raise NotImplementedError("transformer fusion is out of scope (SURVEY.md section 2)")
        self.use_transformer = False
        self.visual_model = VideoSaliencyModel(transformer_in_channel, nhead, use_upsample, num_hier, num_clips)
        self.audionet = SoundNet()
        self.maxpool = nn.MaxPool3d((4, 1, 1), stride=(2, 1, 2), padding=(0, 0, 0))
        self.bilinear = nn.Bilinear(42, 3, 4 * 7 * 12)

    def forward(self, x, audio):
        audio = self.audionet(audio)
        y0, y1, y2, y3 = self.visual_model.backbone(x)
        y0 = self.maxpool(y0)
        fused = self.bilinear(y0.flatten(2), audio.flatten(2))
        fused = fused.view(fused.size(0), fused.size(1), 4, 7, 12)
        return self.visual_model.decoder(fused, y1, y2, y3)


# --------------------------------------------------------------------------
# losses (loss.py:13-99) and glue (utils.py:9-39)
# --------------------------------------------------------------------------

EPS = 2.2204e-16


def kldiv(s_map, gt):
    assert s_map.size() == gt.size()
    b = s_map.size(0)
    p = s_map.reshape(b, -1)
    q = gt.reshape(b, -1)
    p = p / (p.sum(1, keepdim=True) * 1.0)
    q = q / (q.sum(1, keepdim=True) * 1.0)
    return torch.mean(torch.sum(q * torch.log(EPS + q / (p + EPS)), 1))


def normalize_map(s_map):
    b = s_map.size(0)
    flat = s_map.reshape(b, -1)
    lo = flat.min(1)[0].view(b, 1, 1)
    hi = flat.max(1)[0].view(b, 1, 1)
    return (s_map - lo) / (hi - lo * 1.0)


def similarity(s_map, gt):
    b = s_map.size(0)
    p = normalize_map(s_map).reshape(b, -1)
    q = normalize_map(gt).reshape(b, -1)
    p = p / (p.sum(1, keepdim=True) * 1.0)
    q = q / (q.sum(1, keepdim=True) * 1.0)
    return torch.mean(torch.sum(torch.min(p, q), 1))


def cc(s_map, gt):
    assert s_map.size() == gt.size()
    b = s_map.size(0)
    a = s_map.reshape(b, -1)
    g = gt.reshape(b, -1)
    a = (a - a.mean(1, keepdim=True)) / a.std(1, keepdim=True)
    g = (g - g.mean(1, keepdim=True)) / g.std(1, keepdim=True)
    ab = (a * g).sum(1)
    aa = (a * a).sum(1)
    bb = (g * g).sum(1)
    return torch.mean(ab / torch.sqrt(aa * bb))


def nss(s_map, gt):
    """loss.py:101-120, equal-size branch (the size-mismatch branch is a cv2.resize on the host)."""
    assert s_map.size() == gt.size()
    b = s_map.size(0)
    a = s_map.reshape(b, -1)
    g = gt.reshape(b, -1)
    z = (a - a.mean(1, keepdim=True)) / (a.std(1, keepdim=True) + 2.2204e-16)
    return torch.mean((z * g).sum(1) / g.sum(1))


def get_loss(pred_map, gt, args):
    """utils.py:9-20 without the CUDA-only zero tensor."""
    loss = torch.zeros(1, dtype=torch.float32, device=pred_map.device)
    if args.kldiv:
        loss += args.kldiv_coeff * kldiv(pred_map, gt)
    if args.cc:
        loss += args.cc_coeff * cc(pred_map, gt)
    if getattr(args, "l1", False):
        loss += args.l1_coeff * nn.functional.l1_loss(pred_map, gt)
    if args.sim:
        loss += args.sim_coeff * similarity(pred_map, gt)
    return loss


def loss_func(pred_map, gt, args):
    """utils.py:22-39."""
    assert pred_map.size() == gt.size()
    if pred_map.dim() == 4:
        assert pred_map.size(0) == args.batch_size
        pred_map = pred_map.permute((1, 0, 2, 3))
        gt = gt.permute((1, 0, 2, 3))
        loss = torch.zeros(1, dtype=torch.float32, device=pred_map.device)
        for i in range(pred_map.size(0)):
            loss += get_loss(pred_map[i], gt[i], args)
        return loss / pred_map.size(0)
    return get_loss(pred_map, gt, args)
