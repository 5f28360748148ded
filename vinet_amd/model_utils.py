"""Drop-in replacements for the reference's conv blocks (model_utils.py:128-420).

Same class names, constructor arguments and state_dict keys as the reference
(`conv.weight`, `bn.running_mean`, `conv_s.weight`, `bn_t.bias`, `branch1.1.*` ...),
so reference checkpoints load unchanged; every forward/backward runs on the
hand-written HIP kernels of libvinet_hip.so through vinet_amd.engine.  The
nn.Conv3d / nn.BatchNorm3d sub-modules are *parameter containers only* (same
names, shapes and default init as the reference): their own forward is disabled.
"""
import torch
from torch import nn

from . import _lib as L
from . import engine as E


class ConvParams(nn.Conv3d):
    """weight (and bias) holder; compute happens in libvinet_hip.so."""

    def forward(self, x):  # pragma: no cover
        raise RuntimeError("ConvParams only holds parameters; call the owning vinet_amd block")

    def plan(self, stem=False):
        p = self.__dict__.get("_vinet_plan")
        if p is None or p.weight is not self.weight or p.bias is not self.bias:
            p = E.ConvPlan(self.weight, self.bias, self.kernel_size, self.stride, self.padding, stem=stem)
            self.__dict__["_vinet_plan"] = p
        return p


class _BNMixin:
    """running-stat holder.  `num_batches_tracked` is advanced lazily (it is not
    used with a fixed momentum) so a training step issues no extra device op."""

    def forward(self, x):  # pragma: no cover
        raise RuntimeError("BatchNorm parameters only; call the owning vinet_amd block")

    def state(self):
        return E.BNState(self.weight, self.bias, self.running_mean, self.running_var, self.eps, self.momentum,
                         fold_cache=self.__dict__.setdefault("_vinet_fold", {}))

    def note_training_step(self):
        self.__dict__["_vinet_pending"] = self.__dict__.get("_vinet_pending", 0) + 1
        self.__dict__.setdefault("_vinet_fold", {}).clear()   # running statistics changed under the cached fold

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        pend = self.__dict__.get("_vinet_pending", 0)
        if pend and self.num_batches_tracked is not None:
            self.num_batches_tracked += pend
            self.__dict__["_vinet_pending"] = 0
        super()._save_to_state_dict(destination, prefix, keep_vars)


class BNParams3d(_BNMixin, nn.BatchNorm3d):
    pass


class BNParams2d(_BNMixin, nn.BatchNorm2d):
    pass


class _Marker(nn.Module):
    """parameter-free placeholder that keeps nn.Sequential indices (and so the
    state_dict keys) identical to the reference's ReLU / Upsample / MaxPool slots."""

    def __init__(self, what, **kw):
        super().__init__()
        self.what = what
        self.kw = kw

    def extra_repr(self):
        return self.what

    def forward(self, x):  # pragma: no cover
        raise RuntimeError("marker module; call the owning vinet_amd block")


def conv_bn_relu(ctx, conv, bn, x, dst=None, stem=False):
    bs = bn.state()
    out = E.conv_forward(ctx, conv.plan(stem), x, bn=bs, act=L.ACT_RELU, dst=dst)
    if ctx.training:
        bn.note_training_step()
    return out


class _Block(nn.Module):
    """root dispatch shared by the conv blocks."""
    compute_dtype = None   # None -> engine default (bf16); set to engine.F32 for the parity path
    _cpad = None

    def forward(self, x):
        body = E.BlockBody(self, self._fwd, self._cpad)
        return E.run_root(body, [x], list(self.parameters()))[0]


class BasicConv3d(_Block):
    """model_utils.py:128-139."""

    def __init__(self, in_planes, out_planes, kernel_size, stride, padding=0):
        super().__init__()
        self.conv = ConvParams(in_planes, out_planes, kernel_size=kernel_size, stride=stride, padding=padding, bias=False)
        self.bn = BNParams3d(out_planes, eps=1e-3, momentum=0.001, affine=True)
        self.relu = _Marker("relu")

    def _fwd(self, ctx, x, dst=None):
        return conv_bn_relu(ctx, self.conv, self.bn, x, dst)


class SepConv3d(_Block):
    """model_utils.py:141-160."""

    def __init__(self, in_planes, out_planes, kernel_size, stride, padding=0):
        super().__init__()
        k, s, p = kernel_size, stride, padding
        self.conv_s = ConvParams(in_planes, out_planes, kernel_size=(1, k, k), stride=(1, s, s), padding=(0, p, p), bias=False)
        self.bn_s = BNParams3d(out_planes, eps=1e-3, momentum=0.001, affine=True)
        self.relu_s = _Marker("relu")
        self.conv_t = ConvParams(out_planes, out_planes, kernel_size=(k, 1, 1), stride=(s, 1, 1), padding=(p, 0, 0), bias=False)
        self.bn_t = BNParams3d(out_planes, eps=1e-3, momentum=0.001, affine=True)
        self.relu_t = _Marker("relu")
        # the 1x7x7 stride-2 RGB stem gets the W-folded kernel (input padded to 4 channels)
        self._stem = in_planes == 3 and k == 7 and p == 3
        self._cpad = 4 if self._stem else None

    def _fwd(self, ctx, x, dst=None):
        y = conv_bn_relu(ctx, self.conv_s, self.bn_s, x, stem=self._stem)
        return conv_bn_relu(ctx, self.conv_t, self.bn_t, y, dst)


# Inception widths: name -> (in, b0, b1_reduce, b1_out, b2_reduce, b2_out, b3)   model_utils.py:162-420
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


class _Mixed(_Block):
    """Four-branch Inception stage.  The branch outputs are written straight
    into channel slices of one buffer (torch.cat at model_utils.py:187 is never a
    copy) and, in training, their BN scale/shift land in slices of one vector."""
    _name = None

    def __init__(self):
        super().__init__()
        cin, b0, b1r, b1, b2r, b2, b3 = MIXED_WIDTHS[self._name]
        self.widths = (b0, b1, b2, b3)
        self.reduce = (b1r, b2r)
        self.branch0 = nn.Sequential(BasicConv3d(cin, b0, kernel_size=1, stride=1))
        self.branch1 = nn.Sequential(BasicConv3d(cin, b1r, kernel_size=1, stride=1),
                                     SepConv3d(b1r, b1, kernel_size=3, stride=1, padding=1))
        self.branch2 = nn.Sequential(BasicConv3d(cin, b2r, kernel_size=1, stride=1),
                                     SepConv3d(b2r, b2, kernel_size=3, stride=1, padding=1))
        self.branch3 = nn.Sequential(_Marker("maxpool3d", kernel_size=(3, 3, 3), stride=1, padding=1),
                                     BasicConv3d(cin, b3, kernel_size=1, stride=1))

    def _entry(self):
        """the three 1x1x1 convs that read the block input, as one conv with output channels [b1 reduce | b2 reduce | b0]"""
        mods = (self.branch1[0], self.branch2[0], self.branch0[0])
        plans = [m.conv.plan() for m in mods]
        jp = self.__dict__.get("_vinet_joint")
        if jp is None or any(a is not b for a, b in zip(jp.members, plans)):
            jp = self.__dict__["_vinet_joint"] = E.JointConvPlan(plans)
        jbn = E.JointBN([m.bn.state() for m in mods], [p.N for p in plans],
                        fold_cache=self.__dict__.setdefault("_vinet_joint_fold", {}))
        return mods, jp, jbn

    def _fwd_joint(self, ctx, x):
        """One buffer [b1 reduce | b2 reduce | b0 | b1 | b2 | b3] per voxel: the entry conv writes its three outputs
        side by side (x is read once, and in backward the three data gradients are ONE conv whose K runs over
        the same channels of the gradient buffer: x.grad is written once instead of read-modify-written three
        times); the block output is the channel slice behind the two reduce outputs."""
        xv = x.v
        b0, b1, b2, b3 = self.widths
        b1r, b2r = self.reduce
        R = b1r + b2r
        big = E.new_concat(ctx, xv.B, xv.T, xv.H, xv.W, R + b0 + b1 + b2 + b3, pending=(ctx.training or ctx.recording))
        r1, r2, cat = big.region(0, b1r), big.region(b1r, R), big.region(R, R + b0 + b1 + b2 + b3)
        entry = big.sub_chan(0, R + b0)
        entry.ready_of = [r1, r2, cat]
        o1, o2, o3 = b0, b0 + b1, b0 + b1 + b2
        # the pool branch FIRST: backward then reaches the entry conv before the pool, so the entry conv's data gradient is the
        # first writer of x.grad (a plain store: the pointwise streaming kernel, conv_pw.h) and the pool backward accumulates
        fork = ctx.branch_streams(xv.B * xv.T * xv.H * xv.W, xv.B)
        if (fork is not None and ctx.recording and ctx.training and E.BRANCH_STREAMS_BWD and E.PARAM_GRAD_HOOK is None and
                xv.B >= E.BRANCH_STREAMS_BWD_MIN_BATCH):
            return self._fwd_joint_forked_train(ctx, x, fork, big, r1, r2, cat, entry)
        if fork is not None:      # small batches: the branches run side by side (engine.BRANCH_STREAMS_VOX / _TRAIN_VOX)
            main = torch.cuda.current_stream(ctx.device)
            fork[1].wait_stream(main)
            with ctx.on_stream(fork[1]):
                pooled = E.maxpool_forward(ctx, x, (3, 3, 3), (1, 1, 1), (1, 1, 1))
                self.branch3[1]._fwd(ctx, pooled, cat.sub_chan(o3, o3 + b3))
                del pooled
            mods, jp, jbn = self._entry()
            E.conv_forward(ctx, jp, x, bn=jbn, act=L.ACT_RELU, dst=entry)
            if ctx.training:
                for m in mods:
                    m.bn.note_training_step()
                self.__dict__["_vinet_joint_fold"].clear()
            fork[0].wait_stream(main)
            if E.BRANCH_STREAMS_SWAP:
                with ctx.on_stream(fork[0]):
                    self.branch1[1]._fwd(ctx, r1, cat.sub_chan(o1, o2))
                self.branch2[1]._fwd(ctx, r2, cat.sub_chan(o2, o3))
            else:
                with ctx.on_stream(fork[0]):
                    self.branch2[1]._fwd(ctx, r2, cat.sub_chan(o2, o3))
                self.branch1[1]._fwd(ctx, r1, cat.sub_chan(o1, o2))
            main.wait_stream(fork[0])
            main.wait_stream(fork[1])
            return cat
        pooled = E.maxpool_forward(ctx, x, (3, 3, 3), (1, 1, 1), (1, 1, 1))
        self.branch3[1]._fwd(ctx, pooled, cat.sub_chan(o3, o3 + b3))
        mods, jp, jbn = self._entry()
        E.conv_forward(ctx, jp, x, bn=jbn, act=L.ACT_RELU, dst=entry)
        if ctx.training:
            for m in mods:
                m.bn.note_training_step()
            self.__dict__["_vinet_joint_fold"].clear()
        self.branch1[1]._fwd(ctx, r1, cat.sub_chan(o1, o2))
        self.branch2[1]._fwd(ctx, r2, cat.sub_chan(o2, o3))
        return cat

    def _fwd_joint_forked_train(self, ctx, x, fork, big, r1, r2, cat, entry):
        """Small-batch training: the branches side by side in the forward AND in the backward pass.  The tape is run in reverse,
        so the forward records, between the branches' nodes, markers that switch the backward pass's stream: backward runs
        branch 2 on the main stream, branch 1 on fork[0] and branch 3's conv on fork[1] (each chain = BatchNorm backward + data
        gradients, their weight gradients leave for the weight-gradient stream from whichever stream made dy), joins, then the
        entry conv (the first writer of x.grad) and the pool's backward (which accumulates into it) on the main stream as before.
        The branches write disjoint slices of the concat gradient's reduce regions and disjoint parameter gradients.  Under
        capture the markers do nothing (the captured step groups its weight-gradient joins: engine.WGRAD_GROUP_CAPTURE)."""
        b0, b1, b2, b3 = self.widths
        o1, o2, o3 = b0, b0 + b1, b0 + b1 + b2
        dev = ctx.device

        def m_begin():
            if not ctx.capturing:
                main = torch.cuda.current_stream(dev)
                fork[0].wait_stream(main)
                fork[1].wait_stream(main)

        def m_enter(k):
            def f():
                if not ctx.capturing:
                    ctx._bwd_cm = ctx.on_stream(fork[k])
                    ctx._bwd_cm.__enter__()
            return f

        def m_exit():
            if not ctx.capturing:
                ctx._bwd_cm.__exit__(None, None, None)
                ctx._bwd_cm = None

        def m_join():
            if not ctx.capturing:
                main = torch.cuda.current_stream(dev)
                main.wait_stream(fork[0])
                main.wait_stream(fork[1])

        main = torch.cuda.current_stream(dev)
        fork[1].wait_stream(main)
        with ctx.on_stream(fork[1]):
            pooled = E.maxpool_forward(ctx, x, (3, 3, 3), (1, 1, 1), (1, 1, 1))
        mods, jp, jbn = self._entry()
        E.conv_forward(ctx, jp, x, bn=jbn, act=L.ACT_RELU, dst=entry)
        for m in mods:
            m.bn.note_training_step()
        self.__dict__["_vinet_joint_fold"].clear()
        # (recorded in the reverse of the order the backward pass meets them)
        ctx.record(m_join)
        ctx.record(m_exit)
        with ctx.on_stream(fork[1]):
            self.branch3[1]._fwd(ctx, pooled, cat.sub_chan(o3, o3 + b3))
            del pooled
        ctx.record(m_enter(1))
        ctx.record(m_exit)
        fork[0].wait_stream(main)
        with ctx.on_stream(fork[0]):
            self.branch1[1]._fwd(ctx, r1, cat.sub_chan(o1, o2))
        ctx.record(m_enter(0))
        self.branch2[1]._fwd(ctx, r2, cat.sub_chan(o2, o3))
        ctx.record(m_begin)
        main.wait_stream(fork[0])
        main.wait_stream(fork[1])
        return cat

    def _fwd(self, ctx, x, dst=None):
        # (eval-mode BN under autograd keeps per-layer running statistics as the saved mean: per-conv path)
        if dst is None and E.JOINT_ENTRY and (ctx.training or not ctx.recording):
            return self._fwd_joint(ctx, x)
        xv = x.v
        b0, b1, b2, b3 = self.widths
        cat = dst if dst is not None else E.new_concat(ctx, xv.B, xv.T, xv.H, xv.W, b0 + b1 + b2 + b3,
                                                       pending=(ctx.training or ctx.recording))
        o1, o2, o3 = b0, b0 + b1, b0 + b1 + b2
        self.branch0[0]._fwd(ctx, x, cat.sub_chan(0, o1))
        self.branch1[1]._fwd(ctx, self.branch1[0]._fwd(ctx, x), cat.sub_chan(o1, o2))
        self.branch2[1]._fwd(ctx, self.branch2[0]._fwd(ctx, x), cat.sub_chan(o2, o3))
        pooled = E.maxpool_forward(ctx, x, (3, 3, 3), (1, 1, 1), (1, 1, 1))
        self.branch3[1]._fwd(ctx, pooled, cat.sub_chan(o3, o3 + b3))
        return cat


def _mixed(name):
    return type("Mixed_" + name, (_Mixed,), {"_name": name, "__doc__": "model_utils.py Mixed_%s" % name})


Mixed_3b, Mixed_3c = _mixed("3b"), _mixed("3c")
Mixed_4b, Mixed_4c, Mixed_4d, Mixed_4e, Mixed_4f = _mixed("4b"), _mixed("4c"), _mixed("4d"), _mixed("4e"), _mixed("4f")
Mixed_5b, Mixed_5c = _mixed("5b"), _mixed("5c")
