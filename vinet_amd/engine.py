"""Host-side engine of the MI355X ViNet path.

Everything a module forward/backward does on the device goes through this file
as calls into libvinet_hip.so (vinet_amd/_lib.py): there are no torch compute
kernels on the path.  torch is used for device memory (caching allocator),
streams, autograd *entry points* and torch.distributed.

Concepts
  View  channels-last 5-D view [B][T][H][W][C] of a flat torch buffer, with W
        stride `ld` (channel slices of a concat buffer) and batch stride `sB`
        (T slices of a longer buffer).
  Act   a View plus a *pending* per-channel affine(+ReLU).  In training mode a
        conv stores its raw output and BatchNorm's scale/shift only exist once
        the whole tensor has been reduced, so BN+ReLU is applied by whichever
        kernel loads the tensor next (conv / wgrad / pool loaders) instead of in
        a separate pass over HBM.  `Act.grad` is always the gradient w.r.t. the
        transformed value.
  Tape  list of backward closures recorded during a training forward; run in
        reverse by the single autograd node that wraps a root module call.
"""
import ctypes as C
import itertools
import math
import os
import weakref

import torch

from . import _lib as L

F32, BF16 = L.F32, L.BF16
F32S = L.F32S        # fp32 tensors + split-bf16 matrix arithmetic: a COMPUTE mode of the fp32 storage type (Ctx.cdt), never a View dtype
TORCH_DT = {F32: torch.float32, BF16: torch.bfloat16}
ESIZE = {F32: 4, BF16: 2}
EG = {F32: 4, BF16: 8}          # elements per 16 bytes

_DEFAULT_DTYPE = BF16
_WEIGHTS_EPOCH = 0
_UNPACK_TABLES = {}
_SIDE_STREAMS = {}

# ----------------------------------------------------------------------------
# configuration: ONE explicit entry point, engine.configure(**kw).  Nothing here reads the environment (the only
# environment variable of the package is VINET_LIB, the path of another build of the library: _lib.py); tools and
# bench.py pass `--cfg name=value,...` through configure().  The values live as module attributes (tests monkeypatch
# them); config() returns the current set, which bench.py records in its JSON line.
# ----------------------------------------------------------------------------
_CONFIG = dict(
    # weight gradients are off the backward critical path (only the optimizer consumes them): they run on a second
    # HIP stream, concurrently with the data-gradient / BatchNorm-backward chain
    WGRAD_SIDE_STREAM=True,
    # CUs the persistent weight-gradient kernels may occupy while they run beside the main stream (VinetWgradDesc::max_cus,
    # per launch); without a second stream they are alone on the GPU and get all of it
    WGRAD_CUS=208,
    WGRAD_CUS_DEC=208,          # ... the decoder's (BatchNorm-less, deferred) weight gradients
    TAIL_WGRAD_FULL=True,       # the tape's last weight gradient (the RGB stem) takes the whole chip
    # weight-gradient jobs per join with the main stream (1 = a join per job), eager / under capture.  Replayed step: 1: 381 / 546
    # clips/s at 8 / 32 clips, 16: 410 / 583, 64: 367 / 549 -- every fork costs the replay ~12 us (profiles/r4_experiments.txt)
    WGRAD_GROUP=1,
    WGRAD_GROUP_CAPTURE=16,
    # every packed weight gradient of a backward pass unpacked by ONE launch at its end (0 = one vinet_unpack_wgrad per conv).
    # Not used while a parameter-gradient hook is installed (the bucketed all-reduce wants each gradient as soon as it is final).
    MULTI_UNPACK=1,
    N_SIDE_STREAMS=1,           # weight-gradient streams the jobs are dealt over round robin
    # Inference at small batches: the four branches of an Inception stage are independent kernels whose grids cannot fill 256
    # CUs (batch 1: 3 .. 170 workgroups), so below this many input voxels a stage forks them over two more streams and joins at
    # the concat.  Only under capture by default (an eager batch-1 forward is bound by the host's launch rate: 307 -> 263 fps
    # eager; 586 -> 645 fps replayed); BRANCH_STREAMS_EAGER forks in eager too.  0 = never.
    BRANCH_STREAMS_VOX=65536,
    BRANCH_STREAMS_EAGER=False,
    # the training forward forks the same way below this many voxels, at small batches only (4 clips +5 %, 8 clips +1.8 %, neutral
    # at 32; at 192 clips the last two stages would qualify and the step loses 0.15 %, hence the batch bound)
    BRANCH_STREAMS_TRAIN_VOX=800000,
    BRANCH_STREAMS_TRAIN_BATCH=32,
    # ... and in the backward pass (tape markers switch its stream: model_utils._Mixed._fwd_joint_forked_train); eager only, from
    # 8 clips on: +1..1.5 %.  On since round 5: the mismatch that kept it opt-in in round 4 was not the forks' (spin stress on both
    # sides of every fork passes; the soak reproduced the mismatch on ONE stream; root cause: a packed-fp32 erratum, csrc/common.h).
    BRANCH_STREAMS_BWD=True,
    BRANCH_STREAMS_BWD_MIN_BATCH=8,
    # which branch leaves the capturing stream: the longer chain (branch 1) forks (690 -> 737 fps at batch 1); False = branch 2
    BRANCH_STREAMS_SWAP=True,
    STEM_FOLD=True,             # padded / folded RGB stem input (the streaming stem kernels need it)
    JOINT_ENTRY=1,              # Inception blocks run their three input-side 1x1x1 convs as one (model_utils._Mixed._fwd_joint)
    BN_BWD_FUSE=1,              # the stem's BN-backward apply pass folded into its weight-gradient kernel
    SHARE_SKIP_GRAD=True,       # a decoder skip's gradient lives in the T-concat's gradient (no copy in backward)
    # weight gradients of the BN-free (decoder) convs wait until the tape reaches the encoder (+0.5 % on the step at 192 clips)
    DEFER_DECODER_WGRAD=1,
    DEFER_DECODER_WGRAD_F32S=0,  # fp32s: the weight-gradient stream is the longer one there (207.3 -> 212.5 clips/s at 64 clips)
    PERSISTENT_DW=1,            # packed weight-gradient workspaces owned by the conv plans, re-zeroed by the unpack kernel
    # consumer-side BN costs the MFMA kernels 20-27 %; writing relu(bn(x)) out once costs two passes over x: materialise the
    # input of a conv when N * taps is at least this (0 = never; whole step: never 478 clips/s, 400: 483, 800: 539, 2500: 485)
    MATERIALIZE_NT=800,
    SPLIT_WGRAD_BF16=1,         # fp32s weight gradient as three launches of the bf16 kernels over hi / lo planes
    SPLIT_IN_APPLY=1,           # fp32s: the planes of dy leave with the BatchNorm-backward apply pass that produces dy
    # a data gradient that is the LAST writer of the gradient behind BatchNorm + ReLU layers also writes their backward partial
    # sums (VinetConvDesc::bnb_*): no reduce pass over (dz, z) for those layers.  0 = off, 1 = the stem's fused temporal data
    # gradient only (round 4), 2 = every data gradient the library can do it for (round 5: the shared conv epilogue, conv_bnb.hip).
    # Default 1: mode 2 is correct (bit-exact kernel tests, goldens) and removes 16 of the 58 reduce launches (-8 ms of reduce
    # time in the step), but the 32 data gradients that carry the sums get 7.7 ms slower alone (+z read, +8 VALU per element in
    # an issue-bound epilogue): 282.8 -> 286.7 ms per step in alternating same-box runs (profiles/r5_bnb_epilogue_ab.txt)
    DGRAD_BN_STATS=1,
    UPSAMPLE_BWD_RELU=1,        # ReLU backward of conv -> ReLU -> upsample inside the upsample's backward pass
    PARAM_GRAD_MODE="fused",    # see set_param_grad_mode
    # Schedule stress (tests): shader clocks a spin kernel idles on the weight-gradient stream in front of every weight-gradient
    # launch (DBG_SPIN_SIDE) / on the main stream in front of every data gradient (DBG_SPIN_MAIN) / on a branch stream each time
    # a fork enters it (DBG_SPIN_FORK; negative: on the forking stream instead, so the branch streams run ahead).  A result
    # that depends on the streams' relative timing shows up as a mismatch against the one-stream schedule.
    DBG_SPIN_SIDE=0,
    DBG_SPIN_MAIN=0,
    DBG_SPIN_FORK=0,
    # "vinet_conv3d_wgrad,vinet_bn_bwd_reduce,tag:dgrad": entry points (or call-site tags) whose launches are SKIPPED -- what a
    # kernel family costs in the step once the overlap of the two streams is taken into account (tools/ablate.sh).  Results are
    # garbage; the step time is the point.
    ABLATE="",
)
globals().update(_CONFIG)
_ABLATE, _ABLATE_TAGS = set(), []


def configure(**kw):
    """Set engine options (names: the keys of engine._CONFIG, upper or lower case).  Returns the previous values of the keys
    given, so `old = configure(x=1) ... configure(**old)` restores them."""
    global _ABLATE, _ABLATE_TAGS
    old = {}
    for k, v in kw.items():
        K = k.upper()
        if K not in _CONFIG:
            raise KeyError("engine.configure: unknown option %r (known: %s)" % (k, ", ".join(sorted(_CONFIG))))
        d = _CONFIG[K]
        if isinstance(d, bool):
            v = (v not in ("0", "false", "False", "")) if isinstance(v, str) else bool(v)
        elif isinstance(d, int):
            v = int(v)
        elif isinstance(d, str):
            v = str(v)
        old[K] = globals()[K]
        globals()[K] = v
        if K == "PARAM_GRAD_MODE":
            set_param_grad_mode(v)
        if K == "ABLATE":
            items = [a for a in v.split(",") if a]
            _ABLATE = set(a for a in items if not a.startswith("tag:"))
            _ABLATE_TAGS = [a[4:] for a in items if a.startswith("tag:")]
            if _ABLATE_TAGS:
                _ABLATE.add("\0tags")
    return old


def configure_from_string(text):
    """'name=value,name=value' (bench.py --cfg, tools): engine options, and `lib.<option>=<int>` for vinet_set_option"""
    kw = {}
    for item in filter(None, (text or "").split(",")):
        k, _, v = item.partition("=")
        k = k.strip()
        if k.startswith("lib."):
            L.set_option(k[4:], int(v))
        elif k.upper() == "ABLATE":
            kw["ABLATE"] = v.replace("+", ",")          # ('+' separates the entries inside one --cfg value)
        else:
            kw[k] = v
    return configure(**kw)


def config(changed_only=False):
    """current engine options (+ the library options set through _lib.set_option)"""
    cur = {k: globals()[k] for k in _CONFIG}
    if changed_only:
        cur = {k: v for k, v in cur.items() if v != _CONFIG[k]}
    if L.LIB_OPTIONS:
        cur["lib"] = dict(L.LIB_OPTIONS)
    return cur


def set_default_dtype(name):
    """'bf16' (throughput path, default), 'fp32' (exact parity path: fp32 tensors, fp32 MFMA) or 'fp32s' (fp32 tensors, convs on
    three bf16 MFMAs per product over hi / lo halves of both operands: the fast configuration that meets the 1e-3 contract)."""
    global _DEFAULT_DTYPE
    _DEFAULT_DTYPE = {"bf16": BF16, "bfloat16": BF16, "fp32": F32, "float32": F32, "fp32s": F32S, "fp32_split": F32S}[str(name).replace("torch.", "")]


def default_dtype():
    return _DEFAULT_DTYPE


def bump_weights_epoch():
    """Called by anything that rewrites parameters behind torch's back (the fused
    Adam kernel): invalidates every cached weight pack."""
    global _WEIGHTS_EPOCH
    _WEIGHTS_EPOCH += 1


def rup(a, b):
    return (a + b - 1) // b * b


class Profiler:
    """HIP-event timing of library calls on torch's current stream (which is the
    stream every kernel is launched on).  `only` restricts bracketing to a set of
    call-site keys so a timed benchmark region can measure ONE kernel site with two
    event records per launch and nothing else."""

    def __init__(self, only=None):
        self.only = None if only is None else set(only)
        self.records = []

    def want(self, key):
        return self.only is None or key in self.only

    def summary(self):
        torch.cuda.synchronize()
        agg = {}
        for key, e0, e1, work in self.records:
            a = agg.setdefault(key, dict(count=0, ms=0.0, work=work))
            a["count"] += 1
            a["ms"] += e0.elapsed_time(e1)
        return agg


PROFILER = None
LAUNCH_LOG = None      # tests: a list that receives (entry point, stream, tag) of every launch issued through Ctx.call


def set_profiler(p):
    global PROFILER
    PROFILER = p


def _stream_for(device):
    if device.type == "cuda":
        return torch.cuda.current_stream(device).cuda_stream
    if not L.is_test_double():
        raise RuntimeError("vinet_amd runs on MI355X only: tensor on %s and no HIP device (no CPU fallback)" % device)
    return 0


def _ptr(t):
    return t.data_ptr() if t is not None else None


class View:
    __slots__ = ("buf", "off", "B", "T", "H", "W", "C", "ld", "sB", "dt")

    def __init__(self, buf, off, B, T, H, W, Cc, ld, sB, dt):
        self.buf, self.off, self.B, self.T, self.H, self.W, self.C, self.ld, self.sB, self.dt = buf, off, B, T, H, W, Cc, ld, sB, dt

    @staticmethod
    def alloc(B, T, H, W, Cc, dt, device, zero=False):
        # (dense rows.  Rows padded to 128-byte lines were tried: the microbenchmark tools/ubench/l2_to_lds moves 8 rows x 128 B
        #  per LDS-DMA at 33 B/clk/CU from 960-byte rows against 55-65 from line-aligned ones, but no layer of the real step got
        #  faster, and the switch is gone.)
        ld = Cc
        n = B * T * H * W * ld
        buf = (torch.zeros if zero else torch.empty)(n, dtype=TORCH_DT[dt], device=device)
        return View(buf, 0, B, T, H, W, Cc, ld, T * H * W * ld, dt)

    @property
    def device(self):
        return self.buf.device

    @property
    def nvox(self):
        return self.B * self.T * self.H * self.W

    def ptr(self):
        return self.buf.data_ptr() + self.off * ESIZE[self.dt]

    def ct(self):
        return L.CTensor(self.ptr(), self.B, self.T, self.H, self.W, self.C, self.ld, self.sB)

    def chan(self, c0, c1):
        assert 0 <= c0 < c1 <= self.C
        return View(self.buf, self.off + c0, self.B, self.T, self.H, self.W, c1 - c0, self.ld, self.sB, self.dt)

    def tslice(self, t0, t1):
        assert 0 <= t0 < t1 <= self.T
        return View(self.buf, self.off + t0 * self.H * self.W * self.ld, self.B, t1 - t0, self.H, self.W, self.C, self.ld, self.sB, self.dt)

    def quads(self):
        """dense view reinterpreted as [1,1,1,n/4,4] (C==1 tensors for quad kernels)."""
        assert self.ld == self.C and self.sB == self.T * self.H * self.W * self.C
        n = self.nvox * self.C
        assert n % 4 == 0
        return View(self.buf, self.off, 1, 1, 1, n // 4, 4, 4, n, self.dt)

    def torch5(self):
        """[B,T,H,W,C] strided torch view (tests / debugging)."""
        return torch.as_strided(self.buf, (self.B, self.T, self.H, self.W, self.C),
                                (self.sB, self.H * self.W * self.ld, self.W * self.ld, self.ld, 1), self.off)

    def same_dims(self, o):
        return (self.B, self.T, self.H, self.W, self.C) == (o.B, o.T, o.H, o.W, o.C)


class Act:
    """activation = view + pending affine/relu + gradient bookkeeping."""
    __slots__ = ("v", "scale", "shift", "relu", "_grad", "grad_ready", "needs_grad", "parent", "pc0", "pt0", "fold",
                 "indep", "ready_of", "grad_marks", "mean", "invstd", "alias_of", "n_readers", "act_out", "grad_masked")

    def __init__(self, v, scale=None, shift=None, relu=False, needs_grad=False, mean=None, invstd=None):
        self.v, self.scale, self.shift, self.relu = v, scale, shift, relu
        # batch statistics of the training-mode BatchNorm(s) behind the pending affine (per channel, fp32; slices follow the
        # channel slices like scale / shift): what a consumer's data gradient needs to fold the BatchNorm-backward reduce pass in
        self.mean, self.invstd = mean, invstd
        self._grad, self.grad_ready, self.needs_grad = None, False, needs_grad
        self.parent, self.pc0, self.pt0 = None, 0, 0
        self.fold = None
        self.indep = False       # channel region with its own "gradient written" flag (see region())
        self.ready_of = None     # span over several regions: ready when all of them are
        self.grad_marks = 0      # writers of this gradient so far (mark_grad_ready calls on the root)
        self.alias_of = None     # the pending activation this plain one materialises with shared gradient storage (materialize)
        self.n_readers = 0       # consumers recorded in this forward that will write this gradient (counted on the root)
        self.act_out = 0         # activation the producing conv applied in its epilogue (no BatchNorm): conv_forward
        self.grad_masked = -1    # grad_marks at which the gradient already carries that activation's backward (upsample2x backward)

    @property
    def plain(self):
        return self.scale is None and not self.relu

    def affine(self):
        return L.CAffine(_ptr(self.scale), _ptr(self.shift), 1 if self.relu else 0)

    def sub_chan(self, c0, c1):
        """channel slice sharing gradient storage with self (concat member)."""
        a = Act(self.v.chan(c0, c1), None if self.scale is None else self.scale[c0:c1],
                None if self.shift is None else self.shift[c0:c1], self.relu, self.needs_grad,
                None if self.mean is None else self.mean[c0:c1], None if self.invstd is None else self.invstd[c0:c1])
        a.parent, a.pc0, a.pt0 = self, c0, None
        return a

    def region(self, c0, c1):
        """channel slice that shares gradient STORAGE with self but is written by its own set of consumers:
        the first of them stores, the rest accumulate, independently of the neighbouring regions (the fused
        Inception entry conv writes [b1 reduce | b2 reduce | b0] side by side with the concat output)."""
        a = self.sub_chan(c0, c1)
        a.indep = True
        return a

    def sub_t(self, t0, t1):
        a = Act(self.v.tslice(t0, t1), self.scale, self.shift, self.relu, self.needs_grad, self.mean, self.invstd)
        a.parent, a.pt0, a.pc0 = self, t0, None
        return a

    def root(self):
        """owner of the gradient-ready flag"""
        a = self
        while a.parent is not None and not a.indep:
            a = a.parent
        return a

    # ---- gradients ------------------------------------------------------
    def grad_view(self, dt=None, zero=False):
        """gradient storage (allocated on first use; slices resolve into the parent's)."""
        if self.parent is not None:
            if self.indep and zero:
                top = self.parent
                while top.parent is not None:
                    top = top.parent
                assert top._grad is None, "zero-filled gradient requested for a region of live shared storage"
            g = self.parent.grad_view(dt, zero)
            if self.pt0 is None:
                return g.chan(self.pc0, self.pc0 + self.v.C)
            return g.tslice(self.pt0, self.pt0 + self.v.T)
        if self._grad is None:
            v = self.v
            self._grad = View.alloc(v.B, v.T, v.H, v.W, v.C, v.dt if dt is None else dt, v.device, zero=zero)
            if zero:
                self.grad_ready = True
        return self._grad

    def is_grad_ready(self):
        if self.ready_of is not None:
            return all(a.is_grad_ready() for a in self.ready_of)
        return self.root().grad_ready

    def mark_grad_ready(self):
        r = self.root()
        r.grad_ready = True
        r.grad_marks += 1


def _abs_chan(a):
    """(storage owner, absolute channel offset) of an activation inside the buffer that owns its gradient storage; None when a
    T slice lies on the way (decoder concat members: no BatchNorm there)"""
    off = 0
    while a.parent is not None:
        if a.pc0 is None:
            return None
        off += a.pc0
        a = a.parent
    return a, off


def _note_reader(ctx, x):
    """forward: one more consumer of `x` whose backward will write x's gradient"""
    if ctx.recording and x.needs_grad:
        x.root().n_readers += 1


def _register_bnb(ctx, x, ws, rows):
    """a data gradient just wrote x's gradient (already marked) together with the BatchNorm-backward partial sums `ws`
    ([rows][2][x.C]) of the layer(s) behind x's pending affine: remember them with the writer count of that moment -- they are
    valid as long as no later writer touches the gradient (_find_bnb checks)"""
    loc = _abs_chan(x)
    if loc is None:
        return
    owner, off = loc
    r = x.root()
    # (kept on the pass, not on the activation: an Act that referenced its own root would be a reference cycle, and a step's
    #  activations must die by reference count when the tape is dropped -- 190 GB of them at the headline batch)
    ctx._bnb.setdefault(id(owner), []).append((off, off + x.v.C, ws, rows, r, r.grad_marks))


def _find_bnb(ctx, res, c0, width):
    """partial sums covering channels [c0, c0 + width) of `res` left by the LAST writer of that gradient: (ws, rows, column offset,
    row width) or None"""
    loc = _abs_chan(res)
    if loc is None or id(loc[0]) not in ctx._bnb:
        return None
    owner, off = loc
    a0, a1 = off + c0, off + c0 + width
    for r0, r1, ws, rows, root, marks in reversed(ctx._bnb[id(owner)]):
        if r0 <= a0 and a1 <= r1 and root.grad_marks == marks:
            return ws, rows, a0 - r0, r1 - r0
    return None


class Ctx:
    """one forward(/backward) pass: dtype, mode, stream, tape."""

    def __init__(self, device, dt=None, training=False, record=False):
        self.lib = L.get()
        self.device = device
        dt = _DEFAULT_DTYPE if dt is None else dt
        self.dt = F32 if dt == F32S else dt       # storage type of activations / packed weights
        self.cdt = dt                             # what the conv / weight-gradient descriptors carry (F32S: split-bf16 arithmetic)
        self.training = training
        self.tape = [] if record else None
        self.stream = _stream_for(device)
        self._side_keep = []
        self._deferred = []
        self._unpack_jobs = []
        self._dw_plans = []
        self._bnb = {}           # id(storage owner) -> [(c0, c1, partials, rows, root, marks)]: BN-backward partial sums that data gradients left
        self.capturing = False

    @property
    def recording(self):
        return self.tape is not None

    def f32(self, n, zero=False):
        return (torch.zeros if zero else torch.empty)(n, dtype=torch.float32, device=self.device)

    def call(self, name, *args, tag=None, work=None):
        if LAUNCH_LOG is not None:
            LAUNCH_LOG.append((name, args[-1] if args else None, tag))
        if _ABLATE and (name in _ABLATE or (tag is not None and any(a in tag for a in _ABLATE_TAGS))):
            return          # tuning only (VINET_ABLATE): the launch is skipped, results are garbage, the step time is the point
        prof = PROFILER
        if prof is not None and self.device.type == "cuda" and prof.want(tag or name):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = getattr(self.lib, name)(*args)
            e1.record()
            prof.records.append((tag or name, e0, e1, work))
        else:
            rc = getattr(self.lib, name)(*args)
        if rc != 0:
            L.check(rc, name)

    def record(self, fn):
        if self.tape is not None:
            self.tape.append(fn)

    def side_stream(self, k=0):
        """k-th weight-gradient stream of this device (None on the CPU test double)"""
        if not WGRAD_SIDE_STREAM or self.device.type != "cuda":
            return None
        st = _SIDE_STREAMS.get((self.device.index, k))
        if st is None:
            st = _SIDE_STREAMS[(self.device.index, k)] = torch.cuda.Stream(self.device)
        return st

    def branch_streams(self, nvox, batch=0):
        """the two extra streams an Inception stage forks its branches over in small-batch inference (None: run in order)"""
        if self.device.type != "cuda":
            return None
        if self.training or self.recording:
            # the training forward (its tape order is the Python order either way; backward runs on the main stream, behind the
            # joins): GPU-bound from 4 clips on, so eager forks too
            if nvox > BRANCH_STREAMS_TRAIN_VOX or batch > BRANCH_STREAMS_TRAIN_BATCH:
                return None
        else:
            if nvox > BRANCH_STREAMS_VOX:
                return None
            if not BRANCH_STREAMS_EAGER and not torch.cuda.is_current_stream_capturing():
                return None
        out = []
        for k in ("b1", "b2"):
            st = _SIDE_STREAMS.get((self.device.index, k))
            if st is None:
                st = _SIDE_STREAMS[(self.device.index, k)] = torch.cuda.Stream(self.device)
            out.append(st)
        return out

    def on_stream(self, st):
        """with ctx.on_stream(st): launches (and torch allocations) go to `st`"""
        return _OnStream(self, st)

    def side_streams(self):
        """every weight-gradient stream in use (N_SIDE_STREAMS of them; empty without a GPU)"""
        if self.side_stream() is None:
            return []
        return [self.side_stream(k) for k in range(N_SIDE_STREAMS)]

    def keep(self, *tensors):
        """hold tensors that a side-stream kernel reads or writes until run_backward has joined the streams: the caching
        allocator would otherwise hand a block released by the main stream's Python frame to the next main-stream
        allocation while the side kernel is still using it"""
        self._side_keep.extend(t for t in tensors if t is not None)

    def flush_deferred(self):
        """launch the deferred weight-gradient jobs on the weight-gradient stream behind ONE join with the main stream.

        One join for the whole batch is not only cheaper: a join per job (event record on the main stream + wait on the side
        stream, with no main-stream launch in between) gives the main stream's last node one outgoing edge per job, and a
        captured step with such a fan-out REPLAYS WRONG on ROCm 7.2 -- the main stream's next kernel runs before that node
        has finished, although hipGraphDebugDotPrint shows the edge (the two graphs' edge sets differ only by the redundant
        fan-out edges; tools/dot_edges.py, tools/repro_graph_fanout.py, profiles/r4_capture_fanout.txt).
        Eager execution of the same launches is correct under any relative timing of the two streams (spin-kernel stress)."""
        jobs, self._deferred = self._deferred, []
        if not jobs:
            return
        side = self.side_stream()
        if side is not None:
            # (every weight-gradient stream the jobs are dealt over: a job on stream k >= 1 skips its own join when _joined is set)
            cur = torch.cuda.current_stream(self.device)
            for st in self.side_streams():
                st.wait_stream(cur)
            self._joined = True
        try:
            for job in jobs:
                job()
        finally:
            self._joined = False

    def flush_unpack(self):
        """the packed weight gradients collected during this backward -> `.grad`, ONE launch behind the last weight-gradient
        kernel (on the weight-gradient stream); the device-side job table is cached per job list (pointers are stable:
        persistent workspaces, gradient views of the optimizer's flat buffer)"""
        jobs, self._unpack_jobs = getattr(self, "_unpack_jobs", []), []
        if not jobs:
            return
        # A plan that ran twice in this backward (a module called twice in one forward) queued its job twice, but its
        # persistent workspace already holds the SUM of both weight gradients: one job per workspace, or two thread
        # groups of the launch would race on `grad += dw; dw = 0` over the same rows.
        seen, uniq = set(), []
        for j in jobs:
            if j[0] not in seen:
                seen.add(j[0])
                uniq.append(j)
        jobs = uniq
        key = tuple(jobs)
        ent = _UNPACK_TABLES.get(key)
        if ent is None:
            # (never evicted: a captured step has the table's address baked into its launch, and a table is a few KB)
            assert not self.capturing, "weight-gradient unpack table built inside a stream capture (warm the step up first)"
            rows, off = [], 0
            for dw, gw, N, Cin, ntaps, stem, numel in jobs:
                rows.append([dw, gw, N, Cin, ntaps, stem, off, 0])
                off += numel
            rows.append([0, 0, 0, 0, 0, 0, off, 0])
            ent = _UNPACK_TABLES[key] = (torch.tensor(rows, dtype=torch.int64).to(self.device), off)
        table, total = ent
        side = self.side_stream()
        main_ptr = self.stream
        with (torch.cuda.stream(side) if side is not None else _NullCtx()):
            if side is not None:
                self.stream = side.cuda_stream
            try:
                self.call("vinet_unpack_wgrad_multi", table.data_ptr(), len(jobs), total, 3, self.stream)
            finally:
                self.stream = main_ptr

    def run_backward(self):
        self.side_used = False
        self._side_keep = []
        self._deferred = []
        self._unpack_jobs = []
        self.capturing = self.device.type == "cuda" and torch.cuda.is_current_stream_capturing()
        self._tape_left = len(self.tape)         # nodes still to run, the current one included
        self._dw_plans = []
        try:
            for fn in reversed(self.tape):
                fn()
                self._tape_left -= 1
            self.flush_deferred()
            self.flush_unpack()
        except BaseException:
            # every weight-gradient kernel ADDS into its plan's persistent workspace, which only the unpack launch hands back
            # zeroed: a backward pass that dies between the two would leave a residue that every later step silently adds to
            # its gradient.  Drop the workspaces this pass touched (the next use allocates zero-filled ones).
            # The weight-gradient kernels launched so far may still be RUNNING on the side streams, and the workspaces were
            # allocated on the main stream: join them first, or the caching allocator could hand the blocks to the next
            # main-stream allocation while atomics still land in them.  (Not inside a capture: synchronising would invalidate it,
            # and a failed capture is discarded as a whole anyway.)
            if self.device.type == "cuda" and not self.capturing:
                try:
                    for st in self.side_streams():
                        torch.cuda.current_stream(self.device).wait_stream(st)
                    torch.cuda.current_stream(self.device).synchronize()
                except Exception:
                    pass
            for plan in self._dw_plans:
                plan._dw_ws.clear()
            self._deferred, self._unpack_jobs = [], []
            self._side_keep, self._bnb, self.tape = [], {}, []
            raise
        if getattr(self, "side_used", False):
            # the optimizer (and every buffer release that follows) is ordered after the side stream
            for st in self.side_streams():
                torch.cuda.current_stream(self.device).wait_stream(st)
        self._side_keep = []
        self._bnb = {}
        self.tape = []


# ----------------------------------------------------------------------------
# import / export at module boundaries
# ----------------------------------------------------------------------------

def import_ncdhw(ctx, t, cpad=None, needs_grad=False):
    """fp32 NCDHW-logical torch tensor (any strides) -> Act (channels-last, ctx.dt)."""
    assert t.dim() == 5 and t.dtype == torch.float32, "expected fp32 [B,C,T,H,W]"
    B, Cc, T, H, W = t.shape
    cp = rup(Cc, EG[ctx.dt]) if cpad is None else cpad
    v = View.alloc(B, T, H, W, cp, ctx.dt, t.device)
    sb, sc, st, sh, sw = t.stride()
    ctx.call("vinet_import_ncdhw", t.data_ptr(), sb, sc, st, sh, sw, Cc, C.byref(v.ct()), ctx.dt, ctx.stream)
    a = Act(v, needs_grad=needs_grad)
    return a


def import_video_folded(ctx, t):
    """RGB clip [B,3,T,H,W] (any strides, no gradient needed) -> Act over the *overlapped* view
    [B][T][H+6][(W+8)/2][C=32], ld = 8, of a zero-padded 4-channel buffer: the form in which the
    1x7x7 stride-2 stem is a generic 7-tap conv for the LDS-DMA kernels (see vinet_import_ncdhw_pad)."""
    assert t.dim() == 5 and t.dtype == torch.float32 and t.shape[1] == 3
    B, Cc, T, H, W = t.shape
    Hp, Wp = H + 6, W + 8 + (W & 1)
    n = B * T * Hp * Wp * 4
    buf = torch.empty(n + 64, dtype=TORCH_DT[ctx.dt], device=t.device)   # slack: the last positions' rows overhang
    padded = View(buf, 0, B, T, Hp, Wp, 4, 4, T * Hp * Wp * 4, ctx.dt)
    sb, sc, st, sh, sw = t.stride()
    ctx.call("vinet_import_ncdhw_pad", t.data_ptr(), sb, sc, st, sh, sw, Cc, H, W, 3, 3, C.byref(padded.ct()), ctx.dt, ctx.stream)
    a = Act(View(buf, 0, B, T, Hp, Wp // 2, 32, 8, T * Hp * Wp * 4, ctx.dt))
    a.fold = (H, W)     # original extent: output is ((H-1)//2+1, (W-1)//2+1)
    return a


def export_ncdhw(ctx, a, channels=None):
    """Act -> fp32 contiguous NCDHW tensor (pending affine applied)."""
    v = a.v
    Cc = v.C if channels is None else channels
    out = torch.empty((v.B, v.C, v.T, v.H, v.W), dtype=torch.float32, device=v.device)
    sb, sc, st, sh, sw = out.stride()
    ctx.call("vinet_export_ncdhw", C.byref(v.ct()), v.dt, a.affine(), out.data_ptr(), sb, sc, st, sh, sw, 0, ctx.stream)
    return out[:, :Cc] if Cc != v.C else out


def export_grad_ncdhw(ctx, a, channels):
    g = a.grad_view()
    out = torch.empty((g.B, g.C, g.T, g.H, g.W), dtype=torch.float32, device=g.device)
    sb, sc, st, sh, sw = out.stride()
    ctx.call("vinet_export_ncdhw", C.byref(g.ct()), g.dt, L.CAffine(None, None, 0), out.data_ptr(), sb, sc, st, sh, sw, 0, ctx.stream)
    return out[:, :channels] if channels != g.C else out


def import_grad_ncdhw(ctx, a, gt):
    """seed a.grad from an fp32 NCDHW gradient tensor."""
    g = a.grad_view()
    B, Cc, T, H, W = gt.shape
    sb, sc, st, sh, sw = gt.stride()
    ctx.call("vinet_import_ncdhw", gt.data_ptr(), sb, sc, st, sh, sw, Cc, C.byref(g.ct()), g.dt, ctx.stream)
    a.mark_grad_ready()


def materialize(ctx, a, dst=None, out_dt=None, share_grad=False):
    """apply the pending affine (or just copy / convert) into `dst`; returns a plain Act.

    share_grad: `dst` is a slice of a tensor whose gradient is written (stored) before any other consumer of `a` runs its
    backward (a decoder skip: the decoder's backward precedes the encoder's).  `a` then takes dst's gradient storage as
    its own -- the later consumers accumulate into it and a's producer reads it from there -- and backward needs no copy.
    Only an activation that owns its gradient storage can be re-homed (not a member of a concat).

    With a fresh destination of the activation dtype the result SHARES a's gradient storage: the gradient w.r.t. the
    materialised values is the gradient w.r.t. a's post-affine output, which is what a.grad holds (the producer's BN
    backward takes it from there), so backward needs no copy."""
    v = a.v
    alias = dst is None and (out_dt is None or out_dt == v.dt) and v.dt == ctx.dt
    if dst is None:
        dt = ctx.dt if out_dt is None else out_dt
        dst = Act(View.alloc(v.B, v.T, v.H, v.W, v.C, dt, v.device))
    out = dst.v
    assert out.same_dims(v) and dst.plain
    ctx.call("vinet_copy_affine", C.byref(v.ct()), v.dt, a.affine(), C.byref(out.ct()), out.dt, 0, ctx.stream)
    dst.needs_grad = a.needs_grad
    if alias:
        dst.parent, dst.pc0, dst.pt0 = a, 0, None        # identity "slice": grad_view / readiness resolve into a's
        dst.alias_of = a
        return dst
    _note_reader(ctx, a)
    if (share_grad and SHARE_SKIP_GRAD and ctx.recording and a.needs_grad and a.parent is None and a._grad is None
            and dst.parent is not None and out.dt == v.dt):
        a.parent, a.pc0, a.pt0 = dst, 0, None            # a's gradient lives in dst's slice of the concat gradient
        return dst
    if ctx.recording and a.needs_grad:
        def bwd():
            dg = dst.grad_view()
            tg = a.grad_view()
            ctx.call("vinet_copy_affine", C.byref(dg.ct()), dg.dt, L.CAffine(None, None, 0), C.byref(tg.ct()), tg.dt,
                     1 if a.is_grad_ready() else 0, ctx.stream)
            a.mark_grad_ready()
        ctx.record(bwd)
    return dst


def new_concat(ctx, B, T, H, W, Ctot, pending):
    """destination of a channel concat; `pending` reserves per-channel scale/shift
    vectors that the member convs' BN finalize kernels fill in place."""
    v = View.alloc(B, T, H, W, Ctot, ctx.dt, ctx.device)
    if pending:
        # (training: the member BatchNorms also leave their batch mean / invstd in slices of two vectors, so that a consumer of
        #  the whole concat can hand all of them to one data-gradient launch: _conv_backward, VinetConvDesc::bnb_*)
        st = ctx.training
        return Act(v, ctx.f32(Ctot), ctx.f32(Ctot), relu=True, needs_grad=True, mean=ctx.f32(Ctot) if st else None,
                   invstd=ctx.f32(Ctot) if st else None)
    return Act(v, needs_grad=True)


# ----------------------------------------------------------------------------
# convolution plans
# ----------------------------------------------------------------------------

def _phase_taps_1d(I, O, k, s, p):
    """dgrad of a 1-D strided conv as per-phase stride-1 correlations over dy.

    returns list of (r, Q, [(offset, k_index), ...]): input positions r, r+s, ...
    (Q of them) equal  sum_j dy[q + offset_j] * w[k_index_j].
    """
    out = []
    for r in range(s):
        if r >= I:
            continue
        Q = (I - r + s - 1) // s
        d0 = (r + p) % s
        e = (r + p - d0) // s
        taps = []
        j = 0
        while d0 + s * j < k:
            taps.append((e - j, d0 + s * j))
            j += 1
        out.append((r, Q, taps))
    return out


class ConvPlan:
    """static description of one nn.Conv3d: geometry, parameters, cached packs and tap tables."""

    def __init__(self, weight, bias, kernel, stride, padding, stem=False):
        self.weight, self.bias = weight, bias
        self.k, self.s, self.p = tuple(kernel), tuple(stride), tuple(padding)
        self.N, self.Cin = weight.shape[0], weight.shape[1]
        self.ntaps = self.k[0] * self.k[1] * self.k[2]
        self.stem = stem
        # (k,1,1) kernel: tap kt is (kt - pT, 0, 0, kt)
        self.temporal = self.k[1] == 1 and self.k[2] == 1 and self.p[1] == 0 and self.p[2] == 0 and self.s[1] == 1 and self.s[2] == 1
        # 3 x 3 spatial footprint, unit spatial stride, "same" padding: fwd_taps (kt-major) and every stride phase of the data
        # gradient keep |dh|, |dw| <= 1 with equal-dt taps contiguous -- the promise behind VinetConvDesc::tline == 5
        # (... and weight slices < 64: conv_ht.h keeps a tap's slice in six bits of its scalar tap word)
        self.spatial3 = (not stem) and self.k[1:] == (3, 3) and self.p[1:] == (1, 1) and self.s[1:] == (1, 1) and self.ntaps <= 64
        # 1x1x1, unit stride, no padding: forward and data gradient are one tap (0, 0, 0, slice 0) -- VinetConvDesc::tline == 6
        self.pointwise = (not stem) and self.k == (1, 1, 1) and self.s == (1, 1, 1) and self.p == (0, 0, 0)
        if stem:
            assert self.k == (1, 7, 7) and self.Cin == 3 and self.p[2] == 3
        self._packs = {}
        self._taps = {}
        self._dw_ws = {}

    # ---- geometry ---------------------------------------------------------
    def site(self, xv):
        """human-readable call-site key for profiles: channels, kernel, stride, input extent"""
        return "%d->%d k%dx%dx%d s%dx%dx%d in%dx%dx%dx%d" % ((self.Cin, self.N) + self.k + self.s + (xv.B, xv.T, xv.H, xv.W))

    def out_dims(self, T, H, W):
        return tuple((d + 2 * p - k) // s + 1 for d, k, s, p in zip((T, H, W), self.k, self.s, self.p))

    def kp(self, transpose=False):
        if self.stem and not transpose:
            return 32
        return rup(self.N if transpose else self.Cin, 32)

    # ---- device-side constant tables --------------------------------------
    def _dev_taps(self, key, rows, device):
        k = (key, str(device))
        t = self._taps.get(k)
        if t is None:
            t = torch.tensor(rows, dtype=torch.int32).reshape(-1, 4).to(device)
            self._taps[k] = t
        return t

    def folded_taps(self, device):
        """stem over the folded view: one tap per kernel row, no padding (the buffer is padded)"""
        rows = [(0, kh, 0, kh) for kh in range(self.k[1])]
        return self._dev_taps("fold", rows, device), len(rows)

    def fwd_taps(self, device):
        kT, kH, kW = self.k
        pT, pH, pW = self.p
        if self.stem:
            rows = [(0, kh - pH, -pW, kh) for kh in range(kH)]
        else:
            rows = [(kt - pT, kh - pH, kw - pW, (kt * kH + kh) * kW + kw)
                    for kt in range(kT) for kh in range(kH) for kw in range(kW)]
        return self._dev_taps("fwd", rows, device), len(rows)

    def dgrad_phases(self, in_dims, out_dims, device):
        """list of dicts describing one vinet_conv3d launch per stride phase."""
        kT, kH, kW = self.k
        per = [_phase_taps_1d(I, O, k, s, p) for I, O, k, s, p in zip(in_dims, out_dims, self.k, self.s, self.p)]
        phases = []
        full = True
        for (rT, QT, tT), (rH, QH, tH), (rW, QW, tW) in itertools.product(*per):
            rows = [(oT_, oH_, oW_, (a * kH + b) * kW + c) for (oT_, a) in tT for (oH_, b) in tH for (oW_, c) in tW]
            if not rows:
                full = False
                continue
            key = ("dg", in_dims, rT, rH, rW)
            offs = sorted(r_[0] for r_ in rows)
            tline = self.temporal and offs == list(range(offs[0], offs[0] + len(offs))) and self.ntaps <= 64
            tl = 1 if tline else 0
            if self.spatial3:
                dts = [r_[0] for r_ in rows]
                assert all(abs(r_[1]) <= 1 and abs(r_[2]) <= 1 for r_ in rows) and dts == sorted(dts, key=dts.index)
                assert all(dts[i] == dts[i - 1] or dts[i] not in dts[:i] for i in range(1, len(dts))), "equal-dt taps must be contiguous"
                tl = 5
            if self.pointwise:
                assert rows == [(0, 0, 0, 0)]
                tl = 6
            phases.append(dict(taps=self._dev_taps(key, rows, device), ntaps=len(rows), Q=(QT, QH, QW), r=(rT, rH, rW),
                               tline=tl, tpad=-offs[0] if tline else 0))
        covered = all(len(p_) == min(s, I) for p_, s, I in zip(per, self.s, in_dims))
        return phases, (full and covered)

    # ---- packed weights -----------------------------------------------------
    def packed(self, ctx, transpose=False):
        # (keyed and packed by the ARITHMETIC dtype: the split-bf16 form reads hi / lo bf16 planes -- the same bytes per row as
        #  the fp32 pack, a different content: include/vinet_hip.h, vinet_pack_weights)
        key = (ctx.cdt, transpose, str(ctx.device))
        stamp = self._pack_stamp()
        ent = self._packs.get(key)
        if ent is not None and ent[0] == stamp:
            return ent[1]
        if ent is not None and _PACKS.repack_all(ctx, key):     # stale: every registered weight in one launch
            return self._packs[key][1]
        w = self.weight.detach()
        assert w.dtype == torch.float32 and w.is_contiguous()
        buf = ent[1] if ent is not None else torch.empty(self.pack_numel(transpose), dtype=TORCH_DT[ctx.dt], device=ctx.device)
        ctx.call("vinet_pack_weights", w.data_ptr(), self.N, self.Cin, self.ntaps, 1 if transpose else 0,
                 1 if (self.stem and not transpose) else 0, ctx.cdt, buf.data_ptr(), ctx.stream)
        self._packs[key] = (stamp, buf)
        _PACKS.register(self, key)
        return buf

    def _pack_stamp(self):
        return (self.weight._version, _WEIGHTS_EPOCH, self.weight.data_ptr())

    def pack_jobs(self, key):
        """rows of the multi-pack job table (without the prefix field): (w, out, N, Cin, ntaps, flags, elements, ld|col<<32)"""
        tr = 1 if key[1] else 0
        stem = 1 if (self.stem and not key[1]) else 0
        return [(self.weight.data_ptr(), self._packs[key][1].data_ptr(), self.N, self.Cin, self.ntaps, tr | (stem << 1),
                 self.pack_numel(key[1]), 0)]

    def wants_wgrad(self):
        return self.weight.requires_grad

    def unpack_wgrad(self, ctx, dw, clear=False):
        """packed fp32 dw -> += .grad in torch layout (`clear`: the kernel hands dw back zeroed)"""
        gw = _param_grad(self.weight)
        ctx.call("vinet_unpack_wgrad", dw.data_ptr(), self.N, self.Cin, self.ntaps, 1 if self.stem else 0, 3 if clear else 1,
                 gw.data_ptr(), ctx.stream)

    def grad_targets(self):
        return [self.weight]

    def unpack_jobs(self, dw):
        """rows of the multi-unpack job table (without the prefix field): (dw, grad, N, Cin, ntaps, stem, packed elements)"""
        gw = _param_grad(self.weight)
        nsl, kp = (7, 32) if self.stem else (self.ntaps, self.kp(False))
        return [(dw.data_ptr(), gw.data_ptr(), self.N, self.Cin, self.ntaps, 1 if self.stem else 0, nsl * self.N * kp)]

    def dw_workspace(self, ctx, numel):
        """persistent packed fp32 weight-gradient workspace of this conv: zero-filled ONCE; vinet_unpack_wgrad hands it
        back zeroed after every use (flag bit 1), so a training step issues no fill launch per conv."""
        key = (str(ctx.device), numel)
        ws = self._dw_ws.get(key)
        if ws is None:
            ws = self._dw_ws[key] = torch.zeros(numel, dtype=torch.float32, device=ctx.device)
        return ws

    def pack_numel(self, transpose):
        if self.stem and not transpose:
            return 7 * self.N * 32
        if not transpose:
            return self.ntaps * self.N * self.kp(False)
        return self.ntaps * self.Cin * self.kp(True)


def pack_table(jobs, device):
    """device-side job table of vinet_pack_weights_multi (include/vinet_hip.h) -> (table, total elements)"""
    rows, off = [], 0
    for w, out, N, Cin, ntaps, flags, numel, ldcol in jobs:
        rows.append([w, out, N, Cin, ntaps, flags, off, ldcol])
        off += numel
    rows.append([0, 0, 0, 0, 0, 0, off, 0])
    return torch.tensor(rows, dtype=torch.int64).to(device), off


class JointConvPlan(ConvPlan):
    """Several 1x1x1 convs over the SAME input (the entry convs of an Inception block, model_utils.py:176-187)
    run as one conv: forward packs stacked along N, transposed packs side by side along K, one weight-gradient
    launch whose rows are handed back to each member's .grad.  The members keep their own parameters."""

    def __init__(self, members):
        m0 = members[0]
        assert all(m.k == (1, 1, 1) and m.s == (1, 1, 1) and m.p == (0, 0, 0) and m.Cin == m0.Cin and m.bias is None and not m.stem
                   for m in members)
        self.members = list(members)
        self.offs = [sum(m.N for m in members[:i]) for i in range(len(members))]
        self.weight, self.bias = None, None
        self.k, self.s, self.p = (1, 1, 1), (1, 1, 1), (0, 0, 0)
        self.N, self.Cin, self.ntaps, self.stem = sum(m.N for m in members), m0.Cin, 1, False
        self.temporal = self.spatial3 = False
        self.pointwise = True
        self._packs, self._taps, self._dw_ws = {}, {}, {}

    def grad_targets(self):
        return [m.weight for m in self.members]

    def _pack_stamp(self):
        return tuple(m._pack_stamp() for m in self.members)

    def wants_wgrad(self):
        assert len({m.weight.requires_grad for m in self.members}) == 1, "joint conv: freeze all members or none"
        return self.members[0].weight.requires_grad

    def pack_jobs(self, key):
        dt, transpose = key[0], key[1]
        base, es = self._packs[key][1].data_ptr(), (4 if dt == F32S else ESIZE[dt])
        jobs = []
        for m, off in zip(self.members, self.offs):
            w = m.weight.detach()
            assert w.dtype == torch.float32 and w.is_contiguous()
            if transpose:
                jobs.append((w.data_ptr(), base, m.N, m.Cin, 1, 1, m.Cin * rup(m.N, 32), self.kp(True) | (off << 32)))
            else:
                jobs.append((w.data_ptr(), base + off * self.kp(False) * es, m.N, m.Cin, 1, 0, m.N * self.kp(False), 0))
        return jobs

    def packed(self, ctx, transpose=False):
        key = (ctx.cdt, transpose, str(ctx.device))
        stamp = self._pack_stamp()
        ent = self._packs.get(key)
        if ent is not None and ent[0] == stamp:
            return ent[1]
        if ent is not None and _PACKS.repack_all(ctx, key):
            return self._packs[key][1]
        if ent is None:     # zero-filled once: the side-by-side jobs never touch the K padding columns
            self._packs[key] = (None, torch.zeros(self.pack_numel(transpose), dtype=TORCH_DT[ctx.dt], device=ctx.device))
            _PACKS.register(self, key)
        jobs = self.pack_jobs(key)
        table, total = pack_table(jobs, ctx.device)
        ctx.call("vinet_pack_weights_multi", table.data_ptr(), len(jobs), total, ctx.cdt, ctx.stream)
        self._packs[key] = (stamp, self._packs[key][1])
        self._keep_table = table       # the launch reads it asynchronously
        return self._packs[key][1]

    def unpack_jobs(self, dw):
        kp = self.kp(False)
        return [(dw.data_ptr() + off * kp * 4, _param_grad(m.weight).data_ptr(), m.N, m.Cin, 1, 0, m.N * kp)
                for m, off in zip(self.members, self.offs)]

    def unpack_wgrad(self, ctx, dw, clear=False):
        kp = self.kp(False)
        for m, off in zip(self.members, self.offs):
            gw = _param_grad(m.weight)
            ctx.call("vinet_unpack_wgrad", dw.data_ptr() + off * kp * 4, m.N, m.Cin, 1, 0, 3 if clear else 1, gw.data_ptr(), ctx.stream)


class _PackRegistry:
    """Every (plan, dtype, transpose, device) pack that has been built once.  After an optimizer step all of them
    are stale together, so the first stale request re-packs the whole group with ONE vinet_pack_weights_multi
    launch (a device-side job table, rebuilt only when the set of jobs or a weight pointer changes)."""

    def __init__(self):
        self.groups = {}      # (dt, device str) -> dict(jobs=[(weakref(plan), key)], table=None, sig=None, total=0, njobs=0)

    def register(self, plan, key):
        g = self.groups.setdefault((key[0], key[2]), dict(jobs=[], table=None, sig=None, total=0))
        g["jobs"].append((weakref.ref(plan), key))
        g["table"] = None

    def repack_all(self, ctx, key):
        g = self.groups.get((key[0], key[2]))
        if g is None:
            return False
        live = [(r(), k) for r, k in g["jobs"] if r() is not None and k in r()._packs]
        if len(live) != len(g["jobs"]):
            g["jobs"] = [(weakref.ref(p), k) for p, k in live]
            g["table"] = None
        if len(live) < 2:
            return False
        jobs = [j for p, k in live for j in p.pack_jobs(k)]
        sig = tuple((j[0], j[1]) for j in jobs)
        if g["table"] is None or g["sig"] != sig:
            g["table"], g["total"] = pack_table(jobs, ctx.device)
            g["sig"], g["njobs"] = sig, len(jobs)
        ctx.call("vinet_pack_weights_multi", g["table"].data_ptr(), g["njobs"], g["total"], key[0], ctx.stream)
        for p, k in live:
            p._packs[k] = (p._pack_stamp(), p._packs[k][1])
        return True


_PACKS = _PackRegistry()


class BNState:
    """parameters/buffers of one BatchNorm + per-forward scratch."""

    def __init__(self, gamma, beta, running_mean, running_var, eps, momentum, fold_cache=None):
        self.gamma, self.beta, self.rm, self.rv, self.eps, self.momentum = gamma, beta, running_mean, running_var, eps, momentum
        self.steps = 0   # forwards in training mode since the counter buffer was last synced
        # eval-mode (scale, shift) of the folded BN, owned by the module: inference re-folds only when a
        # parameter / buffer / conv bias changed (tensor versions, pointers, optimizer epoch)
        self.fold_cache = fold_cache

    def fold_stamp(self, bias, dt, device):
        ts = (self.gamma, self.beta, self.rm, self.rv, bias)
        return (tuple((t._version, t.data_ptr()) if t is not None else None for t in ts), _WEIGHTS_EPOCH, dt, str(device))

    # (pointer offsets: the joint form below runs the same kernels on channel slices)
    def fold(self, ctx, bias, N, scale, shift, invstd=None, off=0):
        o = 4 * off
        ctx.call("vinet_bn_fold", _ptr(self.gamma), _ptr(self.beta), self.rm.data_ptr(), self.rv.data_ptr(), _ptr(bias),
                 float(self.eps), N, scale.data_ptr() + o, shift.data_ptr() + o,
                 None if invstd is None else invstd.data_ptr() + o, ctx.stream)

    def finalize(self, ctx, stats, rows, N, M, mean, invstd, scale, shift, off=0, ld=0):
        o = 4 * off
        ctx.call("vinet_bn_finalize", stats.data_ptr() + o, rows, N, ld, float(M), _ptr(self.gamma), _ptr(self.beta),
                 float(self.eps), float(self.momentum), self.rm.data_ptr(), self.rv.data_ptr(), mean.data_ptr() + o,
                 invstd.data_ptr() + o, scale.data_ptr() + o, shift.data_ptr() + o, ctx.stream)
        self.steps += 1

    def bwd_finalize(self, ctx, ws, rows, N, M, scale, train_bn, invstd, c1, c2, off=0, ld=0, ws_off=None):
        """`off`: this BatchNorm's first channel in the per-channel vectors; `ws_off` / `ld`: its first column and the row width
        of the partial-sum table (default: the same layout as the vectors)"""
        o = 4 * off
        dg = _param_grad(self.gamma) if self.gamma is not None and self.gamma.requires_grad else None
        db = _param_grad(self.beta) if self.beta is not None and self.beta.requires_grad else None
        ctx.call("vinet_bn_bwd_finalize", ws.data_ptr() + (o if ws_off is None else 4 * ws_off), rows, N, ld, float(M), scale.data_ptr() + o, 1 if train_bn else 0,
                 _ptr(dg), _ptr(db), invstd.data_ptr() + o, c1.data_ptr() + o, c2.data_ptr() + o, ctx.stream)
        _note_param_grad(ctx, self.gamma, self.beta)


class JointBN:
    """the BatchNorms of a JointConvPlan's members, over consecutive channel ranges of the joint output"""

    def __init__(self, members, widths, fold_cache=None):
        self.members, self.widths = list(members), list(widths)
        self.offs = [sum(widths[:i]) for i in range(len(widths))]
        self.N = sum(widths)
        self.fold_cache = fold_cache
        self.eps, self.momentum = members[0].eps, members[0].momentum

    def fold_stamp(self, bias, dt, device):
        return tuple(m.fold_stamp(None, dt, device) for m in self.members)

    def fold(self, ctx, bias, N, scale, shift, invstd=None):
        assert bias is None and N == self.N
        for m, w, o in zip(self.members, self.widths, self.offs):
            m.fold(ctx, None, w, scale, shift, invstd, off=o)

    def finalize(self, ctx, stats, rows, N, M, mean, invstd, scale, shift):
        for m, w, o in zip(self.members, self.widths, self.offs):
            m.finalize(ctx, stats, rows, w, M, mean, invstd, scale, shift, off=o, ld=N)

    def bwd_finalize(self, ctx, ws, rows, N, M, scale, train_bn, invstd, c1, c2):
        for m, w, o in zip(self.members, self.widths, self.offs):
            m.bwd_finalize(ctx, ws, rows, w, M, scale, train_bn, invstd, c1, c2, off=o, ld=N)


# Called as hook(ctx, param) when the tape has LAUNCHED the last kernel that writes `param`'s gradient in this backward
# (weight gradients: on the side stream; BatchNorm / bias gradients: on the main stream).  vinet_amd.parallel's
# GradientBuckets hangs its bucketed, overlapped all-reduce on it.  The current stream is the main stream when it fires.
PARAM_GRAD_HOOK = None


def _note_param_grad(ctx, *params):
    if PARAM_GRAD_HOOK is not None:
        for p in params:
            if p is not None and p.requires_grad:
                PARAM_GRAD_HOOK(ctx, p)


def _param_grad(p):
    """fp32 gradient buffer of a parameter that kernels accumulate into."""
    if p.grad is None:
        p.grad = torch.zeros_like(p, memory_format=torch.contiguous_format)
    return p.grad


def _conv_kernel_name(ctx, d):
    buf = C.create_string_buffer(96)
    ctx.lib.vinet_conv3d_kernel_name(C.byref(d), buf, 96)
    return buf.value.decode() or "conv"


def _wgrad_kernel_name(ctx, wd):
    buf = C.create_string_buffer(96)
    ctx.lib.vinet_conv3d_wgrad_kernel_name(C.byref(wd), buf, 96)
    return buf.value.decode() or "wgrad"


def _splitk_scratch(ctx, d):
    """Grids too small for the chip (batch-1 inference) split their K loop when the caller lends scratch
    memory (include/vinet_hip.h: splitk_ws).  The tensor comes from torch's stream-ordered caching
    allocator, so dropping the reference right after the launch is safe (and graph capture keeps it)."""
    nbytes = ctx.lib.vinet_conv3d_splitk_bytes(C.byref(d))
    if nbytes <= 0:
        return None
    ws = torch.empty(nbytes // 4, dtype=torch.float32, device=ctx.device)
    d.splitk_ws, d.splitk_ws_bytes = ws.data_ptr(), nbytes
    return ws




def conv_forward(ctx, plan, x, bn=None, act=L.ACT_NONE, dst=None, out_dt=None, n_pad=None):
    """x -> conv (-> BN) (-> act).  Returns the output Act.

    training + bn : raw conv output is stored; statistics come from the conv
                    epilogue; the result carries a pending (scale, shift, relu).
    eval + bn     : BN folded into the conv epilogue together with ReLU.
    no bn         : bias / activation in the epilogue.
    `dst`         : optional destination Act (channel / T slice of a concat buffer).
    `n_pad`       : store into a channel-padded output (N not a multiple of the vector width).
    """
    lib_dt = ctx.dt
    _note_reader(ctx, x)
    want_plain = (MATERIALIZE_NT and x.scale is not None and ctx.dt != L.F32 and x.fold is None and not plan.stem and
                  plan.N * plan.ntaps >= MATERIALIZE_NT and x.v.dt == ctx.dt)
    xv = x.v
    folded = plan.stem and x.fold is not None
    if folded:
        oT, oH, oW = xv.T, (x.fold[0] - 1) // 2 + 1, (x.fold[1] - 1) // 2 + 1
    else:
        oT, oH, oW = plan.out_dims(xv.T, xv.H, xv.W)
    Ny = plan.N if n_pad is None else n_pad
    odt = lib_dt if out_dt is None else out_dt
    if dst is None:
        dst = Act(View.alloc(xv.B, oT, oH, oW, Ny, odt, xv.device))
    out = dst.v
    scale_out, shift_out = dst.scale, dst.shift
    mean_out, invstd_out = dst.mean, dst.invstd
    dst.mean = dst.invstd = None        # (set below by the training-mode BatchNorm path only)
    assert (out.B, out.T, out.H, out.W, out.C) == (xv.B, oT, oH, oW, Ny), "conv output view mismatch"
    taps, ntaps = plan.folded_taps(ctx.device) if folded else plan.fwd_taps(ctx.device)
    w = plan.packed(ctx, False)

    d = L.CConvDesc()
    d.dtype, d.out_dtype, d.mode = ctx.cdt, out.dt, (L.CONV_STEM if (plan.stem and not folded) else L.CONV_GENERIC)
    d.x, d.y = xv.ct(), out.ct()
    d.oT, d.oH, d.oW = oT, oH, oW
    d.sT, d.sH, d.sW = (1, 2, 1) if folded else plan.s
    d.omT = d.omH = d.omW = 1
    d.ooT = d.ooH = d.ooW = 0
    d.ntaps, d.taps, d.w, d.Kp = ntaps, taps.data_ptr(), w.data_ptr(), plan.kp(False)
    d.pre = x.affine()
    d.accumulate = 0
    d.n_valid = plan.N if Ny != plan.N else 0
    if plan.pointwise:
        d.tline = 6                         # the single tap (0, 0, 0, slice 0)
    elif not folded and not plan.stem and plan.temporal:
        d.tline, d.tpad = 1, plan.p[0]      # promise to the library (it cannot read the device-side tap table)
    elif folded:
        d.tline = 2                         # taps (0, kh, 0, kh): ConvPlan.folded_taps
    elif plan.spatial3:
        d.tline = 5                         # 3 x 3 spatial footprint, kt-major tap order: ConvPlan.fwd_taps
    if want_plain and not ctx.lib.vinet_conv3d_applies_pre_once(C.byref(d)):
        # (the halo-tile kernel applies a pending BN + ReLU once per staged element: no materialisation pass for its layers)
        x = materialize(ctx, x)
        xv = x.v
        d.x, d.pre = xv.ct(), x.affine()
    M = xv.B * oT * oH * oW
    site = plan.site(xv)
    es = ESIZE[lib_dt]
    work = dict(flops=2.0 * M * plan.N * plan.Cin * plan.ntaps,
                bytes=float(xv.nvox * plan.Cin * es + M * plan.N * ESIZE[out.dt] + plan.N * plan.Cin * plan.ntaps * es))
    conv_tag = _conv_kernel_name(ctx, d) + " | fwd " + site if PROFILER is not None else None

    train_bn = bn is not None and ctx.training
    keep = {}
    res = dst
    res.needs_grad = True
    if bn is None:
        d.out_scale, d.out_shift = None, _ptr(plan.bias)
        d.act, d.stats = act, None
        ws = _splitk_scratch(ctx, d)
        ctx.call("vinet_conv3d", C.byref(d), ctx.stream, tag=conv_tag, work=work)
        res.scale = res.shift = None
        res.relu = False
        res.act_out = act
    elif not train_bn:
        scale = ctx.f32(plan.N) if scale_out is None else scale_out
        shift = ctx.f32(plan.N) if shift_out is None else shift_out
        if ctx.recording:
            # eval-mode BN with gradients: keep the raw conv output (+bias), BN(+ReLU) stays pending
            assert not isinstance(bn, JointBN), "joint convs are not recorded under eval-mode BN"
            invstd = ctx.f32(plan.N)
            bn.fold(ctx, None, plan.N, scale, shift, invstd)
            d.out_scale, d.out_shift, d.act, d.stats = None, _ptr(plan.bias), L.ACT_NONE, None
            ctx.call("vinet_conv3d", C.byref(d), ctx.stream, tag=conv_tag, work=work)
            res.scale, res.shift, res.relu = scale, shift, (act == L.ACT_RELU)
            keep.update(mean=bn.rm, invstd=invstd)
        else:
            cached = None
            if scale_out is None and shift_out is None and bn.fold_cache is not None:
                stamp = bn.fold_stamp(plan.bias, ctx.dt, ctx.device)
                cached = bn.fold_cache.get("fold")
                if cached is not None and cached[0] == stamp:
                    scale, shift = cached[1], cached[2]
                else:
                    cached = None
                    bn.fold_cache["fold"] = (stamp, scale, shift)
            if cached is None:
                bn.fold(ctx, plan.bias, plan.N, scale, shift)
            d.out_scale, d.out_shift, d.act, d.stats = scale.data_ptr(), shift.data_ptr(), act, None
            ws = _splitk_scratch(ctx, d)
            ctx.call("vinet_conv3d", C.byref(d), ctx.stream, tag=conv_tag, work=work)
            res.scale = res.shift = None
            res.relu = False
    else:
        rows = ctx.lib.vinet_conv3d_stats_rows(C.byref(d))
        stats = ctx.f32(rows * 2 * plan.N)
        d.out_scale, d.out_shift, d.act, d.stats = None, _ptr(plan.bias), L.ACT_NONE, stats.data_ptr()
        ctx.call("vinet_conv3d", C.byref(d), ctx.stream, tag=conv_tag, work=work)
        scale = ctx.f32(plan.N) if scale_out is None else scale_out
        shift = ctx.f32(plan.N) if shift_out is None else shift_out
        mean = ctx.f32(plan.N) if mean_out is None else mean_out
        invstd = ctx.f32(plan.N) if invstd_out is None else invstd_out
        if rows >= 1024:        # tall table (early, high-resolution layers): coalesced pre-reduction to 256 rows
            per = (rows + 255) // 256
            rows2 = (rows + per - 1) // per
            folded = ctx.f32(rows2 * 2 * plan.N)
            ctx.call("vinet_bn_partials_fold", stats.data_ptr(), rows, plan.N, folded.data_ptr(), rows2, ctx.stream)
            stats, rows = folded, rows2
        bn.finalize(ctx, stats, rows, plan.N, M, mean, invstd, scale, shift)
        res.scale, res.shift, res.relu = scale, shift, (act == L.ACT_RELU)
        keep.update(mean=mean, invstd=invstd)
        res.mean, res.invstd = mean, invstd      # (a consumer's data gradient may fold this BatchNorm's backward reduce pass in: _conv_backward)

    if ctx.recording:
        ctx.record(lambda: _conv_backward(ctx, plan, x, res, bn, act, train_bn, keep, M))
    return res


class _OnStream:
    def __init__(self, ctx, st):
        self.ctx, self.st = ctx, st

    def __enter__(self):
        self.prev = self.ctx.stream
        self.cm = torch.cuda.stream(self.st)
        self.cm.__enter__()
        self.ctx.stream = self.st.cuda_stream
        if DBG_SPIN_FORK and self.ctx.device.type == "cuda" and not torch.cuda.is_current_stream_capturing():
            # schedule stress: the branch stream (> 0) or the forking stream (< 0) idles first, so the streams drift apart
            self.ctx.lib.vinet_debug_spin(abs(DBG_SPIN_FORK), self.st.cuda_stream if DBG_SPIN_FORK > 0 else self.prev)
        return self

    def __exit__(self, *exc):
        self.ctx.stream = self.prev
        return self.cm.__exit__(*exc)


class _NullCtx:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False




def _split_planes_folded(ctx, x):
    """hi / lo planes of the folded RGB stem input (import_video_folded): the zero-padded 4-channel buffer [B][T][H+6][Wp][4]
    is split as a plain tensor; the planes are the same overlapped view (32 'channels' = 8 pixels, ld = 8) over each half."""
    v = x.v
    B, T, Hp, Wp = v.B, v.T, v.H, 2 * v.W
    n = B * T * Hp * Wp * 4
    src = View(v.buf, v.off, B, T, Hp, Wp, 4, 4, T * Hp * Wp * 4, F32)
    acts = []
    bufs = [torch.empty(n + 64, dtype=TORCH_DT[BF16], device=v.device) for _ in range(2)]
    for b in bufs:
        b[n:].zero_()
    hi, lo = (View(b, 0, B, T, Hp, Wp, 4, 4, T * Hp * Wp * 4, BF16) for b in bufs)
    ctx.call("vinet_split_bf16", C.byref(src.ct()), L.CAffine(None, None, 0), C.byref(hi.ct()), C.byref(lo.ct()), ctx.stream)
    for b in bufs:
        a = Act(View(b, 0, B, T, Hp, Wp // 2, 32, 8, T * Hp * Wp * 4, BF16))
        a.fold = x.fold
        acts.append(a)
    return acts


def _split_planes(ctx, x, dy, dy_planes=None):
    """hi / lo bf16 planes of a conv's input (pending affine applied) and of its output gradient: ((x_hi, x_lo), (dy_hi, dy_lo));
    dy_planes: those of dy exist already (vinet_bn_bwd_apply_split)"""
    out = []
    if x.fold is not None:
        if dy_planes is None:
            hi = View.alloc(dy.B, dy.T, dy.H, dy.W, dy.C, BF16, dy.device)
            lo = View.alloc(dy.B, dy.T, dy.H, dy.W, dy.C, BF16, dy.device)
            ctx.call("vinet_split_bf16", C.byref(dy.ct()), L.CAffine(None, None, 0), C.byref(hi.ct()), C.byref(lo.ct()), ctx.stream)
            dy_planes = (hi, lo)
        return [tuple(_split_planes_folded(ctx, x)), dy_planes]
    for v, aff in ((x.v, x.affine()),) + (((dy, L.CAffine(None, None, 0)),) if dy_planes is None else ()):
        hi = View.alloc(v.B, v.T, v.H, v.W, v.C, BF16, v.device)
        lo = View.alloc(v.B, v.T, v.H, v.W, v.C, BF16, v.device)
        ctx.call("vinet_split_bf16", C.byref(v.ct()), aff, C.byref(hi.ct()), C.byref(lo.ct()), ctx.stream)
        out.append((hi, lo))
    if dy_planes is not None:
        out.append(dy_planes)
    return out


def _wgrad_desc(ctx, plan, x, dy, dw=None):
    folded = plan.stem and x.fold is not None
    taps, ntaps = plan.folded_taps(ctx.device) if folded else plan.fwd_taps(ctx.device)
    wd = L.CWgradDesc()
    wd.dtype, wd.mode = ctx.cdt, (L.CONV_STEM if (plan.stem and not folded) else L.CONV_GENERIC)
    wd.x, wd.dy = x.v.ct(), dy.ct()
    wd.sT, wd.sH, wd.sW = (1, 2, 1) if folded else plan.s
    wd.ntaps, wd.taps, wd.dw, wd.Kp = ntaps, taps.data_ptr(), (dw.data_ptr() if dw is not None else None), plan.kp(False)
    wd.pre = x.affine()
    # a purely temporal kernel (k,1,1): tap kt is (kt - pad, 0, 0, kt) -- the library cannot read the
    # device-side tap table, so the geometry is promised here (include/vinet_hip.h: tline)
    if not folded and not plan.stem and plan.temporal:
        wd.tline, wd.tpad = 1, plan.p[0]
    elif folded:
        wd.tline = 2        # taps (0, kh, 0, kh): ConvPlan.folded_taps
    elif not plan.stem and plan.k[1:] == (3, 3) and plan.p == (0, 1, 1) and plan.s == (plan.k[0], 1, 1):
        wd.tline = 4        # kT x 3 x 3, temporal stride = kT (the decoder): ConvPlan.fwd_taps order
    wd._keep = taps
    return wd


def _conv_backward(ctx, plan, x, res, bn, act, train_bn, keep, M):
    out = res.v
    dz = res.grad_view()
    assert res.is_grad_ready(), "conv backward reached before any consumer produced a gradient"
    Ny = out.C
    fused_bnb = None
    dy_planes = None        # fp32s: (hi, lo) planes of dy written by the BatchNorm-backward apply pass
    # ---- through BN / activation: dz -> dy (w.r.t. the raw conv output) --------
    if bn is not None:
        fwd = res.affine()
        nb = float(dz.nvox * dz.C * ESIZE[dz.dt])
        # per BatchNorm behind this conv (one, or the members of a joint entry conv): its two backward sums either came with the
        # gradient -- the data gradient that wrote dz LAST left their partial rows (VinetConvDesc::bnb_*; _find_bnb checks that
        # no writer came after it) -- or need a reduce pass over (dz, z)
        members = list(zip(bn.members, bn.offs, bn.widths)) if isinstance(bn, JointBN) else [(bn, 0, Ny)]
        c1, c2 = ctx.f32(Ny), ctx.f32(Ny)
        left = []
        for m_, o_, w_ in members:
            part = _find_bnb(ctx, res, o_, w_) if DGRAD_BN_STATS else None
            if part is None:
                left.append((m_, o_, w_))
            else:
                m_.bwd_finalize(ctx, part[0], part[1], w_, M, res.scale, train_bn, keep["invstd"], c1, c2, off=o_, ld=part[3], ws_off=part[2])
        if len(left) == len(members):       # nothing came with the gradient: ONE pass over the whole tensor
            rows = ctx.lib.vinet_stats_rows(C.byref(dz.ct()))
            ws = ctx.f32(rows * 2 * Ny)
            ctx.call("vinet_bn_bwd_reduce", C.byref(dz.ct()), C.byref(out.ct()), dz.dt, fwd, keep["mean"].data_ptr(),
                     keep["invstd"].data_ptr(), ws.data_ptr(), ctx.stream,
                     tag=("vinet_bn_bwd_reduce | C%d x %d voxels" % (dz.C, dz.nvox)) if PROFILER is not None else None,
                     work=dict(flops=0.0, bytes=2 * nb))
            for m_, o_, w_ in left:
                m_.bwd_finalize(ctx, ws, rows, w_, M, res.scale, train_bn, keep["invstd"], c1, c2, off=o_, ld=Ny)
        else:
            for m_, o_, w_ in left:         # (a joint conv: the members whose gradient has several writers)
                dzs, outs = dz.chan(o_, o_ + w_), out.chan(o_, o_ + w_)
                rows = ctx.lib.vinet_stats_rows(C.byref(dzs.ct()))
                ws = ctx.f32(rows * 2 * w_)
                fsl = L.CAffine(res.scale.data_ptr() + 4 * o_, res.shift.data_ptr() + 4 * o_, 1 if res.relu else 0)
                ctx.call("vinet_bn_bwd_reduce", C.byref(dzs.ct()), C.byref(outs.ct()), dz.dt, fsl, keep["mean"].data_ptr() + 4 * o_,
                         keep["invstd"].data_ptr() + 4 * o_, ws.data_ptr(), ctx.stream,
                         tag=("vinet_bn_bwd_reduce | C%d x %d voxels" % (w_, dz.nvox)) if PROFILER is not None else None,
                         work=dict(flops=0.0, bytes=2 * nb * w_ / Ny))
                m_.bwd_finalize(ctx, ws, rows, w_, M, res.scale, train_bn, keep["invstd"], c1, c2, off=o_, ld=w_, ws_off=0)
        # A conv whose input needs no gradient (the RGB stem) has one consumer of dz, its weight gradient: kernels
        # that can form dz from (gradient behind the BN, raw conv output) on the fly spare the apply pass
        if BN_BWD_FUSE and not x.needs_grad and plan.wants_wgrad() and plan.bias is None and out.dt == dz.dt == ctx.dt:
            q = _wgrad_desc(ctx, plan, x, dz)
            q.bnb_z, q.bnb_ld, q.bnb_sB, q.bnb_fwd = out.ptr(), out.ld, out.sB, fwd
            q.bnb_mean, q.bnb_invstd = keep["mean"].data_ptr(), keep["invstd"].data_ptr()
            q.bnb_c1, q.bnb_c2 = c1.data_ptr(), c2.data_ptr()
            if ctx.lib.vinet_conv3d_wgrad_fuses_bn_bwd(C.byref(q)):
                fused_bnb = (out, fwd, keep["mean"], keep["invstd"], c1, c2)
        if (fused_bnb is None and SPLIT_IN_APPLY and SPLIT_WGRAD_BF16 and ctx.cdt == F32S and plan.wants_wgrad() and dz.dt == F32 and
                out.dt == F32 and dz.C % 8 == 0 and dz.ld % 8 == 0 and out.ld % 8 == 0 and ctx.device.type == "cuda"):
            # fp32s: the hi / lo planes of dy (an operand of the three bf16 weight-gradient launches) leave with the apply pass
            dh = View.alloc(dz.B, dz.T, dz.H, dz.W, dz.C, BF16, dz.device)
            dl = View.alloc(dz.B, dz.T, dz.H, dz.W, dz.C, BF16, dz.device)
            ctx.call("vinet_bn_bwd_apply_split", C.byref(dz.ct()), C.byref(out.ct()), fwd, keep["mean"].data_ptr(),
                     keep["invstd"].data_ptr(), c1.data_ptr(), c2.data_ptr(), C.byref(dz.ct()), C.byref(dh.ct()), C.byref(dl.ct()), ctx.stream,
                     tag=("vinet_bn_bwd_apply | C%d x %d voxels" % (dz.C, dz.nvox)) if PROFILER is not None else None,
                     work=dict(flops=0.0, bytes=4 * nb))
            dy_planes = (dh, dl)
        elif fused_bnb is None:
            ctx.call("vinet_bn_bwd_apply", C.byref(dz.ct()), C.byref(out.ct()), dz.dt, fwd, keep["mean"].data_ptr(),
                     keep["invstd"].data_ptr(), c1.data_ptr(), c2.data_ptr(), C.byref(dz.ct()), ctx.stream,
                     tag=("vinet_bn_bwd_apply | C%d x %d voxels" % (dz.C, dz.nvox)) if PROFILER is not None else None,
                     work=dict(flops=0.0, bytes=3 * nb))
        dy = dz
    elif act == L.ACT_RELU and res.grad_masked == res.root().grad_marks and dz.dt == ctx.dt:
        dy = dz         # the only writer of dz (the upsample's backward) already gated it with this ReLU
    elif act != L.ACT_NONE:
        if dz.dt == ctx.dt:
            dy = dz
        else:
            dy = View.alloc(dz.B, dz.T, dz.H, dz.W, dz.C, ctx.dt, dz.device)
        ctx.call("vinet_act_bwd", C.byref(dz.ct()), dz.dt, C.byref(out.ct()), out.dt, act, C.byref(dy.ct()), dy.dt, ctx.stream)
    else:
        dy = dz
        if dz.dt != ctx.dt:
            dy = View.alloc(dz.B, dz.T, dz.H, dz.W, dz.C, ctx.dt, dz.device)
            ctx.call("vinet_copy_affine", C.byref(dz.ct()), dz.dt, L.CAffine(None, None, 0), C.byref(dy.ct()), dy.dt, 0, ctx.stream)
    # ---- bias ------------------------------------------------------------------
    if plan.bias is not None and plan.bias.requires_grad:
        rows = ctx.lib.vinet_stats_rows(C.byref(dy.ct()))
        ws = ctx.f32(rows * 2 * Ny)
        gb = _param_grad(plan.bias)
        # channel-padded head: pad-channel gradients are exactly zero, so folding
        # channels modulo N leaves the real sums untouched
        assert Ny % plan.N == 0 and (Ny == plan.N or plan.N == 1)
        ctx.call("vinet_channel_sum", C.byref(dy.ct()), dy.dt, ws.data_ptr(), plan.N, gb.data_ptr(), 1, ctx.stream)
        _note_param_grad(ctx, plan.bias)
    # ---- weight gradient (side stream) ---------------------------------------------
    if plan.wants_wgrad():
        for w_ in plan.grad_targets():
            _param_grad(w_)          # (allocated on the main stream, not inside the side-stream context)

        # fp32s: hi / lo planes of both operands for the three bf16 launches -- any conv with whole 8-channel groups, and the
        # folded RGB stem (the row-streaming strip kernel, wgrad_hs.hip, over planes of the padded clip)
        want_planes = bool(SPLIT_WGRAD_BF16 and ctx.cdt == F32S and fused_bnb is None and dy.C % 8 == 0 and x.v.dt == F32 and dy.dt == F32 and
                           ((not plan.stem and x.fold is None and x.v.C % 8 == 0) or
                            (plan.stem and x.fold is not None and x.scale is None and x.v.off == 0)))
        planes_main = None

        def wgrad_job():
            ctx._side_rr = (getattr(ctx, "_side_rr", -1) + 1) % N_SIDE_STREAMS
            side = ctx.side_stream(ctx._side_rr)
            main_ptr = ctx.stream
            if side is not None:
                if not getattr(ctx, "_joined", False):
                    side.wait_stream(torch.cuda.current_stream(ctx.device))   # dy (and everything before it) is ready
                ctx.side_used = True
            with (torch.cuda.stream(side) if side is not None else _NullCtx()):
                if side is not None:
                    ctx.stream = side.cuda_stream
                try:
                    kp = plan.kp(False)
                    nsl = 7 if plan.stem else plan.ntaps
                    # persistent workspace, handed back zeroed by the unpack kernel
                    # (a channel-padded head -- rows past plan.N are never unpacked, so whatever accumulates there over the steps is
                    #  never read -- keeps a persistent workspace too: no allocation and no fill on the weight-gradient stream)
                    persistent = bool(PERSISTENT_DW)
                    dw = plan.dw_workspace(ctx, nsl * Ny * kp) if persistent else ctx.f32(nsl * Ny * kp, zero=True)
                    if persistent:
                        ctx._dw_plans.append(plan)
                    if persistent and any(j[0] == dw.data_ptr() for j in ctx._unpack_jobs):
                        # this plan already ran in this backward (a module used twice in one forward): its workspace holds the
                        # first use's gradient, and the kernels may STORE their result (no split-K, no atomics: "dw is zero on
                        # entry").  Hand that gradient over and get the workspace back zeroed first.
                        plan.unpack_wgrad(ctx, dw, clear=True)
                    wd = _wgrad_desc(ctx, plan, x, dy, dw)
                    if fused_bnb is not None:
                        zv, zf, zm, zi, z1, z2 = fused_bnb
                        wd.bnb_z, wd.bnb_ld, wd.bnb_sB, wd.bnb_fwd = zv.ptr(), zv.ld, zv.sB, zf
                        wd.bnb_mean, wd.bnb_invstd, wd.bnb_c1, wd.bnb_c2 = zm.data_ptr(), zi.data_ptr(), z1.data_ptr(), z2.data_ptr()
                    es = ESIZE[ctx.dt]
                    # the LAST node of the tape (the RGB stem: its input needs no gradient, so nothing of the main stream is
                    # left beside its weight gradient) may take the whole chip; any other conv without a data gradient (the first
                    # SoundNet layer) still has main-stream work beside it and keeps the cap
                    tail = TAIL_WGRAD_FULL and side is not None and not x.needs_grad and getattr(ctx, "_tape_left", 0) <= 1
                    wd.max_cus = 256 if (tail or side is None) else (WGRAD_CUS_DEC if bn is None else WGRAD_CUS)
                    if DBG_SPIN_SIDE and side is not None:
                        ctx.lib.vinet_debug_spin(DBG_SPIN_SIDE, ctx.stream)
                    planes = planes_main
                    if planes is None and want_planes:
                        planes = _split_planes(ctx, x, dy, dy_planes)
                    if planes is not None:
                        # the split-bf16 weight gradient as THREE launches of the bf16 kernels (the row- / frame-streaming ones
                        # included) over hi / lo planes of both operands: dw += dy_hi x_hi + dy_lo x_hi + dy_hi x_lo.  Every bf16
                        # weight-gradient kernel adds into dw; the pending affine of x went into its planes.
                        (xh, xl), (dh, dl) = planes
                        for xa, da in ((xh, dl), (xl, dh), (xh, dh)):      # small terms first
                            wq = _wgrad_desc(ctx, plan, xa if isinstance(xa, Act) else Act(xa), da, dw)
                            wq.dtype, wq.max_cus = BF16, wd.max_cus
                            ctx.call("vinet_conv3d_wgrad", C.byref(wq), ctx.stream,
                                     tag=(_wgrad_kernel_name(ctx, wq) + " | wgrad(split x3) " + plan.site(x.v)) if PROFILER is not None else None,
                                     work=dict(flops=2.0 * M * plan.N * plan.Cin * plan.ntaps / 3,
                                               bytes=float(x.v.nvox * plan.Cin * 2 + M * plan.N * 2 + plan.N * plan.Cin * plan.ntaps * 4)))
                        if side is not None:
                            ctx.keep(*((p.v if isinstance(p, Act) else p).buf for p in (xh, xl, dh, dl)))
                    else:
                        ctx.call("vinet_conv3d_wgrad", C.byref(wd), ctx.stream,
                                 tag=(_wgrad_kernel_name(ctx, wd) + " | wgrad " + plan.site(x.v)) if PROFILER is not None else None,
                                 work=dict(flops=2.0 * M * plan.N * plan.Cin * plan.ntaps,
                                           bytes=float(x.v.nvox * plan.Cin * es + M * plan.N * es + plan.N * plan.Cin * plan.ntaps * 4)))
                    if Ny != plan.N:
                        assert plan.ntaps == 1, "channel-padded outputs are only supported for 1x1x1 convs"
                    if persistent and MULTI_UNPACK and PARAM_GRAD_HOOK is None and N_SIDE_STREAMS == 1:
                        ctx._unpack_jobs.extend(plan.unpack_jobs(dw))      # one launch for all of them at the end of backward
                    else:
                        plan.unpack_wgrad(ctx, dw, clear=persistent)
                    if side is not None:
                        ctx.keep(dw, dy.buf, x.v.buf, x.scale, x.shift, *(fused_bnb[2:] if fused_bnb is not None else ()))
                finally:
                    ctx.stream = main_ptr
            _note_param_grad(ctx, *plan.grad_targets())

        # The decoder's weight gradients (convs without BatchNorm: MFMA-bound, persistent, LDS-heavy) would run beside the
        # decoder's data gradients, which are MFMA-bound too: both lose.  Deferred, they start when the tape reaches the
        # encoder, whose BN-backward passes and pools are HBM-bound and share a CU with them at little cost.  dy and x
        # stay untouched meanwhile: gradient buffers are written once per backward and live until the tape is dropped.
        # (Under stream capture too: the wrong encoder gradients of round 3's captured step came from the per-job joins of the
        # flush -- a fan-out of redundant graph edges that ROCm 7.2 replays wrongly, see Ctx.flush_deferred -- not from the
        # deferral; with one join per batch the captured step follows the eager trajectory.)
        defer = DEFER_DECODER_WGRAD_F32S if ctx.cdt == F32S else DEFER_DECODER_WGRAD
        group = WGRAD_GROUP_CAPTURE if ctx.capturing else WGRAD_GROUP
        if defer and bn is None and ctx.side_stream() is not None:
            ctx._deferred.append(wgrad_job)
        elif group > 1 and ctx.side_stream() is not None:
            # weight-gradient jobs leave for their stream `group` at a time behind one join (fewer fork points: every join is an
            # event pair on the host and, in a replayed graph, ~12 us of cross-queue latency on the main stream's next kernel)
            ctx._deferred.append(wgrad_job)
            if len(ctx._deferred) >= group:
                ctx.flush_deferred()
        else:
            ctx.flush_deferred()
            wgrad_job()
    # ---- data gradient -------------------------------------------------------------
    if x.needs_grad:
        xv = x.v
        if DBG_SPIN_MAIN and ctx.device.type == "cuda":
            ctx.lib.vinet_debug_spin(DBG_SPIN_MAIN, ctx.stream)
        phases, full = plan.dgrad_phases((xv.T, xv.H, xv.W), (out.T, out.H, out.W), ctx.device)
        ready = x.is_grad_ready()
        dx = x.grad_view(zero=(not ready and not full))
        acc = 1 if x.is_grad_ready() else 0
        wt = plan.packed(ctx, True)
        if plan.temporal and plan.s[0] > 1 and len(phases) > 1:
            # all stride phases of a temporal data gradient in one launch where the library has a kernel for it
            # (dy is read once instead of once per phase): include/vinet_hip.h, tline == 3
            d = L.CConvDesc()
            d.dtype, d.out_dtype, d.mode = ctx.cdt, dx.dt, L.CONV_GENERIC
            d.x, d.y = dy.ct(), dx.ct()
            d.oT, d.oH, d.oW = xv.T, xv.H, xv.W
            d.sT, d.sH, d.sW = plan.s[0], 1, 1
            d.omT = d.omH = d.omW = 1
            d.ooT = d.ooH = d.ooW = 0
            d.ntaps, d.taps, d.w, d.Kp = plan.k[0], None, wt.data_ptr(), plan.kp(True)
            d.pre = L.CAffine(None, None, 0)
            d.out_scale = d.out_shift = None
            d.act, d.accumulate, d.stats = L.ACT_NONE, acc, None
            d.n_valid = plan.Cin if xv.C != plan.Cin else 0
            d.tline, d.tpad = 3, plan.p[0]
            if ctx.lib.vinet_conv3d_fuses_dgrad_phases(C.byref(d)):
                es = ESIZE[ctx.dt]
                bnb_ws = None
                if (DGRAD_BN_STATS and not acc and x.mean is not None and x.scale is not None and x.parent is None and
                        x.fold is None and xv.dt == dx.dt and xv.C == plan.Cin and dx.same_dims(xv)):
                    # x = relu(bn(z)), this launch is the first (for the stem: the only) writer of its gradient: the partial
                    # sums of that BatchNorm's backward reduce pass leave with the gradient
                    d.bnb_z, d.bnb_ld, d.bnb_sB, d.bnb_fwd = xv.ptr(), xv.ld, xv.sB, x.affine()
                    d.bnb_mean, d.bnb_invstd = x.mean.data_ptr(), x.invstd.data_ptr()
                    brows = ctx.lib.vinet_conv3d_bn_bwd_stats_rows(C.byref(d))
                    if brows > 0:
                        bnb_ws = ctx.f32(brows * 2 * xv.C)
                        d.bnb_partials = bnb_ws.data_ptr()
                ctx.call("vinet_conv3d", C.byref(d), ctx.stream,
                         tag=("conv_tsd_kernel | dgrad " + plan.site(xv)) if PROFILER is not None else None,
                         work=dict(flops=2.0 * M * plan.N * plan.Cin * plan.ntaps,
                                   bytes=float(xv.nvox * plan.Cin * es + M * plan.N * es + plan.N * plan.Cin * plan.ntaps * es)))
                phases = []
                if bnb_ws is not None:
                    x.mark_grad_ready()
                    _register_bnb(ctx, x, bnb_ws, brows)
                    return
        # The BatchNorm(s) behind x's pending affine (x itself, or the pending activation it materialises): when this launch is
        # the LAST writer of x's gradient -- every consumer recorded in forward but this one has written -- and covers it in one
        # launch, it also writes the partial sums of their backward reduce pass (VinetConvDesc::bnb_*, the shared conv epilogue)
        bx = x.alias_of if x.alias_of is not None else x
        bnb_ws, brows = None, 0
        want_bnb = (DGRAD_BN_STATS >= 2 and len(phases) == 1 and full and bx.mean is not None and bx.scale is not None and bx.relu
                    and bx.fold is None and ctx.cdt == BF16 and dx.dt == BF16 and bx.v.dt == BF16 and xv.C == plan.Cin
                    and dx.same_dims(bx.v) and x.root().grad_marks + 1 == x.root().n_readers)
        for ph in phases:
            d = L.CConvDesc()
            d.dtype, d.out_dtype, d.mode = ctx.cdt, dx.dt, L.CONV_GENERIC
            d.x, d.y = dy.ct(), dx.ct()
            d.oT, d.oH, d.oW = ph["Q"]
            d.sT = d.sH = d.sW = 1
            d.omT, d.omH, d.omW = plan.s
            d.ooT, d.ooH, d.ooW = ph["r"]
            d.ntaps, d.taps, d.w, d.Kp = ph["ntaps"], ph["taps"].data_ptr(), wt.data_ptr(), plan.kp(True)
            d.pre = L.CAffine(None, None, 0)
            d.out_scale = d.out_shift = None
            d.act, d.accumulate, d.stats = L.ACT_NONE, acc, None
            d.n_valid = plan.Cin if xv.C != plan.Cin else 0
            d.tline, d.tpad = ph["tline"], ph["tpad"]
            if want_bnb:
                zv = bx.v
                d.bnb_z, d.bnb_ld, d.bnb_sB, d.bnb_fwd = zv.ptr(), zv.ld, zv.sB, bx.affine()
                d.bnb_mean, d.bnb_invstd = bx.mean.data_ptr(), bx.invstd.data_ptr()
                brows = ctx.lib.vinet_conv3d_bn_bwd_stats_rows(C.byref(d))
                if brows > 0:
                    bnb_ws = ctx.f32(brows * 2 * zv.C)
                    d.bnb_partials = bnb_ws.data_ptr()
            es = ESIZE[ctx.dt]
            nph = len(phases)
            ctx.call("vinet_conv3d", C.byref(d), ctx.stream,
                     tag=(_conv_kernel_name(ctx, d) + ("+bnb" if bnb_ws is not None else "") + " | dgrad " + plan.site(xv)) if PROFILER is not None else None,
                     work=dict(flops=2.0 * M * plan.N * plan.Cin * plan.ntaps / nph,
                               bytes=float(xv.nvox * plan.Cin * es + M * plan.N * es + plan.N * plan.Cin * plan.ntaps * es) / nph))
        x.mark_grad_ready()
        if bnb_ws is not None:
            _register_bnb(ctx, bx, bnb_ws, brows)


# ----------------------------------------------------------------------------
# pooling / upsample
# ----------------------------------------------------------------------------

def maxpool_forward(ctx, x, k, s, p, dst=None):
    xv = x.v
    od = tuple((d + 2 * pp - kk) // ss + 1 for d, kk, ss, pp in zip((xv.T, xv.H, xv.W), k, s, p))
    if dst is None:
        dst = Act(View.alloc(xv.B, od[0], od[1], od[2], xv.C, xv.dt, xv.device))
    out = dst.v
    assert (out.T, out.H, out.W, out.C) == (od[0], od[1], od[2], xv.C) and dst.plain
    pd = L.CPoolDesc(xv.dt, k[0], k[1], k[2], s[0], s[1], s[2], p[0], p[1], p[2])
    rec = ctx.recording and x.needs_grad
    _note_reader(ctx, x)
    am = torch.empty(out.nvox * out.C, dtype=torch.uint8, device=xv.device) if rec else None
    ptag = "maxpool k%dx%dx%d s%dx%dx%d C%d in%dx%dx%dx%d" % (k + s + (xv.C, xv.B, xv.T, xv.H, xv.W))
    es = ESIZE[xv.dt]
    ctx.call("vinet_maxpool3d", C.byref(pd), C.byref(xv.ct()), x.affine(), C.byref(out.ct()), _ptr(am), ctx.stream,
             tag="maxpool_fwd_kernel | " + ptag, work=dict(flops=0.0, bytes=float((xv.nvox + out.nvox) * xv.C * es)))
    dst.needs_grad = x.needs_grad
    if rec:
        def bwd():
            dy = dst.grad_view()
            dx = x.grad_view()
            ctx.call("vinet_maxpool3d_bwd", C.byref(pd), C.byref(dy.ct()), am.data_ptr(), C.byref(dx.ct()),
                     1 if x.is_grad_ready() else 0, ctx.stream, tag="maxpool_bwd_kernel | " + ptag,
                     work=dict(flops=0.0, bytes=float((xv.nvox + out.nvox) * xv.C * es + out.nvox * xv.C)))
            x.mark_grad_ready()
        ctx.record(bwd)
    return dst


def upsample2x_forward(ctx, x, dst=None):
    assert x.plain, "upsample input must be materialised"
    xv = x.v
    if dst is None:
        dst = Act(View.alloc(xv.B, xv.T, 2 * xv.H, 2 * xv.W, xv.C, xv.dt, xv.device))
    out = dst.v
    assert dst.plain
    ctx.call("vinet_upsample2x", C.byref(xv.ct()), C.byref(out.ct()), xv.dt, ctx.stream)
    _note_reader(ctx, x)
    dst.needs_grad = x.needs_grad
    if ctx.recording and x.needs_grad:
        def bwd():
            dy = dst.grad_view()
            dx = x.grad_view()
            if UPSAMPLE_BWD_RELU and x.act_out == L.ACT_RELU and not x.is_grad_ready() and x.parent is None and dx.dt == xv.dt:
                # conv -> ReLU -> upsample (the decoder): this pass is the only writer of the conv's output gradient, the ReLU's
                # backward goes with it (no vinet_act_bwd pass in the conv's backward)
                ctx.call("vinet_upsample2x_bwd_relu", C.byref(dy.ct()), C.byref(dx.ct()), C.byref(xv.ct()), dx.dt, ctx.stream)
                x.mark_grad_ready()
                x.grad_masked = x.root().grad_marks
                return
            ctx.call("vinet_upsample2x_bwd", C.byref(dy.ct()), C.byref(dx.ct()), dx.dt, 1 if x.is_grad_ready() else 0, ctx.stream)
            x.mark_grad_ready()
        ctx.record(bwd)
    return dst


def unfold1d_forward(ctx, x, k, stride, pad):
    """[B,L,1,1,C] signal (channel 0) -> [B,(L+2p-k)/s+1,1,1,k] windows (vinet_unfold1d): the input of SoundNet's first conv
    as a pointwise conv.  The waveform needs no gradient, so there is no backward."""
    assert x.plain and not x.needs_grad, "unfold1d: raw input signals only"
    xv = x.v
    assert xv.H == 1 and xv.W == 1
    To = (xv.T + 2 * pad - k) // stride + 1
    dst = Act(View.alloc(xv.B, To, 1, 1, k, xv.dt, xv.device))
    ctx.call("vinet_unfold1d", C.byref(xv.ct()), C.byref(dst.v.ct()), xv.dt, stride, pad, ctx.stream)
    dst.needs_grad = False
    return dst


# ----------------------------------------------------------------------------
# autograd entry point: one node per root module call
# ----------------------------------------------------------------------------

# How parameter gradients leave a root module's backward:
#   "fused"    (default) the tape's kernels accumulate straight into `param.grad` (views of the optimizer's flat buffer
#              when vinet_amd.optim.Adam is used) and the autograd node returns None for parameters: no extra pass;
#   "autograd" the node RETURNS the parameter gradients and autograd's own AccumulateGrad nodes add them to `.grad`,
#              so `torch.autograd.grad(loss, params)`, gradient hooks and torch's DistributedDataParallel (whose
#              reducer hangs its bucket hooks on those nodes, train.py:181-185 wraps the reference module the same
#              way with nn.DataParallel) see this module like any nn.Module.  Costs one fill + one add per parameter.
_PARAM_GRAD_MODE = "fused"


def set_param_grad_mode(mode):
    global _PARAM_GRAD_MODE, PARAM_GRAD_MODE
    assert mode in ("fused", "autograd")
    _PARAM_GRAD_MODE = PARAM_GRAD_MODE = mode


def param_grad_mode():
    return _PARAM_GRAD_MODE


class _TapeFn(torch.autograd.Function):
    """Runs `body.run` on the engine.  Forward records the tape; backward seeds
    the output gradients, replays the tape (which accumulates parameter
    gradients straight into `.grad`) and returns the input gradients -- and, in
    "autograd" mode, the parameter gradients (see _PARAM_GRAD_MODE)."""

    @staticmethod
    def forward(fctx, body, n_in, *tensors):
        inputs = tensors[:n_in]
        ectx = body.make_ctx(inputs[0].device, record=True)
        outs, state = body.run(ectx, inputs, [t.requires_grad for t in inputs])
        fctx.body, fctx.ectx, fctx.state, fctx.n_par = body, ectx, state, len(tensors) - n_in
        fctx.params = tensors[n_in:]
        return tuple(outs)

    @staticmethod
    def backward(fctx, *gouts):
        ectx = fctx.ectx
        ectx.stream = _stream_for(ectx.device)
        if _PARAM_GRAD_MODE == "fused":
            gin = fctx.body.seed(ectx, fctx.state, gouts)
            return (None, None) + tuple(gin) + (None,) * fctx.n_par
        # "autograd": let the tape write into fresh buffers, hand them back, leave `.grad` to AccumulateGrad
        params = fctx.params
        saved = [p.grad for p in params]
        for p in params:
            p.grad = None
        try:
            gin = fctx.body.seed(ectx, fctx.state, gouts)
            grads = [p.grad for p in params]
        finally:
            for p, g in zip(params, saved):
                p.grad = g
        base = 2 + len(gin)
        grads = [g if fctx.needs_input_grad[base + i] else None for i, g in enumerate(grads)]
        return (None, None) + tuple(gin) + tuple(grads)


def run_root(body, inputs, params):
    """Dispatch a root module call: one autograd node when gradients are wanted,
    a plain engine call otherwise.  Returns a tuple of output tensors."""
    want = torch.is_grad_enabled() and (any(t.requires_grad for t in inputs) or any(p.requires_grad for p in params))
    if want:
        # parameters are passed so autograd sees the dependency; their gradients
        # are accumulated into `.grad` by the tape itself
        return _TapeFn.apply(body, len(inputs), *inputs, *[p for p in params if p.requires_grad])
    ectx = body.make_ctx(inputs[0].device, record=False)
    outs, _ = body.run(ectx, inputs, [False] * len(inputs))
    return tuple(outs)


class BlockBody:
    """root wrapper for modules mapping NCDHW fp32 tensors to NCDHW fp32 tensors
    (BasicConv3d, SepConv3d, Mixed_*, BackBoneS3D, decoders used standalone)."""

    def __init__(self, module, fwd, cpad=None):
        self.module, self.fwd, self.cpad = module, fwd, cpad

    def make_ctx(self, device, record):
        return Ctx(device, getattr(self.module, "compute_dtype", None), self.module.training, record)

    def run(self, ectx, inputs, req):
        acts = [import_ncdhw(ectx, t, self.cpad, needs_grad=r) for t, r in zip(inputs, req)]
        outs = self.fwd(ectx, *acts)
        outs = list(outs) if isinstance(outs, (list, tuple)) else [outs]
        for o in outs:
            _note_reader(ectx, o)        # (the seed of backward writes its gradient: import_grad_ncdhw)
        tens = [export_ncdhw(ectx, o, self.out_channels(o)) for o in outs]
        return tens, (acts, outs, [t.shape[1] for t in inputs])

    @staticmethod
    def out_channels(o):
        return o.v.C

    def seed(self, ectx, state, gouts):
        acts, outs, cin = state
        for o, g in zip(outs, gouts):
            import_grad_ncdhw(ectx, o, g)
        ectx.run_backward()
        return [export_grad_ncdhw(ectx, a, c) if (a.needs_grad and a.is_grad_ready()) else None for a, c in zip(acts, cin)]
