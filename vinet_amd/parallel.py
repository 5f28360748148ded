"""Clip-level data parallelism: one process per GPU, RCCL over xGMI.

Replaces the reference's single-process `nn.DataParallel` (train.py:181-185).
Clips are independent, so the only exchange per step is the gradient reduction: an
all-reduce(SUM) over the optimizer's flat fp32 gradient buffer (124.4 MB for
ViNet-32, no packing: the wgrad kernels accumulate straight into that buffer),
averaged by folding 1/world into the fused Adam kernel.

  * `GradientBuckets`: the flat buffer is cut into contiguous buckets in REVERSE
    parameter order (the decoder -- 23.2 M of the 31.1 M parameters -- finishes its
    backward first); a bucket's all-reduce is issued, asynchronously and on RCCL's
    own stream, the moment the tape has launched the last kernel that writes one of
    its gradients (engine.PARAM_GRAD_HOOK), ordered behind BOTH the main stream and
    the weight-gradient side stream by events -- so the exchange overlaps the rest
    of the backward pass.  `allreduce_gradients` is the one-shot form.
  * parameters that never receive a gradient -- SoundNet's conv8_objs / conv8_scns,
    11.48 M parameters the reference's forward never touches (model.py:788-791) --
    stay out of the buffer: `trainable_parameters(model)`.
  * BatchNorm statistics stay per replica, exactly as under DataParallel (whose
    replica-0 statistics are the ones that survive a step); `broadcast_buffers`
    makes every replica adopt rank 0's before a checkpoint or validation.
  * the modules are also plain nn.Modules to torch's DistributedDataParallel:
    `ddp_wrap(model)` switches the engine to engine.set_param_grad_mode("autograd").
"""
import os

import torch
import torch.distributed as dist


# Functional testing on ONE GPU: treat a single-rank process group as distributed, so that every collective of the N > 1 path
# (flat broadcast, bucketed all-reduce on the communication stream with its event joins, scalar means) really goes through
# RCCL.  RCCL refuses two ranks on one device, so this is the only way to execute it without a multi-GPU node.
FORCE_COLLECTIVES = False      # (bench.py --force-collectives; tests set the attribute)


def distributed():
    return dist.is_initialized() and (dist.get_world_size() > 1 or FORCE_COLLECTIVES)


def _free_port():
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def _address_in_use(err):
    msg = str(err).lower()
    return "eaddrinuse" in msg or "address already in use" in msg or "errno: 98" in msg


def init_from_env(backend=None):
    """torchrun / torch.distributed.run environment -> (rank, world, local_rank, device).

    Rendezvous, in this order:
      * VINET_RDZV_FILE=<path>: a `FileStore` -- no TCP port at all (launchers without torchrun; the functional tests);
      * MASTER_ADDR / MASTER_PORT from the launcher (torch.distributed.run hands its workers the agent's store, so nothing
        is bound here);
      * a ONE-rank group (world == 1 with FORCE_COLLECTIVES) has no peer to agree a port with: if the port -- the default
        29500 or one from the environment -- is taken (EADDRINUSE), it retries on ports the kernel hands out.  A multi-rank
        group whose port is taken cannot be repaired from inside one rank: the error names the two ways out."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    use_cuda = torch.cuda.is_available()
    # (local % device_count only matters for functional tests that oversubscribe one GPU)
    device = torch.device("cuda:%d" % (local % torch.cuda.device_count())) if use_cuda else torch.device("cpu")
    if use_cuda:
        torch.cuda.set_device(device)
    if (world > 1 or FORCE_COLLECTIVES) and not dist.is_initialized():
        backend = backend or ("nccl" if use_cuda else "gloo")
        rdzv_file = os.environ.get("VINET_RDZV_FILE")
        if rdzv_file:
            dist.init_process_group(backend, store=dist.FileStore(rdzv_file, world), rank=rank, world_size=world)
            return rank, world, local, device
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        for attempt in range(8):
            try:
                dist.init_process_group(backend, rank=rank, world_size=world)
                break
            except Exception as e:      # torch raises DistNetworkError / RuntimeError depending on where the bind fails
                if not _address_in_use(e):
                    raise
                if world > 1 or attempt == 7:
                    raise RuntimeError("rendezvous port %s:%s is in use: pass another --master-port to the launcher, or set "
                                       "VINET_RDZV_FILE to a fresh path for a file rendezvous"
                                       % (os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"])) from e
                os.environ["MASTER_PORT"] = str(_free_port())
    return rank, world, local, device


def shard_batch(global_batch, rank, world):
    """[start, stop) of this rank's clips; equal shards (the loss is a batch mean,
    loss.py:38, so equal local batches make the averaged gradient exact)."""
    assert global_batch % world == 0, "global batch must divide evenly over ranks"
    b = global_batch // world
    return rank * b, (rank + 1) * b


def broadcast_parameters(optimizer, src=0):
    """make every replica start from rank `src`'s weights (one flat broadcast)."""
    if distributed():
        dist.broadcast(optimizer.flat_p, src=src)
        from . import engine
        engine.bump_weights_epoch()


def allreduce_gradients(optimizer, async_op=False):
    """SUM all-reduce of the flat gradient buffer; the mean is applied inside Adam."""
    if not distributed():
        optimizer.grad_scale = 1.0
        return None
    optimizer.grad_scale = 1.0 / dist.get_world_size()
    return dist.all_reduce(optimizer.flat_g, op=dist.ReduceOp.SUM, async_op=async_op)


def allreduce_scalar_mean(t):
    if distributed():
        t = t.detach().clone()
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        t /= dist.get_world_size()
    return t


def trainable_parameters(model):
    """parameters that take part in training: requires_grad and reachable from the forward pass.  Modules list the
    ones their forward never uses in `unused_parameter_names` (AViNet: SoundNet's two classification heads)."""
    skip = set()
    for prefix, mod in model.named_modules():
        for name in getattr(mod, "unused_parameter_names", ()):
            skip.add((prefix + "." if prefix else "") + name)
    return [p for n, p in model.named_parameters() if p.requires_grad and n not in skip]


def broadcast_buffers(model, src=0):
    """every replica adopts rank `src`'s buffers (BatchNorm running statistics: 85.5 KB for ViNet-32)"""
    if distributed():
        for b in model.buffers():
            dist.broadcast(b, src=src)


def ddp_wrap(model, **kw):
    """torch.nn.parallel.DistributedDataParallel around a vinet_amd module: the root autograd node then returns its
    parameter gradients to autograd (engine 'autograd' mode) so DDP's bucket hooks fire."""
    from . import engine
    engine.set_param_grad_mode("autograd")
    dev = next(model.parameters()).device
    if dev.type == "cuda":
        kw.setdefault("device_ids", [dev.index])
    return torch.nn.parallel.DistributedDataParallel(model, **kw)


class GradientBuckets:
    """Bucketed all-reduce of vinet_amd.optim.Adam's flat gradient buffer, overlapped with the backward pass.

        buckets = GradientBuckets(optimizer)           # once
        optimizer.zero_grad(); buckets.begin_step()
        loss.backward()                                # buckets go out as their gradients complete
        buckets.finish(); optimizer.step()
    """

    def __init__(self, optimizer, bucket_bytes=25 << 20):
        self.opt = optimizer
        ps, offs = optimizer._params, optimizer._offs
        total = optimizer.flat_g.numel()
        self.buckets = []          # [lo, hi) element ranges of flat_g, last parameters first
        self.index = {}
        hi, members = total, []
        for p, o in zip(reversed(ps), reversed(offs)):
            members.append(p)
            if (hi - o) * 4 >= bucket_bytes:
                self._add(o, hi, members)
                hi, members = o, []
        if members:
            self._add(0, hi, members)
        self._works, self._fired, self._members, self._launched = [], [], [], []
        self._seen, self._last_ctx = {}, None
        self._comm_streams = {}
        # timing=True: per-bucket HIP events (issue / completion on the communication stream) and the backward pass's own
        # extent on the main stream -> `timeline()` after finish(): how much of the exchange hid under backward
        self.timing = False
        self._ev = []

    def _add(self, lo, hi, members):
        for p in members:
            self.index[id(p)] = len(self.buckets)
        self.buckets.append((lo, hi, len(members)))

    def active(self):
        return distributed()

    def begin_step(self):
        """ONE backward per begin_step / finish pair: a bucket leaves when every one of its parameters has reported its
        gradient once (a set of parameter ids, so a parameter whose gradient is written by two tape nodes -- a module used
        twice in forward -- is counted once, at its LAST report: see _on_param).  Gradient accumulation over several
        backward() calls must call finish() only after the last one and not use the overlapped form (use
        allreduce_gradients after the last backward instead)."""
        from . import engine
        self._works = []
        self._members = [n for _, _, n in self.buckets]
        self._fired = [set() for _ in self.buckets]
        self._launched = [False] * len(self.buckets)
        self._seen, self._last_ctx = {}, None
        if self.active():
            engine.PARAM_GRAD_HOOK = self._on_param
        self.opt.grad_scale = 1.0 / dist.get_world_size() if self.active() else 1.0
        self._ev, self._t0, self._t1 = [], None, None
        if self.timing and self.active() and self.opt.flat_g.is_cuda:
            self._t0 = torch.cuda.Event(enable_timing=True)
            self._t0.record()

    def expect_reports(self, counts):
        """{id(param): number of tape nodes that write its gradient per backward} for models that use a parameter more than
        once in forward (none of the ViNet / AViNet modules do): the bucket then waits for the last report."""
        self._expected = dict(counts)

    def _on_param(self, ctx, p):
        b = self.index.get(id(p))
        if b is None:
            return
        # a report behind the bucket's launch would add an un-reduced contribution to an already summed slice
        assert not self._launched[b], ("gradient of a parameter reported after its bucket's all-reduce was issued: a parameter "
                                       "written by several tape nodes (or a second backward inside one begin_step/finish "
                                       "pair) needs GradientBuckets.expect_reports")
        self._last_ctx = ctx
        need = getattr(self, "_expected", {}).get(id(p), 1)
        seen = self._seen.get(id(p), 0) + 1 if need > 1 else 1
        if need > 1:
            self._seen[id(p)] = seen
            if seen < need:
                return
        self._fired[b].add(id(p))
        if len(self._fired[b]) == self._members[b]:
            self._launch(b, ctx)

    def _launch(self, b, ctx=None):
        lo, hi, _ = self.buckets[b]
        t = self.opt.flat_g[lo:hi]
        self._launched[b] = True
        if t.is_cuda:
            dev = t.device
            comm = self._comm_streams.get(dev.index)
            if comm is None:
                comm = self._comm_streams[dev.index] = torch.cuda.Stream(dev)
            comm.wait_stream(torch.cuda.current_stream(dev))              # BatchNorm / bias gradients and everything before them
            for side in (ctx.side_streams() if ctx is not None else []):
                comm.wait_stream(side)                                    # the weight-gradient kernels launched so far
            with torch.cuda.stream(comm):
                if self._t0 is not None:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                work = dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=True)
                self._works.append(work)
                if self._t0 is not None:
                    # RCCL runs the collective on the process group's OWN stream: an event recorded on `comm` right behind the
                    # call would stamp its issue, not its end.  work.wait() makes `comm` wait for the collective's end event
                    # (no host block with the nccl backend), so e1 is the completion stamp.  Timing mode only.
                    work.wait()
                    e1.record()
                    self._ev.append((b, (hi - lo) * 4, e0, e1))
        else:
            self._works.append(dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=True))

    def finish(self):
        """buckets whose parameters got no (or not every) gradient this step go out now; then the optimizer's stream waits
        for every bucket.  Called after backward(): engine.Ctx.run_backward has already made the main stream wait for the
        weight-gradient side stream, so the leftovers only need the main stream -- they are launched with the last tape
        context anyway, which joins the side stream explicitly as well."""
        from . import engine
        engine.PARAM_GRAD_HOOK = None
        if not self.active():
            return
        if getattr(self, "_t0", None) is not None:
            self._t1 = torch.cuda.Event(enable_timing=True)       # end of the backward pass on the main stream
            self._t1.record()
        for b in range(len(self.buckets)):
            if not self._launched[b]:
                self._launch(b, self._last_ctx)
        for w in self._works:
            w.wait()
        self._works = []
        self._last_ctx = None

    def timeline(self):
        """after finish() of a step taken with timing = True (and a device synchronisation): per bucket the issue and completion
        time of its all-reduce in ms from begin_step(), the end of the backward pass, and the fraction of the summed all-reduce
        time that lay inside the backward pass (hidden: the optimizer never waited for it)"""
        if not self._ev or self._t0 is None or self._t1 is None:
            return None
        torch.cuda.synchronize()
        bwd_end = self._t0.elapsed_time(self._t1)
        rows, tot, hid = [], 0.0, 0.0
        for b, nbytes, e0, e1 in self._ev:
            t_is, t_done = self._t0.elapsed_time(e0), self._t0.elapsed_time(e1)
            rows.append(dict(bucket=b, mb=round(nbytes / 1e6, 2), issue_ms=round(t_is, 3), done_ms=round(t_done, 3)))
            tot += max(t_done - t_is, 0.0)
            hid += max(min(t_done, bwd_end) - min(t_is, bwd_end), 0.0)
        return dict(buckets=rows, backward_end_ms=round(bwd_end, 3), allreduce_ms=round(tot, 3),
                    hidden_frac=(hid / tot if tot > 0 else None))
