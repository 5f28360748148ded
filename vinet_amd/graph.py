"""hipGraph capture of the inference forward.

At batch 1 a ViNet forward is ~300 kernel launches for ~1 ms of GPU work, i.e. host
launch bound.  All launches go through torch's current stream, all memory comes from
torch's caching allocator and the engine never synchronises, so the whole forward can
be captured once into a hipGraph (`torch.cuda.CUDAGraph` is hipGraph on ROCm) and
replayed with one launch per frame -- the "HIP graphs instead of a tracing compiler"
design point.  Weight packs and tap tables are built during the warm-up calls, so the
captured region contains only the steady-state kernels.
"""
import torch


class GraphedInference:
    """y = GraphedInference(model, example_inputs)(inputs): replays a captured forward.

    Inputs must keep the example's shapes; parameters must not be modified after capture
    (re-create the wrapper after loading new weights)."""

    def __init__(self, model, *example_inputs, warmup=2):
        assert all(t.is_cuda for t in example_inputs), "graph capture needs GPU tensors"
        self.model = model.eval()
        self.static_in = [t.clone() for t in example_inputs]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(warmup):          # builds weight packs, tap tables, LDS attributes
                self.model(*self.static_in)
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph), torch.no_grad():
            self.static_out = self.model(*self.static_in)

    @torch.no_grad()
    def __call__(self, *inputs):
        for dst, src in zip(self.static_in, inputs):
            dst.copy_(src)
        self.graph.replay()
        return self.static_out


class GraphedTrainStep:
    """One training step -- zero_grad, forward, loss, backward -- captured into a hipGraph; the fused Adam launch follows each
    replay (its bias corrections are host scalars).  At the reference's batch sizes (train.py:43: a GLOBAL batch of 8, i.e.
    1-8 clips per GPU) a step is ~1500 kernel launches issued from Python for a few milliseconds of GPU work: host bound.
    Everything the tape does is capturable as it is: all launches go to torch's current stream or to the weight-gradient
    side stream (forked from and joined back to it by events), all memory is the caching allocator's, the weight packs of
    the whole model are rebuilt by ONE multi-pack launch at the top of the step (captured: it re-runs at every replay),
    gradients land in the optimizer's flat buffer at fixed addresses.

        step = GraphedTrainStep(model, optimizer, vinet_amd.loss.kldiv, (clips,), gt)
        loss = step((clips,), gt)          # same shapes as at capture

    Data parallel (one process per GPU): the replayed graph leaves the local gradients in the optimizer's flat buffer; ONE
    all-reduce of that buffer (parallel.allreduce_gradients: RCCL, 1 / world folded into Adam) runs between the replay and the
    fused Adam launch.  At the batch sizes a captured step is for (1-8 clips per GPU, a ~20 ms step) the 124 MB exchange is
    ~1.4 ms on a ring over xGMI: issued in one piece behind the graph it is not worth cutting the graph into segments for.
    Without a GPU (host-logic tests on the ABI emulator) nothing is captured: the same call sequence runs eagerly.
    """

    def __init__(self, model, optimizer, loss_fn, inputs, gt, warmup=2, debug_dot=None, keep_graph=False, launch_log=False):
        self.capture = all(t.is_cuda for t in inputs) and gt.is_cuda
        if not self.capture:
            from . import _lib
            assert _lib.is_test_double(), "graph capture needs GPU tensors"
        self.model, self.opt, self.loss_fn = model.train(), optimizer, loss_fn
        self.static_in = [t.clone() for t in inputs]
        self.static_gt = gt.clone()
        self._bns = [m for m in model.modules() if hasattr(m, "note_training_step")]
        if not self.capture:
            return
        # The warm-up steps are REAL steps (they have to be: they build weight packs, tap tables, persistent workspaces, LDS
        # attributes and autograd's own set-up), so everything they advance is snapshotted and put back: parameters, Adam
        # moments and step count, BatchNorm running statistics and the lazily counted num_batches_tracked.  A run that
        # builds the graph then follows the eager trajectory from the first user-visible step.
        import gc
        gc.collect()          # plans of models that died earlier leave the weight-pack registry now, not in the middle of the capture
        opt = self.opt
        snap_opt = [t.clone() for t in (opt.flat_p, opt.flat_m, opt.flat_v)] if hasattr(opt, "flat_p") else None
        snap_step = getattr(opt, "_step", None)
        snap_sd = None if snap_opt is not None else {k: (v.clone() if torch.is_tensor(v) else v) for k, v in opt.state_dict().items()}
        snap_par = None if snap_opt is not None else [p.detach().clone() for p in model.parameters()]
        snap_buf = [(b, b.clone()) for b in model.buffers()]
        snap_pend = [(bn, bn.__dict__.get("_vinet_pending", 0)) for bn in self._bns]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):          # weight packs, tap tables, persistent workspaces, LDS attributes, autograd's own set-up
                self._body()
                self.opt.step()
            with torch.no_grad():
                if snap_opt is not None:
                    for dst, src in zip((opt.flat_p, opt.flat_m, opt.flat_v), snap_opt):
                        dst.copy_(src)
                    opt._step = snap_step
                else:
                    for p_, v_ in zip(model.parameters(), snap_par):
                        p_.copy_(v_)
                for b, v_ in snap_buf:
                    b.copy_(v_)
            from . import engine
            engine.bump_weights_epoch()       # the packs built from the warm-up weights are stale
        for bn, pend in snap_pend:
            bn.__dict__["_vinet_pending"] = pend
            bn.__dict__.setdefault("_vinet_fold", {}).clear()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        if snap_sd is not None:
            import copy
            opt.load_state_dict(copy.deepcopy(snap_sd))
        self.graph = torch.cuda.CUDAGraph(keep_graph=True) if keep_graph else torch.cuda.CUDAGraph()
        if debug_dot:                          # hipGraphDebugDotPrint of the captured step (tools/dot_edges.py)
            self.graph.enable_debug_mode()
        # (a garbage collection inside the capture could drop plans from the pack registry, whose job table would then be
        # rebuilt -- a host-to-device copy -- in the captured region)
        was_enabled = gc.isenabled()
        gc.disable()
        try:
            with torch.cuda.graph(self.graph):
                if launch_log:                 # (which launch went to which stream, in issue order)
                    engine.LAUNCH_LOG = self.launch_log = []
                    self.capture_stream = torch.cuda.current_stream().cuda_stream
                self.static_loss = self._body()
        finally:
            engine.LAUNCH_LOG = None
            if was_enabled:
                gc.enable()
        if debug_dot:
            self.graph.debug_dump(debug_dot)
        for bn, pend in snap_pend:           # (the captured call ran the Python forward once: its host-side step count does not count)
            bn.__dict__["_vinet_pending"] = pend

    def _body(self):
        self.opt.zero_grad()
        loss = self.loss_fn(self.model(*self.static_in), self.static_gt)
        loss.backward()
        return loss.detach()

    def __call__(self, inputs, gt):
        from . import parallel
        for dst, src in zip(self.static_in, inputs):
            if dst.data_ptr() != src.data_ptr():
                dst.copy_(src)
        if self.static_gt.data_ptr() != gt.data_ptr():
            self.static_gt.copy_(gt)
        if self.capture:
            self.graph.replay()
            for bn in self._bns:             # (host-side bookkeeping the captured body did once: num_batches_tracked)
                bn.note_training_step()
        else:
            self.static_loss = self._body()
        parallel.allreduce_gradients(self.opt)     # N > 1: one RCCL all-reduce of the flat buffer (no-op on one rank)
        self.opt.step()
        return self.static_loss
