"""Fused Adam over one flat fp32 buffer (train.py:188,217 use torch.optim.Adam).

`Adam(params, lr)` flattens the parameters into a single fp32 buffer (each
`param.data` becomes a view of it), keeps one flat gradient buffer whose views
are installed as `param.grad` (so the engine's wgrad kernels accumulate straight
into it and the RCCL all-reduce needs no packing), and performs the whole update
with one `vinet_adam_step` launch.  Same math as torch.optim.Adam with default
betas / eps, no weight decay, no amsgrad.
"""
import torch

from . import _lib as L
from . import engine as E


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        params = [p for p in params if p.requires_grad]
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        assert len(self.param_groups) == 1, "vinet_amd.optim.Adam updates one flat buffer with one set of hyper-parameters: pass a single parameter group"

        self._flatten()
        self._step = 0
        self.grad_scale = 1.0   # e.g. 1/world_size after a SUM all-reduce

    def _flatten(self):
        ps = [p for g in self.param_groups for p in g["params"]]
        assert ps, "no parameters"
        dev = ps[0].device
        assert all(p.dtype == torch.float32 and p.device == dev for p in ps)
        # 16-byte aligned slots
        offs, n = [], 0
        for p in ps:
            offs.append(n)
            n += (p.numel() + 3) // 4 * 4
        self.flat_p = torch.zeros(n, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(n, dtype=torch.float32, device=dev)
        self.flat_m = torch.zeros(n, dtype=torch.float32, device=dev)
        self.flat_v = torch.zeros(n, dtype=torch.float32, device=dev)
        for p, o in zip(ps, offs):
            view = self.flat_p[o:o + p.numel()].view(p.shape)
            view.copy_(p.data)
            p.data = view
            p.grad = self.flat_g[o:o + p.numel()].view(p.shape)
        self._params, self._offs = ps, offs
        E.bump_weights_epoch()

    def zero_grad(self, set_to_none=False):
        """gradients stay allocated (views of the flat buffer); one fill kernel."""
        lib = L.get()
        L.check(lib.vinet_fill_f32(self.flat_g.data_ptr(), self.flat_g.numel(), 0.0, E._stream_for(self.flat_g.device)),
                "vinet_fill_f32")
        for p, o in zip(self._params, self._offs):
            if p.grad is None or p.grad.data_ptr() != self.flat_g.data_ptr() + 4 * o:
                p.grad = self.flat_g[o:o + p.numel()].view(p.shape)

    # the moments and the step count live in flat buffers outside Optimizer.state: carry them through
    # state_dict() / load_state_dict() so a resumed run keeps its moments and bias correction
    def state_dict(self):
        sd = super().state_dict()
        sd["vinet_flat"] = dict(m=self.flat_m.detach().cpu().clone(), v=self.flat_v.detach().cpu().clone(), step=self._step,
                                numel=[p.numel() for p in self._params])
        return sd

    def load_state_dict(self, state_dict):
        state_dict = dict(state_dict)
        flat = state_dict.pop("vinet_flat", None)
        super().load_state_dict(state_dict)
        assert len(self.param_groups) == 1
        if flat is not None:
            assert flat["numel"] == [p.numel() for p in self._params], "optimizer state belongs to a different parameter list"
            self.flat_m.copy_(flat["m"])
            self.flat_v.copy_(flat["v"])
            self._step = int(flat["step"])

    @torch.no_grad()
    def step(self, closure=None):
        assert closure is None
        g = self.param_groups[0]
        self._step += 1
        b1, b2 = g["betas"]
        bc1 = 1.0 - b1 ** self._step
        bc2 = 1.0 - b2 ** self._step
        lib = L.get()
        L.check(lib.vinet_adam_step(self.flat_p.data_ptr(), self.flat_g.data_ptr(), self.flat_m.data_ptr(),
                                    self.flat_v.data_ptr(), self.flat_p.numel(), float(g["lr"]), float(b1), float(b2),
                                    float(g["eps"]), float(bc1), float(bc2), float(self.grad_scale),
                                    E._stream_for(self.flat_p.device)), "vinet_adam_step")
        E.bump_weights_epoch()
