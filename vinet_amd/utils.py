"""Loss glue and meters -- drop-in for the reference's utils.py:9-59."""
import torch

from .loss import cc, kldiv, similarity


def get_loss(pred_map, gt, args):
    """utils.py:9-20: weighted sum of the enabled losses.  (The reference allocates
    its accumulator with `.cuda()`; here it lives on pred_map's device.)"""
    loss = None

    def add(term):
        nonlocal loss
        loss = term if loss is None else loss + term

    if getattr(args, "kldiv", False):
        add(args.kldiv_coeff * kldiv(pred_map, gt))
    if getattr(args, "cc", False):
        add(args.cc_coeff * cc(pred_map, gt))
    if getattr(args, "l1", False):
        raise NotImplementedError("--l1 is broken in the reference too (utils.py:15 uses an unbound `criterion`)")
    if getattr(args, "sim", False):
        add(args.sim_coeff * similarity(pred_map, gt))
    if loss is None:
        loss = torch.zeros((), dtype=torch.float32, device=pred_map.device)
    return loss.reshape(1)


def loss_func(pred_map, gt, args):
    """utils.py:22-39."""
    assert pred_map.size() == gt.size()
    if pred_map.dim() == 4:
        assert pred_map.size(0) == args.batch_size
        pred_map = pred_map.permute((1, 0, 2, 3))
        gt = gt.permute((1, 0, 2, 3))
        total = None
        for i in range(pred_map.size(0)):
            term = get_loss(pred_map[i].contiguous(), gt[i].contiguous(), args)
            total = term if total is None else total + term
        return total / pred_map.size(0)
    return get_loss(pred_map, gt, args)


class AverageMeter:
    """Running mean of a logged quantity; same public surface as the reference's meter
    (utils.py:41-59: attributes val / sum / count / avg, methods reset() and update(val, n=1))."""

    __slots__ = ("val", "sum", "count")

    def __init__(self):
        self.reset()

    def reset(self):
        self.val, self.sum, self.count = 0, 0, 0

    @property
    def avg(self):
        return self.sum / self.count if self.count else 0

    def update(self, val, n=1):
        self.val = val
        self.sum, self.count = self.sum + val * n, self.count + n


def num_params(model):
    return sum(dict((p.data_ptr(), p.numel()) for p in model.parameters()).values())
