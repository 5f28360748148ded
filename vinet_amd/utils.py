"""Loss glue, meters and saliency-map post-processing -- drop-in for the reference's utils.py:9-78.

`blur` / `img_save` keep the reference's names and meaning (utils.py:61-78) but work on device tensors:
cv2.resize + cv2.GaussianBlur + the make_grid normalisation + uint8 conversion run as HIP kernels
(libvinet_hip.so: vinet_resize_blur, vinet_minmax, vinet_normalize_u8), so a predicted map leaves the GPU once, as
bytes.  `resize_blur` and `to_uint8` are the fused forms the harness uses.
"""
import torch

from . import _lib as L
from . import engine as E
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
        # (the reference asserts size(0) == args.batch_size, utils.py:26; under one process per GPU the tensor holds
        #  this rank's shard of the global batch and the last batch of an epoch may be partial, so only the shape
        #  agreement above is required)
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


# ---- post-processing of predicted maps (generate_result.py:95-104, train.py:251-253, utils.py:61-78) ----------
def _maps3(x):
    assert x.dim() in (2, 3), "expected a [H,W] map or a [B,H,W] batch of maps"
    m = x.detach()
    if m.dtype != torch.float32 or not m.is_contiguous():
        m = m.float().contiguous()
    return m.view((-1,) + tuple(m.shape[-2:]))


@torch.no_grad()
def resize_blur(smap, size=None, return_minmax=False):
    """cv2.GaussianBlur(cv2.resize(smap, (W_out, H_out)), (11, 11), 0) on device.  `size` = (H_out, W_out) (None: no
    resize -- cv2.resize to the same size is the identity); [H,W] or [B,H,W] float maps in, float32 maps out."""
    m = _maps3(smap)
    B, H, W = m.shape
    oH, oW = (H, W) if size is None else (int(size[0]), int(size[1]))
    out = torch.empty((B, oH, oW), dtype=torch.float32, device=m.device)
    mm = torch.empty((B, 2), dtype=torch.int32, device=m.device) if return_minmax else None
    L.check(L.get().vinet_resize_blur(m.data_ptr(), B, H, W, out.data_ptr(), oH, oW, mm.data_ptr() if mm is not None else None,
                                      E._stream_for(m.device)), "vinet_resize_blur")
    out = out[0] if smap.dim() == 2 else out
    return (out, mm) if return_minmax else out


def blur(img):
    """utils.py:61-64: cv2.GaussianBlur(img, (11, 11), 0) -> FloatTensor (numpy maps are taken to the current device)."""
    if not torch.is_tensor(img):
        img = torch.as_tensor(img, dtype=torch.float32, device="cpu" if L.is_test_double() else "cuda")
    return resize_blur(img, None)


@torch.no_grad()
def to_uint8(maps, minmax=None):
    """utils.py:66-78 img_save(normalize=True) up to the file write: each map is min-max normalised on its own
    ((x - min) / (max - min + 1e-5)), scaled by 255, + 0.5, clamped, rounded half to even -> uint8 [.., H, W]."""
    m = _maps3(maps)
    B, n = m.shape[0], m.shape[1] * m.shape[2]
    lib, stream = L.get(), E._stream_for(m.device)
    if minmax is None:
        minmax = torch.empty((B, 2), dtype=torch.int32, device=m.device)
        L.check(lib.vinet_minmax(m.data_ptr(), B, n, minmax.data_ptr(), stream), "vinet_minmax")
    out = torch.empty(m.shape, dtype=torch.uint8, device=m.device)
    L.check(lib.vinet_normalize_u8(m.data_ptr(), minmax.data_ptr(), B, n, out.data_ptr(), stream), "vinet_normalize_u8")
    return out[0] if maps.dim() == 2 else out


def postprocess(smap, size):
    """generate_result.py:97-100 in one go: resize to `size` = (H, W), blur, normalise -> uint8 map(s) on device."""
    out, mm = resize_blur(smap, size, return_minmax=True)
    return to_uint8(out, mm)


def img_save(tensor, fp, nrow=8, padding=2, normalize=False, range=None, scale_each=False, pad_value=0, format=None):
    """utils.py:66-78 for the harness's use: ONE [H,W] map, normalize=True.  The grid arguments are accepted for
    signature compatibility; a single map has no grid."""
    assert tensor.dim() == 2 and normalize and range is None, "img_save: one [H,W] map with normalize=True (generate_result.py:100)"
    from PIL import Image
    nd = to_uint8(tensor).cpu().numpy()
    im = Image.fromarray(nd)
    if fp.split('.')[-1] == "png":
        im.save(fp, format=format)
    else:
        im.save(fp, format=format, quality=100)
