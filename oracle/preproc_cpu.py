"""CPU restatement of the reference's input pipeline for frames and ground-truth maps -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and tests/abi_emulator.py may import this module.

    dataloader.py:243-250, generate_result.py:77-88   img_transform = Resize((224, 384)) -> ToTensor() -> Normalize(mean, std)
                                                       on Image.open(path).convert('RGB')
    dataloader.py:283-296                              gt = np.array(Image.open(..).convert('L')).astype('float');
                                                       train: gt = cv2.resize(gt, (384, 224));  if gt.max() > 1: gt /= 255.0

Third-party arithmetic:
  * transforms.Resize -> PIL.Image.resize(size, BILINEAR) (torchvision 0.5.0 functional.resize; Pillow==5.2.0 pinned,
    requirements.txt:106).  Pillow IS in this image (12.2.0): **pinned** -- tests/test_oracle.py compares `pil_resize_bilinear`
    with the real `Image.resize` byte for byte on up- and down-scaling shapes.  Algorithm (libImaging/Resample.c, unchanged
    between those versions for 8-bit images): per axis, support = max(scale, 1), coefficients of the triangle filter over
    [center - support, center + support) normalised in double, converted to 22-bit fixed point, horizontal pass then vertical
    pass, each accumulating integers from 2^21 and clipping (>> 22) to uint8 -- the intermediate image is uint8.
  * ToTensor: float32(byte) / 255;  Normalize: (x - mean) / std in float32 (torchvision absent: restated).
  * cv2.resize on the float64 map: INTER_LINEAR with float32 coefficients and double accumulation (resize.cpp,
    HResizeLinear<double, double, float>) -- cv2 absent: **unpinned**, shares oracle/postproc_cpu.py's coefficient code.

    dataloader.py:65-75    starts / ends: the audio sample range of every video frame (host integer logic)
    dataloader.py:89-122   get_audio_feature: zeros(1, 70560) with float(np.hanning(M)) * wav[:, start:end+1] centred
  * np.hanning: numpy==1.18.5 is pinned (requirements.txt:91): 0.5 - 0.5*cos(2*pi*n/(M-1)) for n = 0..M-1 in double.
    numpy 2.2 here evaluates the algebraically equal 0.5 + 0.5*cos(pi*n'/(M-1)), n' = 1-M, 3-M, ..; the restatement is
    compared with it to float32 round-off (tests/test_oracle.py) -- pinned to that tolerance.
"""
import math

import numpy as np

from . import postproc_cpu as PP

PRECISION_BITS = 32 - 8 - 2
MEAN = (0.485, 0.456, 0.406)
STD = (0.229, 0.224, 0.225)


def resample_coeffs(in_size, out_size):
    """Resample.c precompute_coeffs + normalize_coeffs_8bpc for the bilinear (triangle) filter, box = whole image."""
    scale = filterscale = float(in_size) / float(out_size)
    if filterscale < 1.0:
        filterscale = 1.0
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = [max(0.0, 1.0 - abs((x + xmin - center + 0.5) * ss)) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        for x in range(xmax):
            k = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(0.5 + k * (1 << PRECISION_BITS)) if k >= 0 else int(-0.5 + k * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _pass(img, bounds, kk, axis):
    """one 8-bit pass along `axis` of [..., H, W, C] uint8"""
    src = np.moveaxis(img, axis, -1).astype(np.int64)
    out = np.empty(src.shape[:-1] + (bounds.shape[0],), np.uint8)
    for xx in range(bounds.shape[0]):
        xmin, xmax = int(bounds[xx, 0]), int(bounds[xx, 1])
        acc = (1 << (PRECISION_BITS - 1)) + (src[..., xmin:xmin + xmax] * kk[xx, :xmax].astype(np.int64)).sum(-1)
        out[..., xx] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return np.moveaxis(out, -1, axis)


def pil_resize_bilinear(img, oh, ow):
    """Image.fromarray(img).resize((ow, oh), Image.BILINEAR) for uint8 [..., H, W, C] (or [..., H, W] with C implied 1)."""
    img = np.asarray(img, dtype=np.uint8)
    bh, kh = resample_coeffs(img.shape[-2], ow)
    bv, kv = resample_coeffs(img.shape[-3], oh)
    tmp = _pass(img, bh, kh, -2)              # horizontal first (ImagingResample), uint8 in between
    return _pass(tmp, bv, kv, -3)


def frames_preprocess(frames, oh=224, ow=384, mean=MEAN, std=STD):
    """img_transform on uint8 RGB frames [N, H, W, 3] -> float32 [N, 3, oh, ow]."""
    r = pil_resize_bilinear(frames, oh, ow)
    x = (r.astype(np.float32) / np.float32(255)).astype(np.float32)
    x = (x - np.asarray(mean, np.float32)).astype(np.float32)
    x = (x / np.asarray(std, np.float32)).astype(np.float32)
    return np.ascontiguousarray(np.moveaxis(x, -1, -3))


def gt_preprocess(gt, oh=None, ow=None):
    """dataloader.py:283-296 for uint8 'L' maps [N, H, W]: float64, optional cv2.resize, / 255 if the map's max > 1 -> float32."""
    g = np.asarray(gt, dtype=np.uint8).astype(np.float64)
    H, W = g.shape[-2:]
    if oh is not None and (oh, ow) != (H, W):
        x0, x1, a0, a1 = PP._linear_coeffs(ow, W, True)
        y0, y1, b0, b1 = PP._linear_coeffs(oh, H, False)
        rows = g[..., :, x0] * a0.astype(np.float64) + g[..., :, x1] * a1.astype(np.float64)
        g = rows[..., y0, :] * b0.astype(np.float64)[:, None] + rows[..., y1, :] * b1.astype(np.float64)[:, None]
    out = np.empty(g.shape, np.float32)
    for i in range(g.shape[0]):
        m = g[i]
        out[i] = (m / 255.0 if m.max() > 1.0 else m).astype(np.float32)
    return out


def audio_frame_bounds(n_frames, fps, Fs, n_samples_total):
    """dataloader.py:65-75: starts[f], ends[f] (f = 1..n_frames; index 0 unused = 0) of the audio samples of video frame f."""
    n_samples = Fs / float(fps)
    starts = np.zeros(n_frames + 1, dtype=int)
    ends = np.zeros(n_frames + 1, dtype=int)
    for f in range(1, n_frames + 1):
        starts[f] = int(max(0, ((f - 1) * (1.0 / float(fps)) * Fs) - n_samples / 2))
        ends[f] = int(min(n_samples_total, abs(((f - 1) * (1.0 / float(fps)) * Fs) + n_samples / 2)))
    return starts, ends


def hanning_np118(M):
    """numpy 1.18.5 lib/function_base.py hanning()."""
    if M < 1:
        return np.array([])
    if M == 1:
        return np.ones(1, float)
    n = np.arange(0, M)
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * n / (M - 1))


def audio_excerpt(wav, start, end, win=70560):
    """dataloader.py:89-122 for one video: wav float32 [L] (already * 2**-23) -> float32 [win]."""
    wav = np.asarray(wav, dtype=np.float32)
    seg = wav[start:end + 1]
    M = seg.shape[0]
    out = np.zeros(win, np.float32)
    lo = win // 2 - M // 2
    out[lo:lo + M] = hanning_np118(M).astype(np.float32) * seg
    return out
