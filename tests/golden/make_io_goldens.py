#!/usr/bin/env python3
"""Fixtures for the input / output pipeline rows.  Run from the repo root:

    python tests/golden/make_io_goldens.py

`io_frames.npz`  -- uint8 RGB frames and what the reference's `img_transform` makes of them.  The resize is produced by
                    the REAL Pillow resampler (`Image.resize(BILINEAR)`, what torchvision's `transforms.Resize` calls;
                    dataloader.py:243-250), ToTensor / Normalize by the same float32 expressions in torch.  The script refuses
                    to write unless oracle/preproc_cpu.py reproduces those outputs bit for bit.
`io_maps.npz`    -- float32 saliency maps and the harness's post-processing of them (generate_result.py:95-104) as stated
                    by oracle/postproc_cpu.py.  cv2 / torchvision are absent from this image, so these are REGRESSION
                    vectors of the restatement (its header says "parity unpinned"), not reference outputs; the script
                    cross-checks them against scipy / torch before writing.
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import postproc_cpu as P  # noqa: E402
from oracle import preproc_cpu as Q  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def frames():
    import PIL
    from PIL import Image
    rng = np.random.default_rng(1234)
    cases = {"down_90x160": (2, 90, 160, 56, 96), "up_20x30": (1, 20, 30, 56, 96), "same_w": (1, 70, 96, 56, 96)}
    out = {}
    for name, (n, h, w, oh, ow) in cases.items():
        u8 = rng.integers(0, 256, (n, h, w, 3), dtype=np.uint8)
        pil = np.stack([np.asarray(Image.fromarray(f).resize((ow, oh), Image.BILINEAR)) for f in u8])
        x = torch.from_numpy(pil.copy()).permute(0, 3, 1, 2).float().div(255)
        x = ((x - torch.tensor(Q.MEAN)[None, :, None, None]) / torch.tensor(Q.STD)[None, :, None, None]).numpy()
        assert np.array_equal(Q.pil_resize_bilinear(u8, oh, ow), pil), name
        assert np.array_equal(Q.frames_preprocess(u8, oh, ow), x), name
        out[name + "_in"], out[name + "_resized"], out[name + "_out"] = u8, pil, x
    gt = rng.integers(0, 256, (2, 45, 80), dtype=np.uint8)
    gt[1] = (gt[1] > 230).astype(np.uint8)
    out["gt_in"], out["gt_train"], out["gt_val"] = gt, Q.gt_preprocess(gt, 28, 48), Q.gt_preprocess(gt)
    meta = dict(pillow=PIL.__version__, cases={k: list(v) for k, v in cases.items()}, gt_train_size=[28, 48])
    np.savez_compressed(os.path.join(OUT, "io_frames.npz"), meta=json.dumps(meta), **out)


def maps():
    import scipy.ndimage as ndi
    rng = np.random.default_rng(4321)
    src = (1.0 / (1.0 + np.exp(-(rng.standard_normal((2, 28, 48)) * 2.0 - 1.0)))).astype(np.float32)
    out = {"src": src}
    k = P.gaussian_kernel().astype(np.float64)
    for name, (oh, ow) in {"up_45x80": (45, 80), "same": (28, 48), "down_9x13": (9, 13)}.items():
        r = P.resize_linear(src, oh, ow)
        b = P.gaussian_blur11(r)
        ref = ndi.correlate1d(ndi.correlate1d(r.astype(np.float64), k, axis=2, mode="mirror"), k, axis=1, mode="mirror")
        assert np.abs(b - ref).max() < 1e-6
        t = torch.nn.functional.interpolate(torch.from_numpy(src)[None], size=(oh, ow), mode="bilinear", align_corners=False)[0].numpy()
        assert np.abs(r - t).max() < 1e-4
        out[name + "_blur"], out[name + "_u8"] = b, P.normalize_u8(b)
    np.savez_compressed(os.path.join(OUT, "io_maps.npz"), meta=json.dumps(dict(kernel=[float(v) for v in P.gaussian_kernel()])), **out)


if __name__ == "__main__":
    frames()
    maps()
    print("wrote io_frames.npz, io_maps.npz")
