"""CPU restatement of the saliency-map post-processing of the reference -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and tests/abi_emulator.py may import this module; the product path
(vinet_amd/) never does.

What the reference does with a predicted map (float32 [H, W], sigmoid output of the decoder):

    generate_result.py:95-104  process():  smap = cv2.resize(smap, (img_w, img_h)); smap = blur(smap);
                                           img_save(smap, path, normalize=True)
    train.py:251-253           validate(): pred = cv2.resize(pred, (gt_w, gt_h)); pred = blur(pred)  -> losses
    utils.py:61-64             blur():     cv2.GaussianBlur(img, (11, 11), 0)
    utils.py:66-78             img_save(): torchvision.utils.make_grid(normalize=True) -> mul(255).add_(0.5).clamp_(0, 255)
                                           -> torch.round -> uint8, channel 0

The arithmetic lives in two third-party packages that are ABSENT from this image and from /root/reference:
opencv-python==3.4.3.18 (requirements.txt:96) and torchvision==0.5.0 (requirements.txt:178).  **Parity unpinned**:
there is no cv2 / torchvision here to produce golden vectors, and the reference holds no fixture for this path.  The
functions below restate the published algorithms of those versions; tests/test_oracle.py cross-checks them against
two independent implementations that ARE here -- torch.nn.functional.interpolate(mode="bilinear",
align_corners=False) (same half-pixel sampling as INTER_LINEAR without anti-aliasing) and
scipy.ndimage.correlate1d(mode="mirror") (= BORDER_REFLECT_101).

  * cv2.resize, INTER_LINEAR, CV_32F (imgproc/src/resize.cpp, resizeGeneric_ + HResizeLinear / VResizeLinear):
        scale = 1 / (dsize / ssize)  (double);  f = (float)((d + 0.5) * scale - 0.5);  s = floor(f);  f -= s
        horizontally:  s < 0 -> (s, f) = (0, 0);  s >= ssize - 1 -> (s, f) = (ssize - 1, 0);  weights (1 - f, f)
        vertically:    rows s, s + 1 clipped to [0, ssize - 1], weights (1 - f, f) unchanged
        row pass first (S[s] * a0 + S[s + 1] * a1), then the column pass (R0 * b0 + R1 * b1), all in float32
  * cv2.GaussianBlur(., (11, 11), 0), CV_32F (imgproc/src/smooth.cpp): sigma = 0.3 * ((11 - 1) * 0.5 - 1) + 0.8 = 2.0;
        getGaussianKernel: t_i = (float)exp(-0.5 / sigma^2 * (i - 5)^2) in double, k_i = (float)(t_i * (1 / sum t))
        separable, BORDER_REFLECT_101; row pass = generic RowFilter (s = k0*x0; s += k_j*x_j, j = 1..10), column
        pass = SymmColumnFilter (s = k5*x0 + sum_{j=1..5} k_{5+j} * (x_{+j} + x_{-j})), float32
  * make_grid(normalize=True) on one map (torchvision/utils.py, 0.5.0): clamp to [min, max],
        x = (x + (-min)) / (max - min + 1e-5)   (the divisor is formed in double, the tensor ops run in float32)
  * img_save: y = clamp(x * 255 + 0.5, 0, 255);  round half to even;  uint8
"""
import numpy as np

KSIZE = 11
SIGMA = 0.3 * ((KSIZE - 1) * 0.5 - 1) + 0.8          # 2.0


def gaussian_kernel():
    """cv2.getGaussianKernel(11, -1, CV_32F)."""
    scale2x = -0.5 / (SIGMA * SIGMA)
    x = np.arange(KSIZE, dtype=np.float64) - (KSIZE - 1) * 0.5
    t = np.exp(scale2x * x * x).astype(np.float32)
    inv = 1.0 / float(np.sum(t.astype(np.float64)))
    return (t.astype(np.float64) * inv).astype(np.float32)


def _linear_coeffs(dsize, ssize, zero_at_border):
    scale = 1.0 / (float(dsize) / float(ssize))
    d = np.arange(dsize, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    if zero_at_border:
        lo, hi = s < 0, s >= ssize - 1
        f = np.where(lo | hi, np.float32(0), f).astype(np.float32)
        s = np.where(lo, 0, np.where(hi, ssize - 1, s))
        s1 = np.minimum(s + 1, ssize - 1)            # weight 0 there; never read past the row
    else:
        s1 = np.clip(s + 1, 0, ssize - 1)
        s = np.clip(s, 0, ssize - 1)
    return s, s1, (np.float32(1) - f).astype(np.float32), f


def resize_linear(src, oh, ow):
    """cv2.resize(src, (ow, oh)) for float32 [..., H, W] maps (INTER_LINEAR)."""
    src = np.asarray(src, dtype=np.float32)
    H, W = src.shape[-2:]
    x0, x1, a0, a1 = _linear_coeffs(ow, W, True)
    y0, y1, b0, b1 = _linear_coeffs(oh, H, False)
    rows = (src[..., :, x0] * a0).astype(np.float32) + (src[..., :, x1] * a1).astype(np.float32)
    rows = rows.astype(np.float32)
    out = (rows[..., y0, :] * b0[:, None]).astype(np.float32) + (rows[..., y1, :] * b1[:, None]).astype(np.float32)
    return out.astype(np.float32)


def reflect101(i, n):
    """cv::borderInterpolate(i, n, BORDER_REFLECT_101)."""
    i = np.asarray(i, dtype=np.int64).copy()
    if n == 1:
        return np.zeros_like(i)
    while True:
        lo, hi = i < 0, i >= n
        if not (lo.any() or hi.any()):
            return i
        i = np.where(lo, -i, i)
        i = np.where(hi, 2 * (n - 1) - i, i)


def gaussian_blur11(img):
    """cv2.GaussianBlur(img, (11, 11), 0) for float32 [..., H, W] maps."""
    img = np.asarray(img, dtype=np.float32)
    H, W = img.shape[-2:]
    k = gaussian_kernel()
    r = KSIZE // 2
    xi = [reflect101(np.arange(W) + j - r, W) for j in range(KSIZE)]
    rows = (img[..., :, xi[0]] * k[0]).astype(np.float32)
    for j in range(1, KSIZE):
        rows = (rows + (img[..., :, xi[j]] * k[j]).astype(np.float32)).astype(np.float32)
    yi = [reflect101(np.arange(H) + j - r, H) for j in range(KSIZE)]
    out = (rows[..., yi[r], :] * k[r]).astype(np.float32)
    for j in range(1, r + 1):
        pair = (rows[..., yi[r + j], :] + rows[..., yi[r - j], :]).astype(np.float32)
        out = (out + (pair * k[r + j]).astype(np.float32)).astype(np.float32)
    return out


def resize_blur(smap, oh, ow):
    """generate_result.py:97-98 / train.py:251-252."""
    return gaussian_blur11(resize_linear(smap, oh, ow))


def normalize_u8(img):
    """utils.py:66-78 img_save(normalize=True) of ONE [H, W] map (or a batch [B, H, W] of separately saved maps)."""
    img = np.asarray(img, dtype=np.float32)
    if img.ndim == 3:
        return np.stack([normalize_u8(m) for m in img])
    mn, mx = float(img.min()), float(img.max())
    x = np.clip(img, np.float32(mn), np.float32(mx)).astype(np.float32)
    x = (x + np.float32(-mn)).astype(np.float32)
    x = (x / np.float32(mx - mn + 1e-5)).astype(np.float32)
    y = ((x * np.float32(255)).astype(np.float32) + np.float32(0.5)).astype(np.float32)
    y = np.clip(y, np.float32(0), np.float32(255))
    return np.rint(y).astype(np.uint8)
