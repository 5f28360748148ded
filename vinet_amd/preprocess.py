"""Input pipeline on device -- counterpart of the reference's `img_transform` and ground-truth handling.

    dataloader.py:243-250, generate_result.py:77-88   transforms.Resize((224, 384)) -> ToTensor() -> Normalize(MEAN, STD)
    dataloader.py:283-296                              gt: 'L' bytes -> float -> (train) cv2.resize(gt, (384, 224)) -> / 255

The decoder (PIL) stays on the host; what it produces -- RGB bytes at the image's own size -- is uploaded as is
(0.69 MB for a 640x360 frame instead of 1.03 MB of float32 at 224x384) and resized / scaled / normalised by
libvinet_hip.so (vinet_frames_preprocess, vinet_gt_preprocess), bit-identical to PIL's resampler.
"""
import ctypes as C

import torch

from . import _lib as L
from . import engine as E

MEAN = (0.485, 0.456, 0.406)
STD = (0.229, 0.224, 0.225)
SIZE = (224, 384)


def _ws(nbytes, device):
    return torch.empty((int(nbytes) + 15) // 16 * 16, dtype=torch.uint8, device=device)


@torch.no_grad()
def frames_to_tensor(frames_u8, size=SIZE, mean=MEAN, std=STD):
    """uint8 RGB frames [N,H,W,3] (or one [H,W,3]) on the device -> float32 [N,3,h,w] (or [3,h,w]): img_transform."""
    single = frames_u8.dim() == 3
    f = frames_u8[None] if single else frames_u8
    assert f.dtype == torch.uint8 and f.dim() == 4 and f.shape[-1] == 3, "expected uint8 [N,H,W,3] RGB frames"
    f = f.contiguous()
    N, H, W = f.shape[:3]
    oH, oW = int(size[0]), int(size[1])
    lib = L.get()
    out = torch.empty((N, 3, oH, oW), dtype=torch.float32, device=f.device)
    ws = _ws(lib.vinet_frames_preprocess_ws_bytes(N, H, W, oH, oW), f.device)
    ms = (C.c_float * 6)(*(list(mean) + list(std)))
    L.check(lib.vinet_frames_preprocess(f.data_ptr(), N, H, W, out.data_ptr(), oH, oW, ms, ws.data_ptr(), E._stream_for(f.device)),
            "vinet_frames_preprocess")
    return out[0] if single else out


@torch.no_grad()
def gt_to_tensor(gt_u8, size=None):
    """uint8 'L' maps [N,H,W] (or [H,W]) on the device -> float32 maps: dataloader.py:283-296 (`size` = (h, w) resizes
    like the train mode's cv2.resize(gt, (w, h)); None keeps the resolution, as val mode does)."""
    single = gt_u8.dim() == 2
    g = (gt_u8[None] if single else gt_u8).contiguous()
    assert g.dtype == torch.uint8 and g.dim() == 3, "expected uint8 [N,H,W] maps"
    N, H, W = g.shape
    oH, oW = (H, W) if size is None else (int(size[0]), int(size[1]))
    lib = L.get()
    out = torch.empty((N, oH, oW), dtype=torch.float32, device=g.device)
    ws = _ws(lib.vinet_gt_preprocess_ws_bytes(N, oH, oW), g.device)
    L.check(lib.vinet_gt_preprocess(g.data_ptr(), N, H, W, out.data_ptr(), oH, oW, ws.data_ptr(), E._stream_for(g.device)),
            "vinet_gt_preprocess")
    return out[0] if single else out


def torch_transform(path, device=None):
    """generate_result.py:77-88: (normalised [3,224,384] tensor, PIL size (width, height)) for one image file."""
    import numpy as np
    from PIL import Image
    img = Image.open(path).convert('RGB')
    sz = img.size
    dev = device if device is not None else ("cpu" if L.is_test_double() else "cuda")
    u8 = torch.from_numpy(np.asarray(img).copy()).to(dev)
    return frames_to_tensor(u8), sz


# ---- audio (AViNet): dataloader.py:36-122 ----------------------------------------------------------------------------------
MAX_AUDIO_WIN = int(22050 / 10 * 32)          # dataloader.py:91-94: max_audio_Fs / min_video_fps * 32 = 70560


def audio_frame_bounds(n_frames, fps, Fs, n_samples_total):
    """dataloader.py:65-75: (starts, ends) of the audio samples that belong to video frame f = 1..n_frames."""
    import numpy as np
    n_samples = Fs / float(fps)
    starts = np.zeros(n_frames + 1, dtype=int)
    ends = np.zeros(n_frames + 1, dtype=int)
    for f in range(1, n_frames + 1):
        starts[f] = int(max(0, ((f - 1) * (1.0 / float(fps)) * Fs) - n_samples / 2))
        ends[f] = int(min(n_samples_total, abs(((f - 1) * (1.0 / float(fps)) * Fs) + n_samples / 2)))
    return starts, ends


@torch.no_grad()
def audio_excerpt(wav, start, end, win=MAX_AUDIO_WIN):
    """Hanning-windowed excerpt wav[start:end+1] centred in `win` zeros (dataloader.py:95-121); wav: float32 [L] or [1,L]
    on the device (one upload per video), result float32 [win]."""
    w = wav.reshape(-1)
    assert w.dtype == torch.float32 and w.is_contiguous()
    out = torch.empty(win, dtype=torch.float32, device=w.device)
    L.check(L.get().vinet_audio_excerpt(w.data_ptr(), w.numel(), int(start), int(end), out.data_ptr(), win, E._stream_for(w.device)),
            "vinet_audio_excerpt")
    return out


def get_audio_feature(audioind, audiodata, clip_size, start_idx):
    """dataloader.py:89-122, same arguments and result ([1, 70560, 1]); `audiodata[video]['wav']` is a device tensor."""
    info = audiodata.get(audioind)
    if info is None:
        print(audioind, "not present in data")
        dev = "cpu" if L.is_test_double() else "cuda"
        return torch.zeros(1, MAX_AUDIO_WIN, 1, device=dev)
    start = info['starts'][start_idx + 1]
    if start_idx + clip_size >= len(info['ends']):
        print("Exceeds size", audioind)
        end = info['ends'][-1]
    else:
        end = info['ends'][start_idx + clip_size]
    return audio_excerpt(info['wav'], start, end).view(1, -1, 1)
