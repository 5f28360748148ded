"""DHF1K clips as BYTES -- counterpart of the reference's `DHF1KDataset` (dataloader.py:236-309).

Same constructor, same length, same clip / frame selection (train: one random clip per video; val: every
4*len_snippet-th start; save: every len_snippet-th start plus the tail) -- but `__getitem__` stops after the decoder:
it returns the clip's RGB frames and the ground-truth map as uint8 arrays at the video's own resolution.  Resize,
ToTensor, Normalize and the ground truth's cv2.resize / 255 run on the device for the whole batch
(`DeviceBatch`: vinet_amd.preprocess -> libvinet_hip.so), so a worker process only decodes PNGs and a 640x360 clip of
32 frames crosses PCIe as 22 MB of bytes instead of 33 MB of float32 at 224x384.
"""
import os

import numpy as np
import torch
from torch.utils.data import Dataset


class DHF1KDataset(Dataset):
    """`list_num_frame` keeps the reference's meaning: frame counts per video in train mode (one random clip per video
    and epoch, dataloader.py:251-253,270-272), (video, first frame) pairs otherwise -- every 4*T-th start for "val"
    (dataloader.py:254-258), every T-th start plus the clip that ends with the video for "save" (dataloader.py:259-264)."""

    def __init__(self, path_data, len_snippet, mode="train", multi_frame=0, alternate=1):
        assert mode in ("train", "val", "save")
        self.path_data, self.len_snippet, self.mode = path_data, len_snippet, mode
        self.multi_frame, self.alternate = multi_frame, alternate
        videos = os.listdir(path_data)
        counts = {v: len(os.listdir(os.path.join(path_data, v, 'images'))) for v in videos}
        span = alternate * len_snippet                       # frames a clip covers
        if mode == "train":
            self.video_names = videos
            self.list_num_frame = [counts[v] for v in videos]
            return
        hop = 4 * len_snippet if mode == "val" else len_snippet
        self.list_num_frame = []
        for v in videos:
            self.list_num_frame += [(v, first) for first in range(0, counts[v] - span, hop)]
            if mode == "save":
                self.list_num_frame.append((v, counts[v] - len_snippet))

    def __len__(self):
        return len(self.list_num_frame)

    def __getitem__(self, idx):
        from PIL import Image
        T, step = self.len_snippet, self.alternate
        if self.mode == "train":
            video = self.video_names[idx]
            first = np.random.randint(0, self.list_num_frame[idx] - step * T + 1)       # dataloader.py:272
        else:
            video, first = self.list_num_frame[idx]
        names = ['%04d.png' % (first + step * i + 1) for i in range(T)]                   # 1-based file names
        img_dir, map_dir = os.path.join(self.path_data, video, 'images'), os.path.join(self.path_data, video, 'maps')
        imgs = [Image.open(os.path.join(img_dir, n)).convert('RGB') for n in names]
        clip = torch.from_numpy(np.stack([np.asarray(im) for im in imgs]))                # [T,h,w,3] uint8
        if self.mode == "save":
            return clip, first, video, imgs[-1].size
        wanted = names if self.multi_frame != 0 else names[-1:]                           # dataloader.py:306-308: last frame's map
        gt = torch.from_numpy(np.stack([np.asarray(Image.open(os.path.join(map_dir, n)).convert('L')) for n in wanted]))
        return clip, (gt if self.multi_frame != 0 else gt[0])


def collate_bytes(samples):
    """videos differ in size: keep the per-sample byte tensors in lists (the default collate would try to stack them)"""
    return tuple(list(col) for col in zip(*samples))


class DeviceBatch:
    """bytes -> the tensors the reference's loader yields, on `device`: clips [B,T,3,224,384] float32 and ground truth
    [B,224,384] (train: cv2.resize(gt, (384, 224)), dataloader.py:291-292) or [B,h,w] at the video's resolution (val)."""

    def __init__(self, device, mode="train"):
        self.device, self.mode = device, mode

    def __call__(self, sample):
        from . import preprocess as PR
        clips, gts = sample[0], sample[1]
        x = torch.stack([PR.frames_to_tensor(c.to(self.device, non_blocking=True)) for c in clips])
        size = PR.SIZE if self.mode == "train" else None
        g = [PR.gt_to_tensor(t.to(self.device, non_blocking=True), size) for t in gts]
        g = torch.stack(g) if len(set(tuple(t.shape) for t in g)) == 1 else g
        return x, g
