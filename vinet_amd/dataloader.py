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
    def __init__(self, path_data, len_snippet, mode="train", multi_frame=0, alternate=1):
        ''' mode: train, val, save '''
        self.path_data = path_data
        self.len_snippet = len_snippet
        self.mode = mode
        self.multi_frame = multi_frame
        self.alternate = alternate
        n_img = lambda v: len(os.listdir(os.path.join(path_data, v, 'images')))
        if self.mode == "train":                                          # dataloader.py:251-253
            self.video_names = os.listdir(path_data)
            self.list_num_frame = [n_img(d) for d in self.video_names]
        elif self.mode == "val":                                          # dataloader.py:254-258
            self.list_num_frame = []
            for v in os.listdir(path_data):
                for i in range(0, n_img(v) - self.alternate * self.len_snippet, 4 * self.len_snippet):
                    self.list_num_frame.append((v, i))
        else:                                                             # dataloader.py:259-264
            self.list_num_frame = []
            for v in os.listdir(path_data):
                for i in range(0, n_img(v) - self.alternate * self.len_snippet, self.len_snippet):
                    self.list_num_frame.append((v, i))
                self.list_num_frame.append((v, n_img(v) - self.len_snippet))

    def __len__(self):
        return len(self.list_num_frame)

    def __getitem__(self, idx):
        from PIL import Image
        if self.mode == "train":                                          # dataloader.py:270-272
            file_name = self.video_names[idx]
            start_idx = np.random.randint(0, self.list_num_frame[idx] - self.alternate * self.len_snippet + 1)
        else:
            (file_name, start_idx) = self.list_num_frame[idx]
        path_clip = os.path.join(self.path_data, file_name, 'images')
        path_annt = os.path.join(self.path_data, file_name, 'maps')
        frames, gts, sz = [], [], None
        for i in range(self.len_snippet):
            name = '%04d.png' % (start_idx + self.alternate * i + 1)
            img = Image.open(os.path.join(path_clip, name)).convert('RGB')
            sz = img.size
            frames.append(np.asarray(img))
            if self.mode != "save" and (self.multi_frame != 0 or i == self.len_snippet - 1):
                gts.append(np.asarray(Image.open(os.path.join(path_annt, name)).convert('L')))
        clip = torch.from_numpy(np.stack(frames))                         # [T,h,w,3] uint8
        if self.mode == "save":
            return clip, start_idx, file_name, sz
        gt = torch.from_numpy(np.stack(gts))                              # [1 or T,h,w] uint8
        return clip, (gt[-1] if self.multi_frame == 0 else gt)


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
