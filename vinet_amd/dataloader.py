"""Training / validation clips as BYTES -- counterparts of the reference's `DHF1KDataset` (dataloader.py:236-309),
`SoundDatasetLoader` (dataloader.py:124-233) and `Hollywood_UCFDataset` (dataloader.py:310-391).

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


def _map_is_annotated(path):
    """dataloader.py:190-192 `check_frame`: the ground-truth map has at least one non-zero pixel"""
    from PIL import Image
    return int(np.asarray(Image.open(path).convert('L')).max()) != 0


class SoundDatasetLoader(Dataset):
    """The audio-visual sets (DIEM, Coutrot_db1/2, SumMe, ETMD_av, AVAD: `train.py:72-75`, dataloader.py:124-233).

    Directory contract under `path_data` (the reference hard-codes '/ssd_scratch/cvit/samyak/data/'; here it is a keyword
    with that default): `fold_lists/<list>.txt` (first word of a line = video name; DIEM: `DIEM_list_<mode>_fps.txt`,
    others `<name>_list_<mode>_<split>_fps.txt`), `video_frames/<name>/<video>/img_%05d.jpg`,
    `annotations/<name>/<video>/maps/eyeMap_%05d.jpg`, `video_audio/<name>/<video>/<video>.wav`.
    Selection as in the reference: train = one random clip per video whose LAST frame has a non-empty map (redrawn until
    it has); val / test = every 2*len_snippet-th start whose last frame has one.  An item is (clip uint8 [T,h,w,3],
    map uint8 [h,w] of frame start+T) and, with `use_sound` / `use_vox`, the audio excerpt's (video, first sample,
    last sample): `DeviceBatch(..., audiodata=ds.audiodata)` cuts and windows it on the device (vinet_audio_excerpt)."""

    def __init__(self, len_snippet, dataset_name='DIEM', split=1, mode='train', use_sound=False, use_vox=False,
                 path_data='/ssd_scratch/cvit/samyak/data/'):
        from .generate_result_audio_visual import make_dataset
        self.path_data, self.dataset_name, self.mode, self.len_snippet = path_data, dataset_name, mode, len_snippet
        self.use_sound, self.use_vox = use_sound, use_vox
        if dataset_name == 'DIEM':
            file_name = 'DIEM_list_{}_fps.txt'.format(mode)
        else:
            file_name = '{}_list_{}_{}_fps.txt'.format(dataset_name, mode, split)
        with open(os.path.join(path_data, 'fold_lists', file_name), 'r') as f:
            self.list_indata = sorted(line.split(' ')[0].strip() for line in f.readlines())
        print(self.mode, len(self.list_indata))
        T = len_snippet
        if mode == 'train':
            self.list_num_frame = [len(os.listdir(self._maps(v))) for v in self.list_indata]
        else:
            assert mode in ('test', 'val')
            print("val set")
            self.list_num_frame = []
            for v in self.list_indata:
                n = len(os.listdir(self._maps(v)))
                self.list_num_frame += [(v, i) for i in range(0, n - T, 2 * T) if _map_is_annotated(self._map_file(v, i + T))]
        self.max_audio_win = int(22050 / 10 * 32)
        self.audiodata = {}
        if use_sound or use_vox:
            if mode == 'val':
                file_name = file_name.replace('val', 'test')            # dataloader.py:178-179: the val list's audio is indexed by the test list
            self.audiodata = make_dataset(os.path.join(path_data, 'fold_lists', file_name),
                                          os.path.join(path_data, 'video_audio', dataset_name),
                                          os.path.join(path_data, 'annotations', dataset_name))

    def _maps(self, video):
        return os.path.join(self.path_data, 'annotations', self.dataset_name, video, 'maps')

    def _map_file(self, video, frame):
        return os.path.join(self._maps(video), 'eyeMap_%05d.jpg' % frame)

    def check_frame(self, path):
        return _map_is_annotated(path)

    def __len__(self):
        return len(self.list_num_frame)

    def audio_bounds(self, video, start_idx):
        """sample range of the clip that starts at frame `start_idx` (dataloader.py:100-107), None for a video without audio"""
        info = self.audiodata.get(video)
        if info is None:
            print(video, "not present in data")
            return None
        first = int(info['starts'][start_idx + 1])
        if start_idx + self.len_snippet >= len(info['ends']):
            print("Exceeds size", video)
            return first, int(info['ends'][-1])
        return first, int(info['ends'][start_idx + self.len_snippet])

    def __getitem__(self, idx):
        from PIL import Image
        T = self.len_snippet
        if self.mode == 'train':
            video = self.list_indata[idx]
            while True:
                first = np.random.randint(0, self.list_num_frame[idx] - T + 1)
                if _map_is_annotated(self._map_file(video, first + T)):
                    break
                print("No saliency defined in train dataset")
        else:
            video, first = self.list_num_frame[idx]
        frames = os.path.join(self.path_data, 'video_frames', self.dataset_name, video)
        clip = torch.from_numpy(np.stack([np.asarray(Image.open(os.path.join(frames, 'img_%05d.jpg' % (first + i + 1))).convert('RGB'))
                                          for i in range(T)]))
        gt = torch.from_numpy(np.asarray(Image.open(self._map_file(video, first + T)).convert('L')).copy())
        assert int(gt.max()) != 0, (first, video)
        if self.use_sound or self.use_vox:
            # the third field names THIS dataset's audio table: several datasets use the same folder names (clip_1 ...)
            return clip, gt, (video, self.audio_bounds(video, first), self.table_key)
        return clip, gt

    @property
    def table_key(self):
        return (self.dataset_name, self.mode)


class Hollywood_UCFDataset(Dataset):
    """Hollywood-2 / UCF-Sports (dataloader.py:310-391): `<path_data>/<video>/images/*` and `.../maps/*`, frames taken in
    sorted file-name order.  train = one random clip per video, val = every len_snippet-th start (a video not longer than
    a clip contributes its start 0); a video with fewer maps than len_snippet is padded at the FRONT with its first
    frame / map (dataloader.py:356-365).  Items are bytes: (clip uint8 [T,h,w,3], map uint8 [h,w] of the last frame, or
    [T,h,w] with multi_frame)."""

    def __init__(self, path_data, len_snippet, mode="train", frame_no="last", multi_frame=0):
        assert mode in ("train", "val")
        self.path_data, self.len_snippet, self.mode = path_data, len_snippet, mode
        self.frame_no, self.multi_frame = frame_no, multi_frame
        videos = os.listdir(path_data)
        counts = [len(os.listdir(os.path.join(path_data, v, 'images'))) for v in videos]
        if mode == "train":
            self.video_names, self.list_num_frame = videos, counts
            return
        self.list_num_frame = []
        for v, n in zip(videos, counts):
            self.list_num_frame += [(v, i) for i in range(0, n - len_snippet, len_snippet)]
            if n <= len_snippet:
                self.list_num_frame.append((v, 0))

    def __len__(self):
        return len(self.list_num_frame)

    def __getitem__(self, idx):
        from PIL import Image
        T = self.len_snippet
        if self.mode == "train":
            video = self.video_names[idx]
            first = np.random.randint(0, max(1, self.list_num_frame[idx] - T + 1))
        else:
            video, first = self.list_num_frame[idx]
        img_dir, map_dir = os.path.join(self.path_data, video, 'images'), os.path.join(self.path_data, video, 'maps')
        images, maps = sorted(os.listdir(img_dir)), sorted(os.listdir(map_dir))
        if len(maps) < T:
            images = [images[0]] * (T - len(images)) + images
            maps = [maps[0]] * (T - len(maps)) + maps
            assert len(maps) == T and len(images) == T
        clip = torch.from_numpy(np.stack([np.asarray(Image.open(os.path.join(img_dir, images[first + i])).convert('RGB')) for i in range(T)]))
        wanted = range(T) if self.multi_frame != 0 else [T - 1]
        gt = torch.from_numpy(np.stack([np.asarray(Image.open(os.path.join(map_dir, maps[first + i])).convert('L')) for i in wanted]))
        return clip, (gt if self.multi_frame != 0 else gt[0])


def collate_bytes(samples):
    """videos differ in size: keep the per-sample byte tensors in lists (the default collate would try to stack them)"""
    return tuple(list(col) for col in zip(*samples))


class DeviceBatch:
    """bytes -> the tensors the reference's loaders yield, on `device`: clips [B,T,3,224,384] float32 and ground truth
    [B,224,384] (train: cv2.resize(gt, (384, 224)), dataloader.py:291-292) or [B,h,w] at the video's resolution (val).
    A third column of (video, (first sample, last sample)) references becomes the Hanning-windowed audio excerpts
    [B,1,70560,1] (dataloader.py:89-122); `audiodata` is the dataset's table, each waveform is uploaded once.
    `gt_dtype=torch.float64` reproduces SoundDatasetLoader's double ground truth (it skips the FloatTensor cast the other
    two datasets apply: dataloader.py:217-226 vs 296; the values are the float32 results widened)."""

    WAV_CACHE = 64       # waveforms kept on the device (least recently used first out)

    def __init__(self, device, mode="train", audiodata=None, gt_dtype=None, audio_tables=None):
        """`audiodata`: ONE dataset's table {video: info}; `audio_tables`: {SoundDatasetLoader.table_key: table} when the
        batches come from several datasets (train.run concatenates six): an item's third reference field picks its own
        dataset's table, as the reference keeps one table per SoundDatasetLoader (dataloader.py:181-186)."""
        self.device, self.mode, self.audiodata, self.gt_dtype = device, mode, audiodata or {}, gt_dtype
        self.audio_tables = audio_tables or {}
        from collections import OrderedDict
        self._wav = OrderedDict()

    def _audio(self, ref):
        from . import preprocess as PR
        video, bounds = ref[0], ref[1]
        tkey = ref[2] if len(ref) > 2 else None
        if bounds is None:
            return torch.zeros(1, PR.MAX_AUDIO_WIN, 1, device=self.device)
        table = self.audio_tables.get(tkey) if tkey in self.audio_tables else self.audiodata
        ck = (tkey if tkey in self.audio_tables else None, video)
        w = self._wav.get(ck)
        if w is None:
            w = self._wav[ck] = table[video]['wav'][0].contiguous().to(self.device)
        self._wav.move_to_end(ck)
        while len(self._wav) > max(1, self.WAV_CACHE):
            self._wav.popitem(last=False)
        return PR.audio_excerpt(w, bounds[0], bounds[1]).view(1, -1, 1)

    def __call__(self, sample):
        from . import preprocess as PR
        clips, gts = sample[0], sample[1]
        x = torch.stack([PR.frames_to_tensor(c.to(self.device, non_blocking=True)) for c in clips])
        size = PR.SIZE if self.mode == "train" else None
        g = [PR.gt_to_tensor(t.to(self.device, non_blocking=True), size) for t in gts]
        if self.gt_dtype is not None:
            g = [t.to(self.gt_dtype) for t in g]
        g = torch.stack(g) if len(set(tuple(t.shape) for t in g)) == 1 else g
        if len(sample) > 2:
            return x, g, torch.stack([self._audio(r) for r in sample[2]])
        return x, g
