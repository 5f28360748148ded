"""Deterministic synthetic directory trees for the dataset classes (shared by tests/golden/make_dataset_goldens.py,
which runs the REFERENCE's classes on them, and by tests/test_drivers.py, which runs vinet_amd's on the same trees).
Nothing here is reference code: JPEG / PNG / WAV files with seeded random content in the layout the reference reads."""
import os
import wave

import numpy as np

SOUND_VIDEOS = [("clipA", 41, 25.0), ("clipB", 30, 30.0), ("clipC", 37, 25.0), ("quiet", 28, 25.0)]   # (name, frames, fps)
SOUND_T = 8
FS = 22050


def make_sound_tree(root, dataset_name="DIEM", seed=0):
    """fold_lists + video_frames + annotations + video_audio for SoundDatasetLoader (dataloader.py:124-233).
    Maps 12 and 24 of every video are empty (check_frame must skip the clips that end there); 'quiet' has no .wav."""
    from PIL import Image
    rng = np.random.default_rng(seed)
    os.makedirs(os.path.join(root, "fold_lists"), exist_ok=True)
    lists = {"train": SOUND_VIDEOS[:3] + SOUND_VIDEOS[3:], "test": SOUND_VIDEOS[1:], "val": SOUND_VIDEOS[1:]}
    for mode, vids in lists.items():
        with open(os.path.join(root, "fold_lists", "%s_list_%s_fps.txt" % (dataset_name, mode) if dataset_name == "DIEM"
                               else "%s_list_%s_1_fps.txt" % (dataset_name, mode)), "w") as f:
            for name, n, fps in reversed(vids):                       # unsorted on purpose: the loader sorts
                f.write("%s %d %s\n" % (name, n, fps))
    for vi, (name, n, fps) in enumerate(SOUND_VIDEOS):
        fdir = os.path.join(root, "video_frames", dataset_name, name)
        mdir = os.path.join(root, "annotations", dataset_name, name, "maps")
        os.makedirs(fdir, exist_ok=True)
        os.makedirs(mdir, exist_ok=True)
        h, w = 24 + 2 * vi, 32
        for i in range(1, n + 1):
            Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)).save(os.path.join(fdir, "img_%05d.jpg" % i), quality=95)
            m = np.zeros((h, w), np.uint8) if i in (12, 24) else rng.integers(0, 256, (h, w), dtype=np.uint8)
            Image.fromarray(m).save(os.path.join(mdir, "eyeMap_%05d.jpg" % i), quality=95)
        if name == "quiet":
            continue
        adir = os.path.join(root, "video_audio", dataset_name, name)
        os.makedirs(adir, exist_ok=True)
        nsamp = int(FS * n / fps) - 137 * vi
        pcm = (rng.standard_normal(nsamp) * 3000).clip(-32768, 32767).astype("<i2")
        with wave.open(os.path.join(adir, name + ".wav"), "wb") as wv:
            wv.setnchannels(1)
            wv.setsampwidth(2)
            wv.setframerate(FS)
            wv.writeframes(pcm.tobytes())


HOLLY_VIDEOS = [("actionclip001", 19), ("actionclip002", 5), ("sports07", 8), ("sports11", 13)]
HOLLY_T = 8


def make_hollywood_tree(root, seed=1):
    """<video>/images/*.png + maps/*.png for Hollywood_UCFDataset (dataloader.py:310-391); 'actionclip002' is shorter than
    a clip (front padding with its first frame), 'sports07' is exactly one clip long."""
    from PIL import Image
    rng = np.random.default_rng(seed)
    for vi, (name, n) in enumerate(HOLLY_VIDEOS):
        os.makedirs(os.path.join(root, name, "images"), exist_ok=True)
        os.makedirs(os.path.join(root, name, "maps"), exist_ok=True)
        h, w = 20 + vi, 28
        for i in range(n):
            fn = "%s_%05d.png" % (name, i * 3 + 1)                    # sorted order is the temporal order; numbering has gaps
            Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)).save(os.path.join(root, name, "images", fn))
            lo = 2 if i % 4 == 0 else 256                              # some maps are {0,1}-valued: no /255 for them
            Image.fromarray(rng.integers(0, lo, (h, w), dtype=np.uint8)).save(os.path.join(root, name, "maps", fn))
