#!/usr/bin/env python3
"""Selection / audio goldens of SoundDatasetLoader and Hollywood_UCFDataset captured FROM THE REAL REFERENCE CLASSES.

Runs only in the build container (needs /root/reference).  Imports the reference's dataloader.py unmodified, with stand-in
packages for what the image lacks and the item logic never depends on:
  * cv2: `imread(path, 0)` -> the PIL 'L' decode (only `.max() != 0` is taken from it), `resize` -> a nearest resample
    (the resize arithmetic is row f3's own fixture, tests/golden/io_*.npz; what is recorded here is WHICH map is read);
  * torchvision.transforms: Compose / Resize / ToTensor / Normalize -> a tensor of the decoded bytes (the Pillow-exact
    resampler has its own fixture), so `clip_img` spells out which files were decoded, in which order (CRC of the bytes);
  * torchaudio.load -> vinet_amd's PCM reader in sox's 32-bit sample scale (the WAV scale is NOT pinned by this).
SoundDatasetLoader hard-codes its data root (dataloader.py:127); os.path.join is wrapped for the duration of the run so
that root maps onto the synthetic tree (tests/dataset_trees.py).

Recorded per (class, mode): len, list_num_frame, and for seeded items the decoded frame bytes' CRC, the map CRC + dtype
and the audio excerpt (float32, full 70560 window) -> tests/golden/datasets.json + datasets_audio.npz.

Usage:  python tests/golden/make_dataset_goldens.py
"""
import json
import os
import sys
import tempfile
import zlib

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)


def _install_stubs():
    d = tempfile.mkdtemp(prefix="vinet_stubs_")
    bodies = {
        "cv2": "import numpy as np\nfrom PIL import Image\n"
               "def imread(path, flag=1):\n    assert flag == 0\n    return np.asarray(Image.open(path).convert('L'))\n"
               "def resize(a, size):\n    w, h = size\n    ys = (np.arange(h) * a.shape[0]) // h\n    xs = (np.arange(w) * a.shape[1]) // w\n    return a[ys][:, xs]\n",
        "torchaudio": "def load(path, normalization=True):\n    assert normalization is False\n"
                      "    from vinet_amd.generate_result_audio_visual import load_wav\n    return load_wav(path)\n",
        "torchvision": "from . import transforms, utils\n",
    }
    for pkg, body in bodies.items():
        os.makedirs(os.path.join(d, pkg))
        with open(os.path.join(d, pkg, "__init__.py"), "w") as f:
            f.write(body)
    with open(os.path.join(d, "torchvision", "utils.py"), "w") as f:
        f.write("")
    with open(os.path.join(d, "torchvision", "transforms.py"), "w") as f:
        f.write("import numpy as np, torch\n"
                "class Compose:\n    def __init__(self, ts):\n        self.ts = ts\n    def __call__(self, img):\n        return torch.from_numpy(np.asarray(img).copy()).float()\n"
                "class Resize:\n    def __init__(self, size):\n        pass\n"
                "class ToTensor:\n    pass\n"
                "class Normalize:\n    def __init__(self, mean, std):\n        pass\n")
    sys.path[:0] = [d, REF]


def crc(t):
    import numpy as np
    return int(zlib.crc32(np.ascontiguousarray(t).tobytes()))


def main():
    _install_stubs()
    import numpy as np
    import torch
    from tests import dataset_trees as TR
    import dataloader as RD                        # the reference's module

    out, audio = {}, {}
    # ---- SoundDatasetLoader ------------------------------------------------------------------------------------------
    root = tempfile.mkdtemp(prefix="vinet_sound_")
    TR.make_sound_tree(root)
    hard = '/ssd_scratch/cvit/samyak/data/'
    real_join = os.path.join

    def join(a, *rest):
        if isinstance(a, str) and a.startswith(hard):
            a = real_join(root, a[len(hard):])
        return real_join(a, *rest)
    os.path.join = join
    RD.join = join
    try:
        for mode in ("train", "val", "test"):
            ds = RD.SoundDatasetLoader(TR.SOUND_T, dataset_name='DIEM', split=1, mode=mode, use_sound=True)
            rec = dict(len=len(ds), list_indata=list(ds.list_indata),
                       list_num_frame=[list(x) if isinstance(x, tuple) else int(x) for x in ds.list_num_frame],
                       audio_videos=sorted(ds.audiodata.keys()), items=[])
            for idx in range(len(ds)):
                np.random.seed(100 + idx)
                clip, gt, af = ds[idx]
                key = "sound_%s_%d" % (mode, idx)
                audio[key] = af.numpy().reshape(-1).astype(np.float32)
                rec["items"].append(dict(idx=idx, clip_shape=list(clip.shape), clip_crc=crc(clip.numpy().astype(np.uint8)), gt_dtype=str(gt.dtype),
                                         gt_shape=list(gt.shape), gt_crc=(crc(gt) if mode != "train" else None), audio=key,
                                         audio_shape=list(af.shape)))
            out["SoundDatasetLoader/" + mode] = rec
            print("SoundDatasetLoader", mode, "len", len(ds))
        ds = RD.SoundDatasetLoader(TR.SOUND_T, dataset_name='DIEM', mode="val", use_sound=False)
        item = ds[0]
        out["SoundDatasetLoader/val/no_sound"] = dict(len=len(ds), n_out=len(item))
    finally:
        os.path.join = real_join
    # ---- Hollywood_UCFDataset ----------------------------------------------------------------------------------------
    hroot = tempfile.mkdtemp(prefix="vinet_holly_")
    TR.make_hollywood_tree(hroot)
    real_listdir = os.listdir
    os.listdir = lambda p: sorted(real_listdir(p))            # directory order is file-system dependent: fix it for both sides
    try:
        for mode in ("train", "val"):
            for mf in (0, 1):
                ds = RD.Hollywood_UCFDataset(hroot, TR.HOLLY_T, mode=mode, multi_frame=mf)
                rec = dict(len=len(ds), list_num_frame=[list(x) if isinstance(x, tuple) else int(x) for x in ds.list_num_frame], items=[])
                for idx in range(len(ds)):
                    np.random.seed(7 + idx)
                    clip, gt = ds[idx]
                    rec["items"].append(dict(idx=idx, clip_shape=list(clip.shape), clip_crc=crc(clip.numpy().astype(np.uint8)), gt_dtype=str(gt.dtype),
                                             gt_shape=list(gt.shape), gt_crc=(crc(gt.numpy()) if mode == "val" else None)))
                out["Hollywood_UCFDataset/%s/mf%d" % (mode, mf)] = rec
                print("Hollywood_UCFDataset", mode, mf, "len", len(ds))
    finally:
        os.listdir = real_listdir
    with open(os.path.join(HERE, "datasets.json"), "w") as f:
        json.dump(dict(source="reference dataloader.py:124-233,310-391 run on tests/dataset_trees.py (tests/golden/make_dataset_goldens.py)",
                       cases=out), f)
    np.savez_compressed(os.path.join(HERE, "datasets_audio.npz"), **audio)
    print("wrote datasets.json, datasets_audio.npz")


if __name__ == "__main__":
    main()
