#!/usr/bin/env python3
"""Harness trace golden captured FROM THE REAL REFERENCE LOOP.

Runs only in the build container (needs /root/reference).  Imports the reference's
generate_result.py unmodified (stub packages for cv2 / torchvision / torchaudio / block, which the loop
logic never needs) and runs ITS `validate(args)` (generate_result.py:21-75) on a temporary directory tree
of N frame files per video, with
  * `VideoSaliencyModel` replaced by a recording stub (no network is evaluated),
  * `torch.load` returning an empty state dict,
  * `torch_transform` returning a 1-pixel tensor whose value is the frame's index (so the clip the model
    receives spells out which frames it holds, in which order),
  * `cv2.resize`, `blur`, `img_save` replaced by recorders (the post-processing arithmetic is a different
    row).
What is written: for each case (N frames, clip size T, start_idx / num_parts) the ordered list of
(video name, saved file name, [clip frame indices in model order]) exactly as the reference produced it.

Usage:  python tests/golden/make_trace_golden.py
"""
import json
import os
import sys
import tempfile
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def _install_stubs():
    d = tempfile.mkdtemp(prefix="vinet_stubs_")
    for pkg, body in {"block": "fusions = None\n", "cv2": "", "torchaudio": "",
                      "torchvision": "from . import models, transforms, utils\n"}.items():
        os.makedirs(os.path.join(d, pkg))
        with open(os.path.join(d, pkg, "__init__.py"), "w") as f:
            f.write(body)
    for sub, body in {"models": "vgg19 = None\n", "transforms": "", "utils": ""}.items():
        with open(os.path.join(d, "torchvision", sub + ".py"), "w") as f:
            f.write(body)
    sys.path[:0] = [d, REF]


def main():
    _install_stubs()
    import torch
    argv, sys.argv = sys.argv, ["generate_result.py"]
    import generate_result as G          # the reference's module (its argparse sits under __main__)
    sys.argv = argv

    trace = []

    class StubModel(torch.nn.Module):
        def __init__(self, **kw):
            super().__init__()
            self.kw = kw

        def load_state_dict(self, sd, strict=True):
            return None

        def forward(self, clip):
            # clip [1, 3, T, 1, 1]: channel 0 holds the frame indices in the order the model sees them
            self.last = [int(v) for v in clip[0, 0, :, 0, 0].tolist()]
            return torch.zeros(1, 2, 2)

    box = {}

    def make_model(**kw):
        box["m"] = StubModel(**kw)
        box["ctor"] = kw
        return box["m"]

    def torch_transform(path):
        idx = int(os.path.splitext(os.path.basename(path))[0])
        return torch.full((3, 1, 1), float(idx)), (4, 6)

    G.VideoSaliencyModel = make_model
    G.torch.load = lambda *a, **k: {}
    G.torch_transform = torch_transform
    G.cv2.resize = lambda img, size: img
    G.blur = lambda img: torch.as_tensor(img)
    G.img_save = lambda smap, path, normalize=False: trace.append((os.path.basename(os.path.dirname(path)), os.path.basename(path), list(box["m"].last)))

    cases = []
    for (n_videos, n_frames, T, start_idx, num_parts) in [(1, 11, 4, -1, 4), (1, 7, 4, -1, 4), (1, 6, 4, -1, 4), (3, 9, 3, 2, 3),
                                                           (2, 67, 32, -1, 4), (4, 5, 2, 1, 2)]:
        root = tempfile.mkdtemp(prefix="vinet_trace_")
        for v in range(n_videos):
            os.makedirs(os.path.join(root, "%03d" % (v + 1), "images"))
            for i in range(n_frames + v):
                open(os.path.join(root, "%03d" % (v + 1), "images", "%04d.png" % i), "w").close()
        save = tempfile.mkdtemp(prefix="vinet_trace_out_")
        args = types.SimpleNamespace(path_indata=root, file_weight="unused.pt", clip_size=T, transformer_in_channel=32, nhead=4,
                                     decoder_upsample=1, num_hier=3, save_path=save, start_idx=start_idx, num_parts=num_parts)
        del trace[:]
        G.validate(args)
        cases.append(dict(n_videos=n_videos, n_frames=n_frames, T=T, start_idx=start_idx, num_parts=num_parts,
                          ctor={k: (bool(v) if isinstance(v, bool) else v) for k, v in box["ctor"].items()},
                          calls=[[d, f, c] for d, f, c in trace]))
        print("case", (n_videos, n_frames, T, start_idx, num_parts), "->", len(trace), "model calls")
    with open(os.path.join(HERE, "harness_trace.json"), "w") as f:
        json.dump(dict(source="reference generate_result.py:21-75 run with a recording stub model (tests/golden/make_trace_golden.py)",
                       cases=cases), f)
    print("wrote harness_trace.json")


if __name__ == "__main__":
    main()
