"""train.py / generate_result.py counterparts: flags, schedule and the S3D key remap (CPU)."""
import torch

from vinet_amd import generate_result as GR
from vinet_amd import train as TR


def test_train_flags_match_reference_defaults():
    a = TR.build_parser().parse_args([])
    assert (a.no_epochs, a.lr, a.batch_size, a.clip_size, a.log_interval, a.no_workers) == (40, 1e-4, 8, 32, 5, 4)
    assert a.kldiv is True and a.cc is False and a.sim is False and a.l1 is False
    assert (a.kldiv_coeff, a.cc_coeff, a.sim_coeff) == (1.0, -1.0, -1.0)
    assert a.decoder_upsample == 1 and a.num_hier == 3 and a.load_weight == "None" and a.dataset == "DHF1KDataset"
    assert TR.build_parser().parse_args(["--cc", "True"]).cc is True


def test_sliding_window_schedule_matches_generate_result():
    """generate_result.py:58-73 for T=4, N=9: frames 3..8 from forward clips, frames 0..2 from flipped ones"""
    T, N = 4, 9
    sched = GR.sliding_window_schedule(N, T)
    fwd = [(o, c) for o, c, f in sched if not f]
    rev = [(o, c) for o, c, f in sched if f]
    assert fwd == [(i, list(range(i - T + 1, i + 1))) for i in range(T - 1, N)]
    assert rev == [(i - T + 1, list(range(i, i - T, -1))) for i in range(T - 1, 2 * T - 2)]
    assert sorted(o for o, _, _ in sched) == list(range(N))           # every frame predicted exactly once
    # call order: the flipped call directly follows its forward call (generate_result.py:67-71)
    assert [f for _, _, f in sched[:6]] == [False, True, False, True, False, True]
    assert GR.sliding_window_schedule(2 * T - 2, T) == []               # "more frames are needed"


def test_schedule_equals_trace_captured_from_the_reference_loop():
    """tests/golden/harness_trace.json holds the ordered (video, saved frame, clip frame
    indices) list the REFERENCE's generate_result.validate produced with a recording stub model
    (tests/golden/make_trace_golden.py); our schedule + video selection must reproduce it call for call."""
    import json
    import os
    from tests.goldens import GOLDEN_DIR
    cases = json.load(open(os.path.join(GOLDEN_DIR, "harness_trace.json")))["cases"]
    assert len(cases) >= 6 and any(c["T"] == 32 for c in cases)
    for c in cases:
        T = c["T"]
        assert c["ctor"]["num_clips"] == T and c["ctor"]["use_upsample"] is True and c["ctor"]["num_hier"] == 3
        names = ["%03d" % (v + 1) for v in range(c["n_videos"])]
        a, b = 0, len(names)
        if c["start_idx"] != -1:                      # generate_result.py:44-46
            ln = (1.0 / float(c["num_parts"])) * len(names)
            a, b = int((c["start_idx"] - 1) * ln), int(c["start_idx"] * ln)
        want = []
        for v, name in enumerate(names):
            if not (a <= v < b):
                continue
            for out_frame, clip, flipped in GR.sliding_window_schedule(c["n_frames"] + v, T):
                want.append([name, "%04d.png" % out_frame, clip])
        assert want == c["calls"], (c["n_videos"], c["n_frames"], T)


def test_list_videos_splits_like_the_reference(tmp_path):
    """generate_result.py:40-46: sorted directories, part `start_idx` of `num_parts` (1-based), -1 = all"""
    for n in ("b", "a", "d", "c", "e"):
        (tmp_path / n).mkdir()
    (tmp_path / "file.txt").write_text("x")
    assert GR.list_videos(str(tmp_path), -1, 4) == ["a", "b", "c", "d", "e"]
    for k in (1, 2, 3):
        ln = 5 / 3.0
        assert GR.list_videos(str(tmp_path), k, 3) == ["a", "b", "c", "d", "e"][int((k - 1) * ln):int(k * ln)]


def test_s3d_kinetics_key_remap():
    from vinet_amd import model
    bb = model.BackBoneS3D()
    src = {}
    want = {"base.0.conv_s.weight": "base1.0.conv_s.weight", "base.3.conv_t.weight": "base1.3.conv_t.weight",
            "base.5.branch0.0.conv.weight": "base2.0.branch0.0.conv.weight",
            "module.base.8.branch1.1.conv_s.weight": "base3.0.branch1.1.conv_s.weight",
            "base.15.branch3.1.bn.running_mean": "base4.1.branch3.1.bn.running_mean"}
    sd = bb.state_dict()
    for i, (k, dst) in enumerate(want.items()):
        src[k] = torch.full_like(sd[dst], float(i + 1))
    TR.remap_s3d_kinetics(src, bb)
    sd = bb.state_dict()
    for i, dst in enumerate(want.values()):
        assert torch.equal(sd[dst], torch.full_like(sd[dst], float(i + 1)))


def _fake_dhf1k(root, n_videos=3, n_frames=9, h=20, w=30, seed=0):
    import os
    import numpy as np
    from PIL import Image
    rng = np.random.default_rng(seed)
    data = {}
    for v in range(n_videos):
        name = "%03d" % (v + 1)
        os.makedirs(root / name / "images")
        os.makedirs(root / name / "maps")
        fr = rng.integers(0, 256, (n_frames + v, h + 2 * v, w, 3), dtype=np.uint8)      # videos differ in length and size
        gt = rng.integers(0, 256, (n_frames + v, h + 2 * v, w), dtype=np.uint8)
        for i in range(fr.shape[0]):
            Image.fromarray(fr[i]).save(root / name / "images" / ("%04d.png" % (i + 1)))
            Image.fromarray(gt[i]).save(root / name / "maps" / ("%04d.png" % (i + 1)))
        data[name] = (fr, gt)
    return data


def test_dhf1k_dataset_yields_bytes_with_the_reference_selection(tmp_path):
    """dataloader.py:236-309: lengths, clip starts and frame names per mode; items are the decoder's bytes"""
    import numpy as np
    from vinet_amd import dataloader as DL
    data = _fake_dhf1k(tmp_path)
    T = 3
    tr = DL.DHF1KDataset(str(tmp_path), T, mode="train")
    assert len(tr) == 3
    np.random.seed(5)
    clip, gt = tr[1]
    np.random.seed(5)
    name = tr.video_names[1]
    start = np.random.randint(0, tr.list_num_frame[1] - T + 1)                  # dataloader.py:272
    fr, g = data[name]
    assert clip.dtype == torch.uint8 and np.array_equal(clip.numpy(), fr[start:start + T]) and np.array_equal(gt.numpy(), g[start + T - 1])
    va = DL.DHF1KDataset(str(tmp_path), T, mode="val")
    want = sorted((v, i) for v in data for i in range(0, data[v][0].shape[0] - T, 4 * T))
    assert sorted(va.list_num_frame) == want and len(va) == len(want)
    sv = DL.DHF1KDataset(str(tmp_path), T, mode="save")
    want = sorted([(v, i) for v in data for i in range(0, data[v][0].shape[0] - T, T)] + [(v, data[v][0].shape[0] - T) for v in data])
    assert sorted(sv.list_num_frame) == want
    clip, s0, fname, sz = sv[0]
    assert clip.shape[0] == T and sz == (data[fname][0].shape[2], data[fname][0].shape[1]) and np.array_equal(clip.numpy(), data[fname][0][s0:s0 + T])
    alt = DL.DHF1KDataset(str(tmp_path), T, mode="val", alternate=2)
    v, i = alt.list_num_frame[0]
    assert np.array_equal(alt[0][0].numpy(), data[v][0][i:i + 2 * T:2])
    mf = DL.DHF1KDataset(str(tmp_path), T, mode="val", multi_frame=1)
    assert mf[0][1].shape[0] == T


def test_device_batch_equals_the_reference_transforms(tmp_path):
    """bytes -> network inputs through the C ABI (emulator here) == img_transform / gt handling restated by the oracle"""
    import numpy as np
    from oracle import preproc_cpu as Q
    from tests.abi_emulator import AbiEmulator
    from vinet_amd import _lib as L
    from vinet_amd import dataloader as DL
    data = _fake_dhf1k(tmp_path)
    L._install_test_double(AbiEmulator())
    try:
        ds = DL.DHF1KDataset(str(tmp_path), 3, mode="train")
        np.random.seed(1)
        loader = torch.utils.data.DataLoader(ds, batch_size=3, shuffle=False, collate_fn=DL.collate_bytes)
        raw = next(iter(loader))
        x, g = DL.DeviceBatch(torch.device("cpu"), "train")(raw)
        assert x.shape == (3, 3, 3, 224, 384) and g.shape == (3, 224, 384)
        for b in range(3):
            assert np.array_equal(x[b].numpy(), Q.frames_preprocess(raw[0][b].numpy()))
            assert np.array_equal(g[b].numpy(), Q.gt_preprocess(raw[1][b].numpy()[None], 224, 384)[0])
        va = DL.DHF1KDataset(str(tmp_path), 3, mode="val")
        raw = next(iter(torch.utils.data.DataLoader(va, batch_size=1, collate_fn=DL.collate_bytes)))
        x, g = DL.DeviceBatch(torch.device("cpu"), "val")(raw)
        assert g.shape == (1,) + tuple(raw[1][0].shape) and np.array_equal(g[0].numpy(), Q.gt_preprocess(raw[1][0].numpy()[None])[0])
    finally:
        L._install_test_double(None)


def _crc(a):
    import zlib
    import numpy as np
    return int(zlib.crc32(np.ascontiguousarray(a).tobytes()))


def _dataset_goldens():
    import json
    import os
    import numpy as np
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    with open(os.path.join(here, "datasets.json")) as f:
        return json.load(f)["cases"], np.load(os.path.join(here, "datasets_audio.npz"))


def test_sound_dataset_loader_against_the_reference_selection(tmp_path):
    """dataloader.py:124-233 on the synthetic DIEM tree: lengths, sorted fold list, annotated-clip selection, which frame /
    map files an item decodes (seeded train draws included), the float64 map values and the windowed audio excerpt --
    all against what the reference's own class produced on the same tree (tests/golden/make_dataset_goldens.py)"""
    import numpy as np
    from tests import dataset_trees as TR
    from tests.abi_emulator import AbiEmulator
    from vinet_amd import _lib as L
    from vinet_amd import dataloader as DL
    cases, audio = _dataset_goldens()
    TR.make_sound_tree(str(tmp_path))
    L._install_test_double(AbiEmulator())
    try:
        for mode in ("train", "val", "test"):
            want = cases["SoundDatasetLoader/" + mode]
            ds = DL.SoundDatasetLoader(TR.SOUND_T, dataset_name='DIEM', split=1, mode=mode, use_sound=True, path_data=str(tmp_path))
            assert len(ds) == want["len"] and list(ds.list_indata) == want["list_indata"]
            assert [list(x) if isinstance(x, tuple) else int(x) for x in ds.list_num_frame] == want["list_num_frame"]
            assert sorted(ds.audiodata.keys()) == want["audio_videos"] and ds.max_audio_win == 70560
            batch = DL.DeviceBatch(torch.device("cpu"), "val", audiodata=ds.audiodata, gt_dtype=torch.float64)
            for it in want["items"]:
                np.random.seed(100 + it["idx"])
                clip, gt, ref = ds[it["idx"]]
                assert clip.dtype == torch.uint8 and list(clip.shape) == it["clip_shape"] and _crc(clip.numpy()) == it["clip_crc"]
                g = gt.numpy().astype('float')
                g = g / 255.0 if g.max() > 1.0 else g
                assert it["gt_dtype"] == "float64"                         # F11: this loader's maps reach the loss as doubles
                if it["gt_crc"] is not None:
                    assert list(g.shape) == it["gt_shape"] and _crc(g) == it["gt_crc"]
                x, gd, af = batch(([clip], [gt], [ref]))
                assert gd.dtype == torch.float64 and np.allclose(gd[0].numpy(), g, rtol=0, atol=6e-8)
                assert list(af.shape[1:]) == it["audio_shape"] and np.array_equal(af.view(-1).numpy(), audio[it["audio"]])
        ns = DL.SoundDatasetLoader(TR.SOUND_T, dataset_name='DIEM', mode="val", path_data=str(tmp_path))
        assert len(ns) == cases["SoundDatasetLoader/val/no_sound"]["len"] and len(ns[0]) == cases["SoundDatasetLoader/val/no_sound"]["n_out"]
    finally:
        L._install_test_double(None)


def test_hollywood_ucf_dataset_against_the_reference_selection(tmp_path, monkeypatch):
    """dataloader.py:310-391: per-mode lengths and starts, front padding of a video shorter than a clip, last-frame vs
    multi-frame maps with the per-map /255 rule -- against the reference class's output on the same tree"""
    import os
    import numpy as np
    from tests import dataset_trees as TR
    from tests.abi_emulator import AbiEmulator
    from vinet_amd import _lib as L
    from vinet_amd import dataloader as DL
    cases, _ = _dataset_goldens()
    TR.make_hollywood_tree(str(tmp_path))
    real_listdir = os.listdir
    monkeypatch.setattr(os, "listdir", lambda p: sorted(real_listdir(p)))   # the golden was captured with sorted directory order
    L._install_test_double(AbiEmulator())
    try:
        for mode in ("train", "val"):
            for mf in (0, 1):
                want = cases["Hollywood_UCFDataset/%s/mf%d" % (mode, mf)]
                ds = DL.Hollywood_UCFDataset(str(tmp_path), TR.HOLLY_T, mode=mode, multi_frame=mf)
                assert len(ds) == want["len"]
                assert [list(x) if isinstance(x, tuple) else int(x) for x in ds.list_num_frame] == want["list_num_frame"]
                batch = DL.DeviceBatch(torch.device("cpu"), mode)
                for it in want["items"]:
                    np.random.seed(7 + it["idx"])
                    clip, gt = ds[it["idx"]]
                    assert list(clip.shape) == it["clip_shape"] and _crc(clip.numpy()) == it["clip_crc"]
                    x, gd = batch(([clip], [gt]))
                    assert x.shape == (1, TR.HOLLY_T, 3, 224, 384)
                    if mode == "val":
                        maps = gt.numpy()[None] if mf == 0 else gt.numpy()
                        g = np.stack([(m.astype('float') / 255.0 if m.max() > 1 else m.astype('float')) for m in maps]).astype(np.float32)
                        g = g[0] if mf == 0 else g
                        assert it["gt_dtype"] == "torch.float32" and list(g.shape) == it["gt_shape"] and _crc(g) == it["gt_crc"]
                        assert np.array_equal(gd[0].numpy(), g)
                    else:
                        assert tuple(gd.shape[-2:]) == (224, 384)
    finally:
        L._install_test_double(None)


def test_frame_ring_windows_are_views_of_the_newest_frames():
    """FrameRing: any run of consecutive frames among the newest `capacity` is one contiguous slice of the doubled buffer;
    windows() returns overlapping zero-copy views"""
    import random
    from vinet_amd.generate_result import FrameRing
    fr = torch.randn(41, 3, 4, 5)
    for R in (7, 12):
        ring, pos, T = FrameRing(R, fr.shape[1:], "cpu"), 0, 4
        random.seed(R)
        while pos < fr.shape[0]:
            k = min(random.randint(1, R), fr.shape[0] - pos)
            ring.push(fr[pos:pos + k])
            pos += k
            for first in range(max(0, pos - R), pos - T + 1):
                n = min(R - T + 1, pos - first - T + 1)
                w = ring.windows(first, n, T)
                assert w.data_ptr() >= ring.buf.data_ptr() and w.untyped_storage().data_ptr() == ring.buf.untyped_storage().data_ptr()
                assert torch.equal(w, torch.stack([fr[first + i:first + i + T] for i in range(n)]))


def test_streaming_harness_equals_the_resident_one():
    """predict_stream (frames arrive in chunks, clips are views into a ring of the newest frames) produces exactly the maps of
    predict_video (whole video resident, one gather per call) -- and thereby the reference's schedule (the trace golden above):
    every output frame once, early frames from the time-reversed window"""
    from tests.abi_emulator import AbiEmulator
    from vinet_amd import _lib as L
    from vinet_amd import engine as E
    from vinet_amd import generate_result as GR
    from vinet_amd import model as VM
    from vinet_amd import synth
    L._install_test_double(AbiEmulator())
    try:
        E.set_default_dtype("fp32")
        m = VM.VideoSaliencyModel(num_clips=8).eval()
        m.load_state_dict(synth.synth_state_dict(m.state_dict(), 3))
        N, T = 19, 8
        frames = synth.clip(1, N, 64, 96, 5)[0]                      # [N,3,64,96]
        want = GR.predict_video(m, frames, T, batch=1)
        for batch, chunk in ((1, 1), (3, 5), (2, 32)):
            got = torch.full_like(want, float("nan"))
            seen = []
            for outs, maps in GR.predict_stream(m, (frames[c:c + chunk] for c in range(0, N, chunk)), T, batch=batch):
                for i, mp in zip(outs, maps):
                    got[i] = mp
                    seen.append(i)
            assert sorted(seen) == list(range(N)), (batch, chunk, seen)
            # one clip per call: bit for bit; several per call: the CPU model's convolutions round differently per batch size
            assert torch.equal(got, want) if batch == 1 else float((got - want).abs().max()) < 5e-6, (batch, chunk, float((got - want).abs().max()))
        pp = GR.predict_video(m, frames, T, batch=2, out_size=(45, 80))
        got = torch.zeros_like(pp)
        for outs, maps in GR.predict_stream(m, (frames[c:c + 4] for c in range(0, N, 4)), T, batch=2, out_size=(45, 80)):
            for i, mp in zip(outs, maps):
                got[i] = mp
        d = (got.int() - pp.int()).abs()
        assert int(d.max()) <= 1 and float((d > 0).float().mean()) < 1e-3          # (uint8 maps: a float ulp can move a byte)
    finally:
        L._install_test_double(None)
        E.set_default_dtype("bf16")


def test_fold_list_parser_and_s3d_key_rule(tmp_path):
    """generate_result_audio_visual.py:22-30 (three words per line) and the key rule of train.py:144-170"""
    from vinet_amd import generate_result_audio_visual as AV
    f = tmp_path / "list.txt"
    f.write_text("clip_1 120 25\n\nclip_2  80 29.97\n")
    assert AV.read_sal_text(str(f)) == [("clip_1", "120", "25"), ("clip_2", "80", "29.97")]
    assert TR._s3d_key("base.0.conv_s.weight") == "base1.0.conv_s.weight"
    assert TR._s3d_key("base.4.x") == "base1.4.x" and TR._s3d_key("base.5.x") == "base2.0.x"
    assert TR._s3d_key("base.7.x") == "base2.2.x" and TR._s3d_key("base.8.x") == "base3.0.x"
    assert TR._s3d_key("base.13.x") == "base3.5.x" and TR._s3d_key("base.14.x") == "base4.0.x"
    assert TR._s3d_key("module.base.15.branch3.1.bn.bias") == "base4.1.branch3.1.bn.bias"
    assert TR._s3d_key("fc.0.weight") == "fc.0.weight"


def test_device_batch_keeps_one_audio_table_per_dataset():
    """Coutrot_db1 / Coutrot_db2 style name clashes: two datasets with a video of the SAME folder name and different audio.
    An item carries its dataset's table key, so the excerpt is cut from its own dataset's waveform (the reference keeps one
    table per SoundDatasetLoader, dataloader.py:181-186); the waveform cache is bounded."""
    from tests.abi_emulator import AbiEmulator
    from vinet_amd import _lib as L
    from vinet_amd import dataloader as DL
    from vinet_amd import preprocess as PR
    L._install_test_double(AbiEmulator())
    try:
        n = 3 * PR.MAX_AUDIO_WIN
        wa = torch.linspace(-1, 1, n).view(1, n).contiguous()
        wb = (-wa).contiguous()
        tables = {("Coutrot_db1", "train"): {"clip_1": {"wav": wa}}, ("Coutrot_db2", "train"): {"clip_1": {"wav": wb}}}
        batch = DL.DeviceBatch(torch.device("cpu"), "train", audio_tables=tables)
        lo, hi = 1000, 1000 + 20000
        a = batch._audio(("clip_1", (lo, hi), ("Coutrot_db1", "train")))
        b = batch._audio(("clip_1", (lo, hi), ("Coutrot_db2", "train")))
        assert a.shape == (1, PR.MAX_AUDIO_WIN, 1) and torch.equal(a, -b) and float(a.abs().max()) > 0
        assert torch.equal(a.view(-1), PR.audio_excerpt(wa[0], lo, hi).view(-1))
        # single-table form (one dataset) still works with two-field references
        one = DL.DeviceBatch(torch.device("cpu"), "train", audiodata=tables[("Coutrot_db2", "train")])
        assert torch.equal(one._audio(("clip_1", (lo, hi))), b)
        # bounded cache
        batch.WAV_CACHE = 1
        batch._audio(("clip_1", (lo, hi), ("Coutrot_db1", "train")))
        assert len(batch._wav) == 1
        assert torch.equal(batch._audio(("clip_1", None, ("Coutrot_db1", "train"))), torch.zeros(1, PR.MAX_AUDIO_WIN, 1))
    finally:
        L._install_test_double(None)
