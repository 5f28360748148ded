"""Host logic of vinet_amd (tap tables, dgrad phases, BN bookkeeping, concat
plumbing, backward tape, losses glue, fused optimizer) on CPU, with the C ABI
served by tests/abi_emulator.py, against the golden vectors of the reference."""
import json

import numpy as np
import pytest
import torch

from oracle import vinet_cpu as O
from tests import goldens as G
from tests import model_cases as MC
from tests.abi_emulator import AbiEmulator
from vinet_amd import _lib as L
from vinet_amd import engine as E
from vinet_amd import synth


@pytest.fixture(autouse=True)
def _emulated_abi():
    L._install_test_double(AbiEmulator())
    old = E.default_dtype()
    E.set_default_dtype("fp32")
    yield
    E.set_default_dtype("bf16" if old == E.BF16 else "fp32")
    L._install_test_double(None)


CPU = torch.device("cpu")


@pytest.mark.parametrize("name", ["basic_16_32", "sep_16_32_k3", "sep_3_64_k7s2", "mixed_3b"])
@pytest.mark.parametrize("mode", ["eval", "train"])
def test_blocks(name, mode):
    MC.block_case(name, mode, CPU)


@pytest.mark.parametrize("mode", ["eval", "train"])
def test_mixed_5b_block_at_4x7x12(mode):
    """the M = 336-voxel stage (base4 of a 32 x 224 x 384 clip): outputs, input and parameter gradients of Mixed_5b against the
    reference's (compact golden: strided samples + norms)"""
    MC.block_case_compact("mixed_5b", mode, CPU, 2e-4, 2e-4)


@pytest.mark.slow
@pytest.mark.parametrize("name", ["mixed_3c", "mixed_4b", "mixed_4f", "mixed_5c", "soundnet"])
def test_more_inception_blocks_train_mode(name):
    """round 6: compact goldens from the reference for the Inception stages that had none: the engine
    on the CPU model of the C ABI, training mode (BatchNorm statistics, every gradient)"""
    MC.block_case_compact(name, "train", CPU, 2e-4, 2e-4)


def test_weight_shared_conv_gradients():
    """one ConvPlan run twice in a backward: one unpack job per weight-gradient workspace (engine.Ctx.flush_unpack)"""
    MC.weight_shared_case(torch.device("cpu"), E.F32, 2e-4)


def test_dgrad_phase_decomposition_matches_conv_transpose():
    """every (k, s, p) the nets use, plus a few odd ones, as 1-D identities"""
    for I, k, s, p in [(32, 7, 2, 3), (12, 3, 3, 0), (20, 5, 5, 0), (9, 3, 1, 1), (70560, 64, 2, 32), (11, 4, 2, 2), (7, 2, 3, 0)]:
        O = (I + 2 * p - k) // s + 1
        if I > 1000:
            I, O = 200, (200 + 2 * p - k) // s + 1
        w = torch.randn(1, 1, k, dtype=torch.float64)
        dy = torch.randn(1, 1, O, dtype=torch.float64)
        x = torch.zeros(1, 1, I, dtype=torch.float64, requires_grad=True)
        torch.nn.functional.conv1d(x, w, stride=s, padding=p).backward(dy)
        got = torch.zeros(I, dtype=torch.float64)
        for r, Q, taps in E._phase_taps_1d(I, O, k, s, p):
            for q in range(Q):
                acc = 0.0
                for off, ki in taps:
                    o = q + off
                    if 0 <= o < O:
                        acc += float(dy[0, 0, o]) * float(w[0, 0, ki])
                got[r + q * s] = acc
        assert torch.allclose(got, x.grad[0, 0], atol=1e-12), (I, k, s, p)


def test_losses_match_reference_goldens():
    MC.losses_case(CPU)


def test_decoder8_forward_backward():
    MC.decoder8_case(CPU)


@pytest.mark.parametrize("clips", [16, 32, 48])
def test_decoder16_48_forward_backward(clips):
    """the clip-length specific decoder tails (model.py:339-346, 463-469; the 48-frame one has a biased (3,1,1) conv)"""
    MC.decoder_case(clips, CPU)


def test_loss_func_matches_reference_goldens():
    MC.loss_func_case(CPU)


def test_64_frame_model_ends_at_one_frame():
    """BASELINE config 5 (build-defined tail): host logic against the labelled oracle restatement"""
    from oracle import vinet_cpu as O
    from vinet_amd import model as VM
    from vinet_amd import synth
    m = VM.VideoSaliencyModel(num_clips=64).eval()
    o = O.VideoSaliencyModel(num_clips=64).eval()
    sd = synth.synth_state_dict(o.state_dict(), 3)
    m.load_state_dict(sd)
    o.load_state_dict(sd)
    x = synth.clip(1, 64, 32, 32, 3).permute(0, 2, 1, 3, 4)
    with torch.no_grad():
        MC.close(m(x), o(x), 2e-5, "64-frame map")


@pytest.mark.slow
def test_e2e_8x96x192_inference():
    MC.e2e_case("8x96x192", CPU)


@pytest.mark.slow
def test_train_step_matches_reference():
    MC.train_step_case(CPU)


def test_train_step_with_torch_adam_gives_same_update():
    """drop-in: torch.optim.Adam on our parameters must see the gradients the tape wrote"""
    import torch.optim
    from vinet_amd import model_utils as MU
    m = MU.BasicConv3d(16, 32, 1, 1).train()
    x = synth.normal("ta", (2, 16, 2, 4, 4), 1)
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    opt.zero_grad()
    m(x).sum().backward()
    assert all(p.grad is not None for p in m.parameters())
    before = m.conv.weight.detach().clone()
    opt.step()
    assert not torch.equal(before, m.conv.weight.detach())


def test_eval_fold_cache_follows_parameter_and_statistics_changes():
    """eval-mode BN folds are cached per module; they must be recomputed after load_state_dict, after a
    training-mode forward (running statistics move) and after an optimizer step, and reused otherwise"""
    from vinet_amd import model_utils as MU
    from vinet_amd import optim as VO
    m = MU.BasicConv3d(16, 32, 1, 1)
    o = O.BasicConv3d(16, 32, 1, 1)
    sd = synth.synth_state_dict(o.state_dict(), 5)
    m.load_state_dict(sd)
    o.load_state_dict(sd)
    x = synth.normal("fc", (2, 16, 2, 4, 4), 1)

    def check(what):
        m.eval()
        o.eval()
        with torch.no_grad():
            MC.close(m(x), o(x), 1e-5, what)

    check("initial")
    check("cached")
    sd2 = synth.synth_state_dict(o.state_dict(), 6)
    m.load_state_dict(sd2)
    o.load_state_dict(sd2)
    check("after load_state_dict")
    for mod in (m, o):                       # a training-mode forward moves the running statistics
        mod.train()
        mod(x * 3 + 1)
    check("after a training forward")
    m.train()
    opt = VO.Adam(m.parameters(), lr=1e-2)
    opt.zero_grad()
    m(x).sum().backward()
    opt.step()
    o.load_state_dict(m.state_dict())
    check("after an optimizer step")


def test_product_path_refuses_cpu_without_library_double():
    """no CPU fallback: without the test double a CPU tensor must raise"""
    from vinet_amd import model_utils as MU
    L._install_test_double(None)
    try:
        with pytest.raises(Exception):
            MU.BasicConv3d(16, 32, 1, 1)(torch.zeros(1, 16, 1, 2, 2))
    finally:
        L._install_test_double(AbiEmulator())


def test_postprocess_host_api_follows_process_and_validate():
    """vinet_amd.utils.resize_blur / to_uint8 / postprocess / blur and the harness's out_size path against the
    oracle's statement of generate_result.py:95-104 (through the ABI emulator: host logic only)"""
    from oracle import postproc_cpu as P
    from vinet_amd import generate_result as GR
    from vinet_amd import utils as U
    maps = torch.sigmoid(synth.normal("pp_host", (3, 24, 40), 5))
    ref = P.resize_blur(maps.numpy(), 45, 80)
    got = U.resize_blur(maps, (45, 80))
    assert got.shape == (3, 45, 80) and np.array_equal(got.numpy(), ref)
    assert np.array_equal(U.resize_blur(maps[0], (45, 80)).numpy(), ref[0])                       # [H,W] in -> [H,W] out
    assert np.array_equal(U.blur(maps[1].numpy()).numpy(), P.gaussian_blur11(maps[1].numpy()))    # utils.py:61-64
    u8 = U.postprocess(maps, (45, 80))
    assert u8.dtype == torch.uint8 and np.array_equal(u8.numpy(), P.normalize_u8(ref))
    assert np.array_equal(U.to_uint8(torch.from_numpy(ref[2])).numpy(), P.normalize_u8(ref[2]))

    class Fake(torch.nn.Module):                      # "saliency" = mean over channels and time: checks the frame routing
        def __init__(self):
            super().__init__()
            self.p = torch.nn.Parameter(torch.zeros(1))

        def forward(self, clips):
            return torch.sigmoid(clips.mean((1, 2)) + clips[:, 0, -1])

    T, N = 3, 7
    frames = synth.normal("pp_frames", (N, 3, 8, 12), 2)
    raw = GR.predict_video(Fake(), frames, T, batch=2)
    out = GR.predict_video(Fake(), frames, T, batch=2, out_size=(20, 18))
    assert out.dtype == torch.uint8 and out.shape == (N, 20, 18)
    assert np.array_equal(out.numpy(), P.normalize_u8(P.resize_blur(raw.numpy(), 20, 18)))
    u8 = GR.process(Fake(), frames[:T].permute(1, 0, 2, 3)[None], None, None, None, None, (18, 20))   # img_size = (w, h)
    assert u8.shape == (20, 18)


def test_preprocess_host_api_and_directory_harness(tmp_path):
    """vinet_amd.preprocess against the oracle, then generate_result.validate() on a directory of PNG frames: the saved
    images must be what the reference's flow produces -- PIL resize + normalise, model call per the sliding-window
    schedule, cv2.resize + blur + img_save -- here with every stage after the decoder behind the C ABI"""
    import argparse
    import os
    from PIL import Image
    from oracle import postproc_cpu as P
    from oracle import preproc_cpu as Q
    from vinet_amd import generate_result as GR
    from vinet_amd import preprocess as PR
    rng = np.random.default_rng(11)
    u8 = rng.integers(0, 256, (3, 30, 44, 3), dtype=np.uint8)
    assert np.array_equal(PR.frames_to_tensor(torch.from_numpy(u8), (16, 24)).numpy(), Q.frames_preprocess(u8, 16, 24))
    assert PR.frames_to_tensor(torch.from_numpy(u8[0])).shape == (3, 224, 384)
    g8 = rng.integers(0, 256, (2, 30, 44), dtype=np.uint8)
    assert np.array_equal(PR.gt_to_tensor(torch.from_numpy(g8), (16, 24)).numpy(), Q.gt_preprocess(g8, 16, 24))
    assert np.array_equal(PR.gt_to_tensor(torch.from_numpy(g8[0])).numpy(), Q.gt_preprocess(g8[:1])[0])

    class Fake(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.p = torch.nn.Parameter(torch.zeros(1))

        def forward(self, clips):
            return torch.sigmoid(clips.mean((1, 2)) + clips[:, 0, -1])

    T, N, h, w = 3, 6, 30, 44
    src = tmp_path / "in"
    frames = {}
    for v in ("b_video", "a_video", "short"):
        os.makedirs(src / v / "images")
        n = 2 if v == "short" else N
        frames[v] = rng.integers(0, 256, (n, h, w, 3), dtype=np.uint8)
        for i in range(n):
            Image.fromarray(frames[v][i]).save(src / v / "images" / ("%04d.png" % (i + 1)))
    t, sz = PR.torch_transform(str(src / "a_video" / "images" / "0001.png"))
    assert sz == (w, h) and np.array_equal(t.numpy(), Q.frames_preprocess(frames["a_video"][:1])[0])
    args = argparse.Namespace(path_indata=str(src), save_path=str(tmp_path / "out"), start_idx=-1, num_parts=4, clip_size=T, batch=2, graph=0)
    assert GR.list_videos(str(src)) == ["a_video", "b_video", "short"]
    assert GR.validate(args, Fake(), torch.device("cpu")) == 2 * N          # "short" has fewer than 2T-1 frames: skipped
    for v in ("a_video", "b_video"):
        x = torch.from_numpy(Q.frames_preprocess(frames[v]))
        raw = GR.predict_video(Fake(), x, T)
        want = P.normalize_u8(P.resize_blur(raw.numpy(), h, w))
        for i in range(N):
            got = np.asarray(Image.open(tmp_path / "out" / v / ("%04d.png" % (i + 1))))
            assert got.shape == (h, w) and np.array_equal(got, want[i])
    assert not os.listdir(tmp_path / "out" / "short")


def test_audio_feature_host_api():
    """vinet_amd.preprocess.get_audio_feature keeps dataloader.py:89-122's arguments and result"""
    from oracle import preproc_cpu as Q
    from vinet_amd import preprocess as PR
    rng = np.random.default_rng(4)
    wav = (rng.standard_normal((1, 150000)) * 2 ** -8).astype(np.float32)
    starts, ends = PR.audio_frame_bounds(160, 25.0, 22050, wav.shape[1])
    s2, e2 = Q.audio_frame_bounds(160, 25.0, 22050, wav.shape[1])
    assert np.array_equal(starts, s2) and np.array_equal(ends, e2)
    data = {"vid": dict(wav=torch.from_numpy(wav), starts=starts, ends=ends)}
    f = PR.get_audio_feature("vid", data, 32, 10)
    assert f.shape == (1, 70560, 1) and np.array_equal(f.view(-1).numpy(), Q.audio_excerpt(wav[0], starts[11], ends[42]))
    f = PR.get_audio_feature("vid", data, 32, 140)                         # runs past the last frame: ends[-1]
    assert np.array_equal(f.view(-1).numpy(), Q.audio_excerpt(wav[0], starts[141], ends[-1]))
    assert float(PR.get_audio_feature("nope", data, 32, 0).abs().sum()) == 0.0
    assert PR.MAX_AUDIO_WIN == 70560


def test_audio_visual_directory_harness(tmp_path):
    """generate_result_audio_visual.py:115-192 on a synthetic DIEM-like tree: frames as PNG, 16-bit WAV, fold list; the
    flipped clips must get the time-flipped excerpt"""
    import argparse
    import os
    import wave
    from PIL import Image
    from oracle import postproc_cpu as P
    from oracle import preproc_cpu as Q
    from vinet_amd import generate_result as GR
    from vinet_amd import generate_result_audio_visual as AV
    rng = np.random.default_rng(21)
    T, N, h, w, fps, Fs = 3, 7, 24, 36, 25, 8000
    root = tmp_path / "data"
    os.makedirs(root / "fold_lists")
    os.makedirs(root / "video_frames" / "DIEM" / "clipA")
    os.makedirs(root / "video_audio" / "DIEM" / "clipA")
    os.makedirs(root / "annotations" / "DIEM" / "clipA" / "maps")
    (root / "fold_lists" / "DIEM_list_test_fps.txt").write_text("clipA %d %d\n" % (N, fps))
    u8 = rng.integers(0, 256, (N, h, w, 3), dtype=np.uint8)
    for i in range(N):
        Image.fromarray(u8[i]).save(root / "video_frames" / "DIEM" / "clipA" / ("img_%05d.png" % (i + 1)))
        Image.fromarray(u8[i, :, :, 0]).save(root / "annotations" / "DIEM" / "clipA" / "maps" / ("eyeMap_%05d.png" % (i + 1)))
    pcm = rng.integers(-20000, 20000, 4000, dtype=np.int16)
    with wave.open(str(root / "video_audio" / "DIEM" / "clipA" / "clipA.wav"), "wb") as f:
        f.setnchannels(1); f.setsampwidth(2); f.setframerate(Fs); f.writeframes(pcm.tobytes())
    wav, fs = AV.load_wav(str(root / "video_audio" / "DIEM" / "clipA" / "clipA.wav"))
    assert fs == Fs and wav.shape == (1, 4000) and np.array_equal(wav[0].numpy(), pcm.astype(np.float32) * 65536.0)

    ramp = torch.linspace(-1.0, 1.0, 70560)

    class FakeAV(torch.nn.Module):                    # the audio term is odd under a time flip of the excerpt
        def __init__(self):
            super().__init__()
            self.p = torch.nn.Parameter(torch.zeros(1))

        def forward(self, clips, audio):
            assert audio.shape[1:] == (1, 70560, 1)
            a = (audio.reshape(audio.shape[0], -1) * ramp).sum(1) * 1e-3
            return torch.sigmoid(clips.mean((1, 2)) + clips[:, 0, -1] + a[:, None, None])

    args = argparse.Namespace(path_indata=str(root), save_path=str(tmp_path / "out"), dataset="DIEM", split=1, start_idx=-1, num_parts=4,
                              clip_size=T, use_sound=True, batch=2)
    assert AV.validate(args, FakeAV(), torch.device("cpu")) == N
    # the same thing by hand: oracle pre-processing, the reference's schedule call by call, oracle post-processing
    x = torch.from_numpy(Q.frames_preprocess(u8))
    wav_s = (pcm.astype(np.float32) * np.float32(65536.0) * np.float32(2 ** -23)).astype(np.float32)
    st, en = Q.audio_frame_bounds(N, float(fps), Fs, 4000)
    raw = np.zeros((N, 224, 384), np.float32)
    m = FakeAV()
    for (o, clip, flipped) in GR.sliding_window_schedule(N, T):
        s0 = min(clip)
        e_idx = en[-1] if s0 + T >= len(en) else en[s0 + T]
        a = torch.from_numpy(Q.audio_excerpt(wav_s, st[s0 + 1], e_idx)).view(1, 1, -1, 1)
        if flipped:
            a = torch.flip(a, [2])
        raw[o] = m(x[clip].permute(1, 0, 2, 3)[None], a)[0].detach().numpy()
    want = P.normalize_u8(P.resize_blur(raw, h, w))
    for i in range(N):
        got = np.asarray(Image.open(tmp_path / "out" / "clipA" / ("img_%05d.png" % (i + 1))))
        assert np.array_equal(got, want[i]), i
    # without sound the visual harness is used as is
    args.use_sound = False
    args.save_path = str(tmp_path / "out2")

    class Fake(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.p = torch.nn.Parameter(torch.zeros(1))

        def forward(self, clips):
            return torch.sigmoid(clips.mean((1, 2)))

    assert AV.validate(args, Fake(), torch.device("cpu")) == N
