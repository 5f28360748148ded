#!/usr/bin/env python3
"""Audio-visual inference harness -- counterpart of generate_result_audio_visual.py on the MI355X path.

Same flags, same directory layout (`<path_indata>/fold_lists/<list>`, `video_frames/<dataset>/<video>/`,
`video_audio/<dataset>/<video>/<video>.wav`, `annotations/<dataset>/<video>/maps`), same sliding-window / time-flip
schedule (generate_result_audio_visual.py:165-190: the flipped clip gets the time-flipped audio excerpt).  As in
vinet_amd/generate_result.py only the decoders (PIL, the wav reader) and the image encoder run on the host: frames go
up as bytes, the waveform of a video is uploaded once and every call's Hanning-windowed excerpt is cut on the device
(vinet_audio_excerpt), maps come back as uint8.

`torchaudio.load(path, normalization=False)` (generate_result_audio_visual.py:56; torchaudio==0.4.0 is absent here)
returns sox's 32-bit left-justified samples as floats, i.e. a 16-bit PCM sample s as s * 65536; the reference then scales
by 2**-23 (`:57`).  `load_wav` restates that for PCM WAV files with the standard library's `wave` module.
"""
import argparse
import os
import sys
from os.path import join

import numpy as np
import torch

from . import generate_result as GR
from . import preprocess


def read_sal_text(txt_file):
    """A fold list (`<video> <frame count> <fps>` per line, generate_result_audio_visual.py:22-30) as a list of
    (video, frame count, fps) string triples in file order; blank lines are skipped."""
    with open(txt_file, 'r') as f:
        rows = [ln.split() for ln in f]
    return [(r[0], r[1], r[2]) for r in rows if r]


def load_wav(path):
    """(float32 [channels, samples] in sox's 32-bit sample scale, Fs) -- torchaudio.load(path, normalization=False)."""
    import wave
    with wave.open(path, 'rb') as w:
        nch, width, fs, n = w.getnchannels(), w.getsampwidth(), w.getframerate(), w.getnframes()
        raw = w.readframes(n)
    if width == 2:
        a = np.frombuffer(raw, dtype='<i2').astype(np.float32) * np.float32(65536.0)
    elif width == 4:
        a = np.frombuffer(raw, dtype='<i4').astype(np.float32)
    elif width == 1:
        a = (np.frombuffer(raw, dtype=np.uint8).astype(np.float32) - np.float32(128.0)) * np.float32(16777216.0)
    else:
        raise ValueError("load_wav: %d-byte PCM samples are not supported (%s)" % (width, path))
    return torch.from_numpy(a.reshape(-1, nch).T.copy()), fs


def make_dataset(annotation_path, audio_path, gt_path, device=None):
    """generate_result_audio_visual.py:32-86: per video of the fold list the waveform (scaled by 2**-23, on `device`) and
    the audio sample range of every frame.  Videos with at most one annotated frame or without a .wav are skipped (and
    reported with the reference's messages)."""
    entries = read_sal_text(annotation_path)
    table = {}
    for n, (video, _, fps) in enumerate(entries):
        if n % 100 == 0:
            print('dataset loading [{}/{}]'.format(n, len(entries)))
        n_maps = len(os.listdir(join(gt_path, video, 'maps')))
        if n_maps <= 1:
            print("Less frames")
            continue
        wav_file = join(audio_path, video, video + '.wav')
        if not os.path.exists(wav_file):
            print("Not exists", wav_file)
            continue
        wav, rate = load_wav(wav_file)
        wav = wav * (2 ** -23)
        starts, ends = preprocess.audio_frame_bounds(n_maps, fps, rate, wav.shape[1])
        table[video] = {'audiopath': audio_path, 'video_id': video, 'Fs': rate, 'starts': starts, 'ends': ends,
                        'wav': wav if device is None else wav.to(device)}
    return table


def get_audio_feature(audioind, audiodata, args, start_idx):
    """generate_result_audio_visual.py:88-113: [1, 1, 70560, 1]; the first channel of the waveform, like the `[1, L]`
    excerpt the reference's assignment broadcasts from."""
    info = audiodata.get(audioind)
    if info is None:
        return torch.zeros(1, 1, preprocess.MAX_AUDIO_WIN, 1)
    start = info['starts'][start_idx + 1]
    if start_idx + args.clip_size >= len(info['ends']):
        print("Exceeds size", audioind)
        sys.stdout.flush()
        end = info['ends'][-1]
    else:
        end = info['ends'][start_idx + args.clip_size]
    return preprocess.audio_excerpt(info['wav'][0].contiguous(), start, end).view(1, 1, -1, 1)


@torch.no_grad()
def predict_video(model, frames, T, audio_for_start=None, batch=1, out_size=None):
    """generate_result.predict_video with the audio branch: `audio_for_start(s)` is the excerpt [1,1,70560,1] of the clip
    that starts at frame s; the time-flipped clip gets the flipped excerpt (generate_result_audio_visual.py:183-186)."""
    if audio_for_start is None:
        return GR.predict_video(model, frames, T, batch, out_size)
    from .utils import postprocess
    N = frames.shape[0]
    sched = GR.sliding_window_schedule(N, T)
    assert sched, "more frames are needed (N >= 2T-1)"
    if out_size is None:
        maps = torch.empty((N,) + tuple(frames.shape[2:]), dtype=torch.float32, device=frames.device)
    else:
        maps = torch.empty((N, int(out_size[0]), int(out_size[1])), dtype=torch.uint8, device=frames.device)
    model.eval()
    for s in range(0, len(sched), batch):
        chunk = sched[s:s + batch]
        idx = torch.tensor([c[1] for c in chunk], device=frames.device)
        clips = frames[idx].permute(0, 2, 1, 3, 4)
        audio = []
        for (_, clip, flipped) in chunk:
            a = audio_for_start(min(clip)).to(frames.device)
            audio.append(torch.flip(a, [2]) if flipped else a)
        y = model(clips, torch.cat(audio, 0))
        if out_size is not None:
            y = postprocess(y, out_size)
        maps[torch.tensor([c[0] for c in chunk], device=frames.device)] = y
    return maps


@torch.no_grad()
def validate(args, model=None, device=None):
    """generate_result_audio_visual.py:115-192."""
    from PIL import Image
    dev = device if device is not None else torch.device('cuda')
    T = args.clip_size
    file_name = 'DIEM_list_test_fps.txt' if args.dataset == 'DIEM' else '{}_list_test_{}_fps.txt'.format(args.dataset, args.split)
    list_indata = []
    with open(join(args.path_indata, 'fold_lists', file_name), 'r') as f:
        for line in f.readlines():
            list_indata.append(line.split(' ')[0].strip())
    list_indata.sort()
    audiodata = None
    if args.use_sound:
        audiodata = make_dataset(join(args.path_indata, 'fold_lists', file_name), join(args.path_indata, 'video_audio', args.dataset),
                                 join(args.path_indata, 'annotations', args.dataset), dev)
    if args.start_idx != -1:
        _len = (1.0 / float(args.num_parts)) * len(list_indata)
        list_indata = list_indata[int((args.start_idx - 1) * _len): int(args.start_idx * _len)]
    n_saved = 0
    for dname in list_indata:
        print('processing ' + dname, flush=True)
        img_dir = os.path.join(args.path_indata, 'video_frames', args.dataset, dname)
        list_frames = sorted(f for f in os.listdir(img_dir) if os.path.isfile(os.path.join(img_dir, f)))
        os.makedirs(join(args.save_path, dname), exist_ok=True)
        if len(list_frames) < 2 * T - 1:
            print(' more frames are needed')
            continue
        w, h = Image.open(os.path.join(img_dir, list_frames[0])).size
        chunk = int(getattr(args, "decode_chunk", 32))

        def decoded():      # the streaming schedule of generate_result.predict_stream: 32 frames decoded / uploaded / pre-processed at a time
            for c0 in range(0, len(list_frames), chunk):
                imgs = [Image.open(os.path.join(img_dir, f)).convert('RGB') for f in list_frames[c0:c0 + chunk]]
                assert all(im.size == (w, h) for im in imgs), "frames of one video must share a size (%s)" % dname
                yield preprocess.frames_to_tensor(torch.from_numpy(np.stack([np.asarray(im) for im in imgs])).to(dev))

        def audio(starts, flipped):     # excerpt of the clip that starts at frame s; the time-flipped clip gets the flipped excerpt
            a = torch.cat([get_audio_feature(dname, audiodata, args, s).to(dev) for s in starts], 0)
            return (torch.flip(a, [2]) if flipped else a,)
        pend_i, pend_m = [], []

        def flush():
            nonlocal n_saved
            if pend_i:
                host = torch.cat(pend_m).cpu().numpy()
                for i, m in zip(pend_i, host):
                    fp = join(args.save_path, dname, list_frames[i])
                    im = Image.fromarray(m)
                    im.save(fp) if fp.split('.')[-1] == "png" else im.save(fp, quality=100)
                    n_saved += 1
                del pend_i[:], pend_m[:]
        for outs, maps in GR.predict_stream(model, decoded(), T, getattr(args, "batch", 1), (h, w), False, audio if args.use_sound else None):
            pend_i.extend(outs)
            pend_m.append(maps)
            if len(pend_i) >= chunk:
                flush()
        flush()
    return n_saved


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument('--file_weight', default="./saved_models/no_trans_upsampling_reduced.pt", type=str)
    p.add_argument('--allow_synthetic_weights', default=0, type=int, help="1 = fall back to procedural weights when --file_weight is missing")
    p.add_argument('--nhead', default=4, type=int)
    p.add_argument('--num_encoder_layers', default=3, type=int)
    p.add_argument('--transformer_in_channel', default=512, type=int)
    p.add_argument('--save_path', default='/ssd_scratch/cvit/samyak/Results/AViNet_Diem', type=str)
    p.add_argument('--start_idx', default=-1, type=int)
    p.add_argument('--num_parts', default=4, type=int)
    p.add_argument('--split', default=1, type=int)
    p.add_argument('--path_indata', default='/ssd_scratch/cvit/samyak/data/', type=str)
    p.add_argument('--dataset', default='DIEM', type=str)
    p.add_argument('--multi_frame', default=0, type=int)
    p.add_argument('--decoder_upsample', default=1, type=int)
    p.add_argument('--num_decoder_layers', default=-1, type=int)
    p.add_argument('--num_hier', default=3, type=int)
    p.add_argument('--clip_size', default=32, type=int)
    p.add_argument('--use_sound', default=False, type=bool)
    p.add_argument('--compute_dtype', default="fp32s", choices=["bf16", "fp32", "fp32s"],
                   help="arithmetic of the HIP path.  Default fp32s (split-bf16 products, fp32 tensors): INSIDE the reference contract -- maps within "
                        "1e-3 of the PyTorch-CPU path, exact argmax.  bf16 is the throughput mode (2.9x faster; maps within 2.5e-2, gradients of the "
                        "encoder noisy: DESIGN.md) and must be asked for; fp32 is the exact-fp32-MFMA path")
    p.add_argument('--batch', default=1, type=int)
    return p


def main(argv=None):
    import time
    from . import engine, model, synth
    args = build_parser().parse_args(argv)
    print(args)
    dev = torch.device('cuda')
    engine.set_default_dtype(args.compute_dtype)
    kw = dict(transformer_in_channel=args.transformer_in_channel, nhead=args.nhead, use_upsample=bool(args.decoder_upsample),
              num_hier=args.num_hier, num_clips=args.clip_size)
    m = model.VideoAudioSaliencyModel(**kw) if args.use_sound else model.VideoSaliencyModel(**kw)
    if os.path.isfile(args.file_weight):
        m.load_state_dict(torch.load(args.file_weight, map_location="cpu"))
    elif args.allow_synthetic_weights:
        print("weight file? using procedural weights")
        m.load_state_dict(synth.synth_state_dict(m.state_dict(), 0))
    else:
        # torch.load of the reference fails on a missing checkpoint (generate_result_audio_visual.py:138)
        raise FileNotFoundError("--file_weight %r does not exist (pass --allow_synthetic_weights 1 to run on procedural weights)" % args.file_weight)
    m = m.to(dev).eval()
    t0 = time.time()
    n = validate(args, m, dev)
    print("%d saliency images written in %.3f s" % (n, time.time() - t0))
    return n


if __name__ == "__main__":
    main()
