"""Deterministic, version-independent synthetic weights and inputs.

Real ViNet checkpoints are absent and ``torch.manual_seed``
streams change between torch versions, so every golden vector, parity test and
bench run draws its numbers from this counter-based generator instead:
``value[i] = f(splitmix64(key(name, seed) + i))``.  The same (name, shape,
seed) gives the same bits on any box.

The weight recipe follows SURVEY.md section 8(c): He-scaled conv weights,
BN gamma in [0.5, 1.5], beta ~ N(0, 0.1), running mean ~ N(0, 0.1), running
var in [0.5, 1.5].  The final decoder conv (``decoder.convtsp4.8``) is then
calibrated by the caller (see ``calibrate_head``) so pre-sigmoid logits are
about N(-3, 1) -- default init gives an almost constant map, which would make
a 1e-3 parity check vacuous.
"""
import hashlib
import math

import numpy as np
import torch

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _key(name, seed):
    h = hashlib.sha256(("%s|%d" % (name, seed)).encode()).digest()
    return np.uint64(int.from_bytes(h[:8], "little"))


def _splitmix64(x):
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return z ^ (z >> np.uint64(31))


def uniform01(name, n, seed=0, stream=0):
    """n doubles in [0, 1), 53-bit."""
    with np.errstate(over="ignore"):
        base = _key(name, seed) + np.uint64(stream) * np.uint64(0xD1B54A32D192ED03)
        ctr = base + np.arange(n, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)
    bits = _splitmix64(ctr)
    return (bits >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))


def normal(name, shape, seed=0, mean=0.0, std=1.0):
    n = int(np.prod(shape)) if len(shape) else 1
    u1 = uniform01(name, n, seed, 0)
    u2 = uniform01(name, n, seed, 1)
    z = np.sqrt(-2.0 * np.log(1.0 - u1)) * np.cos(2.0 * math.pi * u2)
    return torch.from_numpy((mean + std * z).astype(np.float32).reshape(shape))


def uniform(name, shape, seed=0, lo=0.0, hi=1.0):
    n = int(np.prod(shape)) if len(shape) else 1
    u = uniform01(name, n, seed, 0)
    return torch.from_numpy((lo + (hi - lo) * u).astype(np.float32).reshape(shape))


def synth_state_dict(reference_state_dict, seed=0):
    """Procedural values for every entry of a (reference-layout) state_dict.

    Only names, shapes and dtypes of ``reference_state_dict`` are used.
    """
    out = {}
    keys = list(reference_state_dict.keys())
    keyset = set(keys)
    for k in keys:
        ref = reference_state_dict[k]
        shape = tuple(ref.shape)
        stem, _, leaf = k.rpartition(".")
        is_bn = (stem + ".running_mean") in keyset
        if leaf == "num_batches_tracked":
            v = torch.zeros(shape, dtype=ref.dtype)
        elif leaf == "running_mean":
            v = normal(k, shape, seed, 0.0, 0.1)
        elif leaf == "running_var":
            v = uniform(k, shape, seed, 0.5, 1.5)
        elif is_bn and leaf == "weight":
            v = uniform(k, shape, seed, 0.5, 1.5)
        elif is_bn and leaf == "bias":
            v = normal(k, shape, seed, 0.0, 0.1)
        elif leaf == "weight":
            fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else shape[0]
            v = normal(k, shape, seed, 0.0, math.sqrt(2.0 / max(fan_in, 1)))
        elif leaf == "bias":
            v = normal(k, shape, seed, 0.0, 0.05)
        else:
            v = normal(k, shape, seed, 0.0, 0.1)
        out[k] = v.to(ref.dtype)
    return out


def calibrate_head(weight, bias, logits_mean, logits_std, target_mean=-3.0, target_std=1.0):
    """Rescale the final 32->1 conv so logits become ~N(target_mean, target_std).

    ``logits_mean/std`` are measured with the un-calibrated head.  Returns new
    (weight, bias).  logit' = (logit - b) * g + b' with g = target_std / std.
    """
    g = target_std / max(float(logits_std), 1e-12)
    new_w = weight * g
    new_b = (bias - logits_mean) * g + target_mean
    return new_w, new_b


def clip(batch, frames, height, width, seed=0):
    """Video clip as the reference's loaders hand it over: ``[B, T, 3, H, W]``
    ~ N(0, 1) (post-Normalize statistics, dataloader.py:246-249); callers apply
    ``permute((0, 2, 1, 3, 4))`` exactly as train.py:205 does."""
    return normal("clip", (batch, frames, 3, height, width), seed)


def gt_map(batch, height, width, seed=0):
    """Ground-truth saliency: normalised sum of three Gaussian blobs in [0, 1]."""
    u = uniform01("gt", batch * 9, seed).reshape(batch, 3, 3)
    ys = torch.arange(height, dtype=torch.float32).view(1, height, 1)
    xs = torch.arange(width, dtype=torch.float32).view(1, 1, width)
    out = torch.zeros(batch, height, width)
    for b in range(batch):
        for j in range(3):
            cy, cx, s = u[b, j]
            cy, cx = float(cy) * height, float(cx) * width
            sig = (0.04 + 0.10 * float(s)) * width
            out[b] += torch.exp(-((ys[0] - cy) ** 2 + (xs[0] - cx) ** 2) / (2 * sig * sig))
        out[b] /= out[b].max()
    return out


def audio(batch, length=70560, seed=0):
    """Waveform ``[B, 1, L, 1]``: Hanning-windowed N(0,1) * 2**-7 in the centre,
    zeros elsewhere (cf. dataloader.py:95-121)."""
    w = normal("audio", (batch, length), seed) * (2.0 ** -7)
    n = length // 2
    win = torch.hann_window(n, periodic=False)
    env = torch.zeros(length)
    start = (length - n) // 2
    env[start:start + n] = win
    return (w * env).view(batch, 1, length, 1)
