"""Seeded synthetic checkpoint and inputs (numpy only).

No trained checkpoint ships with the reference (README.md:125 points at external
downloads), and a freshly constructed reference model outputs exactly 0 because of
``zero_module`` (Diffusion_Stage/models/transformer.py:44-50,65,165,443).  Parity
tests, ``smoke()`` and ``bench.py`` therefore regenerate the same checkpoint and
inputs from seeds on whichever box they run, instead of committing 24 MB of weights.

Each tensor gets its own counter-based stream keyed by (seed, crc32(name)), so the
values do not depend on generation order.
"""
from __future__ import annotations

import zlib
from collections import OrderedDict

import numpy as np

from .param_spec import DenoiserConfig, param_shapes


def _rng(seed: int, tag: str, index: int = 0) -> np.random.Generator:
    key = [int(seed) & 0xFFFFFFFF, zlib.crc32(tag.encode()) & 0xFFFFFFFF]
    ctr = [int(index) & 0xFFFFFFFFFFFFFFFF, 0, 0, 0]
    return np.random.Generator(np.random.Philox(key=key, counter=ctr))


def synthetic_state_dict(cfg: DenoiserConfig = DenoiserConfig(), seed: int = 0):
    """name -> np.ndarray for every state_dict entry of the denoiser.

    Linear/conv weights and biases ~ U(-1/sqrt(fan_in), 1/sqrt(fan_in)) (the bound
    PyTorch's default initialiser uses), *including* the tensors the reference
    zero-initialises - otherwise every Stylization block, FFN and the output
    projection contribute nothing and parity would be vacuous.  Norm affine and
    BatchNorm running statistics are non-trivial.
    """
    out = OrderedDict()
    for name, shape in param_shapes(cfg).items():
        g = _rng(seed, name)
        if name.endswith("num_batches_tracked"):
            out[name] = np.asarray(1000, dtype=np.int64)
            continue
        leaf = name.rsplit(".", 1)[-1]
        if name == "sequence_embedding":
            a = g.standard_normal(shape)
        elif leaf == "running_mean":
            a = 0.1 * g.standard_normal(shape)
        elif leaf == "running_var":
            a = g.uniform(0.5, 1.5, shape)
        elif len(shape) == 1 and (".norm." in name or "text_norm" in name
                                  or "conv2d_layer.1." in name or "residual.1." in name
                                  or "conv4.1." in name):
            # LayerNorm / BatchNorm affine
            a = (1.0 + 0.1 * g.standard_normal(shape)) if leaf == "weight" \
                else 0.1 * g.standard_normal(shape)
        else:
            if leaf == "weight":
                fan_in = int(np.prod(shape[1:]))
            else:  # bias: fan_in of the sibling weight
                wshape = param_shapes(cfg)[name[:-4] + "weight"]
                fan_in = int(np.prod(wshape[1:]))
            bound = 1.0 / np.sqrt(fan_in)
            a = g.uniform(-bound, bound, shape)
        out[name] = np.ascontiguousarray(a, dtype=np.float32)
    return out


def synthetic_mel(clip: int, n_frames: int = 5400, n_bins: int = 128, seed: int = 1):
    """U[0,1) mel for clip index ``clip`` (real mels are normalised to [0,1]:
    Diffusion_Stage/tools/visualization.py:165)."""
    return _rng(seed, "mel", clip).random((n_frames, n_bins), dtype=np.float32)


def synthetic_noise(clip: int, T: int = 1800, P: int = 26, seed: int = 2):
    """x_T ~ N(0,1) for clip index ``clip``."""
    return _rng(seed, "x_T", clip).standard_normal((T, P), dtype=np.float32)


def synthetic_music_features(clip: int, T: int = 1800, C: int = 64, seed: int = 3):
    """Stand-in for MusicEncoder output [T,64] (used where encode_music is not under
    test): roughly unit-scale, like a BatchNorm'ed conv output."""
    return _rng(seed, "xf", clip).standard_normal((T, C), dtype=np.float32)


def batch_mel(B, n_frames=5400, n_bins=128, seed=1, first=0):
    return np.stack([synthetic_mel(first + b, n_frames, n_bins, seed) for b in range(B)])


def batch_noise(B, T=1800, P=26, seed=2, first=0):
    return np.stack([synthetic_noise(first + b, T, P, seed) for b in range(B)])


def batch_music_features(B, T=1800, C=64, seed=3, first=0):
    return np.stack([synthetic_music_features(first + b, T, C, seed) for b in range(B)])


def stress_state_dict(cfg: DenoiserConfig = DenoiserConfig(), seed: int = 0):
    """A "trained-like" stress variant of the synthetic checkpoint: trained denoisers are not at initialisation
    scale, so the parity tests also run on a draw with larger modulation / output weights, spread-out
    LayerNorm gains and a few outlier channels (tests/golden/g8_robust.npz):
      * StylizationBlock ``emb_layers`` / ``out_layers``, ``ffn.linear2`` and ``out`` weights x 3;
      * every LayerNorm gain log-normal with sigma = 0.5;
      * four 10x outlier channels in ``sequence_embedding`` and in ``joint_embed``.
    """
    sd = synthetic_state_dict(cfg, seed)
    for name in sd:
        leaf = name.rsplit(".", 1)[-1]
        if leaf == "weight" and (".emb_layers.1." in name or ".out_layers.2." in name or name.endswith("ffn.linear2.weight")
                                 or name == "out.weight"):
            sd[name] = np.ascontiguousarray(sd[name] * 3.0, dtype=np.float32)
        elif leaf == "weight" and len(sd[name].shape) == 1 and (".norm." in name or "text_norm" in name):
            g = _rng(seed, "stress:" + name)
            sd[name] = np.exp(0.5 * g.standard_normal(sd[name].shape)).astype(np.float32)
    ch = _rng(seed, "stress:outliers").choice(cfg.latent_dim, size=8, replace=False)
    se = sd["sequence_embedding"].copy()
    se[:, ch[:4]] *= 10.0
    sd["sequence_embedding"] = se
    je = sd["joint_embed.weight"].copy()
    je[ch[4:], :] *= 10.0
    sd["joint_embed.weight"] = je
    return sd


def smooth_mel(clip: int, n_frames: int = 5400, n_bins: int = 128, seed: int = 1):
    """A mel in [0,1] with the smoothness of a real spectrogram (white noise low-pass filtered along time and
    frequency, then min-max normalised as tools/visualization.py:165 does) - ``synthetic_mel`` is white."""
    a = _rng(seed, "smooth_mel", clip).standard_normal((n_frames + 64, n_bins + 16))
    kt = np.hanning(65)
    kf = np.hanning(17)
    a = np.apply_along_axis(lambda v: np.convolve(v, kt / kt.sum(), mode="valid"), 0, a)
    a = np.apply_along_axis(lambda v: np.convolve(v, kf / kf.sum(), mode="valid"), 1, a)
    a = a[:n_frames, :n_bins]
    a = (a - a.min()) / (a.max() - a.min())
    return np.ascontiguousarray(a, dtype=np.float32)


def batch_step_noise(S, B, T=1800, P=26, seed=4, first=0):
    """Per-iteration DDIM noise z_i ~ N(0,1) (eta > 0), [S, B, T, P]: iteration i of clip b has its own stream."""
    return np.stack([np.stack([_rng(seed, f"z{i}", first + b).standard_normal((T, P), dtype=np.float32) for b in range(B)])
                     for i in range(S)])
