#!/usr/bin/env python3
"""CPU study behind the MusicEncoder's single-plane fp16 format (dc_music.hip, PL<false>; not collected by pytest).

Emulates the format on top of the oracle's arithmetic - BatchNorm folded into the convolution, folded weights rounded to fp16, every
layer's output rounded to fp16 once, fp32 accumulation, conv1.0 and conv4 exact as in the kernels - and reports
  (a) rel-L2 of the encoder's output against the oracle's fp32 MusicEncoder (transformer.py:313-340), and
  (b) rel-L2 of x0 after the oracle's DDIM-50 when the denoiser is conditioned on the emulated features instead of the exact ones.
Numbers on the seeded synthetic checkpoint, one clip of 1350 mel frames:  (a) 3.75e-4 (the GPU kernels measure 3.6 - 3.9e-4),
(b) 1.26e-4 - against the 5.0e-4 the fp16 denoiser itself sits at, i.e. 5.2e-4 combined in quadrature.
usage: python tests/study_encoder_fp16.py [mel frames, default 1350]"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffusion_conductor_amd.synthetic import synthetic_mel, synthetic_noise, synthetic_state_dict  # noqa: E402
from oracle import ddim_oracle as O  # noqa: E402

p = {k: torch.as_tensor(v) for k, v in synthetic_state_dict().items()}
Tm = int(sys.argv[1]) if len(sys.argv) > 1 else 1350
mel = torch.from_numpy(np.stack([synthetic_mel(0, Tm)])).float()
r16 = lambda x: x.half().float()


def fold(pre, bn):
    w, b = p[pre + ".weight"], p[pre + ".bias"]
    s = p[bn + ".weight"] / torch.sqrt(p[bn + ".running_var"] + 1e-5)
    return w * s.view(-1, *([1] * (w.dim() - 1))), (b - p[bn + ".running_mean"]) * s + p[bn + ".bias"]


def layer(prefix, x, residual, exact_weights=False):
    w, b = fold(prefix + ".conv2d_layer.0", prefix + ".conv2d_layer.1")
    y = F.relu(F.conv2d(F.pad(x, (1, 1, 1, 1), mode="reflect"), w if exact_weights else r16(w), b))
    if residual == "identity":
        y = y + x
    elif residual == "conv":
        rw, rb = fold(prefix + ".residual.0", prefix + ".residual.1")
        y = y + F.conv2d(x, r16(rw), rb)
    return r16(y)


def encoder_fp16(mel):
    pre = "music_encoder"
    x = layer(pre + ".conv1.0", mel.unsqueeze(1), "none", exact_weights=True)      # fp32 FMAs on the vector ALU
    x = layer(pre + ".conv1.1", x, "identity")
    x = layer(pre + ".conv1.2", x, "identity")
    x = F.max_pool2d(x, (5, 5), (1, 2), (2, 2))
    x = layer(pre + ".conv2.0", x, "conv")
    x = layer(pre + ".conv2.1", x, "identity")
    x = F.max_pool2d(x, (5, 5), (3, 2), (2, 2))
    x = layer(pre + ".conv3.0", x, "identity")
    x = layer(pre + ".conv3.1", x, "identity")
    x = F.max_pool2d(x, (3, 3), (1, 2), (1, 1))
    x = x.transpose(1, 2).flatten(start_dim=2).transpose(1, 2)
    w4, b4 = fold(pre + ".conv4.0", pre + ".conv4.1")
    return F.conv1d(x, w4, b4).transpose(1, 2)                                       # conv4: fp16 hi + lo weights, ~exact


rl2 = lambda a, b: float((a - b).norm() / b.norm())
with torch.no_grad():
    ref = O.music_encoder(p, mel)
    emu = encoder_fp16(mel)
    print(f"(a) encoder output, single fp16 plane vs oracle: rel-L2 {rl2(emu, ref):.3e}")
    T = (Tm - 1) // 3 + 1
    noise = torch.from_numpy(np.stack([synthetic_noise(0, T)])).float()
    x0 = {}
    for name, xo in (("exact", ref), ("fp16", emu)):
        x0[name] = O.ddim_sample_loop(p, noise, F.linear(xo, p["proj.weight"], p["proj.bias"]), xo, [T], 50)
    print(f"(b) x0 after DDIM-50, conditioned on the fp16-plane features vs the exact ones: rel-L2 {rl2(x0['fp16'], x0['exact']):.3e}")
