#!/usr/bin/env python3
"""GPU box: the MusicEncoder's two plane formats (DC_ME_PREC=split | f16) against the oracle - rel-L2 of xf_out / xf_proj on a few
shapes - and the time of encode_music for 32 clips of 60 s.  usage: python tools/encoder_formats.py"""
import os
import sys
import time

import torch

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests"))
from helpers import batch_mel, make_model, oracle_params, rel_l2  # noqa: E402
from oracle import ddim_oracle as O  # noqa: E402  (checker only)

m = make_model("fp16")
p = oracle_params()
for B, Tm in ((1, 270), (2, 271), (2, 2700), (1, 5400)):
    mel = torch.from_numpy(batch_mel(B, Tm))
    with torch.no_grad():
        rxp, rx = O.encode_music(p, mel)
    for fmt in ("split", "f16"):
        os.environ["DC_ME_PREC"] = fmt
        xp, x = m.encode_music(mel.cuda(), "cuda:0")
        torch.cuda.synchronize()
        print(f"B={B} Tm={Tm} {fmt:5s}: rel-L2 x {rel_l2(x, rx):.3e} x_proj {rel_l2(xp, rxp):.3e}")
mel = torch.from_numpy(batch_mel(32, 5400)).cuda()
for fmt in ("split", "f16", "split", "f16"):
    os.environ["DC_ME_PREC"] = fmt
    ts = []
    for i in range(6):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        m.encode_music(mel, "cuda:0")
        torch.cuda.synchronize()
        ts.append(1e3 * (time.perf_counter() - t0))
    print(f"encode_music 32 x 60 s, {fmt}: " + " ".join(f"{t:.2f}" for t in ts) + " ms")
