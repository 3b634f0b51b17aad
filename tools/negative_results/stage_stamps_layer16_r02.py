#!/usr/bin/env python3
"""(round 2, rejected 16-waves-per-unit kernel - not the shipped k_layer16: see tools/stage_stamps16.py)  Diagnostic: per-stage time (us) of the 16 waves of one k_layer16 workgroup (layer 3), from s_memrealtime stamps.
Run on the GPU box:  DC_LAYER16=1 DC_STAMPS=1 DC_DISABLE_GRAPH=1 python tools/stage_stamps16.py"""
import os, sys
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from helpers import make_model, make_diffusion, xf_pair, batch_noise
B, T = 32, 1800
m = make_model("fp16")
xfp, xfo = xf_pair(B, T); noise = torch.from_numpy(batch_noise(B, T)).cuda()
nat = m.set_conditioning(xfp.cuda(), xfo.cuda(), [T] * B)
gd = make_diffusion(50)
for _ in range(2):
    nat.ddim_loop(noise, gd.native_coefficients())
torch.cuda.synchronize()
st = nat.debug_read("stamps", np.uint64, 264 + 16 * 32).astype(np.int64)[264:].reshape(16, 32)
names = ["load h, combine, sync", "Q + attend (SA)", "sync", "stylize (SA)", "sync", "Q + attend (CA)", "sync", "stylize (CA)", "sync", "FFN", "sync",
         "stylize (FFN)", "sync", "LN + K + keys", "wait image + barrier", "factors", "V + P round 0", "barrier", "reduce 0", "barrier + V + P 1",
         "barrier", "reduce 1"]
print("wave:                  " + "".join(f"{w:6d}" for w in range(16)))
for k in range(1, 23):
    print(f"{k:2d} {names[k-1][:20]:20s}" + "".join(f"{(st[w, k] - st[w, k-1]) / 100.0:6.2f}" for w in range(16)))
print("total (us)             " + "".join(f"{(st[w, 22] - st[w, 0]) / 100.0:6.1f}" for w in range(16)))
